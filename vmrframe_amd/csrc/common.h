// common.h -- shared device/host helpers for libvmr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vmr_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16_t;   // IEEE half: the second 16-bit activation type (VMR_F16; BASELINE configs[4] runs BAN in fp16)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define WAVE 64

// ------------------------------------------------- LDS reads under LDS-DMA
// ds_read_b64_tr_b16 as inline asm, NOT __builtin_amdgcn_ds_read_tr16_b64_*: hipcc (ROCm 7.2) treats the builtin as an
// LDS access that may alias an LDS-DMA write in flight and puts "s_waitcnt vmcnt(0)" in front of the first one after any
// global_load_lds -- draining the very stage(s) the ring had just requested (a plain ds_read_b128 does not get that wait).
// The asm read is invisible to the compiler's lgkmcnt bookkeeping: every user waits itself (lgkm_wait<N>) and then
// pins the fragment registers (frag_pin) so the MFMAs cannot be scheduled above the wait.
__device__ __forceinline__ s16x4 lds_read_tr(const unsigned char* p) {
  typedef __attribute__((address_space(3))) const unsigned char lds_u8;
  const uint32_t addr = (uint32_t)(uintptr_t)(lds_u8*)p;
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
template <int N> __device__ __forceinline__ void lgkm_wait() {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit field");
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void frag_pin(bf16x8& f) { asm volatile("" : "+v"(f)); }
__device__ __forceinline__ bf16x8 lds_read_b128_asm(const unsigned char* p) {   // same contract, for loops that mix both kinds
  typedef __attribute__((address_space(3))) const unsigned char lds_u8;
  const uint32_t addr = (uint32_t)(uintptr_t)(lds_u8*)p;
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}


// ---------------------------------------------------------------- errors
extern thread_local char g_vmr_err[256];
int vmr_fail(int code, const char* fmt, ...);
#define VMR_CHECK(cond, ...)                        \
  do {                                              \
    if (!(cond)) return vmr_fail(-22, __VA_ARGS__); \
  } while (0)
#define VMR_LAUNCH_CHECK()                                                     \
  do {                                                                         \
    hipError_t e_ = hipGetLastError();                                         \
    if (e_ != hipSuccess) return vmr_fail(-5, "launch: %s", hipGetErrorString(e_)); \
  } while (0)

// ------------------------------------------------------------ conversions
__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }  // v_cvt_pk_bf16_f32 (RNE, NaN safe)

template <typename T> __device__ __forceinline__ float to_f(T x);
template <> __device__ __forceinline__ float to_f<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t x) { return (float)x; }
template <> __device__ __forceinline__ float to_f<f16_t>(f16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float x) { return (bf16_t)x; }
template <> __device__ __forceinline__ f16_t from_f<f16_t>(float x) { return (f16_t)x; }   // v_cvt_f16_f32 (RNE; |x| > 65504 -> inf)

// The 16-bit element types share every data path (LDS images, DMA, fragment registers): kernels carry fragments as raw
// bf16x8 BIT containers and name the element type only where bits meet arithmetic -- the MFMA and the float conversions.
template <typename E> struct is_f16 { static constexpr bool value = false; };
template <> struct is_f16<f16_t> { static constexpr bool value = true; };
template <typename E> __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {   // 16x16x32, fp32 accumulate
  if constexpr (is_f16<E>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// low / high 16-bit element of a packed pair, as float
template <typename E> __device__ __forceinline__ float e16_lo(uint32_t w) {
  if constexpr (is_f16<E>::value) return (float)__builtin_bit_cast(f16_t, (uint16_t)(w & 0xFFFFu));
  else return __uint_as_float(w << 16);
}
template <typename E> __device__ __forceinline__ float e16_hi(uint32_t w) {
  if constexpr (is_f16<E>::value) return (float)__builtin_bit_cast(f16_t, (uint16_t)(w >> 16));
  else return __uint_as_float(w & 0xFFFF0000u);
}
// one element of a raw fragment <-> float
template <typename E> __device__ __forceinline__ float frag_get(const bf16x8& f, int i) {
  if constexpr (is_f16<E>::value) return (float)__builtin_bit_cast(f16x8, f)[i];
  else return (float)f[i];
}
template <typename E> __device__ __forceinline__ bf16_t bits_from_f(float x) {   // the element's bits, carried as bf16_t
  if constexpr (is_f16<E>::value) return __builtin_bit_cast(bf16_t, (f16_t)x);
  else return (bf16_t)x;
}
template <typename E> __device__ __forceinline__ float bits_to_f(bf16_t b) {
  if constexpr (is_f16<E>::value) return (float)__builtin_bit_cast(f16_t, b);
  else return (float)b;
}

// 8-element vector load/store of T as floats (16 B for bf16, 2x16 B for f32)
template <typename T> struct Vec8;
template <> struct Vec8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    bf16x8 x = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)x[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    bf16x8 x;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = x;
  }
};
template <> struct Vec8<f16_t> {
  static __device__ __forceinline__ void load(const f16_t* p, float (&v)[8]) {
    f16x8 x = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)x[i];
  }
  static __device__ __forceinline__ void store(f16_t* p, const float (&v)[8]) {
    f16x8 x;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (f16_t)v[i];
    *reinterpret_cast<f16x8*>(p) = x;
  }
};
template <> struct Vec8<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
  }
};

// 4-element vector load/store of T as floats (8 B for bf16, 16 B for f32): hipcc does not merge
// scalar 2-byte accesses by itself
template <typename T> struct Vec4;
template <> struct Vec4<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
    bf16x4 x = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)x[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
    bf16x4 x;
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x4*>(p) = x;
  }
};
template <> struct Vec4<f16_t> {
  static __device__ __forceinline__ void load(const f16_t* p, float (&v)[4]) {
    f16x4 x = *reinterpret_cast<const f16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)x[i];
  }
  static __device__ __forceinline__ void store(f16_t* p, const float (&v)[4]) {
    f16x4 x;
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = (f16_t)v[i];
    *reinterpret_cast<f16x4*>(p) = x;
  }
};
template <> struct Vec4<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[i];
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    f32x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = v[i];
    *reinterpret_cast<f32x4*>(p) = a;
  }
};

template <typename T> struct Vec2;
template <> struct Vec2<bf16_t> {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[2]) {
    bf16x2 x = *reinterpret_cast<const bf16x2*>(p);
    v[0] = (float)x[0]; v[1] = (float)x[1];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[2]) {
    bf16x2 x; x[0] = (bf16_t)v[0]; x[1] = (bf16_t)v[1];
    *reinterpret_cast<bf16x2*>(p) = x;
  }
};
template <> struct Vec2<f16_t> {
  typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
  static __device__ __forceinline__ void load(const f16_t* p, float (&v)[2]) {
    f16x2 x = *reinterpret_cast<const f16x2*>(p);
    v[0] = (float)x[0]; v[1] = (float)x[1];
  }
  static __device__ __forceinline__ void store(f16_t* p, const float (&v)[2]) {
    f16x2 x; x[0] = (f16_t)v[0]; x[1] = (f16_t)v[1];
    *reinterpret_cast<f16x2*>(p) = x;
  }
};
template <> struct Vec2<float> {
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  static __device__ __forceinline__ void load(const float* p, float (&v)[2]) {
    f32x2 x = *reinterpret_cast<const f32x2*>(p);
    v[0] = x[0]; v[1] = x[1];
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[2]) {
    f32x2 x; x[0] = v[0]; x[1] = v[1];
    *reinterpret_cast<f32x2*>(p) = x;
  }
};

// ---------------------------------------------------------------- dropout
// Counter-based keep decision: a pure function of (seed, element index), so the
// backward pass regenerates the mask instead of storing it.  One strong 2x32-bit
// hash serves a group of 4 consecutive elements (16 random bits each), which is
// what the 8-wide vector paths amortise; the drop probability is therefore
// quantised to 1/65536 (0.2 -> 13107/65536) and the inverted-dropout scale uses
// the NOMINAL p, an expectation error below 2e-5.
__device__ __forceinline__ uint32_t vmr_mix(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
// 64 random bits for the group of 4 elements with index g = idx >> 2
__device__ __forceinline__ uint2 vmr_hash4(uint32_t seed, uint64_t g) {
  const uint32_t lo = (uint32_t)g, hi = (uint32_t)(g >> 32);
  const uint32_t a = vmr_mix(lo * 0x9E3779B1u + hi * 0x85EBCA77u + seed);
  const uint32_t b = vmr_mix(a ^ 0x68E31DA4u) ;
  return make_uint2(a, b);
}
__host__ __device__ __forceinline__ uint32_t vmr_drop_thresh(float p) {   // 16-bit threshold
  const float t = p * 65536.0f;
  return t >= 65535.0f ? 65535u : (uint32_t)t;
}
// effective seed = site seed mixed with an optional device-resident step counter,
// so a captured hipGraph replays with fresh masks every step.
__device__ __forceinline__ uint32_t vmr_seed(uint32_t site_seed, const uint32_t* step) {
  return step ? site_seed ^ (step[0] * 0x9E3779B9u + 0x7F4A7C15u) : site_seed;
}
__device__ __forceinline__ bool vmr_keep(uint32_t seed, uint64_t idx, uint32_t thresh) {
  const uint2 h = vmr_hash4(seed, idx >> 2);
  const uint32_t w = (idx & 2) ? h.y : h.x;
  return ((idx & 1) ? (w >> 16) : (w & 0xFFFFu)) >= thresh;
}
// keep bits of 8 consecutive elements starting at idx8 (a multiple of 8): bit e <=> keep(idx8 + e)
__device__ __forceinline__ uint32_t vmr_keep8(uint32_t seed, uint64_t idx8, uint32_t thresh) {
  const uint2 h0 = vmr_hash4(seed, idx8 >> 2), h1 = vmr_hash4(seed, (idx8 >> 2) + 1);
  uint32_t m = 0;
  m |= ((h0.x & 0xFFFFu) >= thresh) << 0; m |= ((h0.x >> 16) >= thresh) << 1;
  m |= ((h0.y & 0xFFFFu) >= thresh) << 2; m |= ((h0.y >> 16) >= thresh) << 3;
  m |= ((h1.x & 0xFFFFu) >= thresh) << 4; m |= ((h1.x >> 16) >= thresh) << 5;
  m |= ((h1.y & 0xFFFFu) >= thresh) << 6; m |= ((h1.y >> 16) >= thresh) << 7;
  return m;
}

// ------------------------------------------------------------- reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// The same sum on the DPP path: four in-row steps (quad_perm x 2, row_half_mirror, row_mirror: every lane of a 16-lane
// row then holds its row's sum) and the four rows met through SGPRs (v_readlane) -- about 10 short-latency VALU ops
// instead of six dependent ds_bpermute round trips through the LDS crossbar (~100 cycles each; __shfl_xor compiles to
// ds_bpermute_b32 for every distance).  ALL 64 lanes must be active.  The association differs from wave_sum's
// butterfly, so the two are not bit-interchangeable: the LayerNorm family (norm.hip, convblock.hip) uses this one
// throughout.  The result is wave-uniform (it lives in SGPRs).
template <int CTRL> __device__ __forceinline__ float dpp_get(float v) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_get<0xB1>(v);    // quad_perm:[1,0,3,2]
  v += dpp_get<0x4E>(v);    // quad_perm:[2,3,0,1]
  v += dpp_get<0x141>(v);   // row_half_mirror
  v += dpp_get<0x140>(v);   // row_mirror
  const uint32_t u = __float_as_uint(v);
  const float r0 = __uint_as_float(__builtin_amdgcn_readlane(u, 0)), r1 = __uint_as_float(__builtin_amdgcn_readlane(u, 16));
  const float r2 = __uint_as_float(__builtin_amdgcn_readlane(u, 32)), r3 = __uint_as_float(__builtin_amdgcn_readlane(u, 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// dtype codes: VMR_F32 = 0, VMR_BF16 = 1, VMR_F16 = 2
static inline bool vmr_dtype_ok(int dt) { return dt == VMR_F32 || dt == VMR_BF16 || dt == VMR_F16; }
static inline bool vmr_dtype_16(int dt) { return dt == VMR_BF16 || dt == VMR_F16; }
static inline int vmr_dtype_size(int dt) { return dt == VMR_F32 ? 4 : 2; }
// runs `...` with `T` bound to the element type of `dtype` (the caller has checked vmr_dtype_ok)
#define VMR_DISPATCH(dtype, T, ...)                                  \
  do {                                                               \
    if ((dtype) == VMR_BF16) { typedef bf16_t T; __VA_ARGS__; }      \
    else if ((dtype) == VMR_F16) { typedef f16_t T; __VA_ARGS__; }   \
    else { typedef float T; __VA_ARGS__; }                           \
  } while (0)
#define VMR_DISPATCH16(dtype, T, ...)                                \
  do {                                                               \
    if ((dtype) == VMR_F16) { typedef f16_t T; __VA_ARGS__; }        \
    else { typedef bf16_t T; __VA_ARGS__; }                          \
  } while (0)
