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
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define WAVE 64

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
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float x) { return (bf16_t)x; }

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

// ---------------------------------------------------------------- dropout
// Counter-based keep decision: a pure function of (seed, element index), so the
// backward pass regenerates the mask instead of storing it.
__device__ __forceinline__ uint32_t vmr_hash(uint32_t seed, uint64_t idx) {
  uint32_t h = (uint32_t)idx * 0x9E3779B1u + (uint32_t)(idx >> 32) * 0x85EBCA77u;
  h ^= seed;
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__host__ __device__ __forceinline__ uint32_t vmr_drop_thresh(float p) {
  double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}
// effective seed = site seed mixed with an optional device-resident step counter,
// so a captured hipGraph replays with fresh masks every step.
__device__ __forceinline__ uint32_t vmr_seed(uint32_t site_seed, const uint32_t* step) {
  return step ? site_seed ^ (step[0] * 0x9E3779B9u + 0x7F4A7C15u) : site_seed;
}
__device__ __forceinline__ bool vmr_keep(uint32_t seed, uint64_t idx, uint32_t thresh) {
  return vmr_hash(seed, idx) >= thresh;
}

// ------------------------------------------------------------- reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
