// gemm.hip -- MFMA GEMM with fused epilogue for gfx950 (MI355X).
//
// One kernel family serves every dense contraction of the SeqPAN path: the
// pointwise Conv1D layers (reference models/layers.py:15-26), the
// MultiheadAttention projections (layers.py:570) and, batched over (b,h) or
// (t,h), the QK^T / P.V products of the dual and batch-axis attentions
// (layers.py:349-366, 567-574), plus all backward products (dX = dY.W,
// dW = dY^T.X) through the transA/transB forms.
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves, 64x64 each =
// 4x4 MFMA 16x16 tiles), BK=64 (bf16, v_mfma_f32_16x16x32_bf16) or BK=16 (f32,
// v_mfma_f32_16x16x4_f32).  Global->register->LDS staging, two LDS stages, one
// barrier per K-step with the next tile's global loads issued before the MFMAs
// (write-after-barrier).  K-contiguous operands live in LDS as 128-B rows with a
// 16-B-chunk XOR swizzle (conflict-free ds_read_b128); M/N-contiguous operands
// (the transposed forms) live as [k][128] rows with a 32-B-chunk XOR swizzle
// and are read with ds_read_b64_tr_b16, so no operand is ever transposed in HBM.
// The epilogue goes through LDS so bias/ReLU/dropout/residual/aux traffic is
// 16-B coalesced.  blockIdx -> tile mapping is XCD-aware (tiles that share an A
// row-panel run on one XCD / one L2).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int CST_LD = 132;                    // fp32 staging row stride (floats)
constexpr int SMEM_BYTES = BM * CST_LD * 4;    // 67,584 B (>= 2 stages of A+B)

struct TileCoord {
  int tm, tn, zb, ks;
};

__device__ __forceinline__ TileCoord tile_coord(const vmr_gemm_t& g, int tiles_m, int tiles_n) {
  // XCD-aware bijective remap: blocks with equal blockIdx.x % 8 share an XCD/L2;
  // give each XCD a contiguous run of logical tiles (n fastest), so the tiles that
  // re-read one A row-panel hit the same L2.
  const int nblk = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  TileCoord t;
  t.tm = logical / tiles_n;
  t.tn = logical - t.tm * tiles_n;
  const int sk = g.splitk > 1 ? g.splitk : 1;
  t.zb = blockIdx.z / sk;
  t.ks = blockIdx.z - t.zb * sk;
  return t;
}

// swizzle of the [k][128] (M/N-contiguous) bf16 image: 32-B chunk index ^= f(k)
__device__ __forceinline__ int swz_tr(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

// ------------------------------------------------------------------ epilogue
template <typename T, bool ALIGNED>
__device__ __forceinline__ void epilogue(const vmr_gemm_t& g, float* cst, int m0, int n0, int zb,
                                         T* __restrict__ C, const T* __restrict__ Rsd, T* __restrict__ Aux) {
  const int tid = threadIdx.x;
  const int flags = g.flags;
  if (flags & VMR_EPI_ACCUM) {
    float* Cf = reinterpret_cast<float*>(C);
    // 256 contiguous bytes per wave-instruction: the fast shape for float atomics
    for (int pass = 0; pass < 32; ++pass) {
      const int row = pass * 4 + (tid >> 6);
      const int gm = m0 + row;
      if (gm >= g.M) continue;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int col = (tid & 63) + 64 * h;
        const int gn = n0 + col;
        if (gn < g.N) atomicAdd(&Cf[(int64_t)gm * g.ldc + gn], cst[row * CST_LD + col] * g.alpha);
      }
    }
    return;
  }
  const uint32_t thresh = vmr_drop_thresh(g.drop_p);
  const float dscale = (flags & VMR_EPI_DROPOUT) ? 1.0f / (1.0f - g.drop_p) : 1.0f;
  const uint32_t seed = vmr_seed(g.drop_seed, g.drop_step);
  const bool out_f32 = (flags & VMR_EPI_OUT_F32) != 0;
  for (int pass = 0; pass < 8; ++pass) {
    const int row = pass * 16 + (tid >> 4);
    const int col = (tid & 15) * 8;
    const int gm = m0 + row, gn = n0 + col;
    if (gm >= g.M || gn >= g.N) continue;
    float v[8];
    {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&cst[row * CST_LD + col]);
      const f32x4 b = *reinterpret_cast<const f32x4*>(&cst[row * CST_LD + col + 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    }
    const int nvalid = min(8, g.N - gn);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = v[e] * g.alpha;
      if ((flags & VMR_EPI_BIAS) && e < nvalid) x += g.bias[gn + e];
      if (flags & VMR_EPI_RELU) x = fmaxf(x, 0.0f);
      if (flags & VMR_EPI_DROPOUT) {
        const uint64_t idx = ((uint64_t)zb * g.M + gm) * (uint64_t)g.N + (uint64_t)(gn + e);
        x = vmr_keep(seed, idx, thresh) ? x * dscale : 0.0f;
      }
      v[e] = x;
    }
    const bool vec = ALIGNED && nvalid == 8;
    if (flags & VMR_EPI_AUX) {
      T* ap = Aux + (int64_t)gm * g.ldr + gn;
      if (vec) Vec8<T>::store(ap, v);
      else for (int e = 0; e < nvalid; ++e) ap[e] = from_f<T>(v[e]);
    }
    if (flags & VMR_EPI_RESIDUAL) {
      const T* rp = Rsd + (int64_t)gm * g.ldr + gn;
      if (vec) {
        float rv[8];
        Vec8<T>::load(rp, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      } else {
        for (int e = 0; e < nvalid; ++e) v[e] += to_f<T>(rp[e]);
      }
    }
    if (flags & VMR_EPI_ROWSCALE) {
      const float rs = g.rowscale[gm];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= rs;
    }
    if (out_f32) {
      float* cp = reinterpret_cast<float*>(C) + (int64_t)gm * g.ldc + gn;
      if (vec) Vec8<float>::store(cp, v);
      else for (int e = 0; e < nvalid; ++e) cp[e] = v[e];
    } else {
      T* cp = C + (int64_t)gm * g.ldc + gn;
      if (vec) Vec8<T>::store(cp, v);
      else for (int e = 0; e < nvalid; ++e) cp[e] = from_f<T>(v[e]);
    }
  }
}

__device__ __forceinline__ void stage_acc(float* cst, const f32x4 (&acc)[4][4], int wm, int wn, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cst[(wm * 64 + i * 16 + (lane >> 4) * 4 + r) * CST_LD + wn * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
}

// ---------------------------------------------------------------- bf16 kernel
constexpr int BK16 = 64;                      // bf16 K-step
constexpr int OP_BYTES16 = 128 * BK16 * 2;    // 16 KiB per operand per stage

// Load one operand tile (128 x 64 bf16) into 4 x bf16x8 registers per thread.
//   KC (k-contiguous):  tile rows = 128 (m or n), 8 chunks of 8 k each
//   !KC (m/n-contiguous): tile rows = 64 (k), 16 chunks of 8 m each
template <bool KC, bool ALIGNED>
__device__ __forceinline__ void load_operand(const bf16_t* __restrict__ P, int64_t ld, int r0, int R,
                                             int k0, int k_end, bf16x8 (&reg)[4]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + i * 256;
    int row, kk;  // element (index in the m/n space, k index)
    const bf16_t* p;
    bool full, any;
    if (KC) {
      row = r0 + (id >> 3); kk = k0 + (id & 7) * 8;
      p = P + (int64_t)row * ld + kk;
      any = row < R && kk < k_end;
      full = row < R && kk + 8 <= k_end;
    } else {
      kk = k0 + (id >> 4); row = r0 + (id & 15) * 8;
      p = P + (int64_t)kk * ld + row;
      any = kk < k_end && row < R;
      full = kk < k_end && row + 8 <= R;
    }
    bf16x8 v;
    if (ALIGNED && full) v = *reinterpret_cast<const bf16x8*>(p);
    else if (any) {  // unaligned operand or a ragged K / M / N tail: guarded element loads
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool ok = KC ? (kk + e < k_end) : (row + e < R);
        v[e] = ok ? p[e] : (bf16_t)0.0f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.0f;
    }
    reg[i] = v;
  }
}

template <bool KC>
__device__ __forceinline__ void store_operand(unsigned char* lds, const bf16x8 (&reg)[4]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + i * 256;
    int off;
    if (KC) {
      const int row = id >> 3, c = id & 7;
      off = row * 128 + ((c ^ (row & 7)) << 4);
    } else {
      const int row = id >> 4, c16 = id & 15;
      off = row * 256 + ((((c16 >> 1) ^ swz_tr(row))) << 5) + ((c16 & 1) << 4);
    }
    *reinterpret_cast<bf16x8*>(lds + off) = reg[i];
  }
}

// fragment for the 16-row (or 16-col) MFMA tile `t16` (0..7 within the 128 tile), k-substep kk (0..1)
template <bool KC>
__device__ __forceinline__ bf16x8 read_frag(const unsigned char* lds, int t16, int kk, int lane) {
  if (KC) {
    const int row = t16 * 16 + (lane & 15);
    const int chunk = kk * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((chunk ^ (row & 7)) << 4));
  } else {
    const int g = lane >> 4, ii = lane & 15, q = ii >> 2, p = ii & 3;
    const int r = kk * 32 + 8 * g + q;
    const int a0 = r * 256 + ((t16 ^ swz_tr(r)) << 5) + p * 8;
    const int a1 = (r + 4) * 256 + ((t16 ^ swz_tr(r + 4)) << 5) + p * 8;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + a1));
    union { struct { s16x4 l, h; } s; bf16x8 v; } u;
    u.s.l = lo; u.s.h = hi;
    return u.v;
  }
}

template <bool TA, bool TB, bool ALIGNED>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(vmr_gemm_t g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const TileCoord tc = tile_coord(g, tiles_m, tiles_n);
  const int m0 = tc.tm * BM, n0 = tc.tn * BN;
  const int z1 = tc.zb / g.Z2, z2 = tc.zb - z1 * g.Z2;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A) + z1 * g.sA1 + z2 * g.sA2;
  const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B) + z1 * g.sB1 + z2 * g.sB2;
  const int64_t coff = z1 * g.sC1 + z2 * g.sC2;

  int k_begin = 0, k_end = g.K;
  if (g.splitk > 1) {
    int chunk = (g.K + g.splitk - 1) / g.splitk;
    chunk = (chunk + BK16 - 1) / BK16 * BK16;
    k_begin = tc.ks * chunk;
    k_end = min(g.K, k_begin + chunk);
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = k_end > k_begin ? (k_end - k_begin + BK16 - 1) / BK16 : 0;
  bf16x8 ra[4], rb[4];
  if (nk > 0) {
    load_operand<!TA, ALIGNED>(A, g.lda, m0, g.M, k_begin, k_end, ra);
    load_operand<!TB, ALIGNED>(B, g.ldb, n0, g.N, k_begin, k_end, rb);
    store_operand<!TA>(smem, ra);
    store_operand<!TB>(smem + OP_BYTES16, rb);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    unsigned char* cur = smem + (kt & 1) * 2 * OP_BYTES16;
    unsigned char* nxt = smem + ((kt + 1) & 1) * 2 * OP_BYTES16;
    const bool more = kt + 1 < nk;
    if (more) {  // issue the next tile's global loads before the MFMAs (latency hides under compute)
      const int k0 = k_begin + (kt + 1) * BK16;
      load_operand<!TA, ALIGNED>(A, g.lda, m0, g.M, k0, k_end, ra);
      load_operand<!TB, ALIGNED>(B, g.ldb, n0, g.N, k0, k_end, rb);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = read_frag<!TA>(cur, wm * 4 + i, kk, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag<!TB>(cur + OP_BYTES16, wn * 4 + j, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      store_operand<!TA>(nxt, ra);
      store_operand<!TB>(nxt + OP_BYTES16, rb);
    }
    __syncthreads();
  }
  float* cst = reinterpret_cast<float*>(smem);
  stage_acc(cst, acc, wm, wn, lane);
  __syncthreads();
  epilogue<bf16_t, ALIGNED>(g, cst, m0, n0, tc.zb,
                            (g.flags & (VMR_EPI_OUT_F32 | VMR_EPI_ACCUM))
                                ? reinterpret_cast<bf16_t*>(reinterpret_cast<float*>(g.C) + coff)
                                : reinterpret_cast<bf16_t*>(g.C) + coff,
                            reinterpret_cast<const bf16_t*>(g.residual) + coff,
                            reinterpret_cast<bf16_t*>(g.aux) + coff);
}

// ----------------------------------------------------------------- f32 kernel
constexpr int BK32 = 16;
constexpr int OP_FLOATS32 = 2304;  // max(128*17, 16*144) floats per operand per stage
constexpr int LD_KC32 = 17, LD_TR32 = 144;

template <bool KC, bool ALIGNED>
__device__ __forceinline__ void load_operand32(const float* __restrict__ P, int64_t ld, int r0, int R,
                                               int k0, int k_end, f32x4 (&reg)[2]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + i * 256;
    int row, kk;
    const float* p;
    if (KC) { row = r0 + (id >> 2); kk = k0 + (id & 3) * 4; p = P + (int64_t)row * ld + kk; }
    else { kk = k0 + (id >> 5); row = r0 + (id & 31) * 4; p = P + (int64_t)kk * ld + row; }
    f32x4 v;
    const bool full = KC ? (row < R && kk + 4 <= k_end) : (kk < k_end && row + 4 <= R);
    if (ALIGNED && full) v = *reinterpret_cast<const f32x4*>(p);
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = KC ? (row < R && kk + e < k_end) : (kk < k_end && row + e < R);
        v[e] = ok ? p[e] : 0.0f;
      }
    }
    reg[i] = v;
  }
}

template <bool KC>
__device__ __forceinline__ void store_operand32(float* lds, const f32x4 (&reg)[2]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + i * 256;
    if (KC) {
      const int row = id >> 2, c = id & 3;
#pragma unroll
      for (int e = 0; e < 4; ++e) lds[row * LD_KC32 + c * 4 + e] = reg[i][e];
    } else {
      const int row = id >> 5, c = id & 31;
      *reinterpret_cast<f32x4*>(&lds[row * LD_TR32 + c * 4]) = reg[i];
    }
  }
}

template <bool KC>
__device__ __forceinline__ float read_frag32(const float* lds, int t16, int kk, int lane) {
  return KC ? lds[(t16 * 16 + (lane & 15)) * LD_KC32 + kk * 4 + (lane >> 4)]
            : lds[(kk * 4 + (lane >> 4)) * LD_TR32 + t16 * 16 + (lane & 15)];
}

template <bool TA, bool TB, bool ALIGNED>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(vmr_gemm_t g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* lds = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const TileCoord tc = tile_coord(g, tiles_m, tiles_n);
  const int m0 = tc.tm * BM, n0 = tc.tn * BN;
  const int z1 = tc.zb / g.Z2, z2 = tc.zb - z1 * g.Z2;
  const float* A = reinterpret_cast<const float*>(g.A) + z1 * g.sA1 + z2 * g.sA2;
  const float* B = reinterpret_cast<const float*>(g.B) + z1 * g.sB1 + z2 * g.sB2;
  const int64_t coff = z1 * g.sC1 + z2 * g.sC2;
  int k_begin = 0, k_end = g.K;
  if (g.splitk > 1) {
    int chunk = (g.K + g.splitk - 1) / g.splitk;
    chunk = (chunk + BK32 - 1) / BK32 * BK32;
    k_begin = tc.ks * chunk;
    k_end = min(g.K, k_begin + chunk);
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nk = k_end > k_begin ? (k_end - k_begin + BK32 - 1) / BK32 : 0;
  f32x4 ra[2], rb[2];
  if (nk > 0) {
    load_operand32<!TA, ALIGNED>(A, g.lda, m0, g.M, k_begin, k_end, ra);
    load_operand32<!TB, ALIGNED>(B, g.ldb, n0, g.N, k_begin, k_end, rb);
    store_operand32<!TA>(lds, ra);
    store_operand32<!TB>(lds + OP_FLOATS32, rb);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    float* cur = lds + (kt & 1) * 2 * OP_FLOATS32;
    float* nxt = lds + ((kt + 1) & 1) * 2 * OP_FLOATS32;
    const bool more = kt + 1 < nk;
    if (more) {
      const int k0 = k_begin + (kt + 1) * BK32;
      load_operand32<!TA, ALIGNED>(A, g.lda, m0, g.M, k0, k_end, ra);
      load_operand32<!TB, ALIGNED>(B, g.ldb, n0, g.N, k0, k_end, rb);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = read_frag32<!TA>(cur, wm * 4 + i, kk, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag32<!TB>(cur + OP_FLOATS32, wn * 4 + j, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      store_operand32<!TA>(nxt, ra);
      store_operand32<!TB>(nxt + OP_FLOATS32, rb);
    }
    __syncthreads();
  }
  stage_acc(lds, acc, wm, wn, lane);
  __syncthreads();
  epilogue<float, ALIGNED>(g, lds, m0, n0, tc.zb, reinterpret_cast<float*>(g.C) + coff,
                           reinterpret_cast<const float*>(g.residual) + coff,
                           reinterpret_cast<float*>(g.aux) + coff);
}

typedef void (*gemm_fn)(vmr_gemm_t, int, int);

template <bool TA, bool TB, bool AL>
gemm_fn pick_dtype(int dtype) {
  return dtype == VMR_BF16 ? (gemm_fn)gemm_bf16_kernel<TA, TB, AL> : (gemm_fn)gemm_f32_kernel<TA, TB, AL>;
}
template <bool AL>
gemm_fn pick_trans(int ta, int tb, int dtype) {
  if (!ta && !tb) return pick_dtype<false, false, AL>(dtype);
  if (!ta && tb) return pick_dtype<false, true, AL>(dtype);
  if (ta && !tb) return pick_dtype<true, false, AL>(dtype);
  return pick_dtype<true, true, AL>(dtype);
}

inline bool mult(int64_t v, int64_t m) { return (v % m) == 0; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int vmr_gemm(const vmr_gemm_t* gp, void* stream) {
  VMR_CHECK(gp != nullptr, "vmr_gemm: null descriptor");
  vmr_gemm_t g = *gp;
  VMR_CHECK(g.dtype == VMR_F32 || g.dtype == VMR_BF16, "vmr_gemm: bad dtype %d", g.dtype);
  VMR_CHECK(g.M >= 0 && g.N >= 0 && g.K >= 0, "vmr_gemm: negative dim");
  VMR_CHECK(g.A && g.B && g.C, "vmr_gemm: null operand");
  if (g.Z1 <= 0) g.Z1 = 1;
  if (g.Z2 <= 0) g.Z2 = 1;
  if (g.splitk <= 0) g.splitk = 1;
  VMR_CHECK(g.splitk == 1 || (g.flags & VMR_EPI_ACCUM), "vmr_gemm: splitk>1 needs VMR_EPI_ACCUM");
  VMR_CHECK(!(g.flags & VMR_EPI_BIAS) || g.bias, "vmr_gemm: bias flag without pointer");
  VMR_CHECK(!(g.flags & VMR_EPI_RESIDUAL) || g.residual, "vmr_gemm: residual flag without pointer");
  VMR_CHECK(!(g.flags & VMR_EPI_AUX) || g.aux, "vmr_gemm: aux flag without pointer");
  VMR_CHECK(!(g.flags & VMR_EPI_ROWSCALE) || g.rowscale, "vmr_gemm: rowscale flag without pointer");
  VMR_CHECK(!(g.flags & VMR_EPI_DROPOUT) || (g.drop_p >= 0.f && g.drop_p < 1.f), "vmr_gemm: bad drop_p");
  const int64_t Z = (int64_t)g.Z1 * g.Z2 * g.splitk;
  VMR_CHECK(Z <= 65535, "vmr_gemm: too many batches (%lld)", (long long)Z);
  if (g.M == 0 || g.N == 0) return 0;
  VMR_CHECK(g.lda >= (g.transA ? g.M : g.K) && g.ldb >= (g.transB ? g.N : g.K) && g.ldc >= g.N,
            "vmr_gemm: leading dimension too small (lda %lld ldb %lld ldc %lld)", (long long)g.lda,
            (long long)g.ldb, (long long)g.ldc);
  const int v = g.dtype == VMR_BF16 ? 8 : 4;  // elements per 16 B
  const int vc = (g.flags & (VMR_EPI_OUT_F32 | VMR_EPI_ACCUM)) ? 4 : v;
  bool al = aligned16(g.A) && aligned16(g.B) && aligned16(g.C) && mult(g.lda, v) && mult(g.ldb, v) &&
            mult(g.ldc, 8) && mult(g.sA1, v) && mult(g.sA2, v) && mult(g.sB1, v) && mult(g.sB2, v) &&
            mult(g.sC1, 8) && mult(g.sC2, 8);
  (void)vc;
  if (g.flags & (VMR_EPI_RESIDUAL | VMR_EPI_AUX)) {
    al = al && mult(g.ldr, 8);
    if (g.flags & VMR_EPI_RESIDUAL) al = al && aligned16(g.residual);
    if (g.flags & VMR_EPI_AUX) al = al && aligned16(g.aux);
  }
  const int tiles_m = cdiv(g.M, BM), tiles_n = cdiv(g.N, BN);
  gemm_fn fn = al ? pick_trans<true>(g.transA, g.transB, g.dtype) : pick_trans<false>(g.transA, g.transB, g.dtype);
  {  // > 64 KiB of dynamic LDS must be opted into once per kernel
    static thread_local const void* done[16];
    static thread_local int ndone = 0;
    bool seen = false;
    for (int i = 0; i < ndone; ++i) seen = seen || done[i] == reinterpret_cast<const void*>(fn);
    if (!seen) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
      if (e != hipSuccess) return vmr_fail(-5, "vmr_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
      if (ndone < 16) done[ndone++] = reinterpret_cast<const void*>(fn);
    }
  }
  dim3 grid((unsigned)(tiles_m * tiles_n), 1, (unsigned)Z);
  hipLaunchKernelGGL(fn, grid, dim3(256), SMEM_BYTES, (hipStream_t)stream, g, tiles_m, tiles_n);
  VMR_LAUNCH_CHECK();
  return 0;
}
