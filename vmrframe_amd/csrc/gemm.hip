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
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int CST_LD = 132;                    // fp32 staging row stride (floats)
constexpr int CST_BYTES = 64 * CST_LD * 4;     // epilogue staging: HALF a tile (64 rows) at a time = 33,792 B

struct TileCoord {
  int tm, tn, zb, ks;
};

__device__ __forceinline__ TileCoord tile_coord(const vmr_gemm_t& g, int tiles_m, int tiles_n, int bid_x, int bid_z,
                                                int grid_z) {
  // XCD-aware bijective remap: blocks with equal blockIdx.x % 8 share an XCD/L2;
  // give each XCD a contiguous run of logical tiles (n fastest), so the tiles that
  // re-read one A row-panel hit the same L2.
  if (g.splitk > 1 && grid_z == 1) {
    // split-K, unbatched: the K split rides in blockIdx.x (ks = bid % splitk).  Workgroups go to the 8 XCDs round-
    // robin by linear id, so with splitk = 8 each XCD owns ONE K slab of both operands for all output tiles: a
    // [K/8 x M] and a [K/8 x N] panel (2.4 MB each at cfg2) stay in that XCD's L2 and HBM reads each operand once.
    // (With ks in blockIdx.z every XCD swept all K slabs of one operand: 8x the HBM / Infinity-Cache traffic.)
    TileCoord t;
    const int bid = bid_x, sk = g.splitk;
    t.ks = bid % sk;
    const int tile = bid / sk;
    t.tm = tile / tiles_n;
    t.tn = tile - t.tm * tiles_n;
    t.zb = 0;
    return t;
  }
  const int nblk = tiles_m * tiles_n;
  const int bid = bid_x;
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  TileCoord t;
  t.tm = logical / tiles_n;
  t.tn = logical - t.tm * tiles_n;
  const int sk = g.splitk > 1 ? g.splitk : 1;
  t.zb = bid_z / sk;
  t.ks = bid_z - t.zb * sk;
  return t;
}

// Column owned by lane quad q = lane >> 4 in MFMA tile j of the wave's 64-column slab.  PERM (the weight operand
// is row-major, W [N][K]: the x.W^T products): the LDS-DMA source rows of W are permuted (perm_row128 below) so that
// tile j, MFMA row r holds weight row (j>>1)*32 + (r>>2)*8 + (j&1)*4 + (r&3): a lane's 4 values of tiles 2k and
// 2k+1 are then 8 CONSECUTIVE columns, and C / aux / residual move as 16-byte accesses (half the epilogue's
// memory instructions; the 4 lanes of a row cover a 64-byte segment per instruction).
template <bool PERM>
__device__ __forceinline__ int quad_col(int j, int q) {
  return PERM ? ((j >> 1) * 32 + q * 8 + (j & 1) * 4) : (j * 16 + q * 4);
}
__device__ __forceinline__ int perm_row128(int rho) {   // LDS row of the [128][BK] weight tile -> weight row
  const int j = (rho >> 4) & 3, r = rho & 15;
  return (rho & 64) | ((j >> 1) << 5) | ((r >> 2) << 3) | ((j & 1) << 2) | (r & 3);
}

// residual values of the lane, as loaded: 8 bytes per (i, j) quad; with PERM quads 2k and 2k+1 came as one 16-byte load
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ TileCoord tile_coord(const vmr_gemm_t& g, int tiles_m, int tiles_n) {
  return tile_coord(g, tiles_m, tiles_n, blockIdx.x, blockIdx.z, gridDim.z);
}

// swizzle of the [k][128] (M/N-contiguous) bf16 image: 32-B chunk index ^= f(k)
__device__ __forceinline__ int swz_tr(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

// ------------------------------------------------------------------ epilogue
// The accumulators go through LDS one 64-row half tile at a time (33 KiB, so the
// epilogue never decides the workgroup's LDS footprint): the two waves that own
// the half stage it as fp32, then all 256 threads apply bias / ReLU / dropout /
// aux / residual / row-mask with 16-byte coalesced global accesses.
template <typename T, bool ALIGNED, int NT = 256, int ROWS = 64>
__device__ __forceinline__ void epilogue_half(const vmr_gemm_t& g, const float* cst, int m0, int n0, int zb,
                                              T* __restrict__ C, const T* __restrict__ Rsd, T* __restrict__ Aux) {
  const int tid = threadIdx.x;
  const int flags = g.flags;
  if (flags & VMR_EPI_ACCUM) {
    float* Cf = reinterpret_cast<float*>(C);
    // 256 contiguous bytes per wave-instruction: the fast shape for float atomics
    for (int pass = 0; pass < ROWS / (NT / 64); ++pass) {
      const int row = pass * (NT / 64) + (tid >> 6);
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
  constexpr int PASSES = ROWS / (NT / 16);
  // residual rows of ALL passes are requested up front: one exposed HBM latency per tile instead of
  // one per pass (the epilogue is a dependent load -> math -> store chain otherwise)
  float rres[PASSES][8];
  const bool res_vec = ALIGNED && (flags & VMR_EPI_RESIDUAL) && (g.N - n0 - (tid & 15) * 8) >= 8;
  if (res_vec) {
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
      const int gm = min(m0 + pass * (NT / 16) + (tid >> 4), g.M - 1);   // clamped: the load is unconditional
      Vec8<T>::load(Rsd + (int64_t)(gm / g.res_div) * g.ldr + n0 + (tid & 15) * 8, rres[pass]);
    }
  }
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
    const int row = pass * (NT / 16) + (tid >> 4);
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
    uint32_t keep = 0xFFu;
    if (flags & VMR_EPI_DROPOUT) {
      const uint64_t idx0 = ((uint64_t)zb * g.M + gm + g.drop_row0) * (uint64_t)g.N + (uint64_t)gn;
      if ((g.N & 7) == 0) keep = vmr_keep8(seed, idx0, thresh);   // gn is a multiple of 8
      else {
        keep = 0;
        for (int e = 0; e < 8; ++e) keep |= (uint32_t)vmr_keep(seed, idx0 + e, thresh) << e;
      }
    }
    float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (flags & VMR_EPI_RESIDUAL) {
      if (res_vec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) rv[e] = rres[pass][e];
      } else {
        const T* rp = Rsd + (int64_t)(gm / g.res_div) * g.ldr + gn;
        for (int e = 0; e < nvalid; ++e) rv[e] = to_f<T>(rp[e]);
      }
    }
    const bool res_pre = (flags & VMR_EPI_RES_PRE) != 0;   // residual joins the pre-activation (before ReLU / dropout)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = v[e] * g.alpha;
      if ((flags & VMR_EPI_BIAS) && e < nvalid) x += g.bias_scale * g.bias[gn + e] + (g.bias2 ? g.bias2[gn + e] : 0.f);
      if (res_pre) x += rv[e];
      if (flags & VMR_EPI_RELU) x = fmaxf(x, 0.0f);
      if (flags & VMR_EPI_DROPOUT) x = ((keep >> e) & 1) ? x * dscale : 0.0f;
      v[e] = x;
    }
    const bool vec = ALIGNED && nvalid == 8;
    if (flags & VMR_EPI_AUX) {
      T* ap = Aux + (int64_t)gm * g.ldr + gn;
      if (vec) Vec8<T>::store(ap, v);
      else for (int e = 0; e < nvalid; ++e) ap[e] = from_f<T>(v[e]);
    }
    if ((flags & VMR_EPI_RESIDUAL) && !res_pre) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rv[e];
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

// caller has passed the barrier that ends the K loop
template <typename T, bool ALIGNED, bool FULL = false, bool TRANSPOSED = false, bool PERM = false>
__device__ __forceinline__ void epilogue(const vmr_gemm_t& g, float* cst, const f32x4 (&acc)[4][4], int wm, int wn,
                                         int lane, int m0, int n0, int zb, T* __restrict__ C,
                                         const T* __restrict__ Rsd, T* __restrict__ Aux) {
  if (FULL) {  // the kernel owns >= 128*132*4 B of LDS: stage the whole tile at once (one barrier)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (TRANSPOSED) {   // acc[i][j][r] = C[i*16 + (lane&15)][j*16 + (lane>>4)*4 + r]: one 16-byte LDS write
          // (PERM: the weight rows were permuted on their way into LDS, see quad_col in the register-direct epilogue)
          const int cq = PERM ? ((j >> 1) * 32 + (lane >> 4) * 8 + (j & 1) * 4) : (j * 16 + (lane >> 4) * 4);
          *reinterpret_cast<f32x4*>(&cst[(wm * 64 + i * 16 + (lane & 15)) * CST_LD + wn * 64 + cq]) = acc[i][j];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            cst[(wm * 64 + i * 16 + (lane >> 4) * 4 + r) * CST_LD + wn * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
        }
      }
    __syncthreads();
    epilogue_half<T, ALIGNED, 256, 128>(g, cst, m0, n0, zb, C, Rsd, Aux);
    return;
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            cst[(i * 16 + (lane >> 4) * 4 + r) * CST_LD + wn * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
    }
    __syncthreads();
    epilogue_half<T, ALIGNED>(g, cst, m0 + half * 64, n0, zb, C, Rsd, Aux);
    if (half == 0) __syncthreads();
  }
}

// ---------------------------------------------------------------- bf16 kernel
// BK = 64: 2 x 32 KiB of staging -> 2 workgroups / CU.
// BK = 32: 2 x 16 KiB            -> LDS allows 4 / CU, registers 3 / CU: 768 resident tiles, so the
//          592-tile packed-token GEMMs ([9472 x 1024]) run as ONE wave of workgroups instead of two.
__device__ __forceinline__ int swz_kc32(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

// Load one operand tile (128 x BK bf16) into BK/16 x bf16x8 registers per thread.
//   KC (k-contiguous):  tile rows = 128 (m or n), BK/8 chunks of 8 k each
//   !KC (m/n-contiguous): tile rows = BK (k), 16 chunks of 8 m each
template <bool KC, bool ALIGNED, int BK>
__device__ __forceinline__ void load_operand(const bf16_t* __restrict__ P, int64_t ld, int r0, int R,
                                             int k0, int k_end, bf16x8 (&reg)[BK / 16]) {
  const int tid = threadIdx.x;
  constexpr int CPR = BK / 8;  // chunks per row of a k-contiguous tile
#pragma unroll
  for (int i = 0; i < BK / 16; ++i) {
    const int id = tid + i * 256;
    int row, kk;  // element (index in the m/n space, k index)
    const bf16_t* p;
    bool full, any;
    if (KC) {
      row = r0 + id / CPR; kk = k0 + (id % CPR) * 8;
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

template <bool KC, int BK>
__device__ __forceinline__ void store_operand(unsigned char* lds, const bf16x8 (&reg)[BK / 16]) {
  const int tid = threadIdx.x;
  constexpr int CPR = BK / 8;
#pragma unroll
  for (int i = 0; i < BK / 16; ++i) {
    const int id = tid + i * 256;
    int off;
    if (KC) {
      const int row = id / CPR, c = id % CPR;
      off = BK == 64 ? row * 128 + ((c ^ (row & 7)) << 4) : row * 64 + ((c ^ swz_kc32(row)) << 4);
    } else {
      const int row = id >> 4, c16 = id & 15;
      off = row * 256 + ((((c16 >> 1) ^ swz_tr(row))) << 5) + ((c16 & 1) << 4);
    }
    *reinterpret_cast<bf16x8*>(lds + off) = reg[i];
  }
}

// (lds_read_tr / lgkm_wait / frag_pin: common.h)
// fragment for the 16-row (or 16-col) MFMA tile `t16` (0..7 within the 128 tile), 32-deep k-substep kk
// ASMTR: the transposed read is the inline-asm one (LDS-DMA kernels); false: the builtin, waits left to the compiler
template <bool KC, int BK, bool ASMTR = false>
__device__ __forceinline__ bf16x8 read_frag(const unsigned char* lds, int t16, int kk, int lane) {
  if (KC) {
    const int row = t16 * 16 + (lane & 15);
    if (BK == 64) {
      const int chunk = kk * 4 + (lane >> 4);
      return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((chunk ^ (row & 7)) << 4));
    } else {
      return *reinterpret_cast<const bf16x8*>(lds + row * 64 + (((lane >> 4) ^ swz_kc32(row)) << 4));
    }
  } else {
    const int g = lane >> 4, ii = lane & 15, q = ii >> 2, p = ii & 3;
    const int r = kk * 32 + 8 * g + q;
    lds += (t16 >> 3) * (BK * 256);   // a 256-wide operand = two [BK][128] images back to back
    t16 &= 7;
    const int a0 = r * 256 + ((t16 ^ swz_tr(r)) << 5) + p * 8;
    const int a1 = (r + 4) * 256 + ((t16 ^ swz_tr(r + 4)) << 5) + p * 8;
    union { struct { s16x4 l, h; } s; bf16x8 v; } u;
    if constexpr (ASMTR) {
      u.s.l = lds_read_tr(lds + a0);
      u.s.h = lds_read_tr(lds + a1);
    } else {
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
      u.s.l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + a0));
      u.s.h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + a1));
    }
    return u.v;
  }
}

template <int BK> struct Geom16 {
  static constexpr int OP_BYTES = 128 * BK * 2;
  static constexpr int STAGES_BYTES = 4 * OP_BYTES;
  static constexpr int SMEM = STAGES_BYTES > CST_BYTES ? STAGES_BYTES : CST_BYTES;
};

template <typename E, bool TA, bool TB, bool ALIGNED, int BK>
__global__ __launch_bounds__(256, (BK == 32 ? 3 : 2)) void gemm_e16_kernel(vmr_gemm_t g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int OPB = Geom16<BK>::OP_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const TileCoord tc = tile_coord(g, tiles_m, tiles_n);
  const int m0 = tc.tm * BM, n0 = tc.tn * BN;
  const int z1 = tc.zb / g.Z2, z2 = tc.zb - z1 * g.Z2;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A) + z1 * g.sA1 + z2 * g.sA2;
  const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B) + z1 * g.sB1 + z2 * g.sB2;
  const int64_t coff = z1 * g.sC1 + z2 * g.sC2 + ((g.flags & VMR_EPI_SLAB) ? (int64_t)tc.ks * g.M * g.ldc : 0);

  int k_begin = 0, k_end = g.K;
  if (g.splitk > 1) {
    int chunk = (g.K + g.splitk - 1) / g.splitk;
    chunk = (chunk + 63) / 64 * 64;
    k_begin = tc.ks * chunk;
    k_end = min(g.K, k_begin + chunk);
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = k_end > k_begin ? (k_end - k_begin + BK - 1) / BK : 0;
  bf16x8 ra[BK / 16], rb[BK / 16];
  if (nk > 0) {
    load_operand<!TA, ALIGNED, BK>(A, g.lda, m0, g.M, k_begin, k_end, ra);
    load_operand<!TB, ALIGNED, BK>(B, g.ldb, n0, g.N, k_begin, k_end, rb);
    store_operand<!TA, BK>(smem, ra);
    store_operand<!TB, BK>(smem + OPB, rb);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    unsigned char* cur = smem + (kt & 1) * 2 * OPB;
    unsigned char* nxt = smem + ((kt + 1) & 1) * 2 * OPB;
    const bool more = kt + 1 < nk;
    if (more) {  // issue the next tile's global loads before the MFMAs (latency hides under compute)
      const int k0 = k_begin + (kt + 1) * BK;
      load_operand<!TA, ALIGNED, BK>(A, g.lda, m0, g.M, k0, k_end, ra);
      load_operand<!TB, ALIGNED, BK>(B, g.ldb, n0, g.N, k0, k_end, rb);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = read_frag<!TA, BK>(cur, wm * 4 + i, kk, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag<!TB, BK>(cur + OPB, wn * 4 + j, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = mfma16<E>(fa[i], fb[j], acc[i][j]);
    }
    if (more) {
      store_operand<!TA, BK>(nxt, ra);
      store_operand<!TB, BK>(nxt + OPB, rb);
    }
    __syncthreads();
  }
  epilogue<E, ALIGNED>(g, reinterpret_cast<float*>(smem), acc, wm, wn, lane, m0, n0, tc.zb,
                       (g.flags & (VMR_EPI_OUT_F32 | VMR_EPI_ACCUM)) ? reinterpret_cast<E*>(reinterpret_cast<float*>(g.C) + coff)
                                                                     : reinterpret_cast<E*>(g.C) + coff,
                       reinterpret_cast<const E*>(g.residual) + coff, reinterpret_cast<E*>(g.aux) + coff);
}

// ------------------------------------------------------- bf16 LDS-DMA kernel
// Interior path (M, N multiples of 128, K-range multiples of 32, 16-byte aligned rows): the
// operand tiles go global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction,
// no staging VGPRs, no ds_write), through a 4-deep LDS ring (4 x 16 KiB): the loads of K-step
// t+3 are issued while step t is computed, and each step waits with a COUNTED vmcnt (8 loads of
// the two later steps stay in flight) + one raw s_barrier, so HBM/L2 latency (~900 cycles) is
// covered by three K-steps of MFMA instead of stalling every step.  The LDS image is lane-linear
// per wave-instruction, so the XOR swizzles of the register path are applied to the per-lane SOURCE
// address (same image, same fragment reads).
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

// one operand tile (128 rows x BK, or BK x 128 for the transposed image) = 128*BK*2 bytes = NB 1-KiB blocks
// (ROWS = 160 for the 160-row A tile of the ragged-M variant: rows past rmax re-read row rmax)
template <bool KC, int BK, int ROWS = 128, int NW = 4, bool PERM = false>
__device__ __forceinline__ void dma_operand(const bf16_t* __restrict__ P, int64_t ld, int r0, int k0,
                                            unsigned char* lds, int wid, int lane, int rmax = 0x7fffffff) {
  constexpr int NB = ROWS * BK * 2 / 1024;  // 8 (BK=32) or 16 (BK=64) blocks per 128 rows, NB/NW per wave
  static_assert(NB % NW == 0, "operand blocks must divide over the waves");
  static_assert(ROWS == 128 || BK == 64, "tall / wide tiles: BK=64 images only");
  static_assert(KC || ROWS % 128 == 0, "transposed image: whole [BK][128] images");
  static_assert(!PERM || (KC && BK == 64 && ROWS == 128), "row permutation: the k-contiguous 128-row weight tile");
#pragma unroll
  for (int jj = 0; jj < NB / NW; ++jj) {
    const int j = wid * (NB / NW) + jj;
    const bf16_t* src;
    if (KC) {
      if (BK == 64) {       // 128-B rows: one wave-instruction = 8 whole rows = 8 full cache lines
        const int row = 8 * j + (lane >> 3);
        const int c = (lane & 7) ^ (row & 7);
        const int grow = PERM ? perm_row128(row) : row;   // (weight tile of the x.W^T products: see quad_col)
        src = P + (int64_t)min(r0 + grow, rmax) * ld + k0 + c * 8;
      } else {              // 64-B rows
        const int row = 16 * j + (lane >> 2);
        const int c = (lane & 3) ^ swz_kc32(row);
        src = P + (int64_t)(r0 + row) * ld + k0 + c * 8;
      }
    } else {
      constexpr int BPI = BK / 4;                  // 1-KiB blocks per [BK][128] image
      const int r = 4 * (j % BPI) + (lane >> 4);
      const int ph16 = lane & 15;
      const int c32 = (ph16 >> 1) ^ swz_tr(r);
      src = P + (int64_t)(k0 + r) * ld + r0 + (j / BPI) * 128 + (c32 * 2 + (ph16 & 1)) * 8;
    }
    __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(lds + j * 1024), 16, 0, 0);
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
#define VMR_VMCNT_CASE(n) else if (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  VMR_VMCNT_CASE(4); VMR_VMCNT_CASE(5); VMR_VMCNT_CASE(6); VMR_VMCNT_CASE(7); VMR_VMCNT_CASE(8); VMR_VMCNT_CASE(9);
  VMR_VMCNT_CASE(12); VMR_VMCNT_CASE(14); VMR_VMCNT_CASE(16); VMR_VMCNT_CASE(18);
  else static_assert(N == 0 || N == 4 || N == 5 || N == 6 || N == 7 || N == 8 || N == 9 || N == 12 || N == 14 || N == 16 || N == 18,
                     "add the immediate");
#undef VMR_VMCNT_CASE
}

// Register-direct epilogue of the LDS-DMA kernel (bf16 output, interior tiles).  The MFMAs are issued
// with the operands swapped (C^T = W.x^T), so a lane owns FOUR CONSECUTIVE COLUMNS of one output row:
// bias / residual / aux / C move as 8-byte accesses straight from the accumulators -- no LDS staging,
// no workgroup barrier (a wave's stores overlap the other waves' and the co-resident workgroup's
// MFMAs), and one dropout hash serves exactly the lane's 4 elements.
template <typename E, int MT, bool PERM>
__device__ __forceinline__ void epilogue_direct(const vmr_gemm_t& g, const f32x4 (&acc)[MT][4], int wm, int wn, int lane,
                                                int m0, int n0, int zb, E* __restrict__ C,
                                                const u32x2 (&rres)[MT][4], E* __restrict__ Aux) {
  const int flags = g.flags;
  const int rbase = m0 + wm * (MT * 16) + (lane & 15);
  const int q = lane >> 4;
  const int cw = n0 + wn * 64;
  f32x4 bias4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
  {
    bias4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (flags & VMR_EPI_BIAS) {
      bias4[j] = *reinterpret_cast<const f32x4*>(g.bias + cw + quad_col<PERM>(j, q)) * g.bias_scale;
      if (g.bias2) bias4[j] += *reinterpret_cast<const f32x4*>(g.bias2 + cw + quad_col<PERM>(j, q));
    }
  }
  const uint32_t thresh = vmr_drop_thresh(g.drop_p);
  const float dscale = (flags & VMR_EPI_DROPOUT) ? 1.0f / (1.0f - g.drop_p) : 1.0f;
  const uint32_t seed = vmr_seed(g.drop_seed, g.drop_step);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int gm = rbase + i * 16;
    if (MT != 4 && gm >= g.M) continue;   // ragged last tile of the 160-row variant
    const float rs = (flags & VMR_EPI_ROWSCALE) ? g.rowscale[gm] : 1.0f;
    float out[4][4], ax[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = cw + quad_col<PERM>(j, q);
      const float r4[4] = {e16_lo<E>(rres[i][j][0]), e16_hi<E>(rres[i][j][0]), e16_lo<E>(rres[i][j][1]), e16_hi<E>(rres[i][j][1])};
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = acc[i][j][e] * g.alpha + bias4[j][e];
        if (flags & VMR_EPI_RES_PRE) x += r4[e];
        if (flags & VMR_EPI_RELU) x = fmaxf(x, 0.0f);
        v[e] = x;
      }
      if (flags & VMR_EPI_DROPOUT) {
        const uint64_t idx0 = ((uint64_t)zb * g.M + gm + g.drop_row0) * (uint64_t)g.N + (uint64_t)gn;  // multiple of 4
        const uint2 h = vmr_hash4(seed, idx0 >> 2);
        v[0] = (h.x & 0xFFFFu) >= thresh ? v[0] * dscale : 0.f;
        v[1] = (h.x >> 16) >= thresh ? v[1] * dscale : 0.f;
        v[2] = (h.y & 0xFFFFu) >= thresh ? v[2] * dscale : 0.f;
        v[3] = (h.y >> 16) >= thresh ? v[3] * dscale : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) ax[j][e] = v[e];
      if ((flags & VMR_EPI_RESIDUAL) && !(flags & VMR_EPI_RES_PRE)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r4[e];
      }
      if (flags & VMR_EPI_ROWSCALE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= rs;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) out[j][e] = v[e];
    }
    if (flags & VMR_EPI_OUT_F32) {   // fp32 result / split-K slab: one 16-byte store per quad
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(C) + (int64_t)gm * g.ldc + cw + quad_col<PERM>(j, q)) =
            (f32x4){out[j][0], out[j][1], out[j][2], out[j][3]};
    } else if (PERM) {               // quads 2k, 2k+1 are 8 consecutive columns: 16-byte stores
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int gn = cw + quad_col<PERM>(2 * k, q);
        const float o8[8] = {out[2 * k][0], out[2 * k][1], out[2 * k][2], out[2 * k][3],
                             out[2 * k + 1][0], out[2 * k + 1][1], out[2 * k + 1][2], out[2 * k + 1][3]};
        Vec8<E>::store(C + (int64_t)gm * g.ldc + gn, o8);
        if (flags & VMR_EPI_AUX) {
          const float a8[8] = {ax[2 * k][0], ax[2 * k][1], ax[2 * k][2], ax[2 * k][3],
                               ax[2 * k + 1][0], ax[2 * k + 1][1], ax[2 * k + 1][2], ax[2 * k + 1][3]};
          if (flags & VMR_EPI_AUX_BITS) {   // the lane's 8 consecutive columns = one byte of the [M][N/8] bit matrix
            uint32_t bits = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) bits |= (a8[e] != 0.f ? 1u : 0u) << e;
            reinterpret_cast<unsigned char*>(g.aux)[((int64_t)zb * g.M + gm) * (g.N >> 3) + (gn >> 3)] = (unsigned char)bits;
          } else {
            Vec8<E>::store(Aux + (int64_t)gm * g.ldr + gn, a8);
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int gn = cw + quad_col<PERM>(j, q);
        Vec4<E>::store(C + (int64_t)gm * g.ldc + gn, out[j]);
        if (flags & VMR_EPI_AUX) Vec4<E>::store(Aux + (int64_t)gm * g.ldr + gn, ax[j]);
      }
    }
  }
}

// MT = 16-row MFMA tiles per wave along M: 4 -> the 128x128 tile; 5 -> a 160x128 tile (A not
// transposed, register-direct epilogue only, ragged last row tile allowed) that turns the 592-tile,
// 1.16-round grids of the packed [9472 x 1024] products into ONE round of 480 workgroups.
// WM = waves along M: 2 -> 256 threads, two workgroups per CU; 4 -> 512 threads, ONE 256x128 (MT = 4) or
// 320x128 (MT = 5) workgroup per CU.  The K loop is bound by the L2 -> LDS fill (measured: the loop without
// MFMAs takes 90 % of the full time, ~60 GB/s per CU), so bytes per flop decide: one wide tile per CU
// fetches (256 + 128) rows per K-step where two 128x128 tiles fetch 2 x (128 + 128).
template <typename E, bool TA, bool TB, int BK, int NST, int MT = 4, int WM = 2>
__device__ __forceinline__ void gemm_dma_body(const vmr_gemm_t& g, int tiles_m, int tiles_n, unsigned char* smem, int bid_x,
                                              int bid_z, int grid_z) {
  static_assert(MT == 4 || (!TA && BK == 64), "tall tile: A row-major, BK = 64");
  static_assert(WM == 2 || WM == 4, "2 or 4 waves along M");
  constexpr int NW = 2 * WM;                 // waves
  constexpr int TBM = MT * 16 * WM;          // tile rows
  constexpr int OPA = TBM * BK * 2;          // A operand bytes per stage
  constexpr int OPB = 128 * BK * 2;          // B operand bytes per stage
  constexpr int LPS = (OPA + OPB) / 1024 / NW;  // loads per wave per K-step (A + B)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const TileCoord tc = tile_coord(g, tiles_m, tiles_n, bid_x, bid_z, grid_z);
  const int m0 = tc.tm * TBM, n0 = tc.tn * BN;
  const int z1 = tc.zb / g.Z2, z2 = tc.zb - z1 * g.Z2;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A) + z1 * g.sA1 + z2 * g.sA2;
  const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B) + z1 * g.sB1 + z2 * g.sB2;
  const int64_t coff = z1 * g.sC1 + z2 * g.sC2 + ((g.flags & VMR_EPI_SLAB) ? (int64_t)tc.ks * g.M * g.ldc : 0);
  int k_begin = 0, k_end = g.K;
  if (g.splitk > 1) {
    int chunk = (g.K + g.splitk - 1) / g.splitk;
    chunk = (chunk + 63) / 64 * 64;
    k_begin = tc.ks * chunk;
    k_end = min(g.K, k_begin + chunk);
  }
  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nk = k_end > k_begin ? (k_end - k_begin) / BK : 0;
  // The residual quads of the register-direct epilogue are requested FIRST (oldest in the vmcnt
  // order, so every counted wait below also covers them): their HBM latency hides under the whole
  // K loop instead of being exposed once per tile.
  // weight operand row-major and BK = 64: its LDS rows are permuted so the epilogue moves 16 bytes per lane (quad_col)
  constexpr bool PERM = !TB && BK == 64;
  u32x2 rres[MT][4];
  const bool direct = MT != 4 || WM != 2 || !(g.flags & VMR_EPI_ACCUM);   // atomics keep the LDS-staged, 256-B-per-wave shape
  if (direct && (g.flags & VMR_EPI_RESIDUAL)) {
    const bf16_t* Rsd = reinterpret_cast<const bf16_t*>(g.residual) + coff;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const bf16_t* rrow = Rsd + (int64_t)(min(m0 + wm * (MT * 16) + (lane & 15) + i * 16, g.M - 1) / g.res_div) * g.ldr +
                           n0 + wn * 64;
      if constexpr (PERM) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const uint4 w = *reinterpret_cast<const uint4*>(rrow + quad_col<true>(2 * k, lane >> 4));
          rres[i][2 * k] = (u32x2){w.x, w.y};
          rres[i][2 * k + 1] = (u32x2){w.z, w.w};
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) rres[i][j] = *reinterpret_cast<const u32x2*>(rrow + quad_col<false>(j, lane >> 4));
      }
    }
  }

#pragma unroll
  for (int s = 0; s < NST - 1; ++s) {
    if (s < nk) {
      dma_operand<!TA, BK, TBM, NW>(A, g.lda, m0, k_begin + s * BK, smem + s * (OPA + OPB), wid, lane, g.M - 1);
      dma_operand<!TB, BK, 128, NW, PERM>(B, g.ldb, n0, k_begin + s * BK, smem + s * (OPA + OPB) + OPA, wid, lane);
    }
  }
  // Software-pipelined fragment reads: the ds_reads of the next 32-deep k-substep are in flight while
  // the 16 MFMAs of the current one issue (rotated loop: the MFMAs of substep (kt, last) run at the top
  // of iteration kt+1, beside that iteration's first reads).
  constexpr int NKK = BK / 32;
  bf16x8 fa[2][MT], fb[2][4];
  // bias gradient of the weight-gradient product (TA): column sums of A^T = one more MFMA per A fragment against
  // a ones operand, in the waves that own the first 64 columns of the first column tile
  constexpr int NCS = TA ? MT : 1;
  f32x4 accdb[NCS];
#pragma unroll
  for (int i = 0; i < NCS; ++i) accdb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool colsum = TA && g.a_colsum != nullptr && tc.tn == 0 && wn == 0;   // wave-uniform
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = bits_from_f<E>(1.0f);
  auto mma = [&](int buf) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = mfma16<E>(fb[buf][j], fa[buf][i], acc[i][j]);  // C^T tile
    if constexpr (TA) {
      if (colsum) {
#pragma unroll
        for (int i = 0; i < MT; ++i) accdb[i] = mfma16<E>(ones, fa[buf][i], accdb[i]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // transposed operands are read by inline asm (lds_read_tr): wait for the OLDER substep's fragments ourselves -- the LDS
  // returns in order, so "at most as many outstanding as were issued since" retires them -- and pin their registers
  constexpr int NEWER = (TA ? 2 * MT : MT) + (TB ? 8 : 4);          // LDS reads of one substep
  auto settle = [&](int buf, bool last) {
    if constexpr (TA || TB) {
      if (last) lgkm_wait<0>();
      else lgkm_wait<(NEWER < 15 ? NEWER : 15)>();
#pragma unroll
      for (int i = 0; i < MT; ++i) frag_pin(fa[buf][i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) frag_pin(fb[buf][j]);
    }
  };
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's loads of step kt must have landed; those of the later NST-2 steps stay in flight
    const int later = min(nk - 1 - kt, NST - 2);
    if (later >= 2) wait_vmcnt<2 * LPS>();
    else if (later == 1) wait_vmcnt<LPS>();
    else wait_vmcnt<0>();
    // this wave's fragment reads of step kt-1 are complete before any wave's DMA may overwrite that buffer
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's part of step kt is in LDS; step kt-1's buffer is free
    if (kt + NST - 1 < nk) {
      unsigned char* dst = smem + ((kt + NST - 1) % NST) * (OPA + OPB);
      const int k0 = k_begin + (kt + NST - 1) * BK;
      dma_operand<!TA, BK, TBM, NW>(A, g.lda, m0, k0, dst, wid, lane, g.M - 1);
      dma_operand<!TB, BK, 128, NW, PERM>(B, g.ldb, n0, k0, dst + OPA, wid, lane);
    }
    const unsigned char* cur = smem + (kt % NST) * (OPA + OPB);
    if constexpr (NKK == 1) {
      // BK = 32: one substep per step, so there is no "other" fragment set to keep in flight -- read, wait, multiply (the
      // other resident waves cover the wait).  (Until round 2 this case went through the rotated loop below, whose
      // "previous substep" was the set just overwritten: step 0 was dropped and the last step counted twice.  Only
      // VMR_GEMM_DMA=1 ever selected it, and no test did; tests/test_gpu_a_ops.py now covers every ring variant.)
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[0][i] = read_frag<!TA, BK, true>(cur, wm * MT + i, 0, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[0][j] = read_frag<!TB, BK, true>(cur + OPA, wn * 4 + j, 0, lane);
      if constexpr (TA || TB) settle(0, true);
      mma(0);
    } else {
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const int buf = kk & 1;
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[buf][i] = read_frag<!TA, BK, true>(cur, wm * MT + i, kk, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[buf][j] = read_frag<!TB, BK, true>(cur + OPA, wn * 4 + j, kk, lane);
      // while those reads fly: the MFMAs of the PREVIOUS substep (the last one of step kt-1 when kk == 0)
      if (kk > 0) { settle((kk - 1) & 1, false); mma((kk - 1) & 1); }
      else if (kt > 0) { settle((NKK - 1) & 1, false); mma((NKK - 1) & 1); }
    }
    }
  }
  if constexpr (NKK > 1) {
    if (nk > 0) { settle((NKK - 1) & 1, true); mma((NKK - 1) & 1); }
  }
  if constexpr (TA) {
    if (colsum && lane < 16) {   // every row of the ones-product holds the column sums: lanes 0..15 carry m = lane
#pragma unroll
      for (int i = 0; i < MT; ++i) atomicAdd(&g.a_colsum[m0 + wm * (MT * 16) + i * 16 + lane], accdb[i][0]);
    }
  }
  if (direct) {
    epilogue_direct<E, MT, PERM>(g, acc, wm, wn, lane, m0, n0, tc.zb,
                                 (g.flags & VMR_EPI_OUT_F32) ? reinterpret_cast<E*>(reinterpret_cast<float*>(g.C) + coff)
                                                             : reinterpret_cast<E*>(g.C) + coff,
                                 rres, reinterpret_cast<E*>(g.aux) + coff);
    return;
  }
  if constexpr (MT == 4 && WM == 2) {
    __syncthreads();
    epilogue<E, true, true, true, PERM>(g, reinterpret_cast<float*>(smem), acc, wm, wn, lane, m0, n0, tc.zb,
                                        (g.flags & (VMR_EPI_OUT_F32 | VMR_EPI_ACCUM))
                                            ? reinterpret_cast<E*>(reinterpret_cast<float*>(g.C) + coff)
                                            : reinterpret_cast<E*>(g.C) + coff,
                                        reinterpret_cast<const E*>(g.residual) + coff, reinterpret_cast<E*>(g.aux) + coff);
  }
}

template <typename E, bool TA, bool TB, int MT = 4, int WM = 2>
__global__ __launch_bounds__(WM * 128, (WM == 2 ? 2 : 1)) void gemm_e16_dma_kernel(vmr_gemm_t g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  gemm_dma_body<E, TA, TB, 64, 2, MT, WM>(g, tiles_m, tiles_n, smem, blockIdx.x, blockIdx.z, gridDim.z);
}

// ---------------------------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, ONE 512-thread workgroup per CU, eight phases per two K-tiles ("8-phase" schedule, the guide's
// cdna_hip_programming.md section 5 template re-derived for this library's LDS images and swapped-operand epilogue).
//   * 8 waves = 2 (M) x 4 (N), wave tile 128 x 64 = acc[8][4]; a PHASE is one 64 x 32 quadrant of it over the whole
//     K-tile: 16 MFMAs.  Quadrant order (A rows, B cols): (lo,lo) (lo,hi) (hi,hi) (hi,lo); the B-lo fragments stay in
//     registers from phase 1 to phase 4, so a K-tile's four LDS units are last read in phases 1, 1, 2, 3:
//         unit 0 "Alo" = the rows every wave reads in phase 1 (rows {0..63, 128..191} of the A tile),
//         unit 1 "Blo" = weight rows c*64 + [0,32) of the four 64-column groups, unit 2 "Bhi" = c*64 + [32,64),
//         unit 3 "Ahi" = A rows {64..127, 192..255}          (16 KiB each = two 1-KiB LDS-DMA blocks per wave).
//   * two K-tile buffers (2 x 64 KiB).  Phase P (counted over the whole K loop) stages unit P + 6 -- the slot whose last
//     read was two or three phases ago -- so every unit is requested 5-6 phases before its first read, and one counted
//     wait per phase, "all but the four youngest units" (vmcnt(8)), retires exactly the unit the NEXT phase reads first.
//   * the two wave rows (wm = 0 / 1, one wave of each per SIMD) run one barrier apart: while one group issues its 16
//     MFMAs (s_setprio 1) the other issues fragment reads, its share of the LDS-DMA requests and the wait; two raw
//     s_barriers per phase keep them alternating.  Hazards (both groups): a unit is read one phase AFTER the phase whose
//     pre-barrier wait retired it, and re-staged at least two phases after its last read.
// Layouts: TR = false: both operands K-contiguous (x.W^T: the forward products, dX on the K-major weight copies);
// TR = true: both operands stored [K][rows] (the weight gradient dz^T.x), [64][128] images read with ds_read_b64_tr_b16.
constexpr int P8_UNIT = 16384, P8_BUF = 4 * P8_UNIT, P8_SMEM = 2 * P8_BUF;

template <int N> __device__ __forceinline__ void p8_vmcnt() {
  if constexpr (N >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <typename E, bool TR, bool PERM>
__device__ __forceinline__ void gemm_p8_body(const vmr_gemm_t& g, int tiles_m, int tiles_n, unsigned char* smem, int bid) {
  constexpr int dbg = 0;   // (ablation bits of the round-2 study: 1 = no LDS-DMA in the loop, 2 = no MFMAs, 4 = no fragment reads)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  // tile / split of this workgroup (same conventions as tile_coord: the K split rides in the low bits of the id)
  int ks = 0, logical = bid;
  if (g.splitk > 1) {
    ks = bid % g.splitk;
    logical = bid / g.splitk;
  } else {
    const int nblk = tiles_m * tiles_n, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = logical / tiles_n, tn = logical - tm * tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  int k_begin = 0, k_end = g.K;
  if (g.splitk > 1) {
    int chunk = (g.K + g.splitk - 1) / g.splitk;
    chunk = (chunk + 63) / 64 * 64;
    k_begin = ks * chunk;
    k_end = min(g.K, k_begin + chunk);
  }
  const int nk = k_end > k_begin ? (k_end - k_begin) / 64 : 0;
  const int U = 4 * nk;                                    // staging units of this workgroup
  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);
  const int64_t coff = (g.flags & VMR_EPI_SLAB) ? (int64_t)ks * g.M * g.ldc : 0;

  // per-lane source offsets (elements) of this wave's two LDS-DMA blocks of each unit kind
  int offA[2][2], offB[2][2];                              // [lo/hi][block]
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int j = wid * 2 + jj;                            // 1-KiB block of the 16-KiB unit
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
      if constexpr (!TR) {
        const int row = 8 * j + (lane >> 3);               // image row (128 B each), 16-B chunk swizzled by the row
        const int c = (lane & 7) ^ (row & 7);
        const int arow = (row >> 6) * 128 + hi * 64 + (row & 63);
        const int w = row & 31, tj = w >> 4, r = w & 15;
        const int brow = (row >> 5) * 64 + hi * 32 + (PERM ? ((r >> 2) * 8 + tj * 4 + (r & 3)) : w);
        offA[hi][jj] = min(m0 + arow, g.M - 1) * (int)g.lda + c * 8;
        offB[hi][jj] = (n0 + brow) * (int)g.ldb + c * 8;
      } else {
        const int r = 4 * j + (lane >> 4);                 // k row of the [64][128] image (256 B each)
        const int ph16 = lane & 15;
        const int col = (((ph16 >> 1) ^ swz_tr(r)) * 2 + (ph16 & 1)) * 8;
        offA[hi][jj] = r * (int)g.lda + m0 + (col >> 6) * 128 + hi * 64 + (col & 63);
        offB[hi][jj] = r * (int)g.ldb + n0 + (col >> 5) * 64 + hi * 32 + (col & 31);
      }
    }
  }
  // unit u -> (K-tile u >> 2, kind u & 3: Alo, Blo, Bhi, Ahi)
  auto stage = [&](int u) {
    const int t = u >> 2, kind = u & 3;
    unsigned char* dst = smem + (t & 1) * P8_BUF + kind * P8_UNIT + wid * 2048;
    const int k0 = k_begin + t * 64;
    const bool isA = kind == 0 || kind == 3;
    const int hi = kind >> 1;                              // kinds 2, 3 are the hi halves
    const bf16_t* base = isA ? A : B;
    const int64_t kadv = TR ? (int64_t)k0 * (isA ? g.lda : g.ldb) : (int64_t)k0;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int off = isA ? offA[hi][jj] : offB[hi][jj];
      __builtin_amdgcn_global_load_lds((gvoid_t*)(base + kadv + off), (lvoid_t*)(dst + jj * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (nk > 0) {
    // prologue: units 0..5 (K-tile 0 and the first two units of K-tile 1)
#pragma unroll
    for (int u = 0; u < 6; ++u)
      if (u < U) stage(u);
    if (U > 6) p8_vmcnt<8>();       // units 0, 1 landed (this wave's parts)
    else p8_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();             // the second wave row runs one barrier behind
    bf16x8 fa[4][2], flo[2][2], fhi[2][2];
    auto mma = [&](int ih, const bf16x8 (&fb)[2][2], int jh) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[ih * 4 + i][jh * 2 + j] =
                mfma16<E>(fb[j][kk], fa[i][kk], acc[ih * 4 + i][jh * 2 + j]);  // C^T tile
      __builtin_amdgcn_s_setprio(0);
    };
    for (int t = 0; t < nk; ++t) {
      const unsigned char* cur = smem + (t & 1) * P8_BUF;
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        const int P = 4 * t + ph;
        // 1. this phase's fragment reads
        if (dbg & 4) {
        } else if (ph == 0) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) flo[j][kk] = read_frag<!TR, 64, true>(cur + 1 * P8_UNIT, wn * 2 + j, kk, lane);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fa[i][kk] = read_frag<!TR, 64, true>(cur + 0 * P8_UNIT, wm * 4 + i, kk, lane);
        } else if (ph == 1) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fhi[j][kk] = read_frag<!TR, 64, true>(cur + 2 * P8_UNIT, wn * 2 + j, kk, lane);
        } else if (ph == 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fa[i][kk] = read_frag<!TR, 64, true>(cur + 3 * P8_UNIT, wm * 4 + i, kk, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        // 2. one unit of a later K-tile, then "everything the next phase reads has landed" (this wave's parts)
        const int u = P + 6;
        if (u < U) {
          if (!(dbg & 1)) stage(u);
          p8_vmcnt<8>();
        } else {
          const int rem = U - 3 - P;                       // units still allowed in flight: 3, 2, 1, 0
          if (rem >= 3) p8_vmcnt<6>();
          else if (rem == 2) p8_vmcnt<4>();
          else if (rem == 1) p8_vmcnt<2>();
          else p8_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (TR) {          // (asm fragment reads: see lds_read_tr)
          if (ph == 0 || ph == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { frag_pin(fa[i][0]); frag_pin(fa[i][1]); }
          }
          if (ph == 0) { frag_pin(flo[0][0]); frag_pin(flo[0][1]); frag_pin(flo[1][0]); frag_pin(flo[1][1]); }
          if (ph == 1) { frag_pin(fhi[0][0]); frag_pin(fhi[0][1]); frag_pin(fhi[1][0]); frag_pin(fhi[1][1]); }
        }
        __builtin_amdgcn_sched_barrier(0);
        // 3. the quadrant
        if (!(dbg & 2)) {
        if (ph == 0) mma(0, flo, 0);
        else if (ph == 1) mma(0, fhi, 1);
        else if (ph == 2) mma(1, fhi, 1);
        else mma(1, flo, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();             // pairs with the second wave row's last barrier
  }
  // register-direct epilogue (the residual quads are requested here: during the K loop their 64 registers hold fragments)
  u32x2 rres[8][4];
  if (g.flags & VMR_EPI_RESIDUAL) {
    const bf16_t* Rsd = reinterpret_cast<const bf16_t*>(g.residual) + coff;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bf16_t* rrow = Rsd + (int64_t)(min(m0 + wm * 128 + (lane & 15) + i * 16, g.M - 1) / g.res_div) * g.ldr + n0 + wn * 64;
      if constexpr (PERM) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const uint4 w = *reinterpret_cast<const uint4*>(rrow + quad_col<true>(2 * k, lane >> 4));
          rres[i][2 * k] = (u32x2){w.x, w.y};
          rres[i][2 * k + 1] = (u32x2){w.z, w.w};
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) rres[i][j] = *reinterpret_cast<const u32x2*>(rrow + quad_col<false>(j, lane >> 4));
      }
    }
  }
  epilogue_direct<E, 8, PERM>(g, acc, wm, wn, lane, m0, n0, 0,
                              (g.flags & VMR_EPI_OUT_F32) ? reinterpret_cast<E*>(reinterpret_cast<float*>(g.C) + coff)
                                                          : reinterpret_cast<E*>(g.C) + coff,
                              rres, reinterpret_cast<E*>(g.aux) + coff);
}

template <typename E, bool TR>
__global__ __launch_bounds__(512, 1) void gemm_e16_p8_kernel(vmr_gemm_t g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  gemm_p8_body<E, TR, !TR>(g, tiles_m, tiles_n, smem, blockIdx.x);
}

// Two independent products in ONE launch: workgroups [0, nblk2) run problem 2 (a split-K weight-gradient product
// dW = dz^T.x writing slabs), the rest problem 1 (an x.W^T-layout product: the input gradient dX = dz.Wt^T).  Single-round
// grids run every workgroup in phase (fill, K loop, epilogue burst); back to back in one grid, the second problem's
// workgroups start their DMA under the first's store drain and one launch gap disappears.
struct ReduceJob {   // dst[i] += sum_k slab[k][i] in 16-byte quads (the split-K second stage of an EARLIER product)
  const float* slab;
  float* dst;
  int nsplit, cols4, nblocks, valid4;   // valid4 > 0: only the first valid4 float4 columns of a slab row have a destination
  int64_t n4, ld4;
};

template <typename E, int MT1>
__global__ __launch_bounds__(256, 2) void gemm_e16_dma2_kernel(vmr_gemm_t g1, int tm1, int tn1, vmr_gemm_t g2, int tm2, int tn2,
                                                                int nblk2, int nblk12, ReduceJob rj) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // the split-K weight-gradient workgroups go first (measured: 9.32 vs 9.37 ms/step the other way round; alternating
  // the two problems in groups of 8 workgroups was worse than either, 9.53)
  // the slab reduction of the PREVIOUS layer's weight gradient takes the first workgroup ids (measured: first 9.08-9.10,
  // last 9.21 ms/step; 256-384 reduction workgroups, 128 or 512+ are slower)
  const int bid = (int)blockIdx.x < rj.nblocks ? nblk12 + (int)blockIdx.x : (int)blockIdx.x - rj.nblocks;
  if (bid < nblk2) gemm_dma_body<E, true, true, 64, 2, 4, 2>(g2, tm2, tn2, smem, bid, 0, 1);
  else if (bid < nblk12) gemm_dma_body<E, false, false, 64, 2, MT1, 2>(g1, tm1, tn1, smem, bid - nblk2, 0, 1);
  else {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(rj.slab);
    f32x4* d4 = reinterpret_cast<f32x4*>(rj.dst);
    for (int64_t i = (int64_t)(bid - nblk12) * 256 + threadIdx.x; i < rj.n4; i += (int64_t)rj.nblocks * 256) {
      if (rj.valid4 && (int)(i % rj.cols4) >= rj.valid4) continue;
      const int64_t o = rj.cols4 ? (i / rj.cols4) * rj.ld4 + (i % rj.cols4) : i;
      // eight slabs per round trip (index-clamped, unconditional loads; the surplus ones are not added): a "load, add"
      // loop over a runtime nsplit compiles to one dependent HBM round trip per slab, and these workgroups then
      // outlast the two products they ride with.  Same summation order as the stand-alone reduction.
      f32x4 acc = d4[o];
      for (int k0 = 0; k0 < rj.nsplit; k0 += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = s4[(int64_t)min(k0 + u, rj.nsplit - 1) * rj.n4 + i];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u < rj.nsplit) { acc[0] += v[u][0]; acc[1] += v[u][1]; acc[2] += v[u][2]; acc[3] += v[u][3]; }
      }
      d4[o] = acc;
    }
  }
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
  const int64_t coff = z1 * g.sC1 + z2 * g.sC2 + ((g.flags & VMR_EPI_SLAB) ? (int64_t)tc.ks * g.M * g.ldc : 0);
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
  epilogue<float, ALIGNED>(g, lds, acc, wm, wn, lane, m0, n0, tc.zb, reinterpret_cast<float*>(g.C) + coff,
                           reinterpret_cast<const float*>(g.residual) + coff,
                           reinterpret_cast<float*>(g.aux) + coff);
}

typedef void (*gemm_fn)(vmr_gemm_t, int, int);
constexpr int SMEM_F32 = (4 * OP_FLOATS32 * 4) > CST_BYTES ? (4 * OP_FLOATS32 * 4) : CST_BYTES;

struct Pick {
  gemm_fn fn;
  int smem;
};

// ------------------------------------------------------------------------------------------------ library switches
// Read ONCE, at the first GEMM call (or set through vmr_debug_set_gemm_*; tests flip them between calls).  Every switch
// selects between kernels that implement the SAME flag set, except `dma == 0`, which turns the LDS-DMA kernels off:
// VMR_EPI_AUX_BITS is then refused by vmr_gemm (vmr_gemm_aux_bits_supported says so first).  The round-2 A/B variants
// that measured as dead ends (BK = 32 rings, 8-wave 128^2 tiles, one-workgroup 256/320 x 128 rings, the ablation build of
// the 8-phase kernel: DESIGN.md 3.1, 3.1d) are no longer in the library.
struct GemmCfg {
  int dma;     // VMR_GEMM_DMA    1 (default): LDS-DMA kernels where the shape allows; 0: register-staged kernels only
  int p8;      // VMR_GEMM_P8     0 never, 1 (default) where 256 x 256 tiles fill >= 85 % of every round, 2 wherever allowed
  int tall;    // VMR_GEMM_TALL   160 x 128 tiles when they save a (partial) round (default 1)
  int wide;    // VMR_GEMM_WIDE   320 x 128 one-workgroup-per-CU tiles for multi-round x.W^T products (default 1)
  int merge;   // VMR_GEMM_MERGE  vmr_gemm2: both products in one launch (default 1)
};
int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
GemmCfg& cfg() {
  static GemmCfg c = {env_int("VMR_GEMM_DMA", 1) != 0 ? 1 : 0, env_int("VMR_GEMM_P8", 1), env_int("VMR_GEMM_TALL", 1),
                      env_int("VMR_GEMM_WIDE", 1), env_int("VMR_GEMM_MERGE", 1)};
  return c;
}

template <typename E, bool AL>
Pick pick_reg16(int ta, int tb) {
  constexpr int sm = Geom16<32>::SMEM;
  if (!ta && !tb) return {(gemm_fn)gemm_e16_kernel<E, false, false, AL, 32>, sm};
  if (!ta && tb) return {(gemm_fn)gemm_e16_kernel<E, false, true, AL, 32>, sm};
  if (ta && !tb) return {(gemm_fn)gemm_e16_kernel<E, true, false, AL, 32>, sm};
  return {(gemm_fn)gemm_e16_kernel<E, true, true, AL, 32>, sm};
}
template <bool AL>
Pick pick_reg32(int ta, int tb) {
  if (!ta && !tb) return {(gemm_fn)gemm_f32_kernel<false, false, AL>, SMEM_F32};
  if (!ta && tb) return {(gemm_fn)gemm_f32_kernel<false, true, AL>, SMEM_F32};
  if (ta && !tb) return {(gemm_fn)gemm_f32_kernel<true, false, AL>, SMEM_F32};
  return {(gemm_fn)gemm_f32_kernel<true, true, AL>, SMEM_F32};
}
Pick pick_reg(int dtype, int ta, int tb, bool al) {
  if (dtype == VMR_F32) return al ? pick_reg32<true>(ta, tb) : pick_reg32<false>(ta, tb);
  if (dtype == VMR_F16) return al ? pick_reg16<f16_t, true>(ta, tb) : pick_reg16<f16_t, false>(ta, tb);
  return al ? pick_reg16<bf16_t, true>(ta, tb) : pick_reg16<bf16_t, false>(ta, tb);
}

// LDS-DMA kernels: BK = 64 x 2 stages.  64 KiB of stages; 67,584 B so the atomic-accumulate epilogue can stage the whole
// fp32 tile in one pass (still 2 / CU)
constexpr int DMA_SMEM = 128 * CST_LD * 4;
constexpr int TALL_SMEM = 2 * (160 * 64 * 2 + 128 * 64 * 2);   // 73,728 B: two workgroups per CU
constexpr int WIDE_SMEM = 2 * (320 + 128) * 128;               // one 512-thread workgroup per CU
template <typename E>
gemm_fn pick_dma_e(int ta, int tb) {
  if (!ta && !tb) return (gemm_fn)gemm_e16_dma_kernel<E, false, false>;
  if (!ta && tb) return (gemm_fn)gemm_e16_dma_kernel<E, false, true>;
  if (ta && !tb) return (gemm_fn)gemm_e16_dma_kernel<E, true, false>;
  return (gemm_fn)gemm_e16_dma_kernel<E, true, true>;
}
gemm_fn pick_dma(int dtype, int ta, int tb) { return dtype == VMR_F16 ? pick_dma_e<f16_t>(ta, tb) : pick_dma_e<bf16_t>(ta, tb); }
template <typename E>
gemm_fn pick_tall_e(int tb) {
  return tb ? (gemm_fn)gemm_e16_dma_kernel<E, false, true, 5> : (gemm_fn)gemm_e16_dma_kernel<E, false, false, 5>;
}
gemm_fn pick_tall(int dtype, int tb) { return dtype == VMR_F16 ? pick_tall_e<f16_t>(tb) : pick_tall_e<bf16_t>(tb); }
// one 512-thread workgroup per CU: 320x128 tiles (A row-major, ragged M allowed)
gemm_fn pick_wide_tall(int dtype) {
  return dtype == VMR_F16 ? (gemm_fn)gemm_e16_dma_kernel<f16_t, false, false, 5, 4> : (gemm_fn)gemm_e16_dma_kernel<bf16_t, false, false, 5, 4>;
}
gemm_fn pick_p8(int dtype, int tr) {
  if (dtype == VMR_F16) return tr ? (gemm_fn)gemm_e16_p8_kernel<f16_t, true> : (gemm_fn)gemm_e16_p8_kernel<f16_t, false>;
  return tr ? (gemm_fn)gemm_e16_p8_kernel<bf16_t, true> : (gemm_fn)gemm_e16_p8_kernel<bf16_t, false>;
}

inline int set_smem_once(const void* fn, int smem) {   // > 64 KiB of dynamic LDS must be opted into once per kernel
  if (smem <= 64 * 1024) return 0;
  static thread_local const void* done[64];
  static thread_local int ndone = 0;
  for (int i = 0; i < ndone; ++i)
    if (done[i] == fn) return 0;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (e != hipSuccess) return vmr_fail(-5, "vmr_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
  if (ndone < 64) done[ndone++] = fn;
  return 0;
}
inline int set_smem_once(gemm_fn fn, int smem) { return set_smem_once(reinterpret_cast<const void*>(fn), smem); }
// rounds of the 512 resident workgroups (2 per CU) a grid needs, in units of one full 128x128 round:
// a last round that leaves every CU at most one workgroup runs about twice as fast
inline double rounds_cost(int64_t tiles, double tile_weight) {
  const int64_t full = tiles / 512, rem = tiles % 512;
  return ((double)full + (rem == 0 ? 0.0 : (rem <= 256 ? 0.5 : 1.0))) * tile_weight;
}

inline bool mult(int64_t v, int64_t m) { return (v % m) == 0; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool gemm_aligned(const vmr_gemm_t& g) {   // 16-byte accesses everywhere
  const int v = vmr_dtype_16(g.dtype) ? 8 : 4;  // elements per 16 B
  bool al = aligned16(g.A) && aligned16(g.B) && aligned16(g.C) && mult(g.lda, v) && mult(g.ldb, v) &&
            mult(g.ldc, 8) && mult(g.sA1, v) && mult(g.sA2, v) && mult(g.sB1, v) && mult(g.sB2, v) &&
            mult(g.sC1, 8) && mult(g.sC2, 8);
  if (g.flags & (VMR_EPI_RESIDUAL | VMR_EPI_AUX)) {
    al = al && mult(g.ldr, 8);
    if (g.flags & VMR_EPI_RESIDUAL) al = al && aligned16(g.residual);
    if (g.flags & VMR_EPI_AUX) al = al && aligned16(g.aux);
  }
  return al;
}
inline bool bias_ok(const vmr_gemm_t& g) {   // the direct epilogue loads bias as float4
  return !(g.flags & VMR_EPI_BIAS) || (aligned16(g.bias) && aligned16(g.bias2));
}

// shapes the 8-phase kernel takes: 16-bit elements, both operands K-contiguous (ragged M allowed) or both stored
// [K][rows] (M, N multiples of 256), N a multiple of 256, K-ranges of whole 64-deep tiles, plain or slab / fp32 stores
bool p8_ok(const vmr_gemm_t& g, int64_t Z) {
  if (!vmr_dtype_16(g.dtype) || Z != 1 || g.transA != g.transB || !gemm_aligned(g)) return false;
  if (g.N % 256 != 0 || g.K % 64 != 0 || g.M % 8 != 0 || g.M < 256) return false;
  if (g.transA && g.M % 256 != 0) return false;
  if (g.flags & (VMR_EPI_ACCUM | VMR_EPI_AUX_BITS)) return false;
  if (g.a_colsum) return false;
  if (!bias_ok(g)) return false;
  const int sk = g.splitk > 1 ? g.splitk : 1;
  if (sk > 1 && !(g.flags & VMR_EPI_SLAB)) return false;
  int64_t chunk = (g.K + sk - 1) / sk;
  chunk = (chunk + 63) / 64 * 64;
  if (chunk * (sk - 1) >= g.K || g.K - chunk * (sk - 1) < 64) return false;          // every split gets at least one K-tile
  // per-lane source offsets are 32-bit element offsets
  const int64_t ea = g.transA ? (int64_t)64 * g.lda + g.M : (int64_t)g.M * g.lda;
  const int64_t eb = g.transB ? (int64_t)64 * g.ldb + g.N : (int64_t)g.N * g.ldb;
  return ea < (1ll << 31) && eb < (1ll << 31);
}
// Measured (scratch/p8_bench.py, same box, us): [9472,3072,1024] 69 vs 84 (320x128 tiles), 4096^3 113 vs 158, 8192^3 827
// vs 1092 -- but [9472,1024,1024] 33 vs 31, [9472,2048,1024] 60 vs 58, [8192,1024,4096] 87 vs 79: 256 x 256 tiles pay only
// when they fill at least 85 % of the CUs of every round (148 tiles of 256 CUs do not).
bool p8_wins(int64_t tiles) {
  const int64_t rounds = (tiles + 255) / 256;
  return (double)tiles / (double)(rounds * 256) >= 0.85;
}
// the interior shapes of the LDS-DMA kernels (128 x 128 tiles)
bool dma_shape_ok(const vmr_gemm_t& g) {
  return cfg().dma && vmr_dtype_16(g.dtype) && gemm_aligned(g) && g.M % BM == 0 && g.N % BN == 0 && g.K % 64 == 0 &&
         g.K >= 128 * (g.splitk > 1 ? g.splitk : 1) && bias_ok(g);
}
// 160-row tiles: A row-major, ragged M allowed
bool tall_shape_ok(const vmr_gemm_t& g) {
  return cfg().dma && cfg().tall && vmr_dtype_16(g.dtype) && gemm_aligned(g) && !g.transA && g.N % BN == 0 && g.K % 64 == 0 &&
         g.K >= 128 && g.splitk <= 1 && !(g.flags & VMR_EPI_ACCUM) && g.M % 8 == 0 && bias_ok(g);
}
// true when vmr_gemm will take a row-major-weight LDS-DMA kernel with the register-direct 16-byte epilogue -- the only
// one that writes VMR_EPI_AUX_BITS (the weight rows are permuted on their way into LDS so that a lane owns 8 consecutive
// columns = one byte of the bit matrix): the 128 / 160 / 320 x 128 tiles of gemm_dma_body with !TB.  The 8-phase kernel
// does not take products with this flag (p8_ok).
bool gemm_perm_direct(const vmr_gemm_t& g) {
  const int64_t Z = (int64_t)(g.Z1 > 0 ? g.Z1 : 1) * (g.Z2 > 0 ? g.Z2 : 1);
  return cfg().dma && vmr_dtype_16(g.dtype) && !g.transA && !g.transB && Z == 1 && g.splitk <= 1 && gemm_aligned(g) &&
         g.M % BM == 0 && g.N % BN == 0 && g.K % 64 == 0 && g.K >= 128 &&
         !(g.flags & (VMR_EPI_ACCUM | VMR_EPI_OUT_F32 | VMR_EPI_SLAB)) && bias_ok(g);
}

int normalise(vmr_gemm_t& g, const char* who) {
  VMR_CHECK(vmr_dtype_ok(g.dtype), "%s: bad dtype %d", who, g.dtype);
  VMR_CHECK(g.M >= 0 && g.N >= 0 && g.K >= 0, "%s: negative dim", who);
  VMR_CHECK(g.A && g.B && g.C, "%s: null operand", who);
  if (g.Z1 <= 0) g.Z1 = 1;
  if (g.Z2 <= 0) g.Z2 = 1;
  if (g.splitk <= 0) g.splitk = 1;
  VMR_CHECK(g.splitk == 1 || (g.flags & (VMR_EPI_ACCUM | VMR_EPI_SLAB)), "%s: splitk>1 needs VMR_EPI_ACCUM or VMR_EPI_SLAB", who);
  if (g.flags & VMR_EPI_SLAB) g.flags |= VMR_EPI_OUT_F32;
  VMR_CHECK(!(g.flags & VMR_EPI_BIAS) || g.bias, "%s: bias flag without pointer", who);
  if (g.bias_scale == 0.f) g.bias_scale = 1.f;
  if (g.res_div <= 0) g.res_div = 1;
  VMR_CHECK(!g.a_colsum || (g.transA && g.Z1 * g.Z2 == 1), "%s: a_colsum needs transA and Z1*Z2 == 1", who);
  VMR_CHECK(g.res_div == 1 || !(g.flags & VMR_EPI_AUX), "%s: res_div with aux (aux shares ldr) is not supported", who);
  VMR_CHECK(!(g.flags & VMR_EPI_RESIDUAL) || g.residual, "%s: residual flag without pointer", who);
  VMR_CHECK(!(g.flags & VMR_EPI_RES_PRE) || (g.flags & VMR_EPI_RESIDUAL), "%s: VMR_EPI_RES_PRE needs VMR_EPI_RESIDUAL", who);
  VMR_CHECK(!(g.flags & VMR_EPI_AUX) || g.aux, "%s: aux flag without pointer", who);
  VMR_CHECK(!(g.flags & VMR_EPI_ROWSCALE) || g.rowscale, "%s: rowscale flag without pointer", who);
  VMR_CHECK(!(g.flags & VMR_EPI_DROPOUT) || (g.drop_p >= 0.f && g.drop_p < 1.f), "%s: bad drop_p", who);
  // the bit-matrix aux store exists in ONE kernel family; anything else would write bf16 tiles into a 16x smaller buffer
  VMR_CHECK(!(g.flags & VMR_EPI_AUX_BITS) || ((g.flags & VMR_EPI_AUX) && gemm_perm_direct(g)),
            "%s: VMR_EPI_AUX_BITS on a product without the register-direct epilogue (ask vmr_gemm_aux_bits_supported)", who);
  const int64_t Z = (int64_t)g.Z1 * g.Z2 * g.splitk;
  VMR_CHECK(Z <= 65535, "%s: too many batches (%lld)", who, (long long)Z);
  VMR_CHECK(g.M == 0 || g.N == 0 ||
                (g.lda >= (g.transA ? g.M : g.K) && g.ldb >= (g.transB ? g.N : g.K) && g.ldc >= g.N),
            "%s: leading dimension too small (lda %lld ldb %lld ldc %lld)", who, (long long)g.lda, (long long)g.ldb,
            (long long)g.ldc);
  return 0;
}

}  // namespace

extern "C" int vmr_gemm(const vmr_gemm_t* gp, void* stream) {
  VMR_CHECK(gp != nullptr, "vmr_gemm: null descriptor");
  vmr_gemm_t g = *gp;
  if (int rc = normalise(g, "vmr_gemm")) return rc;
  if (g.M == 0 || g.N == 0) return 0;
  const GemmCfg& c = cfg();
  const int64_t Z = (int64_t)g.Z1 * g.Z2 * g.splitk;
  const bool al = gemm_aligned(g);
  const int tiles_m = cdiv(g.M, BM), tiles_n = cdiv(g.N, BN);
  const bool dma_ok = dma_shape_ok(g);
  const bool tall_ok = tall_shape_ok(g);
  hipStream_t st = (hipStream_t)stream;

  // (1) 256 x 256 tiles, 8-phase schedule (gemm_p8_body)
  if (c.dma && c.p8 && p8_ok(g, (int64_t)g.Z1 * g.Z2)) {
    const int tm = cdiv(g.M, 256), tn = g.N / 256;
    const int64_t tiles = (int64_t)tm * tn * g.splitk;
    if (c.p8 >= 2 || p8_wins(tiles)) {
      gemm_fn pf = pick_p8(g.dtype, g.transA);
      if (int rc = set_smem_once(pf, P8_SMEM)) return rc;
      hipLaunchKernelGGL(pf, dim3((unsigned)tiles), dim3(512), P8_SMEM, st, g, tm, tn);
      VMR_LAUNCH_CHECK();
      return 0;
    }
  }
  // 160-row tiles when they save a (partial) round: e.g. [9472 x 1024]: 592 tiles = 1.16 rounds of 128x128
  // -> 480 tiles = one round of 160x128
  const double c128 = dma_ok ? rounds_cost((int64_t)tiles_m * tiles_n * Z, 1.0) : 1e30;
  const double c160 = tall_ok ? rounds_cost((int64_t)cdiv(g.M, 160) * tiles_n * Z, 1.25) : 1e30;
  // (2) wide tiles (ONE 512-thread 320x128 workgroup per CU, 256 per round): 22 % fewer L2 -> LDS bytes per flop than two
  // 160x128 workgroups.  Measured (scratch/gemm_ksweep.py): a win only for row-major x.W^T products that need two or
  // more rounds ([9472 x 2048 x 1024]: 65 -> 59 us); single-round grids and transposed-operand layouts lose 3-15 %
  // (one workgroup per CU leaves nothing to run while its 8 waves sit at the K-step barrier), so they never take it.
  if (c.wide && tall_ok && dma_ok && !g.transB && g.M >= 320) {
    const int tm = cdiv(g.M, 320);
    const int64_t rounds = ((int64_t)tm * tiles_n * Z + 255) / 256;
    if (rounds >= 2 && (double)rounds * 1.25 * 0.9 < (c128 < c160 ? c128 : c160) - 1e-9) {
      gemm_fn wf = pick_wide_tall(g.dtype);
      if (int rc = set_smem_once(wf, WIDE_SMEM)) return rc;
      hipLaunchKernelGGL(wf, dim3((unsigned)(tm * tiles_n), 1, (unsigned)Z), dim3(512), WIDE_SMEM, st, g, tm, tiles_n);
      VMR_LAUNCH_CHECK();
      return 0;
    }
  }
  // (3) 160 x 128 tiles, two workgroups per CU
  if (tall_ok && dma_ok && c160 < c128 - 1e-9) {
    const int tm160 = cdiv(g.M, 160);
    gemm_fn tf = pick_tall(g.dtype, g.transB);
    if (int rc = set_smem_once(tf, TALL_SMEM)) return rc;
    hipLaunchKernelGGL(tf, dim3((unsigned)(tm160 * tiles_n), 1, (unsigned)Z), dim3(256), TALL_SMEM, st, g, tm160, tiles_n);
    VMR_LAUNCH_CHECK();
    return 0;
  }
  // (4) 128 x 128 tiles: LDS-DMA interior kernel, or the register-staged kernel for everything else (any shape /
  // alignment / dtype)
  Pick pk = dma_ok ? Pick{pick_dma(g.dtype, g.transA, g.transB), DMA_SMEM} : pick_reg(g.dtype, g.transA, g.transB, al);
  if (int rc = set_smem_once(pk.fn, pk.smem)) return rc;
  dim3 grid((unsigned)(tiles_m * tiles_n), 1, (unsigned)Z);
  if (g.splitk > 1 && g.Z1 * g.Z2 == 1) grid = dim3((unsigned)(tiles_m * tiles_n * g.splitk), 1, 1);   // see tile_coord
  float* colsum_fallback = nullptr;
  if (g.a_colsum && !dma_ok) {   // only the LDS-DMA kernel folds the column sums into the product
    colsum_fallback = g.a_colsum;
    g.a_colsum = nullptr;
  }
  hipLaunchKernelGGL(pk.fn, grid, dim3(256), pk.smem, st, g, tiles_m, tiles_n);
  VMR_LAUNCH_CHECK();
  if (colsum_fallback)   // A is stored [K][M]: a plain column-sum pass (vmr_relu_bwd_bias mode 0 accumulates)
    return vmr_relu_bwd_bias(0, g.A, nullptr, nullptr, colsum_fallback, g.K, g.M, g.lda, 1.0f, g.dtype, 0.f, 0, nullptr, nullptr,
                             1.0f, stream);
  return 0;
}

// Two independent 16-bit products in one launch where both take the single-round LDS-DMA tiles (see
// gemm_e16_dma2_kernel): g1 = an x.W^T-layout product without split-K (the input gradient on the K-major weight copy),
// g2 = a transposed-operand split-K slab product (the weight gradient).  Anything else falls back to two vmr_gemm calls.
extern "C" int vmr_gemm2_reduce(const vmr_gemm_t* p1, const vmr_gemm_t* p2, const float* slab, float* dst, int nsplit, int64_t n,
                                int cols, int64_t ld_dst, void* stream) {
  VMR_CHECK(p1 && p2, "vmr_gemm2: null descriptor");
  VMR_CHECK(!slab || (dst && nsplit > 0 && n % 4 == 0 && (cols == 0 || (cols % 4 == 0 && ld_dst % 4 == 0 && ld_dst > 0 && n % cols == 0))),
            "vmr_gemm2_reduce: bad reduction job");
  vmr_gemm_t g1 = *p1, g2 = *p2;
  auto norm = [](vmr_gemm_t& g) {
    if (g.Z1 <= 0) g.Z1 = 1;
    if (g.Z2 <= 0) g.Z2 = 1;
    if (g.splitk <= 0) g.splitk = 1;
    if (g.flags & VMR_EPI_SLAB) g.flags |= VMR_EPI_OUT_F32;
    if (g.bias_scale == 0.f) g.bias_scale = 1.f;
    if (g.res_div <= 0) g.res_div = 1;
  };
  norm(g1); norm(g2);
  auto basic = [](const vmr_gemm_t& g) {
    return vmr_dtype_16(g.dtype) && g.A && g.B && g.C && g.Z1 * g.Z2 == 1 && g.M > 0 && g.N % BN == 0 && g.K % 64 == 0 &&
           !(g.flags & (VMR_EPI_ACCUM | VMR_EPI_AUX_BITS)) && gemm_aligned(g) && g.lda >= (g.transA ? g.M : g.K) &&
           g.ldb >= (g.transB ? g.N : g.K) && g.ldc >= g.N && (!(g.flags & VMR_EPI_BIAS) || (g.bias && bias_ok(g))) &&
           (!(g.flags & VMR_EPI_RESIDUAL) || g.residual) && (!(g.flags & VMR_EPI_AUX) || g.aux) &&
           (!(g.flags & VMR_EPI_ROWSCALE) || g.rowscale);
  };
  bool ok = cfg().dma && cfg().merge && basic(g1) && basic(g2) && g1.dtype == g2.dtype;
  // the ridden reduction moves float4s: a misaligned job takes the fallback, where vmr_splitk_reduce reports it
  ok = ok && (!slab || (aligned16(slab) && aligned16(dst)));
  // problem 1: row-major operands, no split; 128- or 160-row tiles exactly as vmr_gemm would pick, single round
  int mt1 = 4, tm1 = 0;
  const int tn1 = g1.N / BN;
  if (ok) {
    ok = !g1.transA && !g1.transB && g1.splitk == 1 && g1.K >= 128 && g1.M % 8 == 0 && !(g1.flags & VMR_EPI_SLAB);
    const double c128 = g1.M % BM == 0 ? rounds_cost((int64_t)(g1.M / BM) * tn1, 1.0) : 1e30;
    const double c160 = cfg().tall ? rounds_cost((int64_t)cdiv(g1.M, 160) * tn1, 1.25) : 1e30;
    mt1 = c160 < c128 - 1e-9 ? 5 : 4;
    tm1 = mt1 == 5 ? cdiv(g1.M, 160) : g1.M / BM;
    ok = ok && (mt1 == 5 || g1.M % BM == 0) && (int64_t)tm1 * tn1 <= 512;
  }
  // problem 2: both operands transposed, split-K slabs
  const int tm2 = g2.M / BM, tn2 = g2.N / BN;
  ok = ok && g2.transA && g2.transB && g2.splitk > 1 && (g2.flags & VMR_EPI_SLAB) && g2.M % BM == 0 &&
       g2.K >= 128 * g2.splitk && (int64_t)tm2 * tn2 * g2.splitk <= 1024;
  if (!ok) {
    if (slab)
      if (int rc = vmr_splitk_reduce(slab, dst, nsplit, n, cols, ld_dst, stream)) return rc;
    if (int rc = vmr_gemm(p1, stream)) return rc;
    return vmr_gemm(p2, stream);
  }
  const int nblk1 = tm1 * tn1, nblk2 = tm2 * tn2 * g2.splitk;
  ReduceJob rj;
  memset(&rj, 0, sizeof(rj));
  if (slab && n > 0) {
    rj.slab = slab; rj.dst = dst; rj.nsplit = nsplit; rj.n4 = n / 4; rj.cols4 = cols / 4; rj.ld4 = ld_dst / 4;
    if (cols && ld_dst == cols) rj.cols4 = 0;                  // dense
    rj.valid4 = (cols && ld_dst < cols) ? (int)(ld_dst / 4) : 0;   // zero-padded K columns: see vmr_splitk_reduce
    rj.nblocks = (int)min((int64_t)384, (n / 4 + 255) / 256);
  }
  const int smem = TALL_SMEM;   // >= both variants' two-stage rings (direct epilogues: no staging tile)
  typedef void (*gemm2_fn)(vmr_gemm_t, int, int, vmr_gemm_t, int, int, int, int, ReduceJob);
  gemm2_fn fn;
  if (g1.dtype == VMR_F16) fn = mt1 == 5 ? gemm_e16_dma2_kernel<f16_t, 5> : gemm_e16_dma2_kernel<f16_t, 4>;
  else fn = mt1 == 5 ? gemm_e16_dma2_kernel<bf16_t, 5> : gemm_e16_dma2_kernel<bf16_t, 4>;
  if (int rc = set_smem_once(reinterpret_cast<const void*>(fn), smem)) return rc;
  hipLaunchKernelGGL(fn, dim3(nblk1 + nblk2 + rj.nblocks), dim3(256), smem, (hipStream_t)stream, g1, tm1, tn1, g2, tm2, tn2, nblk2,
                     nblk1 + nblk2, rj);
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_gemm_aux_bits_supported(const vmr_gemm_t* gp) {
  if (!gp) return 0;
  vmr_gemm_t g = *gp;
  if (g.splitk <= 0) g.splitk = 1;
  return (vmr_dtype_16(g.dtype) && gemm_perm_direct(g)) ? 1 : 0;
}

extern "C" int vmr_debug_set_gemm_p8(int mode) {   // -1: back to VMR_GEMM_P8 / the default
  VMR_CHECK(mode >= -1 && mode <= 2, "vmr_debug_set_gemm_p8: mode %d (0 never, 1 by the rounds model, 2 wherever allowed)", mode);
  cfg().p8 = mode < 0 ? env_int("VMR_GEMM_P8", 1) : mode;
  return 0;
}

extern "C" int vmr_debug_set_gemm_dma(int mode) {  // -1: back to VMR_GEMM_DMA / the default
  VMR_CHECK(mode >= -1 && mode <= 1, "vmr_debug_set_gemm_dma: mode %d (0 register-staged kernels only, 1 LDS-DMA kernels)", mode);
  cfg().dma = mode < 0 ? (env_int("VMR_GEMM_DMA", 1) != 0 ? 1 : 0) : mode;
  return 0;
}

extern "C" int vmr_gemm2(const vmr_gemm_t* p1, const vmr_gemm_t* p2, void* stream) {
  return vmr_gemm2_reduce(p1, p2, nullptr, nullptr, 0, 0, 0, 0, stream);
}
