// norm.hip -- LayerNorm and the fused LayerNorm + depthwise-conv encoder kernels.
//
// Reference semantics: nn.LayerNorm over the last dim (biased variance) at
// models/layers.py:85,116,136,271-273,619-620,650-651; the fused kernel is
// `layer_norms[l]` followed by the depthwise Conv1d(k=7, pad=3, groups=D, no
// bias) of DepthwiseSeparableConvBlock (layers.py:126-148).  The convolution is
// NOT masked: padded frames take part, exactly as in the reference.
//
// All of these are HBM-bound row kernels: one 64-lane wave owns one row, 16-B
// vector loads, fp32 statistics, no re-reads (the row lives in registers).
#include "common.h"

namespace {

constexpr int MAXC_MAX = 4;  // chunks of 8 elements per lane => D <= 64*8*4 = 2048 (kernels are
                             // instantiated for 1/2/4 chunks so D <= 1024 rows cost half the registers)

template <typename T, int MAXC>
__device__ __forceinline__ void load_row(const T* __restrict__ p, int D, int lane, float (&v)[MAXC][8]) {
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    // unconditional, index-clamped load, zeroed afterwards by a select: as "if (i < D) load else zero" every chunk got its
    // own branch AND its own s_waitcnt vmcnt(0) inside it -- ln_dwconv_fwd's four rows x two chunks were EIGHT dependent
    // round trips per wave (found with scratch/isa_events.py).  D is a multiple of 8 (16-byte rows).
    const int i = (c * 64 + lane) * 8;
    Vec8<T>::load(p + min(i, D - 8), v[c]);
    const float keep = i < D ? 1.f : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[c][e] = keep != 0.f ? v[c][e] : 0.f;
  }
}

template <int MAXC>
__device__ __forceinline__ void row_stats(const float (&v)[MAXC][8], int D, int lane, float eps, float& mean,
                                          float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[c][e];
  mean = wave_sum_dpp(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = (c * 64 + lane) * 8;
    if (i < D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; q += d * d; }
    }
  }
  rstd = rsqrtf(wave_sum_dpp(q) / (float)D + eps);
}

// ------------------------------------------------------------ LayerNorm fwd
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps,
                                                     const T* __restrict__ pos, int S, T* __restrict__ y,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                     int64_t rows, int D, float drop_p, uint32_t seed0,
                                                     const uint32_t* __restrict__ step) {
  const uint32_t seed = vmr_seed(seed0, step);
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform: scalar branches
  const uint32_t thresh = vmr_drop_thresh(drop_p);
  const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  // gamma / beta of this lane's columns once, and every operand of a row requested together (unconditional, index-
  // clamped): as loads behind the statistics, each under its own column guard, a row was five dependent round trips
  float gam[MAXC][8], bet[MAXC][8];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ic = min((c * 64 + lane) * 8, D - 8);
    Vec8<float>::load(gamma + ic, gam[c]);
    Vec8<float>::load(beta + ic, bet[c]);
  }
  // TWO rows per wave and iteration, both requested before either is reduced (one row at a time left a single row's loads
  // in flight per wave: 3.6 TB/s)
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t row0 = (int64_t)blockIdx.x * 4 + wid; row0 < rows; row0 += 2 * stride) {
    float v[2][MAXC][8], pv[2][MAXC][8];
    int64_t rw[2] = {row0, row0 + stride};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t rc = min(rw[u], rows - 1);
      load_row<T, MAXC>(x + rc * D, D, lane, v[u]);
      if (pos) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c) Vec8<T>::load(pos + (int64_t)(rc % S) * D + min((c * 64 + lane) * 8, D - 8), pv[u][c]);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t row = rw[u];
      if (row >= rows) continue;          // wave-uniform
      float mean, rstd;
      row_stats<MAXC>(v[u], D, lane, eps, mean, rstd);
      if (lane == 0) {
        if (mean_o) mean_o[row] = mean;
        if (rstd_o) rstd_o[row] = rstd;
      }
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        const int i = (c * 64 + lane) * 8;
        if (i >= D) continue;
        const float (&g)[8] = gam[c];
        const float (&b)[8] = bet[c];
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[u][c][e] - mean) * rstd * g[e] + b[e];
        if (pos) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += pv[u][c][e];
        }
        if (drop_p > 0.f) {
          const uint32_t keep = vmr_keep8(seed, (uint64_t)row * D + i, thresh);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = ((keep >> e) & 1) ? o[e] * dscale : 0.f;
        }
        Vec8<T>::store(y + row * D + i, o);
      }
    }
  }
}

// ------------------------------------------------------------ LayerNorm bwd
// A wave owns TWO consecutive rows whose loads are issued together; a 256-thread workgroup covers
// 8 rows, so a [9472 x 1024] tensor is 1184 workgroups of short waves (parallelism, not per-wave
// pipelining, hides the HBM latency here -- the forward kernel reaches 4 TB/s the same way).
// dgamma/dbeta: the 4 waves' partials meet in LDS in a lane-major (conflict-free) order, and one
// partial row per workgroup goes to the second-stage reduction (colreduce_kernel, which undoes
// the permutation).
constexpr int LNB_ROWS = 8;  // rows per workgroup (4 waves x 2 rows)

template <typename T, int MAXC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ mean_i,
                                                     const float* __restrict__ rstd_i, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ part,
                                                     float* __restrict__ dpos, int S,
                                                     int64_t rows, int D, float drop_p, uint32_t seed0,
                                                     const uint32_t* __restrict__ step) {
  const uint32_t seed = vmr_seed(seed0, step);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][MAXC*8][64 lanes]
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform: scalar branches
  const uint32_t thresh = vmr_drop_thresh(drop_p);
  const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  // Persistent waves: wave w of workgroup g walks rows g*4+w, +4*gridDim, ... ; the three operand rows of
  // the NEXT row (x, dy and the residual-branch gradient) are requested -- unconditionally, clamped --
  // before the current row is reduced, and stay in the STORAGE type (4 VGPRs per bf16 chunk) until then,
  // so every wave keeps one full row set in flight while it computes and the register count leaves
  // 4-5 workgroups per CU.  dgamma/dbeta accumulate in registers over all the wave's rows.
  typedef __attribute__((ext_vector_type(8))) T TV8;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t row = (int64_t)blockIdx.x * 4 + wid;
  TV8 cx[MAXC], cg[MAXC], cr[MAXC];
  {
    const int64_t rc = min(row, rows - 1);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int i = (c * 64 + lane) * 8;
      if (i < D) {
        cx[c] = *reinterpret_cast<const TV8*>(x + rc * D + i);
        cg[c] = *reinterpret_cast<const TV8*>(dy + rc * D + i);
        if (dres) cr[c] = *reinterpret_cast<const TV8*>(dres + rc * D + i);
      }
    }
  }
  float ag[MAXC][8], ab[MAXC][8];
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[c][e] = 0.f; ab[c][e] = 0.f; }
  // gamma of this lane's columns, ONCE: as loads inside the row loop (twice per row) they were the YOUNGEST requests in
  // flight, and waiting for them (vmcnt counts in order) drained the next row's prefetch four times per row -- the
  // prefetch overlapped nothing (found with scratch/isa_events.py)
  float gam[MAXC][8];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = (c * 64 + lane) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) gam[c][e] = 0.f;
    if (i < D) Vec8<float>::load(gamma + i, gam[c]);
  }
  while (row < rows) {
    // The next row's operands: inline-asm loads, waited for by hand AFTER this row's arithmetic (see below).  As ordinary
    // loads they were waited for with vmcnt(0) at the first use of the loop-carried registers -- the compiler does not
    // count across the back edge -- i.e. before the arithmetic they were meant to hide under.
    TV8 nx[MAXC], ng[MAXC], nr[MAXC];
    {
      const int64_t rc = min(row + stride, rows - 1);
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        const int i = min((c * 64 + lane) * 8, D - 8);           // (clamped, unconditional: the columns past D are never used)
        if constexpr (sizeof(TV8) == 16) {                       // bf16: one 16-byte load per operand
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nx[c]) : "v"(x + rc * D + i) : "memory");
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ng[c]) : "v"(dy + rc * D + i) : "memory");
          if (dres) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nr[c]) : "v"(dres + rc * D + i) : "memory");
        } else {                                                 // fp32 (the parity path): ordinary loads
          nx[c] = *reinterpret_cast<const TV8*>(x + rc * D + i);
          ng[c] = *reinterpret_cast<const TV8*>(dy + rc * D + i);
          if (dres) nr[c] = *reinterpret_cast<const TV8*>(dres + rc * D + i);
        }
      }
    }
    const float mean = mean_i[row], rstd = rstd_i[row];
    uint32_t keep[MAXC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int i = (c * 64 + lane) * 8;
      if (i >= D) continue;
      const float (&g)[8] = gam[c];
      keep[c] = drop_p > 0.f ? vmr_keep8(seed, (uint64_t)row * D + i, thresh) : 0xFFu;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float d = (float)cg[c][e];
        if (drop_p > 0.f) d = ((keep[c] >> e) & 1) ? d * dscale : 0.f;
        const float h = ((float)cx[c][e] - mean) * rstd;
        ag[c][e] += d * h;
        ab[c][e] += d;
        const float dh = d * g[e];
        s1 += dh;
        s2 += dh * h;
      }
    }
    s1 = wave_sum_dpp(s1) / (float)D;
    s2 = wave_sum_dpp(s2) / (float)D;
    float o[MAXC][8];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int i = (c * 64 + lane) * 8;
      if (i >= D) continue;
      const float (&g)[8] = gam[c];
#pragma unroll
      for (int e = 0; e < 8; ++e) {   // (xhat and dxhat are recomputed rather than kept: 32 VGPRs less)
        float d = (float)cg[c][e];
        if (drop_p > 0.f) d = ((keep[c] >> e) & 1) ? d * dscale : 0.f;
        const float h = ((float)cx[c][e] - mean) * rstd;
        o[c][e] = rstd * (d * g[e] - s1 - h * s2);
        if (dres) o[c][e] += (float)cr[c][e];
      }
    }
    // the next row has had this row's arithmetic to arrive; its registers become valid HERE (before the stores below are
    // issued, so the wait does not cover them: they fly under the next iteration)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if constexpr (sizeof(TV8) == 16) {
        asm volatile("" : "+v"(nx[c]), "+v"(ng[c]));
        if (dres) asm volatile("" : "+v"(nr[c]));
      }
      cx[c] = nx[c]; cg[c] = ng[c]; cr[c] = nr[c];
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int i = (c * 64 + lane) * 8;
      if (i < D) Vec8<T>::store(dx + row * D + i, o[c]);
    }
    row += stride;
  }
  if (!part) return;
  constexpr int SLOTS = MAXC * 8 * 64;   // permuted row: slot (c*8+e)*64 + lane  <->  column (c*64+lane)*8 + e
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(wid * 2 + 0) * SLOTS + (c * 8 + e) * 64 + lane] = ag[c][e];
      red[(wid * 2 + 1) * SLOTS + (c * 8 + e) * 64 + lane] = ab[c][e];
    }
  __syncthreads();
  float* mine = part + (int64_t)blockIdx.x * 2 * SLOTS;
  for (int i = threadIdx.x; i < 2 * SLOTS; i += 256)
    mine[i] = red[i] + red[2 * SLOTS + i] + red[4 * SLOTS + i] + red[6 * SLOTS + i];
}

// Positional-table gradient of "LN(x) + pos[row % S]" (PositionalEmbedding, reference layers.py:96-107,397):
// dpos[s, :] += sum_b d[b*S + s, :] with d = dropout-masked dy.  One thread per (s, 8 channels) walks the batch;
// this replaced one float atomic PER ELEMENT inside ln_bwd (8.4 M atomics, 64 per address: 242 us at cfg2).
template <typename T>
__global__ __launch_bounds__(256) void dpos_kernel(const T* __restrict__ dy, float* __restrict__ dpos, int64_t rows, int S, int D,
                                                   float drop_p, uint32_t seed0, const uint32_t* __restrict__ step, int nsplit) {
  const uint32_t seed = vmr_seed(seed0, step);
  const uint32_t thresh = vmr_drop_thresh(drop_p);
  const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const int cpr = D / 8;
  const int64_t total = (int64_t)S * cpr;
  // nsplit (1, 2, 4 or 8) ADJACENT lanes share one (s, 8 channels) column: lane sp walks batch-row groups sp, sp + nsplit,
  // ... of 8 rows each, the partial sums meet through lane shuffles and lane 0 of the group does the read-modify-write.
  // (One thread per column was a chain of batch/8 dependent HBM round trips: 16 us for a 17 MB operand at B = 64;
  // meeting through float atomics instead -- 8 per address, 32-byte lane stride -- took 28 us.)
  const int64_t total2 = (total * nsplit + 255) / 256 * 256;     // whole workgroups: every lane reaches the shuffles
  for (int64_t idx2 = (int64_t)blockIdx.x * 256 + threadIdx.x; idx2 < total2; idx2 += (int64_t)gridDim.x * 256) {
    const int sp = (int)(idx2 % nsplit);
    const int64_t idx = min(idx2 / nsplit, total - 1);
    const bool live = idx2 / nsplit < total;
    const int s = (int)(idx / cpr), c = (int)(idx - (int64_t)s * cpr) * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((ext_vector_type(8))) T TV8;
    for (int64_t row0 = s + (int64_t)sp * 8 * S; row0 < rows; row0 += (int64_t)nsplit * 8 * S) {   // 8 batch rows in flight
      TV8 g8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) g8[u] = *reinterpret_cast<const TV8*>(dy + min(row0 + (int64_t)u * S, rows - 1) * D + c);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t row = row0 + (int64_t)u * S;
        if (row >= rows) continue;
        const uint32_t keep = drop_p > 0.f ? vmr_keep8(seed, (uint64_t)row * D + c, thresh) : 0xFFu;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float g = (float)g8[u][e];
          acc[e] += (drop_p > 0.f) ? (((keep >> e) & 1) ? g * dscale : 0.f) : g;
        }
      }
    }
    for (int d = 1; d < nsplit; d <<= 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], d, 64);
    }
    if (sp == 0 && live) {
      float old[8];
      Vec8<float>::load(dpos + (int64_t)s * D + c, old);
#pragma unroll
      for (int e = 0; e < 8; ++e) old[e] += acc[e];
      Vec8<float>::store(dpos + (int64_t)s * D + c, old);
    }
  }
}
static inline int dpos_split(int64_t rows, int S) {   // one round of 8 rows per lane where the batch allows: 1, 2, 4 or 8 ways
  const int64_t groups = (rows / S + 7) / 8;
  return groups >= 8 ? 8 : (groups >= 4 ? 4 : (groups >= 2 ? 2 : 1));
}

// y[r, :] = x[r, :] + pos[r % S, :]  (the positional add of FeatureEncoderPredict, reference layers.py:626-631): pos is the
// fp32 table itself -- no compute-dtype copy of it, no broadcast add in the framework
template <typename T>
__global__ __launch_bounds__(256) void add_pos_kernel(const T* __restrict__ x, const float* __restrict__ pos, T* __restrict__ y,
                                                      int64_t rows, int S, int D) {
  const int cpr = D / 8;
  const int64_t total = rows * cpr;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / cpr;
    const int c = (int)(idx - r * cpr) * 8;
    float v[8], p[8];
    Vec8<T>::load(x + r * D + c, v);
    Vec8<float>::load(pos + (int64_t)(r % S) * D + c, p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += p[e];
    Vec8<T>::store(y + r * D + c, v);
  }
}

// out[j] += sum_b part[b][j]   (j < n): second stage of the column reductions.  blockIdx.y takes
// 16 partial rows (16 independent loads in flight per thread), so only nblocks/16 adders meet on
// an address.
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ part, float* __restrict__ out0,
                                                        float* __restrict__ out1, int nblocks, int n0, int n1,
                                                        int slots /*0: plain rows; else ln_bwd's permuted rows*/) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int n = n0 + n1;
  if (j >= n) return;
  int64_t src = j, stride = n;
  if (slots) {  // column jj of half h sits at slot ((c*8+e)*64 + lane) with jj = (c*64+lane)*8 + e
    const int h = j >= n0, jj = h ? j - n0 : j;
    const int e = jj & 7, cl = jj >> 3, lane = cl & 63, c = cl >> 6;
    src = (int64_t)h * slots + (c * 8 + e) * 64 + lane;
    stride = 2 * (int64_t)slots;
  }
  const int b0 = blockIdx.y * 16;
  float v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = (b0 + k < nblocks) ? part[(int64_t)(b0 + k) * stride + src] : 0.f;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += v[k];
  atomicAdd(j < n0 ? &out0[j] : &out1[j - n0], s);
}

// ----------------------------------------------- fused LayerNorm + dwconv fwd
// Workgroup = (sample b, tile of R output rows).  Phase 1: the 4 waves normalise
// the R+6 rows (3-row halo each side) into LDS; rows outside [0,S) are the
// conv's zero padding.  Phase 2: each thread owns 4 channels and produces the
// R output rows from the LDS tile.
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void ln_dwconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            const float* __restrict__ w, T* __restrict__ u,
                                                            float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                            int S, int D, int R, int tiles, int nb1, int S2,
                                                            int R2, int tiles2, int64_t rows1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* tile = reinterpret_cast<T*>(smem);  // [R+6][D]
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform: scalar branches
  // second sequence group (the query sentences behind the video clips): workgroups >= nb1
  int bid = blockIdx.x;
  if (bid >= nb1) {
    bid -= nb1; S = S2; R = R2; tiles = tiles2;
    x += rows1 * D; u += rows1 * D; mean_o += rows1; rstd_o += rows1;
  }
  const int b = bid / tiles, s0 = (bid % tiles) * R;
  const int nrows = R + 6;
  constexpr int RW = 4;   // rows a wave keeps in flight: their (unconditional, index-clamped) loads issue together
  for (int rb = wid; rb < nrows; rb += 4 * RW) {
    float v[RW][MAXC][8];
    float mean[RW], rstd[RW], msk[RW];
#pragma unroll
    for (int u = 0; u < RW; ++u) {
      const int r = rb + 4 * u, s = s0 - 3 + r;
      msk[u] = (r < nrows && s >= 0 && s < S) ? 1.f : 0.f;   // zero padding of the conv / unused slots
      const int sc = min(max(s, 0), S - 1);
      load_row<T, MAXC>(x + ((int64_t)b * S + sc) * D, D, lane, v[u]);
    }
#pragma unroll
    for (int u = 0; u < RW; ++u) {
      const int r = rb + 4 * u, s = s0 - 3 + r;
      row_stats<MAXC>(v[u], D, lane, eps, mean[u], rstd[u]);
      if (lane == 0 && msk[u] != 0.f && r >= 3 && r < 3 + R) {
        mean_o[(int64_t)b * S + s] = mean[u];
        rstd_o[(int64_t)b * S + s] = rstd[u];
      }
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int i = (c * 64 + lane) * 8;
      if (i >= D) continue;
      float g[8], bb[8];         // (kept inside: hoisting them costs 22 VGPRs = one resident wave per SIMD: 17.5 -> 18.8 us)
      Vec8<float>::load(gamma + i, g);
      Vec8<float>::load(beta + i, bb);
#pragma unroll
      for (int u = 0; u < RW; ++u) {
        const int r = rb + 4 * u;
        if (r >= nrows) continue;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = ((v[u][c][e] - mean[u]) * rstd[u] * g[e] + bb[e]) * msk[u];
        Vec8<T>::store(tile + (int64_t)r * D + i, o);
      }
    }
  }
  __syncthreads();
  for (int cg = threadIdx.x; cg * 4 < D; cg += 256) {
    const int c0 = cg * 4;
    float wk[4][7];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
#pragma unroll
      for (int k = 0; k < 7; ++k) wk[ch][k] = w[(c0 + ch) * 7 + k];
    for (int r = 0; r < R; ++r) {
      const int s = s0 + r;
      if (s >= S) break;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        float nv[4];
        Vec4<T>::load(tile + (int64_t)(r + k) * D + c0, nv);
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) acc[ch] += wk[ch][k] * nv[ch];
      }
      Vec4<T>::store(u + ((int64_t)b * S + s) * D + c0, acc);
    }
  }
}

// ------------------------------------------------------------- dwconv bwd
// Workgroup = (sample b, slice of 128 channels, 4 x 16 rows); wave w streams 16 rows (+3-row
// halos) keeping 7-row windows of du and n=LN(x) in registers; each lane owns 2 channels (small
// register footprint -> many short waves resident: this kernel is latency-, not bandwidth-bound).
// dn[s] = sum_k w[k]*du[s-k+3];  dw[k] += sum_s du[s]*n[s+k-3].  The 4 waves' dw partials are
// combined in LDS and stored as one partial row per workgroup (colreduce_kernel sums them).
#ifndef VMR_DWB_CH
#define VMR_DWB_CH 2
#endif
constexpr int DWB_CH = VMR_DWB_CH, DWB_SLICE = 64 * DWB_CH;
template <typename T, int N> struct VecN;
template <typename T> struct VecN<T, 2> : Vec2<T> {};
template <typename T> struct VecN<T, 4> : Vec4<T> {};

template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const T* __restrict__ du, const T* __restrict__ x,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         const float* __restrict__ mean_i,
                                                         const float* __restrict__ rstd_i,
                                                         const float* __restrict__ w, T* __restrict__ dn,
                                                         float* __restrict__ part, int S, int D, int slices,
                                                         int bps /*workgroups per (sample, slice)*/, int nb1,
                                                         int S2, int bps2, int64_t rows1, int prow1) {
  constexpr int CHN = DWB_CH;
  __shared__ float red[4][DWB_SLICE * 7];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform: scalar branches
  // second sequence group: workgroups >= nb1; its partial rows follow the first group's
  int bid = blockIdx.x;
  if (bid >= nb1) {
    bid -= nb1; S = S2; bps = bps2;
    du += rows1 * D; x += rows1 * D; dn += rows1 * D; mean_i += rows1; rstd_i += rows1;
    part += (int64_t)prow1 * D * 7;
  }
  const int j4 = bid % bps, bs = bid / bps;
  const int b = bs / slices, c0 = (bs % slices) * DWB_SLICE + lane * CHN;
  const bool act = c0 < D;
  constexpr int seg = 16;   // rows per wave
  const int sb = min(S, (j4 * 4 + wid) * seg), se = min(S, sb + seg);
  float wk[CHN][7], aw[CHN][7], g[CHN], bt[CHN];
#pragma unroll
  for (int ch = 0; ch < CHN; ++ch) {
    g[ch] = act ? gamma[c0 + ch] : 0.f;
    bt[ch] = act ? beta[c0 + ch] : 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) { wk[ch][k] = act ? w[(c0 + ch) * 7 + k] : 0.f; aw[ch][k] = 0.f; }
  }
  float wdu[7][CHN], wn[7][CHN];   // window index j holds row (cur - 3 + j)
  // branch-free: out-of-range rows / channels load a clamped (valid) address and are zeroed by a
  // select, so the 16 loads of a chunk issue back to back instead of one predicated region each
  const int c0c = act ? c0 : 0;
  auto fetch = [&](int s, float (&odu)[CHN], float (&on)[CHN]) {
    const bool valid = act && s >= 0 && s < S;
    const int sc = min(max(s, 0), S - 1);
    const int64_t off = ((int64_t)b * S + sc) * D + c0c;
    const float mean = mean_i[(int64_t)b * S + sc], rstd = rstd_i[(int64_t)b * S + sc];
    float xv[CHN], dv[CHN];
    VecN<T, CHN>::load(du + off, dv);
    VecN<T, CHN>::load(x + off, xv);
    // multiply by a 0/1 mask (NOT a select): the compiler must keep the loads unconditional, so all
    // loads of a chunk are in flight together; a select lets it sink each load into its own branch
    const float m = valid ? 1.f : 0.f;
#pragma unroll
    for (int ch = 0; ch < CHN; ++ch) {
      odu[ch] = dv[ch] * m;
      on[ch] = ((xv[ch] - mean) * rstd * g[ch] + bt[ch]) * m;
    }
  };
#pragma unroll
  for (int j = 0; j < 6; ++j) fetch(sb - 3 + j, wdu[j + 1], wn[j + 1]);
  constexpr int CHK = 8;  // rows whose loads are issued together
  for (int s0 = sb; s0 < se; s0 += CHK) {
    float ndu[CHK][CHN], nn[CHK][CHN];
#pragma unroll
    for (int j = 0; j < CHK; ++j) fetch(s0 + 3 + j, ndu[j], nn[j]);
#pragma unroll
    for (int j = 0; j < CHK; ++j) {
      const int s = s0 + j;
      if (s < se) {
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int ch = 0; ch < CHN; ++ch) { wdu[t][ch] = wdu[t + 1][ch]; wn[t][ch] = wn[t + 1][ch]; }
#pragma unroll
        for (int ch = 0; ch < CHN; ++ch) { wdu[6][ch] = ndu[j][ch]; wn[6][ch] = nn[j][ch]; }
        float o[CHN];
#pragma unroll
        for (int ch = 0; ch < CHN; ++ch) {
          float a = 0.f;
#pragma unroll
          for (int k = 0; k < 7; ++k) a += wk[ch][k] * wdu[6 - k][ch];   // dn[s] = sum_k w[k]*du[s+3-k]
          o[ch] = a;
#pragma unroll
          for (int k = 0; k < 7; ++k) aw[ch][k] += wdu[3][ch] * wn[k][ch];  // du[s] * n[s+k-3]
        }
        if (act) VecN<T, CHN>::store(dn + ((int64_t)b * S + s) * D + c0, o);
      }
    }
  }
#pragma unroll
  for (int ch = 0; ch < CHN; ++ch)
#pragma unroll
    for (int k = 0; k < 7; ++k) red[wid][(lane * CHN + ch) * 7 + k] = aw[ch][k];
  __syncthreads();
  // per-workgroup partials [b*bps + j4][D*7] (plain stores); colreduce_kernel sums them
  const int cbase = (bs % slices) * DWB_SLICE;
  float* mine = part + ((int64_t)b * bps + j4) * D * 7;
  for (int i = threadIdx.x; i < DWB_SLICE * 7; i += 256) {
    if (cbase + i / 7 < D) mine[(int64_t)cbase * 7 + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
  }
}

int ln_check(int D) {
  if (D % 8 != 0 || D > 64 * 8 * MAXC_MAX || D <= 0)
    return vmr_fail(-22, "LayerNorm kernels need D %% 8 == 0 and D <= %d (got %d)", 64 * 8 * MAXC_MAX, D);
  return 0;
}

// dispatch on the number of 8-element chunks a lane needs for a row of D elements
#define LN_DISPATCH(D, ...)                                       \
  do {                                                            \
    if ((D) <= 512) { constexpr int MC = 1; __VA_ARGS__; }        \
    else if ((D) <= 1024) { constexpr int MC = 2; __VA_ARGS__; }  \
    else { constexpr int MC = 4; __VA_ARGS__; }                   \
  } while (0)

}  // namespace

extern "C" int vmr_layernorm_fwd(const void* x, const float* gamma, const float* beta, float eps, const void* pos,
                                 int S, void* y, float* mean, float* rstd, int64_t rows, int D, int dtype,
                                 float drop_p, uint32_t drop_seed, const uint32_t* drop_step, void* stream) {
  if (int rc = ln_check(D)) return rc;
  VMR_CHECK(x && gamma && beta && y, "vmr_layernorm_fwd: null pointer");
  VMR_CHECK(!pos || S > 0, "vmr_layernorm_fwd: pos needs S > 0");
  if (rows == 0) return 0;
  const int grid = (int)min((int64_t)4096, (rows + 7) / 8);      // (two rows per wave and pass)
  VMR_DISPATCH(dtype, T, LN_DISPATCH(D, hipLaunchKernelGGL((ln_fwd_kernel<T, MC>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                      (const T*)x, gamma, beta, eps, (const T*)pos, S, (T*)y, mean, rstd,
                                      rows, D, drop_p, drop_seed, drop_step)));
  VMR_LAUNCH_CHECK();
  return 0;
}

static int ln_bwd_impl(const void* dy, const void* x, const float* gamma, const float* mean,
                       const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta,
                       float* dpos, float* workspace, int S, int64_t rows, int D, int dtype, float drop_p,
                       uint32_t drop_seed, const uint32_t* drop_step, void* stream, bool defer, int* nblocks) {
  if (int rc = ln_check(D)) return rc;
  VMR_CHECK(dy && x && gamma && mean && rstd && dx, "vmr_layernorm_bwd: null pointer");
  VMR_CHECK((dgamma == nullptr) == (dbeta == nullptr), "vmr_layernorm_bwd: dgamma/dbeta must come together");
  VMR_CHECK(!(dgamma || defer) || workspace, "vmr_layernorm_bwd: dgamma/dbeta need the workspace (VMR_LN_BWD_WS_FLOATS(rows, D))");
  if (nblocks) *nblocks = 0;
  if (rows == 0) return 0;
  // persistent grid: at most g_lnb_grid workgroups (never more than ceil(rows/8): the workspace bound)
  static int g_lnb_grid = 0;
  if (g_lnb_grid == 0) {
    const char* e = getenv("VMR_LNB_GRID");
    g_lnb_grid = e && atoi(e) > 0 ? atoi(e) : 768;   // 3 resident workgroups x 256 CUs at D = 1024
  }
  const int grid = (int)min((int64_t)min(g_lnb_grid, VMR_LN_BWD_MAX_BLOCKS), (rows + LNB_ROWS - 1) / LNB_ROWS);
  float* part = (dgamma || defer) ? workspace : nullptr;
  if (nblocks) *nblocks = grid;
  const int maxc = D <= 512 ? 1 : (D <= 1024 ? 2 : 4);
  const int slots = maxc * 8 * 64;
  const size_t lds = (size_t)4 * 2 * slots * sizeof(float);
  VMR_DISPATCH(dtype, T, LN_DISPATCH(D, hipLaunchKernelGGL((ln_bwd_kernel<T, MC>), dim3(grid), dim3(256), lds, (hipStream_t)stream,
                                      (const T*)dy, (const T*)x, gamma, mean, rstd, (const T*)dres,
                                      (T*)dx, part, dpos, S, rows, D, drop_p, drop_seed, drop_step)));
  VMR_LAUNCH_CHECK();
  if (dpos) {
    VMR_CHECK(S > 0 && D % 8 == 0, "vmr_layernorm_bwd: dpos needs S > 0");
    const int nsp = dpos_split(rows, S);
    const dim3 gp((unsigned)min((int64_t)4096, ((int64_t)S * (D / 8) * nsp + 255) / 256));
    VMR_DISPATCH(dtype, T, hipLaunchKernelGGL(dpos_kernel<T>, gp, dim3(256), 0, (hipStream_t)stream, (const T*)dy, dpos, rows, S, D, drop_p,
                         drop_seed, drop_step, nsp));
    VMR_LAUNCH_CHECK();
  }
  if (part && !defer) {
    hipLaunchKernelGGL(colreduce_kernel, dim3(cdiv(2 * D, 256), cdiv(grid, 16)), dim3(256), 0, (hipStream_t)stream,
                       part, dgamma, dbeta, grid, D, D, slots);
    VMR_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int vmr_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                 const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta,
                                 float* dpos, float* workspace, int S, int64_t rows, int D, int dtype, float drop_p,
                                 uint32_t drop_seed, const uint32_t* drop_step, void* stream) {
  return ln_bwd_impl(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, dpos, workspace, S, rows, D, dtype, drop_p, drop_seed,
                     drop_step, stream, false, nullptr);
}

extern "C" int vmr_layernorm_bwd_deferred(const void* dy, const void* x, const float* gamma, const float* mean,
                                          const float* rstd, const void* dres, void* dx, float* dpos, float* workspace,
                                          int S, int64_t rows, int D, int dtype, float drop_p, uint32_t drop_seed,
                                          const uint32_t* drop_step, int32_t* nblocks, void* stream) {
  VMR_CHECK(nblocks, "vmr_layernorm_bwd_deferred: null nblocks");
  return ln_bwd_impl(dy, x, gamma, mean, rstd, dres, dx, nullptr, nullptr, dpos, workspace, S, rows, D, dtype, drop_p,
                     drop_seed, drop_step, stream, true, nblocks);
}

// tile height for one sequence group: 16-row LDS tiles (10 output rows + halo): 32 KiB at D=1024 bf16
// -> 4 workgroups / CU, 13 tiles per 128-frame clip; the extra halo re-reads are L2 hits, the
// parallelism hides the row latency
static void dwconv_tiling(int S, int D, size_t esz, int& R, int& tiles) {
  static int cap = -1;
  if (cap < 0) {
    const char* e = getenv("VMR_DWCONV_ROWS");   // LDS tile rows (output rows + 6 halo rows); measured best: see DESIGN
    cap = e ? atoi(e) : 16;
  }
  const int nrows = (int)min((size_t)cap, (size_t)(128 * 1024) / ((size_t)D * esz));
  if (S <= 0 || nrows < 7) { R = 0; tiles = 0; return; }
  R = min(nrows - 6, S);
  tiles = cdiv(S, R);
  R = cdiv(S, tiles);  // balance the tiles
}

// convblock.hip: the persistent form for D = 512 / 1024 (-1: not taken)
int convblock_fwd_launch(const void* x, const float* gamma, const float* beta, float eps, const float* w, void* u, float* mean,
                         float* rstd, int B1, int S1, int B2, int S2, int D, int dtype, void* stream);

extern "C" int vmr_ln_dwconv_fwd2(const void* x, const float* gamma, const float* beta, float eps, const float* w,
                                  void* u, float* mean, float* rstd, int B1, int S1, int B2, int S2, int D,
                                  int dtype, void* stream) {
  if (int rc = ln_check(D)) return rc;
  VMR_CHECK(x && gamma && beta && w && u && mean && rstd, "vmr_ln_dwconv_fwd: null pointer");
  VMR_CHECK(B1 >= 0 && S1 >= 0 && B2 >= 0 && S2 >= 0, "vmr_ln_dwconv_fwd: negative shape");
  if (B1 == 0 || S1 == 0) { B1 = 0; S1 = S1 > 0 ? S1 : 1; }
  if (B2 == 0 || S2 == 0) { B2 = 0; S2 = S2 > 0 ? S2 : 1; }
  if (B1 + B2 == 0) return 0;
  if (const int rc = convblock_fwd_launch(x, gamma, beta, eps, w, u, mean, rstd, B1, S1, B2, S2, D, dtype, stream); rc != -1) return rc;
  const size_t esz = (size_t)vmr_dtype_size(dtype);
  int R1, tiles1, R2, tiles2;
  dwconv_tiling(S1, D, esz, R1, tiles1);
  dwconv_tiling(S2, D, esz, R2, tiles2);
  VMR_CHECK(R1 > 0 && R2 > 0, "vmr_ln_dwconv_fwd: D too large for the LDS tile");
  size_t lds = (size_t)(max(R1, R2) + 6) * D * esz;
  {   // (A/B switch: a larger LDS request lowers the resident workgroups per CU, so the grid runs in rounds whose load
      //  and store bursts overlap)
    static int pad = -1;
    if (pad < 0) {
      const char* e = getenv("VMR_DWCONV_LDS_KB");
      pad = e ? atoi(e) : 0;
    }
    if ((size_t)pad * 1024 > lds) lds = (size_t)pad * 1024;
  }
  const void* fn = nullptr;
  VMR_DISPATCH(dtype, T, LN_DISPATCH(D, fn = (const void*)ln_dwconv_fwd_kernel<T, MC>));
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return vmr_fail(-5, "vmr_ln_dwconv_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const int nb1 = B1 * tiles1, nb = nb1 + B2 * tiles2;
  const int64_t rows1 = (int64_t)B1 * S1;
  VMR_DISPATCH(dtype, T, LN_DISPATCH(D, hipLaunchKernelGGL((ln_dwconv_fwd_kernel<T, MC>), dim3(nb), dim3(256), lds,
                                      (hipStream_t)stream, (const T*)x, gamma, beta, eps, w, (T*)u, mean,
                                      rstd, S1, D, R1, tiles1, nb1, S2, R2, tiles2, rows1)));
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_ln_dwconv_fwd(const void* x, const float* gamma, const float* beta, float eps, const float* w,
                                 void* u, float* mean, float* rstd, int B, int S, int D, int dtype, void* stream) {
  return vmr_ln_dwconv_fwd2(x, gamma, beta, eps, w, u, mean, rstd, B, S, 0, 0, D, dtype, stream);
}

static int dwconv_bwd2_impl(const void* du, const void* x, const float* gamma, const float* beta,
                            const float* mean, const float* rstd, const float* w, void* dn, float* dw,
                            float* workspace, int B1, int S1, int B2, int S2, int D, int dtype, void* stream, bool defer,
                            int* nblocks) {
  VMR_CHECK(du && x && gamma && beta && mean && rstd && w && dn && (dw || defer) && workspace, "vmr_dwconv_bwd: null pointer");
  if (nblocks) *nblocks = 0;
  VMR_CHECK(D % 2 == 0, "vmr_dwconv_bwd: D %% 2 != 0");
  VMR_CHECK(B1 >= 0 && S1 >= 0 && B2 >= 0 && S2 >= 0, "vmr_dwconv_bwd: negative shape");
  if (B1 == 0 || S1 == 0) { B1 = 0; S1 = S1 > 0 ? S1 : 1; }
  if (B2 == 0 || S2 == 0) { B2 = 0; S2 = S2 > 0 ? S2 : 1; }
  if (B1 + B2 == 0) return 0;
  const int slices = cdiv(D, DWB_SLICE);
  const int bps1 = VMR_DWCONV_BWD_BPS(S1), bps2 = VMR_DWCONV_BWD_BPS(S2);
  const int nb1 = B1 * slices * bps1, nb = nb1 + B2 * slices * bps2;
  const int prow1 = B1 * bps1, prows = prow1 + B2 * bps2;
  const int64_t rows1 = (int64_t)B1 * S1;
  VMR_DISPATCH(dtype, T, hipLaunchKernelGGL(dwconv_bwd_kernel<T>, dim3(nb), dim3(256), 0, (hipStream_t)stream,
                       (const T*)du, (const T*)x, gamma, beta, mean, rstd, w, (T*)dn, workspace, S1, D,
                       slices, bps1, nb1, S2, bps2, rows1, prow1));
  VMR_LAUNCH_CHECK();
  if (nblocks) *nblocks = prows;
  if (defer) return 0;
  hipLaunchKernelGGL(colreduce_kernel, dim3(cdiv(D * 7, 256), cdiv(prows, 16)), dim3(256), 0, (hipStream_t)stream,
                     workspace, dw, dw, prows, D * 7, 0, 0);
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_dwconv_bwd2(const void* du, const void* x, const float* gamma, const float* beta,
                               const float* mean, const float* rstd, const float* w, void* dn, float* dw,
                               float* workspace, int B1, int S1, int B2, int S2, int D, int dtype, void* stream) {
  return dwconv_bwd2_impl(du, x, gamma, beta, mean, rstd, w, dn, dw, workspace, B1, S1, B2, S2, D, dtype, stream, false, nullptr);
}

extern "C" int vmr_dwconv_bwd2_deferred(const void* du, const void* x, const float* gamma, const float* beta,
                                        const float* mean, const float* rstd, const float* w, void* dn, float* workspace,
                                        int B1, int S1, int B2, int S2, int D, int dtype, int32_t* nblocks, void* stream) {
  VMR_CHECK(nblocks, "vmr_dwconv_bwd2_deferred: null nblocks");
  return dwconv_bwd2_impl(du, x, gamma, beta, mean, rstd, w, dn, nullptr, workspace, B1, S1, B2, S2, D, dtype, stream, true, nblocks);
}

// ---- batched second stage: every deferred column reduction of a backward pass in ONE launch (blockIdx.z = item)
struct ColItems {
  vmr_colreduce_item_t it[VMR_COLREDUCE_MAX_ITEMS];
};

__global__ __launch_bounds__(256) void colreduce_batched_kernel(ColItems items) {
  const vmr_colreduce_item_t& q = items.it[blockIdx.z];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int n = q.n0 + q.n1;
  const int b0 = blockIdx.y * 16;
  if (j >= n || b0 >= q.nblocks) return;
  int64_t src = j, stride = n;
  if (q.slots) {  // ln_bwd's permuted partial rows (see colreduce_kernel)
    const int h = j >= q.n0, jj = h ? j - q.n0 : j;
    const int e = jj & 7, cl = jj >> 3, lane = cl & 63, c = cl >> 6;
    src = (int64_t)h * q.slots + (c * 8 + e) * 64 + lane;
    stride = 2 * (int64_t)q.slots;
  }
  float v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = (b0 + k < q.nblocks) ? q.part[(int64_t)(b0 + k) * stride + src] : 0.f;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += v[k];
  atomicAdd(j < q.n0 ? &q.out0[j] : &q.out1[j - q.n0], s);
}

extern "C" int vmr_colreduce_batched(const vmr_colreduce_item_t* items, int n, void* stream) {
  VMR_CHECK(items || n == 0, "vmr_colreduce_batched: null items");
  for (int base = 0; base < n; base += VMR_COLREDUCE_MAX_ITEMS) {
    const int cnt = min(VMR_COLREDUCE_MAX_ITEMS, n - base);
    ColItems ci;
    memset(&ci, 0, sizeof(ci));
    int maxn = 0, maxb = 0;
    for (int i = 0; i < cnt; ++i) {
      const vmr_colreduce_item_t& q = items[base + i];
      VMR_CHECK(q.part && q.out0 && (q.out1 || q.n1 == 0) && q.nblocks >= 0 && q.n0 >= 0 && q.n1 >= 0,
                "vmr_colreduce_batched: bad item %d", base + i);
      ci.it[i] = q;
      maxn = max(maxn, q.n0 + q.n1);
      maxb = max(maxb, q.nblocks);
    }
    if (maxn == 0 || maxb == 0) continue;
    hipLaunchKernelGGL(colreduce_batched_kernel, dim3(cdiv(maxn, 256), cdiv(maxb, 16), cnt), dim3(256), 0, (hipStream_t)stream, ci);
    VMR_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int vmr_dwconv_bwd(const void* du, const void* x, const float* gamma, const float* beta,
                              const float* mean, const float* rstd, const float* w, void* dn, float* dw,
                              float* workspace, int B, int S, int D, int dtype, void* stream) {
  return vmr_dwconv_bwd2(du, x, gamma, beta, mean, rstd, w, dn, dw, workspace, B, S, 0, 0, D, dtype, stream);
}


extern "C" int vmr_add_pos_fwd(const void* x, const float* pos, void* y, int64_t rows, int S, int D, int dtype, void* stream) {
  VMR_CHECK(x && pos && y, "vmr_add_pos_fwd: null pointer");
  VMR_CHECK(S > 0 && D % 8 == 0 && rows >= 0, "vmr_add_pos_fwd: need S > 0 and D %% 8 == 0");
  if (rows == 0) return 0;
  const unsigned grid = (unsigned)min((int64_t)4096, (rows * (D / 8) + 255) / 256);
  VMR_DISPATCH(dtype, T, hipLaunchKernelGGL(add_pos_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)x, pos, (T*)y, rows, S, D));
  VMR_LAUNCH_CHECK();
  return 0;
}

// dpos[s, :] += sum_b dy[b*S + s, :]  (ACCUMULATES: dpos may be the table's slot in the gradient arena)
extern "C" int vmr_add_pos_bwd(const void* dy, float* dpos, int64_t rows, int S, int D, int dtype, void* stream) {
  VMR_CHECK(dy && dpos, "vmr_add_pos_bwd: null pointer");
  VMR_CHECK(S > 0 && D % 8 == 0 && rows >= 0, "vmr_add_pos_bwd: need S > 0 and D %% 8 == 0");
  if (rows == 0) return 0;
  const int nsp = dpos_split(rows, S);
  const dim3 gp((unsigned)min((int64_t)4096, ((int64_t)S * (D / 8) * nsp + 255) / 256));
  VMR_DISPATCH(dtype, T, hipLaunchKernelGGL(dpos_kernel<T>, gp, dim3(256), 0, (hipStream_t)stream, (const T*)dy, dpos, rows, S, D, 0.f, 0u, nullptr, nsp));
  VMR_LAUNCH_CHECK();
  return 0;
}
