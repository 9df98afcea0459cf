// softmax.hip -- masked row softmax (+dropout) for the attention scores, fwd/bwd.
//
// mode 0: DualMultiAttention (reference models/layers.py:346-357): z=(b,h),
//         x = scale*S + (1 - rmask[b,r]*cmask[b,c]) * -1e30.  A fully masked
//         query row gives x == -1e30 everywhere -> a uniform softmax, exactly as
//         in the reference (never NaN).
// mode 1: TopSelfAttention2 (layers.py:567-574): nn.MultiheadAttention fed a
//         FLOAT key_padding_mask, which PyTorch ADDS to the logits: z=(t,h),
//         x = scale*S + cmask[c*cm_stride + t].
// One 64-lane wave per row, up to 16 columns per lane (C <= 1024); fp32 math;
// P is written in the activation dtype with the row zero-padded up to ldP so the
// following P.V GEMM can use 16-B loads.
#include "common.h"

namespace {

constexpr int MAXN_MAX = 16;  // kernels are instantiated for 1/2/4/8/16 columns per lane

template <typename T, int MAXN>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, T* __restrict__ P,
                                                          T* __restrict__ Pkeep, const float* __restrict__ rmask,
                                                          const float* __restrict__ cmask, int mode, int64_t nrows,
                                                          int H, int R, int C, int ldS, int ldP, int cm_stride,
                                                          float scale, float drop_p, uint32_t seed0,
                                                          const uint32_t* __restrict__ step) {
  const uint32_t seed = vmr_seed(seed0, step);
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t thresh = vmr_drop_thresh(drop_p);
  const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wid; row < nrows; row += (int64_t)gridDim.x * 4) {
    const int z = (int)(row / R), r = (int)(row - (int64_t)z * R);
    const int zo = z / H;  // b (mode 0) or t (mode 1)
    const float* s = S + row * ldS;
    float rm = 1.f;
    if (mode == 0) rm = rmask[(int64_t)zo * R + r];
    float v[MAXN];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
      const int c = j * 64 + lane;
      if (c < C) {
        float x = s[c] * scale;
        if (mode == 0) x += (1.0f - rm * cmask[(int64_t)zo * C + c]) * VMR_NEG_INF_MASK;
        else x += cmask[(int64_t)c * cm_stride + zo];
        v[j] = x;
        mx = fmaxf(mx, x);
      } else v[j] = -INFINITY;
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
      const int c = j * 64 + lane;
      if (c < C) { v[j] = __expf(v[j] - mx); sum += v[j]; }
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    T* p = P + row * ldP;
    T* pk = Pkeep ? Pkeep + row * ldP : nullptr;
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
      const int c = j * 64 + lane;
      if (c < ldP) {
        float o = 0.f;
        if (c < C) {
          o = v[j] * inv;
          if (pk) pk[c] = from_f<T>(o);
          if (drop_p > 0.f) o = vmr_keep(seed, (uint64_t)row * C + c, thresh) ? o * dscale : 0.f;
        } else if (pk) pk[c] = from_f<T>(0.f);
        p[c] = from_f<T>(o);
      }
    }
  }
}

// dS = scale * Pk * (dPk - sum_c dPk*Pk), where Pk are the pre-dropout
// probabilities and dPk = mask*dscale*dP (mask regenerated from the seed).  With
// dropout active the forward writes both P (dropped, feeds P.V) and Pk.
template <typename T, int MAXN>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dP, const T* __restrict__ Pk,
                                                          T* __restrict__ dS, int64_t nrows, int C, int ldS,
                                                          int ldP, float scale, float drop_p, uint32_t seed0,
                                                          const uint32_t* __restrict__ step) {
  const uint32_t seed = vmr_seed(seed0, step);
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t thresh = vmr_drop_thresh(drop_p);
  const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wid; row < nrows; row += (int64_t)gridDim.x * 4) {
    const float* g = dP + row * ldS;
    const T* p = Pk + row * ldP;
    float pv[MAXN], gv[MAXN];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
      const int c = j * 64 + lane;
      if (c < C) {
        pv[j] = to_f<T>(p[c]);
        float d = g[c];
        if (drop_p > 0.f) d = vmr_keep(seed, (uint64_t)row * C + c, thresh) ? d * dscale : 0.f;
        gv[j] = d;
        dot += d * pv[j];
      } else { pv[j] = 0.f; gv[j] = 0.f; }
    }
    dot = wave_sum(dot);
    T* o = dS + row * ldP;
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
      const int c = j * 64 + lane;
      if (c < ldP) o[c] = from_f<T>(c < C ? scale * pv[j] * (gv[j] - dot) : 0.f);
    }
  }
}

#define SM_DISPATCH(W, ...)                                   \
  do {                                                        \
    if ((W) <= 64) { constexpr int MN = 1; __VA_ARGS__; }     \
    else if ((W) <= 128) { constexpr int MN = 2; __VA_ARGS__; } \
    else if ((W) <= 256) { constexpr int MN = 4; __VA_ARGS__; } \
    else if ((W) <= 512) { constexpr int MN = 8; __VA_ARGS__; } \
    else { constexpr int MN = 16; __VA_ARGS__; }              \
  } while (0)

}  // namespace

extern "C" int vmr_softmax_fwd(const float* S, void* P, void* Pkeep, const float* rmask, const float* cmask, int mode, int Z,
                               int H, int R, int C, int ldS, int ldP, int cm_stride, float scale, int dtype,
                               float drop_p, uint32_t drop_seed, const uint32_t* drop_step, void* stream) {
  VMR_CHECK(S && P && cmask, "vmr_softmax_fwd: null pointer");
  VMR_CHECK(mode == 0 || mode == 1, "vmr_softmax_fwd: bad mode %d", mode);
  VMR_CHECK(mode != 0 || rmask, "vmr_softmax_fwd: mode 0 needs rmask");
  VMR_CHECK(C >= 1 && C <= 64 * MAXN_MAX && ldP <= 64 * MAXN_MAX && ldP >= C && ldS >= C && H >= 1,
            "vmr_softmax_fwd: bad sizes C=%d ldS=%d ldP=%d", C, ldS, ldP);
  const int64_t nrows = (int64_t)Z * R;
  if (nrows == 0) return 0;
  const int grid = (int)min((int64_t)8192, (nrows + 3) / 4);
  VMR_DISPATCH(dtype, T, SM_DISPATCH(ldP, hipLaunchKernelGGL((softmax_fwd_kernel<T, MN>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                        S, (T*)P, (T*)Pkeep, rmask, cmask, mode, nrows, H, R, C, ldS, ldP,
                                        cm_stride, scale, drop_p, drop_seed, drop_step)));
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_softmax_bwd(const float* dP, const void* P, void* dS, int Z, int R, int C, int ldS, int ldP,
                               float scale, int dtype, float drop_p, uint32_t drop_seed, const uint32_t* drop_step,
                               void* stream) {
  VMR_CHECK(dP && P && dS, "vmr_softmax_bwd: null pointer");
  VMR_CHECK(C >= 1 && C <= 64 * MAXN_MAX && ldP >= C && ldP <= 64 * MAXN_MAX && ldS >= C, "vmr_softmax_bwd: bad sizes");
  const int64_t nrows = (int64_t)Z * R;
  if (nrows == 0) return 0;
  const int grid = (int)min((int64_t)8192, (nrows + 3) / 4);
  VMR_DISPATCH(dtype, T, SM_DISPATCH(ldP, hipLaunchKernelGGL((softmax_bwd_kernel<T, MN>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                        dP, (const T*)P, (T*)dS, nrows, C, ldS, ldP, scale, drop_p, drop_seed,
                                        drop_step)));
  VMR_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------ CQ attention softmaxes
// The two masked softmaxes of CQAttention (reference models/layers.py:419-421) over the
// trilinear score S[b,c,q] = S2[b,c,q] + rowterm[b,c] + colterm[b,q]:
//   S_row = softmax_q(S + (1-qmask[b,q]) * -1e30)      (context -> query weights)
//   S_col = softmax_c(S + (1-cmask[b,c]) * -1e30)      (stored [b,c,q]; the reference transposes it)
// One workgroup per sample: the [Lc, Lq] score tile lives in LDS (<= 64 KiB fp32), rows are
// reduced by waves, columns by threads.  Outputs in the activation dtype with a padded leading
// dimension ldP (zero filled) so the following batched GEMMs use 16-byte loads.
namespace {

template <typename T>
__global__ __launch_bounds__(256) void cq_softmax_fwd_kernel(const float* __restrict__ S2, const float* __restrict__ rowterm,
                                                             const float* __restrict__ colterm,
                                                             const float* __restrict__ cmask, const float* __restrict__ qmask,
                                                             T* __restrict__ Srow, T* __restrict__ Scol, int Lc, int Lq,
                                                             int ldS, int ldP) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);   // [Lc][Lq]
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* s2 = S2 + (int64_t)b * Lc * ldS;
  for (int i = threadIdx.x; i < Lc * Lq; i += 256) {
    const int c = i / Lq, q = i - c * Lq;
    float v = s2[(int64_t)c * ldS + q];
    if (rowterm) v += rowterm[(int64_t)b * Lc + c];
    if (colterm) v += colterm[(int64_t)b * Lq + q];
    tile[i] = v;
  }
  __syncthreads();
  // blockIdx.y splits the two softmaxes over two workgroups per sample (each re-reads the small tile)
  const bool do_rows = gridDim.y == 1 || blockIdx.y == 0, do_cols = gridDim.y == 1 || blockIdx.y == 1;
  // rows: softmax over q with the query mask.  Short query axis (Lq <= 32, e.g. 20 words): one THREAD per row
  // (a wave per row would leave 2/3 of its lanes idle and walk 32 rows per wave)
  if (do_rows && Lq <= 32) {
    for (int c = threadIdx.x; c < Lc; c += 256) {
      float v[32];
      float mx = -INFINITY;
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        v[q] = q < Lq ? tile[c * Lq + q] + (1.0f - qmask[(int64_t)b * Lq + q]) * VMR_NEG_INF_MASK : -INFINITY;
        mx = fmaxf(mx, v[q]);
      }
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < 32; ++q) { v[q] = q < Lq ? __expf(v[q] - mx) : 0.f; sum += v[q]; }
      const float inv = 1.f / sum;
      T* o = Srow + ((int64_t)b * Lc + c) * ldP;
#pragma unroll
      for (int q = 0; q < 32; ++q)
        if (q < ldP) o[q] = from_f<T>(v[q] * inv);
      for (int q = 32; q < ldP; ++q) o[q] = from_f<T>(0.f);
    }
  } else if (do_rows)
  for (int c = wid; c < Lc; c += 4) {
    float mx = -INFINITY;
    for (int q = lane; q < Lq; q += 64)
      mx = fmaxf(mx, tile[c * Lq + q] + (1.0f - qmask[(int64_t)b * Lq + q]) * VMR_NEG_INF_MASK);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int q = lane; q < Lq; q += 64)
      sum += __expf(tile[c * Lq + q] + (1.0f - qmask[(int64_t)b * Lq + q]) * VMR_NEG_INF_MASK - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    T* o = Srow + ((int64_t)b * Lc + c) * ldP;
    for (int q = lane; q < ldP; q += 64)
      o[q] = from_f<T>(q < Lq ? __expf(tile[c * Lq + q] + (1.0f - qmask[(int64_t)b * Lq + q]) * VMR_NEG_INF_MASK - mx) * inv
                              : 0.f);
  }
  // columns: softmax over c with the context mask.  256/Lq' threads share a column (each takes a
  // stripe of c), so a short query axis (Lq = 20) does not leave 236 threads idle.
  if (!do_cols) return;
  if (Lq > 256) {   // long query axis: a thread per column keeps every thread busy already
    for (int q = threadIdx.x; q < ldP; q += 256) {
      if (q >= Lq) {
        for (int c = 0; c < Lc; ++c) Scol[((int64_t)b * Lc + c) * ldP + q] = from_f<T>(0.f);
        continue;
      }
      float mx = -INFINITY;
      for (int c = 0; c < Lc; ++c)
        mx = fmaxf(mx, tile[c * Lq + q] + (1.0f - cmask[(int64_t)b * Lc + c]) * VMR_NEG_INF_MASK);
      float sum = 0.f;
      for (int c = 0; c < Lc; ++c)
        sum += __expf(tile[c * Lq + q] + (1.0f - cmask[(int64_t)b * Lc + c]) * VMR_NEG_INF_MASK - mx);
      const float inv = 1.f / sum;
      for (int c = 0; c < Lc; ++c)
        Scol[((int64_t)b * Lc + c) * ldP + q] =
            from_f<T>(__expf(tile[c * Lq + q] + (1.0f - cmask[(int64_t)b * Lc + c]) * VMR_NEG_INF_MASK - mx) * inv);
    }
  } else {
    float* part = tile + Lc * Lq;                      // [NS][Lq] partial max, then partial sums
    const int NS = max(1, min(256 / Lq, Lc));          // stripes per column
    const int q = threadIdx.x % Lq, st = threadIdx.x / Lq;
    const bool act = st < NS;
    float mx = -INFINITY;
    if (act)
      for (int c = st; c < Lc; c += NS)
        mx = fmaxf(mx, tile[c * Lq + q] + (1.0f - cmask[(int64_t)b * Lc + c]) * VMR_NEG_INF_MASK);
    if (act) part[st * Lq + q] = mx;
    __syncthreads();
    if (act)
      for (int k = 0; k < NS; ++k) mx = fmaxf(mx, part[k * Lq + q]);
    __syncthreads();
    float sum = 0.f;
    if (act)
      for (int c = st; c < Lc; c += NS)
        sum += __expf(tile[c * Lq + q] + (1.0f - cmask[(int64_t)b * Lc + c]) * VMR_NEG_INF_MASK - mx);
    if (act) part[st * Lq + q] = sum;
    __syncthreads();
    if (act) {
      sum = 0.f;
      for (int k = 0; k < NS; ++k) sum += part[k * Lq + q];
      const float inv = 1.f / sum;
      for (int c = st; c < Lc; c += NS)
        Scol[((int64_t)b * Lc + c) * ldP + q] =
            from_f<T>(__expf(tile[c * Lq + q] + (1.0f - cmask[(int64_t)b * Lc + c]) * VMR_NEG_INF_MASK - mx) * inv);
    }
    for (int i = threadIdx.x; i < Lc * (ldP - Lq); i += 256) {   // zero the padding columns
      const int c = i / (ldP - Lq), qq = Lq + i % (ldP - Lq);
      Scol[((int64_t)b * Lc + c) * ldP + qq] = from_f<T>(0.f);
    }
  }
}

// dS = Srow*(dSrow - sum_q dSrow*Srow) + Scol*(dScol - sum_c dScol*Scol); drow[c] = sum_q dS, dcol[q] = sum_c dS
template <typename T>
__global__ __launch_bounds__(256) void cq_softmax_bwd_kernel(const T* __restrict__ dSrow, const T* __restrict__ dScol,
                                                             const T* __restrict__ Srow, const T* __restrict__ Scol,
                                                             float* __restrict__ dS2, float* __restrict__ drow,
                                                             float* __restrict__ dcol, int Lc, int Lq, int ldS, int ldP) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);   // [Lc][Lq] accumulates dS
  float* cdot = tile + Lc * Lq;                   // [Lq] column dots
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t base = (int64_t)b * Lc * ldP;
  float* part = cdot + Lq;                           // [NS][Lq] partials
  const bool striped = Lq <= 256;                    // 256/Lq threads share a column (a stripe of c each)
  const int NS = striped ? max(1, min(256 / Lq, Lc)) : 1;
  const int pq = threadIdx.x % Lq, pst = threadIdx.x / Lq;
  if (striped) {
    if (pst < NS) {
      float d = 0.f;
      for (int c = pst; c < Lc; c += NS) d += to_f<T>(dScol[base + (int64_t)c * ldP + pq]) * to_f<T>(Scol[base + (int64_t)c * ldP + pq]);
      part[pst * Lq + pq] = d;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < Lq; q += 256) {
      float d = 0.f;
      for (int k = 0; k < NS; ++k) d += part[k * Lq + q];
      cdot[q] = d;
    }
  } else {
    for (int q = threadIdx.x; q < Lq; q += 256) {
      float d = 0.f;
      for (int c = 0; c < Lc; ++c) d += to_f<T>(dScol[base + (int64_t)c * ldP + q]) * to_f<T>(Scol[base + (int64_t)c * ldP + q]);
      cdot[q] = d;
    }
  }
  __syncthreads();
  for (int c = wid; c < Lc; c += 4) {
    float d = 0.f;
    for (int q = lane; q < Lq; q += 64) d += to_f<T>(dSrow[base + (int64_t)c * ldP + q]) * to_f<T>(Srow[base + (int64_t)c * ldP + q]);
    d = wave_sum(d);
    float rs = 0.f;
    for (int q = lane; q < Lq; q += 64) {
      const int64_t i = base + (int64_t)c * ldP + q;
      const float g = to_f<T>(Srow[i]) * (to_f<T>(dSrow[i]) - d) + to_f<T>(Scol[i]) * (to_f<T>(dScol[i]) - cdot[q]);
      tile[c * Lq + q] = g;
      dS2[((int64_t)b * Lc + c) * ldS + q] = g;
      rs += g;
    }
    rs = wave_sum(rs);
    if (lane == 0 && drow) drow[(int64_t)b * Lc + c] = rs;
  }
  __syncthreads();
  if (dcol) {
    if (striped) {
      if (pst < NS) {
        float cs = 0.f;
        for (int c = pst; c < Lc; c += NS) cs += tile[c * Lq + pq];
        part[pst * Lq + pq] = cs;
      }
      __syncthreads();
      for (int q = threadIdx.x; q < Lq; q += 256) {
        float cs = 0.f;
        for (int k = 0; k < NS; ++k) cs += part[k * Lq + q];
        dcol[(int64_t)b * Lq + q] = cs;
      }
    } else {
      for (int q = threadIdx.x; q < Lq; q += 256) {
        float cs = 0.f;
        for (int c = 0; c < Lc; ++c) cs += tile[c * Lq + q];
        dcol[(int64_t)b * Lq + q] = cs;
      }
    }
  }
}

}  // namespace

extern "C" int vmr_cq_softmax_fwd(const float* S2, const float* rowterm, const float* colterm, const float* cmask,
                                  const float* qmask, void* Srow, void* Scol, int B, int Lc, int Lq, int ldS, int ldP,
                                  int dtype, void* stream) {
  VMR_CHECK(S2 && cmask && qmask && Srow && Scol, "vmr_cq_softmax_fwd: null pointer");
  VMR_CHECK(Lc >= 1 && Lq >= 1 && ldS >= Lq && ldP >= Lq && ((size_t)Lc * Lq + 256 + Lq) * 4 <= 96 * 1024,
            "vmr_cq_softmax_fwd: score tile %dx%d does not fit LDS", Lc, Lq);
  if (B == 0) return 0;
  const size_t lds = ((size_t)Lc * Lq + 256 + Lq) * 4;   // tile + column-stripe partials
  const void* fn = nullptr;
  VMR_DISPATCH(dtype, T, fn = (const void*)cq_softmax_fwd_kernel<T>);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return vmr_fail(-5, "vmr_cq_softmax_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  VMR_DISPATCH(dtype, T, hipLaunchKernelGGL(cq_softmax_fwd_kernel<T>, dim3(B, 2), dim3(256), lds, (hipStream_t)stream, S2, rowterm, colterm,
                       cmask, qmask, (T*)Srow, (T*)Scol, Lc, Lq, ldS, ldP));
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_cq_softmax_bwd(const void* dSrow, const void* dScol, const void* Srow, const void* Scol, float* dS2,
                                  float* drow, float* dcol, int B, int Lc, int Lq, int ldS, int ldP, int dtype,
                                  void* stream) {
  VMR_CHECK(dSrow && dScol && Srow && Scol && dS2, "vmr_cq_softmax_bwd: null pointer");
  VMR_CHECK(Lc >= 1 && Lq >= 1 && ldS >= Lq && ldP >= Lq && ((size_t)Lc * Lq + 2 * Lq + 256) * 4 <= 96 * 1024,
            "vmr_cq_softmax_bwd: score tile %dx%d does not fit LDS", Lc, Lq);
  if (B == 0) return 0;
  const size_t lds = ((size_t)Lc * Lq + 2 * Lq + 256) * 4;   // dS tile + column dots + column-stripe partials
  const void* fn = nullptr;
  VMR_DISPATCH(dtype, T, fn = (const void*)cq_softmax_bwd_kernel<T>);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return vmr_fail(-5, "vmr_cq_softmax_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  VMR_DISPATCH(dtype, T, hipLaunchKernelGGL(cq_softmax_bwd_kernel<T>, dim3(B), dim3(256), lds, (hipStream_t)stream, (const T*)dSrow,
                       (const T*)dScol, (const T*)Srow, (const T*)Scol, dS2, drow, dcol, Lc, Lq, ldS, ldP));
  VMR_LAUNCH_CHECK();
  return 0;
}
