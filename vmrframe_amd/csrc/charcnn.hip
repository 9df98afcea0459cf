// charcnn.hip -- CharacterEmbedding (reference models/layers.py:51-75) as one kernel each way: character
// lookup (padding_idx 0) + dropout + the four Conv2d(char_dim -> 10k, kernel (1,k)), k = 1..4, + ReLU + max over
// positions, written straight into the [words, ldo] embedding matrix that feeds query_conv1d.
// It replaces 4 x {unfold, cast, 128-wide MFMA GEMM with N = 10..40, amax} and their backward chains
// (~110 launches, ~0.8 ms of a cfg2 step).  The work is tiny (0.44 GFLOP forward): plain FMA on LDS-resident
// operands.  Per workgroup: the conv weights TRANSPOSED in LDS ([c*k+j][out channel]: lanes of a wave read
// consecutive out channels), the dropped-out character rows of WPB words in LDS (fp32 values rounded through
// the compute dtype, like the operand of the GEMM path they replace).
// Backward: dW/db partials per workgroup (two-stage, no atomics on the weights), the gradient of the
// character rows scattered into the table gradient with float atomics (6000 addresses, spread in time).
#include "common.h"

namespace {

constexpr int CC_NK = 4;            // kernel widths 1..4
constexpr int CC_MAXC = 16;         // characters per word

struct CharCnnArgs {
  const int64_t* ids;               // [W, C]
  const float* table;               // [num_chars, CD]
  const float* w[CC_NK];            // [oc_k, CD, 1, k] fp32 masters
  const float* b[CC_NK];            // [oc_k]
  int oc[CC_NK];                    // out channels per width
  int woff[CC_NK];                  // element offset of conv k in the LDS / partial weight block
  int coff[CC_NK];                  // first output column of conv k
  int W, C, CD, OT, wtot;           // words, chars per word, char dim, total out channels, total weight elements
  float drop_p; uint32_t seed; const uint32_t* step;
};

template <typename T> __device__ __forceinline__ float round_through(float v) { return to_f<T>(from_f<T>(v)); }

// LDS layout: Wt (T) [wtot] | ce (float) [WPB][C][CD] | (bwd) G float [WPB][OT], A int [WPB][OT]
template <typename T>
__device__ __forceinline__ void stage_weights(const CharCnnArgs& a, T* Wt) {
#pragma unroll
  for (int kk = 0; kk < CC_NK; ++kk) {
    const int k = kk + 1, n = a.oc[kk] * a.CD * k;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {      // i = (o*CD + c)*k + j  (the parameter's own order)
      const int o = i / (a.CD * k), r = i - o * (a.CD * k);
      Wt[a.woff[kk] + r * a.oc[kk] + o] = from_f<T>(a.w[kk][i]);
    }
  }
}

// character rows of the workgroup's words, dropped out, rounded through T, stored [word][channel][CP positions]
// (positions contiguous: the convolution reads them as 16-byte vectors; positions >= C are zero)
// backward layout of the weights: [o][j][c] (channel contiguous: lanes of a wave that differ in c read consecutive
// elements, conflict-free)
template <typename T>
__device__ __forceinline__ void stage_weights_ojc(const CharCnnArgs& a, T* Wt) {
#pragma unroll
  for (int kk = 0; kk < CC_NK; ++kk) {
    const int k = kk + 1, n = a.oc[kk] * a.CD * k;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {      // i = (o*CD + c)*k + j
      const int o = i / (a.CD * k), r = i - o * (a.CD * k), c = r / k, j = r - c * k;
      Wt[a.woff[kk] + (o * k + j) * a.CD + c] = from_f<T>(a.w[kk][i]);
    }
  }
}

// POS_MAJOR (backward): [word][position][channel] instead, channel contiguous (lanes differ in the channel there)
template <typename T, int CP, bool POS_MAJOR = false>
__device__ __forceinline__ void stage_chars(const CharCnnArgs& a, float* ce, int w0, int nw, uint32_t seed, uint32_t thresh,
                                            float dscale) {
  const int per = CP * a.CD;
  for (int i = threadIdx.x; i < nw * per; i += blockDim.x) {
    const int wl = i / per, r = i - wl * per;
    const int d = POS_MAJOR ? r % a.CD : r / CP, pos = POS_MAJOR ? r / a.CD : r - (r / CP) * CP;
    float v = 0.f;
    if (pos < a.C) {
      const int64_t word = w0 + wl;
      const int64_t id = a.ids[word * a.C + pos];
      v = a.table[id * a.CD + d];
      if (a.drop_p > 0.f) v = vmr_keep(seed, (uint64_t)(word * a.C + pos) * a.CD + d, thresh) ? v * dscale : 0.f;
    }
    ce[i] = round_through<T>(v);
  }
}

template <typename T, int K, int CM>
__device__ __forceinline__ void conv_fwd(const CharCnnArgs& a, const T* Wt, const float* ce, int w0, int nw, T* out, int64_t ldo,
                                         int8_t* amax, int tid, int nthr) {
  const int kk = K - 1, oc = a.oc[kk], np = a.C - K + 1;
  const T* wt = Wt + a.woff[kk];
  for (int item = tid; item < oc * nw; item += nthr) {
    const int o = item % oc, wl = item / oc;
    const float* cw = ce + wl * CM * a.CD;
    float acc[CM];
#pragma unroll
    for (int p = 0; p < CM; ++p) acc[p] = 0.f;
    for (int c = 0; c < a.CD; ++c) {
      float wv[K];
#pragma unroll
      for (int j = 0; j < K; ++j) wv[j] = to_f<T>(wt[(c * K + j) * oc + o]);
      float cv[CM];
#pragma unroll
      for (int p4 = 0; p4 < CM; p4 += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(cw + c * CM + p4);
        cv[p4] = t[0]; cv[p4 + 1] = t[1]; cv[p4 + 2] = t[2]; cv[p4 + 3] = t[3];
      }
#pragma unroll
      for (int p = 0; p < CM - K + 1; ++p)
#pragma unroll
        for (int j = 0; j < K; ++j) acc[p] += wv[j] * cv[p + j];
    }
    float best = -INFINITY;
    int arg = 0;
#pragma unroll
    for (int p = 0; p < CM - K + 1; ++p)
      if (p < np && acc[p] > best) { best = acc[p]; arg = p; }
    best += a.b[kk][o];
    const int64_t word = w0 + wl;
    out[word * ldo + a.coff[kk] + o] = from_f<T>(fmaxf(best, 0.f));
    amax[word * a.OT + a.coff[kk] + o] = (int8_t)arg;
  }
}

template <typename T, int WPB, int CM>
__global__ __launch_bounds__(1024) void char_cnn_fwd_kernel(CharCnnArgs a, T* __restrict__ out, int64_t ldo,
                                                           int8_t* __restrict__ amax) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* Wt = reinterpret_cast<T*>(smem);
  float* ce = reinterpret_cast<float*>(smem + ((a.wtot * sizeof(T) + 15) & ~(size_t)15));
  const uint32_t seed = vmr_seed(a.seed, a.step);
  const uint32_t thresh = vmr_drop_thresh(a.drop_p);
  const float dscale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  const int w0 = blockIdx.x * WPB, nw = min(WPB, a.W - w0);
  stage_weights<T>(a, Wt);
  stage_chars<T, CM>(a, ce, w0, nw, seed, thresh, dscale);
  __syncthreads();
  // one item = (output channel, word): 80 .. 320 items per width at 8 words -- run one after the other they kept 2 .. 5 of the
  // 16 waves busy for a sum of four passes; with a wave range per width (2 + 3 + 4 + 5 = 14 waves) the four run side by side
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int n1 = (a.oc[0] * nw + 63) >> 6, n2 = (a.oc[1] * nw + 63) >> 6, n3 = (a.oc[2] * nw + 63) >> 6, n4 = (a.oc[3] * nw + 63) >> 6;
  if (n1 + n2 + n3 + n4 <= (int)(blockDim.x >> 6)) {
    if (wv < n1) conv_fwd<T, 1, CM>(a, Wt, ce, w0, nw, out, ldo, amax, wv * 64 + ln, n1 * 64);
    else if (wv < n1 + n2) conv_fwd<T, 2, CM>(a, Wt, ce, w0, nw, out, ldo, amax, (wv - n1) * 64 + ln, n2 * 64);
    else if (wv < n1 + n2 + n3) conv_fwd<T, 3, CM>(a, Wt, ce, w0, nw, out, ldo, amax, (wv - n1 - n2) * 64 + ln, n3 * 64);
    else if (wv < n1 + n2 + n3 + n4) conv_fwd<T, 4, CM>(a, Wt, ce, w0, nw, out, ldo, amax, (wv - n1 - n2 - n3) * 64 + ln, n4 * 64);
  } else {
    conv_fwd<T, 1, CM>(a, Wt, ce, w0, nw, out, ldo, amax, threadIdx.x, blockDim.x);
    conv_fwd<T, 2, CM>(a, Wt, ce, w0, nw, out, ldo, amax, threadIdx.x, blockDim.x);
    conv_fwd<T, 3, CM>(a, Wt, ce, w0, nw, out, ldo, amax, threadIdx.x, blockDim.x);
    conv_fwd<T, 4, CM>(a, Wt, ce, w0, nw, out, ldo, amax, threadIdx.x, blockDim.x);
  }
}

// ---- backward: partial row of workgroup g = [W_1 | W_2 | W_3 | W_4 | b (OT)]  (wtot + OT floats)
template <typename T, int K, int CP>
__device__ __forceinline__ void conv_dw(const CharCnnArgs& a, const float* ce, const float* G, const int* A, int nw,
                                        float* prow) {
  const int kk = K - 1, oc = a.oc[kk];
  for (int item = threadIdx.x; item < oc * a.CD; item += blockDim.x) {
    const int o = item / a.CD, c = item - o * a.CD;
    float acc[K];
#pragma unroll
    for (int j = 0; j < K; ++j) acc[j] = 0.f;
    for (int wl = 0; wl < nw; ++wl) {
      const float g = G[wl * a.OT + a.coff[kk] + o];
      if (g != 0.f) {
        const int p = A[wl * a.OT + a.coff[kk] + o];
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] += g * ce[(wl * CP + p + j) * a.CD + c];
      }
    }
#pragma unroll
    for (int j = 0; j < K; ++j) prow[a.woff[kk] + (o * a.CD + c) * K + j] = acc[j];
  }
}

template <typename T, int WPB, int CP>
__global__ __launch_bounds__(1024) void char_cnn_bwd_kernel(CharCnnArgs a, const T* __restrict__ dout, const T* __restrict__ out,
                                                           int64_t ldo, const int8_t* __restrict__ amax,
                                                           float* __restrict__ part, float* __restrict__ dtable) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* Wt = reinterpret_cast<T*>(smem);
  float* ce = reinterpret_cast<float*>(smem + ((a.wtot * sizeof(T) + 15) & ~(size_t)15));
  float* G = ce + WPB * CP * a.CD;
  int* A = reinterpret_cast<int*>(G + WPB * a.OT);
  const uint32_t seed = vmr_seed(a.seed, a.step);
  const uint32_t thresh = vmr_drop_thresh(a.drop_p);
  const float dscale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  const int w0 = blockIdx.x * WPB, nw = min(WPB, a.W - w0);
  stage_weights_ojc<T>(a, Wt);
  stage_chars<T, CP, true>(a, ce, w0, nw, seed, thresh, dscale);
  for (int i = threadIdx.x; i < nw * a.OT; i += blockDim.x) {
    const int wl = i / a.OT, o = i - wl * a.OT;
    const int64_t word = w0 + wl;
    const float y = to_f<T>(out[word * ldo + o]);
    G[i] = y > 0.f ? to_f<T>(dout[word * ldo + o]) : 0.f;   // ReLU
    A[i] = amax[word * a.OT + o];
  }
  __syncthreads();
  float* prow = part + (int64_t)blockIdx.x * (a.wtot + a.OT);
  conv_dw<T, 1, CP>(a, ce, G, A, nw, prow);
  conv_dw<T, 2, CP>(a, ce, G, A, nw, prow);
  conv_dw<T, 3, CP>(a, ce, G, A, nw, prow);
  conv_dw<T, 4, CP>(a, ce, G, A, nw, prow);
  for (int o = threadIdx.x; o < a.OT; o += blockDim.x) {     // bias partials
    float s = 0.f;
    for (int wl = 0; wl < nw; ++wl) s += G[wl * a.OT + o];
    prow[a.wtot + o] = s;
  }
  // gradient of the (dropped-out) character rows -> table rows.  Scatter form: a thread owns the (word, channel) column of
  // an LDS image [word][position][channel] (over the character rows, which the weight-gradient phase is done with) and walks
  // the word's output channels once -- each live one (ReLU kills about half) adds its k taps at its arg-max position.
  // (Before: one thread per (word, position, channel) walked ALL output channels and kept the few whose window covered
  // its position: 100 broadcast LDS reads + branches per element, 2500 LDS reads per thread -- a third of the kernel.)
  // Same summation order per element as before ((width, channel) ascending), so the table gradient is bit-identical.
  if (dtable) {
    __syncthreads();                                             // every wave is done with ce
    float* dce = ce;                                             // [nw][CP][CD]
    for (int i = threadIdx.x; i < nw * CP * a.CD; i += blockDim.x) dce[i] = 0.f;
    __syncthreads();
    for (int item = threadIdx.x; item < nw * a.CD; item += blockDim.x) {
      const int wl = item / a.CD, c = item - wl * a.CD;
      float* col = dce + (wl * CP) * a.CD + c;
#pragma unroll
      for (int kk = 0; kk < CC_NK; ++kk) {
        const int k = kk + 1, oc = a.oc[kk];
        const T* wt = Wt + a.woff[kk] + c;
        const float* Gk = G + wl * a.OT + a.coff[kk];
        const int* Ak = A + wl * a.OT + a.coff[kk];
        for (int o = 0; o < oc; ++o) {
          const float g = Gk[o];
          if (g != 0.f) {
            const int p = Ak[o];
#pragma unroll
            for (int j = 0; j < k; ++j) col[(p + j) * a.CD] += g * to_f<T>(wt[(o * k + j) * a.CD]);
          }
        }
      }
    }
    __syncthreads();
    for (int item = threadIdx.x; item < nw * a.CD * a.C; item += blockDim.x) {
      const int wl = item / (a.CD * a.C), r = item - wl * (a.CD * a.C), q = r / a.CD, c = r - q * a.CD;
      const int64_t word = w0 + wl;
      const int64_t id = a.ids[word * a.C + q];
      float d = dce[(wl * CP + q) * a.CD + c];
      if (id == 0) d = 0.f;                                      // padding_idx 0 receives no gradient
      else if (a.drop_p > 0.f && !vmr_keep(seed, (uint64_t)(word * a.C + q) * a.CD + c, thresh)) d = 0.f;
      if (d != 0.f) atomicAdd(&dtable[id * a.CD + c], d * dscale);
    }
  }
}

struct CcDst { float* w[CC_NK]; float* b[CC_NK]; int woff[CC_NK], wn[CC_NK], coff[CC_NK], oc[CC_NK]; };

// dst += sum over partial rows; 16 rows per thread in flight, blockIdx.y walks the row groups
__global__ __launch_bounds__(256) void char_cnn_reduce_kernel(const float* __restrict__ part, CcDst d, int nrows, int wtot, int OT) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= wtot + OT) return;
  const int r0 = blockIdx.y * 16;
  float v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = (r0 + k < nrows) ? part[(int64_t)(r0 + k) * (wtot + OT) + j] : 0.f;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += v[k];
  float* dst = nullptr;
  if (j < wtot) {
#pragma unroll
    for (int kk = 0; kk < CC_NK; ++kk)
      if (j >= d.woff[kk] && j < d.woff[kk] + d.wn[kk]) dst = d.w[kk] + (j - d.woff[kk]);
  } else {
    const int o = j - wtot;
#pragma unroll
    for (int kk = 0; kk < CC_NK; ++kk)
      if (o >= d.coff[kk] && o < d.coff[kk] + d.oc[kk]) dst = d.b[kk] + (o - d.coff[kk]);
  }
  if (dst) atomicAdd(dst, s);
}

int fill_args(CharCnnArgs& a, const int64_t* ids, const float* table, const float* const* w, const float* const* b, const int* oc,
              int W, int C, int CD, float drop_p, uint32_t seed, const uint32_t* step) {
  a.ids = ids; a.table = table;
  int wo = 0, co = 0;
  for (int kk = 0; kk < CC_NK; ++kk) {
    a.w[kk] = w[kk]; a.b[kk] = b[kk]; a.oc[kk] = oc[kk];
    a.woff[kk] = wo; a.coff[kk] = co;
    wo += oc[kk] * CD * (kk + 1);
    co += oc[kk];
  }
  a.W = W; a.C = C; a.CD = CD; a.OT = co; a.wtot = wo;
  a.drop_p = drop_p; a.seed = seed; a.step = step;
  return 0;
}


}  // namespace

// words per workgroup: 8 (bf16) covers 1280 words with 160 workgroups -- fewer than CUs; VMR_CC_WPB=4 doubles them
static int cc_wpb(int dtype) {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("VMR_CC_WPB");
    v = e ? atoi(e) : 8;
    if (v != 4 && v != 8) v = 8;
  }
  return vmr_dtype_16(dtype) ? v : 4;
}

extern "C" int vmr_char_cnn_ws_floats(int W, int CD, const int* oc, int dtype) {
  int wtot = 0, ot = 0;
  for (int kk = 0; kk < CC_NK; ++kk) { wtot += oc[kk] * CD * (kk + 1); ot += oc[kk]; }
  const int wpb = cc_wpb(dtype);   // words per workgroup of the backward kernel
  return ((W + wpb - 1) / wpb) * (wtot + ot);
}

extern "C" int vmr_char_cnn_fwd(const int64_t* char_ids, const float* table, const float* const* w, const float* const* b,
                                const int* oc, void* out, int64_t ldo, int8_t* amax, int W, int C, int CD, int dtype,
                                float drop_p, uint32_t drop_seed, const uint32_t* drop_step, void* stream) {
  VMR_CHECK(char_ids && table && w && b && oc && out && amax, "vmr_char_cnn_fwd: null pointer");
  VMR_CHECK(C >= CC_NK && C <= CC_MAXC && CD >= 1, "vmr_char_cnn_fwd: need %d <= chars per word <= %d", CC_NK, CC_MAXC);
  if (W == 0) return 0;
  CharCnnArgs a;
  fill_args(a, char_ids, table, w, b, oc, W, C, CD, drop_p, drop_seed, drop_step);
  const int esz = vmr_dtype_size(dtype), wpb = cc_wpb(dtype);
  const int cp = C <= 8 ? 8 : 16;
  const size_t lds = ((size_t)a.wtot * esz + 15) / 16 * 16 + (size_t)wpb * cp * CD * 4;
  VMR_CHECK(lds <= 160 * 1024, "vmr_char_cnn_fwd: operands do not fit LDS (%zu B)", lds);
  const dim3 grid((unsigned)((W + wpb - 1) / wpb));
#define VMR_CC_FWD(TT, WPBV, CMV)                                                                                          \
  do {                                                                                                                     \
    const void* fn = (const void*)char_cnn_fwd_kernel<TT, WPBV, CMV>;                                                      \
    if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)    \
      return vmr_fail(-5, "vmr_char_cnn_fwd: hipFuncSetAttribute");                                                        \
    hipLaunchKernelGGL((char_cnn_fwd_kernel<TT, WPBV, CMV>), grid, dim3(1024), lds, (hipStream_t)stream, a, (TT*)out, ldo,  \
                       amax);                                                                                              \
  } while (0)
  if (vmr_dtype_16(dtype) && wpb == 8) VMR_DISPATCH16(dtype, E, { if (C <= 8) VMR_CC_FWD(E, 8, 8); else VMR_CC_FWD(E, 8, 16); });
  else VMR_DISPATCH(dtype, T, { if (C <= 8) VMR_CC_FWD(T, 4, 8); else VMR_CC_FWD(T, 4, 16); });
#undef VMR_CC_FWD
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_char_cnn_bwd(const void* dout, const void* out, int64_t ldo, const int8_t* amax, const int64_t* char_ids,
                                const float* table, const float* const* w, const float* const* b, const int* oc, float* const* dw,
                                float* const* db, float* dtable /*nullable*/, float* workspace, int W, int C, int CD, int dtype,
                                float drop_p, uint32_t drop_seed, const uint32_t* drop_step, void* stream) {
  VMR_CHECK(dout && out && amax && char_ids && table && w && b && oc && dw && db && workspace, "vmr_char_cnn_bwd: null pointer");
  VMR_CHECK(C >= CC_NK && C <= CC_MAXC && CD >= 1, "vmr_char_cnn_bwd: need %d <= chars per word <= %d", CC_NK, CC_MAXC);
  if (W == 0) return 0;
  CharCnnArgs a;
  fill_args(a, char_ids, table, w, b, oc, W, C, CD, drop_p, drop_seed, drop_step);
  const int esz = vmr_dtype_size(dtype), wpb = cc_wpb(dtype), cp = C <= 8 ? 8 : 16;
  const size_t lds = ((size_t)a.wtot * esz + 15) / 16 * 16 + (size_t)wpb * cp * CD * 4 + (size_t)wpb * a.OT * 8;
  VMR_CHECK(lds <= 160 * 1024, "vmr_char_cnn_bwd: operands do not fit LDS (%zu B)", lds);
  const int nblk = (W + wpb - 1) / wpb;
#define VMR_CC_BWD(TT, WPBV, CPV)                                                                                          \
  do {                                                                                                                     \
    const void* fn = (const void*)char_cnn_bwd_kernel<TT, WPBV, CPV>;                                                      \
    if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)    \
      return vmr_fail(-5, "vmr_char_cnn_bwd: hipFuncSetAttribute");                                                        \
    hipLaunchKernelGGL((char_cnn_bwd_kernel<TT, WPBV, CPV>), dim3(nblk), dim3(1024), lds, (hipStream_t)stream, a,          \
                       (const TT*)dout, (const TT*)out, ldo, amax, workspace, dtable);                                    \
  } while (0)
  if (vmr_dtype_16(dtype) && wpb == 8) VMR_DISPATCH16(dtype, E, { if (cp == 8) VMR_CC_BWD(E, 8, 8); else VMR_CC_BWD(E, 8, 16); });
  else VMR_DISPATCH(dtype, T, { if (cp == 8) VMR_CC_BWD(T, 4, 8); else VMR_CC_BWD(T, 4, 16); });
#undef VMR_CC_BWD
  VMR_LAUNCH_CHECK();
  CcDst d;
  for (int kk = 0; kk < CC_NK; ++kk) {
    d.w[kk] = dw[kk]; d.b[kk] = db[kk]; d.woff[kk] = a.woff[kk]; d.wn[kk] = oc[kk] * CD * (kk + 1);
    d.coff[kk] = a.coff[kk]; d.oc[kk] = oc[kk];
  }
  hipLaunchKernelGGL(char_cnn_reduce_kernel, dim3(cdiv(a.wtot + a.OT, 256), cdiv(nblk, 16)), dim3(256), 0, (hipStream_t)stream,
                     workspace, d, nblk, a.wtot, a.OT);
  VMR_LAUNCH_CHECK();
  return 0;
}
