// pool.hip -- small fused heads of the SeqPAN path that were strings of tiny framework kernels:
//   * WeightedPool (reference models/layers.py:440-453): alpha = softmax_l(x.w + mask), pooled = sum_l alpha*x
//   * infer_basic (utils/engine.py:28-44) and the IoU / R1@k / mIoU bookkeeping of the train and eval loops
//     (utils/utils.py:161-185, models/loss.py:83-109)   [SURVEY.md 8f, row N4]
// One workgroup per clip; rows are reduced by waves, columns by threads.  fp32 math, activations
// in the compute dtype.
#include "common.h"

namespace {

constexpr int WP_MAX_L = 2048;

template <typename T>
__global__ __launch_bounds__(256) void weighted_pool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ mask,
                                                                float* __restrict__ alpha, T* __restrict__ pooled,
                                                                int L, int D) {
  __shared__ float sc[WP_MAX_L];
  __shared__ float red[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T* xb = x + (int64_t)b * L * D;
  // scores: one wave per row
  for (int l = wid; l < L; l += 4) {
    float s = 0.f;
    for (int i = lane * 8; i < D; i += 512) {
      float xv[8], wv[8];
      Vec8<T>::load(xb + (int64_t)l * D + i, xv);
      Vec8<float>::load(w + i, wv);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += xv[e] * wv[e];
    }
    s = wave_sum(s);
    if (lane == 0) sc[l] = s + (1.0f - mask[(int64_t)b * L + l]) * VMR_NEG_INF_MASK;
  }
  __syncthreads();
  // softmax over the L scores (L is small: every thread strides over it)
  float mx = -INFINITY;
  for (int l = tid; l < L; l += 256) mx = fmaxf(mx, sc[l]);
  mx = wave_max(mx);
  if (lane == 0) red[wid] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int l = tid; l < L; l += 256) sum += __expf(sc[l] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wid] = sum;
  __syncthreads();
  const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
  for (int l = tid; l < L; l += 256) {
    const float a = __expf(sc[l] - mx) * inv;
    sc[l] = a;
    alpha[(int64_t)b * L + l] = a;
  }
  __syncthreads();
  // pooled[d] = sum_l alpha[l] * x[l, d]
  for (int i = tid * 8; i < D; i += 2048) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < L; ++l) {
      float xv[8];
      Vec8<T>::load(xb + (int64_t)l * D + i, xv);
      const float a = sc[l];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += a * xv[e];
    }
    Vec8<T>::store(pooled + (int64_t)b * D + i, acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void weighted_pool_bwd_kernel(const T* __restrict__ dpooled, const T* __restrict__ x,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ alpha, T* __restrict__ dx,
                                                                float* __restrict__ dw, int L, int D) {
  __shared__ float ds[WP_MAX_L];   // dalpha, then dscore
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T* xb = x + (int64_t)b * L * D;
  const T* dp = dpooled + (int64_t)b * D;
  const float* al = alpha + (int64_t)b * L;
  for (int l = wid; l < L; l += 4) {   // dalpha[l] = dpooled . x[l]
    float s = 0.f;
    for (int i = lane * 8; i < D; i += 512) {
      float xv[8], gv[8];
      Vec8<T>::load(xb + (int64_t)l * D + i, xv);
      Vec8<T>::load(dp + i, gv);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += xv[e] * gv[e];
    }
    s = wave_sum(s);
    if (lane == 0) ds[l] = s;
  }
  __syncthreads();
  float t = 0.f;
  for (int l = tid; l < L; l += 256) t += al[l] * ds[l];
  t = wave_sum(t);
  if (lane == 0) red[wid] = t;
  __syncthreads();
  t = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  for (int l = tid; l < L; l += 256) ds[l] = al[l] * (ds[l] - t);   // dscore
  __syncthreads();
  for (int i = tid * 8; i < D; i += 2048) {
    float gv[8], wv[8], dwv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    Vec8<T>::load(dp + i, gv);
    Vec8<float>::load(w + i, wv);
    for (int l = 0; l < L; ++l) {
      float xv[8], o[8];
      Vec8<T>::load(xb + (int64_t)l * D + i, xv);
      const float a = al[l], d = ds[l];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = a * gv[e] + d * wv[e];
        dwv[e] += d * xv[e];
      }
      Vec8<T>::store(dx + ((int64_t)b * L + l) * D + i, o);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(&dw[i + e], dwv[e]);
  }
}

// ------------------------------------------------------------------ infer_basic
// reference utils/engine.py:28-44: masked softmax of the start / end logits, the upper-triangular
// outer product, and the (first) indices of its maximum.  Because sp[i] >= 0 and rounding is
// monotonic, max_j>=i sp[i]*ep[j] == sp[i]*max_j>=i ep[j] exactly, so the [T,T] product is never
// formed: a suffix max of ep and a prefix max of sp give the row / column maxima.
constexpr int INF_MAX_T = 4096;

__global__ __launch_bounds__(256) void infer_basic_kernel(const float* __restrict__ sl, const float* __restrict__ el,
                                                          const float* __restrict__ vmask, float* __restrict__ frac,
                                                          int* __restrict__ idx, int T) {
  __shared__ float sp[INF_MAX_T], ep[INF_MAX_T];
  __shared__ float redf[8];
  __shared__ int redi[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float nvalid = 0.f;
  for (int h = 0; h < 2; ++h) {   // masked softmax of the two heads into LDS
    const float* z = (h ? el : sl) + (int64_t)b * T;
    float* out = h ? ep : sp;
    float mx = -INFINITY;
    for (int t = tid; t < T; t += 256) {
      const float m = vmask[(int64_t)b * T + t];
      const float v = z[t] + (1.0f - m) * VMR_NEG_INF_MASK;
      out[t] = v;
      mx = fmaxf(mx, v);
      if (h == 0) nvalid += m;
    }
    mx = wave_max(mx);
    if (lane == 0) redf[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    float sum = 0.f;
    for (int t = tid; t < T; t += 256) {
      const float e = expf(out[t] - mx);
      out[t] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) redf[4 + wid] = sum;
    __syncthreads();
    sum = redf[4] + redf[5] + redf[6] + redf[7];
    for (int t = tid; t < T; t += 256) out[t] = out[t] / sum;
    __syncthreads();
  }
  nvalid = wave_sum(nvalid);
  if (lane == 0) redf[wid] = nvalid;
  __syncthreads();
  nvalid = redf[0] + redf[1] + redf[2] + redf[3];
  __syncthreads();
  // row maxima r[i] = sp[i] * max_{j>=i} ep[j]; column maxima c[j] = max_{i<=j} sp[i] * ep[j]
  for (int h = 0; h < 2; ++h) {
    float best = -1.f;
    int bi = 0x7fffffff;
    for (int t = tid; t < T; t += 256) {
      float m = 0.f;
      if (h == 0) { for (int j = t; j < T; ++j) m = fmaxf(m, ep[j]); m = sp[t] * m; }
      else        { for (int i = 0; i <= t; ++i) m = fmaxf(m, sp[i]); m = m * ep[t]; }
      if (m > best) { best = m; bi = t; }   // ascending t within a thread: ties keep the first index
    }
    // (value desc, index asc) reduction over the workgroup
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { redf[wid] = best; redi[wid] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int k = 1; k < 4; ++k)
        if (redf[k] > best || (redf[k] == best && redi[k] < bi)) { best = redf[k]; bi = redi[k]; }
      idx[b * 2 + h] = bi;
      frac[b * 2 + h] = (float)bi / nvalid;
    }
    __syncthreads();
  }
}

// IoU of proposals vs ground truth (utils/utils.py:161-167) and the running R1@{0.3,0.5,0.7} / mIoU
// accumulators of models/loss.py:102-109: acc = {count>=0.3, count>=0.5, count>=0.7, n, sum_iou}
__global__ __launch_bounds__(256) void iou_metrics_kernel(const float* __restrict__ props, const float* __restrict__ gts,
                                                          float* __restrict__ ious, double* __restrict__ acc, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  double c3 = 0, c5 = 0, c7 = 0, s = 0, cnt = 0;
  if (i < n) {
    const float p0 = props[2 * i], p1 = props[2 * i + 1], g0 = gts[2 * i], g1 = gts[2 * i + 1];
    const float u0 = fminf(g0, p0), u1 = fmaxf(g1, p1), i0 = fmaxf(g0, p0), i1 = fminf(g1, p1);
    float iou = 0.f;
    if (u1 - u0 != 0.f) iou = fmaxf(0.f, (i1 - i0) / (u1 - u0));
    if (ious) ious[i] = iou;
    c3 = iou >= 0.3f; c5 = iou >= 0.5f; c7 = iou >= 0.7f; s = iou; cnt = 1;
  }
  __shared__ double red[5][4];
  double v[5] = {c3, c5, c7, cnt, s};
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
    if (lane == 0) red[k][wid] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < 5) atomicAdd(&acc[threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

}  // namespace

extern "C" int vmr_infer_basic(const float* slogits, const float* elogits, const float* vmask, float* frac, int* idx,
                               int B, int T, void* stream) {
  VMR_CHECK(slogits && elogits && vmask && frac && idx, "vmr_infer_basic: null pointer");
  VMR_CHECK(T >= 1 && T <= INF_MAX_T, "vmr_infer_basic: need 1 <= T <= %d", INF_MAX_T);
  if (B == 0) return 0;
  hipLaunchKernelGGL(infer_basic_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, slogits, elogits, vmask, frac, idx, T);
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_iou_metrics(const float* props, const float* gts, float* ious, double* acc, int n, void* stream) {
  VMR_CHECK(props && gts && acc, "vmr_iou_metrics: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(iou_metrics_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, props, gts, ious, acc, n);
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_weighted_pool_fwd(const void* x, const float* w, const float* mask, float* alpha, void* pooled, int B,
                                     int L, int D, int dtype, void* stream) {
  VMR_CHECK(x && w && mask && alpha && pooled, "vmr_weighted_pool_fwd: null pointer");
  VMR_CHECK(L >= 1 && L <= WP_MAX_L && D % 8 == 0, "vmr_weighted_pool_fwd: need 1 <= L <= %d and D %% 8 == 0", WP_MAX_L);
  if (B == 0) return 0;
  if (dtype == VMR_BF16)
    hipLaunchKernelGGL((weighted_pool_fwd_kernel<bf16_t>), dim3(B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w,
                       mask, alpha, (bf16_t*)pooled, L, D);
  else
    hipLaunchKernelGGL((weighted_pool_fwd_kernel<float>), dim3(B), dim3(256), 0, (hipStream_t)stream, (const float*)x, w,
                       mask, alpha, (float*)pooled, L, D);
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_weighted_pool_bwd(const void* dpooled, const void* x, const float* w, const float* alpha, void* dx,
                                     float* dw, int B, int L, int D, int dtype, void* stream) {
  VMR_CHECK(dpooled && x && w && alpha && dx && dw, "vmr_weighted_pool_bwd: null pointer");
  VMR_CHECK(L >= 1 && L <= WP_MAX_L && D % 8 == 0, "vmr_weighted_pool_bwd: need 1 <= L <= %d and D %% 8 == 0", WP_MAX_L);
  if (B == 0) return 0;
  if (dtype == VMR_BF16)
    hipLaunchKernelGGL((weighted_pool_bwd_kernel<bf16_t>), dim3(B), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dpooled, (const bf16_t*)x, w, alpha, (bf16_t*)dx, dw, L, D);
  else
    hipLaunchKernelGGL((weighted_pool_bwd_kernel<float>), dim3(B), dim3(256), 0, (hipStream_t)stream,
                       (const float*)dpooled, (const float*)x, w, alpha, (float*)dx, dw, L, D);
  VMR_LAUNCH_CHECK();
  return 0;
}
