// pool.hip -- small fused heads of the SeqPAN path that were strings of tiny framework kernels:
//   * WeightedPool (reference models/layers.py:440-453): alpha = softmax_l(x.w + mask), pooled = sum_l alpha*x
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

}  // namespace

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
