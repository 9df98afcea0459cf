// pool.hip -- small fused heads of the SeqPAN path that were strings of tiny framework kernels:
//   * WeightedPool (reference models/layers.py:440-453): alpha = softmax_l(x.w + mask), pooled = sum_l alpha*x
//   * infer_basic (utils/engine.py:28-44) and the IoU / R1@k / mIoU bookkeeping of the train and eval loops
//     (utils/utils.py:161-185, models/loss.py:83-109)   [SURVEY.md 8f, row N4]
//   * the narrow (N <= 8) output heads as matrix-vector kernels
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
    s = wave_sum_dpp(s);
    if (lane == 0) ds[l] = s;
  }
  __syncthreads();
  float t = 0.f;
  for (int l = tid; l < L; l += 256) t += al[l] * ds[l];
  t = wave_sum_dpp(t);
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
  VMR_DISPATCH(dtype, T, hipLaunchKernelGGL((weighted_pool_fwd_kernel<T>), dim3(B), dim3(256), 0, (hipStream_t)stream, (const T*)x, w,
                       mask, alpha, (T*)pooled, L, D));
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_weighted_pool_bwd(const void* dpooled, const void* x, const float* w, const float* alpha, void* dx,
                                     float* dw, int B, int L, int D, int dtype, void* stream) {
  VMR_CHECK(dpooled && x && w && alpha && dx && dw, "vmr_weighted_pool_bwd: null pointer");
  VMR_CHECK(L >= 1 && L <= WP_MAX_L && D % 8 == 0, "vmr_weighted_pool_bwd: need 1 <= L <= %d and D %% 8 == 0", WP_MAX_L);
  if (B == 0) return 0;
  VMR_DISPATCH(dtype, T, hipLaunchKernelGGL((weighted_pool_bwd_kernel<T>), dim3(B), dim3(256), 0, (hipStream_t)stream,
                       (const T*)dpooled, (const T*)x, w, alpha, (T*)dx, dw, L, D));
  VMR_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ narrow heads
// y[m, 0:N] = x[m, :] . W[n, :]^T + b  for N <= 8 output channels (match head N = 4, start / end heads
// N = 1; reference models/SeqPAN.py:41,78 and layers.py:659-671 through Conv1D, layers.py:15-26).  A
// 128-wide MFMA tile would be >= 94 % padding: these are bandwidth-bound matrix-vector products.
// fwd: W (fp32 masters) staged in LDS once per workgroup, one wave per row, fp32 logits out.
// bwd: a streaming dx kernel (W in LDS) and a row-slab reduction that ACCUMULATES dW / db (fp32; 8 rows in flight per thread).
namespace {

constexpr int NL_MAXN = 8;
template <typename T> __device__ __forceinline__ float round_through_t(float v) { return to_f<T>(from_f<T>(v)); }

template <typename T, int N>
__global__ __launch_bounds__(256) void narrow_fwd_kernel(const T* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ y, int64_t M,
                                                         int K, int64_t ldx, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Ws = reinterpret_cast<float*>(smem);   // [N][K]
  for (int i = threadIdx.x * 4; i < N * K; i += 1024) *reinterpret_cast<f32x4*>(Ws + i) = *reinterpret_cast<const f32x4*>(W + i);
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  for (int64_t m = r0 + wid; m < min(M, r0 + rows_per_block); m += 4) {
    float acc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = 0.f;
    for (int i = lane * 8; i < K; i += 512) {
      float xv[8];
      Vec8<T>::load(x + m * ldx + i, xv);
#pragma unroll
      for (int n = 0; n < N; ++n) {
        float wv[8];
        Vec8<float>::load(Ws + n * K + i, wv);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[n] += xv[e] * wv[e];
      }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = wave_sum(acc[n]);
    if (lane == 0) {
#pragma unroll
      for (int n = 0; n < N; ++n) y[m * N + n] = acc[n] + (bias ? bias[n] : 0.f);
    }
  }
}

// dx[m, chunk] = sum_n dy[m,n] * W[n, chunk]: pure streaming, one thread per (row, 8-column chunk), W in LDS
template <typename T, int N>
__global__ __launch_bounds__(256) void narrow_dx_kernel(const float* __restrict__ dy, const float* __restrict__ W,
                                                        T* __restrict__ dx, int64_t M, int K, const T* __restrict__ add) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Ws = reinterpret_cast<float*>(smem);   // [N][K]
  for (int i = threadIdx.x * 4; i < N * K; i += 1024) *reinterpret_cast<f32x4*>(Ws + i) = *reinterpret_cast<const f32x4*>(W + i);
  __syncthreads();
  const int cpr = K / 8;
  const int64_t total = M * cpr;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / cpr;
    const int c = (int)(idx - m * cpr) * 8;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (add) Vec8<T>::load(add + m * K + c, o);   // the gradient of x's other consumer joins here (no separate add pass)
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const float g = dy[m * N + n];
      float wv[8];
      Vec8<float>::load(Ws + n * K + c, wv);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += g * wv[e];
    }
    Vec8<T>::store(dx + m * K + c, o);
  }
}

// dW[n, chunk] += sum_m dy[m,n] * x[m, chunk], db[n] += sum_m dy[m,n]: thread = one 8-column chunk over a slab of
// rows, 8 rows in flight per thread (the loop is latency-bound otherwise), one atomic per output per workgroup row group
template <typename T, int N>
__global__ __launch_bounds__(256) void narrow_dw_kernel(const float* __restrict__ dy, const T* __restrict__ x,
                                                        float* __restrict__ part, int64_t M, int K, int64_t ldx,
                                                        int rows_per_block) {
  const int cpr = K / 8;                         // chunks per row
  const int groups = max(1, 256 / cpr);          // row groups working in parallel
  const int chunk = threadIdx.x % cpr, grp = threadIdx.x / cpr;
  if (grp >= groups) return;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float gw[N][8], gb[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    gb[n] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) gw[n][e] = 0.f;
  }
  typedef __attribute__((ext_vector_type(8))) T TV8;
  for (int64_t mb = r0 + grp; mb < r1; mb += (int64_t)groups * 8) {
    TV8 xr[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {      // 8 independent row loads in flight (clamped; masked below)
      const int64_t m = min(mb + (int64_t)u * groups, M - 1);
      xr[u] = *reinterpret_cast<const TV8*>(x + m * ldx + chunk * 8);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t m = mb + (int64_t)u * groups;
      if (m >= r1) continue;
#pragma unroll
      for (int n = 0; n < N; ++n) {
        const float g = dy[m * N + n];
#pragma unroll
        for (int e = 0; e < 8; ++e) gw[n][e] += g * (float)xr[u][e];
        if (chunk == 0) gb[n] += g;
      }
    }
  }
  // one partial row [N*K + N] per (workgroup, row group): float atomics from every workgroup onto the same
  // N*K addresses serialise in L2 (measured 40-70 us for 131-524 k atomics); a second stage sums the rows
  float* prow = part + ((int64_t)blockIdx.x * groups + grp) * ((int64_t)N * K + N);
#pragma unroll
  for (int n = 0; n < N; ++n) {
    Vec8<float>::store(prow + (int64_t)n * K + chunk * 8, gw[n]);
    if (chunk == 0) prow[(int64_t)N * K + n] = gb[n];
  }
}

// out[j] += sum_r part[r][j]   (j < nk: dW, else db): blockIdx.y takes 16 partial rows (16 independent loads in
// flight per thread), so only nrows/16 adders meet on an address
__global__ __launch_bounds__(256) void narrow_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW,
                                                            float* __restrict__ db, int nrows, int nk, int nb) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= nk + nb) return;
  const int r0 = blockIdx.y * 16;
  float v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = (r0 + k < nrows) ? part[(int64_t)(r0 + k) * (nk + nb) + j] : 0.f;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += v[k];
  if (j < nk) atomicAdd(&dW[j], s);
  else if (db) atomicAdd(&db[j - nk], s);
}

template <typename T>
int launch_narrow_fwd(int N, dim3 grid, size_t lds, hipStream_t st, const void* x, const float* W, const float* bias, float* y,
                      int64_t M, int K, int64_t ldx, int rpb) {
#define VMR_NF(NN) hipLaunchKernelGGL((narrow_fwd_kernel<T, NN>), grid, dim3(256), lds, st, (const T*)x, W, bias, y, M, K, ldx, rpb)
  switch (N) {
    case 1: VMR_NF(1); break; case 2: VMR_NF(2); break; case 3: VMR_NF(3); break; case 4: VMR_NF(4); break;
    case 5: VMR_NF(5); break; case 6: VMR_NF(6); break; case 7: VMR_NF(7); break; default: VMR_NF(8); break;
  }
#undef VMR_NF
  return 0;
}

constexpr int NL_RPB = 32;    // rows per workgroup of the dW reduction (128 left 64-74 workgroups on 256 CUs, each thread a chain of 8 dependent row fetches: 15 us)
template <typename T>
int launch_narrow_bwd(int N, hipStream_t st, const float* dy, const void* x, const float* W, void* dx, float* dW, float* db,
                      float* ws, int64_t M, int K, int64_t ldx, const void* dx_add) {
  const dim3 gw((unsigned)((M + NL_RPB - 1) / NL_RPB));
  const dim3 gx((unsigned)min((int64_t)4096, (M * (K / 8) + 255) / 256));
  const size_t lds = (size_t)N * K * 4;
  const int groups = max(1, 256 / (K / 8));
#define VMR_NB(NN)                                                                                                    \
  do {                                                                                                                \
    if (dx) hipLaunchKernelGGL((narrow_dx_kernel<T, NN>), gx, dim3(256), lds, st, dy, W, (T*)dx, M, K, (const T*)dx_add); \
    hipLaunchKernelGGL((narrow_dw_kernel<T, NN>), gw, dim3(256), 0, st, dy, (const T*)x, ws, M, K, ldx, NL_RPB);       \
  } while (0)
  switch (N) {
    case 1: VMR_NB(1); break; case 2: VMR_NB(2); break; case 3: VMR_NB(3); break; case 4: VMR_NB(4); break;
    case 5: VMR_NB(5); break; case 6: VMR_NB(6); break; case 7: VMR_NB(7); break; default: VMR_NB(8); break;
  }
#undef VMR_NB
  hipLaunchKernelGGL(narrow_reduce_kernel, dim3(cdiv(N * K + N, 256), cdiv((int)gw.x * groups, 16)), dim3(256), 0, st, ws, dW, db,
                     (int)gw.x * groups, N * K, N);
  return 0;
}

// ---- the label-embedding fuse of the match head (reference models/SeqPAN.py:80-82):
//      y[m, :] = (res[m, :] + sum_n p[m, n] * E[:, n]) * rs[m],  E = label_embs fp32 [K][N] (N = 4 classes).
// As a GEMM it was a K = 8 (zero-padded) product forward and N = 8 / K = 8 products backward on 128-wide MFMA tiles
// (26 + 42 + 23 us); it is a rank-N update of a streamed matrix.
template <typename T, int N>
__global__ __launch_bounds__(256) void label_fuse_fwd_kernel(const float* __restrict__ p, const float* __restrict__ E,
                                                             const T* __restrict__ res, const float* __restrict__ rs,
                                                             T* __restrict__ y, int64_t M, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Ws = reinterpret_cast<float*>(smem);   // [N][K], transposed while staging
  for (int i = threadIdx.x; i < N * K; i += 256) Ws[(i % N) * K + i / N] = E[i];
  __syncthreads();
  const int cpr = K / 8;
  const int64_t total = M * cpr;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / cpr;
    const int c = (int)(idx - m * cpr) * 8;
    float o[8];
    Vec8<T>::load(res + m * K + c, o);
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const float g = p[m * N + n];
      float wv[8];
      Vec8<float>::load(Ws + n * K + c, wv);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += g * wv[e];
    }
    const float r = rs ? rs[m] : 1.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= r;
    Vec8<T>::store(y + m * K + c, o);
  }
}

// backward, first pass (one wave per row): dres[m, :] = dy[m, :] * rs[m]  and  dp[m, n] = dres[m, :] . E[:, n]
template <typename T, int N>
__global__ __launch_bounds__(256) void label_fuse_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ E,
                                                             const float* __restrict__ rs, T* __restrict__ dres,
                                                             float* __restrict__ dp, int64_t M, int K, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Ws = reinterpret_cast<float*>(smem);
  for (int i = threadIdx.x; i < N * K; i += 256) Ws[(i % N) * K + i / N] = E[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  for (int64_t m = r0 + wid; m < min(M, r0 + rows_per_block); m += 4) {
    float acc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = 0.f;
    const float r = rs ? rs[m] : 1.f;
    for (int i = lane * 8; i < K; i += 512) {
      float xv[8];
      Vec8<T>::load(dy + m * K + i, xv);
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] *= r;
      Vec8<T>::store(dres + m * K + i, xv);
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = round_through_t<T>(xv[e]);   // (the second pass reads dres as stored)
#pragma unroll
      for (int n = 0; n < N; ++n) {
        float wv[8];
        Vec8<float>::load(Ws + n * K + i, wv);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[n] += xv[e] * wv[e];
      }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = wave_sum(acc[n]);
    if (lane == 0) {
#pragma unroll
      for (int n = 0; n < N; ++n) dp[m * N + n] = acc[n];
    }
  }
}

// second stage of narrow_dw_kernel's partial rows into a K-major gradient: dE[k][n] += sum_r part[r][n*K + k]
__global__ __launch_bounds__(256) void narrow_reduce_t_kernel(const float* __restrict__ part, float* __restrict__ dE, int nrows,
                                                              int N, int K) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N * K) return;
  const int r0 = blockIdx.y * 16;
  float v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = (r0 + k < nrows) ? part[(int64_t)(r0 + k) * (N * K + N) + j] : 0.f;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += v[k];
  atomicAdd(&dE[(j % K) * N + j / K], s);
}

}  // namespace

extern "C" int vmr_label_fuse_fwd(const float* p, const float* E, const void* res, const float* rowscale, void* y, int64_t M,
                                  int N, int K, int dtype, void* stream) {
  VMR_CHECK(p && E && res && y, "vmr_label_fuse_fwd: null pointer");
  VMR_CHECK(N == 4 && K % 8 == 0 && K <= 2048, "vmr_label_fuse_fwd: N must be 4, K %% 8 == 0, K <= 2048 (N=%d K=%d)", N, K);
  if (M == 0) return 0;
  const dim3 grid((unsigned)min((int64_t)4096, (M * (K / 8) + 255) / 256));
  const size_t lds = (size_t)N * K * 4;
  VMR_DISPATCH(dtype, T, hipLaunchKernelGGL((label_fuse_fwd_kernel<T, 4>), grid, dim3(256), lds, (hipStream_t)stream, p, E, (const T*)res,
                       rowscale, (T*)y, M, K));
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_label_fuse_bwd(const void* dy, const float* p, const float* E, const float* rowscale, void* dres, float* dp,
                                  float* dE, float* workspace, int64_t M, int N, int K, int dtype, void* stream) {
  VMR_CHECK(dy && p && E && dres && dp && dE && workspace, "vmr_label_fuse_bwd: null pointer");
  VMR_CHECK(N == 4 && K % 8 == 0 && K <= 2048, "vmr_label_fuse_bwd: N must be 4, K %% 8 == 0, K <= 2048 (N=%d K=%d)", N, K);
  if (M == 0) return 0;
  const int rpb = 32;
  const dim3 g1((unsigned)((M + rpb - 1) / rpb));
  const size_t lds = (size_t)N * K * 4;
  const dim3 gw((unsigned)((M + NL_RPB - 1) / NL_RPB));
  const int groups = max(1, 256 / (K / 8));
  VMR_DISPATCH(dtype, T, {
    hipLaunchKernelGGL((label_fuse_bwd_kernel<T, 4>), g1, dim3(256), lds, (hipStream_t)stream, (const T*)dy, E, rowscale,
                       (T*)dres, dp, M, K, rpb);
    hipLaunchKernelGGL((narrow_dw_kernel<T, 4>), gw, dim3(256), 0, (hipStream_t)stream, p, (const T*)dres, workspace, M, K,
                       (int64_t)K, NL_RPB);
  });
  hipLaunchKernelGGL(narrow_reduce_t_kernel, dim3(cdiv(N * K, 256), cdiv((int)gw.x * groups, 16)), dim3(256), 0, (hipStream_t)stream,
                     workspace, dE, (int)gw.x * groups, N, K);
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_narrow_linear_fwd(const void* x, const float* W, const float* bias, float* y, int64_t M, int N, int K,
                                     int64_t ldx, int dtype, void* stream) {
  VMR_CHECK(x && W && y, "vmr_narrow_linear_fwd: null pointer");
  VMR_CHECK(N >= 1 && N <= NL_MAXN && K % 8 == 0 && ldx % 8 == 0 && ldx >= K && (size_t)N * K * 4 <= 64 * 1024,
            "vmr_narrow_linear_fwd: need 1 <= N <= 8, K %% 8 == 0, N*K*4 <= 64 KiB (N=%d K=%d)", N, K);
  if (M == 0) return 0;
  const int rpb = 32;
  dim3 grid((unsigned)((M + rpb - 1) / rpb));
  const size_t lds = (size_t)N * K * 4;
  VMR_DISPATCH(dtype, T, launch_narrow_fwd<T>(N, grid, lds, (hipStream_t)stream, x, W, bias, y, M, K, ldx, rpb));
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_narrow_linear_bwd_add(const float* dy, const void* x, const float* W, void* dx /*nullable*/, const void* dx_add,
                                         float* dW, float* db /*nullable*/, float* workspace, int64_t M, int N, int K,
                                         int64_t ldx, int dtype, void* stream);
extern "C" int vmr_narrow_linear_bwd(const float* dy, const void* x, const float* W, void* dx /*nullable*/, float* dW,
                                     float* db /*nullable*/, float* workspace, int64_t M, int N, int K, int64_t ldx,
                                     int dtype, void* stream) {
  return vmr_narrow_linear_bwd_add(dy, x, W, dx, nullptr, dW, db, workspace, M, N, K, ldx, dtype, stream);
}

extern "C" int vmr_narrow_linear_bwd_add(const float* dy, const void* x, const float* W, void* dx /*nullable*/, const void* dx_add,
                                         float* dW, float* db /*nullable*/, float* workspace, int64_t M, int N, int K,
                                         int64_t ldx, int dtype, void* stream) {
  VMR_CHECK(dy && x && W && dW && workspace, "vmr_narrow_linear_bwd: null pointer");
  VMR_CHECK(!dx_add || dx, "vmr_narrow_linear_bwd_add: dx_add without dx");
  VMR_CHECK(N >= 1 && N <= NL_MAXN && K % 8 == 0 && K <= 2048 && ldx % 8 == 0 && ldx >= K && (size_t)N * K * 4 <= 64 * 1024,
            "vmr_narrow_linear_bwd: need 1 <= N <= 8, K %% 8 == 0, K <= 2048, N*K*4 <= 64 KiB (N=%d K=%d)", N, K);
  if (M == 0) return 0;
  VMR_DISPATCH(dtype, T, launch_narrow_bwd<T>(N, (hipStream_t)stream, dy, x, W, dx, dW, db, workspace, M, K, ldx, dx_add));
  VMR_LAUNCH_CHECK();
  return 0;
}
