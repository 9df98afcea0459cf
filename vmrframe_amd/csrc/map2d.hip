// map2d.hip -- BAN 2-D proposal map: the reference builds a [B, 3F, N, N] map with Python
// loops over diagonals -- SparseMaxPool / DenseMaxPool (models/BANlib/model.py:226-290: cell (i, j) = max over
// frames i..j of the content features, produced by a CASCADE of MaxPool1d(2|3|5, stride 1) so every diagonal
// is one more pooling of the previous one) and SparseBoundaryCat (:293-325: cell (i, j) = [start[i] | end[j]]) --
// then runs Linear(3F -> F) over all N*N cells, two thirds of them zeros.
//
// Here only the cells the mask keeps exist, in COMPACT cell-major order (diagonal by diagonal, as the reference's
// `maskij` list: the main diagonal first, then offset o_1, o_2, ...; within a diagonal by start frame i):
//   M[b, c, :] = max_{t in [i_c, j_c]} x[b, t, :]                    (content; feeds the Wc third of map2d_proj)
//   R[b, c, :] = Ps[b, i_c, :] + Pe[b, j_c, :]                       (boundary: Ps = start.Ws^T, Pe = end.We^T were
//                                                                     projected per FRAME, not per cell)
// so that map2d_proj(cat[start_i, end_j, pool_ij]) = act(M.Wc^T + b + R) is ONE K=F GEMM with R as the
// pre-activation residual (VMR_EPI_RES_PRE) over 1/3 of the cells at 1/3 of the K.
//
// Diagonals are described by `grow[k]` = how many frames diagonal k's window grows over diagonal k-1's
// (= MaxPool1d kernel size - 1: 1 for the first level, 2, 4 for the sparse levels; 1 everywhere for DenseMaxPool).
// Workgroup = (clip b, 64-channel slice); thread = (channel pair, start frame i mod 8); the cascade keeps two
// [N][64] fp32 images in LDS (ping-pong, one barrier per diagonal); every global access is a 4-byte-per-lane,
// 128-byte-per-row segment of a cell row.
#include "common.h"

namespace {

constexpr int MP_CH = 64;      // channels per workgroup
constexpr int MP_NMAX = 160;   // frames (backward LDS: N x 1 KiB of value, arg-max and gradient images)

template <typename T>
__global__ __launch_bounds__(256) void map2d_pool_fwd_kernel(const T* __restrict__ x, const T* __restrict__ ps,
                                                             const T* __restrict__ pe, int64_t ldp,
                                                             const int* __restrict__ grow, int ndiag, T* __restrict__ M,
                                                             T* __restrict__ R, int N, int F, int64_t C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* buf0 = reinterpret_cast<float2*>(smem);            // [N][32] channel pairs
  float2* buf1 = buf0 + (size_t)N * 32;
  const int pc = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int slices = F / MP_CH;
  const int b = blockIdx.x / slices, ch = (blockIdx.x % slices) * MP_CH + pc * 2;
  const T* xb = x + (int64_t)b * N * F + ch;
  const T* psb = ps ? ps + (int64_t)b * N * ldp + ch : nullptr;
  const T* peb = pe ? pe + (int64_t)b * N * ldp + ch : nullptr;
  T* Mb = M + (int64_t)b * C * F + ch;
  T* Rb = R ? R + (int64_t)b * C * F + ch : nullptr;
  for (int t = rg; t < N; t += 8) {
    float v[2];
    Vec2<T>::load(xb + (int64_t)t * F, v);
    buf0[t * 32 + pc] = make_float2(v[0], v[1]);
    Vec2<T>::store(Mb + (int64_t)t * F, v);
    if (Rb) {
      float a[2], e[2];
      Vec2<T>::load(psb + (int64_t)t * ldp, a);
      Vec2<T>::load(peb + (int64_t)t * ldp, e);
      a[0] += e[0]; a[1] += e[1];
      Vec2<T>::store(Rb + (int64_t)t * F, a);
    }
  }
  __syncthreads();
  int64_t cell = N;
  int o = 0;
  float2* src = buf0;
  float2* dst = buf1;
  for (int k = 0; k < ndiag; ++k) {
    const int g = grow[k];
    o += g;
    const int len = N - o;
    for (int i = rg; i < len; i += 8) {
      float2 v = src[i * 32 + pc];
      for (int s = 1; s <= g; ++s) {
        const float2 w = src[(i + s) * 32 + pc];
        v.x = fmaxf(v.x, w.x);
        v.y = fmaxf(v.y, w.y);
      }
      dst[i * 32 + pc] = v;
      float o2[2] = {v.x, v.y};
      Vec2<T>::store(Mb + (cell + i) * F, o2);
      if (Rb) {
        float a[2], e[2];
        Vec2<T>::load(psb + (int64_t)i * ldp, a);
        Vec2<T>::load(peb + (int64_t)(i + o) * ldp, e);
        a[0] += e[0]; a[1] += e[1];
        Vec2<T>::store(Rb + (cell + i) * F, a);
      }
    }
    __syncthreads();
    cell += len > 0 ? len : 0;
    float2* t2 = src; src = dst; dst = t2;
  }
}

// Backward of the max cascade: the arg-max frame of every cell is recomputed with the same cascade (first frame
// wins ties, as the chained MaxPool1d backward does) and dM is added into an LDS image of dx (ds_add_f32: different
// cells of one diagonal may share their arg-max frame).
template <typename T>
__global__ __launch_bounds__(256) void map2d_pool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dM,
                                                             const int* __restrict__ grow, int ndiag, T* __restrict__ dx,
                                                             int N, int F, int64_t C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* buf0 = reinterpret_cast<float2*>(smem);             // values [N][32]
  float2* buf1 = buf0 + (size_t)N * 32;
  int* idx0 = reinterpret_cast<int*>(buf1 + (size_t)N * 32);  // arg-max frames, two 16-bit fields per pair
  int* idx1 = idx0 + (size_t)N * 32;
  float* acc = reinterpret_cast<float*>(idx1 + (size_t)N * 32);   // dx image [N][64]
  const int pc = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int slices = F / MP_CH;
  const int b = blockIdx.x / slices, ch = (blockIdx.x % slices) * MP_CH + pc * 2;
  const T* xb = x + (int64_t)b * N * F + ch;
  const T* gb = dM + (int64_t)b * C * F + ch;
  for (int t = rg; t < N; t += 8) {
    float v[2], d[2];
    Vec2<T>::load(xb + (int64_t)t * F, v);
    Vec2<T>::load(gb + (int64_t)t * F, d);
    buf0[t * 32 + pc] = make_float2(v[0], v[1]);
    idx0[t * 32 + pc] = t | (t << 16);
    acc[t * MP_CH + pc * 2] = d[0];          // main-diagonal cells: the frame itself
    acc[t * MP_CH + pc * 2 + 1] = d[1];
  }
  __syncthreads();
  int64_t cell = N;
  int o = 0;
  float2* src = buf0;
  float2* dst = buf1;
  int* isrc = idx0;
  int* idst = idx1;
  for (int k = 0; k < ndiag; ++k) {
    const int g = grow[k];
    o += g;
    const int len = N - o;
    for (int i = rg; i < len; i += 8) {
      float2 v = src[i * 32 + pc];
      int id = isrc[i * 32 + pc];
      int ix = id & 0xFFFF, iy = id >> 16;
      for (int s = 1; s <= g; ++s) {
        const float2 w = src[(i + s) * 32 + pc];
        const int wi = isrc[(i + s) * 32 + pc];
        if (w.x > v.x) { v.x = w.x; ix = wi & 0xFFFF; }
        if (w.y > v.y) { v.y = w.y; iy = wi >> 16; }
      }
      dst[i * 32 + pc] = v;
      idst[i * 32 + pc] = ix | (iy << 16);
      float d[2];
      Vec2<T>::load(gb + (cell + i) * F, d);
      atomicAdd(&acc[ix * MP_CH + pc * 2], d[0]);
      atomicAdd(&acc[iy * MP_CH + pc * 2 + 1], d[1]);
    }
    __syncthreads();
    cell += len > 0 ? len : 0;
    float2* t2 = src; src = dst; dst = t2;
    int* t3 = isrc; isrc = idst; idst = t3;
  }
  T* dxb = dx + (int64_t)b * N * F + ch;
  for (int t = rg; t < N; t += 8) {
    float v[2] = {acc[t * MP_CH + pc * 2], acc[t * MP_CH + pc * 2 + 1]};
    Vec2<T>::store(dxb + (int64_t)t * F, v);
  }
}

// Backward of R = Ps[i] + Pe[j] in gather form: frame t collects the cells of map row t (into dPs) and of map
// column t (into dPe); one thread = 8 channels of one frame, one 16-byte load per diagonal and direction.
template <typename T>
__global__ __launch_bounds__(64) void map2d_dp_kernel(const T* __restrict__ dR, const int* __restrict__ grow, int ndiag,
                                                      T* __restrict__ dps, T* __restrict__ dpe, int64_t ldp, int N, int F,
                                                      int64_t C) {
  const int b = blockIdx.x / N, t = blockIdx.x % N;
  const T* gb = dR + (int64_t)b * C * F;
  for (int c8 = threadIdx.x * 8; c8 < F; c8 += 64 * 8) {
    float s[8], e[8], v[8];
    Vec8<T>::load(gb + (int64_t)t * F + c8, v);   // main diagonal: cell (t, t)
#pragma unroll
    for (int q = 0; q < 8; ++q) { s[q] = v[q]; e[q] = v[q]; }
    int64_t cell = N;
    int o = 0;
    for (int k = 0; k < ndiag; ++k) {
      o += grow[k];
      const int len = N - o;
      if (len <= 0) break;
      if (t < len) {            // cell (t, t + o): row t
        Vec8<T>::load(gb + (cell + t) * F + c8, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) s[q] += v[q];
      }
      if (t >= o) {             // cell (t - o, t): column t
        Vec8<T>::load(gb + (cell + t - o) * F + c8, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) e[q] += v[q];
      }
      cell += len;
    }
    Vec8<T>::store(dps + ((int64_t)b * N + t) * ldp + c8, s);
    Vec8<T>::store(dpe + ((int64_t)b * N + t) * ldp + c8, e);
  }
}

// dense <- compact: out[b, i, j, :] = cells[b, c(i,j), :] on the mask, `fill[:]` elsewhere (the value the reference
// computes for an all-zero cell: the layers' biases pushed through)
template <typename T>
__global__ __launch_bounds__(256) void map2d_scatter_kernel(const T* __restrict__ cells, const int* __restrict__ cell_of,
                                                            const float* __restrict__ fill, T* __restrict__ out, int N,
                                                            int W, int64_t C, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int w = (int)(idx % W);
    const int64_t bij = idx / W;
    const int ij = (int)(bij % ((int64_t)N * N));
    const int64_t b = bij / ((int64_t)N * N);
    const int c = cell_of[ij];
    out[idx] = c >= 0 ? cells[(b * C + c) * W + w] : from_f<T>(fill ? fill[w] : 0.f);
  }
}

int64_t count_cells(const int* grow_host, int ndiag, int N) {
  int64_t c = N;
  int o = 0;
  for (int k = 0; k < ndiag; ++k) {
    o += grow_host[k];
    if (N - o > 0) c += N - o;
  }
  return c;
}

}  // namespace

extern "C" int vmr_map2d_cells(const int32_t* grow_host, int ndiag, int N) {
  if (!grow_host || ndiag < 0 || N <= 0) return -1;
  return (int)count_cells(grow_host, ndiag, N);
}

extern "C" int vmr_map2d_pool_fwd(const void* x, const void* ps, const void* pe, int64_t ldp, const int32_t* grow,
                                  const int32_t* grow_host, int ndiag, void* M, void* R, int B, int N, int F, int dtype,
                                  void* stream) {
  VMR_CHECK(x && M && ((grow && grow_host) || ndiag == 0), "vmr_map2d_pool_fwd: null pointer");
  VMR_CHECK((R == nullptr) == (ps == nullptr) && (ps == nullptr) == (pe == nullptr),
            "vmr_map2d_pool_fwd: R, ps, pe come together");
  VMR_CHECK(N > 0 && N <= MP_NMAX && F % MP_CH == 0 && (!ps || ldp % 2 == 0),
            "vmr_map2d_pool_fwd: need N <= %d, F %% %d == 0 (N %d F %d)", MP_NMAX, MP_CH, N, F);
  for (int k = 0; k < ndiag; ++k) VMR_CHECK(grow_host[k] >= 1, "vmr_map2d_pool_fwd: grow[%d] < 1", k);
  if (B == 0) return 0;
  const int64_t C = count_cells(grow_host, ndiag, N);
  const size_t lds = (size_t)2 * N * 32 * sizeof(float2);
  if (dtype == VMR_BF16) {
    if (lds > 64 * 1024)
      VMR_CHECK(hipFuncSetAttribute((const void*)map2d_pool_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) == hipSuccess, "vmr_map2d_pool_fwd: LDS opt-in failed");
    hipLaunchKernelGGL(map2d_pool_fwd_kernel<bf16_t>, dim3(B * (F / MP_CH)), dim3(256), lds, (hipStream_t)stream,
                       (const bf16_t*)x, (const bf16_t*)ps, (const bf16_t*)pe, ldp, grow, ndiag, (bf16_t*)M, (bf16_t*)R, N, F, C);
  } else {
    if (lds > 64 * 1024)
      VMR_CHECK(hipFuncSetAttribute((const void*)map2d_pool_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) == hipSuccess, "vmr_map2d_pool_fwd: LDS opt-in failed");
    hipLaunchKernelGGL(map2d_pool_fwd_kernel<float>, dim3(B * (F / MP_CH)), dim3(256), lds, (hipStream_t)stream,
                       (const float*)x, (const float*)ps, (const float*)pe, ldp, grow, ndiag, (float*)M, (float*)R, N, F, C);
  }
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_map2d_pool_bwd(const void* x, const void* dM, const void* dR, const int32_t* grow,
                                  const int32_t* grow_host, int ndiag, void* dx, void* dps, void* dpe, int64_t ldp, int B,
                                  int N, int F, int dtype, void* stream) {
  VMR_CHECK(x && dM && dx && ((grow && grow_host) || ndiag == 0), "vmr_map2d_pool_bwd: null pointer");
  VMR_CHECK((dR == nullptr) == (dps == nullptr) && (dps == nullptr) == (dpe == nullptr),
            "vmr_map2d_pool_bwd: dR, dps, dpe come together");
  VMR_CHECK(N > 0 && N <= MP_NMAX && F % MP_CH == 0 && (!dR || (F % 8 == 0 && ldp % 8 == 0)),
            "vmr_map2d_pool_bwd: need N <= %d, F %% %d == 0 (N %d F %d)", MP_NMAX, MP_CH, N, F);
  if (B == 0) return 0;
  const int64_t C = count_cells(grow_host, ndiag, N);
  const size_t lds = (size_t)2 * N * 32 * sizeof(float2) + (size_t)2 * N * 32 * sizeof(int) + (size_t)N * MP_CH * sizeof(float);
  VMR_CHECK(lds <= 160 * 1024, "vmr_map2d_pool_bwd: N too large for the LDS images");
  if (dtype == VMR_BF16) {
    if (lds > 64 * 1024)
      VMR_CHECK(hipFuncSetAttribute((const void*)map2d_pool_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) == hipSuccess, "vmr_map2d_pool_bwd: LDS opt-in failed");
    hipLaunchKernelGGL(map2d_pool_bwd_kernel<bf16_t>, dim3(B * (F / MP_CH)), dim3(256), lds, (hipStream_t)stream,
                       (const bf16_t*)x, (const bf16_t*)dM, grow, ndiag, (bf16_t*)dx, N, F, C);
    VMR_LAUNCH_CHECK();
    if (dR) {
      hipLaunchKernelGGL(map2d_dp_kernel<bf16_t>, dim3(B * N), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)dR, grow, ndiag,
                         (bf16_t*)dps, (bf16_t*)dpe, ldp, N, F, C);
      VMR_LAUNCH_CHECK();
    }
  } else {
    if (lds > 64 * 1024)
      VMR_CHECK(hipFuncSetAttribute((const void*)map2d_pool_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) == hipSuccess, "vmr_map2d_pool_bwd: LDS opt-in failed");
    hipLaunchKernelGGL(map2d_pool_bwd_kernel<float>, dim3(B * (F / MP_CH)), dim3(256), lds, (hipStream_t)stream,
                       (const float*)x, (const float*)dM, grow, ndiag, (float*)dx, N, F, C);
    VMR_LAUNCH_CHECK();
    if (dR) {
      hipLaunchKernelGGL(map2d_dp_kernel<float>, dim3(B * N), dim3(64), 0, (hipStream_t)stream, (const float*)dR, grow, ndiag,
                         (float*)dps, (float*)dpe, ldp, N, F, C);
      VMR_LAUNCH_CHECK();
    }
  }
  return 0;
}

extern "C" int vmr_map2d_scatter(const void* cells, const int32_t* cell_of, const float* fill, void* out, int B, int N,
                                 int W, int64_t C, int dtype, void* stream) {
  VMR_CHECK(cells && cell_of && out, "vmr_map2d_scatter: null pointer");
  const int64_t total = (int64_t)B * N * N * W;
  if (total == 0) return 0;
  const int grid = (int)min((int64_t)65535, (total + 255) / 256);
  if (dtype == VMR_BF16)
    hipLaunchKernelGGL(map2d_scatter_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)cells, cell_of,
                       fill, (bf16_t*)out, N, W, C, total);
  else
    hipLaunchKernelGGL(map2d_scatter_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)cells, cell_of,
                       fill, (float*)out, N, W, C, total);
  VMR_LAUNCH_CHECK();
  return 0;
}
