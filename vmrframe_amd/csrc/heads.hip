// heads.hip -- the match head and its loss (reference models/SeqPAN.py:78-82, models/loss.py:24-41), which were
// strings of ~50 tiny framework launches on [B,T,4] tensors:
//   * Gumbel-softmax over C <= 8 classes (F.gumbel_softmax, tau = 0.3): noise either given (parity tests) or
//     drawn in-kernel from the counter hash; also emits the zero-padded compute-dtype copy the label-embedding
//     GEMM consumes
//   * lossfun_match: masked mean of -p[label] plus the Frobenius norm of the off-diagonal Gram matrix of the
//     label embeddings; forward and backward in one launch each
#include "common.h"

namespace {

constexpr int HC_MAX = 8;

template <typename T>
__global__ __launch_bounds__(256) void gumbel_softmax_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ noise,
                                                                 float inv_tau, uint32_t seed0, const uint32_t* __restrict__ step,
                                                                 float* __restrict__ probs, T* __restrict__ padded, int64_t R,
                                                                 int C, int ldo) {
  const uint32_t seed = vmr_seed(seed0, step);
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < R; r += (int64_t)gridDim.x * 256) {
    float v[HC_MAX];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < HC_MAX; ++c) {
      if (c < C) {
        float g;
        if (noise) g = noise[r * C + c];
        else {   // g = -log(E), E ~ Exp(1) = -log(U): one 32-bit hash per element, U strictly inside (0,1).
          // 23 bits + 0.5 is exact in fp32 (U <= 1 - 2^-24); with 24 bits the top value rounded to U = 1.0, g = +inf,
          // and one element in 2^24 turned the whole step into NaNs (seen at replay 32 of a 200-step BaseFast run).
          const uint2 h = vmr_hash4(seed, (uint64_t)r * C + c);
          const float u = ((float)(h.x >> 9) + 0.5f) * (1.0f / 8388608.0f);
          g = -__logf(-__logf(u));
        }
        v[c] = (logits[r * C + c] + g) * inv_tau;
        mx = fmaxf(mx, v[c]);
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < HC_MAX; ++c)
      if (c < C) { v[c] = expf(v[c] - mx); sum += v[c]; }
    const float inv = 1.f / sum;
#pragma unroll
    for (int c = 0; c < HC_MAX; ++c)
      if (c < C) probs[r * C + c] = v[c] * inv;
    if (padded)
      for (int c = 0; c < ldo; ++c) padded[r * ldo + c] = from_f<T>(c < C ? v[c] * inv : 0.f);
  }
}

// dlogits = inv_tau * p * (dp - sum_c p*dp), dp = dprobs (nullable) + dpadded[:, :C] (nullable)
template <typename T>
__global__ __launch_bounds__(256) void gumbel_softmax_bwd_kernel(const float* __restrict__ dprobs, const T* __restrict__ dpadded,
                                                                 const float* __restrict__ probs, float inv_tau,
                                                                 float* __restrict__ dlogits, int64_t R, int C, int ldo) {
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < R; r += (int64_t)gridDim.x * 256) {
    float dp[HC_MAX], p[HC_MAX];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < HC_MAX; ++c)
      if (c < C) {
        p[c] = probs[r * C + c];
        dp[c] = (dprobs ? dprobs[r * C + c] : 0.f) + (dpadded ? to_f<T>(dpadded[r * ldo + c]) : 0.f);
        dot += p[c] * dp[c];
      }
#pragma unroll
    for (int c = 0; c < HC_MAX; ++c)
      if (c < C) dlogits[r * C + c] = inv_tau * p[c] * (dp[c] - dot);
  }
}

// stats[2*b .. 2*b+1] = workgroup b's {numerator sum_r -p[r,label]*vmask[r], denominator sum_r vmask[r]}: plain
// stores, summed by the final kernel -- no atomics (deterministic) and nothing to zero beforehand.  (A
// hipMemsetAsync of the accumulators used to sit here; captured into a hipGraph as a memset node it was observed to
// run out of order with the kernels around it on replay, tests/test_gpu_trainer.py::test_graph_replay_equals_eager_steps.)
__global__ __launch_bounds__(256) void match_loss_partial_kernel(const float* __restrict__ probs, const int64_t* __restrict__ labels,
                                                                 const float* __restrict__ vmask, float* __restrict__ stats,
                                                                 int64_t R, int C) {
  float num = 0.f, den = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < R; r += (int64_t)gridDim.x * 256) {
    const float m = vmask[r];
    const int64_t l = labels[r];
    if (l >= 0 && l < C) num -= probs[r * C + l] * m;
    den += m;
  }
  __shared__ float part[2][4];
  num = wave_sum(num);
  den = wave_sum(den);
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = num; part[1][threadIdx.x >> 6] = den; }
  __syncthreads();
  if (threadIdx.x == 0) {
    stats[2 * blockIdx.x] = (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]);
    stats[2 * blockIdx.x + 1] = (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
  }
}

// one workgroup: Gram matrix G = E^T E of the label embeddings E [D,C], off-diagonal Frobenius norm;
// loss = stats[0]/(stats[1]+1e-12) + norm; aux = {G (C*C), norm, denominator}
__global__ __launch_bounds__(256) void match_loss_final_kernel(const float* __restrict__ E, const float* __restrict__ stats,
                                                               int nparts, float* __restrict__ loss, float* __restrict__ aux,
                                                               int D, int C) {
  __shared__ float red[4][HC_MAX * HC_MAX];
  __shared__ float tot[2][4];
  {   // sum of the per-workgroup partials (nparts <= 256: one per thread)
    float n_ = threadIdx.x < nparts ? stats[2 * threadIdx.x] : 0.f;
    float d_ = threadIdx.x < nparts ? stats[2 * threadIdx.x + 1] : 0.f;
    n_ = wave_sum_dpp(n_);
    d_ = wave_sum_dpp(d_);
    if ((threadIdx.x & 63) == 0) { tot[0][threadIdx.x >> 6] = n_; tot[1][threadIdx.x >> 6] = d_; }
  }
  float g[HC_MAX * HC_MAX];
#pragma unroll
  for (int i = 0; i < HC_MAX * HC_MAX; ++i) g[i] = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) {
    float e[HC_MAX];
#pragma unroll
    for (int c = 0; c < HC_MAX; ++c) e[c] = c < C ? E[(int64_t)d * C + c] : 0.f;
#pragma unroll
    for (int i = 0; i < HC_MAX; ++i)
#pragma unroll
      for (int j = 0; j < HC_MAX; ++j) g[i * HC_MAX + j] += e[i] * e[j];
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < HC_MAX * HC_MAX; ++i) {
    const float s = wave_sum_dpp(g[i]);      // (16 + 2 wave reductions: as ds_bpermute butterflies they were most of this kernel)
    if (lane == 0) red[wid][i] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float nrm = 0.f;
    for (int i = 0; i < C; ++i)
      for (int j = 0; j < C; ++j) {
        const float v = red[0][i * HC_MAX + j] + red[1][i * HC_MAX + j] + red[2][i * HC_MAX + j] + red[3][i * HC_MAX + j];
        aux[i * C + j] = v;
        if (i != j) nrm += v * v;
      }
    nrm = sqrtf(nrm);
    const float num = (tot[0][0] + tot[0][1]) + (tot[0][2] + tot[0][3]);
    const float den = (tot[1][0] + tot[1][1]) + (tot[1][2] + tot[1][3]);
    aux[C * C] = nrm;
    aux[C * C + 1] = den + 1e-12f;
    loss[0] = num / (den + 1e-12f) + nrm;
  }
}

// dprobs[r,c] = -dloss * vmask[r] / denom at c == label; dE[d,j] += dloss * 2/norm * sum_{i != j} E[d,i] G[i,j]
__global__ __launch_bounds__(256) void match_loss_bwd_kernel(const float* __restrict__ dloss, const int64_t* __restrict__ labels,
                                                             const float* __restrict__ vmask, const float* __restrict__ E,
                                                             const float* __restrict__ aux, float* __restrict__ dprobs,
                                                             float* __restrict__ dE, int64_t R, int D, int C) {
  const float gl = dloss[0];
  const float nrm = aux[C * C], den = aux[C * C + 1];
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  for (int64_t r = tid; r < R; r += nth) {
    const int64_t l = labels[r];
    const float v = -gl * vmask[r] / den;
    for (int c = 0; c < C; ++c) dprobs[r * C + c] = (c == l) ? v : 0.f;
  }
  if (dE && nrm > 0.f) {
    for (int64_t i = tid; i < (int64_t)D * C; i += nth) {
      const int d = (int)(i / C), j = (int)(i - (int64_t)d * C);
      float s = 0.f;
      for (int k = 0; k < C; ++k)
        if (k != j) s += E[(int64_t)d * C + k] * aux[k * C + j];
      atomicAdd(&dE[i], gl * 2.f * s / nrm);
    }
  }
}

}  // namespace

extern "C" int vmr_gumbel_softmax_fwd(const float* logits, const float* noise, float tau, uint32_t seed, const uint32_t* step,
                                      float* probs, void* padded, int64_t R, int C, int ldo, int dtype, void* stream) {
  VMR_CHECK(logits && probs && tau > 0.f, "vmr_gumbel_softmax_fwd: bad arguments");
  VMR_CHECK(C >= 1 && C <= HC_MAX && (!padded || ldo >= C), "vmr_gumbel_softmax_fwd: need 1 <= C <= %d and ldo >= C", HC_MAX);
  if (R == 0) return 0;
  const dim3 grid((unsigned)min((int64_t)2048, (R + 255) / 256));
  VMR_DISPATCH(dtype, T, hipLaunchKernelGGL(gumbel_softmax_fwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, logits, noise, 1.f / tau, seed,
                       step, probs, (T*)padded, R, C, ldo));
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_gumbel_softmax_bwd(const float* dprobs, const void* dpadded, const float* probs, float tau, float* dlogits,
                                      int64_t R, int C, int ldo, int dtype, void* stream) {
  VMR_CHECK(probs && dlogits && tau > 0.f && (dprobs || dpadded), "vmr_gumbel_softmax_bwd: bad arguments");
  VMR_CHECK(C >= 1 && C <= HC_MAX, "vmr_gumbel_softmax_bwd: need 1 <= C <= %d", HC_MAX);
  if (R == 0) return 0;
  const dim3 grid((unsigned)min((int64_t)2048, (R + 255) / 256));
  VMR_DISPATCH(dtype, T, hipLaunchKernelGGL(gumbel_softmax_bwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, dprobs, (const T*)dpadded,
                       probs, 1.f / tau, dlogits, R, C, ldo));
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_match_loss_fwd(const float* probs, const int64_t* labels, const float* vmask, const float* E, float* loss,
                                  float* aux /*[C*C+2] + VMR_MATCH_LOSS_SCRATCH scratch floats*/, int64_t R, int D, int C,
                                  void* stream) {
  VMR_CHECK(probs && labels && vmask && E && loss && aux, "vmr_match_loss_fwd: null pointer");
  VMR_CHECK(C >= 1 && C <= HC_MAX, "vmr_match_loss_fwd: need 1 <= C <= %d", HC_MAX);
  float* stats = aux + C * C + 2;
  const int nparts = R > 0 ? (int)min((int64_t)256, (R + 255) / 256) : 0;
  static_assert(VMR_MATCH_LOSS_SCRATCH >= 2 * 256, "scratch holds one {num, den} pair per workgroup");
  if (R > 0)
    hipLaunchKernelGGL(match_loss_partial_kernel, dim3((unsigned)nparts), dim3(256), 0, (hipStream_t)stream, probs, labels,
                       vmask, stats, R, C);
  hipLaunchKernelGGL(match_loss_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, E, stats, nparts, loss, aux, D, C);
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_match_loss_bwd(const float* dloss, const int64_t* labels, const float* vmask, const float* E, const float* aux,
                                  float* dprobs, float* dE /*nullable, accumulated*/, int64_t R, int D, int C, void* stream) {
  VMR_CHECK(dloss && labels && vmask && E && aux && dprobs, "vmr_match_loss_bwd: null pointer");
  VMR_CHECK(C >= 1 && C <= HC_MAX, "vmr_match_loss_bwd: need 1 <= C <= %d", HC_MAX);
  const int64_t work = R > (int64_t)D * C ? R : (int64_t)D * C;
  if (work == 0) return 0;
  hipLaunchKernelGGL(match_loss_bwd_kernel, dim3((unsigned)min((int64_t)1024, (work + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, dloss, labels, vmask, E, aux, dprobs, dE, R, D, C);
  VMR_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ CQAttention rank-1 prep
// y[r,:] = x[r,:] * a + b  (a, b fp32 [D]; x, y compute dtype): the operand C*w4mlu + w4Q (or Q*w4mlu + w4C) of
// the trilinear score (reference models/layers.py:427-437), always built on the SHORT stream.
// bwd: dx = dy * a; da[d] += sum_r dy*x; db[d] += sum_r dy (row slabs over blockIdx.y, one atomic per slab).
namespace {

template <typename T>
__global__ __launch_bounds__(256) void scale_shift_fwd_kernel(const T* __restrict__ x, const float* __restrict__ a,
                                                              const float* __restrict__ b, T* __restrict__ y, int64_t rows, int D) {
  const int64_t total = rows * (D / 8);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % (D / 8)) * 8;
    float xv[8], av[8], bv[8];
    Vec8<T>::load(x + i * 8, xv);
    Vec8<float>::load(a + c, av);
    Vec8<float>::load(b + c, bv);
#pragma unroll
    for (int e = 0; e < 8; ++e) xv[e] = xv[e] * av[e] + bv[e];
    Vec8<T>::store(y + i * 8, xv);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void scale_shift_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                              const float* __restrict__ a, T* __restrict__ dx,
                                                              float* __restrict__ da, float* __restrict__ db, int64_t rows, int D,
                                                              int rows_per_block) {
  __shared__ float red[2][8][256];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 256 + cg * 8;
  const int64_t rbeg = (int64_t)blockIdx.y * rows_per_block, rend = min(rows, rbeg + rows_per_block);
  float sa[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < D) {
    float av[8];
    Vec8<float>::load(a + c0, av);
    for (int64_t r = rbeg + rl; r < rend; r += 8) {
      float g[8], xv[8], o[8];
      Vec8<T>::load(dy + r * D + c0, g);
      Vec8<T>::load(x + r * D + c0, xv);
#pragma unroll
      for (int e = 0; e < 8; ++e) { o[e] = g[e] * av[e]; sa[e] += g[e] * xv[e]; sb[e] += g[e]; }
      Vec8<T>::store(dx + r * D + c0, o);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[0][rl][cg * 8 + e] = sa[e]; red[1][rl][cg * 8 + e] = sb[e]; }
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < D) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { s0 += red[0][k][threadIdx.x]; s1 += red[1][k][threadIdx.x]; }
    atomicAdd(&da[c], s0);
    atomicAdd(&db[c], s1);
  }
}

}  // namespace

extern "C" int vmr_scale_shift_fwd(const void* x, const float* a, const float* b, void* y, int64_t rows, int D, int dtype,
                                   void* stream) {
  VMR_CHECK(x && a && b && y && D % 8 == 0, "vmr_scale_shift_fwd: bad arguments");
  if (rows == 0) return 0;
  const dim3 grid((unsigned)min((int64_t)4096, (rows * (D / 8) + 255) / 256));
  VMR_DISPATCH(dtype, T, hipLaunchKernelGGL(scale_shift_fwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, a, b, (T*)y,
                       rows, D));
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_scale_shift_bwd(const void* dy, const void* x, const float* a, void* dx, float* da, float* db, int64_t rows,
                                   int D, int dtype, void* stream) {
  VMR_CHECK(dy && x && a && dx && da && db && D % 8 == 0, "vmr_scale_shift_bwd: bad arguments");
  if (rows == 0) return 0;
  const int gx = cdiv(D, 256);
  int gy = (int)min((int64_t)(512 / gx > 0 ? 512 / gx : 1), (rows + 31) / 32);
  const int rpb = (int)((rows + gy - 1) / gy);
  gy = (int)((rows + rpb - 1) / rpb);
  VMR_DISPATCH(dtype, T, hipLaunchKernelGGL(scale_shift_bwd_kernel<T>, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const T*)dy,
                       (const T*)x, a, (T*)dx, da, db, rows, D, rpb));
  VMR_LAUNCH_CHECK();
  return 0;
}
