// cosine.hip -- sim[b, c] = <q_b, y_bc> / (max(|y_bc|, 1e-30) (1 + 1e-8)) for q already unit-normalised: the similarity
// of BAN's ContrastLoss (reference models/BANlib/model.py:639-671: F.normalize-style division of the sentence projection
// and of every map cell's projection, their dot product over the contrast dimension) over the COMPACT cells [B, C, D].
// As torch ops this was a chain of full-size passes over the [64, 5376, 128] tensor and its fp32 copy (float cast, norm,
// clamp, divide, einsum as bmm -- and their backward: 1.6 ms of the 20 ms BAN step); here one pass forward (read y) and
// one backward (read y, write dy), 16-byte accesses, a row = D / 8 adjacent lanes, reduced with DPP inside a 16-lane row.
#include "common.h"

namespace {

// sum over the LPR adjacent lanes that share a row (LPR a power of two <= 16); every lane of the group gets the sum
template <int LPR> __device__ __forceinline__ float group_sum(float v) {
  if (LPR >= 2) v += dpp_get<0xB1>(v);     // quad_perm:[1,0,3,2]
  if (LPR >= 4) v += dpp_get<0x4E>(v);     // quad_perm:[2,3,0,1]
  if (LPR >= 8) v += dpp_get<0x141>(v);    // row_half_mirror
  if (LPR >= 16) v += dpp_get<0x140>(v);   // row_mirror
  return v;
}

constexpr float COS_K = 1.0f / (1.0f + 1e-8f);

template <typename T, int LPR>
__global__ __launch_bounds__(256) void cos_rows_fwd_kernel(const float* __restrict__ q, const T* __restrict__ y, float* __restrict__ sim,
                                                           float* __restrict__ rnorm, int B, int C) {
  constexpr int D = LPR * 8, RPW = 64 / LPR;                       // rows per wave-instruction
  const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
  const int64_t rows = (int64_t)B * C;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  for (int64_t r0 = wave * RPW; r0 < rows; r0 += nw * RPW) {
    const int64_t r = min(r0 + g, rows - 1);
    const int b = (int)(r / C);
    float yv[8], qv[8];
    Vec8<T>::load(y + r * D + l * 8, yv);
    Vec8<float>::load(q + (int64_t)b * D + l * 8, qv);
    float dot = 0.f, ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { dot += qv[e] * yv[e]; ss += yv[e] * yv[e]; }
    dot = group_sum<LPR>(dot);
    ss = group_sum<LPR>(ss);
    const float rn = 1.f / fmaxf(sqrtf(ss), 1e-30f);
    if (l == 0 && r0 + g < rows) { sim[r] = dot * rn * COS_K; rnorm[r] = rn; }
  }
}

// dy = ds (K r q - s r^2 y);  dq[b] += sum_c ds K r y   (per-workgroup partial in registers -> one atomic per channel)
template <typename T, int LPR>
__global__ __launch_bounds__(256) void cos_rows_bwd_kernel(const float* __restrict__ q, const T* __restrict__ y, const float* __restrict__ sim,
                                                           const float* __restrict__ rnorm, const float* __restrict__ dsim,
                                                           T* __restrict__ dy, float* __restrict__ dq, int B, int C, int chunks) {
  constexpr int D = LPR * 8, RPW = 64 / LPR;
  __shared__ float red[4][64][8];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane / LPR, l = lane % LPR;
  const int b = blockIdx.x / chunks, ch = blockIdx.x % chunks;
  const int per = (C + chunks - 1) / chunks;
  const int c0 = ch * per, c1 = min(C, c0 + per);
  float qv[8], acc[8];
  Vec8<float>::load(q + (int64_t)b * D + l * 8, qv);
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int cc = c0 + w * RPW; cc < c1; cc += 4 * RPW) {
    const int c = min(cc + g, c1 - 1);
    const bool ok = cc + g < c1;
    const int64_t r = (int64_t)b * C + c;
    float yv[8], o[8];
    Vec8<T>::load(y + r * D + l * 8, yv);
    const float ds = ok ? dsim[r] : 0.f, rn = rnorm[r], s = sim[r];
    const float a1 = ds * COS_K * rn, a2 = ds * s * rn * rn;
#pragma unroll
    for (int e = 0; e < 8; ++e) { o[e] = a1 * qv[e] - a2 * yv[e]; acc[e] += a1 * yv[e]; }
    if (ok) Vec8<T>::store(dy + r * D + l * 8, o);
  }
  // lanes with the same l (channel group) across the wave's row groups and across the four waves -> one sum per channel
#pragma unroll
  for (int e = 0; e < 8; ++e) red[w][lane][e] = acc[e];
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += 256) {
    const int ll = i >> 3, e = i & 7;
    float s = 0.f;
    for (int ww = 0; ww < 4; ++ww)
      for (int gg = 0; gg < RPW; ++gg) s += red[ww][gg * LPR + ll][e];
    atomicAdd(dq + (int64_t)b * D + i, s);
  }
}

}  // namespace

extern "C" int vmr_cos_rows_supported(int D) { return D == 8 || D == 16 || D == 32 || D == 64 || D == 128; }

#define COS_DISPATCH(D, ...)                                     \
  do {                                                           \
    switch ((D) / 8) {                                           \
      case 1: { constexpr int LPR = 1; __VA_ARGS__; } break;     \
      case 2: { constexpr int LPR = 2; __VA_ARGS__; } break;     \
      case 4: { constexpr int LPR = 4; __VA_ARGS__; } break;     \
      case 8: { constexpr int LPR = 8; __VA_ARGS__; } break;     \
      default: { constexpr int LPR = 16; __VA_ARGS__; } break;   \
    }                                                            \
  } while (0)

// q: fp32 [B, D] (unit rows), y: `dtype` [B, C, D]; sim, rnorm: fp32 [B, C] (rnorm = 1 / max(|y|, 1e-30), kept for the backward)
extern "C" int vmr_cos_rows_fwd(const float* q, const void* y, float* sim, float* rnorm, int B, int C, int D, int dtype, void* stream) {
  VMR_CHECK(q && y && sim && rnorm, "vmr_cos_rows_fwd: null pointer");
  VMR_CHECK(vmr_cos_rows_supported(D) && vmr_dtype_ok(dtype) && B >= 0 && C >= 0, "vmr_cos_rows_fwd: D = %d (8..128, power of two)", D);
  if (B == 0 || C == 0) return 0;
  const int64_t rows = (int64_t)B * C;
  const int rpb = 4 * (64 / (D / 8));
  const int grid = (int)min((int64_t)4096, (rows + rpb - 1) / rpb);
  VMR_DISPATCH(dtype, T, COS_DISPATCH(D, hipLaunchKernelGGL((cos_rows_fwd_kernel<T, LPR>), dim3(grid), dim3(256), 0, (hipStream_t)stream, q,
                                                             (const T*)y, sim, rnorm, B, C)));
  VMR_LAUNCH_CHECK();
  return 0;
}

// dy: `dtype` [B, C, D] (written); dq: fp32 [B, D] (ACCUMULATED: zero it first)
extern "C" int vmr_cos_rows_bwd(const float* q, const void* y, const float* sim, const float* rnorm, const float* dsim, void* dy,
                                float* dq, int B, int C, int D, int dtype, void* stream) {
  VMR_CHECK(q && y && sim && rnorm && dsim && dy && dq, "vmr_cos_rows_bwd: null pointer");
  VMR_CHECK(vmr_cos_rows_supported(D) && vmr_dtype_ok(dtype) && B >= 0 && C >= 0, "vmr_cos_rows_bwd: D = %d (8..128, power of two)", D);
  if (B == 0 || C == 0) return 0;
  const int chunks = max(1, min(64, (C + 255) / 256));
  VMR_DISPATCH(dtype, T, COS_DISPATCH(D, hipLaunchKernelGGL((cos_rows_bwd_kernel<T, LPR>), dim3(B * chunks), dim3(256), 0, (hipStream_t)stream, q,
                                                             (const T*)y, sim, rnorm, dsim, (T*)dy, dq, B, C, chunks)));
  VMR_LAUNCH_CHECK();
  return 0;
}
