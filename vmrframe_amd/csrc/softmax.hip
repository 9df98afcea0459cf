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
  if (dtype == VMR_BF16)
    SM_DISPATCH(ldP, hipLaunchKernelGGL((softmax_fwd_kernel<bf16_t, MN>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                        S, (bf16_t*)P, (bf16_t*)Pkeep, rmask, cmask, mode, nrows, H, R, C, ldS, ldP,
                                        cm_stride, scale, drop_p, drop_seed, drop_step));
  else
    SM_DISPATCH(ldP, hipLaunchKernelGGL((softmax_fwd_kernel<float, MN>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                        S, (float*)P, (float*)Pkeep, rmask, cmask, mode, nrows, H, R, C, ldS, ldP,
                                        cm_stride, scale, drop_p, drop_seed, drop_step));
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
  if (dtype == VMR_BF16)
    SM_DISPATCH(ldP, hipLaunchKernelGGL((softmax_bwd_kernel<bf16_t, MN>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                        dP, (const bf16_t*)P, (bf16_t*)dS, nrows, C, ldS, ldP, scale, drop_p, drop_seed,
                                        drop_step));
  else
    SM_DISPATCH(ldP, hipLaunchKernelGGL((softmax_bwd_kernel<float, MN>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                        dP, (const float*)P, (float*)dS, nrows, C, ldS, ldP, scale, drop_p, drop_seed,
                                        drop_step));
  VMR_LAUNCH_CHECK();
  return 0;
}
