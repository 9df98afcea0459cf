// staging.hip -- input staging on the device: the per-clip Python loops of the reference's
// data path -- interpolate_avrage / sample_vfeat_linear (utils/data_utils.py:161-201), pad_video_seq (:70-84) and
// convert_length_to_mask (utils/utils.py:125-130) -- as ONE kernel over a device-resident feature arena.
// All video features live once in HBM ([sum of frames, V] fp32, 288 GB per GPU holds whole datasets); a batch is
// described by B row offsets and B x (T+1) segment boundaries (computed on the host exactly as the reference does,
// float32 + round-half-to-even); output row i of clip b is mean(x[s:e]) for s < e, else x[s]; rows >= out_len are
// zero and masked out.
#include "common.h"

namespace {

template <typename TD>
__global__ __launch_bounds__(256) void resample_pad_kernel(const float* __restrict__ arena, const int64_t* __restrict__ row_off,
                                                           const int* __restrict__ seg, const int* __restrict__ out_len,
                                                           TD* __restrict__ out, float* __restrict__ mask, int B, int T, int V,
                                                           int64_t ldo) {
  const int vq = (V + 3) / 4;                                   // 4 columns per thread
  const int64_t total = (int64_t)B * T * vq;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(idx % vq) * 4;
    const int64_t bi = idx / vq;
    const int b = (int)(bi / T), i = (int)(bi - (int64_t)b * T);
    const int n = out_len[b];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < n) {
      const int s = seg[(int64_t)b * (T + 1) + i], e = seg[(int64_t)b * (T + 1) + i + 1];
      const float* base = arena + row_off[b] * V;
      const int last = s < e ? e : s + 1;
      for (int r = s; r < last; ++r) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (c4 + k < V) acc[k] += base[(int64_t)r * V + c4 + k];
      }
      if (s < e) {
        const float cnt = (float)(e - s);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] /= cnt;              // torch.mean: sum, then one division
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (c4 + k < V) out[((int64_t)b * T + i) * ldo + c4 + k] = from_f<TD>(acc[k]);
    if (c4 == 0 && mask) mask[(int64_t)b * T + i] = i < n ? 1.f : 0.f;
  }
}

}  // namespace

extern "C" int vmr_resample_pad(const float* arena, const int64_t* row_off, const int* seg, const int* out_len, void* out,
                                float* mask, int B, int T, int V, int64_t ldo, int out_dtype, void* stream) {
  VMR_CHECK(arena && row_off && seg && out_len && out, "vmr_resample_pad: null pointer");
  VMR_CHECK(T >= 1 && V >= 1 && ldo >= V, "vmr_resample_pad: bad sizes");
  if (B == 0) return 0;
  const int64_t total = (int64_t)B * T * ((V + 3) / 4);
  const dim3 grid((unsigned)min((int64_t)8192, (total + 255) / 256));
  VMR_CHECK(vmr_dtype_ok(out_dtype), "vmr_resample_pad: bad dtype %d", out_dtype);
  VMR_DISPATCH(out_dtype, TO,
               hipLaunchKernelGGL(resample_pad_kernel<TO>, grid, dim3(256), 0, (hipStream_t)stream, arena, row_off, seg, out_len,
                                  (TO*)out, mask, B, T, V, ldo));
  VMR_LAUNCH_CHECK();
  return 0;
}
