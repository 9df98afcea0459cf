// lstm.hip -- the pointwise half of the BAN encoders' bidirectional LSTM (row N2: reference models/BANlib/model.py:8-86,
// QueryEncoder / VisualEncoder = nn.LSTM(batch_first, bidirectional, packed by length, zero initial state)).
//
// One recurrence step s of BOTH directions at once.  The matrix halves (x.W_ih^T + b for all steps up front, h.W_hh^T per
// step) are vmr_gemm launches issued by the host layer; these kernels do the gate math, the state update, the length
// masking and the scatter of h into the [B, T, 2H] output -- and, backward, the gate gradients.
//
// Step order instead of time order: direction 0 handles time t = s, direction 1 time t = len_b - 1 - s, both only while
// s < len_b.  That is exactly what packing does (a sample's reverse pass starts at its own last valid step from a zero
// state; nothing is computed or emitted past its length), and it makes the mask the same for both directions.  The host
// builds direction 1's input projection on the per-sample reversed sequence (vmr_lstm_reverse_rows), so both
// directions index their x-part by s.
//
// Layouts (z = direction, b = sample, j = hidden unit; gate order i, f, g, o as in torch):
//   gx   [2][B][T][4H]  T   x-part of the pre-activations incl. both biases, by STEP
//   gh   [2][B][4H]     f32 h_{s-1} . W_hh^T of this step
//   c    [2][B][H]      f32 cell state (in/out)             hs [2][B][H] T: h for the next step's product (in/out)
//   act  [2][B][T][4H]  T   post-activation gates, by step (saved for backward)
//   cs   [2][B][T][H]   f32 cell state AFTER step s         hp [2][B][T][H] T: h BEFORE step s (= the dW_hh operand)
//   y    [B][T][2H]     T   output, by TIME, zero past len_b (the caller zero-fills once)
#include "common.h"

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) {          // 1 - 2 / (e^{2x} + 1): exact limits at +-inf, no overflow to NaN
  return 1.f - 2.f / (__expf(2.f * x) + 1.f);
}

template <typename T>
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(const T* __restrict__ gx, const float* __restrict__ gh,
                                                            const int* __restrict__ len, float* __restrict__ c,
                                                            T* __restrict__ hs, T* __restrict__ act, float* __restrict__ cs,
                                                            T* __restrict__ hp, T* __restrict__ y, int B, int Tn, int H, int s) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;        // (z, b, j)
  if (idx >= (int64_t)2 * B * H) return;
  const int j = (int)(idx % H), b = (int)((idx / H) % B), z = (int)(idx / ((int64_t)H * B));
  const int64_t zb = (int64_t)z * B + b;
  const int L = len[b];
  const T hprev = hs[zb * H + j];
  hp[(zb * Tn + s) * H + j] = hprev;
  const int64_t ga = (zb * Tn + s) * 4 * H + j;
  if (s >= L) {                       // past this sample's length: state frozen, nothing emitted, zero gate record
    act[ga] = (T)0.f; act[ga + H] = (T)0.f; act[ga + 2 * H] = (T)0.f; act[ga + 3 * H] = (T)0.f;
    cs[(zb * Tn + s) * H + j] = c[zb * H + j];
    return;
  }
  const float* g = gh + zb * 4 * H + j;
  const float ig = sigm((float)gx[ga] + g[0]);
  const float fg = sigm((float)gx[ga + H] + g[H]);
  const float gg = tanh_f((float)gx[ga + 2 * H] + g[2 * H]);
  const float og = sigm((float)gx[ga + 3 * H] + g[3 * H]);
  const float cn = fg * c[zb * H + j] + ig * gg;
  const float hn = og * tanh_f(cn);
  c[zb * H + j] = cn;
  cs[(zb * Tn + s) * H + j] = cn;
  act[ga] = (T)ig; act[ga + H] = (T)fg; act[ga + 2 * H] = (T)gg; act[ga + 3 * H] = (T)og;
  hs[zb * H + j] = (T)hn;
  const int t = z == 0 ? s : L - 1 - s;
  y[((int64_t)b * Tn + t) * 2 * H + z * H + j] = (T)hn;
}

// Backward of step s.  dh (f32 [2][B][H]) holds dgates_{s+1} . W_hh (the host's product; zeros at the last step), dc the
// cell-state gradient carried from step s+1 (in/out).  Activity is a prefix in s (s < len_b), so a sample inactive at s
// is inactive at every later step too: its gate gradients are zero, hence its dh, and its dc was never touched -- no
// state has to be carried across masked steps.  The host then forms dh = dg_s . W_hh for step s-1.
template <typename T>
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ act,
                                                            const float* __restrict__ cs, const int* __restrict__ len,
                                                            const float* __restrict__ dh,
                                                            float* __restrict__ dc, T* __restrict__ dg, int B, int Tn, int H,
                                                            int s) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)2 * B * H) return;
  const int j = (int)(idx % H), b = (int)((idx / H) % B), z = (int)(idx / ((int64_t)H * B));
  const int64_t zb = (int64_t)z * B + b;
  const int L = len[b];
  const int64_t ga = (zb * Tn + s) * 4 * H + j;
  if (s >= L) {       // inactive: no gradient enters or leaves here
    dg[ga] = (T)0.f; dg[ga + H] = (T)0.f; dg[ga + 2 * H] = (T)0.f; dg[ga + 3 * H] = (T)0.f;
    return;
  }
  const int t = z == 0 ? s : L - 1 - s;
  const float dht = (float)dy[((int64_t)b * Tn + t) * 2 * H + z * H + j] + dh[zb * H + j];
  const float ig = (float)act[ga], fg = (float)act[ga + H], gg = (float)act[ga + 2 * H], og = (float)act[ga + 3 * H];
  const float cn = cs[(zb * Tn + s) * H + j];
  const float cp = s > 0 ? cs[(zb * Tn + s - 1) * H + j] : 0.f;
  const float th = tanh_f(cn);
  const float dct = dc[zb * H + j] + dht * og * (1.f - th * th);
  dc[zb * H + j] = dct * fg;
  dg[ga] = (T)(dct * gg * ig * (1.f - ig));
  dg[ga + H] = (T)(dct * cp * fg * (1.f - fg));
  dg[ga + 2 * H] = (T)(dct * ig * (1.f - gg * gg));
  dg[ga + 3 * H] = (T)(dht * th * og * (1.f - og));
}

// dst[b][s][:] = s < len_b ? src[b][len_b - 1 - s][:] : 0   (its own inverse on the valid part: also maps gradients back)
template <typename T>
__global__ __launch_bounds__(256) void lstm_reverse_rows_kernel(const T* __restrict__ src, const int* __restrict__ len,
                                                                T* __restrict__ dst, int B, int Tn, int D8) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;        // (b, s, 16-byte chunk)
  if (idx >= (int64_t)B * Tn * D8) return;
  const int c = (int)(idx % D8), s = (int)((idx / D8) % Tn), b = (int)(idx / ((int64_t)D8 * Tn));
  const int L = len[b];
  u4 v = {0u, 0u, 0u, 0u};
  if (s < L) v = reinterpret_cast<const u4*>(src)[((int64_t)b * Tn + (L - 1 - s)) * D8 + c];
  reinterpret_cast<u4*>(dst)[idx] = v;
}

}  // namespace

extern "C" int vmr_lstm_cell_fwd(const void* gx, const void* gh, const int* len, void* c, void* hs, void* act, void* cs,
                                 void* hp, void* y, int B, int T, int H, int s, int dtype, void* stream) {
  VMR_CHECK(gx && gh && len && c && hs && act && cs && hp && y, "vmr_lstm_cell_fwd: null pointer");
  VMR_CHECK(B > 0 && T > 0 && H > 0 && s >= 0 && s < T, "vmr_lstm_cell_fwd: bad shape B=%d T=%d H=%d s=%d", B, T, H, s);
  VMR_CHECK(dtype == VMR_BF16 || dtype == VMR_F32, "vmr_lstm_cell_fwd: bad dtype %d", dtype);
  const int nblk = (int)(((int64_t)2 * B * H + 255) / 256);
  if (dtype == VMR_BF16)
    hipLaunchKernelGGL(lstm_cell_fwd_kernel<bf16_t>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gx,
                       (const float*)gh, len, (float*)c, (bf16_t*)hs, (bf16_t*)act, (float*)cs, (bf16_t*)hp, (bf16_t*)y, B, T, H, s);
  else
    hipLaunchKernelGGL(lstm_cell_fwd_kernel<float>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const float*)gx,
                       (const float*)gh, len, (float*)c, (float*)hs, (float*)act, (float*)cs, (float*)hp, (float*)y, B, T, H, s);
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_lstm_cell_bwd(const void* dy, const void* act, const void* cs, const int* len, const void* dh,
                                 void* dc, void* dg, int B, int T, int H, int s, int dtype, void* stream) {
  VMR_CHECK(dy && act && cs && len && dh && dc && dg, "vmr_lstm_cell_bwd: null pointer");
  VMR_CHECK(B > 0 && T > 0 && H > 0 && s >= 0 && s < T, "vmr_lstm_cell_bwd: bad shape B=%d T=%d H=%d s=%d", B, T, H, s);
  VMR_CHECK(dtype == VMR_BF16 || dtype == VMR_F32, "vmr_lstm_cell_bwd: bad dtype %d", dtype);
  const int nblk = (int)(((int64_t)2 * B * H + 255) / 256);
  if (dtype == VMR_BF16)
    hipLaunchKernelGGL(lstm_cell_bwd_kernel<bf16_t>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)act, (const float*)cs, len, (const float*)dh, (float*)dc, (bf16_t*)dg, B, T, H, s);
  else
    hipLaunchKernelGGL(lstm_cell_bwd_kernel<float>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const float*)dy,
                       (const float*)act, (const float*)cs, len, (const float*)dh, (float*)dc, (float*)dg, B, T, H, s);
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_lstm_reverse_rows(const void* src, const int* len, void* dst, int B, int T, int D, int dtype, void* stream) {
  VMR_CHECK(src && len && dst, "vmr_lstm_reverse_rows: null pointer");
  VMR_CHECK(dtype == VMR_BF16 || dtype == VMR_F32, "vmr_lstm_reverse_rows: bad dtype %d", dtype);
  const int per16 = dtype == VMR_BF16 ? 8 : 4;
  VMR_CHECK(B > 0 && T > 0 && D > 0 && D % per16 == 0, "vmr_lstm_reverse_rows: D=%d must be a multiple of %d", D, per16);
  VMR_CHECK((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "vmr_lstm_reverse_rows: 16-byte alignment");
  const int D8 = D / per16;
  const int nblk = (int)(((int64_t)B * T * D8 + 255) / 256);
  hipLaunchKernelGGL(lstm_reverse_rows_kernel<float>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const float*)src, len,
                     (float*)dst, B, T, D8);
  VMR_LAUNCH_CHECK();
  return 0;
}
