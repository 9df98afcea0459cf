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
                                                            T* __restrict__ hp, T* __restrict__ y, int B, int Tn, int H, int s, int ND) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;        // (z, b, j)
  if (idx >= (int64_t)ND * B * H) return;
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
  const int t = (z & 1) == 0 ? s : L - 1 - s;
  y[(((int64_t)(z >> 1) * B + b) * Tn + t) * 2 * H + (z & 1) * H + j] = (T)hn;
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
                                                            int s, int ND) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)ND * B * H) return;
  const int j = (int)(idx % H), b = (int)((idx / H) % B), z = (int)(idx / ((int64_t)H * B));
  const int64_t zb = (int64_t)z * B + b;
  const int L = len[b];
  const int64_t ga = (zb * Tn + s) * 4 * H + j;
  if (s >= L) {       // inactive: no gradient enters or leaves here
    dg[ga] = (T)0.f; dg[ga + H] = (T)0.f; dg[ga + 2 * H] = (T)0.f; dg[ga + 3 * H] = (T)0.f;
    return;
  }
  const int t = (z & 1) == 0 ? s : L - 1 - s;
  const float dht = (float)dy[(((int64_t)(z >> 1) * B + b) * Tn + t) * 2 * H + (z & 1) * H + j] + dh[zb * H + j];
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

// ---------------------------------------------------------------------------------------------------------------------
// Fused step (bf16, H = 256 or 512): the recurrent product AND the pointwise half in one launch -- a step is one graph node
// instead of two, and the product stops going through the generic register-staged GEMM (64 batch rows fill no 128-row tile).
// Both operands go global -> fragment registers directly (16-byte loads in MFMA layout; W_hh is L2-resident, h is 32-64 KB).
// What shapes the tiling: ONE WAVE sustains only ~6.7 GB/s of loads, so a step costs (bytes per wave) x 150 ns
// per KiB.  The first version (64 batch x 16 units x 4 gates per workgroup, each wave re-loading the whole 32-KiB weight tile:
// 40-64 KiB per wave) took 8.7 / 11.0 us per step; these tiles load 16 KiB per wave and spread over 128-256 workgroups:
//   forward:  workgroup = (z, 4 units, 64 batch rows), wave = 16 batch rows; the 16 weight rows of the tile are the FOUR
//             gates of the four units (MFMA column n: gate n >> 2, unit n & 3), so one accumulator holds i, f, g, o of a
//             (batch row, unit) in four lanes of the same 16-lane group -- regrouped through 4 KiB of LDS to one
//             (batch row, unit) per thread;
//   backward: workgroup = (z, 16 units, 16 batch rows), wave w = the K range of gate w (K = 4H split four ways), partial
//             dh tiles summed through 4 KiB of LDS; then one (batch row, unit) per thread.
// h ping-pongs between two [2][B][H] buffers (a workgroup reads ALL of h_{s-1} while others write h_s).
// ---------------------------------------------------------------------------------------------------------------------
template <int KS, typename E>
__device__ __forceinline__ void ld_frags(const E* __restrict__ p, bf16x8 (&f)[KS]) {   // KS k-steps of 32, lane offset applied
#pragma unroll
  for (int k = 0; k < KS; ++k) f[k] = *reinterpret_cast<const bf16x8*>(p + k * 32);
}

template <typename E, int HH>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(const E* __restrict__ gx, const E* __restrict__ hprev,
                                                            const E* __restrict__ whh, const int* __restrict__ len,
                                                            float* __restrict__ c, E* __restrict__ hnext,
                                                            E* __restrict__ act, float* __restrict__ cs,
                                                            E* __restrict__ hp, E* __restrict__ y, int B, int Tn, int s) {
  constexpr int H = HH, KS = HH / 32;
  __shared__ float tile[64][17];                     // [batch row of the workgroup][gate * 4 + unit]
  __shared__ bf16x8 wfrag[KS][64];                   // the tile's weight fragments, shared by the four waves (8-16 KiB)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int z = blockIdx.y, u0 = blockIdx.x * 4, b0 = blockIdx.z * 64;
  const int r16 = lane & 15, kq = lane >> 4;
  // this thread's (batch row, unit) of the 64 x 4 tile and its pointwise operands, requested FIRST and unconditionally
  // (row clamped).  One pair per thread keeps the memory instructions per wave low: a wave issues roughly one every
  // 150 ns whatever its width, and the lane-per-MFMA-column form (4 rows x 9 narrow stores per lane) spent 5 us on them.
  const int m = threadIdx.x >> 2, un = threadIdx.x & 3, j = u0 + un;
  const int b = b0 + m, bc = min(b, B - 1);
  const int64_t zb = (int64_t)z * B + bc;
  const int64_t ga = (zb * Tn + s) * 4 * H + j;
  const int L = len[bc];
  float gxv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) gxv[g] = (float)gx[ga + g * H];
  const float cprev = c[zb * H + j];
  const E hpv = hprev[zb * H + j];              // (both ping-pong buffers start zeroed: valid at s = 0 too)
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (s > 0) {                                       // (h_{-1} = 0: step 0 has no recurrent part; s is uniform: no divergent barrier)
    // the four waves multiply their own 16 batch rows by the SAME 16 weight rows: each wave fetches a quarter of the
    // weight fragments into LDS (KS/4 loads instead of KS: a wave's loads are what a step costs), all read all of them
    bf16x8 a[KS];
    ld_frags<KS, E>(hprev + ((int64_t)z * B + min(b0 + w * 16 + r16, B - 1)) * H + kq * 8, a);
    const E* wp = whh + ((int64_t)z * 4 * H + (r16 >> 2) * H + u0 + (r16 & 3)) * H + kq * 8;
#pragma unroll
    for (int k = 0; k < KS / 4; ++k) wfrag[w * (KS / 4) + k][lane] = *reinterpret_cast<const bf16x8*>(wp + (w * (KS / 4) + k) * 32);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KS; ++k) acc = mfma16<E>(a[k], wfrag[k][lane], acc);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[w * 16 + kq * 4 + r][r16] = acc[r];   // D[batch kq*4 + r][n = gate * 4 + unit]
  __syncthreads();
  if (b >= B) return;
  hp[(zb * Tn + s) * H + j] = hpv;
  if (s >= L) {
    act[ga] = (E)0.f; act[ga + H] = (E)0.f; act[ga + 2 * H] = (E)0.f; act[ga + 3 * H] = (E)0.f;
    cs[(zb * Tn + s) * H + j] = cprev;
    hnext[zb * H + j] = hpv;
    return;
  }
  const float ig = sigm(gxv[0] + tile[m][un]);
  const float fg = sigm(gxv[1] + tile[m][4 + un]);
  const float gg = tanh_f(gxv[2] + tile[m][8 + un]);
  const float og = sigm(gxv[3] + tile[m][12 + un]);
  const float cn = fg * cprev + ig * gg;
  const float hn = og * tanh_f(cn);
  c[zb * H + j] = cn;
  cs[(zb * Tn + s) * H + j] = cn;
  act[ga] = (E)ig; act[ga + H] = (E)fg; act[ga + 2 * H] = (E)gg; act[ga + 3 * H] = (E)og;
  hnext[zb * H + j] = (E)hn;
  const int t = (z & 1) == 0 ? s : L - 1 - s;
  y[(((int64_t)(z >> 1) * B + b) * Tn + t) * 2 * H + (z & 1) * H + j] = (E)hn;
}

template <typename E, int HH>
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(const E* __restrict__ dy, const E* __restrict__ act,
                                                            const float* __restrict__ cs, const int* __restrict__ len,
                                                            const E* __restrict__ whht, float* __restrict__ dc,
                                                            E* __restrict__ dg, int B, int Tn, int s) {
  constexpr int H = HH, KS = HH / 32;
  __shared__ float part[4][16][17];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int z = blockIdx.y, j0 = blockIdx.x * 16, b0 = blockIdx.z * 16;
  const int r16 = lane & 15, kq = lane >> 4;
  // this thread's (batch row, unit) of the tile, its pointwise operands first (unconditional, row clamped)
  const int m = threadIdx.x >> 4, j = j0 + (threadIdx.x & 15);
  const int b = b0 + m, bc = min(b, B - 1);
  const int64_t zb = (int64_t)z * B + bc;
  const int64_t ga = (zb * Tn + s) * 4 * H + j;
  const int L = len[bc];
  float av[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) av[g] = (float)act[ga + g * H];
  const float cn = cs[(zb * Tn + s) * H + j];
  const float cpv = cs[(zb * Tn + max(s - 1, 0)) * H + j];
  const float dcv = dc[zb * H + j];
  const int t = min(max((z & 1) == 0 ? s : L - 1 - s, 0), Tn - 1);
  const float dyv = (float)dy[(((int64_t)(z >> 1) * B + bc) * Tn + t) * 2 * H + (z & 1) * H + j];
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (s + 1 < Tn) {                                  // dh = dg_{s+1} . W_hh (nothing flows into the last step); wave w: gate w's K range
    bf16x8 a[KS], bw[KS];
    ld_frags<KS, E>(dg + (((int64_t)z * B + min(b0 + r16, B - 1)) * Tn + s + 1) * 4 * H + w * H + kq * 8, a);
    ld_frags<KS, E>(whht + ((int64_t)z * H + j0 + r16) * 4 * H + w * H + kq * 8, bw);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < KS; ++k) acc = mfma16<E>(a[k], bw[k], acc);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) part[w][kq * 4 + r][r16] = acc[r];      // D[batch kq*4 + r][unit r16]
  __syncthreads();
  if (b >= B) return;
  if (s >= L) {
    dg[ga] = (E)0.f; dg[ga + H] = (E)0.f; dg[ga + 2 * H] = (E)0.f; dg[ga + 3 * H] = (E)0.f;
    return;
  }
  const int n = threadIdx.x & 15;
  const float dht = dyv + part[0][m][n] + part[1][m][n] + part[2][m][n] + part[3][m][n];
  const float ig = av[0], fg = av[1], gg = av[2], og = av[3];
  const float cp = s > 0 ? cpv : 0.f;
  const float th = tanh_f(cn);
  const float dct = dcv + dht * og * (1.f - th * th);
  dc[zb * H + j] = dct * fg;
  dg[ga] = (E)(dct * gg * ig * (1.f - ig));
  dg[ga + H] = (E)(dct * cp * fg * (1.f - fg));
  dg[ga + 2 * H] = (E)(dct * ig * (1.f - gg * gg));
  dg[ga + 3 * H] = (E)(dht * th * og * (1.f - og));
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
                                 void* hp, void* y, int B, int T_, int H, int s, int ndir, int dtype, void* stream) {
  VMR_CHECK(ndir >= 2 && ndir % 2 == 0, "vmr_lstm: ndir must be 2 x the number of independent LSTMs");
  VMR_CHECK(gx && gh && len && c && hs && act && cs && hp && y, "vmr_lstm_cell_fwd: null pointer");
  VMR_CHECK(B > 0 && T_ > 0 && H > 0 && s >= 0 && s < T_, "vmr_lstm_cell_fwd: bad shape B=%d T=%d H=%d s=%d", B, T_, H, s);
  VMR_CHECK(vmr_dtype_ok(dtype), "vmr_lstm_cell_fwd: bad dtype %d", dtype);
  const int nblk = (int)(((int64_t)ndir * B * H + 255) / 256);
  VMR_DISPATCH(dtype, T,
               hipLaunchKernelGGL(lstm_cell_fwd_kernel<T>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const T*)gx, (const float*)gh,
                                  len, (float*)c, (T*)hs, (T*)act, (float*)cs, (T*)hp, (T*)y, B, T_, H, s, ndir));
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_lstm_cell_bwd(const void* dy, const void* act, const void* cs, const int* len, const void* dh,
                                 void* dc, void* dg, int B, int T_, int H, int s, int ndir, int dtype, void* stream) {
  VMR_CHECK(ndir >= 2 && ndir % 2 == 0, "vmr_lstm: ndir must be 2 x the number of independent LSTMs");
  VMR_CHECK(dy && act && cs && len && dh && dc && dg, "vmr_lstm_cell_bwd: null pointer");
  VMR_CHECK(B > 0 && T_ > 0 && H > 0 && s >= 0 && s < T_, "vmr_lstm_cell_bwd: bad shape B=%d T=%d H=%d s=%d", B, T_, H, s);
  VMR_CHECK(vmr_dtype_ok(dtype), "vmr_lstm_cell_bwd: bad dtype %d", dtype);
  const int nblk = (int)(((int64_t)ndir * B * H + 255) / 256);
  VMR_DISPATCH(dtype, T,
               hipLaunchKernelGGL(lstm_cell_bwd_kernel<T>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (const T*)act,
                                  (const float*)cs, len, (const float*)dh, (float*)dc, (T*)dg, B, T_, H, s, ndir));
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_lstm_step_supported(int H, int dtype) { return vmr_dtype_16(dtype) && (H == 256 || H == 512); }

extern "C" int vmr_lstm_step_fwd(const void* gx, const void* hprev, const void* whh, const int* len, void* c, void* hnext,
                                 void* act, void* cs, void* hp, void* y, int B, int T, int H, int s, int ndir, int dtype, void* stream) {
  VMR_CHECK(ndir >= 2 && ndir % 2 == 0, "vmr_lstm: ndir must be 2 x the number of independent LSTMs");
  VMR_CHECK(gx && hprev && whh && len && c && hnext && act && cs && hp && y, "vmr_lstm_step_fwd: null pointer");
  VMR_CHECK(hprev != hnext, "vmr_lstm_step_fwd: h must ping-pong between two buffers");
  VMR_CHECK(vmr_lstm_step_supported(H, dtype), "vmr_lstm_step_fwd: unsupported H=%d dtype=%d", H, dtype);
  VMR_CHECK(B > 0 && T > 0 && s >= 0 && s < T, "vmr_lstm_step_fwd: bad shape B=%d T=%d s=%d", B, T, s);
  VMR_CHECK((((uintptr_t)hprev | (uintptr_t)whh) & 15) == 0, "vmr_lstm_step_fwd: 16-byte alignment");
  const dim3 grid(H / 4, ndir, (B + 63) / 64);
#define VMR_LSTM_FWD(HH)                                                                                               \
  VMR_DISPATCH16(dtype, E,                                                                                             \
                 hipLaunchKernelGGL((lstm_step_fwd_kernel<E, HH>), grid, dim3(256), 0, (hipStream_t)stream, (const E*)gx, \
                                    (const E*)hprev, (const E*)whh, len, (float*)c, (E*)hnext, (E*)act, (float*)cs, (E*)hp, \
                                    (E*)y, B, T, s))
  if (H == 256) VMR_LSTM_FWD(256); else VMR_LSTM_FWD(512);
#undef VMR_LSTM_FWD
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_lstm_step_bwd(const void* dy, const void* act, const void* cs, const int* len, const void* whht, void* dc,
                                 void* dg, int B, int T, int H, int s, int ndir, int dtype, void* stream) {
  VMR_CHECK(ndir >= 2 && ndir % 2 == 0, "vmr_lstm: ndir must be 2 x the number of independent LSTMs");
  VMR_CHECK(dy && act && cs && len && whht && dc && dg, "vmr_lstm_step_bwd: null pointer");
  VMR_CHECK(vmr_lstm_step_supported(H, dtype), "vmr_lstm_step_bwd: unsupported H=%d dtype=%d", H, dtype);
  VMR_CHECK(B > 0 && T > 0 && s >= 0 && s < T, "vmr_lstm_step_bwd: bad shape B=%d T=%d s=%d", B, T, s);
  VMR_CHECK((((uintptr_t)dg | (uintptr_t)whht) & 15) == 0, "vmr_lstm_step_bwd: 16-byte alignment");
  const dim3 grid(H / 16, ndir, (B + 15) / 16);
#define VMR_LSTM_BWD(HH)                                                                                               \
  VMR_DISPATCH16(dtype, E,                                                                                             \
                 hipLaunchKernelGGL((lstm_step_bwd_kernel<E, HH>), grid, dim3(256), 0, (hipStream_t)stream, (const E*)dy, \
                                    (const E*)act, (const float*)cs, len, (const E*)whht, (float*)dc, (E*)dg, B, T, s))
  if (H == 256) VMR_LSTM_BWD(256); else VMR_LSTM_BWD(512);
#undef VMR_LSTM_BWD
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_lstm_reverse_rows(const void* src, const int* len, void* dst, int B, int T, int D, int dtype, void* stream) {
  VMR_CHECK(src && len && dst, "vmr_lstm_reverse_rows: null pointer");
  VMR_CHECK(vmr_dtype_ok(dtype), "vmr_lstm_reverse_rows: bad dtype %d", dtype);
  const int per16 = 16 / vmr_dtype_size(dtype);
  VMR_CHECK(B > 0 && T > 0 && D > 0 && D % per16 == 0, "vmr_lstm_reverse_rows: D=%d must be a multiple of %d", D, per16);
  VMR_CHECK((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "vmr_lstm_reverse_rows: 16-byte alignment");
  const int D8 = D / per16;
  const int nblk = (int)(((int64_t)B * T * D8 + 255) / 256);
  hipLaunchKernelGGL(lstm_reverse_rows_kernel<float>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const float*)src, len,
                     (float*)dst, B, T, D8);
  VMR_LAUNCH_CHECK();
  return 0;
}
