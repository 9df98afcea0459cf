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

// ---------------------------------------------------------------------------------------------------------------------
// The whole recurrence of a layer in ONE launch (bf16 / fp16, H = 256 or 512, B <= 64): persistent workgroups, one
// counter barrier per step instead of one kernel boundary per step.  A graph-captured step kernel costs its 5-9 us PLUS
// a ~1.4 us dependent-kernel boundary, 2 x 128 of them per layer; a step here is: poll the direction's arrival counter,
// fetch h_{s-1} (32-64 KB, L2 / fabric), 32-64 MFMAs per wave against weights that never leave LDS, the gate math in
// registers, publish this workgroup's 2 KB of h_s, arrive.
//
//   workgroup = (direction z, 16 hidden units x all 4 gates x up to 64 batch rows); its 64 rows of W_hh (32-64 KB) are
//               loaded into LDS ONCE; the cell state c and this workgroup's slice of h stay in REGISTERS over all steps;
//   exchange  = a history buffer hist[z][s] of h AFTER step s, slice-major ([H/16][64][16], so a workgroup publishes one
//               contiguous 2 KB block and an MFMA A-fragment of any batch row is one aligned 16-byte load).  Every step
//               writes FRESH addresses.  The protocol is placement-independent (MI355X_MICROARCH "Workgroup dispatch"):
//               payload by write-through `sc1` stores, every storing wave's s_waitcnt vmcnt(0), the workgroup barrier,
//               ONE agent-scope relaxed atomic add on the direction's counter; readers poll that counter with agent-
//               scope (`sc1`) loads + s_sleep and fetch the payload with `sc1` loads (L1-bypassing).  No fence, no
//               __threadfence (an L2 write-back / invalidate per step costs more than the step).
//   placement = none assumed; a direction's workgroups are dealt over its share of the eight XCD groups (blockIdx % 8);
//   liveness  = all 8 x H/16 <= 256 workgroups are resident at once on an otherwise idle GPU (one per CU); every poll
//               is bounded: after ~0.25 s a workgroup raises sync[8] and stops waiting (the host checks the flag).
// Same arithmetic as lstm_step_fwd_kernel / lstm_step_bwd_kernel (same MFMA products in the same k order; the gate math
// may contract into different FMAs: results agree to an ulp of the storage type).
// ---------------------------------------------------------------------------------------------------------------------
struct SeqArgs {
  const void *gx, *whh;
  const int* len;
  void* act; float* cs; void *hp, *y;      // forward outputs (see the layouts at the top)
  const void *dy; float* dcs_unused; void* dg;   // backward
  void* hist;                               // [Z][T][H/16][64][16] (fwd: h after step s; bwd: unused)
  int* sync;                                // [0..7] arrival counters per direction, [8] error flag; zeroed by the caller
  int B, T, ndir, xpd;
  int mode;                                 // bit 0 (forward): the history carries its own readiness (sentinel exchange), no counter
};

constexpr uint32_t SEQ_NOT_YET = 0x7FFFu;   // a NaN no arithmetic produces (bf16 and fp16 alike): "this element has not been written"

__device__ __forceinline__ bf16x8 ld16_sc1(const void* p) {
  bf16x8 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st8_sc1(void* p, u32x2 v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
// wave-uniform wait of thread 0 for `*ctr >= target`; bounded (a lost workgroup must not hang the GPU)
__device__ __forceinline__ void seq_wait(int* ctr, int target, int* err, int* dead) {
  if (threadIdx.x == 0 && !*dead) {
    int it = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++it > (1 << 18)) { *dead = 1; __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void seq_arrive(int* ctr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's payload stores have been written through
  __syncthreads();                                      // ... and every other wave's of the workgroup
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// RB = batch rows per workgroup (64, 32 or 16: RB / 16 waves, 4 * RB threads); a unit slice's row groups are separate
// workgroups, each fetching only its own rows of h (the per-CU fetch is what a step waits for, see the backward kernel)
template <typename E, int HH, int RB>
__global__ __launch_bounds__(4 * RB) void lstm_seq_fwd_kernel(SeqArgs a) {
  constexpr int H = HH, KS = HH / 32, GPB = 64 / RB, WPX = (HH / 16) * GPB, LDW = HH + 8, LDT = 68, NTH = 4 * RB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  E* wS = reinterpret_cast<E*>(smem);                          // [64][LDW]: rows g * 16 + n = gate g of unit u0 + n
  float* tile0 = reinterpret_cast<float*>(wS + 64 * LDW);      // [2][RB][LDT]: pre-activations, [batch row][g * 16 + n], by step parity
  __shared__ int dead;
  // affinity (speed only): blocks with equal blockIdx % 8 share an XCD; direction z takes `xpd` of the 8 groups
  const int grp = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int z = grp % a.ndir, role = grp / a.ndir + a.xpd * kk;
  if (grp / a.ndir >= a.xpd || role >= WPX) return;
  const int slot = role / GPB, b0 = (role % GPB) * RB;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  const int B = a.B, Tn = a.T, u0 = slot * 16;
  const bool sent = (a.mode & 1) != 0;
  const E* gx = reinterpret_cast<const E*>(a.gx);
  const E* whh = reinterpret_cast<const E*>(a.whh);
  E* act = reinterpret_cast<E*>(a.act);
  E* hp = reinterpret_cast<E*>(a.hp);
  E* y = reinterpret_cast<E*>(a.y);
  E* hz = reinterpret_cast<E*>(a.hist) + (int64_t)z * Tn * (H * 64);
  for (int i = tid; i < 64 * (H / 8); i += NTH) {
    const int row = i / (H / 8), ch = i - row * (H / 8);
    *reinterpret_cast<bf16x8*>(wS + row * LDW + ch * 8) =
        *reinterpret_cast<const bf16x8*>(whh + ((int64_t)z * 4 * H + (row >> 4) * H + u0 + (row & 15)) * H + ch * 8);
  }
  if (tid == 0) dead = 0;
  __syncthreads();
  // this thread's cells: batch row m, units j0 .. j0 + 3 (all four gates); rows past B repeat row B - 1 (never stored,
  // except into the history, whose rows other workgroups read as MFMA operand rows: they must be finite)
  const int m = tid >> 2, q = tid & 3, j0 = u0 + 4 * q;      // (m: row within the group, batch row b0 + m)
  const bool real = b0 + m < B;
  const int bc = min(b0 + m, B - 1);
  const int64_t zb = (int64_t)z * B + bc;
  const int L = a.len[bc];
  typedef __attribute__((ext_vector_type(4))) E E4;
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  E4 hprev = {(E)0.f, (E)0.f, (E)0.f, (E)0.f};
  E4 gxv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) gxv[g] = *reinterpret_cast<const E4*>(gx + (zb * Tn + 0) * 4 * H + g * H + j0);
  for (int s = 0; s < Tn; ++s) {
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float* tile = tile0 + (s & 1) * (RB * LDT);
    if (s > 0) {          // (h_{-1} = 0: step 0 has no recurrent part; s is uniform)
      if (!sent) seq_wait(a.sync + z, s * WPX, a.sync + 8, &dead);
      const E* hsrc = hz + (int64_t)(s - 1) * (H * 64);
      bf16x8 af[KS];
      for (int tries = 0;; ++tries) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const int kk = k * 32 + kq * 8;
          af[k] = ld16_sc1(hsrc + ((kk >> 4) * 64 + b0 + w * 16 + r16) * 16 + (kk & 15));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < KS; ++k) asm volatile("" : "+v"(af[k]));
        if (!sent) break;
        // Sentinel exchange: the history was filled with SEQ_NOT_YET before the launch, a producer's 8-byte store (four
        // units of one row, first element never SEQ_NOT_YET) replaces it atomically: a fragment is complete when the first
        // element of both its 8-byte halves is not the sentinel.  No counter, no store acknowledgement on the producer's
        // critical path; the poll IS the operand fetch.  Bounded like seq_wait.
        uint32_t bad = 0u;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const u32x4 raw = __builtin_bit_cast(u32x4, af[k]);
          bad |= (uint32_t)((raw[0] & 0xFFFFu) == SEQ_NOT_YET) | (uint32_t)((raw[2] & 0xFFFFu) == SEQ_NOT_YET);
        }
        if (!__builtin_amdgcn_ballot_w64(bad != 0u)) break;
        if (*reinterpret_cast<volatile int*>(&dead)) break;
        if (tries > (1 << 16)) {
          if (lane == 0) { dead = 1; __hip_atomic_store(a.sync + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < KS; ++k)
          acc[g] = mfma16<E>(af[k], *reinterpret_cast<const bf16x8*>(wS + (g * 16 + r16) * LDW + k * 32 + kq * 8), acc[g]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[(w * 16 + kq * 4 + r) * LDT + g * 16 + r16] = acc[g][r];   // D[batch][gate g, unit r16]
    __syncthreads();
    f32x4 pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) pre[g] = *reinterpret_cast<const f32x4*>(tile + m * LDT + g * 16 + 4 * q);
    const int64_t ga = (zb * Tn + s) * 4 * H + j0;
    const int64_t ca = (zb * Tn + s) * H + j0;
    E4 av[4], hn;
    f32x4 cn4;
    if (s >= L) {          // past this sample's length: state frozen, nothing emitted, zero gate record
#pragma unroll
      for (int g = 0; g < 4; ++g) av[g] = (E4){(E)0.f, (E)0.f, (E)0.f, (E)0.f};
      hn = hprev;
#pragma unroll
      for (int e = 0; e < 4; ++e) cn4[e] = c[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ig = sigm((float)gxv[0][e] + pre[0][e]);
        const float fg = sigm((float)gxv[1][e] + pre[1][e]);
        const float gg = tanh_f((float)gxv[2][e] + pre[2][e]);
        const float og = sigm((float)gxv[3][e] + pre[3][e]);
        const float cn = fg * c[e] + ig * gg;
        c[e] = cn; cn4[e] = cn;
        hn[e] = (E)(og * tanh_f(cn));
        av[0][e] = (E)ig; av[1][e] = (E)fg; av[2][e] = (E)gg; av[3][e] = (E)og;
      }
    }
    // the exchange first: publish h_s and arrive, THEN the records nobody waits for (seq_arrive's vmcnt(0) would
    // otherwise sit out the acknowledgements of seven plain stores every step); they and the next step's gx loads
    // drain under the next poll
    {
      u32x2 hb = __builtin_bit_cast(u32x2, hn);
      if (sent && (hb[0] & 0xFFFFu) == SEQ_NOT_YET) hb[0] ^= 1u;       // (a NaN with exactly the sentinel's payload: another NaN)
      st8_sc1(hz + (int64_t)s * (H * 64) + (slot * 64 + b0 + m) * 16 + 4 * q, hb);
    }
    if (s + 1 < Tn && !sent) seq_arrive(a.sync + z);
    if (real) {
      *reinterpret_cast<E4*>(hp + ca) = hprev;
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<E4*>(act + ga + g * H) = av[g];
      *reinterpret_cast<f32x4*>(a.cs + ca) = cn4;
      if (s < L) {
        const int t = (z & 1) == 0 ? s : L - 1 - s;
        *reinterpret_cast<E4*>(y + (((int64_t)(z >> 1) * B + b0 + m) * Tn + t) * 2 * H + (z & 1) * H + j0) = hn;
      }
    }
    hprev = hn;
    if (s + 1 < Tn) {
#pragma unroll
      for (int g = 0; g < 4; ++g) gxv[g] = *reinterpret_cast<const E4*>(gx + (zb * Tn + s + 1) * 4 * H + g * H + j0);
    }
  }
}

// Backward of the whole recurrence in one launch.  Step s needs dh = dg_{s+1} . W_hh with K = 4H: EVERY workgroup's gate
// gradients of step s + 1 -- the gate-gradient tensor dg [Z][B][T][4H] is itself the exchange buffer (each (b, s) row is
// written once, with write-through stores, and read as MFMA operand rows).  What a step costs is that read: a workgroup
// needs the full 4H-wide rows of its batch rows.  First form: (16 units x 64 batch rows) per workgroup = 128-256 KB per
// workgroup and step through a handful of CUs -- 5.5 / 12 us per step at H = 256 / 512, slower than the per-step
// launches.  Now workgroup = (direction z, 16 units, 16 batch rows): 32-64 KB per step, four times the workgroups (all
// 256 CUs at H = 512), spread over the XCDs (the protocol does not care where they run).  512 threads: wave w takes the
// k-steps w, w + 8, ... of K = 4H (4-8 sixteen-byte operand loads per lane), the eight partial 16 x 16 tiles are summed
// through LDS; threads 0..63 then own (batch row, 4 units) cells; dc stays in registers over all steps.
// Tile (UW units x RB batch rows) per workgroup.  What a step moves over the fabric in total is
// (H / UW) slices x B rows x 4H x 2 B per direction -- every slice re-reads every row -- so wider slices are fewer bytes:
// (16, 16) = 16.8 MB per step at H = 512 and B = 64, (32, 8) = 8.4 MB with the same 256 workgroups (the MFMA tile is then
// half empty in M: rows >= RB load nothing) and 32 rows of W_hh^T (132 KB at H = 512) in LDS.
template <typename E, int HH, int UW, int RB>
__global__ __launch_bounds__(512) void lstm_seq_bwd_kernel(SeqArgs a) {
  constexpr int H = HH, KT = 4 * HH / 32, KW = KT / 8, NT = UW / 16, GPB = 64 / RB, WPX = (HH / UW) * GPB;
  constexpr int LDW = 4 * HH + 8, LDP = UW + 1, QW = UW / 4;
  static_assert(UW % 16 == 0 && (RB == 8 || RB == 16) && RB * QW <= 512, "tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  E* wT = reinterpret_cast<E*>(smem);                          // [UW][LDW]: row n = W_hh^T row of unit u0 + n (K = 4H)
  float* part = reinterpret_cast<float*>(wT + UW * LDW);       // [8 waves][RB][LDP]: the tile's real batch rows only
  __shared__ int dead;
  // affinity (speed only): blocks with equal blockIdx % 8 share an XCD; direction z takes `xpd` of the 8 groups
  const int grp = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int z = grp % a.ndir, role = grp / a.ndir + a.xpd * kk;
  if (grp / a.ndir >= a.xpd || role >= WPX) return;
  const int slot = role / GPB, bq = role % GPB;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  const int B = a.B, Tn = a.T, u0 = slot * UW, b0 = bq * RB;
  const bool sent = RB == 8 && (a.mode & 1) != 0;      // (the 16-row tile keeps the counters)
  const E* whht = reinterpret_cast<const E*>(a.whh);           // [Z][H][4H]
  const E* act = reinterpret_cast<const E*>(a.act);
  const E* dy = reinterpret_cast<const E*>(a.dy);
  E* dg = reinterpret_cast<E*>(a.dg);
  for (int i = tid; i < UW * (4 * H / 8); i += 512) {
    const int row = i / (4 * H / 8), ch = i - row * (4 * H / 8);
    *reinterpret_cast<bf16x8*>(wT + row * LDW + ch * 8) =
        *reinterpret_cast<const bf16x8*>(whht + ((int64_t)z * H + u0 + row) * 4 * H + ch * 8);
  }
  if (tid == 0) dead = 0;
  __syncthreads();
  const bool cell = tid < RB * QW;
  const int m = min(tid / QW, RB - 1), q = tid % QW, j0 = u0 + 4 * q;      // cell: batch row b0 + m, units j0 .. j0 + 3
  const bool real = cell && b0 + m < B;
  const int bc = min(b0 + m, B - 1);
  const int64_t zb = (int64_t)z * B + bc;
  const int L = a.len[bc];
  const int arow = min(b0 + (r16 & (RB - 1)), B - 1);          // this lane's MFMA operand row (batch)
  typedef __attribute__((ext_vector_type(4))) E E4;
  float dcv[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s = Tn - 1; s >= 0; --s) {
    // this step's pointwise operands first (independent of the recurrence: they fly under the wait)
    const int64_t ga = (zb * Tn + s) * 4 * H + j0;
    const int64_t ca = (zb * Tn + s) * H + j0;
    E4 av[4], dyv;
    f32x4 cn4, cp4;
    if (cell) {
#pragma unroll
      for (int g = 0; g < 4; ++g) av[g] = *reinterpret_cast<const E4*>(act + ga + g * H);
      cn4 = *reinterpret_cast<const f32x4*>(a.cs + ca);
      cp4 = *reinterpret_cast<const f32x4*>(a.cs + (zb * Tn + max(s - 1, 0)) * H + j0);
      const int t = min(max((z & 1) == 0 ? s : L - 1 - s, 0), Tn - 1);
      dyv = *reinterpret_cast<const E4*>(dy + (((int64_t)(z >> 1) * B + bc) * Tn + t) * 2 * H + (z & 1) * H + j0);
    }
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (s + 1 < Tn) {        // dh = dg_{s+1} . W_hh (nothing flows into the last step)
      if (!sent) seq_wait(a.sync + z, (Tn - 1 - s) * WPX, a.sync + 8, &dead);
      const E* src = dg + (((int64_t)z * B + arow) * Tn + s + 1) * 4 * H + kq * 8;
      if constexpr (RB == 8) {
        // Only 8 of the MFMA tile's 16 operand rows are real: lanes r16 >= 8 fetch the NEXT k-step of row r16 - 8, so one
        // load instruction covers two adjacent k-steps = 128 contiguous bytes of every row (64 with the idle half), and
        // half as many load instructions.  Wave w takes the k-step pairs (2w, 2w + 1) + 16j.  The even product uses the
        // fragment as loaded, the odd one after a rotation by 8 lanes within each 16-lane row (DPP row_ror:8); the
        // other half of the tile then holds the wrong k-step's rows -- they only reach D rows 8..15, which nobody reads.
        constexpr int KP = KW / 2;
        bf16x8 af[KP];
        for (int tries = 0;; ++tries) {
#pragma unroll
          for (int j = 0; j < KP; ++j) af[j] = ld16_sc1(src + (2 * w + (r16 >> 3) + 16 * j) * 32);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int j = 0; j < KP; ++j) asm volatile("" : "+v"(af[j]));
          if (!sent) break;
          // sentinel exchange (see the forward kernel): dg itself was filled with SEQ_NOT_YET before the launch; a cell's
          // 8-byte store (four units of one gate) replaces it atomically, its first element never the sentinel
          uint32_t bad = 0u;
#pragma unroll
          for (int j = 0; j < KP; ++j) {
            const u32x4 raw = __builtin_bit_cast(u32x4, af[j]);
            bad |= (uint32_t)((raw[0] & 0xFFFFu) == SEQ_NOT_YET) | (uint32_t)((raw[2] & 0xFFFFu) == SEQ_NOT_YET);
          }
          if (!__builtin_amdgcn_ballot_w64(bad != 0u)) break;
          if (*reinterpret_cast<volatile int*>(&dead)) break;
          if (tries > (1 << 16)) {
            if (lane == 0) { dead = 1; __hip_atomic_store(a.sync + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int j = 0; j < KP; ++j) {
          u32x4 raw = __builtin_bit_cast(u32x4, af[j]), rot;
#pragma unroll
          for (int i = 0; i < 4; ++i) rot[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw[i], 0x128, 0xF, 0xF, false);
          const bf16x8 ao = __builtin_bit_cast(bf16x8, rot);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const E* wr = wT + (nt * 16 + r16) * LDW + kq * 8 + (2 * w + 16 * j) * 32;
            acc[nt] = mfma16<E>(af[j], *reinterpret_cast<const bf16x8*>(wr), acc[nt]);
            acc[nt] = mfma16<E>(ao, *reinterpret_cast<const bf16x8*>(wr + 32), acc[nt]);
          }
        }
      } else {
        bf16x8 af[KW];
#pragma unroll
        for (int k = 0; k < KW; ++k) af[k] = ld16_sc1(src + (w + 8 * k) * 32);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < KW; ++k) asm volatile("" : "+v"(af[k]));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const E* wr = wT + (nt * 16 + r16) * LDW + kq * 8;
#pragma unroll
          for (int k = 0; k < KW; ++k) acc[nt] = mfma16<E>(af[k], *reinterpret_cast<const bf16x8*>(wr + (w + 8 * k) * 32), acc[nt]);
        }
      }
    }
    if (kq * 4 < RB) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(w * RB + kq * 4 + r) * LDP + nt * 16 + r16] = acc[nt][r];    // D[batch kq*4 + r][unit]
    }
    __syncthreads();
    if (cell) {
      E4 dgv[4];
      if (s >= L) {
#pragma unroll
        for (int g = 0; g < 4; ++g) dgv[g] = (E4){(E)0.f, (E)0.f, (E)0.f, (E)0.f};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = 4 * q + e;
          float dht = (float)dyv[e];
#pragma unroll
          for (int ww = 0; ww < 8; ++ww) dht += part[(ww * RB + m) * LDP + n];
          const float ig = (float)av[0][e], fg = (float)av[1][e], gg = (float)av[2][e], og = (float)av[3][e];
          const float cp = s > 0 ? cp4[e] : 0.f;
          const float th = tanh_f(cn4[e]);
          const float dct = dcv[e] + dht * og * (1.f - th * th);
          dcv[e] = dct * fg;
          dgv[0][e] = (E)(dct * gg * ig * (1.f - ig));
          dgv[1][e] = (E)(dct * cp * fg * (1.f - fg));
          dgv[2][e] = (E)(dct * ig * (1.f - gg * gg));
          dgv[3][e] = (E)(dht * th * og * (1.f - og));
        }
      }
      if (real) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 v = __builtin_bit_cast(u32x2, dgv[g]);
          if (sent && (v[0] & 0xFFFFu) == SEQ_NOT_YET) v[0] ^= 1u;
          st8_sc1(dg + ga + g * H, v);
        }
      }
    }
    if (s > 0) {
      if (!sent) seq_arrive(a.sync + z);     // (also the barrier that frees `part` for the next step)
      else __syncthreads();                  // `part` is free for the next step
    }
  }
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

extern "C" int vmr_lstm_seq_supported(int B, int H, int ndir, int dtype) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("VMR_LSTM_SEQ");     // A/B switch: 0 = one launch per step
    on = e && atoi(e) == 0 ? 0 : 1;
  }
  // (every workgroup of a launch must be resident at once: forward 8 * H/16, backward ndir * H/16 * 4 of them, one per CU)
  return on && vmr_dtype_16(dtype) && (H == 256 || H == 512) && B >= 1 && B <= 64 && ndir >= 2 && ndir <= 8 && ndir % 2 == 0 &&
         ndir * (H / 16) * 4 <= 256;
}

// 1: vmr_lstm_seq_fwd exchanges h through the history itself -- the caller fills `hist` with the 16-bit pattern 0x7FFF
// ("not yet written") before every launch; 0: arrival counters (VMR_LSTM_SEQ_SENTINEL=0)
extern "C" int vmr_lstm_seq_sentinel(void) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("VMR_LSTM_SEQ_SENTINEL"); on = e ? (atoi(e) != 0) : 1; }
  return on;
}

// 1: vmr_lstm_seq_bwd reads readiness off dg itself: the caller fills dg with the 16-bit pattern 0x7FFF before the launch
// (every row of dg is written by the launch); 0: counters (VMR_LSTM_SEQ_SENTINEL=0, or the (16, 16) tile)
extern "C" int vmr_lstm_seq_bwd_sentinel(void) {
  static int on = -1;
  if (on < 0) {
    const char* t = getenv("VMR_LSTM_SEQ_BWD_TILE");
    on = vmr_lstm_seq_sentinel() && !(t && atoi(t) == 0);
  }
  return on;
}

extern "C" int vmr_lstm_seq_hist_bytes(int T, int H, int ndir, int64_t* bytes) {
  VMR_CHECK(bytes && T > 0 && H > 0 && ndir > 0, "vmr_lstm_seq_hist_bytes: bad argument");
  *bytes = (int64_t)ndir * T * H * 64 * 2;
  return 0;
}

extern "C" int vmr_lstm_seq_fwd(const void* gx, const void* whh, const int* len, void* act, void* cs, void* hp, void* y, void* hist,
                                int* sync, int B, int T, int H, int ndir, int dtype, void* stream) {
  VMR_CHECK(vmr_lstm_seq_supported(B, H, ndir, dtype), "vmr_lstm_seq_fwd: unsupported B=%d H=%d ndir=%d dtype=%d", B, H, ndir, dtype);
  VMR_CHECK(gx && whh && len && act && cs && hp && y && hist && sync, "vmr_lstm_seq_fwd: null pointer");
  VMR_CHECK(T > 0, "vmr_lstm_seq_fwd: T = %d", T);
  VMR_CHECK((((uintptr_t)gx | (uintptr_t)whh | (uintptr_t)act | (uintptr_t)cs | (uintptr_t)hp | (uintptr_t)y | (uintptr_t)hist) & 15) == 0,
            "vmr_lstm_seq_fwd: 16-byte alignment");
  SeqArgs a;
  memset(&a, 0, sizeof(a));
  a.gx = gx; a.whh = whh; a.len = len; a.act = act; a.cs = (float*)cs; a.hp = hp; a.y = y; a.hist = hist; a.sync = sync;
  a.B = B; a.T = T; a.ndir = ndir;
  a.mode = vmr_lstm_seq_sentinel();
  // batch rows per workgroup: 64 (one workgroup per unit slice) / 32 / 16 (VMR_LSTM_SEQ_FWD_ROWS; default below)
  static int rows_env = -1;
  if (rows_env < 0) { const char* e = getenv("VMR_LSTM_SEQ_FWD_ROWS"); rows_env = e ? atoi(e) : 32; }
  int RB = rows_env == 16 ? 16 : (rows_env == 64 ? 64 : 32);
  while (RB < 64 && ndir * (H / 16) * (64 / RB) > 256) RB *= 2;      // every workgroup resident at once
  const size_t lds = (size_t)64 * (H + 8) * 2 + (size_t)2 * RB * 68 * 4;
  const int roles = (H / 16) * (64 / RB);
  // a direction's workgroups share ONE XCD group where they all fit (measured at H = 256, 16 workgroups: 340 us per layer
  // against 388 spread over four; the backward, with four times the workgroups and payload, is the other way round);
  // every workgroup must be resident: at most 32 CUs per XCD x the workgroups of this LDS size a CU holds
  static int xpd_env = -1;
  if (xpd_env < 0) { const char* e = getenv("VMR_LSTM_SEQ_XPD_FWD"); xpd_env = e ? atoi(e) : 0; }
  a.xpd = xpd_env > 0 ? xpd_env : 1;
  const int per_cu = (int)((size_t)160 * 1024 / (lds + 1024)) < 1 ? 1 : (int)((size_t)160 * 1024 / (lds + 1024));
  const int cap = 32 * (per_cu > 2 ? 2 : per_cu);
  if (a.xpd < (roles + cap - 1) / cap) a.xpd = (roles + cap - 1) / cap;
  if (a.xpd * ndir > 8) a.xpd = 8 / ndir;
  const dim3 grid(8 * ((roles + a.xpd - 1) / a.xpd));
#define VMR_LSTM_SEQ_FWD(HH, RBV)                                                                                      \
  VMR_DISPATCH16(dtype, E, do {                                                                                        \
    const void* fn = (const void*)lstm_seq_fwd_kernel<E, HH, RBV>;                                                     \
    if (lds > 64 * 1024) {                                                                                             \
      hipError_t e_ = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
      if (e_ != hipSuccess) return vmr_fail(-5, "vmr_lstm_seq_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e_));   \
    }                                                                                                                  \
    hipLaunchKernelGGL((lstm_seq_fwd_kernel<E, HH, RBV>), grid, dim3(4 * RBV), lds, (hipStream_t)stream, a);           \
  } while (0))
  if (H == 256) { if (RB == 64) VMR_LSTM_SEQ_FWD(256, 64); else if (RB == 32) VMR_LSTM_SEQ_FWD(256, 32); else VMR_LSTM_SEQ_FWD(256, 16); }
  else { if (RB == 64) VMR_LSTM_SEQ_FWD(512, 64); else if (RB == 32) VMR_LSTM_SEQ_FWD(512, 32); else VMR_LSTM_SEQ_FWD(512, 16); }
#undef VMR_LSTM_SEQ_FWD
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_lstm_seq_bwd(const void* dy, const void* act, const void* cs, const int* len, const void* whht, void* dg, int* sync,
                                int B, int T, int H, int ndir, int dtype, void* stream) {
  VMR_CHECK(vmr_lstm_seq_supported(B, H, ndir, dtype), "vmr_lstm_seq_bwd: unsupported B=%d H=%d ndir=%d dtype=%d", B, H, ndir, dtype);
  VMR_CHECK(dy && act && cs && len && whht && dg && sync, "vmr_lstm_seq_bwd: null pointer");
  VMR_CHECK(T > 0, "vmr_lstm_seq_bwd: T = %d", T);
  VMR_CHECK((((uintptr_t)dy | (uintptr_t)act | (uintptr_t)cs | (uintptr_t)whht | (uintptr_t)dg) & 15) == 0, "vmr_lstm_seq_bwd: 16-byte alignment");
  SeqArgs a;
  memset(&a, 0, sizeof(a));
  a.dy = dy; a.act = const_cast<void*>(act); a.cs = (float*)const_cast<void*>(cs); a.len = len; a.whh = whht; a.dg = dg; a.sync = sync;
  a.B = B; a.T = T; a.ndir = ndir;
  a.mode = vmr_lstm_seq_bwd_sentinel();
  // tile per workgroup (see the kernel): the BAN step at B = 64, T = 128 ran 16.57 ms with (16, 16), 15.83 with (32, 8),
  // 15.59 with (64, 8) for the H = 256 layers; VMR_LSTM_SEQ_BWD_TILE = 0 / 1 / 2 for A/B
  static int tile_env = -1;
  if (tile_env < 0) { const char* e = getenv("VMR_LSTM_SEQ_BWD_TILE"); tile_env = e ? atoi(e) : 2; }
  // 0: (16, 16); 1: (32, 8); 2: (64, 8) at H = 256 (its 64 rows of W_hh^T are 132 KB; at H = 512 they do not fit), (32, 8) otherwise
  const int UW = tile_env == 0 ? 16 : (tile_env == 2 && H == 256 ? 64 : 32), RB = tile_env == 0 ? 16 : 8;
  const size_t lds = (size_t)UW * (4 * H + 8) * 2 + (size_t)8 * RB * (UW + 1) * 4;
  // roles per direction = (UW-unit slice, RB-row batch group); a direction's workgroups are spread over its share of
  // the eight XCD groups
  const int roles = (H / UW) * (64 / RB);
  static int xpd_env = -1;
  if (xpd_env < 0) { const char* e = getenv("VMR_LSTM_SEQ_XPD"); xpd_env = e ? atoi(e) : 0; }
  a.xpd = xpd_env > 0 ? xpd_env : 8 / ndir;      // (measured at H = 256: all 64 workgroups on one XCD 652 us per layer, spread over four 415)
  // every workgroup must be RESIDENT (one per CU with these LDS tiles, 32 CUs per XCD): a narrower spread than this leaves
  // workgroups undispatched behind pollers -- every step then runs into the poll bound (measured: 1.4 s per BAN step)
  if (a.xpd < (roles + 31) / 32) a.xpd = (roles + 31) / 32;
  if (a.xpd * ndir > 8) a.xpd = 8 / ndir;
  const dim3 grid(8 * ((roles + a.xpd - 1) / a.xpd));
#define VMR_LSTM_SEQ_BWD(HH, UW_, RB_)                                                                                 \
  VMR_DISPATCH16(dtype, E, do {                                                                                        \
    const void* fn = (const void*)lstm_seq_bwd_kernel<E, HH, UW_, RB_>;                                               \
    if (lds > 64 * 1024) {                                                                                             \
      hipError_t e_ = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
      if (e_ != hipSuccess) return vmr_fail(-5, "vmr_lstm_seq_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e_));   \
    }                                                                                                                  \
    hipLaunchKernelGGL((lstm_seq_bwd_kernel<E, HH, UW_, RB_>), grid, dim3(512), lds, (hipStream_t)stream, a);          \
  } while (0))
  if (UW == 64) {
    VMR_LSTM_SEQ_BWD(256, 64, 8);
  } else if (UW == 32) {
    if (H == 256) VMR_LSTM_SEQ_BWD(256, 32, 8); else VMR_LSTM_SEQ_BWD(512, 32, 8);
  } else {
    if (H == 256) VMR_LSTM_SEQ_BWD(256, 16, 16); else VMR_LSTM_SEQ_BWD(512, 16, 16);
  }
#undef VMR_LSTM_SEQ_BWD
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
