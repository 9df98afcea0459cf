// cqapply.hip -- the apply stage of CQAttention (reference models/layers.py:422-424) and the block's backward, fused:
//     c2q = S_ . Q          mid = S_t^T . C          q2c = S_ . mid   ( == (S_ . S_t^T) . C re-associated )
//     out = [C | c2q | C*c2q | C*q2c]                                  (the input of cqa_linear)
// S_ / S_t are the two masked softmaxes of the trilinear score.
//
// Shapes: one stream is long (video, <= 256 rows), the other short (query, <= 32 rows), both [rows, D].  All three
// contractions have K = 20..128 and N = D: per clip 15-50 MFLOP against 1.3 MB of mandatory traffic (the 4-way concat
// alone is 1 MB) -- HBM-bound, so the kernels are organised around memory, not MFMA:
//   * workgroup = (clip, 128-channel slice): 512 workgroups at cfg2, every CU busy (the composed path ran 14 launches
//     of a 128-wide register-staged GEMM per direction with K or N = 20);
//   * thread = one bf16 channel PAIR (4-byte accesses, 256 B per wave-instruction); the SHORT stream's rows of that
//     pair live in registers; a wave walks rows of the LONG stream;
//   * the probabilities are wave-uniform: they come in as fp32 "long-major" arrays [B, Ll, SP] (SP = short length
//     rounded up to 8, written by cqscore.hip); a row is one coalesced load (lane s holds p[s]) and reaches the FMAs
//     through v_readlane (see ld_probs);
//   * the long stream is read from HBM once per pass, NB = 8 row loads in flight per thread; `mid` / c2q / q2c never
//     exist in HBM; the concat is written straight from registers.
// Backward (per clip and slice): recomputes mid / c2q / q2c, produces dC, dQ in the same layout, and the THREE K = D
// contractions (dS_ = dc2q.Q^T + dq2c.mid^T, dS_t = C.dmid^T) as per-slice fp32 partial [Lc, Lq] tiles on MFMA --
// operands built in fragment layout straight from global 16-byte loads (dc2q = g2 + g3*C, dq2c = g4*C) or read from
// small bf16 LDS images (mid, dmid).  cq_softmax_bwd_parts sums the slices and applies both softmax backwards (dS,
// long-major fp32); cq_score_bwd turns dS into the gradients of the two score operands with the same skeleton.
// bf16 only; anything else stays on the composed path (ops.py).
#include "common.h"

namespace {

constexpr int DS = 128;          // channels per workgroup
constexpr int IMG_LD = DS + 8;   // bf16 LDS image row: 272 B (16 rows hit 16 different 16-byte bank groups)
constexpr int NB = 8;            // long-stream rows requested back to back before any is consumed
typedef __attribute__((ext_vector_type(2))) float f32x2;

struct ApplyArgs {
  const bf16_t* C; const bf16_t* Q;      // ctx [B,Lc,D], qry [B,Lq,D]
  const float* A1; const float* A2;      // S_ and S_t, long-major fp32 [B, Ll, SP]
  bf16_t* out;                           // fwd: cat4 [B*Lc, 4D]
  const bf16_t* g;                       // bwd: dcat4 [B*Lc, 4D]
  bf16_t* dC; bf16_t* dQ; float* parts;  // bwd outputs; parts [B][nsl][2][LcP][LqP] fp32, indexed (c, q)
  int Lc, Lq, D, LcP, LqP;
};

// E = the 16-bit element type (bf16_t / f16_t).  Pointers are carried as raw 16-bit (bf16_t-typed) bits; E names the
// conversions only.
template <typename E> __device__ __forceinline__ void st2(bf16_t* p, float a, float b) {
  const float v[2] = {a, b};
  Vec2<E>::store(reinterpret_cast<E*>(p), v);
}
__device__ __forceinline__ uint32_t ldw(const bf16_t* p) { return *reinterpret_cast<const uint32_t*>(p); }
template <typename E> __device__ __forceinline__ float lo16(uint32_t w) { return e16_lo<E>(w); }
template <typename E> __device__ __forceinline__ float hi16(uint32_t w) { return e16_hi<E>(w); }
// NB rows of one channel pair, unconditional (row index clamped: a conditional load gets its own s_waitcnt)
__device__ __forceinline__ void ld_rows(const bf16_t* base, int64_t ld, int r0, int rlast, uint32_t (&v)[NB]) {
#pragma unroll
  for (int j = 0; j < NB; ++j) v[j] = ldw(base + (int64_t)min(r0 + j, rlast) * ld);
}
// the short stream's rows of this thread's channel pair -> registers.  Rows past `n` hold the clamped last row, NOT zero:
// every use multiplies them by a probability column >= n, which cqscore.hip writes as exactly 0 (a `s < n ? x : 0` select
// per row cost an SGPR pair each -- 48-96 wave-uniform masks that the kernels spilled to VGPR lanes and re-read per use)
template <typename E, int SP>
__device__ __forceinline__ void ld_short(const bf16_t* base, int64_t ld, int n, f32x2 (&r)[SP]) {
  uint32_t w[SP];
#pragma unroll
  for (int s = 0; s < SP; ++s) w[s] = ldw(base + (int64_t)min(s, n - 1) * ld);
#pragma unroll
  for (int s = 0; s < SP; ++s) { r[s][0] = lo16<E>(w[s]); r[s][1] = hi16<E>(w[s]); }
}

// A row of SP probabilities is loaded ONCE per wave as one coalesced vector load -- lane s holds p[s] -- in the same
// batch as the long-stream rows, and broadcast to the FMAs through v_readlane (an SGPR operand).  (Tried before: fp32
// LDS images read as wave-uniform ds_read_b128 -- the LDS pipe became the bound, SQ_WAIT_INST_LDS = half of all wave
// cycles; plain `prow[s]` global reads -- hipcc turned only some of them into s_load, the rest into 64-lane same-address
// vector loads whose latency sat exposed in every row.)
template <int SP>
__device__ __forceinline__ void ld_probs(const float* __restrict__ A, int r0, int rlast, int lane, float (&pv)[NB]) {
  const int col = lane < SP ? lane : 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) pv[j] = A[(int64_t)min(r0 + j, rlast) * SP + col];
}
__device__ __forceinline__ float bcast(float v, int s) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), s)); }
// (the channel pair of a thread is ONE float2: p * r[s] is a v_pk_fma_f32 with the wave-uniform probability splat from an
//  SGPR -- half the VALU instructions of the two scalar FMAs these kernels, which are instruction-bound, used to issue)
template <int SP>
__device__ __forceinline__ void dot_rows(float pv, const f32x2 (&r)[SP], float& o0, float& o1) {
  f32x2 a = {0.f, 0.f};
#pragma unroll
  for (int s = 0; s < SP; ++s) a += bcast(pv, s) * r[s];
  o0 = a[0]; o1 = a[1];
}
template <int SP>
__device__ __forceinline__ void axpy_rows(float pv, float x0, float x1, f32x2 (&acc)[SP]) {
  const f32x2 x = {x0, x1};
#pragma unroll
  for (int s = 0; s < SP; ++s) acc[s] += bcast(pv, s) * x;
}
// Cross-wave sum of per-thread accumulators acc[SP][2]: every wave parks its partial in its own slot of
// red[4][SP][DS] (plain 8-byte stores), a barrier, then whoever needs a row adds the four slots.  (LDS float atomics
// into one shared [SP][DS] tile were the first version: ds_add_f32 runs at ~120 cycles per wave-instruction and the
// 48-144 of them per wave were 2/3 of these kernels' time.)
template <int SP>
__device__ __forceinline__ void red_put(float* __restrict__ red, int w, const f32x2 (&acc)[SP], int lane) {
#pragma unroll
  for (int s = 0; s < SP; ++s) *reinterpret_cast<f32x2*>(red + ((w * SP + s) * DS + 2 * lane)) = acc[s];
}
template <int SP>
__device__ __forceinline__ void red_row(const float* __restrict__ red, int s, int lane, float& o0, float& o1) {
  f32x2 t = *reinterpret_cast<const f32x2*>(red + (s * DS + 2 * lane));
#pragma unroll
  for (int w = 1; w < 4; ++w) t += *reinterpret_cast<const f32x2*>(red + ((w * SP + s) * DS + 2 * lane));
  o0 = t[0]; o1 = t[1];
}
template <int SP>
__device__ __forceinline__ f32x2 red_row2(const float* __restrict__ red, int s, int lane) {
  f32x2 t = *reinterpret_cast<const f32x2*>(red + (s * DS + 2 * lane));
#pragma unroll
  for (int w = 1; w < 4; ++w) t += *reinterpret_cast<const f32x2*>(red + ((w * SP + s) * DS + 2 * lane));
  return t;
}
template <int SP>
__device__ __forceinline__ void red_get(const float* __restrict__ red, f32x2 (&r)[SP], int lane) {
#pragma unroll
  for (int s = 0; s < SP; ++s) r[s] = red_row2<SP>(red, s, lane);
}
#define WAVE_ID() __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))
// One long-stream row at a time: a row's 24-48 broadcast probabilities live in SGPRs (v_readlane), and left alone the
// scheduler interleaves the eight rows of a batch -- 200-400 SGPRs wanted, 100 there: the kernels were spilling SGPRs
// to VGPR lanes (cq_apply_bwd_cshort: 693 v_writelane + 844 s_nop against 1194 FMAs)
#define ROW_FENCE() __builtin_amdgcn_sched_barrier(0)

// =====================================================================================================================
// forward, context = LONG stream (query short): A1[c][q] = S_, A2[c][q] = S_t
// =====================================================================================================================
template <typename E, int SP>
__global__ __launch_bounds__(256) void cq_apply_fwd_clong(ApplyArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);          // [4][SP][DS]
  const int lane = threadIdx.x & 63, w = WAVE_ID();
  const int b = blockIdx.y, d0 = blockIdx.x * DS + 2 * lane, D = a.D;
  const bf16_t* Cb = a.C + (int64_t)b * a.Lc * D + d0;
  const float* A1 = a.A1 + (int64_t)b * a.Lc * SP;
  const float* A2 = a.A2 + (int64_t)b * a.Lc * SP;
  f32x2 Qr[SP], mid[SP];
  ld_short<E, SP>(a.Q + (int64_t)b * a.Lq * D + d0, D, a.Lq, Qr);
#pragma unroll
  for (int q = 0; q < SP; ++q) mid[q][0] = mid[q][1] = 0.f;
  for (int c0 = w * NB; c0 < a.Lc; c0 += 4 * NB) {      // mid = S_t^T . C over this wave's rows
    uint32_t cr[NB];
    float p2[NB];
    ld_rows(Cb, D, c0, a.Lc - 1, cr);
    ld_probs<SP>(A2, c0, a.Lc - 1, lane, p2);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      ROW_FENCE();
      const float ok = c0 + j < a.Lc ? 1.f : 0.f;
      axpy_rows<SP>(p2[j], ok * lo16<E>(cr[j]), ok * hi16<E>(cr[j]), mid);
    }
  }
  red_put<SP>(red, w, mid, lane);
  __syncthreads();
  red_get<SP>(red, mid, lane);
  bf16_t* ob = a.out + (int64_t)b * a.Lc * 4 * D + d0;
  for (int c0 = w * NB; c0 < a.Lc; c0 += 4 * NB) {
    uint32_t cr[NB];
    float p1[NB];
    ld_rows(Cb, D, c0, a.Lc - 1, cr);
    ld_probs<SP>(A1, c0, a.Lc - 1, lane, p1);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      ROW_FENCE();
      const int c = c0 + j;
      float x0, x1, y0, y1;
      const float c0f = lo16<E>(cr[j]), c1f = hi16<E>(cr[j]);
      dot_rows<SP>(p1[j], Qr, x0, x1);                  // c2q
      dot_rows<SP>(p1[j], mid, y0, y1);                 // q2c
      if (c < a.Lc) {
        bf16_t* o = ob + (int64_t)c * 4 * D;
        *reinterpret_cast<uint32_t*>(o) = cr[j];
        st2<E>(o + D, x0, x1);
        st2<E>(o + 2 * D, c0f * x0, c1f * x1);
        st2<E>(o + 3 * D, c0f * y0, c1f * y1);
      }
    }
  }
}

// =====================================================================================================================
// forward, context = SHORT stream (query long): A1[q][c] = S_[c,q], A2[q][c] = S_t[c,q]
// =====================================================================================================================
template <typename E, int SP>
__global__ __launch_bounds__(256) void cq_apply_fwd_cshort(ApplyArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);          // [2][4][SP][DS]
  const int lane = threadIdx.x & 63, w = WAVE_ID();
  const int b = blockIdx.y, d0 = blockIdx.x * DS + 2 * lane, D = a.D;
  const bf16_t* Cb = a.C + (int64_t)b * a.Lc * D + d0;
  const bf16_t* Qb = a.Q + (int64_t)b * a.Lq * D + d0;
  const float* A1 = a.A1 + (int64_t)b * a.Lq * SP;
  const float* A2 = a.A2 + (int64_t)b * a.Lq * SP;
  f32x2 Cr[SP], a1[SP], a2[SP];
  ld_short<E, SP>(Cb, D, a.Lc, Cr);
#pragma unroll
  for (int c = 0; c < SP; ++c) a1[c][0] = a1[c][1] = a2[c][0] = a2[c][1] = 0.f;
  for (int q0 = w * NB; q0 < a.Lq; q0 += 4 * NB) {
    uint32_t qr[NB];
    float p1[NB], p2[NB];
    ld_rows(Qb, D, q0, a.Lq - 1, qr);
    ld_probs<SP>(A1, q0, a.Lq - 1, lane, p1);
    ld_probs<SP>(A2, q0, a.Lq - 1, lane, p2);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      ROW_FENCE();
      const float ok = q0 + j < a.Lq ? 1.f : 0.f;
      float m0, m1;
      dot_rows<SP>(p2[j], Cr, m0, m1);                                  // mid[q] = sum_c S_t[c,q] C[c]
      axpy_rows<SP>(p1[j], ok * lo16<E>(qr[j]), ok * hi16<E>(qr[j]), a1);     // c2q[c] += S_[c,q] Q[q]
      axpy_rows<SP>(p1[j], ok * m0, ok * m1, a2);                       // q2c[c] += S_[c,q] mid[q]
    }
  }
  red_put<SP>(red, w, a1, lane);
  red_put<SP>(red + 4 * SP * DS, w, a2, lane);
  __syncthreads();
  bf16_t* ob = a.out + (int64_t)b * a.Lc * 4 * D + d0;
  for (int c = w; c < a.Lc; c += 4) {
    const uint32_t cw = ldw(Cb + (int64_t)c * D);
    float x0, x1, y0, y1;
    red_row<SP>(red, c, lane, x0, x1);
    red_row<SP>(red + 4 * SP * DS, c, lane, y0, y1);
    bf16_t* o = ob + (int64_t)c * 4 * D;
    *reinterpret_cast<uint32_t*>(o) = cw;
    st2<E>(o + D, x0, x1);
    st2<E>(o + 2 * D, lo16<E>(cw) * x0, hi16<E>(cw) * x1);
    st2<E>(o + 3 * D, lo16<E>(cw) * y0, hi16<E>(cw) * y1);
  }
}

// =====================================================================================================================
// backward: the shared MFMA phase.  parts[0][c][q] = sum_d dc2q[c,d] Q[q,d] + dq2c[c,d] mid[q,d]   (d S_ partial)
//                                   parts[1][c][q] = sum_d C[c,d] dmid[q,d]                          (d S_t partial)
// A operand (rows -> m) = the query side (Q from global; mid, dmid from bf16 LDS images), B operand (rows -> n) = the
// context side, built in fragment layout from global 16-byte loads.  Result lane: c = n0 + (lane & 15), q = m0 + (lane >> 4) * 4 + r.
// =====================================================================================================================
__device__ __forceinline__ bf16x8 ldfrag(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }

template <typename E>
__device__ __forceinline__ void mfma_phase(const ApplyArgs& a, int b, int slice, const bf16_t* midI, const bf16_t* dmidI, int lane,
                                           int w) {
  const int D = a.D, mt_n = a.LqP / 16, nt_n = a.LcP / 16;
  const int kq = (lane >> 4) * 8;
  const bf16_t* Cb = a.C + (int64_t)b * a.Lc * D + slice * DS;
  const bf16_t* Qb = a.Q + (int64_t)b * a.Lq * D + slice * DS;
  const bf16_t* gb = a.g + (int64_t)b * a.Lc * 4 * D + slice * DS;
  float* pb = a.parts + ((int64_t)b * gridDim.x + slice) * 2 * a.LcP * a.LqP;
  for (int t = w; t < mt_n * nt_n; t += 4) {
    const int mt = t % mt_n, nt = t / mt_n;
    const int qrow = mt * 16 + (lane & 15), crow = nt * 16 + (lane & 15);
    const int qg = min(qrow, a.Lq - 1), cg = min(crow, a.Lc - 1);     // clamped: those products land in padding
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < DS / 32; ++ks) {
      const int k = ks * 32 + kq;
      const bf16x8 fq = ldfrag(Qb + (int64_t)qg * D + k);
      const bf16x8 fm = ldfrag(midI + qrow * IMG_LD + k);
      const bf16x8 fd = ldfrag(dmidI + qrow * IMG_LD + k);
      const bf16x8 c8 = ldfrag(Cb + (int64_t)cg * D + k);
      const bf16x8 g2 = ldfrag(gb + (int64_t)cg * 4 * D + D + k);
      const bf16x8 g3 = ldfrag(gb + (int64_t)cg * 4 * D + 2 * D + k);
      const bf16x8 g4 = ldfrag(gb + (int64_t)cg * 4 * D + 3 * D + k);
      bf16x8 dc2q, dq2c;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float cf = frag_get<E>(c8, e);
        dc2q[e] = bits_from_f<E>(frag_get<E>(g2, e) + frag_get<E>(g3, e) * cf);
        dq2c[e] = bits_from_f<E>(frag_get<E>(g4, e) * cf);
      }
      acc1 = mfma16<E>(fq, dc2q, acc1);
      acc1 = mfma16<E>(fm, dq2c, acc1);
      acc2 = mfma16<E>(fd, c8, acc2);
    }
    const int q0 = mt * 16 + (lane >> 4) * 4;
    float* p1 = pb + (int64_t)crow * a.LqP + q0;
    *reinterpret_cast<f32x4*>(p1) = acc1;
    *reinterpret_cast<f32x4*>(p1 + (int64_t)a.LcP * a.LqP) = acc2;
  }
}

// =====================================================================================================================
// backward, context = LONG stream
// =====================================================================================================================
template <typename E, int SP>
__global__ __launch_bounds__(256) void cq_apply_bwd_clong(ApplyArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);                   // [2][4][SP][DS]
  bf16_t* midI = reinterpret_cast<bf16_t*>(red + 8 * SP * DS);   // [LqP][IMG_LD]
  bf16_t* dmidI = midI + a.LqP * IMG_LD;
  const int lane = threadIdx.x & 63, w = WAVE_ID();
  const int b = blockIdx.y, slice = blockIdx.x, d0 = slice * DS + 2 * lane, D = a.D;
  const bf16_t* Cb = a.C + (int64_t)b * a.Lc * D + d0;
  const bf16_t* gb = a.g + (int64_t)b * a.Lc * 4 * D + d0;
  const float* A1 = a.A1 + (int64_t)b * a.Lc * SP;
  const float* A2 = a.A2 + (int64_t)b * a.Lc * SP;
  f32x2 Qr[SP], mid[SP];
  ld_short<E, SP>(a.Q + (int64_t)b * a.Lq * D + d0, D, a.Lq, Qr);
  for (int i = threadIdx.x; i < 2 * a.LqP * IMG_LD; i += 256) midI[i] = (bf16_t)0.f;
#pragma unroll
  for (int q = 0; q < SP; ++q) mid[q][0] = mid[q][1] = 0.f;
  for (int c0 = w * NB; c0 < a.Lc; c0 += 4 * NB) {      // mid (recomputed)
    uint32_t cr[NB];
    float p2[NB];
    ld_rows(Cb, D, c0, a.Lc - 1, cr);
    ld_probs<SP>(A2, c0, a.Lc - 1, lane, p2);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      ROW_FENCE();
      const float ok = c0 + j < a.Lc ? 1.f : 0.f;
      axpy_rows<SP>(p2[j], ok * lo16<E>(cr[j]), ok * hi16<E>(cr[j]), mid);
    }
  }
  red_put<SP>(red, w, mid, lane);
  __syncthreads();
  red_get<SP>(red, mid, lane);
  if (w == 0) {
#pragma unroll
    for (int q = 0; q < SP; ++q)
      if (q < a.Lq) st2<E>(midI + q * IMG_LD + 2 * lane, mid[q][0], mid[q][1]);
  }
  __syncthreads();                                      // everyone has read the mid partials: the slots are re-used below
  // ---- dQ = S_^T . dc2q, dmid = S_^T . dq2c over this wave's rows
  f32x2 dQa[SP], dMa[SP];
#pragma unroll
  for (int q = 0; q < SP; ++q) dQa[q][0] = dQa[q][1] = dMa[q][0] = dMa[q][1] = 0.f;
  for (int c0 = w * NB; c0 < a.Lc; c0 += 4 * NB) {
    uint32_t cr[NB], r2[NB], r3[NB], r4[NB];
    ld_rows(Cb, D, c0, a.Lc - 1, cr);
    ld_rows(gb + D, 4 * D, c0, a.Lc - 1, r2);
    ld_rows(gb + 2 * D, 4 * D, c0, a.Lc - 1, r3);
    ld_rows(gb + 3 * D, 4 * D, c0, a.Lc - 1, r4);
    float p1[NB];
    ld_probs<SP>(A1, c0, a.Lc - 1, lane, p1);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      ROW_FENCE();
      const float ok = c0 + j < a.Lc ? 1.f : 0.f;
      axpy_rows<SP>(p1[j], ok * (lo16<E>(r2[j]) + lo16<E>(r3[j]) * lo16<E>(cr[j])), ok * (hi16<E>(r2[j]) + hi16<E>(r3[j]) * hi16<E>(cr[j])), dQa);
      axpy_rows<SP>(p1[j], ok * lo16<E>(r4[j]) * lo16<E>(cr[j]), ok * hi16<E>(r4[j]) * hi16<E>(cr[j]), dMa);
    }
  }
  red_put<SP>(red, w, dQa, lane);
  red_put<SP>(red + 4 * SP * DS, w, dMa, lane);
  __syncthreads();
  bf16_t* dQb = a.dQ + (int64_t)b * a.Lq * D + d0;
  for (int q = w; q < a.Lq; q += 4) {
    float x0, x1;
    red_row<SP>(red, q, lane, x0, x1);
    st2<E>(dQb + (int64_t)q * D, x0, x1);
  }
  red_get<SP>(red + 4 * SP * DS, dMa, lane);
  if (w == 0) {
#pragma unroll
    for (int q = 0; q < SP; ++q)
      if (q < a.Lq) st2<E>(dmidI + q * IMG_LD + 2 * lane, dMa[q][0], dMa[q][1]);
  }
  // ---- dC = g1 + g3*c2q + g4*q2c + S_t . dmid
  bf16_t* dCb = a.dC + (int64_t)b * a.Lc * D + d0;
  for (int c0 = w * NB; c0 < a.Lc; c0 += 4 * NB) {
    uint32_t r1[NB], r3[NB], r4[NB];
    ld_rows(gb, 4 * D, c0, a.Lc - 1, r1);
    ld_rows(gb + 2 * D, 4 * D, c0, a.Lc - 1, r3);
    ld_rows(gb + 3 * D, 4 * D, c0, a.Lc - 1, r4);
    float p1[NB], p2[NB];
    ld_probs<SP>(A1, c0, a.Lc - 1, lane, p1);
    ld_probs<SP>(A2, c0, a.Lc - 1, lane, p2);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      ROW_FENCE();
      const int c = c0 + j;
      float x0, x1, y0, y1, z0, z1;
      dot_rows<SP>(p1[j], Qr, x0, x1);
      dot_rows<SP>(p1[j], mid, y0, y1);
      dot_rows<SP>(p2[j], dMa, z0, z1);
      if (c < a.Lc)
        st2<E>(dCb + (int64_t)c * D, lo16<E>(r1[j]) + lo16<E>(r3[j]) * x0 + lo16<E>(r4[j]) * y0 + z0,
            hi16<E>(r1[j]) + hi16<E>(r3[j]) * x1 + hi16<E>(r4[j]) * y1 + z1);
    }
  }
  __syncthreads();                                      // the dmid image is complete
  mfma_phase<E>(a, b, slice, midI, dmidI, lane, w);
}

// =====================================================================================================================
// backward, context = SHORT stream
// =====================================================================================================================
template <typename E, int SP>
__global__ __launch_bounds__(256) void cq_apply_bwd_cshort(ApplyArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);                   // [4][SP][DS], used for three reductions in turn
  bf16_t* midI = reinterpret_cast<bf16_t*>(red + 4 * SP * DS);   // [LqP][IMG_LD]
  bf16_t* dmidI = midI + a.LqP * IMG_LD;
  const int lane = threadIdx.x & 63, w = WAVE_ID();
  const int b = blockIdx.y, slice = blockIdx.x, d0 = slice * DS + 2 * lane, D = a.D;
  const bf16_t* Cb = a.C + (int64_t)b * a.Lc * D + d0;
  const bf16_t* Qb = a.Q + (int64_t)b * a.Lq * D + d0;
  const bf16_t* gb = a.g + (int64_t)b * a.Lc * 4 * D + d0;
  const float* A1 = a.A1 + (int64_t)b * a.Lq * SP;
  const float* A2 = a.A2 + (int64_t)b * a.Lq * SP;
  for (int r = a.Lq + w; r < a.LqP; r += 4) {           // image rows past Lq feed padded tile rows: zero
    st2<E>(midI + r * IMG_LD + 2 * lane, 0.f, 0.f);
    st2<E>(dmidI + r * IMG_LD + 2 * lane, 0.f, 0.f);
  }
  __syncthreads();
  bf16_t* dQb = a.dQ + (int64_t)b * a.Lq * D + d0;
  constexpr int RW = SP / 4;                            // context rows c = w, w + 4, ... of this wave (<= SP / 4)
  f32x2 xr[RW], yr[RW], zr[RW];
  {   // pass A: forward quantities (c2q, q2c accumulators; the mid image)
    f32x2 Cr[SP], a1[SP], a2[SP];
    ld_short<E, SP>(Cb, D, a.Lc, Cr);
#pragma unroll
    for (int c = 0; c < SP; ++c) a1[c][0] = a1[c][1] = a2[c][0] = a2[c][1] = 0.f;
    for (int q0 = w * NB; q0 < a.Lq; q0 += 4 * NB) {
      uint32_t qr[NB];
      float p1[NB], p2[NB];
      ld_rows(Qb, D, q0, a.Lq - 1, qr);
      ld_probs<SP>(A1, q0, a.Lq - 1, lane, p1);
      ld_probs<SP>(A2, q0, a.Lq - 1, lane, p2);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        ROW_FENCE();
        const float ok = q0 + j < a.Lq ? 1.f : 0.f;
        float m0, m1;
        dot_rows<SP>(p2[j], Cr, m0, m1);
        if (q0 + j < a.Lq) st2<E>(midI + (q0 + j) * IMG_LD + 2 * lane, m0, m1);
        axpy_rows<SP>(p1[j], ok * lo16<E>(qr[j]), ok * hi16<E>(qr[j]), a1);
        axpy_rows<SP>(p1[j], ok * m0, ok * m1, a2);
      }
    }
    red_put<SP>(red, w, a1, lane);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RW; ++k) xr[k] = red_row2<SP>(red, min(w + 4 * k, SP - 1), lane);
    __syncthreads();
    red_put<SP>(red, w, a2, lane);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RW; ++k) yr[k] = red_row2<SP>(red, min(w + 4 * k, SP - 1), lane);
    __syncthreads();
  }
  {   // pass B: dQ rows, the dmid image, dC_mid accumulators
    f32x2 X1[SP], X2[SP], a3[SP];               // dc2q, dq2c rows of this channel pair
#pragma unroll
    for (int c = 0; c < SP; ++c) {
      const int cc = min(c, a.Lc - 1);
      const bf16_t* gr = gb + (int64_t)cc * 4 * D;
      const uint32_t cv = ldw(Cb + (int64_t)cc * D), g2 = ldw(gr + D), g3 = ldw(gr + 2 * D), g4 = ldw(gr + 3 * D);
      // (rows c >= Lc repeat the last row: their probability columns are exactly 0, see ld_short)
      X1[c][0] = lo16<E>(g2) + lo16<E>(g3) * lo16<E>(cv); X1[c][1] = hi16<E>(g2) + hi16<E>(g3) * hi16<E>(cv);
      X2[c][0] = lo16<E>(g4) * lo16<E>(cv); X2[c][1] = hi16<E>(g4) * hi16<E>(cv);
      a3[c][0] = a3[c][1] = 0.f;
    }
    for (int q0 = w * NB; q0 < a.Lq; q0 += 4 * NB) {
      float p1[NB], p2[NB];
      ld_probs<SP>(A1, q0, a.Lq - 1, lane, p1);
      ld_probs<SP>(A2, q0, a.Lq - 1, lane, p2);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        ROW_FENCE();
        const int q = q0 + j;
        const float ok = q < a.Lq ? 1.f : 0.f;
        float x0, x1, m0, m1;
        dot_rows<SP>(p1[j], X1, x0, x1);                 // dQ[q] = sum_c S_[c,q] dc2q[c]
        dot_rows<SP>(p1[j], X2, m0, m1);                 // dmid[q] = sum_c S_[c,q] dq2c[c]
        if (q < a.Lq) {
          st2<E>(dQb + (int64_t)q * D, x0, x1);
          st2<E>(dmidI + q * IMG_LD + 2 * lane, m0, m1);
        }
        axpy_rows<SP>(p2[j], ok * m0, ok * m1, a3);      // dC_mid[c] += S_t[c,q] dmid[q]
      }
    }
    red_put<SP>(red, w, a3, lane);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < RW; ++k) zr[k] = red_row2<SP>(red, min(w + 4 * k, SP - 1), lane);
  bf16_t* dCb = a.dC + (int64_t)b * a.Lc * D + d0;
#pragma unroll
  for (int k = 0; k < RW; ++k) {
    const int c = w + 4 * k;
    if (c < a.Lc) {
      const bf16_t* gr = gb + (int64_t)c * 4 * D;
      const uint32_t g1 = ldw(gr), g3 = ldw(gr + 2 * D), g4 = ldw(gr + 3 * D);
      st2<E>(dCb + (int64_t)c * D, lo16<E>(g1) + lo16<E>(g3) * xr[k][0] + lo16<E>(g4) * yr[k][0] + zr[k][0],
          hi16<E>(g1) + hi16<E>(g3) * xr[k][1] + hi16<E>(g4) * yr[k][1] + zr[k][1]);
    }
  }
  mfma_phase<E>(a, b, slice, midI, dmidI, lane, w);
}

// =====================================================================================================================
// backward, context = SHORT stream, on MFMA (Lq <= 128, Lc <= 32).  The register kernel above does its six small
// contractions (K = Lc or K = Lq, N = 128 channels) as v_pk_fma on wave-uniform probabilities: 2.4 GFLOP of fp32 VALU work
// per launch at two waves per SIMD = 120 us for 84 MB of traffic.  Here every operand is a 16-bit LDS image and every
// contraction a handful of 16x16x32 MFMAs (the probabilities rounded to the element type, as the fused attention kernels
// do with P):
//   images   Qi [128][IMG_LD] (later dmid), midI [128][IMG_LD], dQ stage [128][IMG_LD], Ci / X1i / X2i [32][IMG_LD]
//            (X1 = dc2q = g2 + g3*C, X2 = dq2c = g4*C), P1i / P2i [128][PLD] = S_ / S_t as [q][c]; rows / columns past
//            the real lengths are ZERO in every image (no reliance on clamped rows);
//   K = c    (mid = P2.C, dQ = P1.X1, dmid = P1.X2): A = a P image row (k contiguous), B = a [c][d] image read transposed;
//   K = q    (c2q = P1^T.Q, q2c = P1^T.mid, dCm = P2^T.dmid): A = a P image read transposed, B = a [q][d] image read
//            transposed; wave w owns channel tiles 2w, 2w + 1 of all three, so dC = g1 + g3*c2q + g4*q2c + dCm is formed
//            in registers;
//   then the shared mfma_phase (dS_ / dS_t partials) on the mid / dmid images.
// =====================================================================================================================
// Geometry of the MFMA kernels: the long stream has at most NL rows; a workgroup takes a DSV-channel slice so that a long image
// is always 16 K elements: NL = 128 -> 128 channels ([128][136] images, P rows of 32 + 8), NL = 256 (BaseFast's T = 256) ->
// 64 channels ([256][72] images, P rows of 32: 157 KB of LDS in the largest kernel).
template <int NL> struct Geo {
  static constexpr int DSV = NL == 256 ? 64 : 128;     // channels per workgroup
  static constexpr int IMG = DSV + 8;                  // image row (elements): 16-byte aligned, rows on different bank groups
  static constexpr int PLD = NL == 256 ? 32 : 40;      // P image row (elements)
  static constexpr int CT = DSV / 16, LT = NL / 16, KSL = NL / 32;
  static constexpr int NN = CT / 4;                    // channel tiles per wave in the K = long products
  static constexpr int MM = LT / 4;                    // long tiles per wave in the K = short products
};

// 16 x 32 fragment (lane & 15 -> column `t * 16 + .` of the image, 8 consecutive image ROWS ks * 32 + 8 * (lane >> 4) ..) from a
// row-major 16-bit image with `rb` bytes per row: two ds_read_b64_tr_b16 (common.h lds_read_tr: the caller waits and pins)
__device__ __forceinline__ bf16x8 img_frag_tr(const bf16_t* img, int rb, int t, int ks, int lane) {
  const int g = lane >> 4, ii = lane & 15, qq = ii >> 2, p = ii & 3;
  const int r = ks * 32 + 8 * g + qq;
  const unsigned char* base = reinterpret_cast<const unsigned char*>(img) + (t << 5) + p * 8;
  union { struct { s16x4 l, h; } s; bf16x8 v; } u;
  u.s.l = lds_read_tr(base + r * rb);
  u.s.h = lds_read_tr(base + (r + 4) * rb);
  return u.v;
}

// four consecutive elements from an accumulator quad, one 8-byte store
template <typename E> __device__ __forceinline__ void st4(bf16_t* p, const f32x4& v) {
  typedef __attribute__((ext_vector_type(4))) bf16_t b4;
  b4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = bits_from_f<E>(v[r]);
  *reinterpret_cast<b4*>(p) = o;
}
// ---- the two product shapes, images in LDS (long stream: NL rows, short: 32 rows; P: [long][short cols]) -----------------
// out[l][d] = sum_s P[l][s] X[s][d] (K = 32 short rows): wave w owns long tiles MM*w .. and all channel tiles; issued as the
// transposed product X^T . P^T, so a lane holds FOUR consecutive channels of one row -- one 8-byte store per tile (four
// 2-byte ones otherwise).  The result goes into the image dst [NL][IMG], or (GLB) to global rows of length ld, rows < n.
template <typename E, int NL, bool GLB = false>
__device__ __forceinline__ void prod_long_rows(const bf16_t* Pi, const bf16_t* Xi, bf16_t* dst, int w, int lane, int64_t ld = 0,
                                               int n = 0) {
  typedef Geo<NL> G;
  const int r16 = lane & 15, kq = lane >> 4;
  bf16x8 fb[G::CT];
#pragma unroll
  for (int nt = 0; nt < G::CT; ++nt) fb[nt] = img_frag_tr(Xi, G::IMG * 2, nt, 0, lane);
  lgkm_wait<0>();
#pragma unroll
  for (int nt = 0; nt < G::CT; ++nt) frag_pin(fb[nt]);
#pragma unroll
  for (int mm = 0; mm < G::MM; ++mm) {
    const int mt = G::MM * w + mm, row = mt * 16 + r16;
    const bf16x8 fa = ldfrag(Pi + row * G::PLD + kq * 8);
#pragma unroll
    for (int nt = 0; nt < G::CT; ++nt) {
      const f32x4 acc = mfma16<E>(fb[nt], fa, (f32x4){0.f, 0.f, 0.f, 0.f});
      if constexpr (GLB) {
        if (row < n) st4<E>(dst + (int64_t)row * ld + nt * 16 + kq * 4, acc);
      } else {
        st4<E>(dst + row * G::IMG + nt * 16 + kq * 4, acc);
      }
    }
  }
}
// acc[st][nn] (+)= sum_l P[l][s] Y[l][d] (K = NL long rows), as the transposed product Y^T . P: wave w owns channel tiles
// NN*w .. and both short tiles; lane (r16, kq) holds row s = st * 16 + r16, channels d = (NN*w + nn) * 16 + kq * 4 + r
template <typename E, int NL>
__device__ __forceinline__ void prod_short_rows(const bf16_t* Pi, const bf16_t* Yi, f32x4 (&acc)[2][Geo<NL>::NN], int w, int lane) {
  typedef Geo<NL> G;
  bf16x8 fa[2][G::KSL], fb[G::NN][G::KSL];
#pragma unroll
  for (int ks = 0; ks < G::KSL; ++ks) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) fa[mt][ks] = img_frag_tr(Pi, G::PLD * 2, mt, ks, lane);
#pragma unroll
    for (int nn = 0; nn < G::NN; ++nn) fb[nn][ks] = img_frag_tr(Yi, G::IMG * 2, G::NN * w + nn, ks, lane);
  }
  lgkm_wait<0>();
#pragma unroll
  for (int ks = 0; ks < G::KSL; ++ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i) frag_pin(fa[i][ks]);
#pragma unroll
    for (int i = 0; i < G::NN; ++i) frag_pin(fb[i][ks]);
  }
#pragma unroll
  for (int ks = 0; ks < G::KSL; ++ks)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nn = 0; nn < G::NN; ++nn) acc[mt][nn] = mfma16<E>(fb[nn][ks], fa[mt][ks], acc[mt][nn]);
}
template <int NN> __device__ __forceinline__ void zero2n(f32x4 (&acc)[2][NN]) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
}
// a [rows <= NR][DSV-channel slice] global matrix -> image (rows >= n zero), 16-byte chunks; NR = image rows (32 or NL)
template <int NL, int NR>
__device__ __forceinline__ void stage_rows(const bf16_t* src, int64_t ld, int n, bf16_t* img, int tid) {
  typedef Geo<NL> G;
  constexpr int CPR = G::DSV / 8;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < (NR * CPR + 255) / 256; ++j) {
    const int i = tid + 256 * j, row = i / CPR, ch = (i % CPR) * 8;
    if (NR * CPR % 256 != 0 && row >= NR) break;
    bf16x8 v = zero8;
    if (row < n) v = ldfrag(src + (int64_t)row * ld + ch);
    *reinterpret_cast<bf16x8*>(img + row * G::IMG + ch) = v;
  }
}
// context rows -> Ci, X1i = dc2q = g2 + g3*C, X2i = dq2c = g4*C (rows >= n zero)
template <typename E, int NL, int NR>
__device__ __forceinline__ void stage_ctx(const bf16_t* Cb, const bf16_t* gb, int D, int n, bf16_t* Ci, bf16_t* X1i, bf16_t* X2i, int tid) {
  typedef Geo<NL> G;
  constexpr int CPR = G::DSV / 8;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < (NR * CPR + 255) / 256; ++j) {
    const int i = tid + 256 * j, row = i / CPR, ch = (i % CPR) * 8;
    if (NR * CPR % 256 != 0 && row >= NR) break;
    bf16x8 c8 = zero8, x1 = zero8, x2 = zero8;
    if (row < n) {
      const bf16_t* gr = gb + (int64_t)row * 4 * D + ch;
      c8 = ldfrag(Cb + (int64_t)row * D + ch);
      const bf16x8 g2 = ldfrag(gr + D), g3 = ldfrag(gr + 2 * D), g4 = ldfrag(gr + 3 * D);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float cf = frag_get<E>(c8, e);
        x1[e] = bits_from_f<E>(frag_get<E>(g2, e) + frag_get<E>(g3, e) * cf);
        x2[e] = bits_from_f<E>(frag_get<E>(g4, e) * cf);
      }
    }
    *reinterpret_cast<bf16x8*>(Ci + row * G::IMG + ch) = c8;
    *reinterpret_cast<bf16x8*>(X1i + row * G::IMG + ch) = x1;
    *reinterpret_cast<bf16x8*>(X2i + row * G::IMG + ch) = x2;
  }
}
// fp32 long-major [nl][SP] rows -> 16-bit image [NL][PLD] (rows >= nl and columns >= SP zero)
template <typename E, int NL>
__device__ __forceinline__ void stage_probs(const float* A, int nl, int SP, bf16_t* Pi, int tid) {
  typedef Geo<NL> G;
#pragma unroll
  for (int j = 0; j < NL * 4 / 256; ++j) {
    const int i = tid + 256 * j, row = i >> 2, cg = (i & 3) * 8;
    bf16x8 p = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < nl && cg < SP) {
      const f32x4 u0 = *reinterpret_cast<const f32x4*>(A + (int64_t)row * SP + cg), u1 = *reinterpret_cast<const f32x4*>(A + (int64_t)row * SP + cg + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { p[e] = bits_from_f<E>(u0[e]); p[4 + e] = bits_from_f<E>(u1[e]); }
    }
    *reinterpret_cast<bf16x8*>(Pi + row * G::PLD + cg) = p;
  }
}
// short-side accumulators -> a [32][IMG] image / a global [n][D] matrix (8-byte stores)
template <typename E, int NL>
__device__ __forceinline__ void put_short_img(const f32x4 (&acc)[2][Geo<NL>::NN], bf16_t* img, int w, int lane) {
  typedef Geo<NL> G;
  const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nn = 0; nn < G::NN; ++nn) st4<E>(img + (mt * 16 + r16) * G::IMG + (G::NN * w + nn) * 16 + kq * 4, acc[mt][nn]);
}
template <typename E, int NL>
__device__ __forceinline__ void put_short_global(const f32x4 (&acc)[2][Geo<NL>::NN], bf16_t* dst, int64_t ld, int n, int w, int lane) {
  typedef Geo<NL> G;
  const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nn = 0; nn < G::NN; ++nn) {
      const int srow = mt * 16 + r16;
      if (srow < n) st4<E>(dst + (int64_t)srow * ld + (G::NN * w + nn) * 16 + kq * 4, acc[mt][nn]);
    }
}

// mfma_phase with the context side (C, dc2q, dq2c) read from the LDS images the MFMA kernels hold anyway (k = channels is the
// contiguous index of every image row: plain 16-byte fragment reads, no global re-read, no conversions); the query rows
// from the image Qi when it is still alive (QIMG), else from global.
template <typename E, int NL, bool QIMG>
__device__ __forceinline__ void mfma_phase_img(const ApplyArgs& a, int b, int slice, const bf16_t* Qi, const bf16_t* midI,
                                               const bf16_t* dmidI, const bf16_t* Ci, const bf16_t* X1i, const bf16_t* X2i,
                                               int lane, int w) {
  typedef Geo<NL> G;
  const int D = a.D, mt_n = a.LqP / 16, nt_n = a.LcP / 16;
  const int kq = (lane >> 4) * 8;
  const bf16_t* Qb = a.Q + (int64_t)b * a.Lq * D + slice * G::DSV;
  float* pb = a.parts + ((int64_t)b * gridDim.x + slice) * 2 * a.LcP * a.LqP;
  for (int t = w; t < mt_n * nt_n; t += 4) {
    const int mt = t % mt_n, nt = t / mt_n;
    const int qrow = mt * 16 + (lane & 15), crow = nt * 16 + (lane & 15);
    const int qg = min(qrow, a.Lq - 1);
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < G::DSV / 32; ++ks) {
      const int k = ks * 32 + kq;
      const bf16x8 fq = QIMG ? ldfrag(Qi + qrow * G::IMG + k) : ldfrag(Qb + (int64_t)qg * D + k);
      const bf16x8 fm = ldfrag(midI + qrow * G::IMG + k);
      const bf16x8 fd = ldfrag(dmidI + qrow * G::IMG + k);
      acc1 = mfma16<E>(fq, ldfrag(X1i + crow * G::IMG + k), acc1);
      acc1 = mfma16<E>(fm, ldfrag(X2i + crow * G::IMG + k), acc1);
      acc2 = mfma16<E>(fd, ldfrag(Ci + crow * G::IMG + k), acc2);
    }
    const int q0 = mt * 16 + (lane >> 4) * 4;
    float* p1 = pb + (int64_t)crow * a.LqP + q0;
    *reinterpret_cast<f32x4*>(p1) = acc1;
    *reinterpret_cast<f32x4*>(p1 + (int64_t)a.LcP * a.LqP) = acc2;
  }
}

// backward, context = SHORT stream (Lq <= NL, Lc <= 32)
template <typename E, int NL>
__global__ __launch_bounds__(256) void cq_apply_bwd_cshort_mfma(ApplyArgs a) {
  typedef Geo<NL> G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Qi = reinterpret_cast<bf16_t*>(smem);          // [NL][IMG]; dmid after c2q is done
  bf16_t* midI = Qi + NL * G::IMG;
  bf16_t* Ci = midI + NL * G::IMG;                       // [32][IMG]
  bf16_t* X1i = Ci + 32 * G::IMG;
  bf16_t* X2i = X1i + 32 * G::IMG;
  bf16_t* P1i = X2i + 32 * G::IMG;                       // [NL][PLD]
  bf16_t* P2i = P1i + NL * G::PLD;
  const int tid = threadIdx.x, lane = tid & 63, w = WAVE_ID();
  const int r16 = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y, slice = blockIdx.x, D = a.D, Lc = a.Lc, Lq = a.Lq;
  const bf16_t* Cb = a.C + (int64_t)b * Lc * D + slice * G::DSV;
  const bf16_t* Qb = a.Q + (int64_t)b * Lq * D + slice * G::DSV;
  const bf16_t* gb = a.g + (int64_t)b * Lc * 4 * D + slice * G::DSV;
  const int SP = (Lc + 7) / 8 * 8;                       // the probability rows' length (cqscore.hip: columns >= Lc are exactly 0)
  stage_rows<NL, NL>(Qb, D, Lq, Qi, tid);
  stage_ctx<E, NL, 32>(Cb, gb, D, Lc, Ci, X1i, X2i, tid);
  stage_probs<E, NL>(a.A1 + (int64_t)b * Lq * SP, Lq, SP, P1i, tid);
  stage_probs<E, NL>(a.A2 + (int64_t)b * Lq * SP, Lq, SP, P2i, tid);
  __syncthreads();
  prod_long_rows<E, NL>(P2i, Ci, midI, w, lane);                                                     // mid[q] = sum_c S_t[c,q] C[c]
  prod_long_rows<E, NL, true>(P1i, X1i, a.dQ + (int64_t)b * Lq * D + slice * G::DSV, w, lane, D, Lq);  // dQ[q] = sum_c S_[c,q] dc2q[c]
  __syncthreads();
  f32x4 c2q[2][G::NN], q2c[2][G::NN], dcm[2][G::NN];
  zero2n(c2q); zero2n(q2c); zero2n(dcm);
  prod_short_rows<E, NL>(P1i, Qi, c2q, w, lane);
  prod_short_rows<E, NL>(P1i, midI, q2c, w, lane);
  __syncthreads();                                       // every wave is done with Qi
  bf16_t* dmidI = Qi;
  prod_long_rows<E, NL>(P1i, X2i, dmidI, w, lane);       // dmid[q] = sum_c S_[c,q] dq2c[c]
  __syncthreads();
  prod_short_rows<E, NL>(P2i, dmidI, dcm, w, lane);      // dC through mid: sum_q S_t[c,q] dmid[q]
  bf16_t* dCb = a.dC + (int64_t)b * Lc * D + slice * G::DSV;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nn = 0; nn < G::NN; ++nn) {
      const int c = mt * 16 + r16, d = (G::NN * w + nn) * 16 + kq * 4;       // this lane: row c, four channels from d
      if (c < Lc) {
        typedef __attribute__((ext_vector_type(4))) bf16_t b4;
        const bf16_t* gr = gb + (int64_t)c * 4 * D + d;
        const b4 g1 = *reinterpret_cast<const b4*>(gr), g3 = *reinterpret_cast<const b4*>(gr + 2 * D), g4 = *reinterpret_cast<const b4*>(gr + 3 * D);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          o[r] = bits_to_f<E>(g1[r]) + bits_to_f<E>(g3[r]) * c2q[mt][nn][r] + bits_to_f<E>(g4[r]) * q2c[mt][nn][r] + dcm[mt][nn][r];
        st4<E>(dCb + (int64_t)c * D + d, o);
      }
    }
  mfma_phase_img<E, NL, false>(a, b, slice, nullptr, midI, dmidI, Ci, X1i, X2i, lane, w);
}

// the 4-way concat of NR context rows from the images Ci (context), Xi (c2q), Yi (q2c): [C | c2q | C*c2q | C*q2c], 16-byte stores
template <typename E, int NL, int NR>
__device__ __forceinline__ void put_cat4(const bf16_t* Ci, const bf16_t* Xi, const bf16_t* Yi, bf16_t* ob, int D, int n, int tid) {
  typedef Geo<NL> G;
  constexpr int CPR = G::DSV / 8;
#pragma unroll
  for (int j = 0; j < (NR * CPR + 255) / 256; ++j) {
    const int i = tid + 256 * j, row = i / CPR, ch = (i % CPR) * 8;
    if (row < n && row < NR) {
      const bf16x8 c8 = *reinterpret_cast<const bf16x8*>(Ci + row * G::IMG + ch), x = *reinterpret_cast<const bf16x8*>(Xi + row * G::IMG + ch);
      const bf16x8 y = *reinterpret_cast<const bf16x8*>(Yi + row * G::IMG + ch);
      bf16x8 cx, cy;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float cf = frag_get<E>(c8, e);
        cx[e] = bits_from_f<E>(cf * frag_get<E>(x, e));
        cy[e] = bits_from_f<E>(cf * frag_get<E>(y, e));
      }
      bf16_t* o = ob + (int64_t)row * 4 * D + ch;
      *reinterpret_cast<bf16x8*>(o) = c8;
      *reinterpret_cast<bf16x8*>(o + D) = x;
      *reinterpret_cast<bf16x8*>(o + 2 * D) = cx;
      *reinterpret_cast<bf16x8*>(o + 3 * D) = cy;
    }
  }
}

// forward on MFMA, context = SHORT stream (Lq <= NL, Lc <= 32): mid = P2.C (K = 32), c2q = P1^T.Q, q2c = P1^T.mid (K = NL)
template <typename E, int NL>
__global__ __launch_bounds__(256) void cq_apply_fwd_cshort_mfma(ApplyArgs a) {
  typedef Geo<NL> G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Qi = reinterpret_cast<bf16_t*>(smem);          // [NL][IMG]
  bf16_t* midI = Qi + NL * G::IMG;                       // [NL][IMG]
  bf16_t* Ci = midI + NL * G::IMG;                       // [32][IMG]
  bf16_t* Xi = Ci + 32 * G::IMG;
  bf16_t* Yi = Xi + 32 * G::IMG;
  bf16_t* P1i = Yi + 32 * G::IMG;                        // [NL][PLD]
  bf16_t* P2i = P1i + NL * G::PLD;
  const int tid = threadIdx.x, lane = tid & 63, w = WAVE_ID();
  const int b = blockIdx.y, slice = blockIdx.x, D = a.D, Lc = a.Lc, Lq = a.Lq;
  const int SP = (Lc + 7) / 8 * 8;
  stage_rows<NL, NL>(a.Q + (int64_t)b * Lq * D + slice * G::DSV, D, Lq, Qi, tid);
  stage_rows<NL, 32>(a.C + (int64_t)b * Lc * D + slice * G::DSV, D, Lc, Ci, tid);
  stage_probs<E, NL>(a.A1 + (int64_t)b * Lq * SP, Lq, SP, P1i, tid);
  stage_probs<E, NL>(a.A2 + (int64_t)b * Lq * SP, Lq, SP, P2i, tid);
  __syncthreads();
  prod_long_rows<E, NL>(P2i, Ci, midI, w, lane);
  f32x4 c2q[2][G::NN], q2c[2][G::NN];
  zero2n(c2q); zero2n(q2c);
  prod_short_rows<E, NL>(P1i, Qi, c2q, w, lane);
  put_short_img<E, NL>(c2q, Xi, w, lane);
  __syncthreads();
  prod_short_rows<E, NL>(P1i, midI, q2c, w, lane);
  put_short_img<E, NL>(q2c, Yi, w, lane);
  __syncthreads();
  put_cat4<E, NL, 32>(Ci, Xi, Yi, a.out + (int64_t)b * Lc * 4 * D + slice * G::DSV, D, Lc, tid);
}

// forward on MFMA, context = LONG stream (Lc <= NL, Lq <= 32): c2q = P1.Q, mid = P2^T.C, q2c = P1.mid
template <typename E, int NL>
__global__ __launch_bounds__(256) void cq_apply_fwd_clong_mfma(ApplyArgs a) {
  typedef Geo<NL> G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Ci = reinterpret_cast<bf16_t*>(smem);          // [NL][IMG]
  bf16_t* Xi = Ci + NL * G::IMG;
  bf16_t* Yi = Xi + NL * G::IMG;
  bf16_t* Qi = Yi + NL * G::IMG;                         // [32][IMG]
  bf16_t* midI = Qi + 32 * G::IMG;
  bf16_t* P1i = midI + 32 * G::IMG;                      // [NL][PLD]
  bf16_t* P2i = P1i + NL * G::PLD;
  const int tid = threadIdx.x, lane = tid & 63, w = WAVE_ID();
  const int b = blockIdx.y, slice = blockIdx.x, D = a.D, Lc = a.Lc, Lq = a.Lq;
  const int SP = (Lq + 7) / 8 * 8;
  stage_rows<NL, NL>(a.C + (int64_t)b * Lc * D + slice * G::DSV, D, Lc, Ci, tid);
  stage_rows<NL, 32>(a.Q + (int64_t)b * Lq * D + slice * G::DSV, D, Lq, Qi, tid);
  stage_probs<E, NL>(a.A1 + (int64_t)b * Lc * SP, Lc, SP, P1i, tid);
  stage_probs<E, NL>(a.A2 + (int64_t)b * Lc * SP, Lc, SP, P2i, tid);
  __syncthreads();
  f32x4 mid[2][G::NN];
  zero2n(mid);
  prod_short_rows<E, NL>(P2i, Ci, mid, w, lane);
  put_short_img<E, NL>(mid, midI, w, lane);
  prod_long_rows<E, NL>(P1i, Qi, Xi, w, lane);
  __syncthreads();
  prod_long_rows<E, NL>(P1i, midI, Yi, w, lane);
  __syncthreads();
  put_cat4<E, NL, NL>(Ci, Xi, Yi, a.out + (int64_t)b * Lc * 4 * D + slice * G::DSV, D, Lc, tid);
}

// backward, context = LONG stream (Lc <= NL, Lq <= 32), same two product shapes wired the other way round:
//   mid = S_t^T.C, dQ = S_^T.dc2q, dmid = S_^T.dq2c   (K = the context rows)       -> 32-row images / global dQ
//   the dS_ / dS_t partial tiles from the images (mfma_phase_img), while Ci / X1i / X2i are still alive
//   c2q = S_.Q, q2c = S_.mid, dCm = S_t.dmid            (K = the 32 query rows)      -> three long images over the dead
//   Ci / X1i / X2i regions; a cooperative epilogue then forms dC = g1 + g3*c2q + g4*q2c + dCm with 16-byte accesses.
template <typename E, int NL>
__global__ __launch_bounds__(256) void cq_apply_bwd_clong_mfma(ApplyArgs a) {
  typedef Geo<NL> G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Ci = reinterpret_cast<bf16_t*>(smem);          // [NL][IMG]; dCm after mid is done
  bf16_t* X1i = Ci + NL * G::IMG;                        // ... c2q
  bf16_t* X2i = X1i + NL * G::IMG;                       // ... q2c
  bf16_t* Qi = X2i + NL * G::IMG;                        // [32][IMG]
  bf16_t* midI = Qi + 32 * G::IMG;
  bf16_t* dmidI = midI + 32 * G::IMG;
  bf16_t* P1i = dmidI + 32 * G::IMG;                     // [NL][PLD]: rows = context, columns = query
  bf16_t* P2i = P1i + NL * G::PLD;
  const int tid = threadIdx.x, lane = tid & 63, w = WAVE_ID();
  const int b = blockIdx.y, slice = blockIdx.x, D = a.D, Lc = a.Lc, Lq = a.Lq;
  const bf16_t* Cb = a.C + (int64_t)b * Lc * D + slice * G::DSV;
  const bf16_t* Qb = a.Q + (int64_t)b * Lq * D + slice * G::DSV;
  const bf16_t* gb = a.g + (int64_t)b * Lc * 4 * D + slice * G::DSV;
  const int SP = (Lq + 7) / 8 * 8;
  stage_ctx<E, NL, NL>(Cb, gb, D, Lc, Ci, X1i, X2i, tid);
  stage_rows<NL, 32>(Qb, D, Lq, Qi, tid);
  stage_probs<E, NL>(a.A1 + (int64_t)b * Lc * SP, Lc, SP, P1i, tid);
  stage_probs<E, NL>(a.A2 + (int64_t)b * Lc * SP, Lc, SP, P2i, tid);
  __syncthreads();
  {
    f32x4 mid[2][G::NN], dq[2][G::NN], dmid[2][G::NN];
    zero2n(mid); zero2n(dq); zero2n(dmid);
    prod_short_rows<E, NL>(P2i, Ci, mid, w, lane);
    prod_short_rows<E, NL>(P1i, X1i, dq, w, lane);
    prod_short_rows<E, NL>(P1i, X2i, dmid, w, lane);
    put_short_img<E, NL>(mid, midI, w, lane);
    put_short_img<E, NL>(dmid, dmidI, w, lane);
    put_short_global<E, NL>(dq, a.dQ + (int64_t)b * Lq * D + slice * G::DSV, D, Lq, w, lane);
  }
  __syncthreads();                                       // mid / dmid complete
  mfma_phase_img<E, NL, true>(a, b, slice, Qi, midI, dmidI, Ci, X1i, X2i, lane, w);     // dS_ / dS_t partials, all operands in LDS
  __syncthreads();                                       // Ci, X1i, X2i free
  prod_long_rows<E, NL>(P1i, Qi, X1i, w, lane);          // c2q
  prod_long_rows<E, NL>(P1i, midI, X2i, w, lane);        // q2c
  prod_long_rows<E, NL>(P2i, dmidI, Ci, w, lane);        // dC through mid
  __syncthreads();
  bf16_t* dCb = a.dC + (int64_t)b * Lc * D + slice * G::DSV;
  constexpr int CPR = G::DSV / 8;
#pragma unroll
  for (int j = 0; j < NL * CPR / 256; ++j) {
    const int i = tid + 256 * j, row = i / CPR, ch = (i % CPR) * 8;
    if (row < Lc) {
      const bf16_t* gr = gb + (int64_t)row * 4 * D + ch;
      const bf16x8 g1 = ldfrag(gr), g3 = ldfrag(gr + 2 * D), g4 = ldfrag(gr + 3 * D);
      const bf16x8 x = *reinterpret_cast<const bf16x8*>(X1i + row * G::IMG + ch), y = *reinterpret_cast<const bf16x8*>(X2i + row * G::IMG + ch);
      const bf16x8 z = *reinterpret_cast<const bf16x8*>(Ci + row * G::IMG + ch);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        o[e] = bits_from_f<E>(frag_get<E>(g1, e) + frag_get<E>(g3, e) * frag_get<E>(x, e) + frag_get<E>(g4, e) * frag_get<E>(y, e) + frag_get<E>(z, e));
      *reinterpret_cast<bf16x8*>(dCb + (int64_t)row * D + ch) = o;
    }
  }
}

// =====================================================================================================================
// both softmax backwards on the summed per-slice partials, everything long-major [Ll][SP] (l = long index, s = short):
//   context long  (c = l, q = s): dS = S_ (dS_ - rowdot_l(dS_ S_)) + S_t (dS_t - coldot_s(dS_t S_t))
//   context short (c = s, q = l): dS = S_ (dS_ - coldot_s(dS_ S_)) + S_t (dS_t - rowdot_l(dS_t S_t))
//   dterm[s] = sum_l dS[l][s]   (the gradient of the rank-1 term that rides on the short stream)
// =====================================================================================================================
// parts[b][0][...] += sum_{p >= 1} parts[b][p][...] over ALL CUs: with one workgroup per clip the softmax-backward kernel
// below pulled the 8 per-slice partial tiles (262 KB per clip, 16.8 MB) through 64 CUs at the per-CU HBM fetch rate
// (~25 GB/s): 22 us of which ~10 were this sum
__global__ __launch_bounds__(256) void cq_parts_presum_kernel(float* __restrict__ parts, int nparts, int64_t per_part4,
                                                              int64_t total4) {
  f32x4* p4 = reinterpret_cast<f32x4*>(parts);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / per_part4, j = i - b * per_part4;
    f32x4* base = p4 + b * nparts * per_part4 + j;
    f32x4 v[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) v[p] = base[(int64_t)min(p, nparts - 1) * per_part4];
    f32x4 acc = v[0];
#pragma unroll
    for (int p = 1; p < 8; ++p)
      if (p < nparts) { acc[0] += v[p][0]; acc[1] += v[p][1]; acc[2] += v[p][2]; acc[3] += v[p][3]; }
    for (int p = 8; p < nparts; ++p) {
      const f32x4 w = base[(int64_t)p * per_part4];
      acc[0] += w[0]; acc[1] += w[1]; acc[2] += w[2]; acc[3] += w[3];
    }
    base[0] = acc;
  }
}

__global__ __launch_bounds__(1024) void cq_softmax_bwd_parts_kernel(const float* __restrict__ parts, int nparts, int nsum,
                                                                   const float* __restrict__ A1, const float* __restrict__ A2,
                                                                   float* __restrict__ dS, float* __restrict__ dterm, int Ll,
                                                                   int Ls, int SP, int LcP, int LqP, int ctx_long) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = Ll * SP;
  float* g1 = reinterpret_cast<float*>(smem);          // dS_ (long-major), later dS
  float* g2 = g1 + n;                                  // dS_t
  float* p1 = g2 + n;                                  // S_
  float* p2 = p1 + n;                                  // S_t
  float* cdot = p2 + n;                                // [SP]
  float* stripe = cdot + SP;                           // [1024]
  const int NT = blockDim.x, NW = NT >> 6;             // 1024 threads: the partial sums are latency-bound, 12 -> 3 rounds
  const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* pb = parts + (int64_t)b * nparts * 2 * LcP * LqP;
  const int64_t tile = (int64_t)LcP * LqP;
  // element (l, s) of the partial tiles: (c, q) = (l, s) or (s, l)
  for (int i = threadIdx.x; i < n; i += NT) {
    const int l = i / SP, s = i - l * SP;
    float s1 = 0.f, s2 = 0.f;
    if (s < Ls) {
      const float* pp = pb + (ctx_long ? (int64_t)l * LqP + s : (int64_t)s * LqP + l);
#pragma unroll 8
      for (int p = 0; p < nsum; ++p) { s1 += pp[(int64_t)p * 2 * tile]; s2 += pp[(int64_t)p * 2 * tile + tile]; }
    }
    g1[i] = s1; g2[i] = s2;
    p1[i] = A1[(int64_t)b * n + i];
    p2[i] = A2[(int64_t)b * n + i];
  }
  __syncthreads();
  // column dots (over l, per s) of the matrix whose softmax ran over the long index: 256 / SP stripes of rows
  const float* gcol = ctx_long ? g2 : g1;
  const float* pcol = ctx_long ? p2 : p1;
  const int NS = NT / SP, ps = threadIdx.x % SP, pst = threadIdx.x / SP;
  {
    float d = 0.f;
    if (pst < NS)
      for (int l = pst; l < Ll; l += NS) d += gcol[l * SP + ps] * pcol[l * SP + ps];
    stripe[threadIdx.x] = d;
    __syncthreads();
    if (threadIdx.x < SP) {
      float t = 0.f;
      for (int k = 0; k < NS; ++k) t += stripe[k * SP + threadIdx.x];
      cdot[threadIdx.x] = t;
    }
  }
  __syncthreads();
  float* grow = ctx_long ? g1 : g2;
  const float* prow = ctx_long ? p1 : p2;
  for (int l = w; l < Ll; l += NW) {                   // row dots (over s, per l) + the combination
    float d = lane < SP ? grow[l * SP + lane] * prow[l * SP + lane] : 0.f;
    d = wave_sum(d);
    if (lane < SP) {
      const int i = l * SP + lane;
      const float g = prow[i] * (grow[i] - d) + pcol[i] * (gcol[i] - cdot[lane]);
      dS[(int64_t)b * n + i] = g;                      // (padding columns: all probabilities are zero there)
      grow[i] = g;                                     // (a lane only ever re-reads its own element)
    }
  }
  __syncthreads();
  {
    float t = 0.f;
    if (pst < NS)
      for (int l = pst; l < Ll; l += NS) t += grow[l * SP + ps];
    stripe[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x < Ls) {
      float u = 0.f;
      for (int k = 0; k < NS; ++k) u += stripe[k * SP + threadIdx.x];
      dterm[(int64_t)b * Ls + threadIdx.x] = u;
    }
  }
}

// =====================================================================================================================
// score backward products: S2[l,s] = long[l,:] . short[s,:]  =>  dlong[l,:] = sum_s dS[l,s] short[s,:],
//                                                                 dshort[s,:] = sum_l dS[l,s] long[l,:]
// dS: fp32 long-major [B, Ll, SP] (scalar-loaded rows)
// =====================================================================================================================
struct ScoreBwdArgs {
  const bf16_t* lng; const bf16_t* sht; const float* dS; bf16_t* dlng; bf16_t* dsht;
  int Ll, Ls, D;
};

template <typename E, int SP>
__global__ __launch_bounds__(256) void cq_score_bwd_kernel(ScoreBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);          // [4][SP][DS]
  const int lane = threadIdx.x & 63, w = WAVE_ID();
  const int b = blockIdx.y, d0 = blockIdx.x * DS + 2 * lane, D = a.D;
  const float* dSb = a.dS + (int64_t)b * a.Ll * SP;
  const bf16_t* Lb = a.lng + (int64_t)b * a.Ll * D + d0;
  bf16_t* dLb = a.dlng + (int64_t)b * a.Ll * D + d0;
  f32x2 Sreg[SP], acc[SP];
  ld_short<E, SP>(a.sht + (int64_t)b * a.Ls * D + d0, D, a.Ls, Sreg);
#pragma unroll
  for (int s = 0; s < SP; ++s) acc[s][0] = acc[s][1] = 0.f;
  for (int r0 = w * NB; r0 < a.Ll; r0 += 4 * NB) {
    uint32_t lr[NB];
    float pd[NB];
    ld_rows(Lb, D, r0, a.Ll - 1, lr);
    ld_probs<SP>(dSb, r0, a.Ll - 1, lane, pd);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      ROW_FENCE();
      const int r = r0 + j;
      const float ok = r < a.Ll ? 1.f : 0.f;
      float x0, x1;
      dot_rows<SP>(pd[j], Sreg, x0, x1);                // d(long)[r] = sum_s dS[r,s] short[s]
      if (r < a.Ll) st2<E>(dLb + (int64_t)r * D, x0, x1);
      axpy_rows<SP>(pd[j], ok * lo16<E>(lr[j]), ok * hi16<E>(lr[j]), acc);   // d(short)[s] += dS[r,s] long[r]
    }
  }
  red_put<SP>(red, w, acc, lane);
  __syncthreads();
  bf16_t* dSh = a.dsht + (int64_t)b * a.Ls * D + d0;
  for (int s = w; s < a.Ls; s += 4) {
    float x0, x1;
    red_row<SP>(red, s, lane, x0, x1);
    st2<E>(dSh + (int64_t)s * D, x0, x1);
  }
}

// The same on MFMA (bf16, Ll <= 128, Ls <= 32): dS rounded to the element type as a [long][short] image,
//   d(long)  = dS . short      (K = 32)  -> staged image, 16-byte stores
//   d(short) = dS^T . long     (K = 128) -> registers, 2-byte stores (32 rows)
template <typename E, int NL>
__global__ __launch_bounds__(256) void cq_score_bwd_mfma(ScoreBwdArgs a) {
  typedef Geo<NL> G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Li = reinterpret_cast<bf16_t*>(smem);          // [NL][IMG]
  bf16_t* Si = Li + NL * G::IMG;                         // [32][IMG]
  bf16_t* Pi = Si + 32 * G::IMG;                         // [NL][PLD]
  const int tid = threadIdx.x, lane = tid & 63, w = WAVE_ID();
  const int b = blockIdx.y, slice = blockIdx.x, D = a.D, Ll = a.Ll, Ls = a.Ls;
  const int SP = (Ls + 7) / 8 * 8;
  stage_rows<NL, NL>(a.lng + (int64_t)b * Ll * D + slice * G::DSV, D, Ll, Li, tid);
  stage_rows<NL, 32>(a.sht + (int64_t)b * Ls * D + slice * G::DSV, D, Ls, Si, tid);
  stage_probs<E, NL>(a.dS + (int64_t)b * Ll * SP, Ll, SP, Pi, tid);
  __syncthreads();
  prod_long_rows<E, NL, true>(Pi, Si, a.dlng + (int64_t)b * Ll * D + slice * G::DSV, w, lane, D, Ll);
  f32x4 ds[2][G::NN];
  zero2n(ds);
  prod_short_rows<E, NL>(Pi, Li, ds, w, lane);
  put_short_global<E, NL>(ds, a.dsht + (int64_t)b * Ls * D + slice * G::DSV, D, Ls, w, lane);
}

inline int sp_of(int n) { return n <= 8 ? 8 : (n <= 16 ? 16 : (n <= 24 ? 24 : 32)); }

// host side of Geo<NL>: the MFMA kernels' long-image rows for a (context, query) length pair, 0 = not theirs
inline int cq_mfma_nl(int Lc, int Lq) {
  const int lo = Lc < Lq ? Lc : Lq, hi = Lc < Lq ? Lq : Lc;
  if (lo > 32 || hi > 256) return 0;
  return hi > 128 ? 256 : 128;
}
template <int NL> constexpr size_t cq_img_bytes(int nlong, int nshort) {      // nlong long images + nshort 32-row images + two P images
  return (size_t)(nlong * NL + nshort * 32) * Geo<NL>::IMG * 2 + (size_t)2 * NL * Geo<NL>::PLD * 2;
}

int cq_mfma_on() {      // A/B: VMR_CQ_MFMA=0 keeps the register kernels
  static int on = -1;
  if (on < 0) { const char* e = getenv("VMR_CQ_MFMA"); on = e ? atoi(e) : 1; }
  return on;
}

int set_lds(const void* fn, size_t bytes, const char* what) {
  if (bytes <= 64 * 1024) return 0;
  // once per kernel and size (the attribute call is tens of microseconds of host time: it showed in eager loops)
  static thread_local const void* done_fn[64];
  static thread_local size_t done_bytes[64];
  static thread_local int ndone = 0;
  for (int i = 0; i < ndone; ++i)
    if (done_fn[i] == fn && done_bytes[i] >= bytes) return 0;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return vmr_fail(-5, "%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
  if (ndone < 64) { done_fn[ndone] = fn; done_bytes[ndone] = bytes; ++ndone; }
  return 0;
}

}  // namespace

// the fused apply kernels take: a 16-bit element type, D a multiple of 128, one stream <= 32 rows, the other <= 128 (256 on MFMA)
extern "C" int vmr_cq_apply_supported(int Lc, int Lq, int D, int dtype) {
  if (!vmr_dtype_16(dtype) || D % DS != 0 || Lc < 1 || Lq < 1) return 0;
  const int shorter = Lc < Lq ? Lc : Lq, longer = Lc < Lq ? Lq : Lc;
  // (more than 128 long rows: the MFMA kernels' 64-channel-slice form only -- the register backward kernels' LDS images
  //  of the long stream stop fitting near 200 rows)
  return shorter <= 32 && (longer <= 128 || (longer <= 256 && cq_mfma_on()));
}

#define CQ_DISPATCH_SP1(E, SPV, KERNEL, ...)                                                          \
  switch (SPV) {                                                                                       \
    case 8: { auto fn = KERNEL<E, 8>; __VA_ARGS__ } break;                                             \
    case 16: { auto fn = KERNEL<E, 16>; __VA_ARGS__ } break;                                           \
    case 24: { auto fn = KERNEL<E, 24>; __VA_ARGS__ } break;                                           \
    default: { auto fn = KERNEL<E, 32>; __VA_ARGS__ } break;                                           \
  }
#define CQ_DISPATCH_SP(SPV, KERNEL, ...)                                                              \
  if (dtype == VMR_F16) { CQ_DISPATCH_SP1(f16_t, SPV, KERNEL, __VA_ARGS__) }                           \
  else { CQ_DISPATCH_SP1(bf16_t, SPV, KERNEL, __VA_ARGS__) }

extern "C" int vmr_cq_apply_fwd(const void* ctx, const void* qry, const float* S_lm, const float* St_lm, void* out, int B, int Lc,
                                int Lq, int D, int dtype, void* stream) {
  VMR_CHECK(ctx && qry && S_lm && St_lm && out, "vmr_cq_apply_fwd: null pointer");
  VMR_CHECK(vmr_cq_apply_supported(Lc, Lq, D, dtype), "vmr_cq_apply_fwd: unsupported shape Lc=%d Lq=%d D=%d", Lc, Lq, D);
  VMR_CHECK((((uintptr_t)S_lm | (uintptr_t)St_lm | (uintptr_t)ctx | (uintptr_t)qry | (uintptr_t)out) & 15) == 0,
            "vmr_cq_apply_fwd: 16-byte alignment");
  if (B == 0) return 0;
  ApplyArgs a;
  memset(&a, 0, sizeof(a));
  a.C = (const bf16_t*)ctx; a.Q = (const bf16_t*)qry; a.A1 = S_lm; a.A2 = St_lm; a.out = (bf16_t*)out;
  a.Lc = Lc; a.Lq = Lq; a.D = D;
  const dim3 grid(D / DS, B);
  static int fwd_env = -1;      // VMR_CQ_MFMA_FWD: bit 0 = context-short forward, bit 1 = context-long forward on MFMA
  if (fwd_env < 0) { const char* e = getenv("VMR_CQ_MFMA_FWD"); fwd_env = e ? atoi(e) : 3; }
  const int shortc = Lq > Lc;
  const int NLv = cq_mfma_nl(Lc, Lq);
  if (cq_mfma_on() && NLv && (fwd_env & (shortc ? 1 : 2))) {
#define CQ_FWD_MFMA(KERNEL, E, NL)                                                            \
    do {                                                                                      \
      const size_t ldsm = shortc ? cq_img_bytes<NL>(2, 3) : cq_img_bytes<NL>(3, 2);           \
      const dim3 gridm(D / Geo<NL>::DSV, B);                                                  \
      auto fn = KERNEL<E, NL>;                                                                \
      if (int rc = set_lds((const void*)fn, ldsm, "vmr_cq_apply_fwd")) return rc;             \
      hipLaunchKernelGGL(fn, gridm, dim3(256), ldsm, (hipStream_t)stream, a);                 \
    } while (0)
#define CQ_FWD_MFMA_E(KERNEL, NL) do { if (dtype == VMR_F16) CQ_FWD_MFMA(KERNEL, f16_t, NL); else CQ_FWD_MFMA(KERNEL, bf16_t, NL); } while (0)
    if (shortc) { if (NLv == 256) CQ_FWD_MFMA_E(cq_apply_fwd_cshort_mfma, 256); else CQ_FWD_MFMA_E(cq_apply_fwd_cshort_mfma, 128); }
    else { if (NLv == 256) CQ_FWD_MFMA_E(cq_apply_fwd_clong_mfma, 256); else CQ_FWD_MFMA_E(cq_apply_fwd_clong_mfma, 128); }
#undef CQ_FWD_MFMA_E
#undef CQ_FWD_MFMA
    VMR_LAUNCH_CHECK();
    return 0;
  }
  if (Lq <= Lc) {
    const int SP = sp_of(Lq);
    const size_t lds = (size_t)4 * SP * DS * 4;
    CQ_DISPATCH_SP(SP, cq_apply_fwd_clong, if (int rc = set_lds((const void*)fn, lds, "vmr_cq_apply_fwd")) return rc;
                   hipLaunchKernelGGL(fn, grid, dim3(256), lds, (hipStream_t)stream, a);)
  } else {
    const int SP = sp_of(Lc);
    const size_t lds = (size_t)8 * SP * DS * 4;
    CQ_DISPATCH_SP(SP, cq_apply_fwd_cshort, if (int rc = set_lds((const void*)fn, lds, "vmr_cq_apply_fwd")) return rc;
                   hipLaunchKernelGGL(fn, grid, dim3(256), lds, (hipStream_t)stream, a);)
  }
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_cq_apply_bwd(const void* dcat4, const void* ctx, const void* qry, const float* S_lm, const float* St_lm,
                                void* dctx, void* dqry, float* parts, int B, int Lc, int Lq, int D, int dtype, void* stream) {
  VMR_CHECK(dcat4 && ctx && qry && S_lm && St_lm && dctx && dqry && parts, "vmr_cq_apply_bwd: null pointer");
  VMR_CHECK(vmr_cq_apply_supported(Lc, Lq, D, dtype), "vmr_cq_apply_bwd: unsupported shape Lc=%d Lq=%d D=%d", Lc, Lq, D);
  VMR_CHECK((((uintptr_t)dcat4 | (uintptr_t)ctx | (uintptr_t)qry | (uintptr_t)parts | (uintptr_t)S_lm | (uintptr_t)St_lm) & 15) == 0,
            "vmr_cq_apply_bwd: 16-byte alignment");
  if (B == 0) return 0;
  ApplyArgs a;
  memset(&a, 0, sizeof(a));
  a.C = (const bf16_t*)ctx; a.Q = (const bf16_t*)qry; a.A1 = S_lm; a.A2 = St_lm; a.g = (const bf16_t*)dcat4;
  a.dC = (bf16_t*)dctx; a.dQ = (bf16_t*)dqry; a.parts = parts;
  a.Lc = Lc; a.Lq = Lq; a.D = D;
  a.LcP = (Lc + 15) / 16 * 16; a.LqP = (Lq + 15) / 16 * 16;
  const dim3 grid(D / DS, B);
  const size_t img = (size_t)2 * a.LqP * IMG_LD * 2;
  if (Lq <= Lc) {
    const int SP = sp_of(Lq);
    if (cq_mfma_on() && cq_mfma_nl(Lc, Lq)) {
#define CQ_BWD_MFMA(KERNEL, E, NL, NLONG, NSHORT)                                             \
      do {                                                                                    \
        const size_t ldsm = cq_img_bytes<NL>(NLONG, NSHORT);                                  \
        const dim3 gridm(D / Geo<NL>::DSV, B);                                                \
        auto fn = KERNEL<E, NL>;                                                              \
        if (int rc = set_lds((const void*)fn, ldsm, "vmr_cq_apply_bwd")) return rc;           \
        hipLaunchKernelGGL(fn, gridm, dim3(256), ldsm, (hipStream_t)stream, a);               \
      } while (0)
#define CQ_BWD_MFMA_E(KERNEL, NL, NLONG, NSHORT) do { if (dtype == VMR_F16) CQ_BWD_MFMA(KERNEL, f16_t, NL, NLONG, NSHORT); else CQ_BWD_MFMA(KERNEL, bf16_t, NL, NLONG, NSHORT); } while (0)
      if (cq_mfma_nl(Lc, Lq) == 256) CQ_BWD_MFMA_E(cq_apply_bwd_clong_mfma, 256, 3, 3); else CQ_BWD_MFMA_E(cq_apply_bwd_clong_mfma, 128, 3, 3);
      VMR_LAUNCH_CHECK();
      return 0;
    }
    const size_t lds = (size_t)8 * SP * DS * 4 + img;
    VMR_CHECK(lds <= 160 * 1024, "vmr_cq_apply_bwd: does not fit LDS (%zu B)", lds);
    CQ_DISPATCH_SP(SP, cq_apply_bwd_clong, if (int rc = set_lds((const void*)fn, lds, "vmr_cq_apply_bwd")) return rc;
                   hipLaunchKernelGGL(fn, grid, dim3(256), lds, (hipStream_t)stream, a);)
  } else {
    const int SP = sp_of(Lc);
    if (cq_mfma_on() && cq_mfma_nl(Lc, Lq)) {
      if (cq_mfma_nl(Lc, Lq) == 256) CQ_BWD_MFMA_E(cq_apply_bwd_cshort_mfma, 256, 2, 3); else CQ_BWD_MFMA_E(cq_apply_bwd_cshort_mfma, 128, 2, 3);
      VMR_LAUNCH_CHECK();
      return 0;
    }
    const size_t lds = (size_t)4 * SP * DS * 4 + img;
    VMR_CHECK(lds <= 160 * 1024, "vmr_cq_apply_bwd: does not fit LDS (%zu B)", lds);
    CQ_DISPATCH_SP(SP, cq_apply_bwd_cshort, if (int rc = set_lds((const void*)fn, lds, "vmr_cq_apply_bwd")) return rc;
                   hipLaunchKernelGGL(fn, grid, dim3(256), lds, (hipStream_t)stream, a);)
  }
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_cq_softmax_bwd_parts(float* parts, const float* S_lm, const float* St_lm, float* dS_lm, float* dterm,
                                        int B, int Lc, int Lq, int D, void* stream) {
  VMR_CHECK(parts && S_lm && St_lm && dS_lm && dterm, "vmr_cq_softmax_bwd_parts: null pointer");
  VMR_CHECK(D % DS == 0 && Lc >= 1 && Lq >= 1 && (Lc <= 32 || Lq <= 32), "vmr_cq_softmax_bwd_parts: bad dims");
  if (B == 0) return 0;
  const int ctx_long = Lq <= Lc ? 1 : 0;
  const int Ll = ctx_long ? Lc : Lq, Ls = ctx_long ? Lq : Lc, SP = sp_of(Ls);
  const size_t lds = ((size_t)4 * Ll * SP + SP + 1024) * 4;
  VMR_CHECK(lds <= 160 * 1024, "vmr_cq_softmax_bwd_parts: score tile %dx%d does not fit LDS", Ll, SP);
  if (int rc = set_lds((const void*)cq_softmax_bwd_parts_kernel, lds, "vmr_cq_softmax_bwd_parts")) return rc;
  // (slices written by vmr_cq_apply_bwd: the MFMA kernels take 64-channel slices when the longer stream exceeds 128 rows)
  const int LcP = (Lc + 15) / 16 * 16, LqP = (Lq + 15) / 16 * 16;
  const int nparts = (cq_mfma_on() && cq_mfma_nl(Lc, Lq) == 256) ? D / 64 : D / DS;
  int nsum = nparts;
  if (nparts > 1 && ((uintptr_t)parts & 15) == 0) {   // (parts is scratch: the slice sum is left in slice 0)
    const int64_t per4 = (int64_t)2 * LcP * LqP / 4, tot4 = per4 * B;
    hipLaunchKernelGGL(cq_parts_presum_kernel, dim3((unsigned)min((int64_t)2048, (tot4 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, parts, nparts, per4, tot4);
    nsum = 1;
  }
  hipLaunchKernelGGL(cq_softmax_bwd_parts_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, parts, nparts, nsum, S_lm, St_lm,
                     dS_lm, dterm, Ll, Ls, SP, LcP, LqP, ctx_long);
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_cq_score_bwd(const void* lng, const void* sht, const float* dS_lm, void* dlng, void* dsht, int B, int Ll, int Ls,
                                int D, int dtype, void* stream) {
  VMR_CHECK(lng && sht && dS_lm && dlng && dsht, "vmr_cq_score_bwd: null pointer");
  VMR_CHECK(vmr_dtype_16(dtype) && D % DS == 0 && Ls >= 1 && Ls <= 32 && Ll >= 1 && ((uintptr_t)dS_lm & 15) == 0,
            "vmr_cq_score_bwd: unsupported shape Ll=%d Ls=%d D=%d", Ll, Ls, D);
  if (B == 0) return 0;
  ScoreBwdArgs a;
  a.lng = (const bf16_t*)lng; a.sht = (const bf16_t*)sht; a.dS = dS_lm; a.dlng = (bf16_t*)dlng; a.dsht = (bf16_t*)dsht;
  a.Ll = Ll; a.Ls = Ls; a.D = D;
  const dim3 grid(D / DS, B);
  const int SP = sp_of(Ls);
  if (cq_mfma_on() && dtype == VMR_BF16 && Ll <= 256) {      // (fp16 keeps dS in fp32: a scaled gradient may not fit the element type)
#define CQ_SB_MFMA(NL)                                                                                          \
    do {                                                                                                        \
      const size_t ldsm = (size_t)(NL + 32) * Geo<NL>::IMG * 2 + (size_t)NL * Geo<NL>::PLD * 2;                 \
      const dim3 gridm(D / Geo<NL>::DSV, B);                                                                    \
      auto fn = cq_score_bwd_mfma<bf16_t, NL>;                                                                  \
      if (int rc = set_lds((const void*)fn, ldsm, "vmr_cq_score_bwd")) return rc;                               \
      hipLaunchKernelGGL(fn, gridm, dim3(256), ldsm, (hipStream_t)stream, a);                                   \
    } while (0)
    if (Ll > 128) CQ_SB_MFMA(256); else CQ_SB_MFMA(128);
#undef CQ_SB_MFMA
    VMR_LAUNCH_CHECK();
    return 0;
  }
  const size_t lds = (size_t)4 * SP * DS * 4;
  CQ_DISPATCH_SP(SP, cq_score_bwd_kernel, if (int rc = set_lds((const void*)fn, lds, "vmr_cq_score_bwd")) return rc;
                 hipLaunchKernelGGL(fn, grid, dim3(256), lds, (hipStream_t)stream, a);)
  VMR_LAUNCH_CHECK();
  return 0;
}
