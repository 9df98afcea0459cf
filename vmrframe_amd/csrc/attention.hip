// attention.hip -- fused attention forward for gfx950: scores (MFMA) + mask + softmax + dropout
// + context gather (MFMA) in ONE kernel, for the three attention forms of the SeqPAN path:
//   * DualMultiAttention self / cross attention (reference models/layers.py:346-367): mode 0,
//     z = (b,h), term = (1 - rmask[b,q] * cmask[b,k]) * -1e30;
//   * TopSelfAttention2 (layers.py:567-574, attention over the BATCH axis per time step): mode 1,
//     z = (t,h), term = cmask[k*cm_stride + t] (a float key-padding mask is ADDED).
// Workgroup = (z, query tile of 32/64/128), one wave per 16 queries.  K and V of the (z) slice go
// global -> LDS by DMA (global_load_lds, <= 64 KiB each: Lk <= 128 keys x hd <= 256; V lands while
// the scores are computed); each wave keeps its 16 x hd query fragments in registers.  Scores are computed
// TRANSPOSED (S^T = K.Q^T) so a lane owns 4 consecutive keys of one query: the key-axis softmax
// is an in-lane reduction plus two cross-lane steps, and P goes to the per-wave LDS image with
// 8-byte writes.  The context is computed as O^T = V^T.P^T (V through ds_read_b64_tr_b16, P as the
// k-contiguous operand) so a lane owns 4 consecutive output channels: 8-byte global stores.
// P (dropped-out) and the pre-dropout probabilities are also written out for the backward pass.
// bf16 only (fp32 / odd shapes use the batched-GEMM + softmax kernels).
#include "common.h"

namespace {

struct AttnArgs {
  const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O;
  bf16_t* P; bf16_t* Pk;                       // [Z, Lq, ldP]; Pk may be null (no dropout)
  int64_t q_s1, q_s2, q_row, k_s1, k_s2, k_row, v_s1, v_s2, v_row, o_s1, o_s2, o_row;
  const float* rmask; const float* cmask;
  int mode, H, Z2, Lq, Lk, ldP, cm_stride;
  float scale, drop_p; uint32_t seed; const uint32_t* step;
};

__device__ __forceinline__ int swz3(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

// One (z, query-tile) per workgroup of NW waves x 16 queries.
template <int HD, int LKP, int NW>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(AttnArgs a) {
  constexpr int RB = HD * 2;            // bytes per K / V row
  constexpr int NJ = LKP / 16;          // key tiles
  constexpr int NT = HD / 16;           // output-channel tiles
  constexpr int PRB = LKP * 2;          // bytes per row of the per-wave P image
  constexpr int PM = (PRB / 16 - 1) & 15;  // swizzle mask: stay inside the row
  constexpr int SPR = RB / 16;          // 16-B slots per K / V row (16 or 32)
  constexpr int NBLK = LKP * RB / 1024; // 1-KiB DMA blocks per operand
  static_assert(NBLK % NW == 0, "DMA blocks must divide over the waves");
  static_assert(NW * 16 * PRB <= LKP * RB, "P images alias the K image");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + LKP * RB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* Ps = smem + wid * 16 * PRB;     // aliases Ks: written only after every wave left the S phase
  const int z = blockIdx.x, z1 = z / a.Z2, z2 = z - z1 * a.Z2;
  const int q0 = blockIdx.y * (NW * 16) + wid * 16;
  const bool active = q0 < a.Lq;        // wave-uniform
  const bf16_t* Kg = a.K + z1 * a.k_s1 + z2 * a.k_s2;
  const bf16_t* Vg = a.V + z1 * a.v_s1 + z2 * a.v_s2;
  const bf16_t* Qg = a.Q + z1 * a.q_s1 + z2 * a.q_s2;
  // ---- this wave's query fragments (B operand of S^T = K.Q^T): lane = (query lane&15, k-chunk lane>>4)
  const int qi = min(q0 + (lane & 15), a.Lq - 1);   // clamped; rows >= Lq are never stored
  bf16x8 qf[HD / 32];
#pragma unroll
  for (int ks = 0; ks < HD / 32; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8*>(Qg + (int64_t)qi * a.q_row + ks * 32 + (lane >> 4) * 8);
  // ---- the masks of this lane's query and keys, requested HERE (index-clamped, unconditional; oldest in the vmcnt
  // order, so the counted wait below covers them): as 4 x NJ guarded loads inside the softmax they each got their own
  // "s_waitcnt vmcnt(0)" from hipcc -- up to 32 dependent round trips per lane after the S phase
  const uint32_t seed = vmr_seed(a.seed, a.step);   // (the device step counter: read with the operands, not mid-kernel)
  const int zo = z / a.H;
  const float rmv = a.mode == 0 ? a.rmask[(int64_t)zo * a.Lq + min(q0 + (lane & 15), a.Lq - 1)] : 1.f;
  float cmv[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kc = min(j * 16 + (lane >> 4) * 4 + r, a.Lk - 1);
      cmv[j][r] = a.mode == 0 ? a.cmask[(int64_t)zo * a.Lk + kc] : a.cmask[(int64_t)kc * a.cm_stride + zo];
    }
  // ---- K and V of this (z) slice: global -> LDS DMA, 1 KiB per wave-instruction, images swizzled through
  // the per-lane SOURCE address.  K: k-contiguous rows, 16-B slot = chunk ^ (row&15) within 256 B;
  // V: [key][channel] rows, 32-B slot = channel-tile ^ swz3(key).  Rows >= Lk re-read row Lk-1 (their
  // probabilities are exactly 0).
#pragma unroll
  for (int jj = 0; jj < NBLK / NW; ++jj) {
    const int j = jj * NW + wid;
    const int row = (64 / SPR) * j + lane / SPR, sl = lane % SPR;
    const int c = (sl & ~15) | ((sl & 15) ^ (row & 15));
    const bf16_t* src = Kg + (int64_t)min(row, a.Lk - 1) * a.k_row + c * 8;
    __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(Ks + j * 1024), 16, 0, 0);
  }
#pragma unroll
  for (int jj = 0; jj < NBLK / NW; ++jj) {
    const int j = jj * NW + wid;
    const int row = (64 / SPR) * j + lane / SPR, sl = lane % SPR;
    const int c = (((sl >> 1) ^ swz3(row)) << 1) | (sl & 1);
    const bf16_t* src = Vg + (int64_t)min(row, a.Lk - 1) * a.v_row + c * 8;
    __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(Vs + j * 1024), 16, 0, 0);
  }
  // K (issued before V) has landed when at most the V blocks are still in flight
  if (NBLK / NW == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if (NBLK / NW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (NBLK / NW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (NBLK / NW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // ---- S^T tiles: rows = keys, cols = queries
  f32x4 st[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) st[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (active) {
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row = j * 16 + (lane & 15);
        const int c = ks * 4 + (lane >> 4);
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + row * RB + (((c & ~15) | ((c & 15) ^ (row & 15))) << 4));
        st[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], st[j], 0, 0, 0);
      }
    }
  }
  // every wave is done reading K (its space becomes the P images) and every wave's V blocks have landed
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (!active) return;
  // lane owns query q = q0 + (lane&15) and keys j*16 + (lane>>4)*4 + r
  const int q = q0 + (lane & 15);
  const bool qok = q < a.Lq;
  const float rm = (a.mode == 0 && qok) ? rmv : 1.f;
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = j * 16 + (lane >> 4) * 4 + r;
      float x = -INFINITY;
      if (key < a.Lk) {
        x = st[j][r] * a.scale;
        if (a.mode == 0) x += (1.0f - rm * cmv[j][r]) * VMR_NEG_INF_MASK;
        else x += cmv[j][r];
      }
      st[j][r] = x;
      mx = fmaxf(mx, x);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = __expf(st[j][r] - mx);   // exp(-inf) = 0 for padded keys
      st[j][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
  const uint32_t thresh = vmr_drop_thresh(a.drop_p);
  const float dscale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  // ---- P -> per-wave LDS image [query][key] (k-contiguous rows, 16-B chunk ^= query & PM), 8-byte writes
  const int64_t prow = ((int64_t)z * a.Lq + q);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    float pk[4], pd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = j * 16 + (lane >> 4) * 4 + r;
      pk[r] = st[j][r] * inv;
      pd[r] = pk[r];
      if (a.drop_p > 0.f) pd[r] = vmr_keep(seed, (uint64_t)prow * a.Lk + key, thresh) ? pk[r] * dscale : 0.f;
    }
    const int key0 = j * 16 + (lane >> 4) * 4;
    const int c = key0 >> 3, half = (key0 >> 2) & 1;
    Vec4<bf16_t>::store(reinterpret_cast<bf16_t*>(Ps + (lane & 15) * PRB + ((c ^ ((lane & 15) & PM)) << 4) + half * 8), pd);
    if (qok && key0 < a.ldP) {   // global copies for the backward pass (zero filled up to ldP)
      Vec4<bf16_t>::store(a.P + prow * a.ldP + key0, pd);
      if (a.Pk) Vec4<bf16_t>::store(a.Pk + prow * a.ldP + key0, pk);
    }
  }
  // (only this wave reads its own P image; LDS ops of one wave complete in order)
  // ---- O^T = V^T . P^T : A = V^T via transposed reads of the [key][channel] image, B = P (k-contiguous)
  f32x4 ot[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) ot[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // Every fragment read below is inline asm (common.h lds_read_tr): with the builtin, hipcc put "s_waitcnt vmcnt(0)" in
  // front of the first transposed read -- it cannot see that the V blocks have landed -- which here waited for the P / Pk
  // global stores just issued.  Groups of 8 channel tiles (16 reads, + the P fragment of the k-step): group g+1 is
  // requested before the MFMAs of g, "at most 15 outstanding" retires g, and the registers are pinned behind the wait.
  constexpr int NH8 = NT / 8, NG = (LKP / 32) * NH8;
  static_assert(NT % 8 == 0, "channel tiles come in groups of 8");
  bf16x8 vf[2][8], pfr[2];
  auto issue = [&](int g) {
    const int ks = g / NH8, hh = g % NH8;
    if (hh == 0) {
      const int prow_l = lane & 15, pc = ks * 4 + (lane >> 4);
      pfr[ks & 1] = lds_read_b128_asm(Ps + prow_l * PRB + ((pc ^ (prow_l & PM)) << 4));
    }
    const int gq = lane >> 4, ii = lane & 15, qq = ii >> 2, p = ii & 3;
    const int r = ks * 32 + 8 * gq + qq;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = hh * 8 + i;
      const int a0 = r * RB + ((t ^ swz3(r)) << 5) + p * 8;
      const int a1 = (r + 4) * RB + ((t ^ swz3(r + 4)) << 5) + p * 8;
      union { struct { s16x4 l, h; } s; bf16x8 v; } u;
      u.s.l = lds_read_tr(Vs + a0);
      u.s.h = lds_read_tr(Vs + a1);
      vf[g & 1][i] = u.v;
    }
  };
  issue(0);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) { issue(g + 1); lgkm_wait<15>(); }
    else lgkm_wait<0>();
    const int ks = g / NH8, hh = g % NH8;
#pragma unroll
    for (int i = 0; i < 8; ++i) frag_pin(vf[g & 1][i]);
    if (hh == 0) frag_pin(pfr[ks & 1]);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      ot[hh * 8 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[g & 1][i], pfr[ks & 1], ot[hh * 8 + i], 0, 0, 0);
  }
  if (qok) {
    bf16_t* Og = a.O + z1 * a.o_s1 + z2 * a.o_s2 + (int64_t)q * a.o_row;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float o4[4] = {ot[t][0], ot[t][1], ot[t][2], ot[t][3]};
      Vec4<bf16_t>::store(Og + t * 16 + (lane >> 4) * 4, o4);
    }
  }
}

template <int HD, int LKP, int NW>
int launch_attn_nw(const AttnArgs& a, int Z, hipStream_t st) {
  constexpr int smem = 2 * LKP * HD * 2;
  const void* fn = (const void*)attn_fwd_kernel<HD, LKP, NW>;
  if (smem > 64 * 1024) {
    static thread_local bool done = false;
    if (!done) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      if (e != hipSuccess) return vmr_fail(-5, "vmr_attention_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
      done = true;
    }
  }
  hipLaunchKernelGGL((attn_fwd_kernel<HD, LKP, NW>), dim3(Z, (a.Lq + NW * 16 - 1) / (NW * 16)), dim3(NW * 64), smem, st, a);
  return 0;
}

template <int HD, int LKP>
int launch_attn(const AttnArgs& a, int Z, hipStream_t st) {
  // waves per workgroup: 8 (128 queries) for long query axes so K/V are staged once per slice
  if (a.Lq > 64) return launch_attn_nw<HD, LKP, 8>(a, Z, st);
  if (a.Lq > 32) return launch_attn_nw<HD, LKP, 4>(a, Z, st);
  return launch_attn_nw<HD, LKP, 2>(a, Z, st);
}

}  // namespace

extern "C" int vmr_attention_fwd_supported(int hd, int Lk, int dtype) {
  return dtype == VMR_BF16 && (hd == 128 || hd == 256) && Lk >= 1 && Lk <= 128;
}

extern "C" int vmr_attention_fwd(const void* Q, const void* K, const void* V, void* O, void* P, void* Pk,
                                 const int64_t* strides /*q,k,v,o x (s1,s2,row)*/, const float* rmask,
                                 const float* cmask, int mode, int Z1, int Z2, int H, int Lq, int Lk, int hd, int ldP,
                                 int cm_stride, float scale, int dtype, float drop_p, uint32_t drop_seed,
                                 const uint32_t* drop_step, void* stream) {
  VMR_CHECK(Q && K && V && O && P && strides && cmask, "vmr_attention_fwd: null pointer");
  VMR_CHECK(vmr_attention_fwd_supported(hd, Lk, dtype), "vmr_attention_fwd: unsupported shape hd=%d Lk=%d dtype=%d", hd, Lk,
            dtype);
  VMR_CHECK(mode == 0 || mode == 1, "vmr_attention_fwd: bad mode");
  VMR_CHECK(mode != 0 || rmask, "vmr_attention_fwd: mode 0 needs rmask");
  VMR_CHECK(ldP % 4 == 0 && ldP >= Lk, "vmr_attention_fwd: ldP must be a multiple of 4 and >= Lk");
  for (int i = 0; i < 12; ++i) VMR_CHECK(strides[i] % 8 == 0, "vmr_attention_fwd: strides must be multiples of 8 elements");
  VMR_CHECK((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O | (uintptr_t)P) & 15) == 0, "vmr_attention_fwd: 16-byte alignment");
  const int Z = Z1 * Z2;
  if (Z == 0 || Lq == 0) return 0;
  VMR_CHECK(Z <= 65535 * 32, "vmr_attention_fwd: too many batches");
  AttnArgs a;
  a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V; a.O = (bf16_t*)O;
  a.P = (bf16_t*)P; a.Pk = (bf16_t*)Pk;
  a.q_s1 = strides[0]; a.q_s2 = strides[1]; a.q_row = strides[2];
  a.k_s1 = strides[3]; a.k_s2 = strides[4]; a.k_row = strides[5];
  a.v_s1 = strides[6]; a.v_s2 = strides[7]; a.v_row = strides[8];
  a.o_s1 = strides[9]; a.o_s2 = strides[10]; a.o_row = strides[11];
  a.rmask = rmask; a.cmask = cmask; a.mode = mode; a.H = H; a.Z2 = Z2; a.Lq = Lq; a.Lk = Lk; a.ldP = ldP;
  a.cm_stride = cm_stride; a.scale = scale; a.drop_p = drop_p; a.seed = drop_seed; a.step = drop_step;
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  const int lkp = Lk <= 32 ? 32 : (Lk <= 64 ? 64 : 128);
  if (hd == 256) rc = lkp == 32 ? launch_attn<256, 32>(a, Z, st) : (lkp == 64 ? launch_attn<256, 64>(a, Z, st) : launch_attn<256, 128>(a, Z, st));
  else rc = lkp == 32 ? launch_attn<128, 32>(a, Z, st) : (lkp == 64 ? launch_attn<128, 64>(a, Z, st) : launch_attn<128, 128>(a, Z, st));
  if (rc) return rc;
  VMR_LAUNCH_CHECK();
  return 0;
}
