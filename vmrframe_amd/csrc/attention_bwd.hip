// attention_bwd.hip -- fused attention backward for gfx950: dP = dO.V^T, the softmax / dropout
// backward, dQ = dS.K, dK = dS^T.Q and dV = P^T.dO in ONE kernel per (z1,z2) slice, for the
// attention forms of attention.hip (DualMultiAttention, reference models/layers.py:346-367, and
// TopSelfAttention2, :567-574).  It replaces four batched GEMM launches + vmr_softmax_bwd and the
// fp32 [Z,Lq,Lk] dP round trip through HBM.
//
// Workgroup = one slice, 8 waves.  Wave w owns query rows 16w..16w+15 (dP, dS, dQ) and key rows
// 16w..16w+15 (dK, dV).  The head dimension is processed in halves of 128 channels so every operand
// tile is <= 128 rows x 256 B = 32 KiB; tiles stream through two LDS buffers by global->LDS DMA
// while the previous tile is consumed:
//   phase 1   V[:, half]   (k-contiguous image)  : dP^T += V_h . dO_h^T            (dO from registers)
//             -> dPk = keep*dscale*dP, dS = scale*Pk*(dPk - sum_k dPk*Pk), P = keep*dscale*Pk;
//                P and dS go to two [query][key] LDS images (bf16)
//   phase 2   K[:, half]   (transposed reads)    : dQ_h^T = K_h^T . dS^T           -> global
//             Q[:, half]                         : dK_h^T = Q_h^T . dS             -> global
//             dO[:, half]                        : dV_h^T = dO_h^T . P             -> global
// All products are issued transposed (C^T = B^T.A^T) so a lane owns 4 consecutive channels of one
// output row: 8-byte global stores, and the key-axis reductions of the softmax backward are in-lane
// plus two cross-lane steps.  The dropout keep bits are regenerated from the forward's counter
// stream; Pk is the forward's pre-dropout probability copy.  bf16 only.
#include "common.h"

namespace {

struct AttnBwdArgs {
  const bf16_t* Q; const bf16_t* K; const bf16_t* V; const bf16_t* dO; const bf16_t* Pk;
  bf16_t* dQ; bf16_t* dK; bf16_t* dV;
  int64_t q_s1, q_s2, q_row, k_s1, k_s2, k_row, v_s1, v_s2, v_row, do_s1, do_s2, do_row;
  int64_t dq_s1, dq_s2, dq_row, dk_s1, dk_s2, dk_row, dv_s1, dv_s2, dv_row;
  int Z2, Lq, Lk, ldP, accum_dq;
  float scale, drop_p; uint32_t seed; const uint32_t* step;
};

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
__device__ __forceinline__ int swz3(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

// [ROWS][128 channels] bf16 tile (256-B rows) global -> LDS by DMA, 1 KiB (4 rows) per wave-instruction,
// 8 waves.  TR: image for transposed reads (32-B slot = channel-tile ^ swz3(row)); else k-contiguous
// image (16-B slot = chunk ^ (row & 15)).  Rows >= valid re-read row valid-1 (their P / dS are 0).
template <int ROWS, bool TR>
__device__ __forceinline__ void dma_tile(const bf16_t* __restrict__ base, int64_t row_stride, int valid, int d_off,
                                         unsigned char* lds, int wid, int lane) {
#pragma unroll
  for (int jj = 0; jj < ROWS / 32; ++jj) {
    const int j = jj * 8 + wid;
    const int row = 4 * j + (lane >> 4), sl = lane & 15;
    const int c = TR ? ((((sl >> 1) ^ swz3(row)) << 1) | (sl & 1)) : (sl ^ (row & 15));
    const bf16_t* src = base + (int64_t)min(row, valid - 1) * row_stride + d_off + c * 8;
    __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(lds + j * 1024), 16, 0, 0);
  }
}

// A operand (m = channel tile t of the half, k = rows ks*32..+31) from a TR image: 16 x 32 fragment
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* img, int rb, int sw_mask, int t, int ks, int lane) {
  const int g = lane >> 4, ii = lane & 15, qq = ii >> 2, p = ii & 3;
  const int r = ks * 32 + 8 * g + qq;
  const int a0 = r * rb + ((t ^ (swz3(r) & sw_mask)) << 5) + p * 8;
  const int a1 = (r + 4) * rb + ((t ^ (swz3(r + 4) & sw_mask)) << 5) + p * 8;
  union { struct { s16x4 l, h; } s; bf16x8 v; } u;
  u.s.l = lds_read_tr(img + a0);      // inline asm (common.h): the builtin drains every LDS-DMA tile in flight
  u.s.h = lds_read_tr(img + a1);
  return u.v;
}

// acc[t] += A_t(ks) . B(ks) over NKS 32-deep steps, t = 8 channel tiles.  Every fragment comes from an inline-asm LDS read
// (ld_a / ld_b), so the waits are ours: group ks+1 (>= 17 reads) is requested before the MFMAs of ks, "at most 15
// outstanding" then retires all of group ks (the LDS returns in order), and the registers are pinned behind the wait.
template <int NKS, class FA, class FB>
__device__ __forceinline__ void asm_product(f32x4 (&acc)[8], FA&& ld_a, FB&& ld_b) {
  bf16x8 fa[2][8], fb[2];
#pragma unroll
  for (int t = 0; t < 8; ++t) fa[0][t] = ld_a(t, 0);
  fb[0] = ld_b(0);
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < NKS) {
#pragma unroll
      for (int t = 0; t < 8; ++t) fa[cur ^ 1][t] = ld_a(t, ks + 1);
      fb[cur ^ 1] = ld_b(ks + 1);
      lgkm_wait<15>();
    } else {
      lgkm_wait<0>();
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) frag_pin(fa[cur][t]);
    frag_pin(fb[cur]);
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[cur][t], fb[cur], acc[t], 0, 0, 0);
  }
}

template <int N> __device__ __forceinline__ void wait_vm() {
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else static_assert(N == 0 || N == 1 || N == 2 || N == 3 || N == 4 || N == 9 || N == 10 || N == 12, "add the immediate");
}

template <int HD, int LQP, int LKP>
__global__ __launch_bounds__(512) void attn_bwd_kernel(AttnBwdArgs a) {
  constexpr int NH = HD / 128;                       // channel halves
  constexpr int TROWS = LQP > LKP ? LQP : LKP;
  constexpr int TBYTES = TROWS * 256;                // one operand buffer
  constexpr int NJ = LKP / 16;                       // key tiles
  constexpr int PRB = LKP * 2;                       // image row bytes ([query][key] bf16)
  constexpr int PSW = NJ - 1;                        // swizzle mask of the images (32-B slots per row - 1)
  constexpr int IMG = LQP * PRB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* bufX = smem;
  unsigned char* bufY = smem + TBYTES;
  unsigned char* Pimg = smem + 2 * TBYTES;
  unsigned char* Simg = Pimg + IMG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int z = blockIdx.x, z1 = z / a.Z2, z2 = z - z1 * a.Z2;
  const bf16_t* Qg = a.Q + z1 * a.q_s1 + z2 * a.q_s2;
  const bf16_t* Kg = a.K + z1 * a.k_s1 + z2 * a.k_s2;
  const bf16_t* Vg = a.V + z1 * a.v_s1 + z2 * a.v_s2;
  const bf16_t* Og = a.dO + z1 * a.do_s1 + z2 * a.do_s2;
  const int r0 = wid * 16;                           // this wave's query rows AND key rows
  const bool qact = r0 < a.Lq && r0 < LQP;           // wave-uniform
  const bool kact = r0 < a.Lk && r0 < LKP;

  // ---------------- phase 1: dP^T = V . dO^T over the whole head dimension
  dma_tile<LKP, false>(Vg, a.v_row, a.Lk, 0, bufX, wid, lane);
  if (NH == 2) dma_tile<LKP, false>(Vg, a.v_row, a.Lk, 128, bufY, wid, lane);
  const int qi = min(r0 + (lane & 15), a.Lq - 1);    // clamped query row of this lane (B operand: n = lane&15)
  bf16x8 dof[HD / 32];
#pragma unroll
  for (int ks = 0; ks < HD / 32; ++ks)
    dof[ks] = *reinterpret_cast<const bf16x8*>(Og + (int64_t)qi * a.do_row + ks * 32 + (lane >> 4) * 8);
  // the forward's probabilities of this lane's (query, 4 keys per tile) pairs, requested with the operands (index-
  // clamped, unconditional): as NJ guarded loads inside the softmax backward each got its own "s_waitcnt vmcnt(0)" from
  // hipcc -- NJ dependent round trips in the middle of the kernel, draining the phase-2 tiles just requested as well
  typedef __attribute__((ext_vector_type(2))) uint32_t pk_raw_t;
  pk_raw_t pkraw[NJ];
  {
    const int64_t prow0 = (int64_t)z * a.Lq + qi;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int kc = min(j * 16 + (lane >> 4) * 4, max(a.ldP - 4, 0));
      pkraw[j] = *reinterpret_cast<const pk_raw_t*>(a.Pk + prow0 * a.ldP + kc);
    }
  }
  const uint32_t seed = vmr_seed(a.seed, a.step);   // (the device step counter: read here, not between the phase-2 tiles)
  // the image rows this wave will not write (query rows of inactive waves) must read as zero
  if (!qact && r0 < LQP) {
    for (int i = lane; i < 16 * PRB / 16; i += 64) {
      *reinterpret_cast<u32x4*>(Pimg + r0 * PRB + i * 16) = (u32x4){0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(Simg + r0 * PRB + i * 16) = (u32x4){0u, 0u, 0u, 0u};
    }
  }
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  f32x4 dpt[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) dpt[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (qact) {
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
      const unsigned char* vb = (ks < 4) ? bufX : bufY;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row = j * 16 + (lane & 15);
        const int c = (ks & 3) * 4 + (lane >> 4);
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vb + row * 256 + ((c ^ (row & 15)) << 4));
        dpt[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[ks], dpt[j], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                       // every wave is done with the V tiles
  // tiles of phase 2, in order: K_0, Q_0, dO_0 [, K_1, Q_1, dO_1]; two in flight
  dma_tile<LKP, true>(Kg, a.k_row, a.Lk, 0, bufX, wid, lane);
  dma_tile<LQP, true>(Qg, a.q_row, a.Lq, 0, bufY, wid, lane);

  // ---------------- softmax / dropout backward on this wave's 16 query rows (lane: query lane&15, 4 keys per tile)
  if (qact) {
    const int q = r0 + (lane & 15);
    const bool qok = q < a.Lq;
    const int64_t prow = (int64_t)z * a.Lq + min(q, a.Lq - 1);
    const uint32_t thresh = vmr_drop_thresh(a.drop_p);
    const float dscale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    float pk[NJ][4], kp[NJ][4];
    float rs = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int key0 = j * 16 + (lane >> 4) * 4;
      const float pv[4] = {__uint_as_float(pkraw[j][0] << 16), __uint_as_float(pkraw[j][0] & 0xFFFF0000u),
                           __uint_as_float(pkraw[j][1] << 16), __uint_as_float(pkraw[j][1] & 0xFFFF0000u)};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + r;
        float p = key < a.Lk ? pv[r] : 0.f;
        float m = dscale;
        if (a.drop_p > 0.f) m = vmr_keep(seed, (uint64_t)prow * a.Lk + key, thresh) ? dscale : 0.f;
        pk[j][r] = p;
        kp[j][r] = m;
        rs += dpt[j][r] * m * p;
      }
    }
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float pd[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pd[r] = qok ? pk[j][r] * kp[j][r] : 0.f;
        ds[r] = qok ? a.scale * pk[j][r] * (dpt[j][r] * kp[j][r] - rs) : 0.f;
      }
      const int off = (r0 + (lane & 15)) * PRB + ((j ^ (swz3(r0 + (lane & 15)) & PSW)) << 5) + (lane >> 4) * 8;
      Vec4<bf16_t>::store(reinterpret_cast<bf16_t*>(Pimg + off), pd);
      Vec4<bf16_t>::store(reinterpret_cast<bf16_t*>(Simg + off), ds);
    }
  }

  // ---------------- phase 2: stream the six operand half-tiles through the two buffers
  constexpr int NBK = LKP / 32, NBQ = LQP / 32;      // DMA instructions per wave per tile
  constexpr int NST = 8;                              // output stores per wave per tile (8 channel tiles x 8 B)
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    // ---- tile K_h in bufX: dQ_h^T[d][q] = sum_key K^T[d][key] dS^T[key][q]   (A: K image TR, B: dS image k-contiguous)
    // K_h landed; Q_h (and, from the second half on, the dV stores issued after it) may still be in flight
    if (h > 0 && kact) wait_vm<NBQ + NST>(); else wait_vm<NBQ>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (qact) {
      const int qr = r0 + (lane & 15), half = (lane >> 4) & 1;
      asm_product<LKP / 32>(
          acc, [&](int t, int ks) { return frag_tr(bufX, 256, 7, t, ks, lane); },
          [&](int ks) {
            const int c32 = ks * 2 + (lane >> 5);
            return lds_read_b128_asm(Simg + qr * PRB + ((c32 ^ (swz3(qr) & PSW)) << 5) + half * 16);
          });
    }
    // dQ accumulation (the second attention form of a query stream adds into the first one's dQ): the eight old quads
    // are requested together and BEFORE the next DMA tile (so the wait for them does not drain it); one load -> wait ->
    // add -> store per channel tile was eight dependent round trips per half
    // (read unconditionally -- dQ is allocated either way -- and masked to +0.0 when not accumulating)
    pk_raw_t oldq[8];
    const uint32_t accm = a.accum_dq ? 0xFFFFFFFFu : 0u;
    {
      const int qc = min(r0 + (lane & 15), a.Lq - 1);
      const bf16_t* dqc = a.dQ + z1 * a.dq_s1 + z2 * a.dq_s2 + (int64_t)qc * a.dq_row + h * 128 + (lane >> 4) * 4;
      // (inline-asm loads with our own counted wait: left to hipcc, the wait for these became "vmcnt(0)" -- after the
      //  dO tile below had been requested -- or, with the first use under a branch, a second full drain after the stores)
#pragma unroll
      for (int t = 0; t < 8; ++t)
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(oldq[t]) : "v"(dqc + t * 16) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                     // bufX is free
    dma_tile<LQP, true>(Og, a.do_row, a.Lq, h * 128, bufX, wid, lane);          // dO_h -> bufX
    wait_vm<NBQ>();                                   // the old dQ values (and Q_h before them) are here; dO_h flies
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      asm volatile("" : "+v"(oldq[t]));
      const uint32_t w0 = oldq[t][0] & accm, w1 = oldq[t][1] & accm;
      acc[t][0] += __uint_as_float(w0 << 16); acc[t][1] += __uint_as_float(w0 & 0xFFFF0000u);
      acc[t][2] += __uint_as_float(w1 << 16); acc[t][3] += __uint_as_float(w1 & 0xFFFF0000u);
    }
    if (qact) {
      const int q = r0 + (lane & 15);
      if (q < a.Lq) {
        bf16_t* dq = a.dQ + z1 * a.dq_s1 + z2 * a.dq_s2 + (int64_t)q * a.dq_row + h * 128 + (lane >> 4) * 4;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          float o4[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
          Vec4<bf16_t>::store(dq + t * 16, o4);
        }
      }
    }
    // ---- tile Q_h in bufY: dK_h^T[d][key] = sum_q Q^T[d][q] dS[q][key]       (A: Q image TR, B: dS image TR)
    if (qact) wait_vm<NBQ + NST>(); else wait_vm<NBQ>();   // Q_h landed; dO_h (+ the dQ stores) may fly
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (kact)
      asm_product<LQP / 32>(
          acc, [&](int t, int ks) { return frag_tr(bufY, 256, 7, t, ks, lane); },
          [&](int ks) { return frag_tr(Simg, PRB, PSW, wid, ks, lane); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                     // bufY is free
    if (h + 1 < NH) dma_tile<LKP, true>(Kg, a.k_row, a.Lk, (h + 1) * 128, bufY, wid, lane);   // K_{h+1} -> bufY (swapped below)
    if (kact) {
      const int key = r0 + (lane & 15);
      if (key < a.Lk) {
        bf16_t* dk = a.dK + z1 * a.dk_s1 + z2 * a.dk_s2 + (int64_t)key * a.dk_row + h * 128 + (lane >> 4) * 4;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          float o4[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
          Vec4<bf16_t>::store(dk + t * 16, o4);
        }
      }
    }
    // ---- tile dO_h in bufX: dV_h^T[d][key] = sum_q dO^T[d][q] P[q][key]      (A: dO image TR, B: P image TR)
    if (h + 1 < NH) { if (kact) wait_vm<NBK + NST>(); else wait_vm<NBK>(); }   // dO_h landed; K_{h+1} (+ dK stores) may fly
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (kact)
      asm_product<LQP / 32>(
          acc, [&](int t, int ks) { return frag_tr(bufX, 256, 7, t, ks, lane); },
          [&](int ks) { return frag_tr(Pimg, PRB, PSW, wid, ks, lane); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                     // bufX is free
    if (h + 1 < NH) dma_tile<LQP, true>(Qg, a.q_row, a.Lq, (h + 1) * 128, bufX, wid, lane);   // Q_{h+1} -> bufX
    if (kact) {
      const int key = r0 + (lane & 15);
      if (key < a.Lk) {
        bf16_t* dv = a.dV + z1 * a.dv_s1 + z2 * a.dv_s2 + (int64_t)key * a.dv_row + h * 128 + (lane >> 4) * 4;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          float o4[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
          Vec4<bf16_t>::store(dv + t * 16, o4);
        }
      }
    }
    // the next half finds K in bufY and Q in bufX: swap the roles
    unsigned char* tmp = bufX; bufX = bufY; bufY = tmp;
  }
}

template <int HD, int LQP, int LKP>
int launch_bwd(const AttnBwdArgs& a, int Z, hipStream_t st) {
  constexpr int TROWS = LQP > LKP ? LQP : LKP;
  constexpr int smem = 2 * TROWS * 256 + 2 * LQP * LKP * 2;
  const void* fn = (const void*)attn_bwd_kernel<HD, LQP, LKP>;
  if (smem > 64 * 1024) {
    static thread_local bool done = false;
    if (!done) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      if (e != hipSuccess) return vmr_fail(-5, "vmr_attention_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
      done = true;
    }
  }
  hipLaunchKernelGGL((attn_bwd_kernel<HD, LQP, LKP>), dim3(Z), dim3(512), smem, st, a);
  return 0;
}

template <int HD, int LQP>
int launch_bwd_k(const AttnBwdArgs& a, int Z, hipStream_t st) {
  if (a.Lk <= 32) return launch_bwd<HD, LQP, 32>(a, Z, st);
  if (a.Lk <= 64) return launch_bwd<HD, LQP, 64>(a, Z, st);
  return launch_bwd<HD, LQP, 128>(a, Z, st);
}

template <int HD>
int launch_bwd_q(const AttnBwdArgs& a, int Z, hipStream_t st) {
  if (a.Lq <= 32) return launch_bwd_k<HD, 32>(a, Z, st);
  if (a.Lq <= 64) return launch_bwd_k<HD, 64>(a, Z, st);
  return launch_bwd_k<HD, 128>(a, Z, st);
}

}  // namespace

extern "C" int vmr_attention_bwd_supported(int hd, int Lq, int Lk, int dtype) {
  return dtype == VMR_BF16 && (hd == 128 || hd == 256) && Lq >= 1 && Lq <= 128 && Lk >= 1 && Lk <= 128;
}

extern "C" int vmr_attention_bwd(const void* dO, const void* Q, const void* K, const void* V, const void* Pkeep, void* dQ,
                                 void* dK, void* dV, const int64_t* strides /*q,k,v,dO,dQ,dK,dV x (s1,s2,row)*/, int Z1,
                                 int Z2, int Lq, int Lk, int hd, int ldP, float scale, int accumulate_dq, int dtype,
                                 float drop_p, uint32_t drop_seed, const uint32_t* drop_step, void* stream) {
  VMR_CHECK(dO && Q && K && V && Pkeep && dQ && dK && dV && strides, "vmr_attention_bwd: null pointer");
  VMR_CHECK(vmr_attention_bwd_supported(hd, Lq, Lk, dtype), "vmr_attention_bwd: unsupported shape hd=%d Lq=%d Lk=%d dtype=%d",
            hd, Lq, Lk, dtype);
  VMR_CHECK(ldP % 4 == 0 && ldP >= Lk, "vmr_attention_bwd: ldP must be a multiple of 4 and >= Lk");
  for (int i = 0; i < 21; ++i) VMR_CHECK(strides[i] % 8 == 0, "vmr_attention_bwd: strides must be multiples of 8 elements");
  VMR_CHECK((((uintptr_t)dO | (uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)Pkeep | (uintptr_t)dQ | (uintptr_t)dK |
              (uintptr_t)dV) & 15) == 0, "vmr_attention_bwd: 16-byte alignment");
  const int Z = Z1 * Z2;
  if (Z == 0) return 0;
  AttnBwdArgs a;
  a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V; a.dO = (const bf16_t*)dO; a.Pk = (const bf16_t*)Pkeep;
  a.dQ = (bf16_t*)dQ; a.dK = (bf16_t*)dK; a.dV = (bf16_t*)dV;
  a.q_s1 = strides[0]; a.q_s2 = strides[1]; a.q_row = strides[2];
  a.k_s1 = strides[3]; a.k_s2 = strides[4]; a.k_row = strides[5];
  a.v_s1 = strides[6]; a.v_s2 = strides[7]; a.v_row = strides[8];
  a.do_s1 = strides[9]; a.do_s2 = strides[10]; a.do_row = strides[11];
  a.dq_s1 = strides[12]; a.dq_s2 = strides[13]; a.dq_row = strides[14];
  a.dk_s1 = strides[15]; a.dk_s2 = strides[16]; a.dk_row = strides[17];
  a.dv_s1 = strides[18]; a.dv_s2 = strides[19]; a.dv_row = strides[20];
  a.Z2 = Z2; a.Lq = Lq; a.Lk = Lk; a.ldP = ldP; a.accum_dq = accumulate_dq;
  a.scale = scale; a.drop_p = drop_p; a.seed = drop_seed; a.step = drop_step;
  int rc = hd == 256 ? launch_bwd_q<256>(a, Z, (hipStream_t)stream) : launch_bwd_q<128>(a, Z, (hipStream_t)stream);
  if (rc) return rc;
  VMR_LAUNCH_CHECK();
  return 0;
}
