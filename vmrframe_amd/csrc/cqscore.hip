// cqscore.hip -- the CQAttention score kernel (reference models/layers.py:417-421,427-437): the trilinear
// similarity S (its rank-1 terms folded onto the short stream by the caller) and BOTH masked softmaxes in one launch:
//     M[v, t] = long[b, v, :] . short_op[b, t, :] + shortterm[b, t]
//     P_t = softmax over t of M + (1 - mask_short[t]) * -1e30        P_v = softmax over v of M + (1 - mask_long[v]) * -1e30
// One workgroup (8 waves) per clip.  The SHORT operand ([L <= 32 words, D], e.g. Q*w4mlu + w4C) is staged once in LDS
// by global->LDS DMA (k-contiguous image, XOR-swizzled); every wave streams its 16 rows of the LONG operand (the
// [T <= 128 frames, D] video tensor) straight from HBM as MFMA fragments -- whole 16-byte chunks per lane, each row read
// exactly once -- and keeps its [16 x 32] score tile in registers.  Scores are computed transposed so a lane owns 4
// consecutive t of one v: the softmax over t is in-lane + 2 cross-lane steps; the softmax over v reduces across lanes
// and, through 1 KiB of LDS, across the 8 waves.  The [T, L] score matrix never goes to HBM in fp32.
// Replaces {zero-fill, batched split-K GEMM, cq_softmax} of the composed path.  bf16 only.
#include "common.h"

namespace {

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

struct CqArgs {
  const bf16_t* lng; const bf16_t* sht; const float* shortterm; const float* mask_long; const float* mask_short;
  bf16_t* Srow; bf16_t* Scol;      // nullable: the composed path's bf16 [B, rows, ldP] pair
  float* Pt32; float* Pv32;        // nullable: fp32 long-major [B, Ll, SP32] (softmax over t / over v) for cqapply.hip
  float* cstat;                    // row-split launches: [B, gridDim.y, 32, 2] local column (max, sum of exp) per workgroup
  int Ll, Ls, D, ldP, orient, SP32;
};

// NKS = D / 32 k-steps, compile-time: the fragment pipeline below must be straight-line code.  (With a run-time trip
// count and guarded re-requests hipcc put "s_waitcnt vmcnt(0)" in front of every MFMA pair -- 16 of the 32 k-steps each
// paid a full HBM round trip, and the eight guarded shortterm / mask loads another eight: 18 us for 20 MB.)
// NW waves = NW * 16 long rows per workgroup.  gridDim.y > 1 splits a clip's rows over workgroups (one workgroup per clip
// pulls its 320 KB through ONE CU at the per-CU HBM fetch rate, ~25 GB/s: 64 active CUs = 15 us for 21 MB): the softmax
// over the long index then leaves exp(s - local max) in Pv32 and the local (max, sum) pairs in cstat; cq_colnorm_kernel
// rescales.
// E = the 16-bit element type (bf16_t / f16_t): CqArgs carries its pointers as raw 16-bit (bf16_t-typed) bits
template <typename E, int NKS, int NW>
__global__ __launch_bounds__(NW * 64) void cq_score_kernel(CqArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int D = NKS * 32, RB = D * 2, CPRW = RB / 16;  // bytes / 16-B chunks per short-operand row
  unsigned char* Ss = smem;                                // [32][RB]
  float* red = reinterpret_cast<float*>(smem + 32 * RB);   // [NW waves][32] column partials
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const bf16_t* Lg = a.lng + (int64_t)b * a.Ll * D;
  const bf16_t* Sg = a.sht + (int64_t)b * a.Ls * D;
  // ---- short operand -> LDS (DMA, 1 KiB per wave-instruction; rows >= Ls re-read row Ls-1: masked out below)
  const int nblk = 32 * RB / 1024;
  for (int j = wid; j < nblk; j += NW) {
    const int flat = j * 64 + lane, row = flat / CPRW, sl = flat - row * CPRW;
    const int c = (sl & ~15) | ((sl & 15) ^ (row & 15));
    const bf16_t* src = Sg + (int64_t)min(row, a.Ls - 1) * D + c * 8;
    __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(Ss + j * 1024), 16, 0, 0);
  }
  // ---- this wave's 16 long rows: fragments straight from HBM, 8 k-steps in flight
  const int v0 = ((int)blockIdx.y * NW + wid) * 16;
  const bool act = v0 < a.Ll;
  const int vi = min(v0 + (lane & 15), a.Ll - 1);
  const bf16_t* lrow = Lg + (int64_t)vi * D + (lane >> 4) * 8;
  f32x4 st[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  // the lane's eight (t) scalars of the short stream, requested before the operand stream (index-clamped, unconditional)
  float stv[2][4], msv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tc = min(j * 16 + (lane >> 4) * 4 + r, a.Ls - 1);
      stv[j][r] = a.shortterm[(int64_t)b * a.Ls + tc];
      msv[j][r] = a.mask_short[(int64_t)b * a.Ls + tc];
    }
  const float mlv = a.mask_long[(int64_t)b * a.Ll + vi];
  // up to 32 fragments (32 KiB per wave) in flight: at D <= 1024 the whole row block is requested at once
  constexpr int RG = NKS < 32 ? NKS : 32;
  bf16x8 fr[RG];
#pragma unroll
  for (int u = 0; u < RG; ++u) fr[u] = *reinterpret_cast<const bf16x8*>(lrow + u * 32);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RG) : "memory");   // the DMA blocks and the scalars (issued first) have landed
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int k0 = 0; k0 < NKS; k0 += RG) {
#pragma unroll
    for (int u = 0; u < RG; ++u) {
      const int ks = k0 + u;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = j * 16 + (lane & 15);
        const int c = ks * 4 + (lane >> 4);
        const bf16x8 sf = *reinterpret_cast<const bf16x8*>(Ss + row * RB + (((c & ~15) | ((c & 15) ^ (row & 15))) << 4));
        st[j] = mfma16<E>(sf, fr[u], st[j]);   // C[t][v]
      }
      if (ks + RG < NKS) fr[u] = *reinterpret_cast<const bf16x8*>(lrow + (ks + RG) * 32);
    }
  }
  // lane owns v = v0 + (lane&15) and t = j*16 + (lane>>4)*4 + r
  const int v = v0 + (lane & 15);
  const bool vok = act && v < a.Ll;
  const float ml = vok ? mlv : 0.f;
  float xt[2][4], xv[2][4];
  float mxt = -INFINITY;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = j * 16 + (lane >> 4) * 4 + r;
      float m = -INFINITY, mv = -INFINITY;
      if (t < a.Ls && vok) {
        const float s = st[j][r] + stv[j][r];
        m = s + (1.0f - msv[j][r]) * VMR_NEG_INF_MASK;   // softmax over t
        mv = s + (1.0f - ml) * VMR_NEG_INF_MASK;                                    // softmax over v
      }
      xt[j][r] = m; xv[j][r] = mv;
      mxt = fmaxf(mxt, m);
    }
  // ---- softmax over t (per v): in-lane + lanes 16 / 32 apart
  mxt = fmaxf(mxt, __shfl_xor(mxt, 16, 64));
  mxt = fmaxf(mxt, __shfl_xor(mxt, 32, 64));
  float sumt = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { xt[j][r] = vok ? __expf(xt[j][r] - mxt) : 0.f; sumt += xt[j][r]; }
  sumt += __shfl_xor(sumt, 16, 64);
  sumt += __shfl_xor(sumt, 32, 64);
  const float invt = vok ? 1.f / sumt : 0.f;
  // ---- softmax over v (per t): lanes 1,2,4,8 apart inside the wave, then across the 8 waves through LDS
  float cm[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float m = xv[j][r];
      m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64));
      m = fmaxf(m, __shfl_xor(m, 4, 64)); m = fmaxf(m, __shfl_xor(m, 8, 64));
      cm[j][r] = m;
      if ((lane & 15) == 0) red[wid * 32 + j * 16 + (lane >> 4) * 4 + r] = m;
    }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = j * 16 + (lane >> 4) * 4 + r;
      float m = red[t];
#pragma unroll
      for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w * 32 + t]);
      cm[j][r] = m;
    }
  __syncthreads();
  float cs[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = (vok && cm[j][r] > -INFINITY) ? __expf(xv[j][r] - cm[j][r]) : 0.f;
      xv[j][r] = e;
      float s = e;
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
      if ((lane & 15) == 0) red[wid * 32 + j * 16 + (lane >> 4) * 4 + r] = s;
    }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = j * 16 + (lane >> 4) * 4 + r;
      float s = red[t];
#pragma unroll
      for (int w = 1; w < NW; ++w) s += red[w * 32 + t];
      if (gridDim.y > 1) {
        if (wid == 0 && (lane & 15) == 0) {
          float* q = a.cstat + (((int64_t)b * gridDim.y + blockIdx.y) * 32 + t) * 2;
          q[0] = cm[j][r];
          q[1] = s;
        }
        cs[j][r] = 1.f;           // (rescaled by cq_colnorm_kernel)
      } else {
        cs[j][r] = s > 0.f ? 1.f / s : 0.f;
      }
    }
  if (!vok) return;
  // ---- outputs.  orient 0 (context = long): Srow = P_t, Scol = P_v, both [b, v, ldP] (8-byte stores);
  //                orient 1 (context = short): Srow = P_v, Scol = P_t, both [b, t, ldP] (column v)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float pt[4], pv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { pt[r] = xt[j][r] * invt; pv[r] = xv[j][r] * cs[j][r]; }
    const int t0 = j * 16 + (lane >> 4) * 4;
    if (a.Pt32 && t0 < a.SP32) {     // (t >= Ls: both probabilities are exactly 0)
      *reinterpret_cast<f32x4*>(a.Pt32 + ((int64_t)b * a.Ll + v) * a.SP32 + t0) = (f32x4){pt[0], pt[1], pt[2], pt[3]};
      *reinterpret_cast<f32x4*>(a.Pv32 + ((int64_t)b * a.Ll + v) * a.SP32 + t0) = (f32x4){pv[0], pv[1], pv[2], pv[3]};
    }
    if (!a.Srow) continue;
    if (a.orient == 0) {
      if (t0 < a.ldP) {
        Vec4<E>::store(reinterpret_cast<E*>(a.Srow) + ((int64_t)b * a.Ll + v) * a.ldP + t0, pt);
        Vec4<E>::store(reinterpret_cast<E*>(a.Scol) + ((int64_t)b * a.Ll + v) * a.ldP + t0, pv);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (t0 + r < a.Ls) {
          a.Srow[((int64_t)b * a.Ls + t0 + r) * a.ldP + v] = bits_from_f<E>(pv[r]);
          a.Scol[((int64_t)b * a.Ls + t0 + r) * a.ldP + v] = bits_from_f<E>(pt[r]);
        }
    }
  }
}

// second pass of a row-split launch: Pv32[b, v, t] *= exp(m_q[t] - M[t]) / sum_q' s_q'[t] exp(m_q'[t] - M[t]), q = the
// workgroup that owned row v.  One workgroup per (clip, row block).
__global__ __launch_bounds__(256) void cq_colnorm_kernel(float* __restrict__ Pv32, const float* __restrict__ cstat, int Ll,
                                                         int SP32, int rows_per) {
  __shared__ float fac[32];
  const int b = blockIdx.x, q = blockIdx.y, nsplit = gridDim.y, tid = threadIdx.x;
  if (tid < 32) {
    const float* st = cstat + (int64_t)b * nsplit * 64 + tid * 2;
    float M = -INFINITY;
    for (int i = 0; i < nsplit; ++i) M = fmaxf(M, st[i * 64]);
    float S = 0.f;
    for (int i = 0; i < nsplit; ++i) S += (st[i * 64] > -INFINITY) ? st[i * 64 + 1] * __expf(st[i * 64] - M) : 0.f;
    fac[tid] = (S > 0.f && st[q * 64] > -INFINITY) ? __expf(st[q * 64] - M) / S : 0.f;
  }
  __syncthreads();
  const int r0 = q * rows_per, nr = min(rows_per, Ll - r0);
  float* P = Pv32 + ((int64_t)b * Ll + r0) * SP32;
  for (int i = tid; i < nr * SP32; i += 256) P[i] *= fac[i % SP32];
}

// orient 1 pads: columns Ll..ldP of every [t] row must be zero (the following GEMMs read ldP columns)
__global__ __launch_bounds__(256) void cq_pad_zero_kernel(bf16_t* __restrict__ A, bf16_t* __restrict__ Bm, int64_t rows, int L,
                                                          int ldP) {
  const int pad = ldP - L;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * pad; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / pad;
    const int c = L + (int)(i - r * pad);
    A[r * ldP + c] = (bf16_t)0.f;
    Bm[r * ldP + c] = (bf16_t)0.f;
  }
}

}  // namespace

extern "C" int vmr_cq_score_supported(int Ll, int Ls, int D, int dtype) {
  return vmr_dtype_16(dtype) && Ll >= 1 && Ll <= 128 && Ls >= 1 && Ls <= 32 && D % 256 == 0 && D >= 256 && D <= 2048;
}
// the row-split form (32 long rows per workgroup + the column-normalisation pass) also takes 128 < Ll <= 256 (BaseFast's
// T = 256): fp32 pair only, with the vmr_cq_score_ws_floats(B) workspace (eight row groups)
extern "C" int vmr_cq_score_split_supported(int Ll, int Ls, int D, int dtype) {
  return vmr_dtype_16(dtype) && Ll >= 1 && Ll <= 256 && Ls >= 1 && Ls <= 32 && D % 256 == 0 && D >= 256 && D <= 2048;
}

static int g_split = -1;   // VMR_CQ_SPLIT / vmr_debug_set_cq_split
extern "C" int vmr_debug_set_cq_split(int mode) {
  g_split = mode;
  return 0;
}
extern "C" int vmr_cq_score_ws_floats(int B) { return B * 8 * 64; }

// colstats != nullptr (vmr_cq_score_ws_floats(B) floats) and only the fp32 pair requested: a clip's rows are split over
// workgroups of 2 waves (256 instead of 64 workgroups at cfg2) + the column-normalisation pass
extern "C" int vmr_cq_score_fwd_ws(const void* lng, const void* sht, const float* shortterm, const float* mask_long,
                                   const float* mask_short, void* Srow, void* Scol, float* Pt_lm, float* Pv_lm, float* colstats,
                                   int B, int Ll, int Ls, int D, int ldP, int orient, int dtype, void* stream) {
  VMR_CHECK(lng && sht && shortterm && mask_long && mask_short, "vmr_cq_score_fwd: null pointer");
  VMR_CHECK((Srow && Scol) || (Pt_lm && Pv_lm), "vmr_cq_score_fwd: no output requested");
  VMR_CHECK((!Srow) == (!Scol) && (!Pt_lm) == (!Pv_lm), "vmr_cq_score_fwd: outputs come in pairs");
  VMR_CHECK((((uintptr_t)Pt_lm | (uintptr_t)Pv_lm) & 15) == 0, "vmr_cq_score_fwd: fp32 outputs must be 16-byte aligned");
  const bool long_rows = Ll > 128;      // only the row-split form takes these
  VMR_CHECK(long_rows ? (vmr_cq_score_split_supported(Ll, Ls, D, dtype) && colstats && !Srow && Pt_lm)
                      : vmr_cq_score_supported(Ll, Ls, D, dtype),
            "vmr_cq_score_fwd: unsupported shape Ll=%d Ls=%d D=%d%s", Ll, Ls, D,
            long_rows ? " (more than 128 long rows: fp32 pair + workspace only)" : "");
  VMR_CHECK(orient == 0 || orient == 1, "vmr_cq_score_fwd: bad orientation");
  VMR_CHECK(!Srow || (ldP % 4 == 0 && ldP >= (orient == 0 ? Ls : Ll)), "vmr_cq_score_fwd: bad ldP");
  if (B == 0) return 0;
  CqArgs a;
  a.lng = (const bf16_t*)lng; a.sht = (const bf16_t*)sht; a.shortterm = shortterm; a.mask_long = mask_long;
  a.mask_short = mask_short; a.Srow = (bf16_t*)Srow; a.Scol = (bf16_t*)Scol;
  a.Pt32 = Pt_lm; a.Pv32 = Pv_lm; a.SP32 = (Ls + 7) / 8 * 8;
  a.Ll = Ll; a.Ls = Ls; a.D = D; a.ldP = ldP; a.orient = orient; a.cstat = colstats;
  const int smem = 32 * D * 2 + 8 * 32 * 4;
  if (g_split < 0) {
    const char* e = getenv("VMR_CQ_SPLIT");
    g_split = e ? atoi(e) : 0;   // measured at cfg2: 10.5 us + 5.1 us (column normalisation) against 15.0 us in one launch
  }
  const bool split = (g_split || long_rows) && colstats && !Srow && Pt_lm && Ll > 32;
  if (smem > 64 * 1024) {
    static thread_local bool done = false;
    if (!done) {
#define VMR_CQ_FNS(E) (const void*)cq_score_kernel<E, 32, 8>, (const void*)cq_score_kernel<E, 40, 8>, (const void*)cq_score_kernel<E, 48, 8>, \
                      (const void*)cq_score_kernel<E, 56, 8>, (const void*)cq_score_kernel<E, 64, 8>, (const void*)cq_score_kernel<E, 32, 2>, \
                      (const void*)cq_score_kernel<E, 40, 2>, (const void*)cq_score_kernel<E, 48, 2>, (const void*)cq_score_kernel<E, 56, 2>, \
                      (const void*)cq_score_kernel<E, 64, 2>
      for (const void* f : {VMR_CQ_FNS(bf16_t), VMR_CQ_FNS(f16_t)}) {
#undef VMR_CQ_FNS
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return vmr_fail(-5, "vmr_cq_score_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
      }
      done = true;
    }
  }
  if (split) {
    const int nsplit = (Ll + 31) / 32;
    switch (D / 256) {
#define VMR_CQ_CASE(n) case n: VMR_DISPATCH16(dtype, E, hipLaunchKernelGGL((cq_score_kernel<E, n * 8, 2>), dim3(B, nsplit), dim3(128), smem, (hipStream_t)stream, a)); break
      VMR_CQ_CASE(1); VMR_CQ_CASE(2); VMR_CQ_CASE(3); VMR_CQ_CASE(4); VMR_CQ_CASE(5); VMR_CQ_CASE(6); VMR_CQ_CASE(7); VMR_CQ_CASE(8);
#undef VMR_CQ_CASE
      default: return vmr_fail(-2, "vmr_cq_score_fwd: unsupported D %d", D);
    }
    VMR_LAUNCH_CHECK();
    hipLaunchKernelGGL(cq_colnorm_kernel, dim3(B, nsplit), dim3(256), 0, (hipStream_t)stream, Pv_lm, colstats, Ll, a.SP32, 32);
    VMR_LAUNCH_CHECK();
    return 0;
  }
  switch (D / 256) {
#define VMR_CQ_CASE(n) case n: VMR_DISPATCH16(dtype, E, hipLaunchKernelGGL((cq_score_kernel<E, n * 8, 8>), dim3(B), dim3(512), smem, (hipStream_t)stream, a)); break
    VMR_CQ_CASE(1); VMR_CQ_CASE(2); VMR_CQ_CASE(3); VMR_CQ_CASE(4); VMR_CQ_CASE(5); VMR_CQ_CASE(6); VMR_CQ_CASE(7); VMR_CQ_CASE(8);
#undef VMR_CQ_CASE
    default: return vmr_fail(-2, "vmr_cq_score_fwd: unsupported D %d", D);
  }
  VMR_LAUNCH_CHECK();
  if (Srow && orient == 1 && ldP > Ll) {
    const int64_t rows = (int64_t)B * Ls;
    hipLaunchKernelGGL(cq_pad_zero_kernel, dim3((unsigned)min((int64_t)1024, (rows * (ldP - Ll) + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (bf16_t*)Srow, (bf16_t*)Scol, rows, Ll, ldP);
    VMR_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int vmr_cq_score_fwd(const void* lng, const void* sht, const float* shortterm, const float* mask_long,
                                const float* mask_short, void* Srow, void* Scol, float* Pt_lm, float* Pv_lm, int B, int Ll,
                                int Ls, int D, int ldP, int orient, int dtype, void* stream) {
  return vmr_cq_score_fwd_ws(lng, sht, shortterm, mask_long, mask_short, Srow, Scol, Pt_lm, Pv_lm, nullptr, B, Ll, Ls, D, ldP,
                             orient, dtype, stream);
}
