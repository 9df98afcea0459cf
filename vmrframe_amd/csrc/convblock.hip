// convblock.hip -- the row half of one DepthwiseSeparableConvBlock layer's backward in ONE pass.
//
// Reference: models/layers.py:126-148.  A layer is x' = drop(relu(pw(dw7(LN(x))) + b)) + x; its pointwise product
// (and that product's dX / dW) run on the GEMM kernels.  Given du (the gradient of u = dw7(LN(x)), the dX product's
// result) and dy (the gradient of the layer's OUTPUT, which is also the residual branch's gradient) this kernel does
// what vmr_dwconv_bwd2 + vmr_layernorm_bwd + the LOWER layer's vmr_relu_bwd_bias(mode 3) did in three launches:
//
//   dn[s]  = sum_k w[k] * du[s+3-k]                      depthwise-conv input gradient (never leaves the chip)
//   dwk[k] += sum_s n[s] * du[s+3-k],  n = LN(x)         conv-weight gradient
//   dx     = LN-backward(dn; x, mean, rstd, gamma) + dy  -> the lower layer's dy
//   dz     = dx * bits * bscale                          the lower layer's ReLU / dropout backward (bit matrix written
//                                                        by that layer's GEMM epilogue, VMR_EPI_AUX_BITS)
//   dgamma, dbeta: column sums  (the lower layer's bias gradient = colsum(dz) rides on its weight-gradient GEMM,
//                                vmr_gemm_t.a_colsum)
//
// HBM traffic per layer at cfg2 (9472 x 1024 bf16): du, x, dy read + dx, dz written = 97 MB against 176 MB for the
// three kernels (dn and dy each made a round trip).  Results are those of the three kernels: dn passes through the
// storage type exactly where it used to be stored, the row reductions keep ln_bwd_kernel's lane mapping.
//
// PERSISTENT workgroups, one per CU (NT = D / 2 threads, 8 waves at D = 1024): a workgroup walks chunks g, g + G, ...
// where a chunk is 8 consecutive frames of one sequence (+ 3 halo rows of du on either side).  Everything that depends
// on the channel only -- conv weights, gamma, beta, and the three parameter-gradient accumulators -- stays in
// registers over all of a workgroup's chunks: the first form (one workgroup per 16-row chunk) spent 9 of its 14 us
// fetching those constants and storing 40 KB of partial sums, per chunk.  Per chunk:
//   phase 1  thread = 2 adjacent channels (float2 arithmetic: v_pk_fma_f32), walks the rows with a 7-row window of du
//            in registers: dn -> LDS (storage type), x -> LDS, parameter-gradient partials in registers;
//   phase 2  wave = one row (lane = 8-channel chunks, ln_bwd_kernel's mapping): the two row sums, dx, dz.
// The NEXT chunk's 22 rows are requested right after phase 1 (into the registers it has just emptied) and arrive under
// phase 2; phase 2's own operands (dy row, mask bytes) are requested before phase 1.
#include "common.h"

namespace {

constexpr int CB_R = 8;   // rows per chunk

struct CBArgs {
  const void *du, *x, *dres;
  const unsigned char* bits;
  float bscale;
  const float *gamma, *beta, *mean, *rstd, *w;
  void *dx, *dz;
  float *part_dw, *part_gb;
  int S1, cps1 /* chunks per sequence */, nc1 /* chunks of group 1 */, S2, cps2, nchunks;
  int64_t rows1;
};

__device__ __forceinline__ float lane_f(float v, int l) {   // lane l's value as a wave-uniform (SGPR) float
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l));
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 and2(f32x2 v, uint32_t m) {   // both lanes' float bits ANDed with a wave-uniform mask
  f32x2 r;
  r[0] = __uint_as_float(__float_as_uint(v[0]) & m);
  r[1] = __uint_as_float(__float_as_uint(v[1]) & m);
  return r;
}

template <typename T, int NT>
__global__ __launch_bounds__(NT, 2) void convblock_bwd_kernel(CBArgs a) {
  constexpr int CPT = 2, D = NT * CPT, MAXC = D / 512, R = CB_R, NW = NT / 64, RPW = R / NW;
  static_assert(D % 512 == 0 && R % NW == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* dnT = reinterpret_cast<T*>(smem);                  // [R][D]
  T* xT = dnT + R * D;                                  // [R][D]
  float* gamS = reinterpret_cast<float*>(xT + R * D);   // [D]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T* du = reinterpret_cast<const T*>(a.du);
  const T* x = reinterpret_cast<const T*>(a.x);
  const T* dres = reinterpret_cast<const T*>(a.dres);
  T* dx = reinterpret_cast<T*>(a.dx);
  T* dz = reinterpret_cast<T*>(a.dz);
  const float* __restrict__ meanp = a.mean;
  const float* __restrict__ rstdp = a.rstd;
  const int c0 = tid * CPT;
  typedef __attribute__((ext_vector_type(CPT))) T TVC;
  typedef __attribute__((ext_vector_type(8))) T TV8;
  auto cvt2 = [](TVC v) -> f32x2 { f32x2 r; r[0] = to_f<T>(v[0]); r[1] = to_f<T>(v[1]); return r; };
  // chunk -> (sequence length, global row of the sequence's frame 0, first frame); wave-uniform
  auto locate = [&](int ci, int& S, int64_t& rb, int& r0) {
    int cps = a.cps1;
    int64_t base = 0;
    S = a.S1;
    if (ci >= a.nc1) { ci -= a.nc1; S = a.S2; cps = a.cps2; base = a.rows1; }
    const int b = ci / cps;
    r0 = (ci - b * cps) * R;
    rb = base + (int64_t)b * S;
  };
  // a chunk's rows: du frames r0-3 .. r0+10 and x frames r0 .. r0+7, in the STORAGE type until used; unconditional loads
  // on clamped frames, zeroed by masks (a select lets hipcc sink each load into its own branch + vmcnt(0))
  TVC pdu[R + 6], px[R];
  auto request = [&](int ci) {
    int S, r0; int64_t rb;
    locate(min(ci, a.nchunks - 1), S, rb, r0);
#pragma unroll
    for (int j = 0; j < R + 6; ++j) pdu[j] = *reinterpret_cast<const TVC*>(du + (rb + min(max(r0 - 3 + j, 0), S - 1)) * D + c0);
#pragma unroll
    for (int j = 0; j < R; ++j) px[j] = *reinterpret_cast<const TVC*>(x + (rb + min(r0 + j, S - 1)) * D + c0);
  };
  request(blockIdx.x);
  // per-channel constants, once per workgroup
  f32x2 wk[7], aw[7], g1, b1, ag1 = {0.f, 0.f}, ab1 = {0.f, 0.f};
  {
    f32x2 gl[D / 2 / NT];
#pragma unroll
    for (int i = 0; i < D / 2 / NT; ++i) gl[i] = *reinterpret_cast<const f32x2*>(a.gamma + (i * NT + tid) * 2);
    g1 = *reinterpret_cast<const f32x2*>(a.gamma + c0);
    b1 = *reinterpret_cast<const f32x2*>(a.beta + c0);
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      wk[k][0] = a.w[c0 * 7 + k]; wk[k][1] = a.w[(c0 + 1) * 7 + k];
      aw[k] = f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < D / 2 / NT; ++i) *reinterpret_cast<f32x2*>(gamS + (i * NT + tid) * 2) = gl[i];
  }
  for (int ci = blockIdx.x; ci < a.nchunks; ci += gridDim.x) {
    int S, r0; int64_t rb;
    locate(ci, S, rb, r0);
    const int nrows = min(R, S - r0);
    // the chunk's row statistics: lanes 0..7 of every wave fetch one row's pair, v_readlane hands them out (the loop
    // stores through other pointers, so as scalar-indexed loads hipcc makes them 16 vector loads + 16 VGPRs)
    const int64_t srow = rb + min(r0 + (lane & (R - 1)), S - 1);
    const float mv = meanp[srow], rv = rstdp[srow];
    // phase 2's operands of this wave's rows, requested now: they arrive under phase 1
    TV8 rr[RPW][MAXC];
    uint32_t bb[RPW][MAXC];
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      const int64_t row = rb + min(r0 + wid + NW * q, S - 1);
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        const int i = (c * 64 + lane) * 8;
        rr[q][c] = *reinterpret_cast<const TV8*>(dres + row * D + i);
        bb[q][c] = a.bits ? a.bits[row * (D >> 3) + (i >> 3)] : 0u;
      }
    }
    // ---- phase 1: conv backward, this thread's 2 channels over the chunk's rows
    // (row masks as INTEGER selects -- SALU, an SGPR each -- ANDed into the float bits: as 0.f / 1.f factors every one
    //  of the wave-uniform masks sat in its own VGPR)
    f32x2 wdu[7];   // window index j holds du frame (cur - 3 + j)
    wdu[0] = f32x2{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int s = r0 - 3 + j;
      wdu[j + 1] = and2(cvt2(pdu[j]), (s >= 0 && s < S) ? 0xFFFFFFFFu : 0u);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int s = r0 + j;
      const uint32_t mdu = (s + 3 < S) ? 0xFFFFFFFFu : 0u;   // (s + 3 >= 0 always)
      const uint32_t valid = j < nrows ? 0xFFFFFFFFu : 0u;
#pragma unroll
      for (int t = 0; t < 6; ++t) wdu[t] = wdu[t + 1];
      wdu[6] = and2(cvt2(pdu[6 + j]), mdu);
      const float mean = lane_f(mv, j), rstd = lane_f(rv, j);
      f32x2 acc = wk[0] * wdu[6];                            // dn[s] = sum_k w[k] * du[s+3-k]
#pragma unroll
      for (int k = 1; k < 7; ++k) acc += wk[k] * wdu[6 - k];
      TVC dnv;
      dnv[0] = from_f<T>(acc[0]); dnv[1] = from_f<T>(acc[1]);
      const f32x2 h = (cvt2(px[j]) - mean) * rstd;
      const f32x2 n = and2(h * g1 + b1, valid);
#pragma unroll
      for (int k = 0; k < 7; ++k) aw[k] += n * wdu[6 - k];   // dw[k] += n[s] * du[s+3-k]
      const f32x2 dr = and2(cvt2(dnv), valid);               // what the LayerNorm backward reads
      ag1 += dr * h;
      ab1 += dr;
      *reinterpret_cast<TVC*>(dnT + j * D + c0) = dnv;
      *reinterpret_cast<TVC*>(xT + j * D + c0) = px[j];
    }
    // the next chunk's rows, into the registers phase 1 has just emptied: they fly under phase 2
    request(ci + gridDim.x);
    __syncthreads();
    // ---- phase 2: LayerNorm backward + residual + the lower layer's mask, one wave per row (operands re-read from
    //      LDS in the second pass rather than kept: registers)
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      const int jj = wid + NW * q;
      if (jj < nrows) {   // wave-uniform
        const int64_t row = rb + r0 + jj;
        const float mean = lane_f(mv, jj), rstd = lane_f(rv, jj);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          const int i = (c * 64 + lane) * 8;
          float d[8], xv[8], g[8];
          Vec8<T>::load(dnT + jj * D + i, d);
          Vec8<T>::load(xT + jj * D + i, xv);
          Vec8<float>::load(gamS + i, g);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float h = (xv[e] - mean) * rstd;
            const float dh = d[e] * g[e];
            s1 += dh;
            s2 += dh * h;
          }
        }
        s1 = wave_sum_dpp(s1) / (float)D;
        s2 = wave_sum_dpp(s2) / (float)D;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          const int i = (c * 64 + lane) * 8;
          float d[8], xv[8], g[8], o[8];
          Vec8<T>::load(dnT + jj * D + i, d);
          Vec8<T>::load(xT + jj * D + i, xv);
          Vec8<float>::load(gamS + i, g);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float h = (xv[e] - mean) * rstd;
            o[e] = rstd * (d[e] * g[e] - s1 - h * s2) + to_f<T>(rr[q][c][e]);
          }
          Vec8<T>::store(dx + row * D + i, o);
          if (dz) {
            float z[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float ov = to_f<T>(from_f<T>(o[e]));   // the stored dx is what the lower layer's backward reads
              z[e] = ((bb[q][c] >> e) & 1) ? ov * a.bscale : 0.f;
            }
            Vec8<T>::store(dz + row * D + i, z);
          }
        }
      }
    }
    __syncthreads();
  }
  // ---- ONE partial row per workgroup (plain stores; vmr_colreduce_batched sums them)
  float* pdw = a.part_dw + (int64_t)blockIdx.x * D * 7;
#pragma unroll
  for (int ch = 0; ch < CPT; ++ch)
#pragma unroll
    for (int k = 0; k < 7; ++k) pdw[(c0 + ch) * 7 + k] = aw[k][ch];
  float* pgb = a.part_gb + (int64_t)blockIdx.x * 2 * D;
  *reinterpret_cast<f32x2*>(pgb + c0) = ag1;
  *reinterpret_cast<f32x2*>(pgb + D + c0) = ab1;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: u = dw7(LN(x)) with the same persistent skeleton (replaces ln_dwconv_fwd_kernel of norm.hip for D = 512 / 1024:
// that kernel's workgroups normalised 16 rows to produce 10 -- a 60 % halo -- and ran load / normalise / barrier /
// convolve / store in lock-step; here a workgroup keeps gamma, beta and its channels' taps in registers over all its
// chunks, the NEXT chunk's rows are requested before the current one is convolved, and two LDS tiles alternate so a
// chunk costs one barrier).  Chunk = 8 output frames; its 14 input frames (3-frame halo each side) are normalised by
// one wave per row (ln_fwd_kernel's lane mapping and two-pass statistics) into an LDS tile in the storage type -- what
// the two-phase kernel staged too -- and the convolution runs thread = 2 channels over a sliding window read from it.
// ---------------------------------------------------------------------------------------------------------------------
struct CFArgs {
  const void* x;
  const float *gamma, *beta, *w;
  float eps;
  void* u;
  float *mean, *rstd;
  int S1, cps1, nc1, S2, cps2, nchunks;
  int64_t rows1;
};

template <typename T, int NT>
__global__ __launch_bounds__(NT, 2) void convblock_fwd_kernel(CFArgs a) {
  constexpr int CPT = 2, D = NT * CPT, MAXC = D / 512, R = CB_R, NR = R + 6, NW = NT / 64, RPW = (NR + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* tiles = reinterpret_cast<T*>(smem);   // [2][NR][D]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T* x = reinterpret_cast<const T*>(a.x);
  T* u = reinterpret_cast<T*>(a.u);
  const int c0 = tid * CPT;
  typedef __attribute__((ext_vector_type(CPT))) T TVC;
  typedef __attribute__((ext_vector_type(8))) T TV8;
  auto locate = [&](int ci, int& S, int64_t& rb, int& r0) {
    int cps = a.cps1;
    int64_t base = 0;
    S = a.S1;
    if (ci >= a.nc1) { ci -= a.nc1; S = a.S2; cps = a.cps2; base = a.rows1; }
    const int b = ci / cps;
    r0 = (ci - b * cps) * R;
    rb = base + (int64_t)b * S;
  };
  // this wave's rows of a chunk (tile rows wid, wid + NW, ...: frames r0 - 3 + row), clamped, in the storage type
  TV8 px[RPW][MAXC];
  auto request = [&](int ci) {
    int S, r0; int64_t rb;
    locate(min(ci, a.nchunks - 1), S, rb, r0);
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      const int s = min(max(r0 - 3 + wid + NW * q, 0), S - 1);
#pragma unroll
      for (int c = 0; c < MAXC; ++c) px[q][c] = *reinterpret_cast<const TV8*>(x + (rb + s) * D + (c * 64 + lane) * 8);
    }
  };
  request(blockIdx.x);
  float gam[MAXC][8], bet[MAXC][8];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    Vec8<float>::load(a.gamma + (c * 64 + lane) * 8, gam[c]);
    Vec8<float>::load(a.beta + (c * 64 + lane) * 8, bet[c]);
  }
  f32x2 wk[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) { wk[k][0] = a.w[c0 * 7 + k]; wk[k][1] = a.w[(c0 + 1) * 7 + k]; }
  int buf = 0;
  for (int ci = blockIdx.x; ci < a.nchunks; ci += gridDim.x, buf ^= 1) {
    int S, r0; int64_t rb;
    locate(ci, S, rb, r0);
    const int nrows = min(R, S - r0);
    T* tile = tiles + buf * NR * D;
    // ---- phase A: LayerNorm of the chunk's 14 input frames, one wave per frame
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      const int r = wid + NW * q;        // tile row
      if (r < NR) {                      // wave-uniform
        const int s = r0 - 3 + r;
        float v[MAXC][8];
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[c][e] = to_f<T>(px[q][c][e]);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
#pragma unroll
          for (int e = 0; e < 8; ++e) sum += v[c][e];
        const float mean = wave_sum_dpp(sum) / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; sq += d * d; }
        const float rstd = rsqrtf(wave_sum_dpp(sq) / (float)D + a.eps);
        const bool inside = s >= 0 && s < S;                 // outside the sequence: the conv's zero padding
        if (lane == 0 && inside && r >= 3 && r < 3 + nrows) {
          a.mean[rb + s] = mean;
          a.rstd[rb + s] = rstd;
        }
        const float msk = inside ? 1.f : 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = ((v[c][e] - mean) * rstd * gam[c][e] + bet[c][e]) * msk;
          Vec8<T>::store(tile + r * D + (c * 64 + lane) * 8, o);
        }
      }
    }
    request(ci + gridDim.x);             // the next chunk's rows fly under the convolution below
    __syncthreads();
    // ---- phase B: depthwise conv (k = 7), this thread's 2 channels, sliding window over the tile
    f32x2 win[7];
    auto tl = [&](int r) -> f32x2 {
      const TVC t = *reinterpret_cast<const TVC*>(tile + r * D + c0);
      f32x2 o; o[0] = to_f<T>(t[0]); o[1] = to_f<T>(t[1]); return o;
    };
#pragma unroll
    for (int k = 0; k < 6; ++k) win[k + 1] = tl(k);
#pragma unroll
    for (int j = 0; j < R; ++j) {
#pragma unroll
      for (int k = 0; k < 6; ++k) win[k] = win[k + 1];
      win[6] = tl(j + 6);
      f32x2 acc = wk[0] * win[0];        // u[s] = sum_k w[k] * n[s + k - 3]
#pragma unroll
      for (int k = 1; k < 7; ++k) acc += wk[k] * win[k];
      if (j < nrows) {
        TVC o;
        o[0] = from_f<T>(acc[0]); o[1] = from_f<T>(acc[1]);
        *reinterpret_cast<TVC*>(u + (rb + r0 + j) * D + c0) = o;
      }
    }
    // (no second barrier: the next chunk's phase A writes the OTHER tile; the one after that comes behind the next
    //  chunk's barrier, which every wave reaches only after finishing this convolution)
  }
}

// persistent grid: one 512-thread (two 256-thread) workgroup(s) per CU, never more than there are chunks
int cb_grid(int nchunks, int D) {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
    const char* e = getenv("VMR_CONVBLOCK_GRID");   // A/B switch: workgroups per launch
    if (e && atoi(e) > 0) ncu = atoi(e);
  }
  int g = D == 512 ? 2 * ncu : ncu;
  if (g > VMR_CONVBLOCK_BWD_MAX_BLOCKS) g = VMR_CONVBLOCK_BWD_MAX_BLOCKS;
  return g < nchunks ? g : nchunks;
}

}  // namespace

extern "C" int vmr_convblock_bwd_blocks(int B1, int S1, int B2, int S2, int D) {
  const int nchunks = (B1 > 0 && S1 > 0 ? B1 * cdiv(S1, CB_R) : 0) + (B2 > 0 && S2 > 0 ? B2 * cdiv(S2, CB_R) : 0);
  return cb_grid(nchunks, D);
}

// called by vmr_ln_dwconv_fwd2 (norm.hip) for the widths this file is built for; -1 = not taken
int convblock_fwd_launch(const void* x, const float* gamma, const float* beta, float eps, const float* w, void* u, float* mean,
                         float* rstd, int B1, int S1, int B2, int S2, int D, int dtype, void* stream) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("VMR_CONVBLOCK_FWD");     // A/B switch: 0 = the two-phase tile kernel of norm.hip
    on = e && atoi(e) == 0 ? 0 : 1;
  }
  if (!on || !(D == 512 || D == 1024) || !vmr_dtype_ok(dtype)) return -1;
  CFArgs a;
  a.x = x; a.gamma = gamma; a.beta = beta; a.w = w; a.eps = eps; a.u = u; a.mean = mean; a.rstd = rstd;
  a.S1 = S1; a.S2 = S2;
  a.cps1 = cdiv(S1, CB_R); a.cps2 = cdiv(S2, CB_R);
  a.nc1 = B1 * a.cps1;
  a.nchunks = a.nc1 + B2 * a.cps2;
  a.rows1 = (int64_t)B1 * S1;
  if (a.nchunks == 0) return 0;
  const int nb = cb_grid(a.nchunks, D);
  const size_t lds = (size_t)2 * (CB_R + 6) * D * vmr_dtype_size(dtype);
  hipStream_t st = (hipStream_t)stream;
#define CF_LAUNCH(T, NT)                                                                                            \
  do {                                                                                                              \
    const void* fn = (const void*)convblock_fwd_kernel<T, NT>;                                                      \
    if (lds > 64 * 1024) {                                                                                          \
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
      if (e != hipSuccess) return vmr_fail(-5, "vmr_ln_dwconv_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); \
    }                                                                                                               \
    hipLaunchKernelGGL((convblock_fwd_kernel<T, NT>), dim3(nb), dim3(NT), lds, st, a);                              \
  } while (0)
  if (D == 512) VMR_DISPATCH(dtype, T, CF_LAUNCH(T, 256));
  else VMR_DISPATCH(dtype, T, CF_LAUNCH(T, 512));
#undef CF_LAUNCH
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_convblock_bwd_supported(int D, int dtype) {
  return (D == 512 || D == 1024) && vmr_dtype_ok(dtype) ? 1 : 0;
}

extern "C" int vmr_convblock_bwd(const void* du, const void* x, const void* dres, const unsigned char* bits,
                                 float bscale, const float* gamma, const float* beta, const float* mean,
                                 const float* rstd, const float* w, void* dx, void* dz, float* part_dw,
                                 float* part_gb, int B1, int S1, int B2, int S2, int D, int dtype,
                                 int32_t* nblocks, void* stream) {
  VMR_CHECK(vmr_convblock_bwd_supported(D, dtype), "vmr_convblock_bwd: D = %d / dtype %d not supported (D in {512, 1024})", D, dtype);
  VMR_CHECK(du && x && dres && gamma && beta && mean && rstd && w && dx && part_dw && part_gb && nblocks,
            "vmr_convblock_bwd: null pointer");
  VMR_CHECK((dz == nullptr) == (bits == nullptr), "vmr_convblock_bwd: dz and bits come together");
  VMR_CHECK(B1 >= 0 && S1 >= 0 && B2 >= 0 && S2 >= 0, "vmr_convblock_bwd: negative shape");
  *nblocks = 0;
  if (B1 == 0 || S1 == 0) { B1 = 0; S1 = S1 > 0 ? S1 : 1; }
  if (B2 == 0 || S2 == 0) { B2 = 0; S2 = S2 > 0 ? S2 : 1; }
  if (B1 + B2 == 0) return 0;
  CBArgs a;
  a.du = du; a.x = x; a.dres = dres; a.bits = bits; a.bscale = bscale;
  a.gamma = gamma; a.beta = beta; a.mean = mean; a.rstd = rstd; a.w = w;
  a.dx = dx; a.dz = dz; a.part_dw = part_dw; a.part_gb = part_gb;
  a.S1 = S1; a.S2 = S2;
  a.cps1 = cdiv(S1, CB_R); a.cps2 = cdiv(S2, CB_R);
  a.nc1 = B1 * a.cps1;
  a.nchunks = a.nc1 + B2 * a.cps2;
  a.rows1 = (int64_t)B1 * S1;
  const int nb = cb_grid(a.nchunks, D);
  *nblocks = nb;
  const size_t esz = (size_t)vmr_dtype_size(dtype);
  const size_t lds = (size_t)2 * CB_R * D * esz + (size_t)D * 4;
  hipStream_t st = (hipStream_t)stream;
#define CB_LAUNCH(T, NT)                                                                                      \
  do {                                                                                                             \
    const void* fn = (const void*)convblock_bwd_kernel<T, NT>;                                                \
    if (lds > 64 * 1024) {                                                                                         \
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
      if (e != hipSuccess) return vmr_fail(-5, "vmr_convblock_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); \
    }                                                                                                              \
    hipLaunchKernelGGL((convblock_bwd_kernel<T, NT>), dim3(nb), dim3(NT), lds, st, a);                       \
  } while (0)
  if (D == 512) VMR_DISPATCH(dtype, T, CB_LAUNCH(T, 256));
  else VMR_DISPATCH(dtype, T, CB_LAUNCH(T, 512));
#undef CB_LAUNCH
  VMR_LAUNCH_CHECK();
  return 0;
}
