// sampler.hip -- BAN's adaptive proposal sampling ON THE DEVICE, one workgroup per clip.
//
// Reference: models/BANlib/model.py:357-435 (`iou`, `proposal_selection_with_negative`, `Aaptive_Proposal_Sampling`): per
// clip a greedy loop over the kept cells of the score map in descending score order -- pick the best unsuppressed
// moment, mark every later-ranked moment whose IoU with it exceeds `thresh` as suppressed, keep the first `neighbor` of
// those as its neighbours, stop after `topk` picks; then [negatives (worst unsuppressed first) | padding (best
// unsuppressed) | selected, in rank order].  Same arithmetic, same order and same tie rule (equal scores keep cell
// order; NaN scores first) as the host routine of sampler_host.hip -- the two are compared for equality in the tests.
//
// Why on the device: the host version cost a [B, C] copy down, ~1 ms of host threads and a copy up BETWEEN two hipGraphs
// (the GPU idles 2-3 ms per BAN step); here the step is one graph.
//   sort     bitonic, 64-bit keys (order-flipped score bits << 32 | cell index: unique keys, so stability is free) in LDS;
//   greedy   thread t owns the EPT consecutive ranks t * EPT ..: "next unsuppressed rank" is a block min, the sweep of a pick
//            is one IoU per owned rank, the first `neighbor` hits in rank order come from a block prefix sum of hit counts;
//   output   two more prefix sums (unsuppressed, selected) give every output slot's source rank.
#include "common.h"

namespace {

constexpr int SMP_T = 1024;          // threads per clip
constexpr int SMP_MAXC = 8192;       // cells per clip (padded to a power of two <= this)

// block-wide exclusive prefix sum of one int per thread (1024 threads = 16 waves); returns the exclusive prefix, *total = sum
__device__ __forceinline__ int block_excl_scan(int v, int* wsum /*[17] LDS*/, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) wsum[w] = x;
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int i = 0; i < SMP_T / 64; ++i) { const int t = wsum[i]; wsum[i] = acc; acc += t; }
    wsum[SMP_T / 64] = acc;
  }
  __syncthreads();
  const int res = wsum[w] + x - v;
  *total = wsum[SMP_T / 64];
  __syncthreads();
  return res;
}

__device__ __forceinline__ int block_min(int v, int* wred /*[16] LDS*/) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = v;
  __syncthreads();
  int r = wred[0];
#pragma unroll
  for (int i = 1; i < SMP_T / 64; ++i) r = min(r, wred[i]);
  __syncthreads();
  return r;
}

template <int EPT>
__global__ __launch_bounds__(SMP_T) void ban_sample_kernel(const float* __restrict__ scores, const int32_t* __restrict__ cells,
                                                           int C, float thresh, int topk, int neighbor, int negative,
                                                           int n_out, int64_t* __restrict__ out, int32_t* __restrict__ status) {
  constexpr int NP = EPT * SMP_T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* key = reinterpret_cast<uint64_t*>(smem);                 // [NP]
  uint32_t* se = reinterpret_cast<uint32_t*>(key + NP);              // [NP]: start | (end + 1) << 16, by rank
  unsigned char* sup = reinterpret_cast<unsigned char*>(se + NP);    // [NP] suppressed
  unsigned char* sel = sup + NP;                                     // [NP] selected
  __shared__ int wsum[SMP_T / 64 + 1];
  __shared__ int wred[SMP_T / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* sc = scores + (int64_t)b * C;
  for (int i = tid; i < NP; i += SMP_T) {
    uint64_t k = ~0ull;                                               // padding sorts last
    if (i < C) {
      const float f = sc[i];
      // ascending key order == descending score order; a NaN ranks with +inf (the host comparator's key: NaN -> +inf,
      // i.e. first, as torch.sort(descending=True) puts it), equal scores by cell index (the low word)
      const uint32_t u = f != f ? 0x7F800000u : __float_as_uint(f);
      const uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      const uint32_t dk = ~asc;
      k = ((uint64_t)dk << 32) | (uint32_t)i;
    }
    key[i] = k;
  }
  __syncthreads();
  // ---- bitonic sort, ascending in the 64-bit key
  for (int k2 = 2; k2 <= NP; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < NP / 2; t += SMP_T) {
        const int i = 2 * t - (t & (j - 1));           // index with bit j clear
        const int p = i + j;
        const bool up = (i & k2) == 0;
        const uint64_t a = key[i], c = key[p];
        if ((a > c) == up) { key[i] = c; key[p] = a; }
      }
      __syncthreads();
    }
  }
  // (-0.f and +0.f compare equal on the host and differ in their bit keys here; the scores are sigmoids, > 0)
  for (int r = tid; r < NP; r += SMP_T) {
    uint32_t v = 0u;
    if (r < C) {
      const int ci = (int)(uint32_t)key[r];
      v = (uint32_t)cells[2 * ci] | ((uint32_t)(cells[2 * ci + 1] + 1) << 16);
    }
    se[r] = v;
    sup[r] = 0; sel[r] = 0;
  }
  __syncthreads();
  // ---- greedy pick-and-suppress; this thread owns ranks r0 .. r0 + EPT - 1
  const int r0 = tid * EPT;
  int prev = -1, count = 0;
  while (true) {
    int cand = 0x7FFFFFFF;
#pragma unroll
    for (int e = EPT - 1; e >= 0; --e) {
      const int r = r0 + e;
      if (r > prev && r < C && !sup[r]) cand = r;
    }
    const int i = block_min(cand, wred);
    if (i >= C - 1) break;                              // the reference loop runs i = 0 .. C - 2
    const float s = (float)(se[i] & 0xFFFFu), en = (float)(se[i] >> 16);
    int hits = 0;
    uint32_t hm = 0u;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int r = r0 + e;
      if (r > i && r < C) {
        const float sr = (float)(se[r] & 0xFFFFu), er = (float)(se[r] >> 16);
        const float inter = fminf(er, en) - fmaxf(sr, s);
        const float uni = fmaxf(er, en) - fminf(sr, s);
        if (fmaxf(inter, 0.f) / uni > thresh) { hm |= 1u << e; ++hits; }
      }
    }
    int tot;
    int rank = block_excl_scan(hits, wsum, &tot);       // hits before this thread's ranks, in rank order
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      if (hm & (1u << e)) {
        const int r = r0 + e;
        if (rank < neighbor) sel[r] = 1;
        ++rank;
        sup[r] = 1;
      }
    }
    if (tid == 0) { sup[i] = 1; sel[i] = 1; }
    __syncthreads();
    prev = i;
    if (++count == topk) break;
  }
  // ---- output: [negatives: the last `negative` unsuppressed ranks, worst first | padding: the first unsuppressed ranks,
  //      while fewer than topk * (neighbor + 1) are selected | the selected ranks in order]
  int nf = 0, ns = 0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int r = r0 + e;
    if (r < C) { nf += !sup[r]; ns += sel[r]; }
  }
  int F, S;
  int fi = block_excl_scan(nf, wsum, &F);
  int si = block_excl_scan(ns, wsum, &S);
  const int total = topk * (neighbor + 1);
  const int nneg = min(negative, F);
  const int npad = S < total ? min(total - S, F) : 0;
  int64_t* ob = out + (int64_t)b * n_out * 2;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int r = r0 + e;
    if (r >= C) continue;
    const int64_t st = (int64_t)(se[r] & 0xFFFFu), e1 = (int64_t)(se[r] >> 16);
    if (!sup[r]) {
      if (fi >= F - nneg) { const int pos = F - 1 - fi; if (pos < n_out) { ob[2 * pos] = st; ob[2 * pos + 1] = e1; } }
      if (fi < npad) { const int pos = nneg + fi; if (pos < n_out) { ob[2 * pos] = st; ob[2 * pos + 1] = e1; } }
      ++fi;
    }
    if (sel[r]) {
      const int pos = nneg + npad + si;
      if (pos < n_out) { ob[2 * pos] = st; ob[2 * pos + 1] = e1; }
      ++si;
    }
  }
  if (tid == 0) status[b] = nneg + npad + S;
}

}  // namespace

// scores [B][C] DEVICE float (score_pred at the kept cells in mask.nonzero() row-major order), cells [C][2] DEVICE int32 (i, j);
// out [B][n_out][2] DEVICE int64 = (start, end + 1) in the reference's order; status [B] DEVICE int32 = proposals produced
// per clip (== n_out when the reference's .view(B, prop_num, 2) would succeed; the caller may check it off the hot path).
extern "C" int vmr_ban_sample(const float* scores, const int32_t* cells, int B, int C, float thresh, int topk, int neighbor,
                              int negative, int n_out, int64_t* out, int32_t* status, void* stream) {
  VMR_CHECK(scores && cells && out && status, "vmr_ban_sample: null pointer");
  VMR_CHECK(B >= 0 && C > 1 && topk > 0 && neighbor >= 0 && negative >= 0 && n_out > 0, "vmr_ban_sample: bad arguments");
  VMR_CHECK(C <= SMP_MAXC, "vmr_ban_sample: %d cells per clip (at most %d)", C, SMP_MAXC);
  if (B == 0) return 0;
  int np = SMP_T;
  while (np < C) np <<= 1;
  const size_t lds = (size_t)np * (8 + 4 + 2);
  hipStream_t st = (hipStream_t)stream;
#define SMP_LAUNCH(EPT)                                                                                               \
  do {                                                                                                                \
    const void* fn = (const void*)ban_sample_kernel<EPT>;                                                             \
    if (lds > 64 * 1024) {                                                                                            \
      hipError_t e_ = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
      if (e_ != hipSuccess) return vmr_fail(-5, "vmr_ban_sample: hipFuncSetAttribute: %s", hipGetErrorString(e_));    \
    }                                                                                                                 \
    hipLaunchKernelGGL(ban_sample_kernel<EPT>, dim3(B), dim3(SMP_T), lds, st, scores, cells, C, thresh, topk, neighbor, \
                       negative, n_out, out, status);                                                                 \
  } while (0)
  switch (np / SMP_T) {
    case 1: SMP_LAUNCH(1); break;
    case 2: SMP_LAUNCH(2); break;
    case 4: SMP_LAUNCH(4); break;
    default: SMP_LAUNCH(8); break;
  }
#undef SMP_LAUNCH
  VMR_LAUNCH_CHECK();
  return 0;
}
