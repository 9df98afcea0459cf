// map2d.hip -- BAN 2-D proposal map: the reference builds a [B, 3F, N, N] map with Python
// loops over diagonals -- SparseMaxPool / DenseMaxPool (models/BANlib/model.py:226-290: cell (i, j) = max over
// frames i..j of the content features, produced by a CASCADE of MaxPool1d(2|3|5, stride 1) so every diagonal
// is one more pooling of the previous one) and SparseBoundaryCat (:293-325: cell (i, j) = [start[i] | end[j]]) --
// then runs Linear(3F -> F) over all N*N cells, two thirds of them zeros.
//
// Here only the cells the mask keeps exist, in COMPACT cell-major order (diagonal by diagonal, as the reference's
// `maskij` list: the main diagonal first, then offset o_1, o_2, ...; within a diagonal by start frame i):
//   M[b, c, :] = max_{t in [i_c, j_c]} x[b, t, :]                    (content; feeds the Wc third of map2d_proj)
//   R[b, c, :] = Ps[b, i_c, :] + Pe[b, j_c, :]                       (boundary: Ps = start.Ws^T, Pe = end.We^T were
//                                                                     projected per FRAME, not per cell)
// so that map2d_proj(cat[start_i, end_j, pool_ij]) = act(M.Wc^T + b + R) is ONE K=F GEMM with R as the
// pre-activation residual (VMR_EPI_RES_PRE) over 1/3 of the cells at 1/3 of the K.
//
// Diagonals are described by `grow[k]` = how many frames diagonal k's window grows over diagonal k-1's
// (= MaxPool1d kernel size - 1: 1 for the first level, 2, 4 for the sparse levels; 1 everywhere for DenseMaxPool).
// Workgroup = (clip b, 64-channel slice); thread = (channel pair, run of consecutive start frames); every global
// access is a 4-byte-per-lane, 128-byte-per-row segment of a cell row, and the rows the NEXT diagonal needs (Pe / dM)
// are requested as soon as the current diagonal has been written.
#include "common.h"

namespace {

constexpr int MP_CH = 64;      // channels per workgroup
constexpr int MP_NMAX = 160;   // frames (8-bit arg-max fields; 20 start frames per thread held in registers)

constexpr int MP_ITEMS = MP_NMAX / 8;   // consecutive start frames per thread (the largest RUN instantiated)

// Both cascade kernels keep the running window maxima in REGISTERS: thread (pc, rg) owns channel pair pc of the
// RUN consecutive start frames i = rg*RUN .. rg*RUN + RUN-1 (RUN >= ceil(N / 8): 8, 16 or 20, compile-time).  Growing every window by one frame
// is new[i] = max(cur[i], cur[i+1]): in-register except for the last frame of the run, whose right neighbour is
// the first frame of the next thread's run -- one 8-byte LDS word per thread per step (double-buffered, one
// barrier).  A diagonal whose window grows by g frames (MaxPool1d(g+1, 1)) is g such steps, cells are written
// after the last one.  LDS holds only that halo (and, in the backward, the dx image).
// A channel pair as loaded (converted to float only where it is consumed): the prefetching loads below are
// UNCONDITIONAL (row index clamped into the tensor) and carry no dependent ALU work, so the compiler neither
// wraps each in its own exec-masked branch nor waits for it on the spot -- predicated loads with an immediate
// bf16 -> f32 conversion ran the backward at one HBM round trip per load (1.0 ms instead of 0.2 ms).
template <typename T> struct Raw;
template <> struct Raw<float> {
  float2 v;
  __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float2*>(p); }
  __device__ __forceinline__ float x() const { return v.x; }
  __device__ __forceinline__ float y() const { return v.y; }
};
template <> struct Raw<bf16_t> {
  uint32_t v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint32_t*>(p); }
  __device__ __forceinline__ float x() const { return __uint_as_float(v << 16); }
  __device__ __forceinline__ float y() const { return __uint_as_float(v & 0xFFFF0000u); }
};

template <> struct Raw<f16_t> {
  uint32_t v;
  __device__ __forceinline__ void load(const f16_t* p) { v = *reinterpret_cast<const uint32_t*>(p); }
  __device__ __forceinline__ float x() const { return e16_lo<f16_t>(v); }
  __device__ __forceinline__ float y() const { return e16_hi<f16_t>(v); }
};

template <typename T, int RUN>
__global__ __launch_bounds__(256) void map2d_pool_fwd_kernel(const T* __restrict__ x, const T* __restrict__ ps,
                                                             const T* __restrict__ pe, int64_t ldp,
                                                             const int* __restrict__ grow, int ndiag, T* __restrict__ M,
                                                             T* __restrict__ R, int N, int F, int64_t C) {
  __shared__ float2 halo[2][9][32];
  const int pc = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int slices = F / MP_CH;
  const int b = blockIdx.x / slices, ch = (blockIdx.x % slices) * MP_CH + pc * 2;
  const int i0 = rg * RUN;           // RUN >= ceil(N / 8) (compile-time: the cascade is straight-line VALU code)
  const T* xb = x + (int64_t)b * N * F + ch;
  const T* psb = ps ? ps + (int64_t)b * N * ldp + ch : nullptr;
  const T* peb = pe ? pe + (int64_t)b * N * ldp + ch : nullptr;
  T* Mb = M + (int64_t)b * C * F + ch;
  T* Rb = R ? R + (int64_t)b * C * F + ch : nullptr;
  float cx[RUN], cy[RUN];
  Raw<T> psv[RUN], pev[RUN], xr[RUN], pe0[RUN];
  if (threadIdx.x < 64) halo[threadIdx.x >> 5][8][pc] = make_float2(-INFINITY, -INFINITY);   // right of the last run
  int o = ndiag > 0 ? grow[0] : 0;     // offset of the next diagonal to be written
  const T* psc = Rb ? psb : xb;        // (no boundary operands: the loads below read x instead and are ignored)
  const T* pec = Rb ? peb : xb;
  const int64_t ldc = Rb ? ldp : (int64_t)F;
#pragma unroll
  for (int u = 0; u < RUN; ++u) {
    const int t = min(i0 + u, N - 1);
    xr[u].load(xb + (int64_t)t * F);
    psv[u].load(psc + (int64_t)t * ldc);
    pe0[u].load(pec + (int64_t)t * ldc);
    pev[u].load(pec + (int64_t)min(t + o, N - 1) * ldc);          // first diagonal's end rows
  }
#pragma unroll
  for (int u = 0; u < RUN; ++u) {
    const int t = i0 + u;
    cx[u] = t < N ? xr[u].x() : -INFINITY;
    cy[u] = t < N ? xr[u].y() : -INFINITY;
    if (t < N) {
      float v[2] = {cx[u], cy[u]};
      Vec2<T>::store(Mb + (int64_t)t * F, v);
      if (Rb) {
        float a[2] = {psv[u].x() + pe0[u].x(), psv[u].y() + pe0[u].y()};
        Vec2<T>::store(Rb + (int64_t)t * F, a);
      }
    }
  }
  int64_t cell = N;
  int step = 0;
  for (int k = 0; k < ndiag; ++k) {
    const int g = grow[k];
    for (int s = 0; s < g; ++s, ++step) {
      halo[step & 1][rg][pc] = make_float2(cx[0], cy[0]);
      __syncthreads();
      const float2 h = halo[step & 1][rg + 1][pc];
#pragma unroll
      for (int u = 0; u < RUN; ++u) {
        cx[u] = fmaxf(cx[u], u + 1 < RUN ? cx[u + 1 < RUN ? u + 1 : u] : h.x);
        cy[u] = fmaxf(cy[u], u + 1 < RUN ? cy[u + 1 < RUN ? u + 1 : u] : h.y);
      }
    }
    const int len = N - o;
    const int on = o + (k + 1 < ndiag ? grow[k + 1] : N);   // next diagonal's offset (N: none)
    float ra[RUN][2];
#pragma unroll
    for (int u = 0; u < RUN; ++u) {
      ra[u][0] = psv[u].x() + pev[u].x();
      ra[u][1] = psv[u].y() + pev[u].y();
      pev[u].load(pec + (int64_t)min(i0 + u + on, N - 1) * ldc);   // next diagonal's end rows: land during its steps
    }
#pragma unroll
    for (int u = 0; u < RUN; ++u) {
      const int i = i0 + u;
      if (i < len) {
        float o2[2] = {cx[u], cy[u]};
        Vec2<T>::store(Mb + (cell + i) * F, o2);
        if (Rb) Vec2<T>::store(Rb + (cell + i) * F, ra[u]);
      }
    }
    cell += len > 0 ? len : 0;
    o = on;
  }
}

// Backward of the max cascade: the same register cascade carrying the arg-max frame of every window (the later
// frame replaces the running maximum only if STRICTLY greater: first frame wins ties, as the chained MaxPool1d
// backward does); dM of every written cell is added into an LDS image of dx.  The dM rows of the next diagonal are requested as soon as the current one has been consumed.
template <typename T, int RUN>
__global__ __launch_bounds__(256) void map2d_pool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dM,
                                                             const int* __restrict__ grow, int ndiag, T* __restrict__ dx,
                                                             int N, int F, int64_t C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* acc = reinterpret_cast<float*>(smem);           // dx image [N][64]
  __shared__ float2 halo[2][9][32];
  __shared__ int hidx[2][9][32];
  const int pc = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int slices = F / MP_CH;
  const int b = blockIdx.x / slices, ch = (blockIdx.x % slices) * MP_CH + pc * 2;
  const int i0 = rg * RUN;
  const T* xb = x + (int64_t)b * N * F + ch;
  const T* gb = dM + (int64_t)b * C * F + ch;
  float cx[RUN], cy[RUN];
  Raw<T> dn[RUN], xr[RUN], d0[RUN];
  int ax[RUN], ay[RUN];                                 // arg-max frames of the two channels
  if (threadIdx.x < 64) halo[threadIdx.x >> 5][8][pc] = make_float2(-INFINITY, -INFINITY);
  int o = ndiag > 0 ? grow[0] : N;
  int64_t cell = N;
#pragma unroll
  for (int u = 0; u < RUN; ++u) {
    const int t = min(i0 + u, N - 1);
    xr[u].load(xb + (int64_t)t * F);
    d0[u].load(gb + (int64_t)t * F);
    dn[u].load(gb + min(cell + t, C - 1) * F);          // first diagonal's cells
  }
#pragma unroll
  for (int u = 0; u < RUN; ++u) {
    const int t = i0 + u;
    cx[u] = t < N ? xr[u].x() : -INFINITY;
    cy[u] = t < N ? xr[u].y() : -INFINITY;
    ax[u] = ay[u] = t;
    if (t < N) {
      acc[t * MP_CH + pc * 2] = d0[u].x();          // main-diagonal cells: the frame itself
      acc[t * MP_CH + pc * 2 + 1] = d0[u].y();
    }
  }
  __syncthreads();
  int step = 0;
  for (int k = 0; k < ndiag; ++k) {
    const int g = grow[k];
    for (int s = 0; s < g; ++s, ++step) {
      halo[step & 1][rg][pc] = make_float2(cx[0], cy[0]);
      hidx[step & 1][rg][pc] = ax[0] | (ay[0] << 8);
      __syncthreads();
      const float2 h = halo[step & 1][rg + 1][pc];
      const int hi = rg < 7 ? hidx[step & 1][rg + 1][pc] : 0;
#pragma unroll
      for (int u = 0; u < RUN; ++u) {
        const int un = u + 1 < RUN ? u + 1 : u;
        const float nx = u + 1 < RUN ? cx[un] : h.x, ny = u + 1 < RUN ? cy[un] : h.y;
        const int nix = u + 1 < RUN ? ax[un] : (hi & 0xFF), niy = u + 1 < RUN ? ay[un] : (hi >> 8);
        ax[u] = nx > cx[u] ? nix : ax[u];
        ay[u] = ny > cy[u] ? niy : ay[u];
        cx[u] = fmaxf(cx[u], nx);
        cy[u] = fmaxf(cy[u], ny);
      }
    }
    const int len = N - o;
    const int64_t ncell = cell + (len > 0 ? len : 0);
    const int on = o + (k + 1 < ndiag ? grow[k + 1] : N);
    float dc[RUN][2];
#pragma unroll
    for (int u = 0; u < RUN; ++u) {
      dc[u][0] = dn[u].x(); dc[u][1] = dn[u].y();
      dn[u].load(gb + min(ncell + i0 + u, C - 1) * F);   // next diagonal's cells: land during its steps
    }
    // Along one diagonal the arg-max frame never decreases with the start frame, so within this thread's run equal
    // frames are adjacent and only the run's FIRST and LAST frame can also belong to a neighbouring thread's run:
    // those two sums go through ds_add_f32 (measured: ~120 cycles per wave-instruction -- 16 per channel and
    // diagonal cost 0.72 ms of this kernel's 0.96), every frame strictly between them is exclusive to this thread
    // and takes a plain read-modify-write.
    const int nv = len - i0;                       // items u < nv are cells of this diagonal
    if (nv > 0) {
      int fx = ax[0], lx = ax[0], fy = ay[0], ly = ay[0];
#pragma unroll
      for (int u = 1; u < RUN; ++u) {
        lx = u < nv ? ax[u] : lx;
        ly = u < nv ? ay[u] : ly;
      }
      float fsx = 0.f, lsx = 0.f, fsy = 0.f, lsy = 0.f;
#pragma unroll
      for (int u = 0; u < RUN; ++u) {
        const bool v = u < nv;
        const float d0 = v ? dc[u][0] : 0.f, d1 = v ? dc[u][1] : 0.f;
        const bool f0 = ax[u] == fx, l0 = ax[u] == lx, f1 = ay[u] == fy, l1 = ay[u] == ly;
        fsx += f0 ? d0 : 0.f;
        lsx += (l0 && !f0) ? d0 : 0.f;
        fsy += f1 ? d1 : 0.f;
        lsy += (l1 && !f1) ? d1 : 0.f;
        if (v && !f0 && !l0) acc[ax[u] * MP_CH + pc * 2] += d0;
        if (v && !f1 && !l1) acc[ay[u] * MP_CH + pc * 2 + 1] += d1;
      }
      atomicAdd(&acc[fx * MP_CH + pc * 2], fsx);
      atomicAdd(&acc[lx * MP_CH + pc * 2], lsx);            // (+0 when the run has a single arg-max frame)
      atomicAdd(&acc[fy * MP_CH + pc * 2 + 1], fsy);
      atomicAdd(&acc[ly * MP_CH + pc * 2 + 1], lsy);
    }
    cell = ncell;
    o = on;
  }
  __syncthreads();
  T* dxb = dx + (int64_t)b * N * F + ch;
  for (int t = rg; t < N; t += 8) {
    float v[2] = {acc[t * MP_CH + pc * 2], acc[t * MP_CH + pc * 2 + 1]};
    Vec2<T>::store(dxb + (int64_t)t * F, v);
  }
}

// Backward of R = Ps[i] + Pe[j] in gather form: frame t collects the cells of map row t (into dPs) and of map
// column t (into dPe); one thread = 8 channels of one frame, one 16-byte load per diagonal and direction.
template <typename T>
__global__ __launch_bounds__(64) void map2d_dp_kernel(const T* __restrict__ dR, const int* __restrict__ grow, int ndiag,
                                                      T* __restrict__ dps, T* __restrict__ dpe, int64_t ldp, int N, int F,
                                                      int64_t C) {
  const int b = blockIdx.x / N, t = blockIdx.x % N;
  const T* gb = dR + (int64_t)b * C * F;
  for (int c8 = threadIdx.x * 8; c8 < F; c8 += 64 * 8) {
    float s[8], e[8], v[8];
    Vec8<T>::load(gb + (int64_t)t * F + c8, v);   // main diagonal: cell (t, t)
#pragma unroll
    for (int q = 0; q < 8; ++q) { s[q] = v[q]; e[q] = v[q]; }
    int64_t cell = N;
    int o = 0;
    for (int k = 0; k < ndiag; ++k) {
      o += grow[k];
      const int len = N - o;
      if (len <= 0) break;
      if (t < len) {            // cell (t, t + o): row t
        Vec8<T>::load(gb + (cell + t) * F + c8, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) s[q] += v[q];
      }
      if (t >= o) {             // cell (t - o, t): column t
        Vec8<T>::load(gb + (cell + t - o) * F + c8, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) e[q] += v[q];
      }
      cell += len;
    }
    Vec8<T>::store(dps + ((int64_t)b * N + t) * ldp + c8, s);
    Vec8<T>::store(dpe + ((int64_t)b * N + t) * ldp + c8, e);
  }
}

// dense <- compact: out[b, i, j, :] = cells[b, c(i,j), :] on the mask, `fill[:]` elsewhere (the value the reference
// computes for an all-zero cell: the layers' biases pushed through)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void map2d_scatter_kernel(const T* __restrict__ cells, const int* __restrict__ cell_of,
                                                            const float* __restrict__ fill, T* __restrict__ out, int N,
                                                            int W, int64_t C, int64_t ncell /* B*N*N */) {
  // one thread = one VEC-wide piece of one dense cell; threads of a cell are consecutive (32-bit index math only)
  const int wv = W / VEC;                       // pieces per cell
  const int cpb = 256 / wv > 0 ? 256 / wv : 1;  // cells per workgroup pass
  const int piece = threadIdx.x % wv, sub = threadIdx.x / wv;
  const int NN = N * N;
  for (int64_t cell0 = (int64_t)blockIdx.x * cpb; cell0 < ncell; cell0 += (int64_t)gridDim.x * cpb) {
    const int64_t cell = cell0 + sub;
    if (sub >= cpb || cell >= ncell) continue;
    const int b = (int)(cell / NN), ij = (int)(cell - (int64_t)b * NN);
    const int c = cell_of[ij];
    const int w = piece * VEC;
    T* o = out + cell * W + w;
    if (VEC == 8) {
      float v[8];
      if (c >= 0) Vec8<T>::load(cells + ((int64_t)b * C + c) * W + w, v);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fill ? fill[w + e] : 0.f;
      }
      Vec8<T>::store(o, v);
    } else {
      *o = c >= 0 ? cells[((int64_t)b * C + c) * W + w] : from_f<T>(fill ? fill[w] : 0.f);
    }
  }
}

int64_t count_cells(const int* grow_host, int ndiag, int N) {
  int64_t c = N;
  int o = 0;
  for (int k = 0; k < ndiag; ++k) {
    o += grow_host[k];
    if (N - o > 0) c += N - o;
  }
  return c;
}

}  // namespace

extern "C" int vmr_map2d_cells(const int32_t* grow_host, int ndiag, int N) {
  if ((!grow_host && ndiag > 0) || ndiag < 0 || N <= 0) return -1;
  return (int)count_cells(grow_host, ndiag, N);
}

// RUN (start frames per thread, compile-time) for N frames
#define MP_RUN_DISPATCH(N, ...)                                  \
  do {                                                           \
    if ((N) <= 64) { constexpr int RUN = 8; __VA_ARGS__; }       \
    else if ((N) <= 128) { constexpr int RUN = 16; __VA_ARGS__; } \
    else { constexpr int RUN = MP_ITEMS; __VA_ARGS__; }          \
  } while (0)

extern "C" int vmr_map2d_pool_fwd(const void* x, const void* ps, const void* pe, int64_t ldp, const int32_t* grow,
                                  const int32_t* grow_host, int ndiag, void* M, void* R, int B, int N, int F, int dtype,
                                  void* stream) {
  VMR_CHECK(x && M && ((grow && grow_host) || ndiag == 0), "vmr_map2d_pool_fwd: null pointer");
  VMR_CHECK((R == nullptr) == (ps == nullptr) && (ps == nullptr) == (pe == nullptr),
            "vmr_map2d_pool_fwd: R, ps, pe come together");
  VMR_CHECK(N > 0 && N <= MP_NMAX && F % MP_CH == 0 && (!ps || ldp % 2 == 0),
            "vmr_map2d_pool_fwd: need N <= %d, F %% %d == 0 (N %d F %d)", MP_NMAX, MP_CH, N, F);
  for (int k = 0; k < ndiag; ++k) VMR_CHECK(grow_host[k] >= 1, "vmr_map2d_pool_fwd: grow[%d] < 1", k);
  if (B == 0) return 0;
  const int64_t C = count_cells(grow_host, ndiag, N);
  const dim3 grid(B * (F / MP_CH));
  VMR_CHECK(vmr_dtype_ok(dtype), "vmr_map2d_pool_fwd: bad dtype %d", dtype);
  VMR_DISPATCH(dtype, T,
               MP_RUN_DISPATCH(N, hipLaunchKernelGGL((map2d_pool_fwd_kernel<T, RUN>), grid, dim3(256), 0, (hipStream_t)stream,
                                                     (const T*)x, (const T*)ps, (const T*)pe, ldp, grow, ndiag, (T*)M, (T*)R, N, F, C)));
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_map2d_pool_bwd(const void* x, const void* dM, const void* dR, const int32_t* grow,
                                  const int32_t* grow_host, int ndiag, void* dx, void* dps, void* dpe, int64_t ldp, int B,
                                  int N, int F, int dtype, void* stream) {
  VMR_CHECK(x && dM && dx && ((grow && grow_host) || ndiag == 0), "vmr_map2d_pool_bwd: null pointer");
  VMR_CHECK((dR == nullptr) == (dps == nullptr) && (dps == nullptr) == (dpe == nullptr),
            "vmr_map2d_pool_bwd: dR, dps, dpe come together");
  VMR_CHECK(N > 0 && N <= MP_NMAX && F % MP_CH == 0 && (!dR || (F % 8 == 0 && ldp % 8 == 0)),
            "vmr_map2d_pool_bwd: need N <= %d, F %% %d == 0 (N %d F %d)", MP_NMAX, MP_CH, N, F);
  for (int k = 0; k < ndiag; ++k) VMR_CHECK(grow_host[k] >= 1, "vmr_map2d_pool_bwd: grow[%d] < 1", k);
  if (B == 0) return 0;
  const int64_t C = count_cells(grow_host, ndiag, N);
  const size_t lds = (size_t)N * MP_CH * sizeof(float);      // the dx image (<= 40 KiB)
  const dim3 grid(B * (F / MP_CH));
  VMR_CHECK(vmr_dtype_ok(dtype), "vmr_map2d_pool_bwd: bad dtype %d", dtype);
  VMR_DISPATCH(dtype, T,
               MP_RUN_DISPATCH(N, hipLaunchKernelGGL((map2d_pool_bwd_kernel<T, RUN>), grid, dim3(256), lds, (hipStream_t)stream,
                                                     (const T*)x, (const T*)dM, grow, ndiag, (T*)dx, N, F, C)));
  VMR_LAUNCH_CHECK();
  if (dR) {
    VMR_DISPATCH(dtype, T,
                 hipLaunchKernelGGL(map2d_dp_kernel<T>, dim3(B * N), dim3(64), 0, (hipStream_t)stream, (const T*)dR, grow, ndiag,
                                    (T*)dps, (T*)dpe, ldp, N, F, C));
    VMR_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int vmr_map2d_scatter(const void* cells, const int32_t* cell_of, const float* fill, void* out, int B, int N,
                                 int W, int64_t C, int dtype, void* stream) {
  VMR_CHECK(cells && cell_of && out, "vmr_map2d_scatter: null pointer");
  const bool v8 = W % 8 == 0 && W / 8 <= 256;
  VMR_CHECK(v8 || W <= 256, "vmr_map2d_scatter: W too large for the scalar path");
  const int64_t total = (int64_t)B * N * N;         // dense cells
  if (total == 0) return 0;
  const int wv = v8 ? W / 8 : W;
  const int cpb = 256 / wv > 0 ? 256 / wv : 1;
  const int grid = (int)min((int64_t)65535, (total + cpb - 1) / cpb);
  VMR_CHECK(vmr_dtype_ok(dtype), "vmr_map2d_scatter: bad dtype %d", dtype);
  VMR_DISPATCH(dtype, T, {
    if (v8) hipLaunchKernelGGL((map2d_scatter_kernel<T, 8>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)cells, cell_of,
                               fill, (T*)out, N, W, C, total);
    else hipLaunchKernelGGL((map2d_scatter_kernel<T, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)cells, cell_of,
                            fill, (T*)out, N, W, C, total);
  });
  VMR_LAUNCH_CHECK();
  return 0;
}
