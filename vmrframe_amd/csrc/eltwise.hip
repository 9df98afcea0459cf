// eltwise.hip -- fused elementwise programs of the dual-attention gating and the
// CQAttention concat (reference models/layers.py:370-380, 424).  One pass over the
// operands instead of one torch kernel per arithmetic op; 16-byte accesses;
// HBM-bound by construction (algorithmic bytes = operands read + results written once).
#include "common.h"

namespace {

// op 0  GATE_FWD   : o0 = a*d + c*b          (a=s_score, b=s_value, c=x_score, d=x_value)   layers.py:374
// op 1  GATE_BWD   : g=a(=do); inputs b..e = s_score,s_value,x_score,x_value -> o0..o3 = d s_score, d s_value, d x_score, d x_value
// op 2  SIGGATE_FWD: a = sv [rows,2D] (scores | values); o0 = sigmoid(scores + (1-m)*-1e30) * values   layers.py:380
// op 3  SIGGATE_BWD: a = do [rows,D], b = sv [rows,2D]; o0 = dsv [rows,2D]
// op 4  CAT4_FWD   : a = C, b = c2q, c = q2c; o0 = [C, c2q, C*c2q, C*q2c] [rows,4D]            layers.py:424
// op 5  CAT4_BWD   : a = dcat [rows,4D], b = C, c = c2q, d = q2c; o0 = dC, o1 = dc2q, o2 = dq2c
template <typename T, int OP>
__global__ __launch_bounds__(256) void eltwise_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                      const T* __restrict__ c, const T* __restrict__ d,
                                                      const T* __restrict__ e5, const float* __restrict__ rowmask,
                                                      T* __restrict__ o0, T* __restrict__ o1, T* __restrict__ o2,
                                                      T* __restrict__ o3, int64_t rows, int D) {
  const int D8 = D >> 3;
  const int64_t total = rows * D8;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t r = t / D8;
    const int c8 = (int)(t - r * D8) * 8;
    const int64_t i = r * D + c8;
    float va[8], vb[8], vc[8], vd[8], ve[8], w0[8], w1[8], w2[8], w3[8];
    if (OP == 0) {
      Vec8<T>::load(a + i, va); Vec8<T>::load(b + i, vb); Vec8<T>::load(c + i, vc); Vec8<T>::load(d + i, vd);
#pragma unroll
      for (int k = 0; k < 8; ++k) w0[k] = va[k] * vd[k] + vc[k] * vb[k];
      Vec8<T>::store(o0 + i, w0);
    } else if (OP == 1) {
      Vec8<T>::load(a + i, va); Vec8<T>::load(b + i, vb); Vec8<T>::load(c + i, vc); Vec8<T>::load(d + i, vd);
      Vec8<T>::load(e5 + i, ve);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        w0[k] = va[k] * ve[k];  // d s_score = do * x_value
        w1[k] = va[k] * vd[k];  // d s_value = do * x_score
        w2[k] = va[k] * vc[k];  // d x_score = do * s_value
        w3[k] = va[k] * vb[k];  // d x_value = do * s_score
      }
      Vec8<T>::store(o0 + i, w0); Vec8<T>::store(o1 + i, w1); Vec8<T>::store(o2 + i, w2); Vec8<T>::store(o3 + i, w3);
    } else if (OP == 2) {
      const int64_t j = r * 2 * D + c8;
      Vec8<T>::load(a + j, va); Vec8<T>::load(a + j + D, vb);
      const float pen = (1.0f - rowmask[r]) * VMR_NEG_INF_MASK;
#pragma unroll
      for (int k = 0; k < 8; ++k) w0[k] = vb[k] / (1.0f + __expf(-(va[k] + pen)));
      Vec8<T>::store(o0 + i, w0);
    } else if (OP == 3) {
      const int64_t j = r * 2 * D + c8;
      Vec8<T>::load(a + i, va); Vec8<T>::load(b + j, vb); Vec8<T>::load(b + j + D, vc);
      const float pen = (1.0f - rowmask[r]) * VMR_NEG_INF_MASK;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float sg = 1.0f / (1.0f + __expf(-(vb[k] + pen)));
        w0[k] = va[k] * vc[k] * sg * (1.0f - sg);
        w1[k] = va[k] * sg;
      }
      Vec8<T>::store(o0 + j, w0); Vec8<T>::store(o0 + j + D, w1);
    } else if (OP == 4) {
      const int64_t j = r * 4 * D + c8;
      Vec8<T>::load(a + i, va); Vec8<T>::load(b + i, vb); Vec8<T>::load(c + i, vc);
#pragma unroll
      for (int k = 0; k < 8; ++k) { w0[k] = va[k] * vb[k]; w1[k] = va[k] * vc[k]; }
      Vec8<T>::store(o0 + j, va); Vec8<T>::store(o0 + j + D, vb);
      Vec8<T>::store(o0 + j + 2 * D, w0); Vec8<T>::store(o0 + j + 3 * D, w1);
    } else {
      const int64_t j = r * 4 * D + c8;
      Vec8<T>::load(b + i, vb); Vec8<T>::load(c + i, vc); Vec8<T>::load(d + i, vd);
      float g0[8], g1[8], g2[8], g3[8];
      Vec8<T>::load(a + j, g0); Vec8<T>::load(a + j + D, g1); Vec8<T>::load(a + j + 2 * D, g2);
      Vec8<T>::load(a + j + 3 * D, g3);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        w0[k] = g0[k] + g2[k] * vc[k] + g3[k] * vd[k];  // dC
        w1[k] = g1[k] + g2[k] * vb[k];                  // dc2q
        w2[k] = g3[k] * vb[k];                          // dq2c
      }
      Vec8<T>::store(o0 + i, w0); Vec8<T>::store(o1 + i, w1); Vec8<T>::store(o2 + i, w2);
    }
  }
}

// dst[i] += sum_k slab[k][i]   (split-K second stage: fp32, 16-byte accesses); dst rows of cols4
// quads with leading dimension ld4 (quads)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dst,
                                                            int nsplit, int64_t n4, int64_t stride4, int cols4,
                                                            int64_t ld4, int valid4) {
  const f32x4* s4 = reinterpret_cast<const f32x4*>(slab);
  f32x4* d4 = reinterpret_cast<f32x4*>(dst);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    if (valid4 && (int)(i % cols4) >= valid4) continue;   // zero-padded K columns of the slab: no destination
    const int64_t o = cols4 ? (i / cols4) * ld4 + (i % cols4) : i;
    f32x4 acc = d4[o];
    for (int k0 = 0; k0 < nsplit; k0 += 8) {      // eight slabs per round trip (see the rider in gemm.hip)
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = s4[(int64_t)min(k0 + u, nsplit - 1) * stride4 + i];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + u < nsplit) { acc[0] += v[u][0]; acc[1] += v[u][1]; acc[2] += v[u][2]; acc[3] += v[u][3]; }
    }
    d4[o] = acc;
  }
}

// out[m, n] = T(sum_k slab[k][m, n] + bias[n] + addend[m, n]): the second stage of a few-tile split-K product WITH its
// epilogue (bias of the layer, or the gradient of the input's other consumer) and the cast to the compute dtype --
// one launch instead of {zero-fill, atomics, add, cast}, and a fixed summation order
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_cast_kernel(const float* __restrict__ slab, int nsplit, int64_t n4,
                                                                 int cols4, const float* __restrict__ bias,
                                                                 const T* __restrict__ addend, T* __restrict__ out) {
  const f32x4* s4 = reinterpret_cast<const f32x4*>(slab);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 acc = s4[i];
    for (int k0 = 1; k0 < nsplit; k0 += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = s4[(int64_t)min(k0 + u, nsplit - 1) * n4 + i];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + u < nsplit) { acc[0] += v[u][0]; acc[1] += v[u][1]; acc[2] += v[u][2]; acc[3] += v[u][3]; }
    }
    if (bias) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + (i % cols4) * 4);
      acc[0] += bv[0]; acc[1] += bv[1]; acc[2] += bv[2]; acc[3] += bv[3];
    }
    float o[4] = {acc[0], acc[1], acc[2], acc[3]};
    if (addend) {
      float av[4];
      Vec4<T>::load(addend + i * 4, av);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += av[e];
    }
    Vec4<T>::store(out + i * 4, o);
  }
}

template <typename T>
void launch_elt(int op, dim3 grid, hipStream_t st, const void* a, const void* b, const void* c, const void* d,
                const void* e5, const float* rowmask, void* o0, void* o1, void* o2, void* o3, int64_t rows, int D) {
#define VMR_ELT(OPN)                                                                                             \
  hipLaunchKernelGGL((eltwise_kernel<T, OPN>), grid, dim3(256), 0, st, (const T*)a, (const T*)b, (const T*)c,    \
                     (const T*)d, (const T*)e5, rowmask, (T*)o0, (T*)o1, (T*)o2, (T*)o3, rows, D)
  switch (op) {
    case 0: VMR_ELT(0); break;
    case 1: VMR_ELT(1); break;
    case 2: VMR_ELT(2); break;
    case 3: VMR_ELT(3); break;
    case 4: VMR_ELT(4); break;
    default: VMR_ELT(5); break;
  }
#undef VMR_ELT
}

}  // namespace

extern "C" int vmr_eltwise(int op, const void* a, const void* b, const void* c, const void* d, const void* e,
                           const float* rowmask, void* o0, void* o1, void* o2, void* o3, int64_t rows, int D,
                           int dtype, void* stream) {
  VMR_CHECK(op >= 0 && op <= 5, "vmr_eltwise: bad op %d", op);
  VMR_CHECK(a && o0, "vmr_eltwise: null pointer");
  VMR_CHECK(D % 8 == 0, "vmr_eltwise: D %% 8 != 0");
  VMR_CHECK((op != 2 && op != 3) || rowmask, "vmr_eltwise: sigmoid gate needs the row mask");
  if (rows == 0) return 0;
  const int64_t total = rows * (D / 8);
  dim3 grid((unsigned)min((int64_t)8192, (total + 255) / 256));
  VMR_DISPATCH(dtype, T, launch_elt<T>(op, grid, (hipStream_t)stream, a, b, c, d, e, rowmask, o0, o1, o2, o3, rows, D));
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_splitk_reduce(const float* slab, float* dst, int nsplit, int64_t n, int cols, int64_t ld_dst,
                                 void* stream) {
  VMR_CHECK(slab && dst && nsplit >= 1, "vmr_splitk_reduce: bad arguments");
  VMR_CHECK(n % 4 == 0 && ((reinterpret_cast<uintptr_t>(slab) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0,
            "vmr_splitk_reduce: needs 16-byte aligned buffers and n %% 4 == 0");
  if (cols == ld_dst) cols = 0;
  VMR_CHECK(cols == 0 || (cols % 4 == 0 && ld_dst % 4 == 0 && ld_dst > 0 && n % cols == 0),
            "vmr_splitk_reduce: cols / ld_dst must be multiples of 4");
  if (n == 0) return 0;
  // ld_dst < cols: the slab rows carry zero-padded K columns (a [N, 500] weight whose product ran at K = 512): the
  // destination is dense [rows, ld_dst] and only the first ld_dst columns of each slab row are reduced
  const int valid4 = (cols && ld_dst < cols) ? (int)(ld_dst / 4) : 0;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)min((int64_t)4096, (n / 4 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, slab, dst, nsplit, n / 4, n / 4, cols / 4, ld_dst / 4, valid4);
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_splitk_reduce_cast(const float* slab, int nsplit, int64_t rows, int cols, const float* bias, const void* addend,
                                      void* out, int dtype, void* stream) {
  VMR_CHECK(slab && out && nsplit >= 1 && rows >= 0 && cols > 0, "vmr_splitk_reduce_cast: bad arguments");
  VMR_CHECK(cols % 4 == 0 && ((reinterpret_cast<uintptr_t>(slab) | reinterpret_cast<uintptr_t>(out) |
                               reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(addend)) & 7) == 0 &&
                (reinterpret_cast<uintptr_t>(slab) & 15) == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0,
            "vmr_splitk_reduce_cast: cols %% 4 == 0, 16-byte aligned slabs / bias, 8-byte aligned out / addend");
  VMR_CHECK(vmr_dtype_ok(dtype), "vmr_splitk_reduce_cast: bad dtype");
  const int64_t n4 = rows * cols / 4;
  if (n4 == 0) return 0;
  const dim3 grid((unsigned)min((int64_t)4096, (n4 + 255) / 256));
  VMR_DISPATCH(dtype, T, hipLaunchKernelGGL(splitk_reduce_cast_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, slab, nsplit, n4, cols / 4, bias,
                       (const T*)addend, (T*)out));
  VMR_LAUNCH_CHECK();
  return 0;
}
