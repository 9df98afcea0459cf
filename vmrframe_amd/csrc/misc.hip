// misc.hip -- streaming helpers of the SeqPAN path: dtype casts with padding and
// input dropout, ReLU/dropout backward with the fused bias reduction, column
// sums (bias gradients), the boundary-label cross-entropy (reference
// models/loss.py:43-54) and the fused AdamW / grad-norm optimizer step
// (utils/utils.py:87-97, main.py:95).  All HBM-bound: 16-B accesses, grid-stride.
#include "common.h"

namespace {

// ------------------------------------------------------------------- cast
// dst[r, 0:cols] = drop(src[r, 0:cols]); dst[r, cols:ld_dst] = 0
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int64_t rows,
                                                   int cols, int64_t ld_src, int64_t ld_dst, float drop_p,
                                                   uint32_t seed0, const uint32_t* __restrict__ step) {
  const uint32_t seed = vmr_seed(seed0, step);
  const uint32_t thresh = vmr_drop_thresh(drop_p);
  const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const int64_t total = rows * ld_dst;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / ld_dst;
    const int c = (int)(i - r * ld_dst);
    float v = 0.f;
    if (c < cols) {
      v = to_f<TS>(src[r * ld_src + c]);
      if (drop_p > 0.f) v = vmr_keep(seed, (uint64_t)r * cols + c, thresh) ? v * dscale : 0.f;
    }
    dst[i] = from_f<TD>(v);
  }
}

// 8 destination columns per thread (ld_dst % 8 == 0; source rows 16-byte aligned where SRC_VEC): 16-byte stores,
// one hash pair per 8 elements -- the [tokens, D] dropout copies of the CQ score path and the [B*T, V] input cast
template <typename TS, typename TD, bool SRC_VEC>
__global__ __launch_bounds__(256) void cast8_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int64_t rows, int cols,
                                                    int64_t ld_src, int64_t ld_dst, float drop_p, uint32_t seed0,
                                                    const uint32_t* __restrict__ step) {
  const uint32_t seed = vmr_seed(seed0, step);
  const uint32_t thresh = vmr_drop_thresh(drop_p);
  const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const int cpr = (int)(ld_dst / 8);
  const int64_t total = rows * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cpr;
    const int c = (int)(i - r * cpr) * 8;
    float v[8];
    if (SRC_VEC && c + 8 <= cols) Vec8<TS>::load(src + r * ld_src + c, v);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (c + e < cols) ? to_f<TS>(src[r * ld_src + c + e]) : 0.f;
    }
    if (drop_p > 0.f) {
      const uint64_t idx = (uint64_t)r * cols + c;
      if ((idx & 3) == 0) {
        const uint32_t keep = vmr_keep8(seed, idx, thresh);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ((keep >> e) & 1) ? v[e] * dscale : 0.f;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = vmr_keep(seed, idx + e, thresh) ? v[e] * dscale : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) if (c + e >= cols) v[e] = 0.f;
    }
    Vec8<TD>::store(dst + r * ld_dst + c, v);
  }
}

// ------------------------------------------------ relu/dropout bwd + bias grad
// MODE 0: db[c] += sum_r dy[r,c]                                   (plain bias)
// MODE 1: dz = dy * scale * (h > 0)            ; db += colsum(dz)  (ReLU [+dropout]: h is the
//         saved post-dropout ReLU output, so h > 0 <=> positive AND kept)
// MODE 2: dz = dy * scale * keep(seed, r*D+c)  ; db += colsum(dz)  (dropout without ReLU)
// block = 256 threads = 32 column-groups (8 cols) x 8 row lanes; columns tile = 256
template <typename T, int MODE>
__global__ __launch_bounds__(256) void relu_bwd_bias_kernel(const T* __restrict__ dy, const T* __restrict__ h,
                                                            T* __restrict__ dz, float* __restrict__ db, int64_t rows,
                                                            int D, int64_t ld, float scale, int rows_per_block,
                                                            float drop_p, uint32_t seed0,
                                                            const uint32_t* __restrict__ step,
                                                            float* __restrict__ db2, float db_scale) {
  __shared__ float red[8][256];
  const uint32_t seed = vmr_seed(seed0, step);
  const uint32_t thresh = vmr_drop_thresh(drop_p);
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 256 + cg * 8;
  const int64_t rbeg = (int64_t)blockIdx.y * rows_per_block;
  const int64_t rend = min(rows, rbeg + rows_per_block);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < D) {
    // RB rows per iteration: their loads (index-clamped, unconditional) are issued together -- one row per iteration was a
    // chain of dependent HBM round trips (3.2 TB/s)
#ifndef VMR_RBB_RB
#define VMR_RBB_RB 4
#endif
    constexpr int RB = VMR_RBB_RB;
    typedef __attribute__((ext_vector_type(8))) T TV8;
    for (int64_t r0 = rbeg + rl; r0 < rend; r0 += 8 * RB) {
      TV8 gv[RB], hv8[RB];
      uint32_t hb[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int64_t r = min(r0 + 8 * u, rows - 1);
        gv[u] = *reinterpret_cast<const TV8*>(dy + r * ld + c0);
        if (MODE == 1 || MODE == 4) hv8[u] = *reinterpret_cast<const TV8*>(h + r * ld + c0);
        if (MODE == 3) hb[u] = reinterpret_cast<const unsigned char*>(h)[r * (D >> 3) + (c0 >> 3)];
      }
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int64_t r = r0 + 8 * u;
        if (r >= rend) continue;
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = (float)gv[u][e];
        if (MODE == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = (float)hv8[u][e] > 0.f ? g[e] * scale : 0.f;
          Vec8<T>::store(dz + r * ld + c0, g);
        } else if (MODE == 3) {   // h = bit matrix uint8 [rows][D/8]: this thread's 8 columns are one byte
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = ((hb[u] >> e) & 1) ? g[e] * scale : 0.f;
          Vec8<T>::store(dz + r * ld + c0, g);
        } else if (MODE == 2 || MODE == 4) {
          const uint32_t keep = vmr_keep8(seed, (uint64_t)r * D + c0, thresh);
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = ((keep >> e) & 1) ? g[e] * scale : 0.f;
          if (MODE == 4) {   // + the gradient of the tensor's other consumer (h), one pass instead of a separate add
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] += (float)hv8[u][e];
          }
          Vec8<T>::store(dz + r * ld + c0, g);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += g[e];
      }
    }
  }
  if (!db) return;
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = acc[e];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < D) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][threadIdx.x];
    atomicAdd(&db[c], s * db_scale);
    if (db2) atomicAdd(&db2[c], s);
  }
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ m, int64_t n, float drop_p,
                                                           uint32_t seed) {
  const uint32_t thresh = vmr_drop_thresh(drop_p);
  const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    m[i] = (drop_p > 0.f ? vmr_keep(seed, (uint64_t)i, thresh) : true) ? dscale : 0.f;
}

// ------------------------------------------------------- embedding gather
// out[i,:] = table[idx[i],:]  (reference WordEmbedding / CharacterEmbedding lookups,
// models/layers.py:42-48,66); backward: dtable[idx[i],:] += dout[i,:] for idx != padding_idx
// (nn.Embedding(padding_idx=0) semantics).  One wave per looked-up row; fp32; float atomics.
__global__ __launch_bounds__(256) void gather_rows_kernel(const int64_t* __restrict__ idx,
                                                          const float* __restrict__ table, float* __restrict__ out,
                                                          int64_t n, int D, int64_t nrows) {
  const int lane = threadIdx.x & 63;
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
    int64_t r = idx[i];
    r = r < 0 ? 0 : (r >= nrows ? nrows - 1 : r);   // never read outside the table
    for (int c = lane; c < D; c += 64) out[i * D + c] = table[r * D + c];
  }
}

__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const int64_t* __restrict__ idx,
                                                               const float* __restrict__ dout,
                                                               float* __restrict__ dtable, int64_t n, int D,
                                                               int64_t nrows, int64_t padding_idx) {
  const int lane = threadIdx.x & 63;
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
    const int64_t r = idx[i];
    if (r == padding_idx || r < 0 || r >= nrows) continue;
    for (int c = lane; c < D; c += 64) atomicAdd(&dtable[r * D + c], dout[i * D + c]);
  }
}

// -------------------------------------------- WordEmbedding (reference models/layers.py:28-48)
// out[i, 0:wd] = drop(row(ids[i])), row(0) = pad_vec, row(1) = unk_vec, row(k >= 2) = glove_vec[k - 2]: the reference
// concatenates the three parameters into one table on every forward (4.8 MB at 4000 words) and gathers from that;
// here the three pieces are indexed in place, the embedding dropout (layers.py:46) is applied on the way and the
// result lands in the compute dtype directly inside the [words, ldo] matrix that feeds query_conv1d.  Columns
// [zero_from, zero_to) of every row are zeroed as well (the K padding of that GEMM); the columns between belong to the
// character CNN (charcnn.hip), which writes them itself.
template <typename T>
__global__ __launch_bounds__(256) void word_embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ pad,
                                                             const float* __restrict__ unk, const float* __restrict__ glove,
                                                             T* __restrict__ out, int64_t n, int wd, int64_t nglove, int64_t ldo,
                                                             int zero_from, int zero_to, float drop_p, uint32_t seed0,
                                                             const uint32_t* __restrict__ step) {
  const uint32_t seed = vmr_seed(seed0, step);
  const uint32_t thresh = vmr_drop_thresh(drop_p);
  const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const int q4 = wd / 4, z4 = (zero_to - zero_from) / 4, per = q4 + z4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n * per; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / per;
    const int local = (int)(i - row * per);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    int col;
    if (local < q4) {
      col = local * 4;
      int64_t id = ids[row];
      id = id < 0 ? 0 : (id > nglove + 1 ? nglove + 1 : id);           // never read outside the pieces
      const float* src = id == 0 ? pad : (id == 1 ? unk : glove + (id - 2) * wd);
      Vec4<float>::load(src + col, v);
      if (drop_p > 0.f) {
        const uint2 h = vmr_hash4(seed, (uint64_t)(row * wd + col) >> 2);
        v[0] = (h.x & 0xFFFFu) >= thresh ? v[0] * dscale : 0.f;
        v[1] = (h.x >> 16) >= thresh ? v[1] * dscale : 0.f;
        v[2] = (h.y & 0xFFFFu) >= thresh ? v[2] * dscale : 0.f;
        v[3] = (h.y >> 16) >= thresh ? v[3] * dscale : 0.f;
      }
    } else {
      col = zero_from + (local - q4) * 4;
    }
    Vec4<T>::store(out + row * ldo + col, v);
  }
}

// the only trainable row is unk_vec (pad_vec and glove_vec are frozen, layers.py:33,37): dunk[c] += sum over the words
// with id 1 of the dropped gradient.  One wave per word; unknown words are rare, so the float atomics are too.
template <typename T>
__global__ __launch_bounds__(256) void word_embed_bwd_kernel(const int64_t* __restrict__ ids, const T* __restrict__ dout,
                                                             float* __restrict__ dunk, int64_t n, int wd, int64_t ldo, float drop_p,
                                                             uint32_t seed0, const uint32_t* __restrict__ step) {
  const uint32_t seed = vmr_seed(seed0, step);
  const uint32_t thresh = vmr_drop_thresh(drop_p);
  const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const int lane = threadIdx.x & 63;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += (int64_t)gridDim.x * 4) {
    if (ids[row] != 1) continue;
    for (int c = lane; c < wd; c += 64) {
      float g = to_f<T>(dout[row * ldo + c]);
      if (drop_p > 0.f) g = vmr_keep(seed, (uint64_t)(row * wd + c), thresh) ? g * dscale : 0.f;
      atomicAdd(&dunk[c], g);
    }
  }
}

// -------------------------------------------- boundary-label cross-entropy
// loss = mean_b( -sum_t ys*log_softmax(zs) ) + same for the end logits.
// grid = 2*B waves (one wave per (which, b) row); T <= 64*16.
__global__ __launch_bounds__(256) void soft_ce_fwd_kernel(const float* __restrict__ zs, const float* __restrict__ ze,
                                                          const float* __restrict__ ys, const float* __restrict__ ye,
                                                          float* __restrict__ loss, float* __restrict__ lse, int B,
                                                          int T) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= 2 * B) return;
  const int which = row / B, b = row - which * B;
  const float* z = (which ? ze : zs) + (int64_t)b * T;
  const float* y = (which ? ye : ys) + (int64_t)b * T;
  float mx = -INFINITY;
  for (int t = lane; t < T; t += 64) mx = fmaxf(mx, z[t]);
  mx = wave_max(mx);
  float se = 0.f, sy = 0.f, syz = 0.f;
  for (int t = lane; t < T; t += 64) {
    se += __expf(z[t] - mx);
    sy += y[t];
    syz += y[t] * z[t];
  }
  se = wave_sum(se); sy = wave_sum(sy); syz = wave_sum(syz);
  const float l = mx + __logf(se);
  if (lane == 0) {
    lse[row] = l;
    atomicAdd(loss, (l * sy - syz) / (float)B);  // -sum y*(z - lse)
  }
}

// dz = dloss/B * (softmax(z)*sum(y) - y)
__global__ __launch_bounds__(256) void soft_ce_bwd_kernel(const float* __restrict__ zs, const float* __restrict__ ze,
                                                          const float* __restrict__ ys, const float* __restrict__ ye,
                                                          const float* __restrict__ lse, const float* __restrict__ dloss,
                                                          float* __restrict__ dzs, float* __restrict__ dze, int B,
                                                          int T) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= 2 * B) return;
  const int which = row / B, b = row - which * B;
  const float* z = (which ? ze : zs) + (int64_t)b * T;
  const float* y = (which ? ye : ys) + (int64_t)b * T;
  float* dz = (which ? dze : dzs) + (int64_t)b * T;
  float sy = 0.f;
  for (int t = lane; t < T; t += 64) sy += y[t];
  sy = wave_sum(sy);
  const float l = lse[row], g = dloss[0] / (float)B;
  for (int t = lane; t < T; t += 64) dz[t] = g * (__expf(z[t] - l) * sy - y[t]);
}

// --------------------------------------------------------------- optimizer
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, float* __restrict__ out, int64_t n) {
  __shared__ float red[4];
  float s = 0.f;
  const int64_t n4 = n >> 2;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  // four independent 16-byte loads in flight per thread (a single dependent load per iteration ran at 3.8 TB/s)
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f32x4 v0 = g4[i], v1 = g4[i + stride], v2 = g4[i + 2 * stride], v3 = g4[i + 3 * stride];
    s += v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2] + v0[3] * v0[3];
    s1 += v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2] + v1[3] * v1[3];
    s2 += v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2] + v2[3] * v2[3];
    s3 += v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2] + v3[3] * v3[3];
  }
  for (; i < n4; i += stride) {
    const f32x4 v = g4[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  s += s1 + s2 + s3;
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// torch.optim.AdamW semantics (decoupled decay, bias correction, eps outside sqrt)
// with clip_grad_norm_'s scale min(1, max_norm/(norm+1e-6)) folded into the read of g.
// Loss scaling (16-bit activation gradients, VMR_F16): `scale` = device float[1] holding S; g and gnorm_sq were produced
// from S * loss, so both are divided by S here; a non-finite norm (an inf / NaN anywhere in the gradients) SKIPS the
// update -- p, m, v and the mirror stay as they are; loss_scale_update_kernel below then halves S and holds the step count.
template <typename TM>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    const uint8_t* __restrict__ decay, TM* __restrict__ pb,
                                                    const float* __restrict__ gnorm_sq, float max_norm, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1,
                                                    float bc2, const int* __restrict__ step_dev,
                                                    float warmup_steps, float total_steps,
                                                    const float* __restrict__ scale, int64_t n) {
  if (step_dev) {   // device-resident step: a captured hipGraph replays with the current lr / bias corrections
    const float s = (float)step_dev[0];
    const float t = s + 1.f;
    bc1 = 1.f - powf(beta1, t);
    bc2 = 1.f - powf(beta2, t);
    if (total_steps > 0.f)
      lr *= s < warmup_steps ? s / fmaxf(1.f, warmup_steps)
                             : fmaxf(0.f, total_steps - s) / fmaxf(1.f, total_steps - warmup_steps);
  }
  float clip = scale ? 1.f / scale[0] : 1.f;
  if (gnorm_sq) {
    const float nrm = sqrtf(gnorm_sq[0]) * clip;
    if (scale && !(nrm <= 3.0e38f)) return;          // inf / NaN: skipped step (uniform over the grid)
    if (max_norm > 0.f) clip *= fminf(1.f, max_norm / (nrm + 1e-6f));
  }
  auto upd = [&](float gi, float& pi, float& mi, float& vi, bool dec) {
    gi *= clip;
    if (dec) pi *= 1.f - lr * wd;
    mi = beta1 * mi + (1.f - beta1) * gi;
    vi = beta2 * vi + (1.f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    pi -= (lr / bc1) * (mi / denom);
  };
  // four elements per thread and pass: 16-byte accesses on the four fp32 streams, 8 bytes on the 16-bit mirror (as one
  // element per thread the kernel issued four times the memory instructions for the same 31 bytes per parameter)
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                     reinterpret_cast<uintptr_t>(v)) & 15) == 0 && (reinterpret_cast<uintptr_t>(decay) & 3) == 0 &&
                   (!pb || (reinterpret_cast<uintptr_t>(pb) & 7) == 0);
  const int64_t n4 = vec ? n >> 2 : 0;
  typedef __attribute__((ext_vector_type(4))) TM TM4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 g4 = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 p4 = reinterpret_cast<f32x4*>(p)[i], m4 = reinterpret_cast<f32x4*>(m)[i], v4 = reinterpret_cast<f32x4*>(v)[i];
    const uint32_t d4 = reinterpret_cast<const uint32_t*>(decay)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = p4[e], me = m4[e], ve = v4[e];
      upd(g4[e], pe, me, ve, ((d4 >> (8 * e)) & 0xFFu) != 0u);
      p4[e] = pe; m4[e] = me; v4[e] = ve;
    }
    reinterpret_cast<f32x4*>(m)[i] = m4;
    reinterpret_cast<f32x4*>(v)[i] = v4;
    reinterpret_cast<f32x4*>(p)[i] = p4;
    if (pb) {
      TM4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = from_f<TM>(p4[e]);
      reinterpret_cast<TM4*>(pb)[i] = o;
    }
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float pi = p[i], mi = m[i], vi = v[i];
    upd(g[i], pi, mi, vi, decay[i] != 0);
    m[i] = mi; v[i] = vi; p[i] = pi;
    if (pb) pb[i] = from_f<TM>(pi);
  }
}

// The dynamic loss scaler's bookkeeping, on the device so a captured step replays it: a non-finite gradient norm halves
// S and resets the streak (the optimizer step was skipped and the step counter holds); otherwise the step counter advances
// and after `growth_interval` clean steps in a row S doubles (capped at 2^24).  state = {S, clean streak}.
__global__ void loss_scale_update_kernel(float* __restrict__ state, const float* __restrict__ gnorm_sq, int* __restrict__ step_dev,
                                         int growth_interval, float min_scale) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float S = state[0];
  const float nrm = sqrtf(gnorm_sq[0]) / S;
  if (!(nrm <= 3.0e38f)) {
    state[0] = fmaxf(S * 0.5f, min_scale);
    state[1] = 0.f;
  } else {
    if (step_dev) step_dev[0] += 1;
    const float streak = state[1] + 1.f;
    if (growth_interval > 0 && streak >= (float)growth_interval) {
      state[0] = fminf(S * 2.f, 16777216.f);
      state[1] = 0.f;
    } else {
      state[1] = streak;
    }
  }
}

template <typename TS>
int launch_cast(const void* src, void* dst, int dst_dtype, int64_t rows, int cols, int64_t ld_src, int64_t ld_dst,
                float drop_p, uint32_t seed, const uint32_t* step, hipStream_t st) {
  const int64_t total = rows * ld_dst;
  const int spv = 16 / (int)sizeof(TS);   // source elements per 16 bytes
  const bool dst_ok = ld_dst % 8 == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
  if (dst_ok) {   // 8 columns per thread, 16-byte stores (and loads where the source rows are 16-byte aligned)
    const bool src_vec = ld_src % spv == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    const int g8 = (int)min((int64_t)8192, (total / 8 + 255) / 256);
#define VMR_CAST8(TD, SV)                                                                                          \
  hipLaunchKernelGGL((cast8_kernel<TS, TD, SV>), dim3(g8), dim3(256), 0, st, (const TS*)src, (TD*)dst, rows, cols, \
                     ld_src, ld_dst, drop_p, seed, step)
    VMR_DISPATCH(dst_dtype, T, { if (src_vec) VMR_CAST8(T, true); else VMR_CAST8(T, false); });
#undef VMR_CAST8
    return 0;
  }
  const int grid = (int)min((int64_t)8192, (total + 255) / 256);
  VMR_DISPATCH(dst_dtype, T, hipLaunchKernelGGL((cast_kernel<TS, T>), dim3(grid), dim3(256), 0, st, (const TS*)src, (T*)dst, rows, cols,
                       ld_src, ld_dst, drop_p, seed, step));
  return 0;
}

}  // namespace

// ---- batched bf16 transpose: dst[c][r] = src[r][c] for every item of the list in one launch (blockIdx.z = item).
// Keeps a K-major copy of every weight matrix next to its bf16 mirror, so the dX = dz.W products run as x.W'^T with
// the row-major-weight kernel (k-contiguous fragment reads, 16-byte epilogue) instead of the transposed-read layout.
struct TransItems {
  vmr_transpose_item_t it[VMR_TRANSPOSE_MAX_ITEMS];
};

__global__ __launch_bounds__(256) void transpose_batched_kernel(TransItems items) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[64][72];
  const vmr_transpose_item_t& q = items.it[blockIdx.z];
  const int tr = (q.rows + 63) / 64, tc = (q.cols + 63) / 64;
  const unsigned short* src = reinterpret_cast<const unsigned short*>(q.src);
  unsigned short* dst = reinterpret_cast<unsigned short*>(q.dst);
  // 16-byte accesses on both sides when the matrix is made of whole 64 x 64 tiles and 16-byte aligned (every weight
  // matrix of the model): a tile row is 8 chunks of 8 elements; 8 lanes move one 128-byte row segment per instruction,
  // and the transposition is 8 two-byte LDS reads per stored chunk.  (4-byte accesses ran at 3.9 TB/s.)
  if (((q.rows | q.cols) & 63) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
    for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
      const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int id = threadIdx.x + 256 * k, r = id >> 3, ch = id & 7;
        const uint4 w = *reinterpret_cast<const uint4*>(src + (int64_t)(r0 + r) * q.cols + c0 + ch * 8);
        *reinterpret_cast<uint4*>(&tile[r][ch * 8]) = w;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int id = threadIdx.x + 256 * k, c = id >> 3, rc = id & 7;
        unsigned short v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[rc * 8 + e][c];
        uint4 w;
        w.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16); w.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
        w.z = (uint32_t)v[4] | ((uint32_t)v[5] << 16); w.w = (uint32_t)v[6] | ((uint32_t)v[7] << 16);
        *reinterpret_cast<uint4*>(dst + (int64_t)(c0 + c) * q.rows + r0 + rc * 8) = w;
      }
      __syncthreads();
    }
    return;
  }
  // 4-byte accesses on both sides when rows and cols are even (every weight matrix of the model): a wave moves two
  // 128-byte row segments per instruction
  const bool even = ((q.rows | q.cols) & 1) == 0;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // pair of columns, row within a pass of 8
  for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
    const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = r0 + ty + 8 * k, c = c0 + 2 * tx;
      unsigned short a = 0, b = 0;
      if (r < q.rows) {
        if (even && c + 1 < q.cols) {
          const uint32_t w = *reinterpret_cast<const uint32_t*>(src + (int64_t)r * q.cols + c);
          a = (unsigned short)(w & 0xFFFFu); b = (unsigned short)(w >> 16);
        } else {
          if (c < q.cols) a = src[(int64_t)r * q.cols + c];
          if (c + 1 < q.cols) b = src[(int64_t)r * q.cols + c + 1];
        }
      }
      tile[ty + 8 * k][2 * tx] = a;
      tile[ty + 8 * k][2 * tx + 1] = b;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = c0 + ty + 8 * k, r = r0 + 2 * tx;      // dst row c, elements r, r+1
      if (c < q.cols) {
        const unsigned short a = tile[2 * tx][ty + 8 * k], b = tile[2 * tx + 1][ty + 8 * k];
        if (even && r + 1 < q.rows) *reinterpret_cast<uint32_t*>(dst + (int64_t)c * q.rows + r) = (uint32_t)a | ((uint32_t)b << 16);
        else {
          if (r < q.rows) dst[(int64_t)c * q.rows + r] = a;
          if (r + 1 < q.rows) dst[(int64_t)c * q.rows + r + 1] = b;
        }
      }
    }
    __syncthreads();
  }
}

extern "C" int vmr_transpose_batched(const vmr_transpose_item_t* items, int n, void* stream) {
  VMR_CHECK(items || n == 0, "vmr_transpose_batched: null items");
  for (int base = 0; base < n; base += VMR_TRANSPOSE_MAX_ITEMS) {
    const int cnt = min(VMR_TRANSPOSE_MAX_ITEMS, n - base);
    TransItems ti;
    memset(&ti, 0, sizeof(ti));
    int64_t maxt = 0;
    for (int i = 0; i < cnt; ++i) {
      const vmr_transpose_item_t& q = items[base + i];
      VMR_CHECK(q.src && q.dst && q.rows > 0 && q.cols > 0, "vmr_transpose_batched: bad item %d", base + i);
      ti.it[i] = q;
      maxt = max(maxt, (int64_t)((q.rows + 63) / 64) * ((q.cols + 63) / 64));
    }
    hipLaunchKernelGGL(transpose_batched_kernel, dim3((unsigned)min(maxt, (int64_t)1024), 1, cnt), dim3(256), 0, (hipStream_t)stream, ti);
    VMR_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int vmr_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t rows, int cols,
                        int64_t ld_src, int64_t ld_dst, float drop_p, uint32_t drop_seed,
                        const uint32_t* drop_step, void* stream) {
  VMR_CHECK(src && dst, "vmr_cast: null pointer");
  VMR_CHECK(ld_src >= cols && ld_dst >= cols, "vmr_cast: leading dim < cols");
  if (rows == 0 || ld_dst == 0) return 0;
  VMR_DISPATCH(src_dtype, T, launch_cast<T>(src, dst, dst_dtype, rows, cols, ld_src, ld_dst, drop_p, drop_seed, drop_step, (hipStream_t)stream));
  VMR_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static void launch_rbb(int mode, dim3 grid, hipStream_t st, const void* dy, const void* h, void* dz, float* db,
                       int64_t rows, int D, int64_t ld, float scale, int rpb, float drop_p, uint32_t seed,
                       const uint32_t* step, float* db2, float db_scale) {
  if (mode == 0)
    hipLaunchKernelGGL((relu_bwd_bias_kernel<T, 0>), grid, dim3(256), 0, st, (const T*)dy, (const T*)h, (T*)dz, db, rows,
                       D, ld, scale, rpb, drop_p, seed, step, db2, db_scale);
  else if (mode == 1)
    hipLaunchKernelGGL((relu_bwd_bias_kernel<T, 1>), grid, dim3(256), 0, st, (const T*)dy, (const T*)h, (T*)dz, db, rows,
                       D, ld, scale, rpb, drop_p, seed, step, db2, db_scale);
  else if (mode == 3)
    hipLaunchKernelGGL((relu_bwd_bias_kernel<T, 3>), grid, dim3(256), 0, st, (const T*)dy, (const T*)h, (T*)dz, db, rows,
                       D, ld, scale, rpb, drop_p, seed, step, db2, db_scale);
  else if (mode == 4)
    hipLaunchKernelGGL((relu_bwd_bias_kernel<T, 4>), grid, dim3(256), 0, st, (const T*)dy, (const T*)h, (T*)dz, db, rows,
                       D, ld, scale, rpb, drop_p, seed, step, db2, db_scale);
  else
    hipLaunchKernelGGL((relu_bwd_bias_kernel<T, 2>), grid, dim3(256), 0, st, (const T*)dy, (const T*)h, (T*)dz, db, rows,
                       D, ld, scale, rpb, drop_p, seed, step, db2, db_scale);
}

extern "C" int vmr_relu_bwd_bias(int mode, const void* dy, const void* h, void* dz, float* db, int64_t rows, int D,
                                 int64_t ld, float scale, int dtype, float drop_p, uint32_t drop_seed,
                                 const uint32_t* drop_step, float* db2, float db_scale, void* stream) {
  VMR_CHECK(mode >= 0 && mode <= 4, "vmr_relu_bwd_bias: bad mode %d", mode);
  VMR_CHECK(!db2 || db, "vmr_relu_bwd_bias: db2 without db");
  if (db_scale == 0.f) db_scale = 1.f;
  VMR_CHECK(dy && (db || mode != 0), "vmr_relu_bwd_bias: null pointer");
  VMR_CHECK(D % 8 == 0 && ld % 8 == 0, "vmr_relu_bwd_bias: D and ld must be multiples of 8 (D=%d)", D);
  VMR_CHECK((mode != 1 && mode != 3 && mode != 4) || h, "vmr_relu_bwd_bias: modes 1/3/4 need h");
  VMR_CHECK(mode == 0 || dz, "vmr_relu_bwd_bias: modes 1/2 need dz");
  if (rows == 0) return 0;
  const int gx = cdiv(D, 256);
  int gy = (int)min((int64_t)(1024 / gx > 0 ? 1024 / gx : 1), (rows + 63) / 64);
  const int rpb = (int)((rows + gy - 1) / gy);
  gy = (int)((rows + rpb - 1) / rpb);
  VMR_DISPATCH(dtype, T, launch_rbb<T>(mode, dim3(gx, gy), (hipStream_t)stream, dy, h, dz, db, rows, D, ld, scale, rpb, drop_p,
                       drop_seed, drop_step, db2, db_scale));
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_dropout_mask(float* m, int64_t n, float drop_p, uint32_t seed, void* stream) {
  VMR_CHECK(m, "vmr_dropout_mask: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((int)min((int64_t)4096, (n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, m, n, drop_p, seed);
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_soft_ce_fwd(const float* zs, const float* ze, const float* ys, const float* ye, float* loss,
                               float* lse, int B, int T, void* stream) {
  VMR_CHECK(zs && ze && ys && ye && loss && lse, "vmr_soft_ce_fwd: null pointer");
  if (B == 0) return 0;
  hipLaunchKernelGGL(soft_ce_fwd_kernel, dim3(cdiv(2 * B, 4)), dim3(256), 0, (hipStream_t)stream, zs, ze, ys, ye, loss,
                     lse, B, T);
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_soft_ce_bwd(const float* zs, const float* ze, const float* ys, const float* ye, const float* lse,
                               const float* dloss, float* dzs, float* dze, int B, int T, void* stream) {
  VMR_CHECK(zs && ze && ys && ye && lse && dloss && dzs && dze, "vmr_soft_ce_bwd: null pointer");
  if (B == 0) return 0;
  hipLaunchKernelGGL(soft_ce_bwd_kernel, dim3(cdiv(2 * B, 4)), dim3(256), 0, (hipStream_t)stream, zs, ze, ys, ye, lse,
                     dloss, dzs, dze, B, T);
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_sumsq(const float* g, float* out, int64_t n, void* stream) {
  VMR_CHECK(g && out, "vmr_sumsq: null pointer");
  VMR_CHECK((reinterpret_cast<uintptr_t>(g) & 15) == 0, "vmr_sumsq: g must be 16-byte aligned");
  if (n == 0) return 0;
  hipLaunchKernelGGL(sumsq_kernel, dim3((int)min((int64_t)2048, (n / 4 + 255) / 256 + 1)), dim3(256), 0,
                     (hipStream_t)stream, g, out, n);
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_adamw(float* p, const float* g, float* m, float* v, const uint8_t* decay, void* p16, int p16_dtype,
                         const float* gnorm_sq, float max_norm, float lr, float beta1, float beta2, float eps,
                         float wd, int step, const int* step_dev, float warmup_steps, float total_steps,
                         const float* loss_scale, int64_t n, void* stream) {
  VMR_CHECK(p && g && m && v && decay, "vmr_adamw: null pointer");
  VMR_CHECK(step >= 1 || step_dev, "vmr_adamw: step starts at 1");
  VMR_CHECK(!p16 || vmr_dtype_16(p16_dtype), "vmr_adamw: the compute-dtype mirror is bf16 or f16 (got %d)", p16_dtype);
  VMR_CHECK(!loss_scale || gnorm_sq, "vmr_adamw: loss scaling needs the gradient norm (the overflow check)");
  if (n == 0) return 0;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  const dim3 grid((int)min((int64_t)4096, (n / 4 + 255) / 256 + 1));
  VMR_DISPATCH16(p16_dtype, TM,
                 hipLaunchKernelGGL(adamw_kernel<TM>, grid, dim3(256), 0, (hipStream_t)stream, p, g, m, v, decay, (TM*)p16, gnorm_sq,
                                    max_norm, lr, beta1, beta2, eps, wd, bc1, bc2, step_dev, warmup_steps, total_steps, loss_scale, n));
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_loss_scale_update(float* state, const float* gnorm_sq, int* step_dev, int growth_interval, float min_scale,
                                     void* stream) {
  VMR_CHECK(state && gnorm_sq, "vmr_loss_scale_update: null pointer");
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, gnorm_sq, step_dev,
                     growth_interval, min_scale > 0.f ? min_scale : 1.f);
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_word_embedding_fwd(const int64_t* ids, const float* pad_vec, const float* unk_vec, const float* glove_vec,
                                      void* out, int64_t n, int wd, int64_t nglove, int64_t ldo, int zero_from, int zero_to,
                                      int dtype, float drop_p, uint32_t drop_seed, const uint32_t* drop_step, void* stream) {
  VMR_CHECK(ids && pad_vec && unk_vec && glove_vec && out, "vmr_word_embedding_fwd: null pointer");
  VMR_CHECK(vmr_dtype_ok(dtype), "vmr_word_embedding_fwd: bad dtype");
  VMR_CHECK(wd > 0 && wd % 4 == 0 && ldo % 4 == 0 && ldo >= wd && nglove >= 0, "vmr_word_embedding_fwd: wd / ldo must be multiples of 4");
  VMR_CHECK(zero_to >= zero_from && (zero_to - zero_from) % 4 == 0 && zero_from % 4 == 0 && zero_to <= ldo && (zero_to == zero_from || zero_from >= wd),
            "vmr_word_embedding_fwd: bad zero range");
  VMR_CHECK((((uintptr_t)pad_vec | (uintptr_t)unk_vec | (uintptr_t)glove_vec | (uintptr_t)out) & 15) == 0, "vmr_word_embedding_fwd: 16-byte alignment");
  if (n == 0) return 0;
  const int64_t work = n * (wd / 4 + (zero_to - zero_from) / 4);
  const int grid = (int)min((int64_t)4096, (work + 255) / 256);
  VMR_DISPATCH(dtype, T, hipLaunchKernelGGL(word_embed_fwd_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, ids, pad_vec, unk_vec, glove_vec,
                       (T*)out, n, wd, nglove, ldo, zero_from, zero_to, drop_p, drop_seed, drop_step));
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_word_embedding_bwd(const int64_t* ids, const void* dout, float* dunk, int64_t n, int wd, int64_t ldo, int dtype,
                                      float drop_p, uint32_t drop_seed, const uint32_t* drop_step, void* stream) {
  VMR_CHECK(ids && dout && dunk, "vmr_word_embedding_bwd: null pointer");
  VMR_CHECK(vmr_dtype_ok(dtype), "vmr_word_embedding_bwd: bad dtype");
  if (n == 0) return 0;
  const int grid = (int)min((int64_t)1024, (n + 3) / 4);
  VMR_DISPATCH(dtype, T, hipLaunchKernelGGL(word_embed_bwd_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, ids, (const T*)dout, dunk, n,
                       wd, ldo, drop_p, drop_seed, drop_step));
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_embedding_fwd(const int64_t* idx, const float* table, float* out, int64_t n, int D, int64_t nrows,
                                 void* stream) {
  VMR_CHECK(idx && table && out && nrows > 0, "vmr_embedding_fwd: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((int)min((int64_t)4096, (n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     idx, table, out, n, D, nrows);
  VMR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmr_embedding_bwd(const int64_t* idx, const float* dout, float* dtable, int64_t n, int D, int64_t nrows,
                                 int64_t padding_idx, void* stream) {
  VMR_CHECK(idx && dout && dtable && nrows > 0, "vmr_embedding_bwd: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((int)min((int64_t)4096, (n + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, idx, dout, dtable, n, D, nrows, padding_idx);
  VMR_LAUNCH_CHECK();
  return 0;
}
