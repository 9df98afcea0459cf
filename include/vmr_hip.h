/* vmr_hip.h -- C ABI of libvmr_hip.so: the MI355X (gfx950) kernels behind the
 * SeqPAN cross-modal matching path.
 *
 * The reference (renjie-liang/VMRFrame) has NO FFI / operator registry: its
 * boundary is the Python naming convention of main.py:21,87,99.
 * Each entry point below therefore cites the reference *function* whose
 * arithmetic it replaces (file:line relative to the reference tree); the
 * Python host layer (vmrframe_amd/) binds them with ctypes and mirrors the
 * reference's module/engine interface.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *     stated; the caller owns all memory, the library never retains pointers.
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*) and
 *     returns without synchronising: 0 on success, negative on error
 *     (vmr_last_error() gives a thread-local message).
 *   - tensors are row-major, last dim contiguous. `dtype` is VMR_F32,
 *     VMR_BF16 or VMR_F16 for activations; statistics, biases, LayerNorm affine
 *     parameters, losses and weight gradients are always fp32.
 *   - VMR_F16 (IEEE half; BASELINE configs[4] "BAN ... fp16"): same kernels and
 *     layouts as VMR_BF16 (v_mfma_f32_16x16x32_f16 has the bf16 rate).  The
 *     reference's -1e30 masks (models/layers.py:9-12) never exist in the
 *     activation dtype: every masked softmax / pool applies them in fp32 inside
 *     the kernel, so nothing overflows half's +-65504.  Activation GRADIENTS in
 *     half need a loss scale: the caller multiplies the loss by S, weight
 *     gradients (always fp32) come out scaled by S and vmr_adamw divides by S
 *     (its `grad_scale`); a non-finite gradient norm makes vmr_adamw skip the
 *     update, which is what a dynamic scaler keys on (vmrframe_amd/optim.py).
 */
#ifndef VMR_HIP_H
#define VMR_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VMR_F32 0
#define VMR_BF16 1
#define VMR_F16 2

#define VMR_NEG_INF_MASK (-1e30f) /* models/layers.py:9 mask_value */

int vmr_version(void);
const char* vmr_last_error(void);
int vmr_sizeof_gemm_desc(void); /* ABI guard for foreign-language bindings of vmr_gemm_t */
/* Test utility (no reference counterpart): fills the whole LDS of every CU with `pattern`, so that a kernel reading
 * LDS it never wrote yields the pattern instead of its predecessor's leftovers.  scratch_u32: 4 device bytes. */
int vmr_debug_poison_lds(uint32_t pattern, void* scratch_u32, void* stream);
/* Test / A-B utility: tile-variant policy of vmr_gemm for the 256 x 256 "8-phase" kernel: 0 never, 1 where the rounds
 * model picks it (default, also VMR_GEMM_P8), 2 wherever the shape allows; -1 re-reads the environment. */
int vmr_debug_set_gemm_p8(int mode);
/* Test / A-B utility: operand-ring variant of the LDS-DMA GEMM kernels: 0 register-staged kernel only, 1 BK = 32 x 4 stages,
 * 2 BK = 64 x 2 stages (default, also VMR_GEMM_DMA), 3 BK = 32 x 3 stages with three workgroups per CU; -1 re-reads the
 * environment. */
int vmr_debug_set_gemm_dma(int mode);

/* ------------------------------------------------------------------ GEMM
 * C[z] = epilogue(alpha * opA(A[z]) . opB(B[z]))
 *   transA=0: A is [M,K] (k contiguous, leading dim lda); 1: A is [K,M]
 *   transB=0: B is [N,K] (k contiguous -- the layout of a Conv1D/Linear
 *             weight [out,in]);                           1: B is [K,N]
 * Replaces every pointwise Conv1D (models/layers.py:15-26 == x.W^T+b), the
 * nn.MultiheadAttention in/out projections (layers.py:570) and, batched, the
 * QK^T / P.V contractions of DualMultiAttention (layers.py:349-366) and
 * TopSelfAttention2 (layers.py:567-574), plus all their backward products.
 * bf16 runs on v_mfma_f32_16x16x32_bf16, f32 on v_mfma_f32_16x16x4_f32.
 */
#define VMR_EPI_BIAS 1      /* + bias[n] (fp32)                               */
#define VMR_EPI_RELU 2      /* max(.,0)                                        */
#define VMR_EPI_DROPOUT 4   /* inverted dropout, counter-based mask            */
#define VMR_EPI_RESIDUAL 8  /* + residual[m,n] (dtype, leading dim ldr)        */
#define VMR_EPI_AUX 16      /* also store the pre-residual value to aux        */
#define VMR_EPI_OUT_F32 32  /* C is fp32 regardless of dtype                   */
#define VMR_EPI_ACCUM 64    /* C (fp32) += result, via atomics (split-K safe)  */
#define VMR_EPI_ROWSCALE 128 /* multiply row m by rowscale[m] (fp32) at the end */
#define VMR_EPI_SLAB 256     /* split-K without atomics: split ks writes fp32 C + ks*M*ldc (then vmr_splitk_reduce) */
#define VMR_EPI_RES_PRE 512  /* with VMR_EPI_RESIDUAL: the residual joins the PRE-activation, act(x.W^T + b + residual),
                               * instead of being added after activation / dropout (BAN map2d_proj on a concatenation) */
#define VMR_EPI_AUX_BITS 1024 /* with VMR_EPI_AUX: aux is a BIT matrix, uint8 [M][N/8] (bit c%8 of byte [m][c/8] = the
                               * pre-residual value is non-zero) -- all the ReLU / dropout backward needs, 1/16 of the bytes.
                               * Only on the register-direct epilogue of the row-major-weight LDS-DMA kernels:
                               * ask vmr_gemm_aux_bits_supported() first; vmr_gemm refuses otherwise. */

typedef struct {
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  const void* residual;
  void* aux;
  const float* rowscale;
  int64_t lda, ldb, ldc, ldr;
  int32_t M, N, K;
  int32_t transA, transB;
  int32_t dtype;
  int32_t flags;
  float alpha;
  /* batching: z in [0, Z1*Z2); z1 = z / Z2, z2 = z % Z2; element strides */
  int32_t Z1, Z2;
  int64_t sA1, sA2, sB1, sB2, sC1, sC2;
  int32_t splitk; /* >1 requires VMR_EPI_ACCUM */
  float drop_p;
  uint32_t drop_seed;
  uint32_t drop_row0; /* added to the row index of the dropout counter (a GEMM split by rows) */
  const uint32_t* drop_step; /* nullable device counter mixed into the seed (hipGraph replay) */
  /* second bias term: with VMR_EPI_BIAS the epilogue adds bias_scale*bias[n] + bias2[n] (bias2 nullable;
   * bias_scale 0 is read as 1).  BiLinear (layers.py:257-263): dense_1(a)+dense_1(b)+bias_value
   * = dense_1.W.(a+b) + 2*dense_1.bias + bias_value, fed straight from the parameter arena. */
  const float* bias2;
  float bias_scale;
  /* residual row broadcast: with res_div > 1 row m adds residual[(m / res_div)*ldr + n], i.e. one
   * residual row per group of res_div consecutive rows (a per-clip term added to all T tokens of the
   * clip: the pooled-query half of CQConcatenate, layers.py:462-468, without materialising the cat). */
  int32_t res_div;
  /* bias gradient for free: with transA (A stored [K][M], the weight-gradient product dW = dz^T.x) the call
   * also ACCUMULATES a_colsum[m] += sum_k A[k][m] (fp32) -- inside the LDS-DMA kernel as one extra MFMA per
   * A fragment against a ones operand in the first column tile, otherwise by a column-sum launch.
   * Replaces the separate bias-gradient pass over dz (layers.py:15-26 backward).  Z1*Z2 must be 1. */
  float* a_colsum;
} vmr_gemm_t;

int vmr_gemm(const vmr_gemm_t* g, void* stream);
/* 1 if vmr_gemm would run g on a kernel whose epilogue can write VMR_EPI_AUX_BITS (g.aux / that flag need not be set
 * yet: the answer depends on shapes, layouts, alignment and the other flags only). */
int vmr_gemm_aux_bits_supported(const vmr_gemm_t* g);
/* Two INDEPENDENT products in one launch when both qualify for the single-round LDS-DMA tiles -- g1: row-major
 * operands without split-K (the input gradient dX = dY.W on the K-major weight copy), g2: both operands transposed,
 * split-K slabs (the weight gradient dW = dY^T.x) -- so that one problem's fill overlaps the other's store drain;
 * otherwise exactly vmr_gemm(g1) followed by vmr_gemm(g2). */
int vmr_gemm2(const vmr_gemm_t* g1, const vmr_gemm_t* g2, void* stream);
/* ... plus, optionally (slab != NULL), the vmr_splitk_reduce of an EARLIER product's slabs as extra workgroups of the
 * same grid (dst += sum_k slab[k], arguments as vmr_splitk_reduce): the previous layer's weight-gradient reduction
 * runs under this layer's products instead of being a launch of its own. */
int vmr_gemm2_reduce(const vmr_gemm_t* g1, const vmr_gemm_t* g2, const float* slab, float* dst, int nsplit, int64_t n,
                     int cols, int64_t ld_dst, void* stream);

/* ------------------------------------------------------------- LayerNorm
 * y = (x-mean)/sqrt(var+eps)*gamma+beta over the last dim D, optional
 * "+ pos[row % S]" (PositionalEmbedding add of FeatureEncoder,
 * layers.py:96-107,397) and optional dropout on the result.
 * nn.LayerNorm call sites: layers.py:85,116,136,271-273,619-620,650-651. */
int vmr_layernorm_fwd(const void* x, const float* gamma, const float* beta, float eps,
                      const void* pos, int S, void* y, float* mean, float* rstd,
                      int64_t rows, int D, int dtype, float drop_p, uint32_t drop_seed,
                      const uint32_t* drop_step, void* stream);
/* dx = LN backward of dy (dropout mask regenerated from the seed), optionally
 * dx += dres (gradient arriving through a residual branch); dgamma/dbeta are
 * ACCUMULATED (+=, two-stage reduction through `workspace`, fp32
 * VMR_LN_BWD_WS_FLOATS(rows, D) floats, caller-owned scratch) and must be zeroed by the
 * caller when needed.  dpos (optional, fp32 [S,D]) accumulates the
 * positional-table gradient. */
#define VMR_LN_BWD_MAX_BLOCKS 8192 /* 8 rows per block: up to 65,536 rows per call */
#define VMR_LN_BWD_WS_FLOATS(rows, D) ((((int64_t)(rows) + 7) / 8) * 2 * ((D) <= 512 ? 512 : ((D) <= 1024 ? 1024 : 2048)))
int vmr_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                      const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta,
                      float* dpos, float* workspace, int S, int64_t rows, int D, int dtype, float drop_p,
                      uint32_t drop_seed, const uint32_t* drop_step, void* stream);

/* ------------------------------------------ fused LayerNorm + depthwise conv
 * u[b,s,:] = sum_k w[:,k] * LN(x)[b,s+k-3,:]   (k=7, zero padded, UNMASKED)
 * = layer_norms[l] + depthwise Conv1d(groups=D) of DepthwiseSeparableConvBlock
 * (layers.py:139-148).  x,u: [B,S,D]; w: fp32 [D,7]; mean/rstd: fp32 [B*S]. */
int vmr_ln_dwconv_fwd(const void* x, const float* gamma, const float* beta, float eps,
                      const float* w, void* u, float* mean, float* rstd,
                      int B, int S, int D, int dtype, void* stream);
/* backward of the depthwise conv alone: dn = conv^T(du); dw += sum du*n where
 * n = LN(x) is recomputed from x, mean, rstd.  dn then goes to
 * vmr_layernorm_bwd. dw is accumulated (+=) through `workspace` (fp32
 * [B * VMR_DWCONV_BWD_BPS(S), D*7] caller-owned scratch: per-workgroup partials,
 * then one reduction). */
#define VMR_DWCONV_BWD_BPS(S) (((S) + 63) / 64)
int vmr_dwconv_bwd(const void* du, const void* x, const float* gamma, const float* beta,
                   const float* mean, const float* rstd, const float* w, void* dn, float* dw,
                   float* workspace, int B, int S, int D, int dtype, void* stream);
/* Two sequence groups in ONE launch: x holds [B1,S1,D] (the video clips) followed
 * by [B2,S2,D] (the query sentences) — the shared encoder (SeqPAN.py:77-80) runs
 * both through the same conv block.  A group with B == 0 is skipped.  Workspace
 * of the backward: fp32 [B1*VMR_DWCONV_BWD_BPS(S1) + B2*VMR_DWCONV_BWD_BPS(S2), D*7]. */
int vmr_ln_dwconv_fwd2(const void* x, const float* gamma, const float* beta, float eps,
                       const float* w, void* u, float* mean, float* rstd,
                       int B1, int S1, int B2, int S2, int D, int dtype, void* stream);
int vmr_dwconv_bwd2(const void* du, const void* x, const float* gamma, const float* beta,
                    const float* mean, const float* rstd, const float* w, void* dn, float* dw,
                    float* workspace, int B1, int S1, int B2, int S2, int D, int dtype, void* stream);

/* The row half of one conv-block layer's backward in ONE pass (csrc/convblock.hip; reference layers.py:126-148:
 * x' = drop(relu(pw(dw7(LN(x))) + b)) + x).  Given du = the gradient of u = dw7(LN(x)) and dres = the gradient of the
 * layer's output (= the residual branch's gradient) it produces what vmr_dwconv_bwd2 + vmr_layernorm_bwd + the LOWER
 * layer's vmr_relu_bwd_bias(mode 3) produce in three launches, with the same arithmetic:
 *   dx = LN-backward(conv^T(du)) + dres;   dz = dx * bit * bscale  (optional: `bits` = the lower layer's
 *   VMR_EPI_AUX_BITS mask [rows, D/8], bscale = its 1/(1-p));
 * and leaves ONE partial row per (persistent) workgroup for vmr_colreduce_batched: part_dw [nblocks, 7*D] (item
 * {part_dw, dw, dw, nblocks, 7*D, 0, 0}) and part_gb [nblocks, 2*D] ({part_gb, dgamma, dbeta, nblocks, D, D, 0}); the
 * lower layer's bias gradient is colsum(dz): give its weight-gradient product vmr_gemm_t.a_colsum.
 * *nblocks = vmr_convblock_bwd_blocks(...) <= VMR_CONVBLOCK_BWD_MAX_BLOCKS (one workgroup per CU).  D in {512, 1024}
 * (vmr_convblock_bwd_supported); other widths take the three separate kernels.  dz and bits are given together or
 * not at all. */
#define VMR_CONVBLOCK_BWD_MAX_BLOCKS 1024
int vmr_convblock_bwd_supported(int D, int dtype);
int vmr_convblock_bwd_blocks(int B1, int S1, int B2, int S2, int D);
int vmr_convblock_bwd(const void* du, const void* x, const void* dres, const unsigned char* bits, float bscale,
                      const float* gamma, const float* beta, const float* mean, const float* rstd,
                      const float* w, void* dx, void* dz, float* part_dw, float* part_gb,
                      int B1, int S1, int B2, int S2, int D, int dtype, int32_t* nblocks, void* stream);

/* Batched bf16 transpose: dst[c*rows + r] = src[r*cols + c] for every item, one launch per 64 items (the list
 * travels as kernel arguments).  The optimizer keeps a K-major copy of every weight matrix this way, so the input-
 * gradient products dX = dY.W (backward of Conv1D, models/layers.py:15-26) run as row-major-weight products. */
typedef struct {
  const void* src;
  void* dst;
  int32_t rows, cols;
} vmr_transpose_item_t;
#define VMR_TRANSPOSE_MAX_ITEMS 64
int vmr_transpose_batched(const vmr_transpose_item_t* items /* host array */, int n, void* stream);

/* Deferred parameter-gradient reductions.  The LayerNorm / depthwise-conv backward kernels leave per-workgroup
 * partial rows in `workspace`; the plain entry points reduce them at once (one small launch each, 38 per SeqPAN
 * step), the *_deferred forms only report how many partial rows they wrote (*nblocks) and the caller reduces ALL of
 * a backward pass's partials with ONE vmr_colreduce_batched launch before the optimizer reads the gradients
 * (workspaces must stay alive, and distinct, until then).  Item for vmr_layernorm_bwd_deferred: {workspace, dgamma,
 * dbeta, nblocks, D, D, slots = D <= 512 ? 512 : D <= 1024 ? 1024 : 2048}; for vmr_dwconv_bwd2_deferred:
 * {workspace, dw, dw, nblocks, 7*D, 0, 0}.  out += sum (accumulating, like the plain forms). */
typedef struct {
  const float* part;
  float* out0;
  float* out1;
  int32_t nblocks, n0, n1, slots;
} vmr_colreduce_item_t;
#define VMR_COLREDUCE_MAX_ITEMS 64   /* per launch (items travel as kernel arguments); longer lists are chunked */
int vmr_layernorm_bwd_deferred(const void* dy, const void* x, const float* gamma, const float* mean,
                               const float* rstd, const void* dres, void* dx, float* dpos, float* workspace,
                               int S, int64_t rows, int D, int dtype, float drop_p, uint32_t drop_seed,
                               const uint32_t* drop_step, int32_t* nblocks, void* stream);
int vmr_dwconv_bwd2_deferred(const void* du, const void* x, const float* gamma, const float* beta,
                             const float* mean, const float* rstd, const float* w, void* dn, float* workspace,
                             int B1, int S1, int B2, int S2, int D, int dtype, int32_t* nblocks, void* stream);
int vmr_colreduce_batched(const vmr_colreduce_item_t* items /* host array */, int n, void* stream);

/* ----------------------------------------------------------- masked softmax
 * P[z,r,:] = softmax_c( scale*S[z,r,c] + term ) with dropout on P.
 *   mode 0 (DualMultiAttention, layers.py:346-357): z=(b,h);
 *          term = (1 - rmask[b,r]*cmask[b,c]) * -1e30
 *   mode 1 (TopSelfAttention2 float key_padding_mask, layers.py:573): z=(t,h);
 *          term = cmask[c*cm_stride + t]   (ADDED, +1 for valid keys)
 * S: fp32 [Z,R,ldS]; P: dtype [Z,R,ldP] (columns >= C are zero filled up to ldP). */
int vmr_softmax_fwd(const float* S, void* P, void* Pkeep /*nullable: pre-dropout probs*/,
                    const float* rmask, const float* cmask,
                    int mode, int Z, int H, int R, int C, int ldS, int ldP, int cm_stride,
                    float scale, int dtype, float drop_p, uint32_t drop_seed,
                    const uint32_t* drop_step, void* stream);
/* dS = scale * Pk*(dP' - sum_c dP'*Pk) with dP' = dropout-mask(dP) regenerated
 * from the seed; P here is the PRE-dropout probability (Pkeep of the forward,
 * or its P when drop_p == 0). dP: fp32 [Z,R,ldS]; dS: dtype [Z,R,ldP] (pad columns zeroed). */
int vmr_softmax_bwd(const float* dP, const void* P, void* dS /*dtype, ld = ldP*/, int Z, int R, int C, int ldS,
                    int ldP, float scale, int dtype, float drop_p, uint32_t drop_seed,
                    const uint32_t* drop_step, void* stream);

/* ------------------------------------------------- fused attention (forward)
 * O = dropout(softmax(scale*Q.K^T + term)).V in ONE kernel (scores and context on MFMA, K/V of a
 * (z1,z2) slice staged in LDS, P never leaves the chip except as the copy the backward needs).
 * Replaces the three launches vmr_gemm + vmr_softmax_fwd + vmr_gemm for DualMultiAttention
 * (layers.py:346-367, mode 0) and TopSelfAttention2 (layers.py:567-574, mode 1); `term`, the
 * dropout stream and the zero-padded P/Pkeep layout are exactly vmr_softmax_fwd's, so
 * vmr_softmax_bwd and the backward GEMMs consume its outputs unchanged.
 * Q,K,V,O: bf16 4-D strided views [Z1,Z2,rows,hd] (unit stride along hd); strides[12] =
 * {s1,s2,row} for Q,K,V,O in elements (multiples of 8).  Supported: bf16, hd in {128,256},
 * 1 <= Lk <= 128 (vmr_attention_fwd_supported); anything else returns an error. */
int vmr_attention_fwd_supported(int hd, int Lk, int dtype);
int vmr_attention_fwd(const void* Q, const void* K, const void* V, void* O, void* P,
                      void* Pkeep /*nullable*/, const int64_t* strides, const float* rmask,
                      const float* cmask, int mode, int Z1, int Z2, int H, int Lq, int Lk, int hd,
                      int ldP, int cm_stride, float scale, int dtype, float drop_p,
                      uint32_t drop_seed, const uint32_t* drop_step, void* stream);

/* ------------------------------------------------- fused attention (backward)
 * dQ (+=), dK, dV of vmr_attention_fwd's O in ONE kernel per (z1,z2) slice: dP = dO.V^T, the softmax and
 * dropout backward (mask regenerated from the forward's counter stream, Pkeep = the forward's
 * pre-dropout probabilities), dQ = dS.K, dK = dS^T.Q, dV = P^T.dO -- replaces four batched vmr_gemm
 * launches + vmr_softmax_bwd and the fp32 dP round trip.  strides[21] = {s1,s2,row} for
 * Q,K,V,dO,dQ,dK,dV (elements, multiples of 8).  Supported: bf16, hd in {128,256}, Lq,Lk <= 128. */
int vmr_attention_bwd_supported(int hd, int Lq, int Lk, int dtype);
int vmr_attention_bwd(const void* dO, const void* Q, const void* K, const void* V, const void* Pkeep,
                      void* dQ, void* dK, void* dV, const int64_t* strides, int Z1, int Z2, int Lq,
                      int Lk, int hd, int ldP, float scale, int accumulate_dq, int dtype,
                      float drop_p, uint32_t drop_seed, const uint32_t* drop_step, void* stream);

/* ------------------------------------------------- CQAttention softmaxes
 * The two masked softmaxes of CQAttention (layers.py:419-421) over the trilinear
 * score S = S2 + rowterm[b,c] + colterm[b,q] (S2 from the batched MFMA GEMM, the
 * rank-1 terms optional): Srow = softmax_q(S + qmask term), Scol = softmax_c(S +
 * cmask term), both [B,Lc,ldP] in the activation dtype (pad columns zeroed).
 * bwd: dS2 [B,Lc,ldS] fp32, drow[b,c] = sum_q dS, dcol[b,q] = sum_c dS (nullable). */
int vmr_cq_softmax_fwd(const float* S2, const float* rowterm, const float* colterm,
                       const float* cmask, const float* qmask, void* Srow, void* Scol,
                       int B, int Lc, int Lq, int ldS, int ldP, int dtype, void* stream);
int vmr_cq_softmax_bwd(const void* dSrow, const void* dScol, const void* Srow, const void* Scol,
                       float* dS2, float* drow, float* dcol, int B, int Lc, int Lq, int ldS,
                       int ldP, int dtype, void* stream);

/* ------------------------------------------------------------------ losses
 * lossfun_loc (models/loss.py:43-54): mean_b( -sum_t y[b,t]*log_softmax(z[b,:])[t] ),
 * start + end in one launch. loss: fp32[1] (accumulated, caller zeroes). */
int vmr_soft_ce_fwd(const float* zs, const float* ze, const float* ys, const float* ye,
                    float* loss, float* lse /*[2,B]*/, int B, int T, void* stream);
int vmr_soft_ce_bwd(const float* zs, const float* ze, const float* ys, const float* ye,
                    const float* lse, const float* dloss, float* dzs, float* dze, int B, int T,
                    void* stream);
/* ------------------------------------------------------------- elementwise */
/* dst[r,0:cols] = dropout(cast(src[r,0:cols])), dst[r,cols:ld_dst] = 0.  Used for
 * the fp32 -> compute-dtype copy of the [B,T,V] video features (with the input
 * dropout of VisualProjection, layers.py:120) and of weights whose K is padded
 * to a multiple of 8 (V=500 -> 504) so the GEMM keeps 16-byte loads. */
int vmr_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t rows, int cols,
             int64_t ld_src, int64_t ld_dst, float drop_p, uint32_t drop_seed,
             const uint32_t* drop_step, void* stream);
/* Backward of the pointwise-conv epilogue (layers.py:133-146, 288-296):
 *  mode 0: db[c] += sum_rows dy[r,c]                       (bias gradient only)
 *  mode 1: dz = dy*scale*(h > 0), db += colsum(dz)         (ReLU [+dropout]; h = saved
 *          post-dropout ReLU output, scale = 1/(1-p))
 *  mode 2: dz = dy*scale*keep(seed, r*D+c), db += colsum(dz) (dropout without ReLU; the mask
 *          is regenerated from the seed the forward GEMM epilogue used)
 *  mode 3: as mode 1 with h the BIT matrix VMR_EPI_AUX_BITS wrote (uint8 [rows][D/8])
 *  mode 4: dz = dy*scale*keep(seed, r*D+c) + h   (stand-alone dropout whose input has a second consumer: h = that
 *          consumer's gradient, dtype [rows, ld]; the db sums are of the dropout term + h and normally unused)
 * db may be NULL in modes 1/2/3.  The column sums are ACCUMULATED: db += db_scale*colsum and, when
 * db2 is given, db2 += colsum (the two bias terms of vmr_gemm_t.bias/bias2; db_scale 0 reads as 1). */
int vmr_relu_bwd_bias(int mode, const void* dy, const void* h, void* dz, float* db, int64_t rows,
                      int D, int64_t ld, float scale, int dtype, float drop_p, uint32_t drop_seed,
                      const uint32_t* drop_step, float* db2, float db_scale, void* stream);
/* The positional add of FeatureEncoderPredict (layers.py:626-631): y[r,:] = x[r,:] + pos[r % S,:], pos = the fp32
 * table (rows >= S); bwd: dpos[s,:] += sum_b dy[b*S+s,:] (ACCUMULATED; dx = dy needs no kernel). */
int vmr_add_pos_fwd(const void* x, const float* pos, void* y, int64_t rows, int S, int D, int dtype, void* stream);
int vmr_add_pos_bwd(const void* dy, float* dpos, int64_t rows, int S, int D, int dtype, void* stream);
/* WordEmbedding.forward (layers.py:28-48) without the per-call table concat: out[i, 0:wd] = drop(row(ids[i])) in
 * `dtype`, row(0) = pad_vec, row(1) = unk_vec, row(k>=2) = glove_vec[k-2] (fp32 [1,wd], [1,wd], [nglove,wd]); out is
 * the [n, ldo] matrix that feeds query_conv1d; columns [zero_from, zero_to) are zero-filled (K padding), the columns
 * in between belong to vmr_char_cnn_fwd.  Dropout: counter stream index i*wd + c.
 * bwd: dunk[c] += sum_{i: ids[i]==1} drop'(dout[i,c]) (the only trainable row; ACCUMULATED). */
int vmr_word_embedding_fwd(const int64_t* ids, const float* pad_vec, const float* unk_vec, const float* glove_vec,
                           void* out, int64_t n, int wd, int64_t nglove, int64_t ldo, int zero_from, int zero_to,
                           int dtype, float drop_p, uint32_t drop_seed, const uint32_t* drop_step, void* stream);
int vmr_word_embedding_bwd(const int64_t* ids, const void* dout, float* dunk, int64_t n, int wd, int64_t ldo, int dtype,
                           float drop_p, uint32_t drop_seed, const uint32_t* drop_step, void* stream);
/* out[i,:] = table[idx[i],:] (fp32): WordEmbedding / CharacterEmbedding lookups
 * (layers.py:42-48,66).  bwd: dtable[idx[i],:] += dout[i,:] for idx != padding_idx
 * (dtable accumulated with float atomics; the caller zeroes it). */
int vmr_embedding_fwd(const int64_t* idx, const float* table, float* out, int64_t n, int D,
                      int64_t nrows, void* stream);
int vmr_embedding_bwd(const int64_t* idx, const float* dout, float* dtable, int64_t n, int D,
                      int64_t nrows, int64_t padding_idx, void* stream);
/* dropout mask materialisation (tests): m[i] = keep(seed,i) ? 1/(1-p) : 0 */
int vmr_dropout_mask(float* m, int64_t n, float drop_p, uint32_t seed, void* stream);

/* fused elementwise programs of the dual-attention gating and the CQAttention concat
 * (layers.py:370-380, 424); rows x D, activation dtype.  op:
 *  0 GATE_FWD   o0 = a*d + c*b                       (a=s_score b=s_value c=x_score d=x_value)
 *  1 GATE_BWD   a=do, b..e = s_score,s_value,x_score,x_value -> o0..o3 = their gradients
 *  2 SIGGATE_FWD a = [scores|values] [rows,2D] -> o0 = sigmoid(scores + (1-rowmask)*-1e30)*values
 *  3 SIGGATE_BWD a = do, b = [scores|values] -> o0 = d[scores|values] [rows,2D]
 *  4 CAT4_FWD   a=C b=c2q c=q2c -> o0 = [C, c2q, C*c2q, C*q2c] [rows,4D]
 *  5 CAT4_BWD   a=dcat [rows,4D], b=C c=c2q d=q2c -> o0=dC o1=dc2q o2=dq2c */
int vmr_eltwise(int op, const void* a, const void* b, const void* c, const void* d, const void* e,
                const float* rowmask, void* o0, void* o1, void* o2, void* o3, int64_t rows, int D,
                int dtype, void* stream);
/* dst += sum_k slab[k] (fp32): second stage of a split-K GEMM whose splits were written as plain
 * slabs (VMR_EPI_SLAB) instead of float atomics.  slab: [nsplit][n] contiguous; dst: n elements,
 * contiguous when cols == 0, else rows of `cols` elements with leading dimension ld_dst (the
 * gradient of a column slice of a weight matrix inside the arena).  ld_dst < cols: the slab rows carry zero-padded K
 * columns (a [N, 500] weight whose product ran at K = 512); dst is dense [rows, ld_dst] and only the first ld_dst
 * columns of every slab row are reduced. */
int vmr_splitk_reduce(const float* slab, float* dst, int nsplit, int64_t n, int cols, int64_t ld_dst,
                      void* stream);
/* out[m,n] = dtype(sum_k slab[k][m,n] + bias[n] + addend[m,n]) (bias fp32 [cols], addend dtype [rows,cols], both
 * nullable): second stage of a few-tile split-K product together with its epilogue and the cast. */
int vmr_splitk_reduce_cast(const float* slab, int nsplit, int64_t rows, int cols, const float* bias, const void* addend,
                           void* out, int dtype, void* stream);

/* ------------------------------------------------- CQAttention score kernel
 * The trilinear similarity of CQAttention (models/layers.py:417-421,427-437; rank-1 terms folded onto the short
 * stream by the caller) and BOTH masked softmaxes in one launch, one workgroup per clip:
 *   M[v,t] = long[b,v,:] . short_op[b,t,:] + shortterm[b,t]          (long: [B,Ll<=128,D], short_op: [B,Ls<=32,D])
 *   P_t = softmax_t(M + (1-mask_short[b,t])*-1e30),  P_v = softmax_v(M + (1-mask_long[b,v])*-1e30)
 * orient 0 (context = long stream):  Srow = P_t, Scol = P_v as [B,Ll,ldP] (columns t, zero padded to ldP);
 * orient 1 (context = short stream): Srow = P_v, Scol = P_t as [B,Ls,ldP] (columns v, zero padded to ldP)
 * -- exactly the (S_row, S_col) layout of vmr_cq_softmax_fwd, whose backward consumes them.  The short operand is
 * staged in LDS by DMA, the long operand is streamed from HBM once as MFMA fragments, the score tile stays in
 * registers.  bf16, D %% 256 == 0 (vmr_cq_score_supported).
 * Second output form (either pair may be null): Pt_lm / Pv_lm, fp32 "long-major" [B, Ll, SP], SP = Ls rounded up to 8
 * (P_t = softmax over t, P_v = softmax over v; columns t >= Ls are 0) -- what the fused apply kernels read row by
 * row with scalar loads, in both orientations. */
int vmr_cq_score_supported(int Ll, int Ls, int D, int dtype);
/* 1 if vmr_cq_score_fwd_ws takes the shape in its row-split form (128 < Ll <= 256 included): fp32 pair + workspace only */
int vmr_cq_score_split_supported(int Ll, int Ls, int D, int dtype);
int vmr_cq_score_fwd(const void* lng, const void* short_op, const float* shortterm, const float* mask_long,
                     const float* mask_short, void* Srow, void* Scol, float* Pt_lm, float* Pv_lm, int B, int Ll,
                     int Ls, int D, int ldP, int orient, int dtype, void* stream);
/* Same, with a workspace of vmr_cq_score_ws_floats(B) floats: when only the fp32 pair is requested, a clip's long
 * rows are split over 4 workgroups (256 instead of 64 at cfg2: one workgroup per clip pulls its operands through ONE
 * CU at the per-CU HBM fetch rate) and a second small launch normalises the softmax over the long index from the
 * per-workgroup (max, sum) pairs. */
int vmr_cq_score_ws_floats(int B);
int vmr_debug_set_cq_split(int mode); /* test / A-B utility: 1 = row-split form where possible, 0 = single launch (default), -1 = re-read VMR_CQ_SPLIT */
int vmr_cq_score_fwd_ws(const void* lng, const void* short_op, const float* shortterm, const float* mask_long,
                        const float* mask_short, void* Srow, void* Scol, float* Pt_lm, float* Pv_lm, float* colstats, int B,
                        int Ll, int Ls, int D, int ldP, int orient, int dtype, void* stream);

/* ------------------------------------------------- CQAttention apply stage (fused) and the block's backward
 * reference models/layers.py:422-424: c2q = S_.Q, q2c = (S_.S_t^T).C [re-associated as S_.(S_t^T.C)],
 * out = cat[C, c2q, C*c2q, C*q2c] -> [B*Lc, 4D], the input of cqa_linear.  ctx [B,Lc,D], qry [B,Lq,D] bf16.
 * S_lm / St_lm: S_ (softmax over q) and S_t (softmax over c) as fp32 LONG-MAJOR arrays [B, Ll, SP] (Ll = the longer
 * of Lc / Lq, SP = the shorter rounded up to 8): with the context long they are indexed [c][q], with the context
 * short [q][c] -- i.e. (Pt_lm, Pv_lm) of vmr_cq_score_fwd for orient 0 and (Pv_lm, Pt_lm) for orient 1.
 * One workgroup per (clip, 128-channel slice); c2q / mid / q2c never exist in HBM.  bf16, D %% 128 == 0, one stream
 * <= 32 rows and the other <= 256 (vmr_cq_apply_supported).
 * vmr_cq_apply_bwd: dcat4 [B*Lc,4D] -> dctx, dqry (the apply stage's share) and `parts`, fp32
 *   [B][D/128][2][LcP][LqP] (LcP, LqP = Lc, Lq rounded up to 16): per-slice partials of dS_ and dS_t, indexed (c, q).
 * vmr_cq_softmax_bwd_parts: sums the slices and applies both softmax backwards: dS_lm fp32 long-major [B, Ll, SP]
 *   and dterm[b, s] = sum_l dS (the gradient of the rank-1 term on the short stream).
 * vmr_cq_score_bwd: S2[l,s] = lng[l,:].sht[s,:] => dlng = dS.sht, dsht = dS^T.lng (the two score operands). */
/* (one partial tile pair per channel slice: 128-channel slices, 64-channel ones when the longer stream has more than 128 rows) */
#define VMR_CQ_APPLY_PARTS_FLOATS(B, Lc, Lq, D) \
  ((int64_t)(B) * ((D) / (((Lc) > 128 || (Lq) > 128) ? 64 : 128)) * 2 * (((Lc) + 15) / 16 * 16) * (((Lq) + 15) / 16 * 16))
int vmr_cq_apply_supported(int Lc, int Lq, int D, int dtype);
int vmr_cq_apply_fwd(const void* ctx, const void* qry, const float* S_lm, const float* St_lm, void* out, int B, int Lc,
                     int Lq, int D, int dtype, void* stream);
int vmr_cq_apply_bwd(const void* dcat4, const void* ctx, const void* qry, const float* S_lm, const float* St_lm,
                     void* dctx, void* dqry, float* parts, int B, int Lc, int Lq, int D, int dtype, void* stream);
int vmr_cq_softmax_bwd_parts(float* parts /* scratch: summed in place */, const float* S_lm, const float* St_lm, float* dS_lm, float* dterm,
                             int B, int Lc, int Lq, int D, void* stream);
int vmr_cq_score_bwd(const void* lng, const void* sht, const float* dS_lm, void* dlng, void* dsht, int B, int Ll, int Ls,
                     int D, int dtype, void* stream);

/* ------------------------------------------------------------ WeightedPool
 * reference models/layers.py:440-453: alpha = softmax_l(x[b,l,:].w + (1-mask[b,l])*-1e30),
 * pooled[b,:] = sum_l alpha[b,l]*x[b,l,:].  x: [B,L,D] dtype; w, mask, alpha fp32; pooled [B,D] dtype.
 * bwd: dx [B,L,D]; dw[D] is ACCUMULATED (float atomics). */
int vmr_weighted_pool_fwd(const void* x, const float* w, const float* mask, float* alpha, void* pooled,
                          int B, int L, int D, int dtype, void* stream);
int vmr_weighted_pool_bwd(const void* dpooled, const void* x, const float* w, const float* alpha,
                          void* dx, float* dw, int B, int L, int D, int dtype, void* stream);

/* ------------------------------------------------- inference head + IoU metrics
 * vmr_infer_basic (reference utils/engine.py:28-44): per clip, the masked softmax of the start and
 * end logits, and the first indices (i*, j*) of max_{i<=j} sp[i]*ep[j] by rows and by columns;
 * frac[b] = (i*, j*) / sum(vmask[b]) (fp32 [B,2]), idx[b] = (i*, j*) (int32 [B,2]).  The [B,T,T] outer
 * product of the reference is never materialised; results are bit-identical to it.
 * vmr_iou_metrics (utils/utils.py:161-185, models/loss.py:83-109): ious[i] = IoU(props[i], gts[i])
 * (nullable output) and acc[5] (fp64, ACCUMULATED) += {#iou>=0.3, #iou>=0.5, #iou>=0.7, n, sum iou}. */
int vmr_infer_basic(const float* slogits, const float* elogits, const float* vmask, float* frac, int* idx,
                    int B, int T, void* stream);
int vmr_iou_metrics(const float* props, const float* gts, float* ious, double* acc, int n, void* stream);

/* ------------------------------------------------------------ narrow heads
 * Conv1D with N <= 8 output channels (match head N = 4: models/SeqPAN.py:41,78; start / end heads N = 1:
 * layers.py:659-671) as bandwidth-bound matrix-vector kernels instead of a >= 94 %-padded MFMA tile.
 * fwd: y[m,0:N] (fp32, dense [M,N]) = x[m,:] . W[n,:]^T + bias;  x: [M,K] dtype with row stride ldx; W: fp32 [N,K].
 * bwd: dx[m,:] = sum_n dy[m,n] W[n,:] (dtype, dense [M,K], nullable) and dW (fp32 [N,K]) / db ([N], nullable)
 *      are ACCUMULATED (two-stage through `workspace`, VMR_NARROW_WS_FLOATS(M,N,K) fp32, caller-owned
 *      scratch).  K %% 8 == 0, K <= 2048. */
#define VMR_NARROW_WS_FLOATS(M, N, K) \
  ((((int64_t)(M) + 31) / 32) * ((K) >= 2048 ? 1 : 256 / ((K) / 8)) * ((int64_t)(N) * (K) + (N)))
int vmr_narrow_linear_fwd(const void* x, const float* W, const float* bias, float* y, int64_t M, int N,
                          int K, int64_t ldx, int dtype, void* stream);
int vmr_narrow_linear_bwd(const float* dy, const void* x, const float* W, void* dx, float* dW, float* db,
                          float* workspace, int64_t M, int N, int K, int64_t ldx, int dtype, void* stream);
/* Same with dx = dx_add + sum_n dy[m,n] W[n,:] (dx_add: dtype, dense [M,K]: the gradient of x's other consumer). */
int vmr_narrow_linear_bwd_add(const float* dy, const void* x, const float* W, void* dx, const void* dx_add, float* dW,
                              float* db, float* workspace, int64_t M, int N, int K, int64_t ldx, int dtype, void* stream);
/* The label-embedding fuse of the match head (models/SeqPAN.py:80-82): y = (res + p . E^T) * rowscale with p fp32 [M,N]
 * (the Gumbel-softmax probabilities, N = 4), E = label_embs fp32 [K,N], res / y [M,K] dtype, rowscale fp32 [M] or NULL.
 * bwd: dres = dy * rowscale (dtype [M,K]), dp[m,n] = dres[m,:] . E[:,n] (fp32 [M,N]), dE[k,n] += sum_m p[m,n] dres[m,k]
 * (ACCUMULATED, fp32 [K,N]); workspace: VMR_NARROW_WS_FLOATS(M, N, K) floats. */
int vmr_label_fuse_fwd(const float* p, const float* E, const void* res, const float* rowscale, void* y, int64_t M, int N,
                       int K, int dtype, void* stream);
int vmr_label_fuse_bwd(const void* dy, const float* p, const float* E, const float* rowscale, void* dres, float* dp,
                       float* dE, float* workspace, int64_t M, int N, int K, int dtype, void* stream);

/* ------------------------------------------------- match head + its loss
 * vmr_gumbel_softmax_fwd/bwd: F.gumbel_softmax(logits, tau) of models/SeqPAN.py:79 over C <= 8 classes:
 *   probs = softmax((logits + g)/tau), g = -log(Exp(1)) noise -- given (`noise`, fp32 [R,C]) or drawn in the
 *   kernel from the counter hash (seed, optional device step counter: hipGraph replay draws fresh noise).
 *   `padded` (nullable, dtype [R,ldo]) receives the same probabilities zero-padded to ldo columns (the
 *   operand of the label-embedding product, models/SeqPAN.py:80-82).  bwd: dlogits from dprobs and/or dpadded.
 * vmr_match_loss_fwd/bwd: lossfun_match (models/loss.py:24-41): sum_r -probs[r,label[r]]*vmask[r] /
 *   (sum_r vmask[r] + 1e-12) + || offdiag(E^T E) ||_F, E = label_embs fp32 [D,C].  aux: fp32
 *   [C*C + 2 + VMR_MATCH_LOSS_SCRATCH] (Gram matrix, norm, denominator, then per-workgroup partial sums --
 *   no atomics, nothing for the caller to zero) kept for the backward; dE is ACCUMULATED. */
#define VMR_MATCH_LOSS_SCRATCH 512
int vmr_gumbel_softmax_fwd(const float* logits, const float* noise /*nullable*/, float tau, uint32_t seed,
                           const uint32_t* step, float* probs, void* padded, int64_t R, int C, int ldo,
                           int dtype, void* stream);
int vmr_gumbel_softmax_bwd(const float* dprobs /*nullable*/, const void* dpadded /*nullable*/,
                           const float* probs, float tau, float* dlogits, int64_t R, int C, int ldo,
                           int dtype, void* stream);
int vmr_match_loss_fwd(const float* probs, const int64_t* labels, const float* vmask, const float* E,
                       float* loss, float* aux, int64_t R, int D, int C, void* stream);
int vmr_match_loss_bwd(const float* dloss, const int64_t* labels, const float* vmask, const float* E,
                       const float* aux, float* dprobs, float* dE, int64_t R, int D, int C, void* stream);

/* y = x*a + b over the last dim (a, b fp32 [D]): the rank-1-folded operand of the CQAttention trilinear score
 * (models/layers.py:427-437).  bwd: dx = dy*a; da += colsum(dy*x); db += colsum(dy) (ACCUMULATED). */
int vmr_scale_shift_fwd(const void* x, const float* a, const float* b, void* y, int64_t rows, int D, int dtype,
                        void* stream);
int vmr_scale_shift_bwd(const void* dy, const void* x, const float* a, void* dx, float* da, float* db,
                        int64_t rows, int D, int dtype, void* stream);

/* ------------------------------------------------------------ CharacterEmbedding
 * reference models/layers.py:51-75 in one kernel each way: out[word, coff_k + o] = max_p relu(b_k[o] +
 * sum_{c,j} W_k[o,c,0,j] * drop(table[ids[word,p+j], c])) for the four kernel widths k = 1..4 (oc[k-1] out
 * channels each, concatenated along the columns: OT = sum oc).  ids: int64 [W, C] (4 <= C <= 16), table fp32
 * [num_chars, CD] (row 0 = padding: no gradient), w[k-1]: fp32 [oc, CD, 1, k], b[k-1]: fp32 [oc]; out: dtype
 * [W, ldo]; amax: int8 [W, OT] (arg-max positions, kept for the backward).  Dropout acts on the gathered
 * character rows (counter stream index (word*C + p)*CD + c).
 * bwd: dw[k-1] / db[k-1] (fp32, shapes of w / b) and dtable (nullable) are ACCUMULATED; workspace:
 * vmr_char_cnn_ws_floats(W, CD, oc, dtype) fp32 of caller-owned scratch. */
int vmr_char_cnn_ws_floats(int W, int CD, const int* oc, int dtype);
int vmr_char_cnn_fwd(const int64_t* char_ids, const float* table, const float* const* w, const float* const* b,
                     const int* oc, void* out, int64_t ldo, int8_t* amax, int W, int C, int CD, int dtype,
                     float drop_p, uint32_t drop_seed, const uint32_t* drop_step, void* stream);
int vmr_char_cnn_bwd(const void* dout, const void* out, int64_t ldo, const int8_t* amax, const int64_t* char_ids,
                     const float* table, const float* const* w, const float* const* b, const int* oc,
                     float* const* dw, float* const* db, float* dtable, float* workspace, int W, int C, int CD,
                     int dtype, float drop_p, uint32_t drop_seed, const uint32_t* drop_step, void* stream);

/* ------------------------------------------------------------ input staging (next row N3)
 * out[b, i, :] = mean(arena[row_off[b] + seg[b,i] : row_off[b] + seg[b,i+1], :]) when seg[b,i] < seg[b,i+1],
 * else arena[row_off[b] + seg[b,i], :], for i < out_len[b]; rows i >= out_len[b] are zero; mask[b,i] = i < out_len[b].
 * = interpolate_avrage / sample_vfeat_linear + pad_video_seq + convert_length_to_mask of the reference
 * (utils/data_utils.py:70-84,161-201, utils/utils.py:125-130) for a whole batch in one launch, reading a
 * device-resident feature arena (fp32 [frames, V]).  seg: int32 [B, T+1]; out: fp32 or bf16 [B, T, ldo]. */
int vmr_resample_pad(const float* arena, const int64_t* row_off, const int* seg, const int* out_len, void* out,
                     float* mask /*nullable*/, int B, int T, int V, int64_t ldo, int out_dtype, void* stream);

/* --------------------------------------------------------------- optimizer
 * fused AdamW over a flat fp32 parameter arena (utils/utils.py:87-97:
 * AdamW, weight_decay 0.01 except names containing bias/layer_norm) with the
 * clip_grad_norm_ scale (main.py:95) folded in; also refreshes the 16-bit
 * compute copy of the weights (p16: bf16 or f16 per p16_dtype, nullable).
 * decay: per-element 0/1 mask as uint8.
 * loss_scale (nullable, device float[1] = S): g and gnorm_sq were produced from
 * S * loss; the kernel divides both by S, and a non-finite norm SKIPS the whole
 * update.  vmr_loss_scale_update then keeps the dynamic scaler's state
 * {S, clean streak} on the device: overflow -> S /= 2 (>= min_scale), the step
 * counter holds; else step_dev += 1 and S *= 2 every growth_interval clean steps. */
int vmr_sumsq(const float* g, float* out /*[1], accumulated*/, int64_t n, void* stream);
int vmr_loss_scale_update(float* state /*[2]*/, const float* gnorm_sq, int* step_dev /*nullable*/, int growth_interval,
                          float min_scale, void* stream);
int vmr_adamw(float* p, const float* g, float* m, float* v, const uint8_t* decay, void* p16, int p16_dtype,
              const float* gnorm_sq, float max_norm, float lr, float beta1, float beta2, float eps,
              float wd, int step,
              const int* step_dev /*nullable: 0-based step count on the device, overrides step*/,
              float warmup_steps, float total_steps /* with step_dev and total_steps > 0: lr is scaled
              by transformers' linear warm-up/decay multiplier of step_dev[0] (utils/utils.py:95-96),
              evaluated on the device so a captured hipGraph replays the schedule */,
              const float* loss_scale /*nullable*/, int64_t n, void* stream);

/* ------------------------------------------------ BAN 2-D proposal map (N2)
 * Replaces the diagonal loops of SparseMaxPool / DenseMaxPool (models/BANlib/model.py:226-290) and
 * SparseBoundaryCat (:293-325) feeding map2d_proj (models/BAN.py:87-93).  Only the cells the reference's mask2d
 * keeps exist, in compact cell-major order: the main diagonal (i, i), then diagonal k = 1..ndiag at offset
 * o_k = grow[0] + .. + grow[k-1] (cells (i, i + o_k), i ascending) -- the order of the reference's `maskij`.
 * grow[k] = MaxPool1d kernel size - 1 of pooler k (1 | 2 | 4 for pooling_counts levels 0 | 1 | 2; all 1 for
 * DenseMaxPool); `grow` is the device copy, `grow_host` the host copy of the same int32 array.
 *   M[b, c, :] = max_{t in [i_c, j_c]} x[b, t, :]          x: [B, N, F] (fuse_feature), M: [B, C, F]
 *   R[b, c, :] = ps[b, i_c, :] + pe[b, j_c, :]             ps / pe: [B, N, F] views with row stride ldp (the start /
 *                                                           end thirds of map2d_proj applied per frame); R, ps, pe
 *                                                           may all be NULL
 * C = vmr_map2d_cells(grow_host, ndiag, N).  N <= 160, F % 64 == 0.
 * Backward: dx[b, t, :] = sum of dM over the cells whose arg-max frame is t (first frame wins ties, as the chained
 * MaxPool1d backward does); dps[b, i, :] = sum_{c: i_c = i} dR, dpe[b, j, :] = sum_{c: j_c = j} dR. */
int vmr_map2d_cells(const int32_t* grow_host, int ndiag, int N);
int vmr_map2d_pool_fwd(const void* x, const void* ps, const void* pe, int64_t ldp, const int32_t* grow,
                       const int32_t* grow_host, int ndiag, void* M, void* R, int B, int N, int F, int dtype,
                       void* stream);
int vmr_map2d_pool_bwd(const void* x, const void* dM, const void* dR, const int32_t* grow,
                       const int32_t* grow_host, int ndiag, void* dx, void* dps, void* dpe, int64_t ldp, int B,
                       int N, int F, int dtype, void* stream);
/* dense [B, N, N, W] <- compact [B, C, W]: cell_of[i*N + j] = compact index or -1; cells off the mask get fill[W]
 * (fp32, nullable = 0): what the reference computes there from an all-zero input (tmap / map2d_proj of BAN.forward). */
int vmr_map2d_scatter(const void* cells, const int32_t* cell_of, const float* fill, void* out, int B, int N, int W,
                      int64_t C, int dtype, void* stream);

/* ------------------------------------------------ BAN encoders: bidirectional LSTM, pointwise half (N2)
 * Replaces the recurrence inside nn.LSTM(batch_first, bidirectional) as QueryEncoder / VisualEncoder use it
 * (models/BANlib/model.py:27-45,60-72: packed by length, zero initial state, outputs zero past each length).  The matrix
 * halves are vmr_gemm launches of the caller; one call = step s of BOTH directions (z = 0 forward in time, 1 backward).
 * Step order: direction 0 handles time t = s, direction 1 time t = len[b] - 1 - s, both only while s < len[b]; the caller
 * builds direction 1's input projection on the per-sample reversed sequence (vmr_lstm_reverse_rows).
 *   gx  [2][B][T][4H] dtype  x-part of the gate pre-activations incl. both biases, by step (gate order i, f, g, o)
 *   gh  [2][B][4H]    f32    h_{s-1} . W_hh^T of this step          c  [2][B][H] f32 cell state (in/out, start at 0)
 *   hs  [2][B][H]     dtype  h for the next step's product (in/out, start at 0)
 *   act [2][B][T][4H] dtype  post-activation gates by step, cs [2][B][T][H] f32 cell state after step s,
 *   hp  [2][B][T][H]  dtype  h before step s                        (all three: saved for the backward)
 *   y   [B][T][2H]    dtype  output by TIME; rows past len[b] are not written (zero-fill once before step 0)
 * Backward, step s (descending): dy [B][T][2H]; dh [2][B][H] f32 = dg_{s+1} . W_hh (zeros at s = T-1); dc [2][B][H] f32
 * carried cell gradient (in/out, start at 0); dg [2][B][T][4H] dtype = gradient of the gate pre-activations of step s
 * (= gradient of gx; dW_hh = sum_s dg_s^T hp_s).  dtype VMR_F32 or VMR_BF16.
 * ndir = 2 x the number of INDEPENDENT LSTMs advanced by the call (same B, T, H and lengths, their own weights and inputs:
 * TemporalDifference's two, models/BANlib/model.py:168-171): every leading [2] above becomes [ndir] (direction z belongs to LSTM
 * z >> 1 and runs backward in time when z is odd) and y / dy become [ndir/2][B][T][2H]. */
int vmr_lstm_cell_fwd(const void* gx, const void* gh, const int32_t* len, void* c, void* hs, void* act, void* cs, void* hp,
                      void* y, int B, int T, int H, int s, int ndir, int dtype, void* stream);
int vmr_lstm_cell_bwd(const void* dy, const void* act, const void* cs, const int32_t* len, const void* dh, void* dc, void* dg,
                      int B, int T, int H, int s, int ndir, int dtype, void* stream);
/* Fused step (bf16, H = 256 or 512: vmr_lstm_step_supported): the recurrent product and the pointwise half in ONE launch.
 * Forward: hprev / hnext are two distinct [2][B][H] buffers (h ping-pongs; hprev is not read at s = 0), whh [2][4H][H];
 * everything else as vmr_lstm_cell_fwd.  Backward: whht [2][H][4H] = W_hh transposed; reads dg[:, :, s+1, :] (written by
 * the call for step s+1; nothing at s = T-1), writes dg[:, :, s, :]; dc as vmr_lstm_cell_bwd. */
int vmr_lstm_step_supported(int H, int dtype);
int vmr_lstm_step_fwd(const void* gx, const void* hprev, const void* whh, const int32_t* len, void* c, void* hnext, void* act,
                      void* cs, void* hp, void* y, int B, int T, int H, int s, int ndir, int dtype, void* stream);
int vmr_lstm_step_bwd(const void* dy, const void* act, const void* cs, const int32_t* len, const void* whht, void* dc, void* dg,
                      int B, int T, int H, int s, int ndir, int dtype, void* stream);
/* The whole recurrence of a bi-LSTM layer in ONE launch (bf16 / fp16, H = 256 or 512, B <= 64, ndir <= 8): persistent
 * workgroups (direction z = blockIdx % 8, 16 hidden units each) advance all T steps with one counter barrier per step
 * instead of one kernel per step; results are those of T calls of vmr_lstm_step_fwd / _bwd (same layouts, same MFMA
 * products; the gate math may contract into different FMAs: agreement to an ulp of the storage type).
 * hist: caller-owned scratch of *bytes from vmr_lstm_seq_hist_bytes(T, H, ndir, &bytes) (the h exchange: every step writes fresh
 * addresses); sync: int32[16] device words the caller ZEROES before each call ([0..7] arrival counters, [8] is raised
 * if a workgroup gave up waiting -- bounded polls: a lost workgroup cannot hang the GPU; the results are then invalid).
 * The exchange protocol is placement-independent (write-through stores, agent-scope counter, L1-bypassing loads); the
 * GPU must otherwise be idle enough for all 8 * H / 16 workgroups to be resident at once.  whht (backward) = W_hh
 * transposed, [ndir][H][4H]. */
int vmr_lstm_seq_supported(int B, int H, int ndir, int dtype);
int vmr_lstm_seq_hist_bytes(int T, int H, int ndir, int64_t* bytes);
/* 1: vmr_lstm_seq_fwd exchanges h through `hist` itself (no arrival counter): the caller fills hist with the 16-bit
 * pattern 0x7FFF ("not yet written": a NaN no arithmetic produces) before EVERY launch; 0: counters, hist need not be
 * initialised.  Under hipGraph capture fill with a kernel, not hipMemsetAsync (memset nodes have been seen to run out of
 * order with their neighbours on replay). */
int vmr_lstm_seq_sentinel(void);
/* the same for vmr_lstm_seq_bwd and its exchange buffer dg (every row of dg is written by the launch) */
int vmr_lstm_seq_bwd_sentinel(void);
int vmr_lstm_seq_fwd(const void* gx, const void* whh, const int* len, void* act, void* cs, void* hp, void* y, void* hist,
                     int* sync, int B, int T, int H, int ndir, int dtype, void* stream);
int vmr_lstm_seq_bwd(const void* dy, const void* act, const void* cs, const int* len, const void* whht, void* dg,
                     int* sync, int B, int T, int H, int ndir, int dtype, void* stream);
/* dst[b][s][:] = s < len[b] ? src[b][len[b]-1-s][:] : 0 for [B][T][D] rows (its own inverse on the valid part).  D a
 * multiple of 16 bytes, 16-byte aligned pointers. */
int vmr_lstm_reverse_rows(const void* src, const int32_t* len, void* dst, int B, int T, int D, int dtype, void* stream);

/* ------------------------------------------------ BAN proposal sampling (N2), HOST routine
 * Replaces Aaptive_Proposal_Sampling / proposal_selection_with_negative (models/BANlib/model.py:371-435): per clip, the
 * greedy pick-and-suppress loop over the kept cells of the score map in descending score order.  Sequential and tiny, so it
 * runs on the host (8 threads over the clips) between the map stage and the proposal head.  All pointers are HOST memory:
 * scores [B][C] = sigmoid(tmap) at the kept cells in mask.nonzero() (row-major) order, cells [C][2] = (i, j) of those cells;
 * out [B][n_out][2] int64 receives (start, end + 1) in the reference's order [negatives | padding | selected by rank].
 * Equal scores keep cell order.  Fails (-22) when a clip yields a count != n_out. */
int vmr_ban_sample_host(const float* scores, const int32_t* cells, int B, int C, float thresh, int topk, int neighbor,
                        int negative, int n_out, int64_t* out);
/* The same sampling ON THE DEVICE (csrc/sampler.hip), one workgroup per clip: bitonic sort of the scores in LDS, the greedy
 * pick-and-suppress with block-wide min / prefix sums, the output assembly.  Same order, tie rule (equal scores keep cell
 * order, NaN first) and float arithmetic as vmr_ban_sample_host; lets the BAN train step be ONE captured graph (no
 * device-to-host hop between the map stage and the proposal head).  scores, cells, out as above but DEVICE pointers;
 * status: device int32 [B] = proposals produced per clip (== n_out when the reference's view would succeed).  C <= 8192. */
int vmr_ban_sample(const float* scores, const int32_t* cells, int B, int C, float thresh, int topk, int neighbor,
                   int negative, int n_out, int64_t* out, int32_t* status, void* stream);

/* sim[b, c] = <q_b, y_bc> / (max(|y_bc|, 1e-30) * (1 + 1e-8)) with q unit-normalised by the caller: the cosine similarity of
 * BAN's ContrastLoss (reference models/BANlib/model.py:639-671) between the sentence projection and every compact map cell's
 * projection, one pass over y [B, C, D] forward and one backward (dy written, dq [B, D] fp32 ACCUMULATED).  rnorm [B, C] =
 * 1 / max(|y|, 1e-30) is kept for the backward.  D in {8, 16, 32, 64, 128}. */
int vmr_cos_rows_supported(int D);
int vmr_cos_rows_fwd(const float* q, const void* y, float* sim, float* rnorm, int B, int C, int D, int dtype, void* stream);
int vmr_cos_rows_bwd(const float* q, const void* y, const float* sim, const float* rnorm, const float* dsim, void* dy,
                     float* dq, int B, int C, int D, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif
