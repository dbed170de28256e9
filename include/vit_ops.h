/*
 * vit_ops.h -- C ABI of libvit_hip.so: gfx950 kernels for the ViT side of the
 * Styl3R hot path.  Plain C, device pointers,
 * hipStream_t as void*, 0 / negative return codes, no torch types.
 *
 * Reference interface each entry point replaces:
 *   vit_rope2d        <- curope.rope_2d(tokens, positions, base, F0)
 *                        src/model/encoder/backbone/croco/curope/curope.cpp:49-65,
 *                        kernels.cu:17-108 (in place, forward F0 / backward -F0),
 *                        called from cuRoPE2D_func curope2d.py:12-29
 *   vit_attention_fwd <- xformers.ops.memory_efficient_attention(q, k, v, scale, p=0)
 *                        call sites blocks.py:129,195 (fp32, (B,N,H,64), no mask)
 *   vit_attention_bwd <- its autograd backward
 *   vit_linear_fwd    <- nn.Linear (+ nn.GELU / residual add) of Mlp / Attention / Block,
 *                        blocks.py:76-82,100,131,149-152
 */
#ifndef VIT_OPS_H
#define VIT_OPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIT_OK 0
#define VIT_EINVAL (-1)
#define VIT_ELAUNCH (-3)

/*
 * In-place 2-D RoPE on tokens viewed as (B, N, H, D) with element strides
 * (stride_b, stride_n, stride_h), innermost dimension contiguous, D % 4 == 0.
 * The D features of a token are four quarters [u_Y, v_Y, u_X, v_X] (Q = D/4); pair d < Q is rotated by
 * theta = pos * sign / base^(d/Q), the Y quarters by pos[b,n,0], the X quarters by pos[b,n,1]:
 *     u' = u cos(theta) - v sin(theta),   v' = v cos(theta) + u sin(theta).
 * cos_tab / sin_tab are (P, Q) tables for positions 0..P-1 at sign = +1 (built once by the host with
 * the reference's own formula, pos_embed.py:121-128); `sign` = +1 forward, -1 backward (curope's F0 / -F0).
 * positions: int64 (B, N, 2), every value in [0, P).
 */
int vit_rope2d(float *tokens, const int64_t *positions, const float *cos_tab, const float *sin_tab, int B, int N,
               int H, int D, int P, int64_t stride_b, int64_t stride_n, int64_t stride_h, float sign, void *stream);

/*
 * softmax(q k^T * scale) v for head_dim 64, fp32, no mask, no dropout.
 * q (B, Nq, H, 64), k/v (B, Nk, H, 64), out (B, Nq, H, 64), given by element strides of the (b, n, h)
 * dimensions (innermost contiguous), so q/k/v may be views into a fused qkv buffer.
 * lse (B, H, Nq) receives log-sum-exp (natural log) of the scaled scores, needed by the backward.
 * If cos_tab != NULL the 2-D RoPE of vit_rope2d is applied to q (positions qpos) and k (positions kpos)
 * on the fly while the tiles are loaded (the buffers are NOT modified).
 */
typedef struct VitAttnArgs {
    int32_t B, H, Nq, Nk;
    float scale;
    int64_t q_sb, q_sn, q_sh;
    int64_t k_sb, k_sn, k_sh;
    int64_t v_sb, v_sn, v_sh;
    int64_t o_sb, o_sn, o_sh;
    const int64_t *qpos, *kpos;       /* (B,Nq,2) / (B,Nk,2) or NULL */
    const float *cos_tab, *sin_tab;   /* (P,16) or NULL */
    int32_t P;
    /* vit_attention_bwd only: token stride (in floats) of the gradient outputs dq / (dk, dv); 0 = H * 64, i.e. contiguous
     * (B,N,H,64) tensors.  3 * H * 64 with dq, dk, dv pointing at planes 0 / 1 / 2 of ONE (B,N,3,H,64) buffer writes the gradient
     * of a packed qkv projection in place (no select_backward fills / copies / adds behind the kernel). */
    int64_t dq_sn, dkv_sn;
    /* optional |max| words (see vit_amax below: 8 KiB, zeroed by the caller) that the kernels fill with the largest |value| they STORE -- the
     * f16x3 scale of the consumer (the proj / qkv / projq / projk / projv Linear layers behind and in front of the attention) then costs no
     * vit_amax pass of its own.  Forward: amax_out (of `out`).  Backward: amax_dq, amax_dk, amax_dv (may all point at ONE word: the packed
     * qkv gradient).  NULL = not wanted.  Launches that fall back to the exact-f32 kernels fill them with a pass over the contiguous result. */
    uint32_t *amax_out, *amax_dq, *amax_dk, *amax_dv;
    /* arithmetic mode 3 ("f16x3": two fp16 pieces per operand, three products per contraction step) only: the |max| words of the INPUT tensors q, k,
     * v (may be one word: a packed qkv projection) and, for the backward, of dout -- the power-of-two operand scales come from them.  Required in
     * that mode (VIT_EINVAL otherwise: no guessed scale); ignored in every other mode. */
    const uint32_t *amax_q, *amax_k, *amax_v, *amax_g;
} VitAttnArgs;

int vit_attention_fwd(const VitAttnArgs *a, const float *q, const float *k, const float *v, float *out, float *lse,
                      void *stream);
/*
 * Arithmetic of the contractions of vit_attention_fwd and vit_attention_bwd: 1 (default) = bf16x6 split arithmetic on the bf16
 * MFMA (csrc/vit_attention_x6.hip, vit_attention_bwd_x6.hip; fp32 round-off accuracy -- measured at or below the f32 kernels'
 * error against float64 -- forward 1.3 - 1.6x, backward 1.3 - 1.55x faster), 0 = exact-f32 MFMA, 2 = the split-arithmetic kernels with three
 * instead of six partial products per contraction step ("bf16x3": operands good to 2^-18, the attention counterpart of vit_x6_set_products(3)),
 * 3 = "f16x3" (round 6): two fp16 pieces per operand x power-of-two scale, three products on v_mfma_f32_32x32x16_f16 -- 2^-22 per product, the accuracy
 * class of mode 1 at the MFMA count of mode 2; needs VitAttnArgs.amax_q / _k / _v (/ _g); the dS operand of the backward carries a per-lane running
 * scale (csrc/vit_attention_bwd_x6.hip).  Strides that are not multiples
 * of 4 floats (or bases that are not 16-byte aligned) always take the f32 kernels.  Per calling thread (thread_local), read at launch time on that thread.
 */
int vit_attention_set_arith(int mode);
int vit_attention_arith(void);

/*
 * Backward of vit_attention_fwd.  q/k/v as in the forward (same strides / RoPE arguments); out, dout,
 * dq (B,Nq,H,64) and dk, dv (B,Nk,H,64) contiguous; lse from the forward; delta_ws: float workspace of
 * B*H*Nq elements.  With fused RoPE the returned dq / dk are gradients w.r.t. the UNROTATED q / k.
 */
int vit_attention_bwd(const VitAttnArgs *a, const float *q, const float *k, const float *v, const float *out,
                      const float *lse, const float *dout, float *dq, float *dk, float *dv, float *delta_ws,
                      void *stream);

/*
 * Fused fp32 Linear: out (M,N) = [residual (M,N) +] act( x (M,K) . w^T + bias ), w (N,K) row-major as
 * nn.Linear stores it (blocks.py: qkv / proj / fc1 / fc2 / projq / projk / projv).  act: 0 identity,
 * 1 exact (erf) GELU = nn.GELU() default (blocks.py:64).  bias, residual, pre may be NULL; `pre`
 * receives the pre-activation (x w^T + bias) for the backward.  K % 16 == 0; all tensors contiguous.
 */
int vit_linear_fwd(const float *x, const float *w, const float *bias, const float *residual, float *out, float *pre,
                   int M, int N, int K, int act, void *stream);

/*
 * The same Linear at fp32 accuracy on the bf16 matrix cores ("bf16x6": every fp32 operand is split exactly into three
 * bf16 pieces, the six leading partial products are accumulated in fp32; csrc/vit_gemm_x6.hip).  The weight is split
 * once per optimizer step with vit_split_weight:
 *   transpose = 0: w (rows = N, cols = K)  ->  packed weight for  out = x . w^T       (the forward, nn.Linear)
 *   transpose = 1: w (rows = N, cols = K)  ->  packed w^T for     dX  = dY . w        (the input gradient):
 *                  call vit_linear_x6_fwd(dY, packed, NULL, NULL, dX, NULL, M, K, N, 0)
 * `packed` holds vit_split_weight_bytes(rows, cols) = 6 bytes per element (+ 8 KiB: the |max| word right behind the pieces carries the
 * weight's |max| bit pattern in the f16x3 mode), layout [out_row][k/8][piece][8] bf16.
 * The contraction length must be a multiple of 16.
 */
size_t vit_split_weight_bytes(int rows, int cols);
/*
 * Partial products per launch of every bf16x6 kernel below (Linear, weight gradient, convolutions): 6 = fp32 round-off accuracy
 * (default); 3 = "bf16x3", a0 b0 + a0 b1 + a1 b0 only: ~3.5e-6 of the output scale per GEMM (two orders tighter than the TF32 the
 * reference enables, croco.py:13) for half the MFMA work.  Per calling thread (thread_local), read when that thread launches a kernel.  Returns VIT_EINVAL for
 * any other n.
 *
 * n = 2 selects "f16x3": every operand is split into TWO fp16 pieces (11-bit significands) of value * s, s = the power of two that puts
 * the operand TENSOR's absolute maximum into [2^14, 2^15), and the three products h h' + h l' + l h' run on v_mfma_f32_32x32x16_f16:
 * 2^-22 per product (bf16x3: 2^-16; bf16x6: 2^-24) at the MFMA count and data path of bf16x3.  The kernels then need the |max| of their
 * ACTIVATION operands: compute it with vit_amax (an exact integer max over the fp32 bit patterns of |x|.  A "word" is 64 uint32 slots, one per 128-byte cache
 * line -- 8 KiB, 4-byte aligned: producers fold their maxima into slot (workgroup + wave) & 63, readers take the max over the slots, so no
 * single cache line sees thousands of serialised atomics; the 8 KiB must hold zeros before the launch; several launches may accumulate into one word) and announce the device addresses with vit_x6_set_operand_amax right before the
 * launch they belong to, on the launching thread: a = x for vit_linear_x6_fwd / vit_conv_x6_fwd (forward and input-gradient uses alike),
 * a = dY, b = x for vit_linear_x6_wgrad / vit_linear_x6_wgrad_acc / vit_conv_x6_wgrad.  The pair is consumed by that launch; a launch in
 * this mode without it returns VIT_EINVAL (no guessed scale, no fallback).  Weights get their scale inside vit_split_weight (called in
 * this mode).  Scales are powers of two, so applying and removing them is exact; fp16 range is the only reason they exist.
 */
int vit_x6_set_products(int n);
int vit_x6_products(void);
int vit_x6_set_operand_amax(const void *a_word, const void *b_word);
/* `word` (zeroed): the next vit_linear_x6_fwd / vit_linear_x6r_fwd (cfg 1, 3) / vit_conv_x6_fwd / vit_layernorm_fwd / vit_layernorm_bwd launch on
 * this thread writes the |max| of its OUTPUT there -- the epilogue sees every value it stores -- so the consumer of that output needs no vit_amax
 * pass of its own.  Any arithmetic mode.  Consumed by that launch; launches whose workgroups only see partial sums (split-K) and the kernels
 * without the epilogue hook (k_conv_x6) run a vit_amax pass over their result instead; the lockstep ring kernel (cfg 2) returns VIT_EINVAL. */
int vit_x6_set_output_amax(void *word);
int vit_amax(const float *x, int64_t n, void *out_word, void *stream);
int vit_split_weight(const float *w, void *packed, int rows, int cols, int transpose, void *stream);
/*
 * The same Linear with an LDS-DMA operand ring (csrc/vit_gemm_x6r.hip).  Its weight operand is the BLOCK layout
 * packed[row / 64][k / 8][piece][row % 64][8] bf16 written by vit_split_weight_block (rows padded to a multiple of 64 with
 * zeros; vit_split_weight_block_bytes gives the size).  cfg: 1 = 128 x 128 ring, 2 = 256 x 256 ring, 3 = 256 x 256 with one
 * conversion per workgroup and ping-pong wave pairs, K % 16 == 0.  Experimental: measured against vit_linear_x6_fwd in
 * DESIGN.md 9.2; the Python layer dispatches the six-product Linear to cfg 3 / cfg 1 on the shapes where they measured faster
 * (styl3r_amd/vit_ops.py `_RING_SHAPES`).
 */
size_t vit_split_weight_block_bytes(int rows, int cols, int transpose);
int vit_split_weight_block(const float *w, void *packed, int rows, int cols, int transpose, void *stream);
/*
 * The forward image AND the transposed (input-gradient) image of one Linear weight w (rows, cols) in ONE launch -- what a trainable layer
 * needs again after every optimizer step; the fp32 weight is read once.  block_fwd / block_t: 1 = the block layout of vit_split_weight_block
 * (size vit_split_weight_block_bytes(rows, cols, 0 / 1)), 0 = the row layout of vit_split_weight (vit_split_weight_bytes).  rows % 8 == 0,
 * cols % 8 == 0.  f16x3: announce the weight's |max| word first (vit_x6_set_operand_amax(word, NULL)).  Bytes identical to the single calls.
 */
int vit_split_weight_pair(const float *w, void *packed_fwd, void *packed_t, int rows, int cols, int block_fwd, int block_t, void *stream);
/*
 * The operand images of a convolution weight w (Co, Ci, k, k), k in {1, 3}, straight from the parameter in ONE launch: packed_fwd = the image
 * vit_conv_x6_fwd takes for the forward (vit_split_weight_bytes(Co, k*k*Ci)); packed_dx (nullable) = the image of the spatially flipped,
 * channel-transposed weight the same entry takes for the input gradient (vit_split_weight_bytes(Ci, k*k*Co)).  Ci % 8 == 0 (and Co % 8 == 0
 * with packed_dx).  Bytes identical to vit_split_weight on the rearranged copies.  f16x3: announce the weight's |max| word first.
 */
int vit_split_conv_weight_pair(const float *w, void *packed_fwd, void *packed_dx, int Co, int Ci, int ksize, void *stream);
/*
 * Every weight image of a model in ONE launch (what an optimizer step invalidates): jobs = device array sorted by first_block, one per image.
 *   kind bit 0: pack w^T (as transpose = 1 above); bit 1: the BLOCK layout of vit_split_weight_block, else the layout of vit_split_weight.
 *   first_block / nbx: the job owns workgroups [first_block, first_block + nbx * ceil(R / 64)), nbx = ceil(Kc / 64), with R x Kc the
 *   output rows x contraction length of the image (R = transpose ? cols : rows).  amax / tail: f16x3 only -- the weight's |max| word and
 *   where the image keeps its copy (right behind the pieces, as the single-weight calls place it).
 * The arithmetic mode is the calling thread's (vit_x6_set_products).  Results are byte-identical to the single-weight calls.
 */
typedef struct VitSplitJob {
    const float *w;
    void *packed;
    const uint32_t *amax;
    uint32_t *tail;
    int32_t rows, cols;
    int32_t kind;
    uint32_t first_block;
    uint32_t nbx;
    uint32_t reserved;
} VitSplitJob;
int vit_split_weights_many(const VitSplitJob *jobs_device, int n_jobs, uint32_t total_blocks, void *stream);
int vit_linear_x6r_fwd(const float *x, const void *w_packed, const float *bias, const float *residual, float *out, float *pre,
                       int M, int N, int K, int act, int cfg, void *stream);

/* Small row counts (round 6; batch-1 serving: 257 / 514 token rows -- infer_model_re10k.py:262-560): vit_linear_x6r_fwd with cfg = 5 runs
 * csrc/vit_gemm_sm.hip on the BLOCK image: 32- / 64-row x 64-column tiles, the contraction split over the 4 / 8 waves of a workgroup, every wave
 * streaming its own operands without a workgroup barrier (weight pieces straight from the block image into MFMA registers, activations
 * row-contiguous -> split -> a wave-private LDS double buffer), the partial tiles reduced once through LDS in wave order (deterministic) and the
 * FULL epilogue (bias, GELU / GELU', residual, `pre`, the |max| word) in the same launch: no zero-fill launch, no atomics, no vit_amax pass
 * behind it.  Same operands, same split functions as the other kernels.  vit_linear_sm_ok(M, N, K): 1 when the shape is served (M <= max_rows,
 * N % 64 == 0, K a multiple of 64 x waves), else the caller keeps vit_linear_x6_fwd; cfg 5 on an unserved shape is VIT_EINVAL.
 * vit_linear_sm_set: max_rows (default 1024; 0 = never, the A/B switch), tile_row_blocks (0 | 1 | 2) and waves (0 | 4 | 8) force the tile rows
 * (x 32) and the waves per workgroup, 0 = the built-in rule.  Per host thread. */
int vit_linear_sm_set(int max_rows, int tile_row_blocks, int waves);
int vit_linear_sm_ok(int M, int N, int K);

/* Serving path of the dual decoders (backbone_croco_multiview.py:147-188 at two context views: decoder 1 on view 0, decoder 2 on view 1, the
 * same layer shapes with two weight sets): TWO problems of one shape per launch.  The arrays hold `groups` (1 or 2) HOST-side arrays of device
 * pointers -- they are read at the call and travel as kernel arguments.  vit_linear_sm_grouped: out_g = [residual_g +] act(x_g . w_g^T + bias_g)
 * on the block images w_block[g] (the shape must satisfy vit_linear_sm_ok); bias / residual may be NULL (or hold NULL entries).  The activation
 * |max| word announced with vit_x6_set_operand_amax covers BOTH inputs, the word of vit_x6_set_output_amax both outputs (the stacked tensors are
 * one tensor with one scale).  vit_layernorm_fwd_grouped: y_g = LayerNorm(x_g; gamma_g, beta_g) over M rows of C each, the operations of
 * vit_layernorm_fwd (results within 1 ulp of it; no mean / rstd: forward only). */
int vit_linear_sm_grouped(const float *const *x, const void *const *w_block, const float *const *bias, const float *const *residual,
                          float *const *out, int groups, int M, int N, int K, int act, void *stream);
int vit_layernorm_fwd_grouped(const float *const *x, const float *const *gamma, const float *const *beta, float *const *y, int groups,
                              int M, int C, float eps, void *stream);
/*
 * The ping-pong kernel (cfg 3 above: 256 x 256 tiles, one workgroup per CU) with an S-way split of K whose partial tiles meet in a
 * caller-owned workspace (vit_linear_x6c_workspace_bytes; plain stores, one ticket per tile, the last arriver reduces and runs the
 * full epilogue, activation and `pre` included).  vit_linear_x6c_choose_splits returns the S that fills the chip's 256 CUs best for
 * a shape, or 0 when 256 x 256 tiles do not fit it (keep vit_linear_x6_fwd then).  Same weight layout as vit_linear_x6r_fwd.
 */
size_t vit_linear_x6c_workspace_bytes(int M, int N, int splits);
int vit_linear_x6c_choose_splits(int M, int N, int K);
int vit_linear_x6c_fwd(const float *x, const void *w_packed, const float *bias, const float *residual, float *out, float *pre,
                       int M, int N, int K, int act, int splits, void *workspace, size_t workspace_bytes, void *stream);
/* hipGraph capture: every FORWARD entry point of this header only enqueues kernels on `stream` (the zero fill in front of a split-contraction
 * launch is a kernel, not hipMemsetAsync -- that call's graph node does not replay faithfully on this runtime), so a serving forward can be
 * captured and replayed (styl3r_amd/graphs.py).  The backward entry points still use hipMemsetAsync for their accumulators: eager only. */
/* act: 0 = none, 1 = exact GELU (optionally storing the pre-activation in `pre`), 2 = out = (x . W^T) * gelu'(residual): the
 * input-gradient GEMM of the layer behind a GELU with the GELU's backward in its epilogue (`residual` = the saved pre-activation of the
 * GELU, no bias, `pre` must be NULL) -- the separate GeluBackward pass over the (M, 4 dim) hidden gradient disappears. */
int vit_linear_x6_fwd(const float *x, const void *w_packed, const float *bias, const float *residual, float *out, float *pre,
                      int M, int N, int K, int act, void *stream);


/*
 * Weight and bias gradient of the same Linear in bf16x6 arithmetic:
 *   dw (N,K) = dy^T (M,N)^T . x (M,K),     dbias (N) = column sums of dy   (dbias may be NULL)
 * dy, x row-major fp32 (autograd's grad_output and the saved input); both outputs are overwritten (the call zeroes
 * them itself when it splits M across workgroups and accumulates with fp32 atomics).
 */
int vit_linear_x6_wgrad(const float *dy, const float *x, float *dw, float *dbias, int M, int N, int K, void *stream);
/* same, ADDING to the values dw / dbias already hold (no zeroing): lets the host point them at the zeroed slices of a
 * gradient all-reduce bucket, so the gradient is produced in place (no memset per layer, no pack copy) and a weight used
 * twice in one backward accumulates correctly */
int vit_linear_x6_wgrad_acc(const float *dy, const float *x, float *dw, float *dbias, int M, int N, int K, void *stream);

/*
 * Convolution of the DPT heads (dpt_block.py:79-218,350-419; 3x3 pad 1 stride 1, or 1x1) as a bf16x6 implicit GEMM on NCHW
 * fp32 tensors:  out (B,Co,H,W) = [residual +] bias + conv(f(in (B,Ci,H,W)), w),  f = ReLU if relu_in else identity
 * (ResidualConvUnit applies its activation BEFORE each convolution, dpt_block.py:  out = conv2(act(conv1(act(x)))) + x).
 * w_packed = vit_split_weight of the weight rearranged to (Co, ksize*ksize*Ci) with k = tap * Ci + ci, tap = ky * ksize + kx.
 * The same entry computes the input gradient from the spatially flipped, channel-transposed weight.  Ci % 16 == 0.
 * `relu_in` is a flag word: bit 0 = ReLU on the input; bit 1 (VIT_CONV_GATE) = `residual` is not added but used as a sign
 * gate, out = residual > 0 ? conv : 0 -- the input gradient of a ReLU-fused convolution with residual := its forward
 * input (bias must be NULL).
 */
#define VIT_CONV_RELU_IN 1
#define VIT_CONV_GATE 2
int vit_conv_x6_fwd(const float *in, const void *w_packed, const float *bias, const float *residual, float *out, int B, int Ci,
                    int Co, int H, int W, int ksize, int relu_in, void *stream);

/*
 * Weight / bias gradient of that convolution: dw (Co,Ci,k,k) and dbias (Co, may be NULL) from dy (B,Co,H,W) and the
 * forward input `in` (B,Ci,H,W) [through ReLU if relu_in]; both outputs are overwritten.  W % 8 == 0, (H*W) % 16 == 0.
 */
int vit_conv_x6_wgrad(const float *dy, const float *in, float *dw, float *dbias, int B, int Ci, int Co, int H, int W, int ksize,
                      int relu_in, void *stream);

/*
 * LayerNorm over the last dimension of a row-major (M, C) fp32 tensor (croco/blocks.py:144-152,205-222 norm1..3 / norm_y,
 * the trunks' enc_norm / dec_norm; nn.LayerNorm(C, eps=1e-6) semantics: biased variance, y = (x - mean) * rstd * gamma + beta).
 * C % 256 == 0, C <= 2048.  The forward also returns the per-row mean / rstd the backward needs.
 * Backward: dx = LayerNorm input gradient [+ dskip]; `dskip` (nullable) is the gradient that arrived at the residual
 * branch of a pre-norm block (x + f(LN(x))): adding it here replaces the framework's separate (M, C) add.  dgamma / dbeta
 * (dbeta nullable) are overwritten, or added to when accumulate != 0.  scratch: vit_layernorm_scratch_bytes(M, C) bytes.
 */
size_t vit_layernorm_scratch_bytes(int M, int C);
int vit_layernorm_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *rstd, int M, int C,
                      float eps, void *stream);
int vit_layernorm_bwd(const float *dy, const float *x, const float *mean, const float *rstd, const float *gamma,
                      const float *dskip, float *dx, float *dgamma, float *dbeta, void *scratch, int M, int C, int accumulate,
                      void *stream);

/*
 * F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True) of the DPT heads (dpt_block.py Interpolate /
 * FeatureFusionBlock): in (planes, H, W) -> out (planes, 2H, 2W), planes = B*C of a contiguous NCHW tensor; W even.
 */
/* (out must be 16-byte aligned; a 16-byte aligned `in` / `dout` with power-of-two maps takes the LDS-staged fast kernels, anything else the generic ones) */
int vit_upsample2x_fwd(const float *in, float *out, int64_t planes, int H, int W, void *stream);
/* its input gradient: dout (planes, 2H, 2W) -> din (planes, H, W), a gather (no atomics), overwritten */
int vit_upsample2x_bwd(const float *dout, float *din, int64_t planes, int H, int W, void *stream);

/*
 * ReLU followed by Dropout(p) of the 'gs_params' DPT heads (dpt_block.py:332-340) in one pass: y = keep ? max(x, 0) / (1 - p) : 0
 * (y may alias x), keep drawn from Philox-4x32-10 keyed by (seed, element index / 4) -- no mask tensor; the backward reads y only:
 * dx = y > 0 ? g / (1 - p) : 0 (dx may alias g).  n = number of floats, a multiple of 4; 0 <= p < 1.
 */
int vit_relu_dropout_fwd(const float *x, float *y, int64_t n, float p, uint64_t seed, void *stream);
int vit_relu_dropout_bwd(const float *y, const float *g, float *dx, int64_t n, float p, void *stream);

/*
 * The tail of every DPT head -- ReLU [-> Dropout(p)] -> Conv2d(C, CO, kernel_size = 1) with CO = 3 or 8 output channels
 * (heads/dpt_block.py:319-320 'regression', :337-339 'gs_params') -- as ONE pass over the (B, C, H, W) activation each way
 * (csrc/vit_head_tail.hip; HBM-bound: 4 B C HW bytes read forward, 8 B C HW moved backward):
 *   y (B, CO, HW) = bias + W (CO, C) . a(h),   a(h) = keep max(h, 0) / (1 - p), keep from the generator of vit_relu_dropout_fwd
 *   dh = a'(h) W^T dy,   dw (CO, C) = sum dy a(h)^T,   db (CO) = sum dy       (dw / db are zeroed by the call, then accumulated with atomics)
 * h is the raw (pre-ReLU) output of the convolution in front; nothing else is stored for the backward.  C % 8 == 0, C <= 256, HW % 4 == 0,
 * p = 0 selects plain ReLU (the seed is ignored).  Any other CO: VIT_EINVAL (callers keep the separate ReLU / Dropout / convolution kernels).
 */
int vit_head_tail_fwd(const float *h, const float *w, const float *bias, float *y, int B, int C, int CO, int64_t HW, float p, uint64_t seed,
                      void *stream);
int vit_head_tail_bwd(const float *h, const float *w, const float *dy, float *dh, float *dw, float *db, int B, int C, int CO, int64_t HW,
                      float p, uint64_t seed, void *stream);

/*
 * Pieces of the 'gs' head's input merger, `feat_up(path_1) + ReLU(Conv2d(3, 256, 7, 1, 3)(imgs))` (dpt_gs_head.py:113-118,146-148):
 * vit_im2col7 writes the 7x7 / padding-3 patches of a (B, 3, H, W) image as (B, 160, H, W) planes (147 taps in the weight's
 * (ci, ky, kx) order + 13 zero planes), over which the convolution is a 1x1 convolution on vit_conv_x6_fwd / vit_conv_x6_wgrad;
 * vit_upsample2x_add_relu_fwd = vit_upsample2x_fwd + max(addend, 0) in the same pass (addend: the convolution's pre-activation,
 * shape of the output).  W % 4 == 0.
 */
int vit_im2col7(const float *img, float *cols, int B, int H, int W, void *stream);
/*
 * 3x3 / stride 1 / padding 1 patches of an NCHW tensor as pixel-major rows: cols (B H W, 9 Ci), column (tap, ci) with tap = 3 dy + dx,
 * zeros outside the image, ReLU applied on the way when relu != 0.  The weight gradient of a 3x3 convolution over few pixels is then
 * vit_linear_x6_wgrad(dY as (B H W, Co) rows, cols): the small stages of the DPT heads (dpt_block.py:79-218) need no library kernel.
 */
int vit_im2col3_rows(const float *in, float *cols, int B, int Ci, int H, int W, int relu, void *stream);
int vit_upsample2x_add_relu_fwd(const float *in, const float *addend, float *out, int64_t planes, int H, int W, void *stream);

/*
 * Head tails + Gaussian adapter in one pass: reg_dense_depth(mode='exp') (postprocess.py:22-60),
 * sigmoid + map_pdf_to_opacity (encoder_noposplat_multi_token_style.py:115-128,205-209), UnifiedGaussianAdapter.forward
 * (gaussian_adapter.py:122-153) and build_covariance / quaternion_to_matrix (gaussians.py:8-44), from the DPT heads' NCHW
 * outputs straight to the Gaussians in the rasterizer's layout (Gaussian g = view * H*W + pixel of scene bi):
 *   means (b, v*H*W, 3), cov (b, v*H*W, 3, 3), sh (b, v*H*W, 3, d_sh), opac (b, v*H*W); scales (.., 3) / rot (.., 4,
 *   normalised xyzw) are optional (visualization_dump) and may be NULL.
 * Inputs: pts0 (b,3,H,W) = downstream_head1's raw output for view 0, ptsr (b*(v-1),3,H,W) = downstream_head2's for views
 * 1..v-1 (NULL when v == 1); par0 / parr likewise with par_channels >= 8 channels [density, scale x3, rotation xyzw x4, ...];
 * app (b*v, 3*d_sh, H, W) = the appearance head (channel c*d_sh + k), or NULL: the SH channels are then channels 8.. of
 * par0 / parr (the non-style encoders' single gs head, par_channels == 8 + 3*d_sh).  sh_mask: (d_sh) device floats.
 * opacity_exponent = 2^x of map_pdf_to_opacity (1 at the reference's settings).
 * The backward overwrites every input gradient (same shapes as the inputs; d_par* only channels it owns: all of them).
 */
typedef struct VitAdapterArgs {
    int32_t b, v, H, W, d_sh, par_channels;
    float opacity_exponent;
    const float *pts0, *ptsr, *par0, *parr, *app, *sh_mask;
} VitAdapterArgs;
int vit_adapter_fwd(const VitAdapterArgs *a, float *means, float *cov, float *sh, float *opac, float *scales, float *rot,
                    void *stream);
int vit_adapter_bwd(const VitAdapterArgs *a, const float *d_means, const float *d_cov, const float *d_sh, const float *d_opac,
                    float *d_pts0, float *d_ptsr, float *d_par0, float *d_parr, float *d_app, void *stream);

const char *vit_version(void);
const char *vit_last_error(void);

/* ---- optimizer pass (vit_optim.hip): AdamW of model_wrapper_style.py:885-895 over every parameter of a group in ONE launch ----
 * chunks: device array; chunk i = n contiguous fp32 elements of one parameter (p), its gradient (g), first / second moment (m, v) and the
 * parameter's device-resident step counter (float, already incremented for this step, torch.optim.AdamW's state["step"] of a fused optimizer);
 * vec != 0 promises that p, g, m, v are 16-byte aligned.  grad_scale (device, may be NULL): every gradient is divided by it while it is read
 * (the clip coefficient deferred by ddp.BucketedGradReducer.clip_grad_norm_(defer_to=...)); the stored gradient is left as it is.
 * Arithmetic and update order of the framework's fused AdamW (amsgrad = maximize = false). */
typedef struct VitAdamChunk {
    float *p;
    const float *g;
    float *m;
    float *v;
    const float *step;
    int32_t n;
    int32_t vec;
    uint32_t *amax;     /* NULL, or the (zeroed) |max| word of the parameter this chunk belongs to: the kernel folds the |max| of the UPDATED values into
                           it, so the f16x3 mode's per-weight scale (vit_split_weight) costs no vit_amax pass after an optimizer step */
} VitAdamChunk;
int vit_adamw_step(const VitAdamChunk *chunks, int n_chunks, float lr, float beta1, float beta2, float eps, float weight_decay,
                   const float *grad_scale, void *stream);

#ifdef __cplusplus
}
#endif
#endif
