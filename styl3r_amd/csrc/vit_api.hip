// vit_api.hip -- the C ABI declared in include/vit_ops.h.
#include <hip/hip_runtime.h>

#include "../../include/vit_ops.h"

namespace vit {
thread_local hipError_t g_last_hip_error = hipSuccess;
int rope2d(float *tokens, const int64_t *positions, const float *cos_tab, const float *sin_tab, int B, int N, int H,
           int D, int P, int64_t sb, int64_t sn, int64_t sh, float sign, hipStream_t stream);
int attention_fwd(const VitAttnArgs &a, const float *q, const float *k, const float *v, float *out, float *lse,
                  hipStream_t stream);
int attention_bwd(const VitAttnArgs &a, const float *q, const float *k, const float *v, const float *out, const float *lse,
                  const float *dout, float *dq, float *dk, float *dv, float *delta_ws, hipStream_t stream);
int linear_fwd(const float *x, const float *w, const float *bias, const float *residual, float *out, float *pre, int M, int N,
               int K, int act, hipStream_t stream);
int split_weight(const float *w, void *packed, int rows, int cols, int transpose, hipStream_t stream);
int split_conv_weight_pair(const float *w, void *packed_fwd, void *packed_dx, int Co, int Ci, int ksize, hipStream_t stream);
int conv_x6_fwd(const float *in, const void *wp, const float *bias, const float *residual, float *out, int B, int Ci, int Co,
                int H, int W, int ksize, int relu_in, hipStream_t stream);
int conv_x6_wgrad(const float *dy, const float *in, float *dw, float *dbias, int B, int Ci, int Co, int H, int W, int ksize,
                  int relu_in, hipStream_t stream);
int upsample2x_fwd(const float *in, float *out, int64_t planes, int H, int W, hipStream_t stream);
int relu_dropout_fwd(const float *x, float *y, int64_t n, float p, uint64_t seed, hipStream_t stream);
int relu_dropout_bwd(const float *y, const float *g, float *dx, int64_t n, float p, hipStream_t stream);
size_t layernorm_scratch_bytes(int M, int C);
int layernorm_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *rstd, int M, int C,
                  float eps, hipStream_t stream);
int layernorm_bwd(const float *dy, const float *x, const float *mean, const float *rstd, const float *gamma, const float *dskip,
                  float *dx, float *dgamma, float *dbeta, float *scratch, int M, int C, int accumulate, hipStream_t stream);
int upsample2x_bwd(const float *dout, float *din, int64_t planes, int H, int W, hipStream_t stream);
int upsample2x_add_relu_fwd(const float *in, const float *addend, float *out, int64_t planes, int H, int W, hipStream_t stream);
int im2col7(const float *img, float *cols, int B, int H, int W, hipStream_t stream);
int im2col3_rows(const float *in, float *cols, int B, int Ci, int H, int W, int relu, hipStream_t stream);
int head_tail_fwd(const float *h, const float *w, const float *bias, float *y, int B, int C, int CO, int64_t HW, float p, uint64_t seed,
                  hipStream_t stream);
int head_tail_bwd(const float *h, const float *w, const float *dy, float *dh, float *dw, float *db, int B, int C, int CO, int64_t HW, float p,
                  uint64_t seed, hipStream_t stream);
int adapter_fwd(const VitAdapterArgs *a, float *means, float *cov, float *sh, float *opac, float *scales, float *rot, hipStream_t s);
int adapter_bwd(const VitAdapterArgs *a, const float *d_means, const float *d_cov, const float *d_sh, const float *d_opac,
                float *d_pts0, float *d_ptsr, float *d_par0, float *d_parr, float *d_app, hipStream_t s);
int linear_x6_wgrad(const float *dy, const float *x, float *dw, float *dbias, int M, int N, int K, int accumulate,
                    hipStream_t stream);
int attention_set_arith(int mode);
int attention_arith();
size_t x6c_workspace_bytes(int M, int N, int splits);
int x6c_choose_splits(int M, int N, int K);
int linear_x6c_fwd(const float *x, const void *wp, const float *bias, const float *residual, float *out, float *pre, int M, int N, int K,
                   int act, int splits, void *workspace, size_t workspace_bytes, hipStream_t stream);
int adamw_step(const VitAdamChunk *chunks, int n_chunks, float lr, float beta1, float beta2, float eps, float weight_decay,
               const float *grad_scale, hipStream_t stream);
int x6_set_products(int n);
int x6_products();
int x6_set_operand_amax(const void *a, const void *b);
int x6_set_output_amax(void *word);
int amax(const float *x, int64_t n, void *out, hipStream_t stream);
int split_weight_block(const float *w, void *packed, int rows, int cols, int transpose, hipStream_t stream);
int split_weight_pair(const float *w, void *packed_f, void *packed_t, int rows, int cols, int block_f, int block_t, hipStream_t stream);
int split_weights_many(const VitSplitJob *jobs_dev, int njobs, uint32_t total_blocks, hipStream_t stream);
int linear_x6r_fwd(const float *x, const void *wp, const float *bias, const float *residual, float *out, float *pre, int M, int N,
                   int K, int act, int cfg, hipStream_t stream);
int linear_sm_set(int max_rows, int tm, int nw);
int linear_sm_ok(int M, int N, int K);
int linear_sm_grouped(const float *const *x, const void *const *wpb, const float *const *bias, const float *const *residual, float *const *out,
                      int groups, int M, int N, int K, int act, hipStream_t stream);
int layernorm_fwd_grouped(const float *const *x, const float *const *gamma, const float *const *beta, float *const *y, int groups, int M, int C,
                          float eps, hipStream_t stream);
int linear_x6_fwd(const float *x, const void *wp, const float *bias, const float *residual, float *out, float *pre, int M, int N,
                  int K, int act, hipStream_t stream);
}  // namespace vit

#define VIT_EXPORT extern "C" __attribute__((visibility("default")))

VIT_EXPORT int vit_rope2d(float *tokens, const int64_t *positions, const float *cos_tab, const float *sin_tab, int B,
                          int N, int H, int D, int P, int64_t stride_b, int64_t stride_n, int64_t stride_h, float sign,
                          void *stream)
{
    return vit::rope2d(tokens, positions, cos_tab, sin_tab, B, N, H, D, P, stride_b, stride_n, stride_h, sign,
                       static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_attention_fwd(const VitAttnArgs *a, const float *q, const float *k, const float *v, float *out,
                                 float *lse, void *stream)
{
    if (!a) return VIT_EINVAL;
    return vit::attention_fwd(*a, q, k, v, out, lse, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_attention_bwd(const VitAttnArgs *a, const float *q, const float *k, const float *v, const float *out,
                                 const float *lse, const float *dout, float *dq, float *dk, float *dv, float *delta_ws,
                                 void *stream)
{
    if (!a) return VIT_EINVAL;
    return vit::attention_bwd(*a, q, k, v, out, lse, dout, dq, dk, dv, delta_ws, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_linear_fwd(const float *x, const float *w, const float *bias, const float *residual, float *out,
                              float *pre, int M, int N, int K, int act, void *stream)
{
    return vit::linear_fwd(x, w, bias, residual, out, pre, M, N, K, act, static_cast<hipStream_t>(stream));
}

// (+ 8 KiB: the |max| word -- 64 slots, one per cache line -- of the f16x3 mode right behind the pieces, vit_x6_set_products(2))
VIT_EXPORT size_t vit_split_weight_bytes(int rows, int cols) { return (size_t)rows * (size_t)cols * 6 + 8192; }

VIT_EXPORT int vit_split_weight(const float *w, void *packed, int rows, int cols, int transpose, void *stream)
{
    return vit::split_weight(w, packed, rows, cols, transpose, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_linear_x6_fwd(const float *x, const void *w_packed, const float *bias, const float *residual, float *out,
                                 float *pre, int M, int N, int K, int act, void *stream)
{
    return vit::linear_x6_fwd(x, w_packed, bias, residual, out, pre, M, N, K, act, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_attention_set_arith(int mode) { return vit::attention_set_arith(mode); }
VIT_EXPORT int vit_attention_arith(void) { return vit::attention_arith(); }
VIT_EXPORT int vit_adamw_step(const VitAdamChunk *chunks, int n_chunks, float lr, float beta1, float beta2, float eps, float weight_decay,
                              const float *grad_scale, void *stream)
{
    return vit::adamw_step(chunks, n_chunks, lr, beta1, beta2, eps, weight_decay, grad_scale, static_cast<hipStream_t>(stream));
}
VIT_EXPORT int vit_x6_set_products(int n) { return vit::x6_set_products(n); }
VIT_EXPORT int vit_x6_products(void) { return vit::x6_products(); }
VIT_EXPORT int vit_x6_set_operand_amax(const void *a, const void *b) { return vit::x6_set_operand_amax(a, b); }
VIT_EXPORT int vit_x6_set_output_amax(void *word) { return vit::x6_set_output_amax(word); }
VIT_EXPORT int vit_amax(const float *x, int64_t n, void *out_word, void *stream) { return vit::amax(x, n, out_word, static_cast<hipStream_t>(stream)); }

VIT_EXPORT size_t vit_split_weight_block_bytes(int rows, int cols, int transpose)
{
    const size_t R = transpose ? cols : rows, K = transpose ? rows : cols;
    return ((R + 63) / 64) * 64 * K * 6 + 8192;      // (+ 8 KiB: the weight's |max| word of the f16x3 mode sits right behind the pieces)
}

VIT_EXPORT int vit_split_weight_block(const float *w, void *packed, int rows, int cols, int transpose, void *stream)
{
    return vit::split_weight_block(w, packed, rows, cols, transpose, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_split_conv_weight_pair(const float *w, void *packed_fwd, void *packed_dx, int Co, int Ci, int ksize, void *stream)
{
    return vit::split_conv_weight_pair(w, packed_fwd, packed_dx, Co, Ci, ksize, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_split_weight_pair(const float *w, void *packed_fwd, void *packed_t, int rows, int cols, int block_fwd, int block_t, void *stream)
{
    return vit::split_weight_pair(w, packed_fwd, packed_t, rows, cols, block_fwd, block_t, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_split_weights_many(const VitSplitJob *jobs_device, int n_jobs, uint32_t total_blocks, void *stream)
{
    return vit::split_weights_many(jobs_device, n_jobs, total_blocks, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_linear_sm_set(int max_rows, int tile_row_blocks, int waves) { return vit::linear_sm_set(max_rows, tile_row_blocks, waves); }
VIT_EXPORT int vit_linear_sm_ok(int M, int N, int K) { return vit::linear_sm_ok(M, N, K); }
VIT_EXPORT int vit_linear_sm_grouped(const float *const *x, const void *const *w_block, const float *const *bias, const float *const *residual,
                                    float *const *out, int groups, int M, int N, int K, int act, void *stream)
{
    return vit::linear_sm_grouped(x, w_block, bias, residual, out, groups, M, N, K, act, static_cast<hipStream_t>(stream));
}
VIT_EXPORT int vit_layernorm_fwd_grouped(const float *const *x, const float *const *gamma, const float *const *beta, float *const *y, int groups,
                                         int M, int C, float eps, void *stream)
{
    return vit::layernorm_fwd_grouped(x, gamma, beta, y, groups, M, C, eps, static_cast<hipStream_t>(stream));
}

VIT_EXPORT size_t vit_linear_x6c_workspace_bytes(int M, int N, int splits) { return vit::x6c_workspace_bytes(M, N, splits); }
VIT_EXPORT int vit_linear_x6c_choose_splits(int M, int N, int K) { return vit::x6c_choose_splits(M, N, K); }

VIT_EXPORT int vit_linear_x6c_fwd(const float *x, const void *w_packed, const float *bias, const float *residual, float *out, float *pre,
                                  int M, int N, int K, int act, int splits, void *workspace, size_t workspace_bytes, void *stream)
{
    return vit::linear_x6c_fwd(x, w_packed, bias, residual, out, pre, M, N, K, act, splits, workspace, workspace_bytes,
                               static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_linear_x6r_fwd(const float *x, const void *w_packed, const float *bias, const float *residual, float *out,
                                  float *pre, int M, int N, int K, int act, int cfg, void *stream)
{
    return vit::linear_x6r_fwd(x, w_packed, bias, residual, out, pre, M, N, K, act, cfg, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_linear_x6_wgrad(const float *dy, const float *x, float *dw, float *dbias, int M, int N, int K, void *stream)
{
    return vit::linear_x6_wgrad(dy, x, dw, dbias, M, N, K, 0, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_linear_x6_wgrad_acc(const float *dy, const float *x, float *dw, float *dbias, int M, int N, int K, void *stream)
{
    return vit::linear_x6_wgrad(dy, x, dw, dbias, M, N, K, 1, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_conv_x6_fwd(const float *in, const void *w_packed, const float *bias, const float *residual, float *out,
                               int B, int Ci, int Co, int H, int W, int ksize, int relu_in, void *stream)
{
    return vit::conv_x6_fwd(in, w_packed, bias, residual, out, B, Ci, Co, H, W, ksize, relu_in, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_conv_x6_wgrad(const float *dy, const float *in, float *dw, float *dbias, int B, int Ci, int Co, int H, int W,
                                 int ksize, int relu_in, void *stream)
{
    return vit::conv_x6_wgrad(dy, in, dw, dbias, B, Ci, Co, H, W, ksize, relu_in, static_cast<hipStream_t>(stream));
}

VIT_EXPORT size_t vit_layernorm_scratch_bytes(int M, int C) { return vit::layernorm_scratch_bytes(M, C); }

VIT_EXPORT int vit_layernorm_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *rstd, int M,
                                 int C, float eps, void *stream)
{
    return vit::layernorm_fwd(x, gamma, beta, y, mean, rstd, M, C, eps, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_layernorm_bwd(const float *dy, const float *x, const float *mean, const float *rstd, const float *gamma,
                                 const float *dskip, float *dx, float *dgamma, float *dbeta, void *scratch, int M, int C,
                                 int accumulate, void *stream)
{
    return vit::layernorm_bwd(dy, x, mean, rstd, gamma, dskip, dx, dgamma, dbeta, static_cast<float *>(scratch), M, C, accumulate,
                              static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_relu_dropout_fwd(const float *x, float *y, int64_t n, float p, uint64_t seed, void *stream)
{
    return vit::relu_dropout_fwd(x, y, n, p, seed, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_relu_dropout_bwd(const float *y, const float *g, float *dx, int64_t n, float p, void *stream)
{
    return vit::relu_dropout_bwd(y, g, dx, n, p, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_upsample2x_fwd(const float *in, float *out, int64_t planes, int H, int W, void *stream)
{
    return vit::upsample2x_fwd(in, out, planes, H, W, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_upsample2x_bwd(const float *dout, float *din, int64_t planes, int H, int W, void *stream)
{
    return vit::upsample2x_bwd(dout, din, planes, H, W, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_upsample2x_add_relu_fwd(const float *in, const float *addend, float *out, int64_t planes, int H, int W, void *stream)
{
    return vit::upsample2x_add_relu_fwd(in, addend, out, planes, H, W, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_im2col7(const float *img, float *cols, int B, int H, int W, void *stream)
{
    return vit::im2col7(img, cols, B, H, W, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_im2col3_rows(const float *in, float *cols, int B, int Ci, int H, int W, int relu, void *stream)
{
    return vit::im2col3_rows(in, cols, B, Ci, H, W, relu, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_head_tail_fwd(const float *h, const float *w, const float *bias, float *y, int B, int C, int CO, int64_t HW, float p,
                                 uint64_t seed, void *stream)
{
    return vit::head_tail_fwd(h, w, bias, y, B, C, CO, HW, p, seed, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_head_tail_bwd(const float *h, const float *w, const float *dy, float *dh, float *dw, float *db, int B, int C, int CO,
                                 int64_t HW, float p, uint64_t seed, void *stream)
{
    return vit::head_tail_bwd(h, w, dy, dh, dw, db, B, C, CO, HW, p, seed, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_adapter_fwd(const VitAdapterArgs *a, float *means, float *cov, float *sh, float *opac, float *scales, float *rot,
                               void *stream)
{
    return vit::adapter_fwd(a, means, cov, sh, opac, scales, rot, static_cast<hipStream_t>(stream));
}

VIT_EXPORT int vit_adapter_bwd(const VitAdapterArgs *a, const float *d_means, const float *d_cov, const float *d_sh,
                               const float *d_opac, float *d_pts0, float *d_ptsr, float *d_par0, float *d_parr, float *d_app,
                               void *stream)
{
    return vit::adapter_bwd(a, d_means, d_cov, d_sh, d_opac, d_pts0, d_ptsr, d_par0, d_parr, d_app, static_cast<hipStream_t>(stream));
}

VIT_EXPORT const char *vit_version(void) { return "vit-hip gfx950 0.1.0"; }
VIT_EXPORT const char *vit_last_error(void)
{
    return vit::g_last_hip_error == hipSuccess ? "" : hipGetErrorString(vit::g_last_hip_error);
}
