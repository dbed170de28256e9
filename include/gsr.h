/*
 * gsr.h -- C ABI of libgsr_hip.so: the MI355X (gfx950) Gaussian-splatting
 * rasterizer that replaces the third-party CUDA extension
 * `diff_gaussian_rasterization` (requirements.txt:17 of the reference) behind
 * the reference's own call site src/model/decoder/cuda_splatting.py:101-129.
 *
 * Plain C: device pointers, sizes, a hipStream_t passed as void*.  No torch
 * types, no global state, re-entrant per stream, never throws; every entry
 * point returns 0 or a negative GSR_E* code.  The caller owns ALL memory
 * (outputs, the opaque workspace that carries geometry/binning/image state
 * from forward to backward -- the counterpart of upstream's geomBuffer /
 * binningBuffer / imgBuffer).
 *
 * Reference interface each entry point replaces:
 *   gsr_forward   <- _C.rasterize_gaussians(...)           as invoked by
 *                    GaussianRasterizer.forward, cuda_splatting.py:120-129
 *                    (one call per view there; V views per call here)
 *   gsr_backward  <- _C.rasterize_gaussians_backward(...)  invoked by autograd
 *                    for the same call (grads for means3D, means2D, shs /
 *                    colors_precomp, opacities, cov3D_precomp, theta, rho)
 *   GsrView       <- GaussianRasterizationSettings fields  cuda_splatting.py:101-115
 */
#ifndef GSR_H
#define GSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_OK 0
#define GSR_EINVAL (-1)     /* bad dimension / null pointer / unsupported degree */
#define GSR_ENOSPACE (-2)   /* workspace_bytes too small for (dims, pair_capacity) */
#define GSR_ELAUNCH (-3)    /* a kernel launch failed (hipGetLastError != success) */

#define GSR_TILE 16
#define GSR_VIEW_FLOATS 64

/* Per-view camera block, 64 floats = 256 B, array of V in DEVICE memory.
 * Matrices are the row-vector ("transposed") 4x4 form the reference passes
 * (cuda_splatting.py:85-88), flat index m[4*r+c]. */
typedef struct GsrView {
    float viewmatrix[16];     /* settings.viewmatrix      (:108) */
    float projmatrix[16];     /* settings.projmatrix      (:109) */
    float projmatrix_raw[16]; /* settings.projmatrix_raw  (:110) */
    float campos[3];          /* settings.campos          (:112) */
    float tanfovx, tanfovy;   /* settings.tanfovx/y       (:104-105) */
    float bg[3];              /* settings.bg              (:106) */
    float scale;              /* scene scale folded into the kernel: means*scale, cov*scale^2
                                 (the make_scale_invariant step, cuda_splatting.py:65-72); 1 = none */
    float pad[7];
} GsrView;

/* Problem dimensions.  V = B * Vt views are rendered per call; view v looks at
 * the Gaussians of scene v / Vt (the (b v) flattening of
 * decoder_splatting_cuda.py:51-63 WITHOUT replicating the Gaussian arrays). */
typedef struct GsrDims {
    int32_t B;          /* scenes (independent Gaussian sets) */
    int32_t Vt;         /* views per scene */
    int32_t G;          /* Gaussians per scene */
    int32_t H, W;       /* image size (all views) */
    int32_t M;          /* SH coefficients per channel in `shs` (G,M,3); 0 => `shs` is precomputed RGB (G,3) */
    int32_t sh_degree;  /* active degree 0..4, (sh_degree+1)^2 <= M */
    int32_t flags;      /* GSR_FLAG_* */
    void *profile;      /* optional GsrProfile* (gsr_profile_create): per-stage hipEvent timing; NULL = off */
} GsrDims;

#define GSR_FLAG_NTOUCHED 1  /* forward: also count n_touched (one extra global atomic per tile and splat) */
#define GSR_FLAG_COV9 2      /* cov6 / dL_dcov6 are full row-major 3x3 matrices (B,G,3,3): the kernels read the
                                upper triangle and write the gradient there (lower triangle 0), which is what
                                autograd produces for covariances[:, triu] (cuda_splatting.py:118,126) */
#define GSR_FLAG_PHASE_BIN 4     /* gsr_forward: enqueue only preprocess + tile scan.  `status` is FINAL after them
                                    (pair count, overflow), so the host can read it back while nothing expensive is
                                    queued, grow the workspace if needed, and then ...                              */
#define GSR_FLAG_PHASE_RENDER 8  /* ... call gsr_forward again with identical arguments + this flag: scatter, per-tile
                                    sort and composite run on the workspace the PHASE_BIN call prepared.  The host
                                    then returns with ~1 ms of GPU work still queued, which hides the launch latency
                                    of everything it enqueues next (loss, backward).  Neither flag: all stages.     */

#define GSR_FLAG_PREZERO_GRADS 16 /* gsr_forward: the composite kernel also zeroes the workspace's per-(view, Gaussian) gradient
                                    accumulators (126 MB at the headline size; that kernel is VALU-bound, HBM is idle under it).
                                    gsr_backward with the same flag then skips its own memset.  Valid for the FIRST backward
                                    after the forward only: the backward leaves the accumulators dirty. */
#define GSR_FLAG_BIN_BALLOT 32    /* K1 / K3 bin with wave-aggregated global atomics (the path of images beyond 4 096 tiles per view) even where
                                    the per-workgroup LDS histograms apply: A/B runs and the equivalence test */
#define GSR_FLAG_SORT_KEYS_SHIFT 8   /* bits 8-9: LDS budget of the per-tile depth sort: 0 = 4096 keys (default), 1 = 1024,
                                       2 = 2048.  Pick the smallest budget >= the longest per-tile list expected
                                       (status[GSR_ST_MAX_TILE] of an earlier call): more workgroups fit a CU.  Longer lists
                                       remain correct (sorted in global memory), only slower. */

/* status words written by gsr_forward (device int32[GSR_STATUS_WORDS]) */
#define GSR_STATUS_WORDS 8
#define GSR_ST_PAIRS 0       /* R: total (tile, Gaussian) pairs over all views (low 32 bits) */
#define GSR_ST_OVERFLOW 1    /* 1 if R > pair_capacity: outputs are INVALID, re-run with a larger capacity */
#define GSR_ST_MAX_TILE 2    /* longest per-tile list */
#define GSR_ST_PAIRS_HI 3    /* high 32 bits of R */

/* Named offsets into the workspace (bytes), for the parity tests and the bench. */
typedef struct GsrLayout {
    size_t records;      /* SplatRec[V*G], 48 B each: x,y,depth,radius+flags | conic A,B,C,opacity | r,g,b,extents */
    size_t tile_count;   /* uint32[V*T]     per-tile list length */
    size_t tile_offset;  /* uint32[V*T+1]   exclusive scan; ranges[t] = [off[t], off[t+1]) */
    size_t tile_cursor;  /* uint32[V*T]     scatter cursors */
    size_t pairs;        /* uint64[cap]     (depth_bits << 32 | id), bucketed by (view, tile) */
    size_t point_list;   /* uint32[cap]     per-tile depth-sorted Gaussian ids in bits 0..27 (GSR_ID_MASK; G < 2^28 by the record-offset limit
                            below).  Bits 28..31 are written by the composite forward and read by the backward: bit 28 + q = the entry's
                            alpha >= 1/255 footprint can reach 8x8 quadrant q (TL, TR, BL, BR) of its tile -- what the forward hands
                            the backward instead of a separate mask stream (round 6; rounds 2-5: a byte, then a 32-bit word per entry) */
    size_t final_T;      /* float[V*H*W] */
    size_t n_contrib;    /* uint32[V*H*W] */
    size_t grad_rec;     /* float[V*G*12]   backward per-(view,Gaussian) accumulators */
    size_t status;       /* int32[GSR_STATUS_WORDS] internal copy */
    size_t tile_order;   /* uint32[V*T]     (view*T + tile) ids, longest list first: launch order of the composite kernels */
    size_t pairs_alt;    /* uint64[cap]     bucket space of the per-tile sort for lists longer than its LDS budget */
    size_t loss_partial; /* float[V*T + V]  fused MSE: per-tile sums of squared errors, then per-view sums (fixed-order reduction) */
    size_t loss_ticket;  /* uint32[V + 1]   fused MSE: arrival counters of a view's tiles / of the views; zeroed by the tile scan */
    size_t loss_diff;    /* float[V*3*H*W]  fused MSE: image - target as the composite forward had it in registers; the backward scales it
                            to dL/dimage (one 12-byte read per pixel, as with a dL_dimage from outside) */
    size_t total;        /* total bytes */
} GsrLayout;

#define GSR_ID_MASK 0x0fffffffu
#define GSR_QUAD_SHIFT 28

/* Size/offsets of the workspace for (dims, pair_capacity).  Returns GSR_OK or GSR_EINVAL.
 * Limits: pair_capacity < 2^32; G * 48 bytes < 2^32 (the composite kernels address one view's records with 32-bit byte offsets; hence
 * ids < 2^27 and the four mask bits of a point_list word are free). */
int gsr_workspace_layout(const GsrDims *dims, int64_t pair_capacity, GsrLayout *out);

/*
 * Forward for V = B*Vt views.  All pointers are device pointers.
 *   views   GsrView[V]
 *   means   float (B,G,3); cov6 float (B,G,6) xx,xy,xz,yy,yz,zz; opac float (B,G);
 *   shs     float (B,G,M,3) or, when M == 0, RGB (B,G,3)
 * Outputs: image (V,3,H,W), depth (V,H,W), opacity (V,H,W), radii int32 (V,G),
 *          n_touched int32 (V,G) (may be NULL unless GSR_FLAG_NTOUCHED),
 *          status int32[GSR_STATUS_WORDS] -- device memory, or device-accessible pinned HOST memory: the tile scan stores the
 *          words there directly and the host reads them behind an event, no copy kernel.
 * The call only enqueues work on `stream` (no host sync).  If status[GSR_ST_OVERFLOW]
 * is set the images are invalid and the call must be repeated with
 * pair_capacity >= status[GSR_ST_PAIRS].
 */
int gsr_forward(const GsrDims *dims, const GsrView *views, const float *means, const float *cov6,
                const float *opac, const float *shs, int64_t pair_capacity, void *workspace,
                size_t workspace_bytes, float *image, float *depth, float *opacity, int32_t *radii,
                int32_t *n_touched, int32_t *status, void *stream);

/*
 * Backward of the same call (workspace must be the one the forward filled).
 *   dL_dimage (V,3,H,W); dL_ddepth (V,H,W) or NULL.
 * Outputs (overwritten): dL_dmeans (B,G,3), dL_dcov6 (B,G,6), dL_dopac (B,G),
 *   dL_dshs (B,G,M,3) or (B,G,3); optional (NULL to skip): dL_dmeans2D (V,G,3)
 *   screen-space mean gradient (z = 0), dL_dtau (V,6) = (rho, theta) pose gradient
 *   of a left se(3) perturbation of each view's world->camera transform.
 */
int gsr_backward(const GsrDims *dims, const GsrView *views, const float *means, const float *cov6,
                 const float *shs, int64_t pair_capacity, void *workspace, size_t workspace_bytes,
                 const float *dL_dimage, const float *dL_ddepth, float *dL_dmeans, float *dL_dcov6,
                 float *dL_dopac, float *dL_dshs, float *dL_dmeans2D, float *dL_dtau, void *stream);

/*
 * The same two calls with the per-step host / launch overhead folded into the kernels (round 6).  `fx` may be NULL (= the plain calls).
 *   tile_count   optional PERSISTENT per-tile counters, uint32[V*T] of device memory the caller zeroed ONCE at allocation and keeps per
 *                stream: the preprocess counts into them and the tile scan leaves them zero again, so no memset runs in front of the
 *                forward.  NULL: the counters of the workspace, zeroed by the call itself.
 *   mse_target   optional (V,3,H,W): `LossMse.forward` (src/loss/loss_mse.py:22-31) fused into the composite kernels --
 *                forward: the composite kernel sums (image - target)^2 per tile while the pixels are still in registers, the last tile of
 *                  every view and the last view add the partials in index order (deterministic), mse_loss[0] = mse_weight * mean;
 *                backward: the composite kernel forms dL/dimage = dL_dimage (may be NULL) + 2 mse_weight / n * mse_grad_loss[0] *
 *                  (image - target) in its prologue from the difference the forward left in the workspace -- two kernels less than
 *                  gsr_mse_forward / gsr_mse_backward around the plain calls, same values.  (The backward needs mse_target only as the
 *                  "fused" switch and mse_weight / mse_grad_loss for the coefficient.)
 *   mse_grad_loss  device fp32[1], the upstream gradient of the loss; NULL = 1.
 */
typedef struct GsrFused {
    uint32_t *tile_count;
    const float *mse_target;
    float mse_weight;
    float *mse_loss;             /* forward: out, device fp32[1] */
    const float *mse_grad_loss;  /* backward: in */
} GsrFused;
int gsr_forward_fused(const GsrDims *dims, const GsrView *views, const float *means, const float *cov6,
                      const float *opac, const float *shs, int64_t pair_capacity, void *workspace,
                      size_t workspace_bytes, float *image, float *depth, float *opacity, int32_t *radii,
                      int32_t *n_touched, int32_t *status, const GsrFused *fx, void *stream);
int gsr_backward_fused(const GsrDims *dims, const GsrView *views, const float *means, const float *cov6,
                       const float *shs, int64_t pair_capacity, void *workspace, size_t workspace_bytes,
                       const float *dL_dimage, const float *dL_ddepth, float *dL_dmeans, float *dL_dcov6,
                       float *dL_dopac, float *dL_dshs, float *dL_dmeans2D, float *dL_dtau, const GsrFused *fx,
                       void *stream);

/*
 * Camera set-up of `render_cuda` (cuda_splatting.py:65-88) for V views in ONE launch: the 1/near rescale
 * of make_scale_invariant, get_fov (projection.py:247-261), get_projection_matrix (:16-43), inverse(c2w)^T
 * and view @ proj, written as GsrView[V].  Replaces ~100 tiny device ops of the torch formulation.
 *   c2w (V,4,4) camera-to-world, K (V,3,3) normalised intrinsics, near/far (V), bg (V,3) -- device, row-major.
 */
int gsr_build_views(const float *c2w, const float *K, const float *near, const float *far, const float *bg, int32_t V,
                    int32_t scale_invariant, GsrView *out, void *stream);

/*
 * MSE consumer of the rendered colour, `LossMse.forward` (src/loss/loss_mse.py:22-31):
 *   loss = weight * mean((pred - target)^2)            -- one launch, deterministic (ticketed partial sums)
 *   dL/dpred = (2 weight / n) * grad_loss[0] * (pred - target)   -- one launch, upstream gradient read on device
 * pred/target/grad_pred: device fp32[n], 16-byte aligned; loss, grad_loss: device fp32[1];
 * scratch: device buffer of gsr_mse_scratch_bytes() bytes, zeroed ONCE by the caller at allocation (the
 * kernel re-arms it), not shared between streams.
 */
size_t gsr_mse_scratch_bytes(void);
int gsr_mse_forward(const float *pred, const float *target, int64_t n, float weight, void *scratch, float *loss,
                    void *stream);
int gsr_mse_backward(const float *pred, const float *target, const float *grad_loss, int64_t n, float weight,
                     float *grad_pred, void *stream);

/*
 * Optional per-stage timing with hipEvents recorded on the caller's stream
 * between the kernels of gsr_forward / gsr_backward (bench.py's live roofline
 * measurement).  A profile holds event pairs for `max_calls` forward and
 * `max_calls` backward calls; gsr_profile_read waits for them, returns the
 * summed milliseconds and launch counts per stage and resets the profile.
 */
#define GSR_N_STAGES 7
#define GSR_STAGE_PREPROCESS 0
#define GSR_STAGE_SCAN 1
#define GSR_STAGE_SCATTER 2
#define GSR_STAGE_SORT 3
#define GSR_STAGE_COMPOSITE_FWD 4
#define GSR_STAGE_COMPOSITE_BWD 5
#define GSR_STAGE_PREPROCESS_BWD 6
typedef struct GsrProfile GsrProfile;
GsrProfile *gsr_profile_create(int max_calls);
void gsr_profile_destroy(GsrProfile *prof);
int gsr_profile_read(GsrProfile *prof, float *ms_sum /* [GSR_N_STAGES] */, int32_t *count /* [GSR_N_STAGES] */);
/* Restrict the timing to the stages whose bit (1 << GSR_STAGE_*) is set in `stage_mask` (default: all).  Every timed stage puts two event
 * records between kernels that would otherwise follow each other back to back (~5 us per boundary); a caller that wants the duration of the
 * composite kernels INSIDE a wall-clock-timed region asks for those two only.  Stages outside the mask read back with count 0. */
int gsr_profile_set_stages(GsrProfile *prof, uint32_t stage_mask);

/* Text of the HIP error behind the calling thread's last GSR_ELAUNCH ("" if none). */
const char *gsr_last_error(void);

/* Library / build identification ("gsr-hip gfx950 <version>"). */
const char *gsr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H */
