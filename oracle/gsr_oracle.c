/*
 * gsr_oracle.c -- CPU ORACLE for the Gaussian-splatting rasterizer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (styl3r_amd/, the
 * C-ABI library, the drop-in diff_gaussian_rasterization module) may import,
 * link or call this file.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * PARITY UNPINNED: the algorithm lives in a third-party dependency that is NOT
 * vendored in the reference tree: requirements.txt:17 installs
 *   git+https://github.com/rmurai0610/diff-gaussian-rasterization-w-pose.git
 * (no commit pin; a fork of graphdeco-inria/diff-gaussian-rasterization with
 * the MonoGS pose-gradient additions).  It is CUDA-only, absent from this
 * image and there is no network, so no golden vector from the reference can
 * pin this restatement.  The restatement follows the published 3DGS algorithm
 * (upstream cuda_rasterizer/{forward,backward,
 * rasterizer_impl}.cu) and is anchored on the reference's own call site
 * src/model/decoder/cuda_splatting.py:101-129 (argument layouts, the 5-tuple
 * return).  It pins itself with analytic known-answer cases, fp64
 * finite-difference gradient checks (tests/test_oracle_*.py) and an independent
 * dense PyTorch fp64 restatement whose gradients come from autograd
 * (tests/test_oracle_vs_autograd.py: images / depth / radii / n_contrib equal,
 * every analytic gradient incl. the pose tau within 1e-7).
 *
 * The file is compiled twice: -DGSO_REAL=float (libgsr_oracle_f32.so, mirrors
 * the fp32 arithmetic of the GPU path operation-for-operation; build with
 * -ffp-contract=off so integer outputs are bit-reproducible) and
 * -DGSO_REAL=double (libgsr_oracle_f64.so, the authority for gradients).
 *
 * Conventions (cuda_splatting.py:85-88): matrices are row-vector 4x4, flat
 * index m[4*r+c]; p_view[c] = sum_r p[r]*view[4r+c] + view[12+c].
 * cov3D is the 6-vector xx,xy,xz,yy,yz,zz (= triu_indices order,
 * cuda_splatting.py:118,126).  SH layout (G, M, 3) coefficient-major
 * (cuda_splatting.py:76).  Tiles are 16x16, tile id = ty*grid_x + tx.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef GSO_REAL
#define GSO_REAL float
#endif
typedef GSO_REAL real;

#define TILE 16
#define TILE_PIX 256

static inline real r_sqrt(real x) { return sizeof(real) == 4 ? (real)sqrtf((float)x) : (real)sqrt((double)x); }
static inline real r_exp(real x) { return sizeof(real) == 4 ? (real)expf((float)x) : (real)exp((double)x); }
static inline real r_ceil(real x) { return sizeof(real) == 4 ? (real)ceilf((float)x) : (real)ceil((double)x); }
static inline real r_max(real a, real b) { return a > b ? a : b; }
static inline real r_min(real a, real b) { return a < b ? a : b; }

typedef struct {
    int32_t G;          /* number of Gaussians */
    int32_t H, W;       /* image size */
    int32_t sh_degree;  /* 0..4 active degree */
    int32_t M;          /* SH coefficients per Gaussian per channel; 0 => colours precomputed (G,3) */
    int32_t want_tau;   /* backward: also produce dL/dtau */
    real tanfovx, tanfovy;
    real bg[3];
    real view[16];      /* viewmatrix     (cuda_splatting.py:108) */
    real proj[16];      /* projmatrix     (full, :109) */
    real proj_raw[16];  /* projmatrix_raw (:110) */
    real campos[3];     /* :112 */
} gso_params;

/* real-SH constants: bands 0-3 are the published 3DGS tables; band 4 is the
 * build's own extension (standard orthonormal real SH, checked for
 * orthonormality in tests/test_oracle_gradients.py::test_sh_tables_orthonormal). */
static const double SH_C0 = 0.28209479177387814;
static const double SH_C1 = 0.4886025119029199;
static const double SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                                -1.0925484305920792, 0.5462742152960396};
static const double SH_C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                                -0.4570457994644658, 1.445305721320277, -0.5900435899266435};
static const double SH_C4[9] = {2.5033429417967046, -1.7701307697799304, 0.9461746957575601,
                                -0.6690465435572892, 0.10578554691520431, -0.6690465435572892,
                                0.47308734787878004, -1.7701307697799304, 0.6258357354491761};

/* SH basis values b[0..n) for unit direction (x,y,z), n = (deg+1)^2. */
static void sh_basis(int deg, real x, real y, real z, real *b)
{
    b[0] = (real)SH_C0;
    if (deg < 1) return;
    b[1] = -(real)SH_C1 * y;
    b[2] = (real)SH_C1 * z;
    b[3] = -(real)SH_C1 * x;
    if (deg < 2) return;
    real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = (real)SH_C2[0] * xy;
    b[5] = (real)SH_C2[1] * yz;
    b[6] = (real)SH_C2[2] * ((real)2 * zz - xx - yy);
    b[7] = (real)SH_C2[3] * xz;
    b[8] = (real)SH_C2[4] * (xx - yy);
    if (deg < 3) return;
    b[9] = (real)SH_C3[0] * y * ((real)3 * xx - yy);
    b[10] = (real)SH_C3[1] * xy * z;
    b[11] = (real)SH_C3[2] * y * ((real)4 * zz - xx - yy);
    b[12] = (real)SH_C3[3] * z * ((real)2 * zz - (real)3 * xx - (real)3 * yy);
    b[13] = (real)SH_C3[4] * x * ((real)4 * zz - xx - yy);
    b[14] = (real)SH_C3[5] * z * (xx - yy);
    b[15] = (real)SH_C3[6] * x * (xx - (real)3 * yy);
    if (deg < 4) return;
    b[16] = (real)SH_C4[0] * xy * (xx - yy);
    b[17] = (real)SH_C4[1] * yz * ((real)3 * xx - yy);
    b[18] = (real)SH_C4[2] * xy * ((real)7 * zz - (real)1);
    b[19] = (real)SH_C4[3] * yz * ((real)7 * zz - (real)3);
    b[20] = (real)SH_C4[4] * (zz * ((real)35 * zz - (real)30) + (real)3);
    b[21] = (real)SH_C4[5] * xz * ((real)7 * zz - (real)3);
    b[22] = (real)SH_C4[6] * (xx - yy) * ((real)7 * zz - (real)1);
    b[23] = (real)SH_C4[7] * xz * (xx - (real)3 * yy);
    b[24] = (real)SH_C4[8] * (xx * (xx - (real)3 * yy) - yy * ((real)3 * xx - yy));
}

/* d(basis)/d(x,y,z) treating x,y,z as free variables (not constrained to the
 * unit sphere); the chain through normalize() is applied by the caller. */
static void sh_basis_grad(int deg, real x, real y, real z, real *dx, real *dy, real *dz)
{
    dx[0] = dy[0] = dz[0] = 0;
    if (deg < 1) return;
    dx[1] = 0; dy[1] = -(real)SH_C1; dz[1] = 0;
    dx[2] = 0; dy[2] = 0; dz[2] = (real)SH_C1;
    dx[3] = -(real)SH_C1; dy[3] = 0; dz[3] = 0;
    if (deg < 2) return;
    real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    dx[4] = (real)SH_C2[0] * y; dy[4] = (real)SH_C2[0] * x; dz[4] = 0;
    dx[5] = 0; dy[5] = (real)SH_C2[1] * z; dz[5] = (real)SH_C2[1] * y;
    dx[6] = (real)SH_C2[2] * (-(real)2 * x); dy[6] = (real)SH_C2[2] * (-(real)2 * y); dz[6] = (real)SH_C2[2] * ((real)4 * z);
    dx[7] = (real)SH_C2[3] * z; dy[7] = 0; dz[7] = (real)SH_C2[3] * x;
    dx[8] = (real)SH_C2[4] * ((real)2 * x); dy[8] = (real)SH_C2[4] * (-(real)2 * y); dz[8] = 0;
    if (deg < 3) return;
    dx[9] = (real)SH_C3[0] * ((real)6 * xy); dy[9] = (real)SH_C3[0] * ((real)3 * xx - (real)3 * yy); dz[9] = 0;
    dx[10] = (real)SH_C3[1] * yz; dy[10] = (real)SH_C3[1] * xz; dz[10] = (real)SH_C3[1] * xy;
    dx[11] = (real)SH_C3[2] * (-(real)2 * xy); dy[11] = (real)SH_C3[2] * ((real)4 * zz - xx - (real)3 * yy); dz[11] = (real)SH_C3[2] * ((real)8 * yz);
    dx[12] = (real)SH_C3[3] * (-(real)6 * xz); dy[12] = (real)SH_C3[3] * (-(real)6 * yz); dz[12] = (real)SH_C3[3] * ((real)6 * zz - (real)3 * xx - (real)3 * yy);
    dx[13] = (real)SH_C3[4] * ((real)4 * zz - (real)3 * xx - yy); dy[13] = (real)SH_C3[4] * (-(real)2 * xy); dz[13] = (real)SH_C3[4] * ((real)8 * xz);
    dx[14] = (real)SH_C3[5] * ((real)2 * xz); dy[14] = (real)SH_C3[5] * (-(real)2 * yz); dz[14] = (real)SH_C3[5] * (xx - yy);
    dx[15] = (real)SH_C3[6] * ((real)3 * xx - (real)3 * yy); dy[15] = (real)SH_C3[6] * (-(real)6 * xy); dz[15] = 0;
    if (deg < 4) return;
    dx[16] = (real)SH_C4[0] * y * ((real)3 * xx - yy); dy[16] = (real)SH_C4[0] * x * (xx - (real)3 * yy); dz[16] = 0;
    dx[17] = (real)SH_C4[1] * ((real)6 * xy * z); dy[17] = (real)SH_C4[1] * z * ((real)3 * xx - (real)3 * yy); dz[17] = (real)SH_C4[1] * y * ((real)3 * xx - yy);
    dx[18] = (real)SH_C4[2] * y * ((real)7 * zz - (real)1); dy[18] = (real)SH_C4[2] * x * ((real)7 * zz - (real)1); dz[18] = (real)SH_C4[2] * ((real)14 * xy * z);
    dx[19] = 0; dy[19] = (real)SH_C4[3] * z * ((real)7 * zz - (real)3); dz[19] = (real)SH_C4[3] * y * ((real)21 * zz - (real)3);
    dx[20] = 0; dy[20] = 0; dz[20] = (real)SH_C4[4] * ((real)140 * zz * z - (real)60 * z);
    dx[21] = (real)SH_C4[5] * z * ((real)7 * zz - (real)3); dy[21] = 0; dz[21] = (real)SH_C4[5] * x * ((real)21 * zz - (real)3);
    dx[22] = (real)SH_C4[6] * ((real)2 * x) * ((real)7 * zz - (real)1); dy[22] = (real)SH_C4[6] * (-(real)2 * y) * ((real)7 * zz - (real)1); dz[22] = (real)SH_C4[6] * (xx - yy) * ((real)14 * z);
    dx[23] = (real)SH_C4[7] * z * ((real)3 * xx - (real)3 * yy); dy[23] = (real)SH_C4[7] * (-(real)6 * xy * z); dz[23] = (real)SH_C4[7] * x * (xx - (real)3 * yy);
    dx[24] = (real)SH_C4[8] * ((real)4 * xx * x - (real)12 * x * yy); dy[24] = (real)SH_C4[8] * (-(real)12 * xx * y + (real)4 * yy * y); dz[24] = 0;
}

/* exported for the orthonormality test */
void gso_sh_basis(int deg, real x, real y, real z, real *b) { sh_basis(deg, x, y, z, b); }
void gso_sh_basis_grad(int deg, real x, real y, real z, real *dx, real *dy, real *dz) { sh_basis_grad(deg, x, y, z, dx, dy, dz); }
int gso_sizeof_real(void) { return (int)sizeof(real); }
/* OpenMP threads of the per-Gaussian stages (preprocess, key emission + radix sort, preprocess backward); the per-tile
 * stages take theirs as an argument.  Every parallel loop writes disjoint outputs: results do not depend on the count. */
static int g_threads = 1;
void gso_set_threads(int n) { g_threads = n > 0 ? (n > 32 ? 32 : n) : 1; }   /* these stages are short: wide teams only buy fork / barrier time */
int gso_sizeof_params(void) { return (int)sizeof(gso_params); }

/* ------------------------------------------------------------------------
 * A1. preprocess  (upstream forward.cu preprocessCUDA / computeCov2D /
 * computeColorFromSH).  One Gaussian at a time, every
 * fp operation written out in the order the HIP kernel uses.
 * Outputs (all length G unless stated):
 *   depth, xy (G,2), conic_opacity (G,4), rgb (G,3), radii, tiles_touched,
 *   rect (G,4) = minx,miny,maxx,maxy in tiles, clamped (G,3) uint8.
 * ---------------------------------------------------------------------- */
typedef struct {
    real t[3];        /* camera-space mean (unclamped) */
    real txc, tyc;    /* clamped t.x, t.y used in J */
    int clampx, clampy;
    real J00, J02, J11, J12;
    real M0[3], M1[3];
    real a, b, c;     /* cov2D after +0.3 */
} gso_geom;

static int gso_geom_eval(const gso_params *p, const real *mean, const real *cov6, gso_geom *g)
{
    const real *V = p->view;
    real px = mean[0], py = mean[1], pz = mean[2];
    for (int c = 0; c < 3; ++c) g->t[c] = px * V[0 + c] + py * V[4 + c] + pz * V[8 + c] + V[12 + c];
    if (g->t[2] <= (real)0.2) return 0;
    real fx = (real)p->W / ((real)2 * p->tanfovx);
    real fy = (real)p->H / ((real)2 * p->tanfovy);
    real limx = (real)1.3 * p->tanfovx, limy = (real)1.3 * p->tanfovy;
    real tz = g->t[2];
    real txtz = g->t[0] / tz, tytz = g->t[1] / tz;
    real cx = r_min(limx, r_max(-limx, txtz));
    real cy = r_min(limy, r_max(-limy, tytz));
    g->clampx = (txtz < -limx || txtz > limx);
    g->clampy = (tytz < -limy || tytz > limy);
    g->txc = cx * tz;
    g->tyc = cy * tz;
    g->J00 = fx / tz;
    g->J02 = -(fx * g->txc) / (tz * tz);
    g->J11 = fy / tz;
    g->J12 = -(fy * g->tyc) / (tz * tz);
    /* R[i][j] = V[4*j+i] (column-vector world->camera rotation); M = J R */
    for (int j = 0; j < 3; ++j) {
        real R0 = V[4 * j + 0], R1 = V[4 * j + 1], R2 = V[4 * j + 2];
        g->M0[j] = g->J00 * R0 + g->J02 * R2;
        g->M1[j] = g->J11 * R1 + g->J12 * R2;
    }
    real S00 = cov6[0], S01 = cov6[1], S02 = cov6[2], S11 = cov6[3], S12 = cov6[4], S22 = cov6[5];
    real v0[3], v1[3];
    v0[0] = S00 * g->M0[0] + S01 * g->M0[1] + S02 * g->M0[2];
    v0[1] = S01 * g->M0[0] + S11 * g->M0[1] + S12 * g->M0[2];
    v0[2] = S02 * g->M0[0] + S12 * g->M0[1] + S22 * g->M0[2];
    v1[0] = S00 * g->M1[0] + S01 * g->M1[1] + S02 * g->M1[2];
    v1[1] = S01 * g->M1[0] + S11 * g->M1[1] + S12 * g->M1[2];
    v1[2] = S02 * g->M1[0] + S12 * g->M1[1] + S22 * g->M1[2];
    g->a = (g->M0[0] * v0[0] + g->M0[1] * v0[1] + g->M0[2] * v0[2]) + (real)0.3;
    g->b = g->M1[0] * v0[0] + g->M1[1] * v0[1] + g->M1[2] * v0[2];
    g->c = (g->M1[0] * v1[0] + g->M1[1] * v1[1] + g->M1[2] * v1[2]) + (real)0.3;
    return 1;
}

static void gso_eval_color(const gso_params *p, const real *mean, const real *sh /* (M,3) */, real *rgb, uint8_t *clamped)
{
    real dx = mean[0] - p->campos[0], dy = mean[1] - p->campos[1], dz = mean[2] - p->campos[2];
    real len = r_sqrt(dx * dx + dy * dy + dz * dz);
    real x = dx / len, y = dy / len, z = dz / len;
    real b[25];
    sh_basis(p->sh_degree, x, y, z, b);
    int n = (p->sh_degree + 1) * (p->sh_degree + 1);
    for (int c = 0; c < 3; ++c) {
        real acc = b[0] * sh[c];
        for (int k = 1; k < n; ++k) acc = acc + b[k] * sh[3 * k + c];
        acc = acc + (real)0.5;
        clamped[c] = acc < 0;
        rgb[c] = r_max(acc, (real)0);
    }
}

void gso_preprocess(const gso_params *p, const real *means, const real *cov6, const real *opac,
                    const real *shs_or_rgb, real *depth, real *xy, real *conic_opacity, real *rgb,
                    int32_t *radii, int32_t *tiles_touched, int32_t *rect, uint8_t *clamped)
{
    int gx = (p->W + TILE - 1) / TILE, gy = (p->H + TILE - 1) / TILE;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int i = 0; i < p->G; ++i) {
        radii[i] = 0; tiles_touched[i] = 0;
        depth[i] = 0; xy[2 * i] = xy[2 * i + 1] = 0;
        for (int k = 0; k < 4; ++k) { conic_opacity[4 * i + k] = 0; rect[4 * i + k] = 0; }
        for (int k = 0; k < 3; ++k) { rgb[3 * i + k] = 0; clamped[3 * i + k] = 0; }
        gso_geom g;
        const real *m = means + 3 * i;
        if (!gso_geom_eval(p, m, cov6 + 6 * i, &g)) continue;
        const real *P = p->proj;
        real hx = m[0] * P[0] + m[1] * P[4] + m[2] * P[8] + P[12];
        real hy = m[0] * P[1] + m[1] * P[5] + m[2] * P[9] + P[13];
        real hw = m[0] * P[3] + m[1] * P[7] + m[2] * P[11] + P[15];
        real pw = (real)1 / (hw + (real)0.0000001);
        real ndcx = hx * pw, ndcy = hy * pw;
        real det = g.a * g.c - g.b * g.b;
        if (det == (real)0) continue;
        real det_inv = (real)1 / det;
        real mid = (real)0.5 * (g.a + g.c);
        real disc = r_sqrt(r_max((real)0.1, mid * mid - det));
        real lam1 = mid + disc, lam2 = mid - disc;
        real radf = r_ceil((real)3 * r_sqrt(r_max(lam1, lam2)));
        int rad = (int)radf;
        real pxx = ((ndcx + (real)1) * (real)p->W - (real)1) * (real)0.5;
        real pxy = ((ndcy + (real)1) * (real)p->H - (real)1) * (real)0.5;
        int minx = (int)((pxx - (real)rad) / (real)TILE); if (minx < 0) minx = 0; if (minx > gx) minx = gx;
        int miny = (int)((pxy - (real)rad) / (real)TILE); if (miny < 0) miny = 0; if (miny > gy) miny = gy;
        int maxx = (int)((pxx + (real)rad + (real)(TILE - 1)) / (real)TILE); if (maxx < 0) maxx = 0; if (maxx > gx) maxx = gx;
        int maxy = (int)((pxy + (real)rad + (real)(TILE - 1)) / (real)TILE); if (maxy < 0) maxy = 0; if (maxy > gy) maxy = gy;
        if ((maxx - minx) * (maxy - miny) == 0) continue;
        if (p->M > 0) gso_eval_color(p, m, shs_or_rgb + (size_t)3 * p->M * i, rgb + 3 * i, clamped + 3 * i);
        else for (int c = 0; c < 3; ++c) rgb[3 * i + c] = shs_or_rgb[3 * i + c];
        depth[i] = g.t[2];
        radii[i] = rad;
        xy[2 * i] = pxx; xy[2 * i + 1] = pxy;
        conic_opacity[4 * i + 0] = g.c * det_inv;
        conic_opacity[4 * i + 1] = -g.b * det_inv;
        conic_opacity[4 * i + 2] = g.a * det_inv;
        conic_opacity[4 * i + 3] = opac[i];
        rect[4 * i + 0] = minx; rect[4 * i + 1] = miny; rect[4 * i + 2] = maxx; rect[4 * i + 3] = maxy;
        tiles_touched[i] = (maxx - minx) * (maxy - miny);
    }
}

/* ------------------------------------------------------------------------
 * A2. binning + sort (upstream rasterizer_impl.cu: InclusiveSum,
 * duplicateWithKeys, DeviceRadixSort::SortPairs, identifyTileRanges).
 * Keys are (tile << 32) | bits(float depth); the sort is a stable LSD radix
 * sort so equal-depth entries of one tile stay in ascending Gaussian id.
 * depth is always taken as FLOAT32 bits (the f64 build rounds to f32 first so
 * both builds agree on the ordering).
 * Returns R; writes point_list (R), ranges (T,2).  If cap < R nothing is
 * written to point_list and -R is returned.
 * ---------------------------------------------------------------------- */
int64_t gso_bin_sort(const gso_params *p, const real *depth, const int32_t *tiles_touched, const int32_t *rect,
                     int64_t cap, int32_t *point_list, int32_t *ranges)
{
    int gx = (p->W + TILE - 1) / TILE, gy = (p->H + TILE - 1) / TILE;
    int T = gx * gy;
    int64_t R = 0;
    for (int i = 0; i < p->G; ++i) R += tiles_touched[i];
    for (int t = 0; t < T; ++t) ranges[2 * t] = ranges[2 * t + 1] = 0;
    if (R > cap) return -R;
    if (R == 0) return 0;
    uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)R * 2);
    uint32_t *vals = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)R * 2);
    uint64_t *k0 = keys, *k1 = keys + R;
    uint32_t *v0 = vals, *v1 = vals + R;
    /* key emission in Gaussian order: offsets = exclusive scan of tiles_touched (serial, cheap), emission in parallel */
    int64_t *goff = (int64_t *)malloc(sizeof(int64_t) * (size_t)p->G);
    {
        int64_t acc = 0;
        for (int i = 0; i < p->G; ++i) { goff[i] = acc; acc += tiles_touched[i]; }
    }
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int i = 0; i < p->G; ++i) {
        if (tiles_touched[i] == 0) continue;
        float d = (float)depth[i];
        uint32_t bits; memcpy(&bits, &d, 4);
        int64_t off = goff[i];
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
            for (int x = rect[4 * i + 0]; x < rect[4 * i + 2]; ++x) {
                k0[off] = ((uint64_t)(uint32_t)(y * gx + x) << 32) | bits;
                v0[off] = (uint32_t)i;
                ++off;
            }
    }
    free(goff);
    /* stable LSD radix sort, 8 passes of 8 bits; each thread owns a contiguous chunk: per-(digit, thread) offsets keep the
     * order inside a digit = (thread, position) = the input order, i.e. the pass is stable exactly like the serial one */
    {
        int nt = g_threads > 16 ? 16 : g_threads;   /* barrier per pass: wide teams only add barrier time to this small stage */
        if ((int64_t)nt > R) nt = (int)R;
        size_t *hist = (size_t *)malloc(sizeof(size_t) * 256 * (size_t)nt);
        for (int pass = 0; pass < 8; ++pass) {
            int sh = pass * 8;
#pragma omp parallel num_threads(nt)
            {
                int t = omp_get_thread_num();
                int64_t lo = R * t / nt, hi = R * (t + 1) / nt;
                size_t *h = hist + 256 * (size_t)t;
                memset(h, 0, sizeof(size_t) * 256);
                for (int64_t j = lo; j < hi; ++j) h[(k0[j] >> sh) & 0xff]++;
#pragma omp barrier
#pragma omp single
                {
                    size_t acc = 0;
                    for (int d = 0; d < 256; ++d)
                        for (int u = 0; u < nt; ++u) { size_t c = hist[256 * (size_t)u + d]; hist[256 * (size_t)u + d] = acc; acc += c; }
                }
                for (int64_t j = lo; j < hi; ++j) {
                    size_t dst = h[(k0[j] >> sh) & 0xff]++;
                    k1[dst] = k0[j]; v1[dst] = v0[j];
                }
            }
            uint64_t *tk = k0; k0 = k1; k1 = tk;
            uint32_t *tv = v0; v0 = v1; v1 = tv;
        }
        free(hist);
    }
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int64_t j = 0; j < R; ++j) {
        point_list[j] = (int32_t)v0[j];
        uint32_t tile = (uint32_t)(k0[j] >> 32);
        if (j == 0 || (uint32_t)(k0[j - 1] >> 32) != tile) ranges[2 * tile] = (int32_t)j;
        if (j == R - 1 || (uint32_t)(k0[j + 1] >> 32) != tile) ranges[2 * tile + 1] = (int32_t)(j + 1);
    }
    free(keys); free(vals);
    return R;
}

/* ------------------------------------------------------------------------
 * A3. composite forward (upstream forward.cu renderCUDA + the -w-pose fork's
 * depth / opacity / n_touched outputs).
 * `fragile` (H*W uint8, may be NULL) marks pixels where some threshold test
 * (alpha<1/255, test_T<1e-4, test_T>0.5, power>0) was decided with a relative
 * margin below 1e-4 -- a different exp() rounding can legitimately flip those,
 * so integer comparisons (n_contrib, n_touched) exclude them.
 * ---------------------------------------------------------------------- */
void gso_render_fwd(const gso_params *p, const int32_t *point_list, const int32_t *ranges, const real *xy,
                    const real *conic_opacity, const real *rgb, const real *depth, real *image, real *out_depth,
                    real *out_opacity, real *final_T, int32_t *n_contrib, int32_t *n_touched, uint8_t *fragile,
                    int nthreads)
{
    int gx = (p->W + TILE - 1) / TILE, gy = (p->H + TILE - 1) / TILE;
    int H = p->H, W = p->W;
    for (int i = 0; i < p->G; ++i) n_touched[i] = 0;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int tile = 0; tile < gx * gy; ++tile) {
        int tx = tile % gx, ty = tile / gx;
        int s = ranges[2 * tile], e = ranges[2 * tile + 1];
        for (int ly = 0; ly < TILE; ++ly)
            for (int lx = 0; lx < TILE; ++lx) {
                int x = tx * TILE + lx, y = ty * TILE + ly;
                if (x >= W || y >= H) continue;
                real T = 1, C[3] = {0, 0, 0}, D = 0, O = 0;
                int contributor = 0, last = 0, frag = 0;
                for (int j = s; j < e; ++j) {
                    int id = point_list[j];
                    contributor++;
                    real dx = xy[2 * id] - (real)x, dy = xy[2 * id + 1] - (real)y;
                    real A = conic_opacity[4 * id], B = conic_opacity[4 * id + 1], Cc = conic_opacity[4 * id + 2];
                    real op = conic_opacity[4 * id + 3];
                    real power = (real)-0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy;
                    if (power > (real)0) { if (power < (real)1e-6) frag = 1; continue; }
                    if (power > (real)-1e-6) frag = 1;
                    real alpha = r_min((real)0.99, op * r_exp(power));
                    real thr = (real)1 / (real)255;
                    if (fabs((double)(alpha - thr)) < 1e-4 * (double)thr) frag = 1;
                    if (alpha < thr) continue;
                    real test_T = T * ((real)1 - alpha);
                    if (fabs((double)test_T - 1e-4) < 1e-8) frag = 1;
                    if (test_T < (real)0.0001) break;
                    real w = alpha * T;
                    C[0] += rgb[3 * id] * w; C[1] += rgb[3 * id + 1] * w; C[2] += rgb[3 * id + 2] * w;
                    D += depth[id] * w;
                    O += w;
                    if (fabs((double)test_T - 0.5) < 5e-5) frag = 1;
                    if (test_T > (real)0.5) {
#ifdef _OPENMP
#pragma omp atomic
#endif
                        n_touched[id]++;
                    }
                    T = test_T;
                    last = contributor;
                }
                int pix = y * W + x;
                final_T[pix] = T;
                n_contrib[pix] = last;
                for (int c = 0; c < 3; ++c) image[c * H * W + pix] = C[c] + T * p->bg[c];
                out_depth[pix] = D;
                out_opacity[pix] = O;
                if (fragile) fragile[pix] = (uint8_t)frag;
            }
    }
}

/* ------------------------------------------------------------------------
 * A4. composite backward (upstream backward.cu renderCUDA + fork's depth).
 * Accumulates into dL_dmean2D (G,2), dL_dconic (G,3: A, B(half convention), C),
 * dL_dopacity (G), dL_drgb (G,3), dL_ddepth (G).  Caller zeroes them.
 * Pixel order inside a tile and tile order are fixed (deterministic sums) when
 * nthreads <= 1.
 * ---------------------------------------------------------------------- */
void gso_render_bwd(const gso_params *p, const int32_t *point_list, const int32_t *ranges, const real *xy,
                    const real *conic_opacity, const real *rgb, const real *depth, const real *final_T,
                    const int32_t *n_contrib, const real *dL_dimage, const real *dL_dout_depth, real *dL_dmean2D,
                    real *dL_dconic, real *dL_dopacity, real *dL_drgb, real *dL_ddepth, int nthreads)
{
    int gx = (p->W + TILE - 1) / TILE, gy = (p->H + TILE - 1) / TILE;
    int H = p->H, W = p->W;
    real ddelx_dx = (real)0.5 * (real)W, ddely_dy = (real)0.5 * (real)H;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int tile = 0; tile < gx * gy; ++tile) {
        int tx = tile % gx, ty = tile / gx;
        int s = ranges[2 * tile];
        for (int ly = 0; ly < TILE; ++ly)
            for (int lx = 0; lx < TILE; ++lx) {
                int x = tx * TILE + lx, y = ty * TILE + ly;
                if (x >= W || y >= H) continue;
                int pix = y * W + x;
                real Tf = final_T[pix], T = Tf;
                int last = n_contrib[pix];
                real dLp[3] = {dL_dimage[pix], dL_dimage[H * W + pix], dL_dimage[2 * H * W + pix]};
                real dLd = dL_dout_depth ? dL_dout_depth[pix] : (real)0;
                real accum[3] = {0, 0, 0}, accum_d = 0, last_alpha = 0, last_col[3] = {0, 0, 0}, last_d = 0;
                real bg_dot = p->bg[0] * dLp[0] + p->bg[1] * dLp[1] + p->bg[2] * dLp[2];
                for (int j = s + last - 1; j >= s; --j) {
                    int id = point_list[j];
                    real dx = xy[2 * id] - (real)x, dy = xy[2 * id + 1] - (real)y;
                    real A = conic_opacity[4 * id], B = conic_opacity[4 * id + 1], Cc = conic_opacity[4 * id + 2];
                    real op = conic_opacity[4 * id + 3];
                    real power = (real)-0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy;
                    if (power > (real)0) continue;
                    real Gv = r_exp(power);
                    real alpha = r_min((real)0.99, op * Gv);
                    if (alpha < (real)1 / (real)255) continue;
                    T = T / ((real)1 - alpha);
                    real w = alpha * T;
                    real dL_dalpha = 0;
                    real add[10];
                    for (int c = 0; c < 3; ++c) {
                        real col = rgb[3 * id + c];
                        accum[c] = last_alpha * last_col[c] + ((real)1 - last_alpha) * accum[c];
                        last_col[c] = col;
                        dL_dalpha += (col - accum[c]) * dLp[c];
                        add[c] = w * dLp[c];
                    }
                    real dep = depth[id];
                    accum_d = last_alpha * last_d + ((real)1 - last_alpha) * accum_d;
                    last_d = dep;
                    dL_dalpha += (dep - accum_d) * dLd;
                    add[3] = w * dLd;
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-Tf / ((real)1 - alpha)) * bg_dot;
                    real dL_dG = op * dL_dalpha;
                    real gdx = Gv * dx, gdy = Gv * dy;
                    real dG_ddelx = -gdx * A - gdy * B;
                    real dG_ddely = -gdy * Cc - gdx * B;
                    add[4] = dL_dG * dG_ddelx * ddelx_dx;
                    add[5] = dL_dG * dG_ddely * ddely_dy;
                    add[6] = (real)-0.5 * gdx * dx * dL_dG;
                    add[7] = (real)-0.5 * gdx * dy * dL_dG;
                    add[8] = (real)-0.5 * gdy * dy * dL_dG;
                    add[9] = Gv * dL_dalpha;
                    real *dst[10] = {&dL_drgb[3 * id], &dL_drgb[3 * id + 1], &dL_drgb[3 * id + 2], &dL_ddepth[id],
                                     &dL_dmean2D[2 * id], &dL_dmean2D[2 * id + 1], &dL_dconic[3 * id],
                                     &dL_dconic[3 * id + 1], &dL_dconic[3 * id + 2], &dL_dopacity[id]};
                    for (int k = 0; k < 10; ++k) {
#ifdef _OPENMP
#pragma omp atomic
#endif
                        *dst[k] += add[k];
                    }
                }
            }
    }
}

/* ------------------------------------------------------------------------
 * A5. preprocess backward (upstream backward.cu computeCov2DCUDA +
 * preprocessCUDA + computeColorFromSH, and the fork's dL/dtau).
 * Inputs: the per-Gaussian gradients from A4.  Outputs (overwritten):
 *   dL_dmeans (G,3), dL_dcov6 (G,6), dL_dshs (G,M,3) or dL_dcolors (G,3),
 *   dL_dopac is A4's dL_dopacity unchanged (not touched here),
 *   dL_dtau (G,6) = (rho, theta) per Gaussian when want_tau (else untouched).
 * tau is a LEFT se(3) perturbation of the world->camera transform,
 * T_cw' = exp(tau) T_cw (src/misc/cam_utils.py:118-137 update_pose), campos
 * held fixed.
 * ---------------------------------------------------------------------- */
void gso_preprocess_bwd(const gso_params *p, const real *means, const real *cov6, const real *shs_or_rgb,
                        const int32_t *radii, const uint8_t *clamped, const real *dL_dmean2D,
                        const real *dL_dconic, const real *dL_drgb, const real *dL_ddepth, real *dL_dmeans,
                        real *dL_dcov6, real *dL_dshs, real *dL_dtau)
{
    const real *V = p->view, *P = p->proj, *Q = p->proj_raw;
    real fx = (real)p->W / ((real)2 * p->tanfovx);
    real fy = (real)p->H / ((real)2 * p->tanfovy);
    int n = (p->sh_degree + 1) * (p->sh_degree + 1);
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int i = 0; i < p->G; ++i) {
        for (int k = 0; k < 3; ++k) dL_dmeans[3 * i + k] = 0;
        for (int k = 0; k < 6; ++k) dL_dcov6[6 * i + k] = 0;
        if (p->M > 0) for (int k = 0; k < 3 * p->M; ++k) dL_dshs[(size_t)3 * p->M * i + k] = 0;
        else for (int k = 0; k < 3; ++k) dL_dshs[3 * i + k] = 0;
        if (dL_dtau) for (int k = 0; k < 6; ++k) dL_dtau[6 * i + k] = 0;
        if (radii[i] <= 0) continue;
        const real *m = means + 3 * i;
        gso_geom g;
        gso_geom_eval(p, m, cov6 + 6 * i, &g);
        real dmean[3] = {0, 0, 0};
        real dt_pose[3] = {0, 0, 0}; /* dL/dp_C for the pose path */
        real dtheta_R[3] = {0, 0, 0};

        /* ---- colour (SH) ---- */
        if (p->M > 0) {
            real ddx = m[0] - p->campos[0], ddy = m[1] - p->campos[1], ddz = m[2] - p->campos[2];
            real len = r_sqrt(ddx * ddx + ddy * ddy + ddz * ddz);
            real x = ddx / len, y = ddy / len, z = ddz / len;
            real b[25], bx[25], by[25], bz[25];
            sh_basis(p->sh_degree, x, y, z, b);
            sh_basis_grad(p->sh_degree, x, y, z, bx, by, bz);
            const real *sh = shs_or_rgb + (size_t)3 * p->M * i;
            real dLdx = 0, dLdy = 0, dLdz = 0;
            for (int c = 0; c < 3; ++c) {
                real gcol = clamped[3 * i + c] ? (real)0 : dL_drgb[3 * i + c];
                for (int k = 0; k < n; ++k) {
                    dL_dshs[(size_t)3 * p->M * i + 3 * k + c] = b[k] * gcol;
                    dLdx += bx[k] * sh[3 * k + c] * gcol;
                    dLdy += by[k] * sh[3 * k + c] * gcol;
                    dLdz += bz[k] * sh[3 * k + c] * gcol;
                }
            }
            /* through normalize(): d(dir)/d(v) = (I - dir dir^T)/len */
            real dot = x * dLdx + y * dLdy + z * dLdz;
            dmean[0] += (dLdx - x * dot) / len;
            dmean[1] += (dLdy - y * dot) / len;
            dmean[2] += (dLdz - z * dot) / len;
        } else {
            for (int c = 0; c < 3; ++c) dL_dshs[3 * i + c] = dL_drgb[3 * i + c];
        }

        /* ---- conic -> cov2D (a,b,c) ---- */
        real a = g.a, b = g.b, c = g.c;
        real denom = a * c - b * b;
        real k2 = (real)1 / (denom * denom + (real)0.0000001);
        real gA = dL_dconic[3 * i], gB = dL_dconic[3 * i + 1], gC = dL_dconic[3 * i + 2];
        real ga = k2 * (-c * c * gA + (real)2 * b * c * gB + (denom - a * c) * gC);
        real gc = k2 * (-a * a * gC + (real)2 * a * b * gB + (denom - a * c) * gA);
        real gb = k2 * (real)2 * (b * c * gA - (denom + (real)2 * b * b) * gB + a * b * gC);

        /* ---- cov2D -> cov3D (6) ---- */
        const real *M0 = g.M0, *M1 = g.M1;
        real *dc = dL_dcov6 + 6 * i;
        dc[0] = M0[0] * M0[0] * ga + M0[0] * M1[0] * gb + M1[0] * M1[0] * gc;
        dc[3] = M0[1] * M0[1] * ga + M0[1] * M1[1] * gb + M1[1] * M1[1] * gc;
        dc[5] = M0[2] * M0[2] * ga + M0[2] * M1[2] * gb + M1[2] * M1[2] * gc;
        dc[1] = (real)2 * M0[0] * M0[1] * ga + (M0[0] * M1[1] + M0[1] * M1[0]) * gb + (real)2 * M1[0] * M1[1] * gc;
        dc[2] = (real)2 * M0[0] * M0[2] * ga + (M0[0] * M1[2] + M0[2] * M1[0]) * gb + (real)2 * M1[0] * M1[2] * gc;
        dc[4] = (real)2 * M0[2] * M0[1] * ga + (M0[1] * M1[2] + M0[2] * M1[1]) * gb + (real)2 * M1[1] * M1[2] * gc;

        /* ---- cov2D -> M = J R ---- */
        const real *S = cov6 + 6 * i;
        real SM0[3] = {S[0] * M0[0] + S[1] * M0[1] + S[2] * M0[2], S[1] * M0[0] + S[3] * M0[1] + S[4] * M0[2],
                       S[2] * M0[0] + S[4] * M0[1] + S[5] * M0[2]};
        real SM1[3] = {S[0] * M1[0] + S[1] * M1[1] + S[2] * M1[2], S[1] * M1[0] + S[3] * M1[1] + S[4] * M1[2],
                       S[2] * M1[0] + S[4] * M1[1] + S[5] * M1[2]};
        real dM0[3], dM1[3];
        for (int j = 0; j < 3; ++j) {
            dM0[j] = (real)2 * ga * SM0[j] + gb * SM1[j];
            dM1[j] = gb * SM0[j] + (real)2 * gc * SM1[j];
        }
        /* R[k][j] = V[4*j+k];  dL/dJ_ik = sum_j dM_ij R[k][j] */
        real dJ00 = 0, dJ02 = 0, dJ11 = 0, dJ12 = 0;
        for (int j = 0; j < 3; ++j) {
            dJ00 += dM0[j] * V[4 * j + 0];
            dJ02 += dM0[j] * V[4 * j + 2];
            dJ11 += dM1[j] * V[4 * j + 1];
            dJ12 += dM1[j] * V[4 * j + 2];
        }
        real tz = (real)1 / g.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        real xmul = g.clampx ? (real)0 : (real)1, ymul = g.clampy ? (real)0 : (real)1;
        real dt_cov[3];
        dt_cov[0] = xmul * -fx * tz2 * dJ02;
        dt_cov[1] = ymul * -fy * tz2 * dJ12;
        dt_cov[2] = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + ((real)2 * fx * g.txc) * tz3 * dJ02 +
                    ((real)2 * fy * g.tyc) * tz3 * dJ12;
        /* dL/dmean = R^T dL/dt : mean_k gets sum_c V[4k+c] dt[c] */
        for (int k = 0; k < 3; ++k)
            dmean[k] += V[4 * k + 0] * dt_cov[0] + V[4 * k + 1] * dt_cov[1] + V[4 * k + 2] * dt_cov[2];

        /* ---- mean2D -> mean through the FULL projection (upstream form) ---- */
        real hx = m[0] * P[0] + m[1] * P[4] + m[2] * P[8] + P[12];
        real hy = m[0] * P[1] + m[1] * P[5] + m[2] * P[9] + P[13];
        real hw = m[0] * P[3] + m[1] * P[7] + m[2] * P[11] + P[15];
        real mw = (real)1 / (hw + (real)0.0000001);
        real mul1 = hx * mw * mw, mul2 = hy * mw * mw;
        real g2x = dL_dmean2D[2 * i], g2y = dL_dmean2D[2 * i + 1];
        for (int k = 0; k < 3; ++k)
            dmean[k] += (P[4 * k + 0] * mw - P[4 * k + 3] * mul1) * g2x + (P[4 * k + 1] * mw - P[4 * k + 3] * mul2) * g2y;

        /* ---- depth -> mean:  depth = t.z = sum_k m_k V[4k+2] + V[14] ---- */
        real gd = dL_ddepth[i];
        for (int k = 0; k < 3; ++k) dmean[k] += V[4 * k + 2] * gd;

        for (int k = 0; k < 3; ++k) dL_dmeans[3 * i + k] = dmean[k];

        /* ---- pose (tau) ---- */
        if (dL_dtau && p->want_tau) {
            /* mean2D through proj_raw applied to p_C: h = [t,1] Q */
            real t0 = g.t[0], t1 = g.t[1], t2 = g.t[2];
            real qx = t0 * Q[0] + t1 * Q[4] + t2 * Q[8] + Q[12];
            real qy = t0 * Q[1] + t1 * Q[5] + t2 * Q[9] + Q[13];
            real qw = t0 * Q[3] + t1 * Q[7] + t2 * Q[11] + Q[15];
            real w1 = (real)1 / (qw + (real)0.0000001);
            real m1 = qx * w1 * w1, m2 = qy * w1 * w1;
            for (int k = 0; k < 3; ++k)
                dt_pose[k] = (Q[4 * k + 0] * w1 - Q[4 * k + 3] * m1) * g2x + (Q[4 * k + 1] * w1 - Q[4 * k + 3] * m2) * g2y;
            dt_pose[2] += gd;
            for (int k = 0; k < 3; ++k) dt_pose[k] += dt_cov[k];
            /* rotation path of cov2D: dL/dR = J^T dM ; A = dL/dR R^T */
            real dR[3][3];
            for (int j = 0; j < 3; ++j) {
                dR[0][j] = g.J00 * dM0[j];
                dR[1][j] = g.J11 * dM1[j];
                dR[2][j] = g.J02 * dM0[j] + g.J12 * dM1[j];
            }
            real Am[3][3];
            for (int r = 0; r < 3; ++r)
                for (int l = 0; l < 3; ++l) {
                    real acc = 0;
                    for (int j = 0; j < 3; ++j) acc += dR[r][j] * V[4 * j + l]; /* R[l][j] = V[4j+l] */
                    Am[r][l] = acc;
                }
            dtheta_R[0] = Am[2][1] - Am[1][2];
            dtheta_R[1] = Am[0][2] - Am[2][0];
            dtheta_R[2] = Am[1][0] - Am[0][1];
            real *tau = dL_dtau + 6 * i;
            tau[0] = dt_pose[0]; tau[1] = dt_pose[1]; tau[2] = dt_pose[2];
            tau[3] = (t1 * dt_pose[2] - t2 * dt_pose[1]) + dtheta_R[0];
            tau[4] = (t2 * dt_pose[0] - t0 * dt_pose[2]) + dtheta_R[1];
            tau[5] = (t0 * dt_pose[1] - t1 * dt_pose[0]) + dtheta_R[2];
        }
    }
}
