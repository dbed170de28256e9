// gsr_common.h -- shared device helpers of the gfx950 rasterizer.
// Written for CDNA4 only: 64-wide wavefronts, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsr.h"

namespace gsr {

constexpr int TILE = GSR_TILE;
constexpr int TILE_PIX = TILE * TILE;   // 256 pixels = 4 wavefronts
constexpr int WAVE = 64;

// 48-byte per-(view,Gaussian) splat record; three aligned float4 so a lane gathers it with three dwordx4 loads.
// (Padding it to a 64-byte slot, so that the per-tile sort's random gather touches exactly one 64-byte line instead of
// 1.5 on average, was measured: that kernel did not move, the kernels that stream the records got 10-30 % slower.)
struct alignas(16) SplatRec {
    float x, y, depth;
    uint32_t rad_flags;     // bits 0..23 radius (0 => culled / not rasterized), bits 24..26 SH channel clamped at 0
    float A, B, C, opacity; // conic + opacity
    float r, g, b;
    uint32_t ext;           // half extents (pixels, rounded up) of the alpha >= 1/255 footprint: hx | hy << 16
};
constexpr int LDS_TILES_MAX = 4096;   // K1 / K3 keep per-tile counters of one view in LDS up to this many tiles (1024 x 1024 pixels); larger images bin with global atomics
static_assert(sizeof(SplatRec) == 48, "SplatRec must be 48 bytes");

// 48-byte list entry as the composite kernels park it in LDS (three float4 per entry): the splat record re-ordered for
// their broadcast reads, plus the Gaussian id and the quadrant mask.
struct alignas(16) QueueRec {
    float x, y, A, B;
    float C, opacity, depth; uint32_t id;
    float r, g, b; uint32_t quad;  // bit k: the footprint may touch 8x8 quadrant k of the tile
};
static_assert(sizeof(QueueRec) == 48, "QueueRec must be 48 bytes");

constexpr int GR_STRIDE = 12;   // floats per (view, Gaussian) gradient accumulator record (layout: gsr_backward.hip)

struct Ptrs {             // carved workspace
    SplatRec *records;
    uint32_t *tile_count, *tile_offset, *tile_cursor, *tile_order;
    unsigned long long *pairs, *pairs_alt;
    uint32_t *point_list;   // ids | quadrant mask << GSR_QUAD_SHIFT (the mask bits are the composite forward's)
    float *loss_partial;    // fused MSE: [V*T] tile partials, [V] view partials
    uint32_t *loss_ticket;  // fused MSE: [V] tiles arrived per view, [1] views arrived
    float *loss_diff;       // fused MSE: image - target, (V,3,H,W)
    float *final_T;
    uint32_t *n_contrib;
    float *grad_rec;
    int32_t *status;
};

}  // namespace gsr

// host-side stage timer (include/gsr.h GsrProfile)
struct GsrProfile {
    int max_calls;
    int next_fwd, next_bwd;
    uint32_t mask;   // stages that record events (gsr_profile_set_stages)
    hipEvent_t *ev;  // [max_calls][GSR_N_STAGES][2]
    hipEvent_t &at(int call, int stage, int which) { return ev[((size_t)call * GSR_N_STAGES + stage) * 2 + which]; }
};

namespace gsr {

// remembers the HIP error behind a GSR_ELAUNCH for gsr_last_error()
extern thread_local hipError_t g_last_hip_error;
inline int launch_status()
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return GSR_OK;
    g_last_hip_error = e;
    return GSR_ELAUNCH;
}
inline bool hip_ok(hipError_t e)
{
    if (e == hipSuccess) return true;
    g_last_hip_error = e;
    return false;
}

struct StageTimer {
    GsrProfile *p; int slot; hipStream_t s;
    // resume: second half of a two-phase forward (GSR_FLAG_PHASE_RENDER) -> keep writing into the slot the first half opened
    StageTimer(void *prof, bool fwd, hipStream_t stream, bool resume = false)
        : p(static_cast<GsrProfile *>(prof)), slot(-1), s(stream)
    {
        if (!p) return;
        int &n = fwd ? p->next_fwd : p->next_bwd;
        if (resume) { if (n > 0 && n <= p->max_calls) slot = n - 1; }
        else if (n < p->max_calls) slot = n++;
    }
    void begin(int stage) { if (slot >= 0 && ((p->mask >> stage) & 1u)) (void)hipEventRecord(p->at(slot, stage, 0), s); }
    void end(int stage) { if (slot >= 0 && ((p->mask >> stage) & 1u)) (void)hipEventRecord(p->at(slot, stage, 1), s); }
};

__host__ __device__ inline int tiles_x(int W) { return (W + TILE - 1) / TILE; }
__host__ __device__ inline int tiles_y(int H) { return (H + TILE - 1) / TILE; }

// ---- real SH tables (bands 0-3 = published 3DGS constants, band 4 = standard real SH) ----
__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f, SH_C2_2 = 0.31539156525252005f,
                           SH_C2_3 = -1.0925484305920792f, SH_C2_4 = 0.5462742152960396f;
__device__ constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f, SH_C3_2 = -0.4570457994644658f,
                           SH_C3_3 = 0.3731763325901154f, SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                           SH_C3_6 = -0.5900435899266435f;
__device__ constexpr float SH_C4_0 = 2.5033429417967046f, SH_C4_1 = -1.7701307697799304f, SH_C4_2 = 0.9461746957575601f,
                           SH_C4_3 = -0.6690465435572892f, SH_C4_4 = 0.10578554691520431f, SH_C4_5 = -0.6690465435572892f,
                           SH_C4_6 = 0.47308734787878004f, SH_C4_7 = -1.7701307697799304f, SH_C4_8 = 0.6258357354491761f;

// Everything that must agree bit-for-bit with the fp32 oracle is compiled with
// fp contraction OFF and written in the oracle's operation order.
#pragma clang fp contract(off)

__device__ inline void sh_basis(int deg, float x, float y, float z, float *b)
{
    b[0] = SH_C0;
    if (deg < 1) return;
    b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
    if (deg < 2) return;
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = SH_C2_0 * xy; b[5] = SH_C2_1 * yz; b[6] = SH_C2_2 * (2.0f * zz - xx - yy);
    b[7] = SH_C2_3 * xz; b[8] = SH_C2_4 * (xx - yy);
    if (deg < 3) return;
    b[9] = SH_C3_0 * y * (3.0f * xx - yy);
    b[10] = SH_C3_1 * xy * z;
    b[11] = SH_C3_2 * y * (4.0f * zz - xx - yy);
    b[12] = SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    b[13] = SH_C3_4 * x * (4.0f * zz - xx - yy);
    b[14] = SH_C3_5 * z * (xx - yy);
    b[15] = SH_C3_6 * x * (xx - 3.0f * yy);
    if (deg < 4) return;
    b[16] = SH_C4_0 * xy * (xx - yy);
    b[17] = SH_C4_1 * yz * (3.0f * xx - yy);
    b[18] = SH_C4_2 * xy * (7.0f * zz - 1.0f);
    b[19] = SH_C4_3 * yz * (7.0f * zz - 3.0f);
    b[20] = SH_C4_4 * (zz * (35.0f * zz - 30.0f) + 3.0f);
    b[21] = SH_C4_5 * xz * (7.0f * zz - 3.0f);
    b[22] = SH_C4_6 * (xx - yy) * (7.0f * zz - 1.0f);
    b[23] = SH_C4_7 * xz * (xx - 3.0f * yy);
    b[24] = SH_C4_8 * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
}

__device__ inline void sh_basis_grad(int deg, float x, float y, float z, float *dx, float *dy, float *dz)
{
    dx[0] = dy[0] = dz[0] = 0.f;
    if (deg < 1) return;
    dx[1] = 0.f; dy[1] = -SH_C1; dz[1] = 0.f;
    dx[2] = 0.f; dy[2] = 0.f; dz[2] = SH_C1;
    dx[3] = -SH_C1; dy[3] = 0.f; dz[3] = 0.f;
    if (deg < 2) return;
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    dx[4] = SH_C2_0 * y; dy[4] = SH_C2_0 * x; dz[4] = 0.f;
    dx[5] = 0.f; dy[5] = SH_C2_1 * z; dz[5] = SH_C2_1 * y;
    dx[6] = SH_C2_2 * (-2.0f * x); dy[6] = SH_C2_2 * (-2.0f * y); dz[6] = SH_C2_2 * (4.0f * z);
    dx[7] = SH_C2_3 * z; dy[7] = 0.f; dz[7] = SH_C2_3 * x;
    dx[8] = SH_C2_4 * (2.0f * x); dy[8] = SH_C2_4 * (-2.0f * y); dz[8] = 0.f;
    if (deg < 3) return;
    dx[9] = SH_C3_0 * (6.0f * xy); dy[9] = SH_C3_0 * (3.0f * xx - 3.0f * yy); dz[9] = 0.f;
    dx[10] = SH_C3_1 * yz; dy[10] = SH_C3_1 * xz; dz[10] = SH_C3_1 * xy;
    dx[11] = SH_C3_2 * (-2.0f * xy); dy[11] = SH_C3_2 * (4.0f * zz - xx - 3.0f * yy); dz[11] = SH_C3_2 * (8.0f * yz);
    dx[12] = SH_C3_3 * (-6.0f * xz); dy[12] = SH_C3_3 * (-6.0f * yz); dz[12] = SH_C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy);
    dx[13] = SH_C3_4 * (4.0f * zz - 3.0f * xx - yy); dy[13] = SH_C3_4 * (-2.0f * xy); dz[13] = SH_C3_4 * (8.0f * xz);
    dx[14] = SH_C3_5 * (2.0f * xz); dy[14] = SH_C3_5 * (-2.0f * yz); dz[14] = SH_C3_5 * (xx - yy);
    dx[15] = SH_C3_6 * (3.0f * xx - 3.0f * yy); dy[15] = SH_C3_6 * (-6.0f * xy); dz[15] = 0.f;
    if (deg < 4) return;
    dx[16] = SH_C4_0 * y * (3.0f * xx - yy); dy[16] = SH_C4_0 * x * (xx - 3.0f * yy); dz[16] = 0.f;
    dx[17] = SH_C4_1 * (6.0f * xy * z); dy[17] = SH_C4_1 * z * (3.0f * xx - 3.0f * yy); dz[17] = SH_C4_1 * y * (3.0f * xx - yy);
    dx[18] = SH_C4_2 * y * (7.0f * zz - 1.0f); dy[18] = SH_C4_2 * x * (7.0f * zz - 1.0f); dz[18] = SH_C4_2 * (14.0f * xy * z);
    dx[19] = 0.f; dy[19] = SH_C4_3 * z * (7.0f * zz - 3.0f); dz[19] = SH_C4_3 * y * (21.0f * zz - 3.0f);
    dx[20] = 0.f; dy[20] = 0.f; dz[20] = SH_C4_4 * (140.0f * zz * z - 60.0f * z);
    dx[21] = SH_C4_5 * z * (7.0f * zz - 3.0f); dy[21] = 0.f; dz[21] = SH_C4_5 * x * (21.0f * zz - 3.0f);
    dx[22] = SH_C4_6 * (2.0f * x) * (7.0f * zz - 1.0f); dy[22] = SH_C4_6 * (-2.0f * y) * (7.0f * zz - 1.0f); dz[22] = SH_C4_6 * (xx - yy) * (14.0f * z);
    dx[23] = SH_C4_7 * z * (3.0f * xx - 3.0f * yy); dy[23] = SH_C4_7 * (-6.0f * xy * z); dz[23] = SH_C4_7 * x * (xx - 3.0f * yy);
    dx[24] = SH_C4_8 * (4.0f * xx * x - 12.0f * x * yy); dy[24] = SH_C4_8 * (-12.0f * xx * y + 4.0f * yy * y); dz[24] = 0.f;
}

// Camera-space geometry of one Gaussian (mirrors gso_geom_eval of the oracle
// operation for operation).
struct Geom {
    float t[3];
    float txc, tyc;
    bool clampx, clampy;
    float J00, J02, J11, J12;
    float M0[3], M1[3];
    float a, b, c;
};

__device__ inline bool geom_eval(const float *V, float tanfovx, float tanfovy, int W, int H, const float *mean,
                                 const float *S, Geom &g)
{
    float px = mean[0], py = mean[1], pz = mean[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) g.t[c] = px * V[0 + c] + py * V[4 + c] + pz * V[8 + c] + V[12 + c];
    if (g.t[2] <= 0.2f) return false;
    float fx = (float)W / (2.0f * tanfovx);
    float fy = (float)H / (2.0f * tanfovy);
    float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    float tz = g.t[2];
    float txtz = g.t[0] / tz, tytz = g.t[1] / tz;
    float cx = fminf(limx, fmaxf(-limx, txtz));
    float cy = fminf(limy, fmaxf(-limy, tytz));
    g.clampx = (txtz < -limx || txtz > limx);
    g.clampy = (tytz < -limy || tytz > limy);
    g.txc = cx * tz;
    g.tyc = cy * tz;
    g.J00 = fx / tz;
    g.J02 = -(fx * g.txc) / (tz * tz);
    g.J11 = fy / tz;
    g.J12 = -(fy * g.tyc) / (tz * tz);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float R0 = V[4 * j + 0], R1 = V[4 * j + 1], R2 = V[4 * j + 2];
        g.M0[j] = g.J00 * R0 + g.J02 * R2;
        g.M1[j] = g.J11 * R1 + g.J12 * R2;
    }
    float v0[3], v1[3];
    v0[0] = S[0] * g.M0[0] + S[1] * g.M0[1] + S[2] * g.M0[2];
    v0[1] = S[1] * g.M0[0] + S[3] * g.M0[1] + S[4] * g.M0[2];
    v0[2] = S[2] * g.M0[0] + S[4] * g.M0[1] + S[5] * g.M0[2];
    v1[0] = S[0] * g.M1[0] + S[1] * g.M1[1] + S[2] * g.M1[2];
    v1[1] = S[1] * g.M1[0] + S[3] * g.M1[1] + S[4] * g.M1[2];
    v1[2] = S[2] * g.M1[0] + S[4] * g.M1[1] + S[5] * g.M1[2];
    g.a = (g.M0[0] * v0[0] + g.M0[1] * v0[1] + g.M0[2] * v0[2]) + 0.3f;
    g.b = g.M1[0] * v0[0] + g.M1[1] * v0[1] + g.M1[2] * v0[2];
    g.c = (g.M1[0] * v1[0] + g.M1[1] * v1[1] + g.M1[2] * v1[2]) + 0.3f;
    return true;
}

// covariance of Gaussian `sg` as xx,xy,xz,yy,yz,zz from the (.,6) or the row-major (.,3,3) layout
__device__ inline void load_cov(const float *__restrict__ cov, size_t sg, bool cov9, float *S)
{
    if (cov9) {
        const float *c = cov + 9 * sg;
        S[0] = c[0]; S[1] = c[1]; S[2] = c[2]; S[3] = c[4]; S[4] = c[5]; S[5] = c[8];
    } else {
        const float *c = cov + 6 * sg;
#pragma unroll
        for (int k = 0; k < 6; ++k) S[k] = c[k];
    }
}

// tile rectangle of a splat (upstream getRect); returns the covered tile count.
__device__ inline int tile_rect(float px, float py, int rad, int gx, int gy, int &minx, int &miny, int &maxx, int &maxy)
{
    minx = min(gx, max(0, (int)((px - (float)rad) / (float)TILE)));
    miny = min(gy, max(0, (int)((py - (float)rad) / (float)TILE)));
    maxx = min(gx, max(0, (int)((px + (float)rad + (float)(TILE - 1)) / (float)TILE)));
    maxy = min(gy, max(0, (int)((py + (float)rad + (float)(TILE - 1)) / (float)TILE)));
    return (maxx - minx) * (maxy - miny);
}

#pragma clang fp contract(fast)

// 4-bit mask of the 8x8 quadrants of the tile at (ox, oy) that the splat's alpha >= 1/255 footprint can touch
__device__ inline uint32_t quadrant_mask(const float4 q0, const float4 q1, uint32_t ext, int ox, int oy)
{
    uint32_t quad = 0;
    if (ext) {
        const float hx = (float)(ext & 0xffffu), hy = (float)(ext >> 16);
        const float fox = (float)ox - q0.x, foy = (float)oy - q0.y;      // tile origin relative to the splat centre
        // a quadrant can see the splat only if min over its pixel rectangle of d^T conic d <= 2 ln(255 opacity):
        // bounding box first, then the exact minimum of the convex quadratic over the rectangle (centre inside -> 0,
        // otherwise it sits on one of the four edges at the clamped 1-D minimiser).  Conservative by the 0.2 % margin.
        const float A = q1.x, B = q1.y, C = q1.z;
        const float lim = 2.0f * __logf(255.0f * q1.w) * 1.002f + 1e-3f;
#pragma unroll         // (rolled, the loop costs the composite kernels 45 MORE registers: measured)
        for (int k = 0; k < 4; ++k) {
            const float x0 = fox + (float)((k & 1) * 8), x1 = x0 + 7.f;
            const float y0 = foy + (float)((k >> 1) * 8), y1 = y0 + 7.f;
            if (x0 > hx || x1 < -hx || y0 > hy || y1 < -hy) continue;
            float best = 0.f;
            if (!(x0 <= 0.f && x1 >= 0.f && y0 <= 0.f && y1 >= 0.f)) {
                best = 3.0e38f;
                // (v_rcp_f32, 1 ulp: the minimiser only has to land within the 0.2 % margin; an IEEE divide is ~10 instructions)
                const float iC = -B * __builtin_amdgcn_rcpf(C), iA = -B * __builtin_amdgcn_rcpf(A);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float xe = e ? x1 : x0;
                    const float dy = fminf(fmaxf(iC * xe, y0), y1);
                    best = fminf(best, A * xe * xe + 2.f * B * xe * dy + C * dy * dy);
                    const float ye = e ? y1 : y0;
                    const float dx = fminf(fmaxf(iA * ye, x0), x1);
                    best = fminf(best, A * dx * dx + 2.f * B * dx * ye + C * ye * ye);
                }
            }
            if (best <= lim) quad |= 1u << k;
        }
    }
    return quad;
}

// The composite kernels evaluate G = exp(power) as ONE v_exp_f32 (= 2^x): the conic is multiplied by log2(e) once per staged entry, so that
// `power` comes out in base-2 units.  Both kernels stage through the functions below and evaluate the same expression, so the forward's and
// the backward's alpha agree bit for bit (the backward re-derives which pixels a splat was composited into from it).
// (the round-2..4 form, exp(power) = v_mul(log2 e) + v_exp_f32 on the unscaled conic, was the A/B of round 5: profiles/r05_k6_cleanup_ab.md)
#define CONIC_PRESCALE 1.4426950408889634f
__device__ inline float footprint_exp(float power) { return __builtin_amdgcn_exp2f(power); }
// One list entry of a composite batch: gather the splat record of Gaussian `id` (three dwordx4 loads) and park it as a
// QueueRec in the three float4 LDS slots at `dst`.  The forward also marks the quadrants of the tile at (ox, oy) the
// footprint can touch; the kernel leaves that mask in the top four bits of the entry's point_list word and the backward reads it back
// instead of recomputing it (the mask code is ~150 instructions and 15 registers the backward's 8-waves-per-SIMD budget does not have).
__device__ inline uint32_t stage_entry_fwd(const SplatRec *__restrict__ recs, uint32_t id, int ox, int oy, float4 *dst)
{
    const float4 *r = reinterpret_cast<const float4 *>(recs + id);
    const float4 q0 = r[0], q1 = r[1], q2 = r[2];
    const uint32_t quad = quadrant_mask(q0, q1, __float_as_uint(q2.w), ox, oy);
    dst[0] = make_float4(q0.x, q0.y, q1.x * CONIC_PRESCALE, q1.y * CONIC_PRESCALE);
    dst[1] = make_float4(q1.z * CONIC_PRESCALE, q1.w, q0.z, __uint_as_float(id));
    dst[2] = make_float4(q2.x, q2.y, q2.z, __uint_as_float(quad));
    return quad;
}
__device__ inline void stage_entry_bwd(const SplatRec *__restrict__ recs, uint32_t id, uint32_t quad, float4 *dst)
{
    const float4 *r = reinterpret_cast<const float4 *>(recs + id);
    const float4 q0 = r[0], q1 = r[1], q2 = r[2];
    dst[0] = make_float4(q0.x, q0.y, q1.x * CONIC_PRESCALE, q1.y * CONIC_PRESCALE);
    dst[1] = make_float4(q1.z * CONIC_PRESCALE, q1.w, q0.z, __uint_as_float(id));
    dst[2] = make_float4(q2.x, q2.y, q2.z, __uint_as_float(quad));
}

// Select by a SCALAR lane mask (one bit per lane, e.g. the AND of several compares' ballots): v_cndmask_b32 reads the mask straight from its SGPR
// pair.  In C++ the same select needs a per-lane bool, i.e. the mask expanded to a register and compared again.
__device__ inline float sel_f(unsigned long long m, float t, float f)
{
    float r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
    return r;
}
__device__ inline float sel0_f(unsigned long long m, float t)         // m ? t : 0
{
    float r;
    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(r) : "v"(t), "s"(m));
    return r;
}
__device__ inline uint32_t sel_u(unsigned long long m, uint32_t t, uint32_t f)
{
    uint32_t r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
    return r;
}

// ---- wave64 helpers ------------------------------------------------------
// Sum over the 64 lanes with DPP row operations (no LDS traffic); the total
// lands in lane 63.  gfx9-family DPP: quad_perm, row_ror, row_bcast15/31.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND = false>
__device__ inline float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, BOUND));
}

__device__ inline float wave_sum_to_lane63(float v)
{
    v += dpp_mov<0xb1, 0xf, 0xf, true>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4e, 0xf, 0xf, true>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x124, 0xf, 0xf, true>(v);  // row_ror:4
    v += dpp_mov<0x128, 0xf, 0xf, true>(v);  // row_ror:8  (every lane of a row now holds the row sum)
    v += dpp_mov<0x142, 0xa, 0xf, false>(v); // row_bcast:15 -> rows 1,3
    v += dpp_mov<0x143, 0xc, 0xf, false>(v); // row_bcast:31 -> rows 2,3
    return v;
}

// ---- wave-aggregated counter bump ---------------------------------------------------
// Every valid lane wants counters[t] += 1 (and, if SLOT, its unique old value).  Lanes of a
// wavefront handle neighbouring Gaussians, which mostly fall into the same few tiles: group the
// lanes by target with ballots and issue ONE atomic per distinct tile instead of one per lane.
template <bool SLOT>
__device__ inline uint32_t wave_agg_inc(uint32_t *__restrict__ counters, uint32_t t, bool valid)
{
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(valid);
    // pass 1 (registers only): group the lanes by target; every lane learns its group's leader, size and its rank
    int my_leader = lane;
    uint32_t cnt = 0, rank = 0;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t tl = (uint32_t)__builtin_amdgcn_readlane((int)t, leader);
        const bool mine = valid && t == tl;
        const unsigned long long m = __ballot(mine);
        if (mine) {
            my_leader = leader;
            cnt = (uint32_t)__popcll(m);
            rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        }
        todo &= ~m;
    }
    // pass 2: ONE atomic instruction, executed by all group leaders at once (one memory round trip per call instead
    // of one per distinct target), then every member fetches its leader's base
    uint32_t base = 0;
    if (valid && lane == my_leader) {
        if (SLOT) base = atomicAdd(counters + t, cnt);
        else atomicAdd(counters + t, cnt);
    }
    if (!SLOT) return 0;
    base = (uint32_t)__shfl((int)base, my_leader, 64);
    return base + rank;
}

// ---- wave reductions of the composite backward -------------------------------------
// Sums nine / ten per-lane values over the 64 lanes: level 1 pairs values with v_permlane32_swap so that each half-wave keeps one value of
// the pair, level 2 does the same across the 16-lane rows with v_permlane16_swap, level 3 finishes inside the rows with bank-packed DPP adds
// (rows_packed_sum).  A plain per-value DPP reduction costs 6 DPP adds per value.
// The swaps are the compiler builtins (round 5; inline asm before): the compiler places their wait states (s_nop, never v_nop: one v_nop
// holds the SIMD's VALU port for ~9.5 ns against 0.55 ns for an s_nop state, profiles/r02_issue_cost.md) and, unlike asm operands tied
// "+v", they do not force copies of the inputs.  Round 2 saw a hipcc build fold the builtin's result pair r[0] + r[1] into r[0] + r[0];
// tests/test_host_boundary.py::test_swap_reductions_add_both_results checks the generated ISA of THIS build (every swap's two registers
// feed one v_add_f32), and the gradient parity tests would not survive a wrong sum.
__device__ inline float row_allsum(float v)
{
    v += dpp_mov<0xb1, 0xf, 0xf, true>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4e, 0xf, 0xf, true>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x124, 0xf, 0xf, true>(v);  // row_ror:4
    v += dpp_mov<0x128, 0xf, 0xf, true>(v);  // row_ror:8
    return v;
}
// (round 5: the swaps are compiler builtins, not inline asm, so that the compiler schedules them and their wait states itself)
__device__ inline float swapsum32(float x, float y)     // lanes < 32: sum of x's two halves, lanes >= 32: of y's
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ inline float swapsum16(float x, float y)     // rows 0, 2: x's row pairs (0,1), (2,3); rows 1, 3: y's
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// ---- the last level: several row-level registers summed into ONE packed register -------------------------------------------------------
// After the two swap levels a register holds four values, one per 16-lane row, and each still has to be summed over its row.  One at a time
// that is four DPP adds per register (quad_perm x 2, row_ror:4, row_ror:8).  DPP instructions write only the banks (4-lane groups of a row)
// named in bank_mask, so the halving steps can PACK instead: after the row_ror:8 step a value needs 8 lanes (two registers share one), after the
// rotate-by-4 step 4 lanes (a third moves into a free bank), and the two quad_perm steps then serve every bank at once.
//   X -> bank 0, Y -> bank 2, U -> bank 3 (lanes 12..15: lane 15 feeds row_bcast, which is how the depth-free kernel's ninth value, ONE value
//   over all 64 lanes, gets its cross-row sum).  Rotation: row_ror:12 (lane i reads lane i - 12 = i + 4 of its row).
// Wait states: a DPP read of a register written by the previous VALU instruction needs two (s_nop 1); inline asm is not scheduled.
__device__ inline float rows_packed_sum(float x, float y, float u, const bool u_whole_wave)
{
    float z, u8;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %1, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_ror:12 row_mask:0xf bank_mask:0x8\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                 : "=&v"(z), "=&v"(u8) : "v"(x), "v"(y), "v"(u));
    if (u_whole_wave)     // lanes 12..15 of every row hold u's row total: rows 1, 3 += row 0, 2 ; rows 2, 3 += row 1 -> lane 63: the wave total
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0x8\n\t"
                     "s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0x8" : "+v"(z));
    return z;
}
// Ten values -> one register: lanes 16r (+0..3) hold a0 a1 a5 a6 (r = 0..3), lanes 16r + 8 a2 a3 a7 a8, lanes 16r + 12 a4 - a9 -.
// 8 swaps + 8 adds + 7 DPP adds (round 2-4: 8 + 8 + 12, and two selects to pick among three registers).
__device__ inline float wave_reduce10(const float *a)
{
    // level 1: halves.  b_i: lanes < 32 hold a_i, lanes >= 32 hold a_{i+5}
    const float b0 = swapsum32(a[0], a[5]), b1 = swapsum32(a[1], a[6]), b2 = swapsum32(a[2], a[7]), b3 = swapsum32(a[3], a[8]),
                b4 = swapsum32(a[4], a[9]);
    // level 2: rows.  (b0,b1) -> a0 a1 a5 a6 ; (b2,b3) -> a2 a3 a7 a8 ; (b4,0) -> a4 - a9 -
    return rows_packed_sum(swapsum16(b0, b1), swapsum16(b2, b3), swapsum16(b4, 0.f), false);
}
__device__ inline int reduce10_slot(int lane)      // which of the ten values a lane publishes, else -1
{
    const int r = lane >> 4, s = lane & 15;
    if (s == 0) return (r < 2) ? r : r + 3;        // 0,1,5,6
    if (s == 8) return (r < 2) ? r + 2 : r + 5;    // 2,3,7,8
    if (s == 12) return (r == 0) ? 4 : (r == 2 ? 9 : -1);
    return -1;
}
// Nine values (the depth-free instantiation of the composite backward): eight go through the two swap levels, the ninth through DPP steps only
// (it rides in bank 3 of the packed register and is summed across the rows by the two row_bcast steps): lanes 16r hold v0 v1 v4 v5,
// lanes 16r + 8 v2 v3 v6 v7, lane 63 v8.  6 swaps + 6 adds + 9 DPP adds.
__device__ inline float wave_reduce9(const float *v)
{
    const float b0 = swapsum32(v[0], v[4]), b1 = swapsum32(v[1], v[5]), b2 = swapsum32(v[2], v[6]), b3 = swapsum32(v[3], v[7]);
    return rows_packed_sum(swapsum16(b0, b1), swapsum16(b2, b3), v[8], true);
}
__device__ inline int reduce9_slot(int lane)       // index into wave_reduce9's nine inputs, else -1
{
    const int r = lane >> 4, s = lane & 15;
    if (s == 0) return (r < 2) ? r : r + 2;        // 0,1,4,5
    if (s == 8) return (r < 2) ? r + 2 : r + 4;    // 2,3,6,7
    return lane == 63 ? 8 : -1;
}

}  // namespace gsr
