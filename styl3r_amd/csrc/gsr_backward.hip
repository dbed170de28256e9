// gsr_backward.hip -- backward pipeline of the gfx950 rasterizer.
//
//   K6 composite_bwd    same tiling as the forward (one wavefront per tile, 4 pixels per
//                       lane); the tile's sorted id list is walked back-to-front from the
//                       tile's deepest contributor, 64 entries at a time: each lane gathers ONE
//                       entry's 48-byte record into a wave-private LDS slot (the quadrant mask
//                       the forward left in the top bits of the list word rides along), then the
//                       wave evaluates the batch from LDS broadcasts as straight-line code under
//                       scalar lane masks; each splat's nine (ten with a depth gradient) partial
//                       gradients are summed across the 64 lanes in registers (permlane swaps +
//                       bank-packed DPP row sums, no LDS) and leave the wave as ONE atomic
//                       instruction per (tile, splat), one lane per component.  Optional
//                       prologue: dL/dimage of the fused LossMse from the forward's image.  (upstream R7)
//   K7 preprocess_bwd   per Gaussian, loops over the scene's views and sums their
//                       contributions in registers (no atomics, deterministic): conic ->
//                       cov2D -> cov3D / mean, projection, depth, SH, pose (tau).   (R8)
//
// (The row-packed, systolic re-decomposition of K6 measured in round 5 -- DESIGN.md R5.1, 1.5 x slower than the tile kernel -- lives in
//  tools/probes/k6_rows_experiment.patch, not in the product library.)
#include <type_traits>

#include "gsr_common.h"

namespace gsr {

// grad_rec layout, 12 floats per (view, Gaussian)
enum { GR_RGB = 0, GR_DEPTH = 3, GR_MX = 4, GR_MY = 5, GR_CA = 6, GR_CB = 7, GR_CC = 8, GR_OP = 9 };   // GR_STRIDE: gsr_common.h

// ------------------------------------------------------------------ K6
// One wavefront per tile, 4 pixels per lane (same pixel <-> lane map as the forward).
// A lane first adds the partial gradients of its own pixels in registers, then ONE
// ten-value wave reduction per splat (wave_reduce10) and ten lanes issue the tile's single
// atomic per component.  No LDS atomics, no workgroup barriers.
// DEPTH = false: no dL/ddepth was passed (Styl3R trains on colour only): the depth terms drop out of the evaluation
// MSE: dL/dimage += 2 weight / n * upstream gradient * (image - target), LossMse's backward (loss_mse.py:22-31), formed here from the difference
// the composite forward left in the workspace (gsr_forward_fused with a target); dL_dimage may then be nullptr (nothing else consumed the image).
template <bool DEPTH>
__global__ void __launch_bounds__(64) k_composite_bwd(GsrDims d, const GsrView *__restrict__ views, Ptrs ws,
                                                     const float *__restrict__ dL_dimage,
                                                     const float *__restrict__ dL_ddepth,
                                                     bool mse, float mse_weight, const float *__restrict__ mse_grad_loss)
{
    if (ws.status[GSR_ST_OVERFLOW]) return;
    __shared__ float4 s_q[64 * 3];

    const int gx = tiles_x(d.W), T = gx * tiles_y(d.H);
    const uint32_t tv = ws.tile_order[blockIdx.y * gridDim.x + blockIdx.x];   // longest lists are launched first
    const int tile = (int)(tv % (uint32_t)T), v = (int)(tv / (uint32_t)T);
    const int lane = threadIdx.x;
    const int ox = (tile % gx) * TILE + (lane & 7), oy = (tile / gx) * TILE + (lane >> 3);
    const size_t P = (size_t)d.H * d.W;

    const size_t t = (size_t)v * T + tile;
    const uint32_t start = ws.tile_offset[t], end = ws.tile_offset[t + 1];
    if (start == end) return;
    const uint32_t *__restrict__ plist = ws.point_list + start;
    const SplatRec *__restrict__ recs = ws.records + (size_t)v * d.G;
    float *grad = ws.grad_rec + (size_t)v * d.G * GR_STRIDE;
    const GsrView &vw = views[v];

    // Per pixel the colour/depth recurrences of upstream (accum_rec[ch], last_color[ch]) only ever enter through their dot
    // product with the pixel's incoming gradient, and both are linear, so ONE scalar on u = rgb.g + depth*gd replaces four.
    // It is kept in ABSOLUTE form: S = T_final (bg.g) + sum over the splats behind the current one of w_i u_i.  With
    // T_j accum_rec_j = (sum_{i>j} w_i u_i) / (1 - alpha_j), upstream's
    //     dL/dalpha_j = (u_j - accum_rec_j) T_j - T_final (bg.g) / (1 - alpha_j)        becomes      T_j u_j - S / (1 - alpha_j)
    // (identical up to rounding): one live register per pixel instead of four (accum_rec, last_alpha, last_u, the background
    // term) and 3 instructions instead of 8 per evaluation -- the 12 VGPRs this frees are a sixth wave per SIMD.
    // The pixel coordinates are the lane's base (fx0, fy0) plus the quadrant's constant offset, not eight more registers.
    const float fx0 = (float)ox, fy0 = (float)oy;
    float Tr[4], g0[4], g1[4], g2[4], gd[4], S[4];
    uint32_t last[4];
    uint32_t mx = 0;
    // the coefficient of gsr_mse_backward, same expression: 2 weight / n * upstream gradient
    const float mse_c = mse ? 2.0f * mse_weight / (float)((size_t)d.B * d.Vt * 3 * P) * (mse_grad_loss ? mse_grad_loss[0] : 1.f) : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int px = ox + (k & 1) * 8, py = oy + (k >> 1) * 8;
        const bool inside = px < d.W && py < d.H;
        const size_t pix = (size_t)py * d.W + px;
        const float Tf = inside ? ws.final_T[v * P + pix] : 0.f;
        last[k] = inside ? ws.n_contrib[v * P + pix] : 0u;
        g0[k] = (inside && dL_dimage) ? dL_dimage[(v * 3 + 0) * P + pix] : 0.f;
        g1[k] = (inside && dL_dimage) ? dL_dimage[(v * 3 + 1) * P + pix] : 0.f;
        g2[k] = (inside && dL_dimage) ? dL_dimage[(v * 3 + 2) * P + pix] : 0.f;
        if (mse && inside) {       // (mse: wave-uniform)
            const float e0 = mse_c * ws.loss_diff[(v * 3 + 0) * P + pix], e1 = mse_c * ws.loss_diff[(v * 3 + 1) * P + pix],
                        e2 = mse_c * ws.loss_diff[(v * 3 + 2) * P + pix];
            if (dL_dimage) { g0[k] += e0; g1[k] += e1; g2[k] += e2; }
            else { g0[k] = e0; g1[k] = e1; g2[k] = e2; }
        }
        gd[k] = (DEPTH && inside) ? dL_ddepth[v * P + pix] : 0.f;
        S[k] = Tf * (vw.bg[0] * g0[k] + vw.bg[1] * g1[k] + vw.bg[2] * g2[k]);
        Tr[k] = Tf;
        mx = max(mx, last[k]);
    }
    const float ddelx_dx = 0.5f * (float)d.W, ddely_dy = 0.5f * (float)d.H;
    // deepest contributor of the tile: nothing behind it received any light
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    const int max_last = __builtin_amdgcn_readfirstlane((int)mx);   // wave-uniform (all-lanes maximum): batch / entry counters and the loop controls stay scalar
    // DEPTH: ten sums, reduce10; depth-free: the nine live sums go through wave_reduce9 (GR_DEPTH's column of grad_rec keeps its zero)
    constexpr bool NINE = !DEPTH;
    int slot;                                   // the grad_rec column this lane publishes, -1: none
    if (NINE) { const int i9 = reduce9_slot(lane); slot = i9 < 0 ? -1 : (i9 < 3 ? i9 : i9 + 1); }      // v = s[0..2], s[4..9]
    else slot = reduce10_slot(lane);

    // entries [0, max_last) of the sorted list, in batches from the back; slot l of a batch = entry hi-1-l.  Each lane
    // gathers one entry's record and parks it in LDS at the start of the batch (stage_entry): holding the NEXT batch in 12
    // registers across the whole evaluation (round 1) hid one ~1 us load per 64 entries (< 1 % of a batch's time) and cost
    // the kernel two waves per SIMD of occupancy.
    for (int hi = max_last; hi > 0; hi -= 64) {
        const int cnt = min(64, hi);
        // (single-wave workgroup: its LDS instructions execute in order, so a compiler barrier orders the staging stores against the
        //  previous batch's reads -- a __syncthreads would also wait, s_waitcnt vmcnt(0), for every gradient atomic still in flight)
        asm volatile("" ::: "memory");
        uint32_t bm = 0;
        if (lane < cnt) {
            const uint32_t word = plist[hi - 1 - lane];  // id | the forward's quadrant mask (entries it never staged or that miss the tile: 0)
            bm = word >> GSR_QUAD_SHIFT;
            stage_entry_bwd(recs, word & GSR_ID_MASK, bm, s_q + lane * 3);
        }
        // the slots whose footprint touches the tile at all, as a scalar bit mask: the entry loop visits only those (no per-entry LDS read
        // + readfirstlane + branch just to find out that there is nothing to evaluate, and the entry's record is read in one go)
        unsigned long long todo = __builtin_amdgcn_ballot_w64(bm != 0u);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // (reading entry j + 1 ahead of time was measured: +12 VGPRs drop the kernel from 5 to 4 waves per SIMD, -6 %)
        for (; todo; todo &= todo - 1ull) {
            const int j = __builtin_ctzll(todo);
            const uint32_t entry = (uint32_t)(hi - 1 - j);  // 0-based position in the list
            const float4 a = s_q[j * 3 + 0];                // x, y, A, B
            const float4 b = s_q[j * 3 + 1];                // C, opacity, depth, id
            const float4 c = s_q[j * 3 + 2];                // r, g, b, quad
            const uint32_t quad = __builtin_amdgcn_readfirstlane(__float_as_uint(c.w));
            float s[10];
            // "Defined" here by an empty asm: on the paths the compiler cannot rule out (an accumulating quadrant without an assigning one)
            // the sums would otherwise be the PREVIOUS entry's, i.e. live across the reduction -- ten v_mov copies in front of its swaps
#pragma unroll
            for (int i = 0; i < 10; ++i) asm volatile("" : "=v"(s[i]));
            if (!DEPTH && !NINE) s[GR_DEPTH] = 0.f;
            unsigned long long anym = 0ull;   // lanes that composited this entry, over its quadrants (a scalar OR of the compare masks: no VALU)
            // one evaluation of quadrant k.  FIRST (the first quadrant of this entry, wave-uniform) ASSIGNS the ten partial
            // sums, later ones accumulate: no per-entry zeroing of ten registers (a tenth of the kernel's VALU slots when
            // 1.4 quadrants are evaluated per entry)
            auto eval = [&](const bool FIRST, const int k) {      // FIRST: wave-uniform
                // Straight-line, predicated by `valid`: an invalid (pixel, splat) pair runs with alpha = G = 0, which
                // leaves T and S unchanged (w = 0) and contributes exactly 0 to every sum.
                const float dx = (a.x - fx0) - (float)((k & 1) * 8), dy = (a.y - fy0) - (float)((k >> 1) * 8);
                const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
                // (no clamp of `power`: where it is positive the splat is invalid whatever G is -- +inf, or NaN * 0 behind it, never reaches a sum:
                //  alpha and Gv are selected to 0 by `valid`, and v_min_f32 returns the number of (0.99, NaN))
                const float Graw = footprint_exp(power);
                const float araw = fminf(0.99f, b.y * Graw);
                // validity as a SCALAR lane mask: the AND of the compares' own ballots (the ballot of a per-lane AND is a v_cndmask + v_cmp round trip
                // through a register), the two selects read it from its SGPR pair.
                // (Measured and dropped: a chain of VALU selects instead of the scalar AND in front of the two selects, 0.887 vs 0.871 ms)
                const unsigned long long vm = __builtin_amdgcn_ballot_w64(entry < last[k]) & __builtin_amdgcn_ballot_w64(power <= 0.f) &
                                              __builtin_amdgcn_ballot_w64(araw >= (1.f / 255.f));
                anym |= vm;
                const float alpha = sel0_f(vm, araw);
                const float Gv = sel0_f(vm, Graw);
                const float inv = __builtin_amdgcn_rcpf(1.f - alpha);   // v_rcp_f32 (1 ulp) for both 1/(1-alpha) uses
                Tr[k] *= inv;
                const float w = alpha * Tr[k];
                float u = c.x * g0[k] + c.y * g1[k] + c.z * g2[k];
                if (DEPTH) u += b.z * gd[k];
                const float dL_dalpha = Tr[k] * u - S[k] * inv;
                S[k] += w * u;
                // dL/dmean2D = sum dL_dG * (-G dx A - G dy B, -G dy C - G dx B) * (W/2, H/2) is linear in the two sums
                // hx = sum(-dL_dG/2 * G dx), hy = sum(-dL_dG/2 * G dy), which the conic gradients need anyway: accumulate
                // those (two adds per evaluation instead of six multiply-adds) and apply A, B, C once per Gaussian in K7
                const float t = Gv * dL_dalpha;            // dL/dopacity term; dL_dG * G = opacity * t
                // (the factor -opacity / 2 of dL_dG G = opacity t is per Gaussian and view: K7 applies it to the five summed moments)
                const float hx = t * dx, hy = t * dy;
                if (FIRST) {
                    s[GR_RGB + 0] = w * g0[k]; s[GR_RGB + 1] = w * g1[k]; s[GR_RGB + 2] = w * g2[k];
                    if (DEPTH) s[GR_DEPTH] = w * gd[k];
                    s[GR_OP] = t; s[GR_MX] = hx; s[GR_MY] = hy;
                    s[GR_CA] = hx * dx; s[GR_CB] = hx * dy; s[GR_CC] = hy * dy;
                } else {
                    s[GR_RGB + 0] += w * g0[k]; s[GR_RGB + 1] += w * g1[k]; s[GR_RGB + 2] += w * g2[k];
                    if (DEPTH) s[GR_DEPTH] += w * gd[k];
                    s[GR_OP] += t; s[GR_MX] += hx; s[GR_MY] += hy;
                    s[GR_CA] += hx * dx; s[GR_CB] += hx * dy; s[GR_CC] += hy * dy;
                }
            };
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!(quad & (1u << k))) continue;  // scalar branches: the only control flow of the evaluation
                eval((quad & ((1u << k) - 1u)) == 0u, k);
            }
            if (anym == 0ull) continue;  // wave-uniform
            float val;             // the packed register of the reduction: every publishing lane finds its total in its own lane
            if (NINE) {
                const float v9[9] = {s[0], s[1], s[2], s[4], s[5], s[6], s[7], s[8], s[9]};
                val = wave_reduce9(v9);
            } else {
                val = wave_reduce10(s);
            }
            if (slot >= 0) {
                // uniform base + 32-bit byte offset (the saddr form of the atomic instead of a 64-bit multiply-add per lane)
                static_assert(GR_STRIDE * 4 == 48, "48-byte gradient records: id * 48 = (id << 5) + (id << 4)");
                uint32_t id16 = __float_as_uint(b.w) << 4;
                asm("" : "+v"(id16));          // (keeps the two v_lshl_add_u32: recombined to id * 48 it becomes a quarter-rate v_mad_u64_u32)
                const uint32_t off = (id16 << 1) + ((uint32_t)slot * 4u + id16);
                if (val != 0.f) atomicAdd(reinterpret_cast<float *>(reinterpret_cast<char *>(grad) + off), val);
            }
        }
    }
}

// ------------------------------------------------------------------ K7
__device__ inline float block_sum_256(float v, float *s_tmp, int tid)
{
    v = wave_sum_to_lane63(v);
    __syncthreads();
    if ((tid & 63) == 63) s_tmp[tid >> 6] = v;
    __syncthreads();
    return s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
}

#pragma clang fp contract(off)
// DEG: active SH degree 0..4, or -1 for precomputed colours.  A template parameter so that the SH tables and the
// per-coefficient accumulators are fully unrolled register arrays (degree 0 carries none of the degree-4 baggage).
template <int DEG>
__global__ void __launch_bounds__(256) k_preprocess_bwd(GsrDims d, const GsrView *__restrict__ views,
                                                        const float *__restrict__ means, const float *__restrict__ cov6,
                                                        const float *__restrict__ shs, Ptrs ws,
                                                        float *__restrict__ dL_dmeans, float *__restrict__ dL_dcov6,
                                                        float *__restrict__ dL_dopac, float *__restrict__ dL_dshs,
                                                        float *__restrict__ dL_dmeans2D, float *__restrict__ dL_dtau)
{
    __shared__ float s_tmp[4];
    const int tid = threadIdx.x;
    const int g = blockIdx.x * blockDim.x + tid;
    const int b = blockIdx.y;
    const bool valid = g < d.G;
    const size_t sg = (size_t)b * d.G + (valid ? g : 0);
    const float m0[3] = {means[3 * sg], means[3 * sg + 1], means[3 * sg + 2]};
    float S0[6];
    const bool cov9 = (d.flags & GSR_FLAG_COV9) != 0;
    load_cov(cov6, sg, cov9, S0);
    constexpr int NC = DEG < 0 ? 1 : (DEG + 1) * (DEG + 1);
    const int ncol = d.M > 0 ? 3 * d.M : 3;
    float *dsh = dL_dshs + sg * (size_t)ncol;
    float dsh_acc[3 * NC];            // summed over the scene's views in registers, stored once
#pragma unroll
    for (int k = 0; k < 3 * NC; ++k) dsh_acc[k] = 0.f;

    float dmean[3] = {0.f, 0.f, 0.f}, dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dop = 0.f;

    for (int j = 0; j < d.Vt; ++j) {
        const int v = b * d.Vt + j;
        const GsrView &vw = views[v];
        const float *V = vw.viewmatrix, *Pm = vw.projmatrix, *Q = vw.projmatrix_raw;
        const size_t vg = (size_t)v * d.G + (valid ? g : 0);
        float tau[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float g2x = 0.f, g2y = 0.f;
        // the record and the summed gradient record of this (view, Gaussian): five independent 16-byte loads issued together, in front of
        // the visibility test.  The kernel is LATENCY-bound (round 5: with contraction on and v_rcp_f32 for its reciprocals, -13 % instructions,
        // it did not move); before, the gradient record was fetched inside `if (vis)`, one memory round trip behind the record: 0.067 -> 0.052 ms
        // at the headline, 0.258 -> 0.215 at 262 144 Gaussians.  (Fetching one view AHEAD on top of this: no further gain, +12 registers.)
        const float4 q0 = reinterpret_cast<const float4 *>(ws.records + vg)[0];
        const float4 q1 = reinterpret_cast<const float4 *>(ws.records + vg)[1];      // A, B, C, opacity
        float gr[GR_STRIDE];
        {
            const float4 *g4 = reinterpret_cast<const float4 *>(ws.grad_rec + vg * GR_STRIDE);
            const float4 ga = g4[0], gb = g4[1], gc = g4[2];
            gr[0] = ga.x; gr[1] = ga.y; gr[2] = ga.z; gr[3] = ga.w; gr[4] = gb.x; gr[5] = gb.y; gr[6] = gb.z; gr[7] = gb.w;
            gr[8] = gc.x; gr[9] = gc.y; gr[10] = gc.z; gr[11] = gc.w;
        }
        const uint32_t rad_flags = __float_as_uint(q0.w);
        const bool vis = valid && (rad_flags & 0xffffffu) != 0;
        if (vis) {
            const uint32_t aux = rad_flags >> 24;
            const float s = vw.scale, s2 = s * s;
            const float m[3] = {m0[0] * s, m0[1] * s, m0[2] * s};
            float S[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) S[k] = S0[k] * s2;
            Geom ge;
            geom_eval(V, vw.tanfovx, vw.tanfovy, d.W, d.H, m, S, ge);
            const float fx = (float)d.W / (2.0f * vw.tanfovx);
            const float fy = (float)d.H / (2.0f * vw.tanfovy);
            float dm[3] = {0.f, 0.f, 0.f};

            // ---- colour ----
            if (DEG >= 0) {
                float ddx = m[0] - vw.campos[0], ddy = m[1] - vw.campos[1], ddz = m[2] - vw.campos[2];
                float len = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
                float x = ddx / len, y = ddy / len, z = ddz / len;
                float bs[NC];
                sh_basis(DEG < 0 ? 0 : DEG, x, y, z, bs);
                float gcol[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) gcol[c] = ((aux >> c) & 1u) ? 0.f : gr[GR_RGB + c];
#pragma unroll
                for (int k = 0; k < NC; ++k)
#pragma unroll
                    for (int c = 0; c < 3; ++c) dsh_acc[3 * k + c] += bs[k] * gcol[c];
                if (DEG > 0) {
                    float bx[NC], by[NC], bz[NC];
                    sh_basis_grad(DEG < 0 ? 0 : DEG, x, y, z, bx, by, bz);
                    const float *sh = shs + sg * 3 * (size_t)d.M;
                    float dLdx = 0.f, dLdy = 0.f, dLdz = 0.f;
#pragma unroll
                    for (int k = 1; k < NC; ++k) {
                        const float t = sh[3 * k] * gcol[0] + sh[3 * k + 1] * gcol[1] + sh[3 * k + 2] * gcol[2];
                        dLdx += bx[k] * t; dLdy += by[k] * t; dLdz += bz[k] * t;
                    }
                    float dot = x * dLdx + y * dLdy + z * dLdz;
                    dm[0] += (dLdx - x * dot) / len;
                    dm[1] += (dLdy - y * dot) / len;
                    dm[2] += (dLdz - z * dot) / len;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) dsh_acc[c] += gr[GR_RGB + c];
            }

            // ---- conic -> cov2D ----
            const float a = ge.a, bb = ge.b, c = ge.c;
            const float denom = a * c - bb * bb;
            const float k2 = 1.0f / (denom * denom + 0.0000001f);
            // K6 leaves the moments of t = dL/dalpha G over the splat's pixels: sum t dx, sum t dy, sum t dx^2, sum t dx dy, sum t dy^2;
            // dL_dG G = opacity t, and the conic / mean terms carry -1/2 of it: one factor per (view, Gaussian), applied here
            const float kop = -0.5f * q1.w;
            const float gA = kop * gr[GR_CA], gB = kop * gr[GR_CB], gC = kop * gr[GR_CC];
            const float ga = k2 * (-c * c * gA + 2.0f * bb * c * gB + (denom - a * c) * gC);
            const float gc = k2 * (-a * a * gC + 2.0f * a * bb * gB + (denom - a * c) * gA);
            const float gb = k2 * 2.0f * (bb * c * gA - (denom + 2.0f * bb * bb) * gB + a * bb * gC);

            // ---- cov2D -> cov3D ----
            const float *M0 = ge.M0, *M1 = ge.M1;
            float dc[6];
            dc[0] = M0[0] * M0[0] * ga + M0[0] * M1[0] * gb + M1[0] * M1[0] * gc;
            dc[3] = M0[1] * M0[1] * ga + M0[1] * M1[1] * gb + M1[1] * M1[1] * gc;
            dc[5] = M0[2] * M0[2] * ga + M0[2] * M1[2] * gb + M1[2] * M1[2] * gc;
            dc[1] = 2.0f * M0[0] * M0[1] * ga + (M0[0] * M1[1] + M0[1] * M1[0]) * gb + 2.0f * M1[0] * M1[1] * gc;
            dc[2] = 2.0f * M0[0] * M0[2] * ga + (M0[0] * M1[2] + M0[2] * M1[0]) * gb + 2.0f * M1[0] * M1[2] * gc;
            dc[4] = 2.0f * M0[2] * M0[1] * ga + (M0[1] * M1[2] + M0[2] * M1[1]) * gb + 2.0f * M1[1] * M1[2] * gc;
#pragma unroll
            for (int k = 0; k < 6; ++k) dcov[k] += dc[k] * s2;

            // ---- cov2D -> M = J R -> t ----
            float SM0[3] = {S[0] * M0[0] + S[1] * M0[1] + S[2] * M0[2], S[1] * M0[0] + S[3] * M0[1] + S[4] * M0[2],
                            S[2] * M0[0] + S[4] * M0[1] + S[5] * M0[2]};
            float SM1[3] = {S[0] * M1[0] + S[1] * M1[1] + S[2] * M1[2], S[1] * M1[0] + S[3] * M1[1] + S[4] * M1[2],
                            S[2] * M1[0] + S[4] * M1[1] + S[5] * M1[2]};
            float dM0[3], dM1[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                dM0[k] = 2.0f * ga * SM0[k] + gb * SM1[k];
                dM1[k] = gb * SM0[k] + 2.0f * gc * SM1[k];
            }
            float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                dJ00 += dM0[k] * V[4 * k + 0];
                dJ02 += dM0[k] * V[4 * k + 2];
                dJ11 += dM1[k] * V[4 * k + 1];
                dJ12 += dM1[k] * V[4 * k + 2];
            }
            const float tz = 1.0f / ge.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
            const float xmul = ge.clampx ? 0.f : 1.f, ymul = ge.clampy ? 0.f : 1.f;
            float dt_cov[3];
            dt_cov[0] = xmul * -fx * tz2 * dJ02;
            dt_cov[1] = ymul * -fy * tz2 * dJ12;
            dt_cov[2] = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.0f * fx * ge.txc) * tz3 * dJ02 +
                        (2.0f * fy * ge.tyc) * tz3 * dJ12;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                dm[k] += V[4 * k + 0] * dt_cov[0] + V[4 * k + 1] * dt_cov[1] + V[4 * k + 2] * dt_cov[2];

            // ---- mean2D -> mean (full projection) ----
            const float hx = m[0] * Pm[0] + m[1] * Pm[4] + m[2] * Pm[8] + Pm[12];
            const float hy = m[0] * Pm[1] + m[1] * Pm[5] + m[2] * Pm[9] + Pm[13];
            const float hw = m[0] * Pm[3] + m[1] * Pm[7] + m[2] * Pm[11] + Pm[15];
            const float mw = 1.0f / (hw + 0.0000001f);
            const float mul1 = hx * mw * mw, mul2 = hy * mw * mw;
            {   // K6 accumulated hx = sum(-dL_dG/2 G dx), hy likewise: mean2D gradient = (A hx + B hy) W, (C hy + B hx) H
                const float sx = kop * gr[GR_MX], sy = kop * gr[GR_MY];
                g2x = (q1.x * sx + q1.y * sy) * (float)d.W;
                g2y = (q1.z * sy + q1.y * sx) * (float)d.H;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k)
                dm[k] += (Pm[4 * k + 0] * mw - Pm[4 * k + 3] * mul1) * g2x + (Pm[4 * k + 1] * mw - Pm[4 * k + 3] * mul2) * g2y;

            // ---- depth -> mean ----
            const float gdep = gr[GR_DEPTH];
#pragma unroll
            for (int k = 0; k < 3; ++k) dm[k] += V[4 * k + 2] * gdep;

#pragma unroll
            for (int k = 0; k < 3; ++k) dmean[k] += dm[k] * s;
            dop += gr[GR_OP];

            // ---- pose ----
            if (dL_dtau) {
                const float t0 = ge.t[0], t1 = ge.t[1], t2 = ge.t[2];
                const float qx = t0 * Q[0] + t1 * Q[4] + t2 * Q[8] + Q[12];
                const float qy = t0 * Q[1] + t1 * Q[5] + t2 * Q[9] + Q[13];
                const float qw = t0 * Q[3] + t1 * Q[7] + t2 * Q[11] + Q[15];
                const float w1 = 1.0f / (qw + 0.0000001f);
                const float m1 = qx * w1 * w1, m2 = qy * w1 * w1;
                float dtp[3];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    dtp[k] = (Q[4 * k + 0] * w1 - Q[4 * k + 3] * m1) * g2x + (Q[4 * k + 1] * w1 - Q[4 * k + 3] * m2) * g2y;
                dtp[2] += gdep;
#pragma unroll
                for (int k = 0; k < 3; ++k) dtp[k] += dt_cov[k];
                float dR[3][3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    dR[0][k] = ge.J00 * dM0[k];
                    dR[1][k] = ge.J11 * dM1[k];
                    dR[2][k] = ge.J02 * dM0[k] + ge.J12 * dM1[k];
                }
                float Am[3][3];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int l = 0; l < 3; ++l) {
                        float acc = 0.f;
#pragma unroll
                        for (int k = 0; k < 3; ++k) acc += dR[r][k] * V[4 * k + l];
                        Am[r][l] = acc;
                    }
                tau[0] = dtp[0]; tau[1] = dtp[1]; tau[2] = dtp[2];
                tau[3] = (t1 * dtp[2] - t2 * dtp[1]) + (Am[2][1] - Am[1][2]);
                tau[4] = (t2 * dtp[0] - t0 * dtp[2]) + (Am[0][2] - Am[2][0]);
                tau[5] = (t0 * dtp[1] - t1 * dtp[0]) + (Am[1][0] - Am[0][1]);
            }
        }
        if (dL_dmeans2D && valid) {
            float *o = dL_dmeans2D + vg * 3;
            o[0] = g2x; o[1] = g2y; o[2] = 0.f;
        }
        if (dL_dtau) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                float sum = block_sum_256(tau[k], s_tmp, tid);
                if (tid == 0 && sum != 0.f) atomicAdd(dL_dtau + (size_t)v * 6 + k, sum);
            }
        }
    }
    if (valid) {
#pragma unroll
        for (int k = 0; k < 3; ++k) dL_dmeans[3 * sg + k] = dmean[k];
        if (cov9) {
            float *o = dL_dcov6 + 9 * sg;
            o[0] = dcov[0]; o[1] = dcov[1]; o[2] = dcov[2]; o[3] = 0.f; o[4] = dcov[3]; o[5] = dcov[4];
            o[6] = 0.f; o[7] = 0.f; o[8] = dcov[5];
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) dL_dcov6[6 * sg + k] = dcov[k];
        }
        dL_dopac[sg] = dop;
#pragma unroll
        for (int k = 0; k < 3 * NC; ++k) dsh[k] = dsh_acc[k];
        for (int k = 3 * NC; k < ncol; ++k) dsh[k] = 0.f;      // coefficients above the active degree
    }
}
#pragma clang fp contract(fast)

// ------------------------------------------------------------------ host
int layout(const GsrDims &d, long long cap, GsrLayout &L);
Ptrs carve(void *base, const GsrLayout &L);

int backward(const GsrDims &d, const GsrView *views, const float *means, const float *cov6, const float *shs,
             long long cap, void *workspace, size_t workspace_bytes, const float *dL_dimage, const float *dL_ddepth,
             float *dL_dmeans, float *dL_dcov6, float *dL_dopac, float *dL_dshs, float *dL_dmeans2D, float *dL_dtau,
             const GsrFused *fx, hipStream_t stream)
{
    GsrLayout L;
    int rc = layout(d, cap, L);
    if (rc != GSR_OK) return rc;
    const float *mse_target = fx ? fx->mse_target : nullptr;
    if (!views || !means || !cov6 || !shs || !workspace || !dL_dmeans || !dL_dcov6 || !dL_dopac || !dL_dshs) return GSR_EINVAL;
    if (!dL_dimage && !mse_target) return GSR_EINVAL;        // some gradient of the image has to come from somewhere
    if (workspace_bytes < L.total) return GSR_ENOSPACE;
    Ptrs ws = carve(workspace, L);
    const int V = d.B * d.Vt, T = tiles_x(d.W) * tiles_y(d.H);
    (void)hipGetLastError();
    StageTimer tm(d.profile, false, stream);
    // (GSR_FLAG_PREZERO_GRADS: the forward's composite kernel already zeroed the accumulators)
    if (!(d.flags & GSR_FLAG_PREZERO_GRADS) &&
        !hip_ok(hipMemsetAsync(ws.grad_rec, 0, (size_t)V * d.G * GR_STRIDE * 4, stream))) return GSR_ELAUNCH;
    if (dL_dtau && !hip_ok(hipMemsetAsync(dL_dtau, 0, (size_t)V * 6 * 4, stream))) return GSR_ELAUNCH;
    tm.begin(GSR_STAGE_COMPOSITE_BWD);
    {
        const float *mg = fx ? fx->mse_grad_loss : nullptr;
        const float mw = fx ? fx->mse_weight : 0.f;
        const bool mse = mse_target != nullptr;
        if (dL_ddepth) hipLaunchKernelGGL(k_composite_bwd<true>, dim3(T, V), dim3(64), 0, stream, d, views, ws, dL_dimage, dL_ddepth, mse, mw, mg);
        else hipLaunchKernelGGL(k_composite_bwd<false>, dim3(T, V), dim3(64), 0, stream, d, views, ws, dL_dimage, dL_ddepth, mse, mw, mg);
    }
    tm.end(GSR_STAGE_COMPOSITE_BWD); tm.begin(GSR_STAGE_PREPROCESS_BWD);
    const dim3 gG((d.G + 255) / 256, d.B);
#define GSR_LAUNCH_K7(DEG) hipLaunchKernelGGL(k_preprocess_bwd<DEG>, gG, dim3(256), 0, stream, d, views, means, cov6, shs, ws, \
                                              dL_dmeans, dL_dcov6, dL_dopac, dL_dshs, dL_dmeans2D, dL_dtau)
    switch (d.M > 0 ? d.sh_degree : -1) {
        case -1: GSR_LAUNCH_K7(-1); break;
        case 0: GSR_LAUNCH_K7(0); break;
        case 1: GSR_LAUNCH_K7(1); break;
        case 2: GSR_LAUNCH_K7(2); break;
        case 3: GSR_LAUNCH_K7(3); break;
        default: GSR_LAUNCH_K7(4); break;
    }
#undef GSR_LAUNCH_K7
    tm.end(GSR_STAGE_PREPROCESS_BWD);
    return launch_status();
}

}  // namespace gsr
