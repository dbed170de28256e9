// gsr_forward.hip -- forward pipeline of the gfx950 rasterizer.
//
//   K1 preprocess      per Gaussian, loops over the scene's views: cull, cov3D->2D, conic,
//                      radius, tile rect, SH colour; writes one 48-B SplatRec per (view,
//                      Gaussian) and counts the per-tile list lengths through a per-workgroup
//                      LDS histogram (one global atomic per non-empty (workgroup, tile)). (upstream R1)
//   K2 scan_tiles      exclusive scan of the V*T tile counters (one workgroup), the status
//                      words (pair count, overflow, longest list), the longest-list-first
//                      launch order of K4 - K6; re-arms persistent counters.             (R2)
//   K3 scatter         every (view, Gaussian, tile) pair -> its tile bucket as
//                      (depth_bits<<32 | id), slots handed out from LDS.                 (R3)
//   K4 tile_sort       per-tile depth sort in LDS: one MSD radix pass over the 8 most
//                      significant VARYING depth bits of the tile (LDS-atomic histogram,
//                      scan, scatter), then an exact rank inside each bucket on the full
//                      64-bit (depth, id) key == upstream's stable radix order; writes
//                      the sorted id list.                                       (R4,R5)
//   K5 composite_fwd   ONE WAVEFRONT per 16x16 tile, 4 pixels per lane (one per quadrant):
//                      walks the sorted list 64 entries at a time -- each lane gathers one
//                      entry's 48-B splat record, computes the 4-bit mask of the 8x8
//                      quadrants its alpha>=1/255 footprint can reach (left in the top bits
//                      of the entry's point_list word for the backward), and parks the
//                      record in a wave-private LDS slot; the wave then reads the entries
//                      back as LDS broadcasts and evaluates them as STRAIGHT-LINE code: the
//                      four conditions of upstream's loop body are ballots combined in
//                      SGPRs, the selects read those scalar lane masks, quadrants are
//                      skipped by scalar branches on the mask; it exits as soon as its 256
//                      pixels are saturated.  Optional epilogue: the tile's share of
//                      LossMse, reduced in fixed order by last-arriver tickets.          (R6)
//
// The upstream design sorts all pairs of one view globally on 64-bit keys (6-8 radix
// passes over HBM, ~120 B/pair) and reads R back to the host.  Here the tile id never
// enters a sort: pairs are bucketed by tile with one counting pass (K1 counters + K3
// scatter, 8 B/pair written) and each bucket is depth-sorted inside the CU's LDS (8 B/pair
// read + 4 B/pair written), with no host round trip: ~28 B/pair of binning traffic.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "gsr_common.h"

namespace gsr {

// ------------------------------------------------------------------ K1
#pragma clang fp contract(off)
template <int DEG>   // active SH degree 0..4, -1 = precomputed colours: the SH table is an unrolled register array
__global__ void __launch_bounds__(256) k_preprocess(GsrDims d, const GsrView *__restrict__ views,
                                                    const float *__restrict__ means, const float *__restrict__ cov6,
                                                    const float *__restrict__ opac, const float *__restrict__ shs,
                                                    Ptrs ws, int32_t *__restrict__ radii, int lds_tiles)
{
    extern __shared__ uint32_t s_tiles[];      // lds_tiles != 0: this workgroup's per-tile counts of the current view (T words)
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    const bool live = g < d.G;   // no early return: the tile counting below is a wave-level operation
    const size_t sg = (size_t)b * d.G + (live ? g : 0);
    const float m0[3] = {means[3 * sg], means[3 * sg + 1], means[3 * sg + 2]};
    float S0[6];
    load_cov(cov6, sg, (d.flags & GSR_FLAG_COV9) != 0, S0);
    const float op = opac[sg];
    const int gx = tiles_x(d.W), gy = tiles_y(d.H), T = gx * gy;
    constexpr int NC = DEG < 0 ? 1 : (DEG + 1) * (DEG + 1);
    if (lds_tiles) {
        for (int t = threadIdx.x; t < T; t += 256) s_tiles[t] = 0u;
        __syncthreads();
    }

    for (int j = 0; j < d.Vt; ++j) {
        const int v = b * d.Vt + j;
        const GsrView &vw = views[v];
        SplatRec rec;
        rec.x = rec.y = rec.depth = 0.f; rec.rad_flags = 0;
        rec.A = rec.B = rec.C = rec.opacity = 0.f;
        rec.r = rec.g = rec.b = 0.f; rec.ext = 0;
        uint32_t clampbits = 0;
        const size_t vg = (size_t)v * d.G + (live ? g : 0);

        const float s = vw.scale, s2 = s * s;
        const float m[3] = {m0[0] * s, m0[1] * s, m0[2] * s};
        float S[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) S[k] = S0[k] * s2;

        Geom ge;
        bool ok = live && geom_eval(vw.viewmatrix, vw.tanfovx, vw.tanfovy, d.W, d.H, m, S, ge);
        int rad = 0, minx = 0, miny = 0, maxx = 0, maxy = 0;
        float pxx = 0.f, pxy = 0.f, det_inv = 0.f;
        if (ok) {
            const float *P = vw.projmatrix;
            float hx = m[0] * P[0] + m[1] * P[4] + m[2] * P[8] + P[12];
            float hy = m[0] * P[1] + m[1] * P[5] + m[2] * P[9] + P[13];
            float hw = m[0] * P[3] + m[1] * P[7] + m[2] * P[11] + P[15];
            float pw = 1.0f / (hw + 0.0000001f);
            float ndcx = hx * pw, ndcy = hy * pw;
            float det = ge.a * ge.c - ge.b * ge.b;
            ok = (det != 0.0f);
            if (ok) {
                det_inv = 1.0f / det;
                float mid = 0.5f * (ge.a + ge.c);
                float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
                float lam1 = mid + disc, lam2 = mid - disc;
                rad = (int)ceilf(3.0f * sqrtf(fmaxf(lam1, lam2)));
                pxx = ((ndcx + 1.0f) * (float)d.W - 1.0f) * 0.5f;
                pxy = ((ndcy + 1.0f) * (float)d.H - 1.0f) * 0.5f;
                ok = tile_rect(pxx, pxy, rad, gx, gy, minx, miny, maxx, maxy) != 0;
            }
        }
        if (ok) {
            if (DEG >= 0) {
                float dx = m[0] - vw.campos[0], dy = m[1] - vw.campos[1], dz = m[2] - vw.campos[2];
                float len = sqrtf(dx * dx + dy * dy + dz * dz);
                float x = dx / len, y = dy / len, z = dz / len;
                float bs[NC];
                sh_basis(DEG < 0 ? 0 : DEG, x, y, z, bs);
                const float *sh = shs + sg * 3 * (size_t)d.M;
                float col[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float acc = bs[0] * sh[c];
#pragma unroll
                    for (int k = 1; k < NC; ++k) acc = acc + bs[k] * sh[3 * k + c];
                    acc = acc + 0.5f;
                    if (acc < 0.f) clampbits |= (1u << c);
                    col[c] = fmaxf(acc, 0.f);
                }
                rec.r = col[0]; rec.g = col[1]; rec.b = col[2];
            } else {
                rec.r = shs[3 * sg]; rec.g = shs[3 * sg + 1]; rec.b = shs[3 * sg + 2];
            }
            rec.x = pxx; rec.y = pxy; rec.depth = ge.t[2];
            rec.rad_flags = (uint32_t)min(rad, 0xffffff) | (clampbits << 24);
            // conservative half extents of {alpha >= 1/255}: d^T conic d <= 2 ln(255 op), conic^-1 = [[a,b],[b,c]]
            if (op >= (1.f / 255.f)) {
                const float tau2 = 2.0f * __logf(255.0f * op) * 1.002f + 1e-3f;
                const float hx = ceilf(sqrtf(tau2 * ge.a) + 0.05f), hy = ceilf(sqrtf(tau2 * ge.c) + 0.05f);
                rec.ext = (uint32_t)fminf(hx, 65535.f) | ((uint32_t)fminf(hy, 65535.f) << 16);
            }
            rec.A = ge.c * det_inv; rec.B = -ge.b * det_inv; rec.C = ge.a * det_inv; rec.opacity = op;
        }
        // per-tile list lengths.  A workgroup's 256 Gaussians are neighbours (pixel-aligned Gaussians of one image row) and fall into a few
        // dozen tiles of a view: they are counted in an LDS histogram (one ds_add_u32 per lane and covered tile) and only the non-zero bins
        // go to the global counters (round 5; before: every wavefront grouped its lanes by target tile with ballot loops, ~10 instructions per
        // distinct tile and covered-tile index, and issued one global atomic per group -- still the path for images of more than
        // LDS_TILES_MAX tiles)
        if (lds_tiles) {
            const int rw = maxx - minx, area = ok ? rw * (maxy - miny) : 0;
            for (int k = 0; k < area; ++k) atomicAdd(&s_tiles[(miny + k / rw) * gx + minx + k % rw], 1u);
            __syncthreads();
            uint32_t *cnt = ws.tile_count + (size_t)v * T;
            for (int t = threadIdx.x; t < T; t += 256) {
                const uint32_t c = s_tiles[t];
                if (c) { atomicAdd(cnt + t, c); s_tiles[t] = 0u; }
            }
            __syncthreads();
        } else {
            const int rw = maxx - minx, area = ok ? rw * (maxy - miny) : 0;
            int amax = area;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) amax = max(amax, __shfl_xor(amax, o, 64));
            uint32_t *cnt = ws.tile_count + (size_t)v * T;
            for (int k = 0; k < amax; ++k) {
                const bool has = k < area;
                const int ty = has ? miny + k / rw : 0, tx = has ? minx + k % rw : 0;
                wave_agg_inc<false>(cnt, (uint32_t)(ty * gx + tx), has);
            }
        }
        if (live) {
            float4 *dst = reinterpret_cast<float4 *>(ws.records + vg);
            const float4 *src = reinterpret_cast<const float4 *>(&rec);
            dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
            radii[vg] = rad * (ok ? 1 : 0);
        }
    }
}
#pragma clang fp contract(fast)

// ------------------------------------------------------------------ K2
// One workgroup of 1024 threads scans n = V*T counters (n is at most a few 10^4).
// Fast path (n <= 16 384 counters, e.g. 40 views x 256 tiles): the counters go through LDS once (coalesced loads), every
// thread then owns PER consecutive counters in registers: serial scan of its own, one wave scan + one cross-wave fix-up
// of the thread totals -- 6 barriers instead of 3 per 1 024 counters.
constexpr int SCAN_PER_MAX = 16;

// `rearm`: the counters are the caller's persistent ones (GsrFused.tile_count): leave them zero for the next forward.
// `nticket`: fused-MSE arrival counters of this workspace to zero (0 = none).
__global__ void __launch_bounds__(1024) k_scan_tiles(int n, long long capacity, Ptrs ws, int32_t *__restrict__ status, int rearm, int nticket)
{
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned long long s_carry;
    __shared__ uint32_t s_max[16];
    __shared__ uint32_t s_bin[256], s_shift;
    __shared__ uint32_t s_cnt[1024 * SCAN_PER_MAX];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t mx = 0;
    const int per = (n + 1023) / 1024;
    const bool fast = per <= SCAN_PER_MAX;
    uint32_t c[SCAN_PER_MAX];
    if (fast) {
        for (int i = tid; i < per * 1024; i += 1024) s_cnt[i] = (i < n) ? ws.tile_count[i] : 0u;
        if (rearm) for (int i = tid; i < n; i += 1024) ws.tile_count[i] = 0u;       // (read above by the same thread)
        for (int i = tid; i < nticket; i += 1024) ws.loss_ticket[i] = 0u;
        __syncthreads();
        unsigned long long tot = 0;
#pragma unroll
        for (int e = 0; e < SCAN_PER_MAX; ++e) {
            c[e] = (e < per) ? s_cnt[tid * per + e] : 0u;
            mx = max(mx, c[e]);
            tot += c[e];
        }
        unsigned long long x = tot;                       // inclusive scan of the thread totals inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned long long y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) s_wave[wid] = x;
        __syncthreads();
        unsigned long long off = x - tot;
        for (int w = 0; w < wid; ++w) off += s_wave[w];
        if (tid == 1023) s_carry = off + tot;
#pragma unroll
        for (int e = 0; e < SCAN_PER_MAX; ++e) {
            const int i = tid * per + e;
            if (e < per && i < n) {
                ws.tile_offset[i] = (uint32_t)min(off, (unsigned long long)0xffffffffu);
                ws.tile_cursor[i] = 0u;
            }
            off += c[e];
        }
    } else {
        if (tid == 0) s_carry = 0;
        for (int i = tid; i < nticket; i += 1024) ws.loss_ticket[i] = 0u;
        __syncthreads();
        for (int base = 0; base < n; base += 1024) {
            int i = base + tid;
            uint32_t cc = (i < n) ? ws.tile_count[i] : 0u;
            mx = max(mx, cc);
            unsigned long long x = cc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                unsigned long long y = __shfl_up(x, o, 64);
                if (lane >= o) x += y;
            }
            if (lane == 63) s_wave[wid] = x;
            __syncthreads();
            unsigned long long wave_off = 0;
            for (int w = 0; w < wid; ++w) wave_off += s_wave[w];
            unsigned long long carry = s_carry;
            unsigned long long excl = carry + wave_off + x - cc;
            if (i < n) {
                ws.tile_offset[i] = (uint32_t)min(excl, (unsigned long long)0xffffffffu);
                ws.tile_cursor[i] = 0u;
            }
            __syncthreads();
            if (tid == 1023) s_carry = carry + wave_off + x;
            __syncthreads();
        }
    }
    // block max of the tile lengths
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    if (lane == 0) s_max[wid] = mx;
    if (tid < 256) s_bin[tid] = 0u;
    __syncthreads();
    if (tid == 0) {
        uint32_t m = 0;
        for (int w = 0; w < 16; ++w) m = max(m, s_max[w]);
        unsigned long long R = s_carry;
        ws.tile_offset[n] = (uint32_t)min(R, (unsigned long long)0xffffffffu);
        int ovf = (R > (unsigned long long)capacity || R > 0xffffffffull) ? 1 : 0;
        int32_t st[GSR_STATUS_WORDS] = {(int32_t)(R & 0xffffffffull), ovf, (int32_t)m, (int32_t)(R >> 32), 0, 0, 0, 0};
        for (int k = 0; k < GSR_STATUS_WORDS; ++k) { status[k] = st[k]; ws.status[k] = st[k]; }
        uint32_t sh = 0;                       // 256 length classes covering [0, m]
        while ((m >> sh) > 255u) ++sh;
        s_shift = sh;
    }
    __syncthreads();
    // Launch order of the composite kernels: longest lists first (counting sort into 256 length classes), so the
    // workgroups that take longest start first and the tail of K5 / K6 is made of short tiles (LPT scheduling).
    const uint32_t sh = s_shift;
    if (fast) {
#pragma unroll
        for (int e = 0; e < SCAN_PER_MAX; ++e)
            if (e < per && tid * per + e < n) atomicAdd(&s_bin[255u - min(c[e] >> sh, 255u)], 1u);
    } else {
        for (int i = tid; i < n; i += 1024) atomicAdd(&s_bin[255u - min(ws.tile_count[i] >> sh, 255u)], 1u);
    }
    __syncthreads();
    if (tid < 64) {                            // exclusive scan of the 256 bins by one wavefront (4 bins per lane)
        uint32_t b0 = s_bin[tid * 4], b1 = s_bin[tid * 4 + 1], b2 = s_bin[tid * 4 + 2], b3 = s_bin[tid * 4 + 3];
        uint32_t t4 = b0 + b1 + b2 + b3, x = t4;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t y = (uint32_t)__shfl_up((int)x, o, 64);
            if (lane >= o) x += y;
        }
        uint32_t off = x - t4;
        s_bin[tid * 4] = off; s_bin[tid * 4 + 1] = off + b0; s_bin[tid * 4 + 2] = off + b0 + b1; s_bin[tid * 4 + 3] = off + b0 + b1 + b2;
    }
    __syncthreads();
    if (fast) {
#pragma unroll
        for (int e = 0; e < SCAN_PER_MAX; ++e) {
            const int i = tid * per + e;
            if (e < per && i < n) ws.tile_order[atomicAdd(&s_bin[255u - min(c[e] >> sh, 255u)], 1u)] = (uint32_t)i;
        }
    } else {
        for (int i = tid; i < n; i += 1024) {
            const uint32_t pos = atomicAdd(&s_bin[255u - min(ws.tile_count[i] >> sh, 255u)], 1u);
            ws.tile_order[pos] = (uint32_t)i;
            if (rearm) ws.tile_count[i] = 0u;      // its last read
        }
    }
}

// ------------------------------------------------------------------ K3
#pragma clang fp contract(off)
__global__ void __launch_bounds__(256) k_scatter(GsrDims d, Ptrs ws)
{
    if (ws.status[GSR_ST_OVERFLOW]) return;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = blockIdx.y;
    const bool live = g < d.G;
    const size_t vg = (size_t)v * d.G + (live ? g : 0);
    const float4 q0 = reinterpret_cast<const float4 *>(ws.records + vg)[0];
    const int rad = live ? (int)(__float_as_uint(q0.w) & 0xffffffu) : 0;
    const int gx = tiles_x(d.W), gy = tiles_y(d.H), T = gx * gy;
    int minx = 0, miny = 0, maxx = 0, maxy = 0;
    if (rad > 0) tile_rect(q0.x, q0.y, rad, gx, gy, minx, miny, maxx, maxy);
    const unsigned long long key = ((unsigned long long)__float_as_uint(q0.z) << 32) | (unsigned)g;
    const size_t tb = (size_t)v * T;
    const int rw = maxx - minx, area = rw * (maxy - miny);
    int amax = area;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = max(amax, __shfl_xor(amax, o, 64));
    for (int k = 0; k < amax; ++k) {
        const bool has = k < area;
        const int ty = has ? miny + k / rw : 0, tx = has ? minx + k % rw : 0;
        const uint32_t tl = (uint32_t)(ty * gx + tx);
        const uint32_t slot = wave_agg_inc<true>(ws.tile_cursor + tb, tl, has);
        if (has) ws.pairs[ws.tile_offset[tb + tl] + slot] = key;
    }
}
#pragma clang fp contract(fast)

// K3 with the slots handed out from LDS (images of up to LDS_TILES_MAX tiles): count the workgroup's pairs per tile in LDS, reserve ONE range per
// non-empty (workgroup, tile) in the global cursor, then every pair takes base + a returning LDS add.  ~2.5 x fewer global atomics than one per
// (wavefront, distinct tile), and no ballot loops.  The order of a tile's pairs before the sort is arbitrary either way.
__global__ void __launch_bounds__(256) k_scatter_lds(GsrDims d, Ptrs ws)
{
    if (ws.status[GSR_ST_OVERFLOW]) return;
    extern __shared__ uint32_t s_sc[];          // [T] counts, later running offsets ; [T] bases
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = blockIdx.y;
    const bool live = g < d.G;
    const size_t vg = (size_t)v * d.G + (live ? g : 0);
    const float4 q0 = reinterpret_cast<const float4 *>(ws.records + vg)[0];
    const int rad = live ? (int)(__float_as_uint(q0.w) & 0xffffffu) : 0;
    const int gx = tiles_x(d.W), gy = tiles_y(d.H), T = gx * gy;
    uint32_t *s_cnt = s_sc, *s_base = s_sc + T;
    for (int t = threadIdx.x; t < T; t += 256) s_cnt[t] = 0u;
    int minx = 0, miny = 0, maxx = 0, maxy = 0;
    if (rad > 0) tile_rect(q0.x, q0.y, rad, gx, gy, minx, miny, maxx, maxy);
    const unsigned long long key = ((unsigned long long)__float_as_uint(q0.z) << 32) | (unsigned)g;
    const size_t tb = (size_t)v * T;
    const int rw = maxx - minx, area = rw * (maxy - miny);
    __syncthreads();
    for (int k = 0; k < area; ++k) atomicAdd(&s_cnt[(miny + k / rw) * gx + minx + k % rw], 1u);
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) {
        const uint32_t c = s_cnt[t];
        if (c) { s_base[t] = ws.tile_offset[tb + t] + atomicAdd(ws.tile_cursor + tb + t, c); s_cnt[t] = 0u; }
    }
    __syncthreads();
    for (int k = 0; k < area; ++k) {
        const int tl = (miny + k / rw) * gx + minx + k % rw;
        ws.pairs[s_base[tl] + atomicAdd(&s_cnt[tl], 1u)] = key;
    }
}

// ------------------------------------------------------------------ K4
// Normalised bitonic network (every compare-exchange is ascending, so entries at or
// beyond n behave as +inf without being stored).  Keys are unique (id in the low word).
template <typename KeyPtr>
__device__ inline void bitonic_sort_block(KeyPtr key, uint32_t n, int tid, int nthreads)
{
    uint32_t N = 1;
    while (N < n) N <<= 1;
    for (uint32_t k = 2; k <= N; k <<= 1) {
        // first sub-step of the merge: mirror partner inside each block of k
        {
            const uint32_t half = k >> 1;
            for (uint32_t p = tid; p < (N >> 1); p += nthreads) {
                uint32_t blk = p / half, q = p - blk * half;
                uint32_t i = blk * k + q, l = blk * k + (k - 1 - q);
                if (l < n) {
                    unsigned long long a = key[i], c = key[l];
                    if (a > c) { key[i] = c; key[l] = a; }
                }
            }
            __syncthreads();
        }
        for (uint32_t j = k >> 2; j > 0; j >>= 1) {
            for (uint32_t p = tid; p < (N >> 1); p += nthreads) {
                uint32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                uint32_t l = i | j;
                if (l < n) {
                    unsigned long long a = key[i], c = key[l];
                    if (a > c) { key[i] = c; key[l] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// Register-blocked bitonic sort for lists of <= 1024 keys (the common case), 256 threads x 4 keys: every pass takes its
// four keys to registers and applies TWO consecutive strides (j, j/2) before writing back, so N = 1024 needs 29 LDS round
// trips and barriers instead of 55.  Standard (directional) network on a list padded to N with all-ones keys, which sort
// to the top and are never stored back.  Keys are unique, so the result equals any other correct sort bit for bit.
__device__ inline void ce(unsigned long long &a, unsigned long long &b, bool asc)
{
    const bool sw = asc ? (a > b) : (a < b);
    const unsigned long long t = sw ? b : a;
    b = sw ? a : b; a = t;
}
__device__ inline void bitonic_sort_1024(unsigned long long *s, uint32_t n, int tid)
{
    uint32_t N = 4;
    while (N < n) N <<= 1;
    for (uint32_t i = n + tid; i < N; i += 256) s[i] = ~0ull;
    __syncthreads();
    const uint32_t G = N >> 2;                               // four-key groups
    if ((uint32_t)tid < G) {                                 // merges k = 2 and k = 4 on four consecutive keys
        const uint32_t i = (uint32_t)tid << 2;
        unsigned long long e0 = s[i], e1 = s[i + 1], e2 = s[i + 2], e3 = s[i + 3];
        ce(e0, e1, true); ce(e2, e3, false);
        const bool asc = (i & 4u) == 0;
        ce(e0, e2, asc); ce(e1, e3, asc); ce(e0, e1, asc); ce(e2, e3, asc);
        s[i] = e0; s[i + 1] = e1; s[i + 2] = e2; s[i + 3] = e3;
    }
    __syncthreads();
    for (uint32_t k = 8; k <= N; k <<= 1) {
        uint32_t j = k >> 1;
        for (; j >= 2; j >>= 2) {                            // strides j and j/2 in one pass
            if ((uint32_t)tid < G) {
                const uint32_t h = j >> 1, g = (uint32_t)tid;
                const uint32_t i0 = (g / h) * (j << 1) + (g % h);
                const bool asc = (i0 & k) == 0;
                unsigned long long e0 = s[i0], e1 = s[i0 + h], e2 = s[i0 + j], e3 = s[i0 + j + h];
                ce(e0, e2, asc); ce(e1, e3, asc); ce(e0, e1, asc); ce(e2, e3, asc);
                s[i0] = e0; s[i0 + h] = e1; s[i0 + j] = e2; s[i0 + j + h] = e3;
            }
            __syncthreads();
        }
        if (j == 1) {                                        // odd number of strides in this merge: the last one alone
            for (uint32_t p = tid; p < (N >> 1); p += 256) {
                const uint32_t i = p << 1;
                unsigned long long a = s[i], b = s[i + 1];
                ce(a, b, (i & k) == 0);
                s[i] = a; s[i + 1] = b;
            }
            __syncthreads();
        }
    }
}

constexpr uint32_t SORT_LDS_KEYS = 4096;  // at most 32 KiB of (dynamic) LDS per workgroup; GSR_FLAG_SORT_KEYS_* ask for less

// ---- per-tile radix (bucket) sort -----------------------------------------------------------------------------
// One most-significant-digit radix pass over the DEPTH bits that actually vary inside the tile, then an exact rank
// inside every bucket:
//   1. min / max of the depth words of the tile's n keys (keys stay in registers, <= 16 per thread);
//   2. digit = (depth - min) >> shift, shift chosen so that max maps into [0, 255]: the 8 most significant VARYING bits
//      (the constant high bytes of the float never enter -- what a byte-aligned LSD sort would spend passes on);
//   3. 256-bin histogram with LDS atomics, exclusive scan, scatter into the buckets (returning LDS atomics; the order
//      inside a bucket is arbitrary and does not matter);
//   4. every key's final position = bucket start + number of keys of its bucket that compare smaller on the full 64-bit
//      (depth, id) key -- buckets hold n / 256 keys on average, so this is a short loop over LDS words that neighbouring
//      lanes read at the same address (broadcast); the sorted id goes straight to global memory (`point_list`);
//   5. a bucket with more than RANK_MAX keys (depth clustered into a sliver of the tile's range: a near outlier in front
//      of a far plane, or all depths equal) is sorted by the bitonic network instead, in place, by the whole workgroup.
// O(n) + O(n * bucket size) instead of O(n log^2 n); no stability requirement anywhere because positions come from
// comparisons of unique keys: the result is bit-identical to any other correct sort of the (depth, id) keys.
constexpr uint32_t RADIX_BINS = 256, RANK_MAX = 128;

__device__ inline uint32_t radix_digit(unsigned long long key, uint32_t dmin, uint32_t shift)
{
    return ((uint32_t)(key >> 32) - dmin) >> shift;
}

// where a sort step of one tile places its results: the sorted id list (== upstream's sorted value list).
// Round 1 (and the first half of round 2) also gathered the 48-B splat record of every list entry here and wrote it, with
// the quadrant mask, into a per-tile QUEUE that the composite kernels streamed.  Measured (GSR_K4X builds, r02l): the
// sort proper is 0.04 ms of this kernel's 0.17 at the headline size and 0.24 of 0.77 at 262 144 Gaussians -- the rest
// was that gather + queue write, bound by bytes (4.5-4.8 TB/s), and the composite kernels then read the 48 B a third
// and a fourth time.  Now the composite kernels gather the record themselves while they stage a batch (one gather per
// lane per 64 entries, under VALU-bound work that leaves the memory system idle): 192 -> 104 bytes moved per pair.
struct TileOut {
    uint32_t *point_list;   // already offset to the tile's range
    __device__ void put(uint32_t pos, unsigned long long key) const { point_list[pos] = (uint32_t)(key & 0xffffffffull); }
};

// s_key[0, m) holds the buckets [b0, b1) of the tile, bucket b at local offset s_start[b] - s_start[b0].
// Exact rank inside every bucket of at most RANK_MAX keys; an oversized one is sorted in place by the bitonic network
// (whole workgroup) and written linearly.  Ends with the LDS array free for the next group.
__device__ inline void rank_and_emit(unsigned long long *s_key, uint32_t m, uint32_t b0, uint32_t b1, const uint32_t *s_start,
                                     const uint32_t *s_big, uint32_t nbig, uint32_t dmin, uint32_t shift, const TileOut &out, int tid)
{
    const uint32_t base = s_start[b0];
    for (uint32_t p = tid; p < m; p += 256) {
        const unsigned long long key = s_key[p];
        const uint32_t dg = radix_digit(key, dmin, shift);
        const uint32_t s0 = s_start[dg] - base, s1 = s_start[dg + 1] - base;
        if (s1 - s0 > RANK_MAX) continue;
        uint32_t rank = s0;
        for (uint32_t q = s0; q < s1; ++q) rank += (s_key[q] < key) ? 1u : 0u;
        out.put(base + rank, key);
    }
    for (uint32_t b = 0; b < nbig; ++b) {
        const uint32_t dg = s_big[b];
        if (dg < b0 || dg >= b1) continue;        // (uniform)
        const uint32_t s0 = s_start[dg] - base, cnt = s_start[dg + 1] - s_start[dg];
        __syncthreads();                          // the rank phase (and the previous bucket's output) are done with s_key
        bitonic_sort_block(s_key + s0, cnt, tid, 256);
        for (uint32_t i = tid; i < cnt; i += 256) out.put(base + s0 + i, s_key[s0 + i]);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_tile_sort(GsrDims d, Ptrs ws, uint32_t lds_keys)
{
    if (ws.status[GSR_ST_OVERFLOW]) return;
    extern __shared__ unsigned long long s_key[];   // lds_keys entries
    __shared__ uint32_t s_hist[RADIX_BINS], s_start[RADIX_BINS + 1], s_big[RADIX_BINS];
    __shared__ uint32_t s_red[12], s_nbig;
    const int gx = tiles_x(d.W), T = gx * tiles_y(d.H);
    // (a natural (view, tile) order cut into one contiguous range per XCD, so that the ~2.5 tiles a splat lies in would gather
    //  its record through one L2, was measured: 2-6 % SLOWER than this order at every size)
    const uint32_t tv = ws.tile_order[blockIdx.y * gridDim.x + blockIdx.x];   // longest lists first (as K5 / K6)
    const int tile = (int)(tv % (uint32_t)T), v = (int)(tv / (uint32_t)T);
    const size_t t = (size_t)v * T + tile;
    const uint32_t start = ws.tile_offset[t];
    const uint32_t n = ws.tile_offset[t + 1] - start;
    if (n == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    TileOut out;
    out.point_list = ws.point_list + start;
    const unsigned long long *gk = ws.pairs + start;
    if (n <= 64) {
        // short list: the register-blocked bitonic network (a handful of passes at this size)
        for (uint32_t i = tid; i < n; i += 256) s_key[i] = gk[i];
        __syncthreads();
        if (n > 1) bitonic_sort_1024(s_key, n, tid);
        for (uint32_t i = tid; i < n; i += 256) out.put(i, s_key[i]);
        return;
    }
    // ---- 1. depth range of the tile.  Lists of up to 1 024 keys (the common case: 620 on average at the headline) are read ONCE, four keys
    //      per thread in registers (round 5: the kernel is a chain of dependent passes per workgroup, and each re-read of the keys from L2 was
    //      one more memory round trip in that chain); longer lists re-read their keys in every pass, 8 B each ----
    const bool in_regs = n <= 1024u;
    unsigned long long kreg[4] = {0ull, 0ull, 0ull, 0ull};
    if (in_regs) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const uint32_t i = tid + 256u * e; if (i < n) kreg[e] = gk[i]; }
    }
    // fn(key) for every key of the list this thread owns (i = tid, tid + 256, ...)
    auto for_my_keys = [&](auto fn) {
        if (in_regs) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (tid + 256u * e < n) fn(kreg[e]);
        } else {
            for (uint32_t i = tid; i < n; i += 256) fn(gk[i]);
        }
    };
    uint32_t mn = 0xffffffffu, mx = 0u;
    for_my_keys([&](unsigned long long key) { const uint32_t dep = (uint32_t)(key >> 32); mn = min(mn, dep); mx = max(mx, dep); });
    s_hist[tid] = 0u;
    if (tid == 0) s_nbig = 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    }
    if (lane == 0) { s_red[wid] = mn; s_red[4 + wid] = mx; }
    __syncthreads();
    const uint32_t dmin = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
    const uint32_t dmax = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
    const uint32_t range = dmax - dmin;
    const uint32_t nbits = range ? 32u - (uint32_t)__clz((int)range) : 0u;
    const uint32_t shift = nbits > 8u ? nbits - 8u : 0u;
    // ---- 2. histogram of the 8 most significant varying depth bits ----
    for_my_keys([&](unsigned long long key) { atomicAdd(&s_hist[radix_digit(key, dmin, shift)], 1u); });
    __syncthreads();
    // ---- 3. exclusive scan of the 256 bins (thread = bin), oversized bins noted ----
    {
        const uint32_t c = s_hist[tid];
        uint32_t x = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) s_red[8 + wid] = x;
        __syncthreads();
        uint32_t off = x - c;
        for (int w = 0; w < wid; ++w) off += s_red[8 + w];
        s_start[tid] = off;
        if (tid == 255) s_start[256] = off + c;
        if (c > RANK_MAX) s_big[atomicAdd(&s_nbig, 1u)] = (uint32_t)tid;
        s_hist[tid] = 0u;                    // becomes the bucket cursor
    }
    __syncthreads();
    const uint32_t nbig = s_nbig;
    if (n <= lds_keys) {
        // ---- 4. scatter into the buckets in LDS, 5. rank + emit ----
        for_my_keys([&](unsigned long long key) {
            const uint32_t dg = radix_digit(key, dmin, shift);
            s_key[s_start[dg] + atomicAdd(&s_hist[dg], 1u)] = key;
        });
        __syncthreads();
        rank_and_emit(s_key, n, 0u, RADIX_BINS, s_start, s_big, nbig, dmin, shift, out, tid);
        return;
    }
    // ---- list longer than the LDS budget: bucket it in global memory (pairs_alt), then take runs of consecutive buckets that
    //      fit through LDS one after the other.  (One workgroup owns the list: __syncthreads orders its own global accesses
    //      through the CU's write-through L1.) ----
    unsigned long long *alt = ws.pairs_alt + start;
    for (uint32_t i = tid; i < n; i += 256) {
        const unsigned long long key = gk[i];
        const uint32_t dg = radix_digit(key, dmin, shift);
        alt[s_start[dg] + atomicAdd(&s_hist[dg], 1u)] = key;
    }
    __syncthreads();
    uint32_t b0 = 0;
    while (b0 < RADIX_BINS) {
        uint32_t b1 = b0 + 1;
        while (b1 < RADIX_BINS && s_start[b1 + 1] - s_start[b0] <= lds_keys) ++b1;
        const uint32_t base = s_start[b0], m = s_start[b1] - base;
        if (m > lds_keys) {
            // a single bucket beyond the LDS budget (depths clustered AND a very long list): bitonic network in global memory
            bitonic_sort_block(alt + base, m, tid, 256);
            for (uint32_t i = tid; i < m; i += 256) out.put(base + i, alt[base + i]);
            __syncthreads();
        } else if (m > 0) {
            for (uint32_t i = tid; i < m; i += 256) s_key[i] = alt[base + i];
            __syncthreads();
            rank_and_emit(s_key, m, b0, b1, s_start, s_big, nbig, dmin, shift, out, tid);
        }
        b0 = b1;
    }
}

// ------------------------------------------------------------------ K5
// One wavefront per tile.  Lane l owns pixel (l&7, l>>3) of each 8x8 quadrant k = 0..3
// (TL, TR, BL, BR), so a splat whose footprint misses a quadrant costs that quadrant nothing
// (wave-uniform scalar branch on the entry's quadrant mask).
//
// Fused LossMse (mse_target != nullptr; src/loss/loss_mse.py:22-31): the tile's sum of (image - target)^2 is formed from the registers
// that hold the finished pixels, published with a device-scope atomic, and the LAST tile of a view to arrive (ticket) adds the view's
// T partials in index order; the last view adds the V view sums the same way and writes weight * total / n.  Fixed summation order =>
// bitwise deterministic, no extra launch and no second pass over the image.  Publishing and reading go through returning agent-scope
// atomics (performed at the coherence point of the eight per-XCD L2s) and an exact `s_waitcnt vmcnt(0)` in front of the ticket: the
// release fence of the stand-alone kernel (an L2 write-back per workgroup) would be paid 10 240 times here.
__device__ inline void agent_publish(float *p, float v)      // returning integer swap: vmcnt counts it, the value is at the coherence point when it is back
{
    (void)__hip_atomic_exchange(reinterpret_cast<uint32_t *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline float agent_read(float *p)
{
    return __uint_as_float(__hip_atomic_fetch_or(reinterpret_cast<uint32_t *>(p), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

template <bool NTOUCH>
__global__ void __launch_bounds__(64) k_composite_fwd(GsrDims d, const GsrView *__restrict__ views, Ptrs ws,
                                                     float *__restrict__ image, float *__restrict__ out_depth,
                                                     float *__restrict__ out_opacity, int32_t *__restrict__ n_touched,
                                                     const float *__restrict__ mse_target, float mse_weight,
                                                     float *__restrict__ mse_loss)
{
    if (ws.status[GSR_ST_OVERFLOW]) return;
    __shared__ float4 s_q[64 * 3];

    const int gx = tiles_x(d.W), T = gx * tiles_y(d.H);
    const uint32_t tv = ws.tile_order[blockIdx.y * gridDim.x + blockIdx.x];   // longest lists are launched first
    const int tile = (int)(tv % (uint32_t)T), v = (int)(tv / (uint32_t)T);
    const int lane = threadIdx.x;
    const int ox = (tile % gx) * TILE + (lane & 7), oy = (tile / gx) * TILE + (lane >> 3);

    if (d.flags & GSR_FLAG_PREZERO_GRADS) {
        // side job: this wavefront's slice of the backward's gradient accumulators (stores only; the kernel's own work is
        // VALU-bound, so the 48 B per (view, Gaussian) ride on otherwise idle HBM bandwidth instead of a separate memset)
        const size_t n4 = (size_t)d.B * d.Vt * d.G * (GR_STRIDE / 4);
        const size_t nwg = (size_t)gridDim.x * gridDim.y, wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        const size_t per = (n4 + nwg - 1) / nwg, lo = wg * per, hi = lo + per < n4 ? lo + per : n4;
        float4 *gz = reinterpret_cast<float4 *>(ws.grad_rec);
        for (size_t i = lo + lane; i < hi; i += 64) gz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    const size_t t = (size_t)v * T + tile;
    const uint32_t start = ws.tile_offset[t];
    const int n = (int)(ws.tile_offset[t + 1] - start);
    uint32_t *__restrict__ plist = ws.point_list + start;
    const SplatRec *__restrict__ recs = ws.records + (size_t)v * d.G;
    const int tile_ox = (tile % gx) * TILE, tile_oy = (tile / gx) * TILE;

    // (pixel coordinates from the lane's base + a per-quadrant constant, as in the backward: the registers this frees are a wave per SIMD)
    float fx0 = (float)ox, fy0 = (float)oy;
    asm volatile("" : "+v"(fx0), "+v"(fy0));     // (kept in registers: the compiler re-converted ox per list entry)
    float Tr[4], C0[4], C1[4], C2[4], D[4], O[4];
    uint32_t last[4];
    bool inside[4];
    // finished pixels (outside the image, or T fell below 1e-4) as SCALAR lane masks, one per quadrant: the evaluation below is straight-line
    // code whose selects read scalar masks (round 5; the round-1..4 form nested four exec-mask branches per evaluation -- done, power > 0,
    // alpha < 1/255, T < 1e-4 -- and kept `done` as a register that every evaluation tested with two VALU instructions)
    unsigned long long dmask[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int px = ox + (k & 1) * 8, py = oy + (k >> 1) * 8;
        inside[k] = px < d.W && py < d.H;
        Tr[k] = 1.f; C0[k] = C1[k] = C2[k] = D[k] = O[k] = 0.f;
        last[k] = 0;
        dmask[k] = __builtin_amdgcn_ballot_w64(!inside[k]);
    }

    for (int base = 0; base < n; base += 64) {
        const int cnt = __builtin_amdgcn_readfirstlane(min(64, n - base));     // (scalar loop control)
        __syncthreads();  // single-wave workgroup: orders this wave's LDS reads of the previous batch
        uint32_t qm = 0;
        if (lane < cnt) {  // one list entry per lane; the geometric quadrant mask steers this kernel's own scalar skips and, through the top bits
                           // of the entry's list word, the backward's (entries the footprint cannot reach keep their zero bits: nothing is written)
            const uint32_t id = plist[base + lane] & GSR_ID_MASK;
            qm = stage_entry_fwd(recs, id, tile_ox, tile_oy, s_q + lane * 3);
            if (qm) plist[base + lane] = id | (qm << GSR_QUAD_SHIFT);
        }
        __syncthreads();

        // the entry loop visits only the slots whose footprint can touch the tile (a scalar bit mask of the batch)
        // (measured and dropped: handing the backward the quadrants that really composited something -- four scalar masks per batch -- instead of
        //  the geometric ones: K6 0.768 -> 0.761 ms, this kernel 0.394 -> 0.418: the ellipse-vs-rectangle test is already that tight)
        // (reading entry j + 1 ahead of entry j's evaluation was measured: +7 VGPRs, 8 -> 7 waves per SIMD, -6 %)
        for (unsigned long long todo = __builtin_amdgcn_ballot_w64(qm != 0u); todo; todo &= todo - 1ull) {
            const int j = __builtin_ctzll(todo);
            const float4 a = s_q[j * 3 + 0];
            const float4 b = s_q[j * 3 + 1];
            const float4 c = s_q[j * 3 + 2];
            const uint32_t quad = __builtin_amdgcn_readfirstlane(__float_as_uint(c.w));
            uint32_t touched_tot = 0;             // scalar: pixels of this tile whose T stays above 1/2 behind the splat
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!(quad & (1u << k))) continue;        // scalar branches: the only control flow of the evaluation
                if (dmask[k] == ~0ull) continue;
                const float dx = (a.x - fx0) - (float)((k & 1) * 8), dy = (a.y - fy0) - (float)((k >> 1) * 8);
                const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
                // (evaluated for every lane: a positive power gives G > 1 or +inf, alpha = 0.99 after the clamp -- v_min_f32 returns the number
                //  of (0.99, NaN) -- and the lane is masked out by `live`)
                const float alpha = fminf(0.99f, b.y * footprint_exp(power));
                const float test_T = Tr[k] * (1.f - alpha);
                // (measured and dropped: a per-entry flag "this conic cannot produce a positive power" and a scalar branch around that compare:
                //  this kernel 0.393 -> 0.402 ms, the backward 0.768 -> 0.762)
                const unsigned long long live = __builtin_amdgcn_ballot_w64(!(power > 0.f)) & __builtin_amdgcn_ballot_w64(!(alpha < (1.f / 255.f))) & ~dmask[k];
                const unsigned long long keep = __builtin_amdgcn_ballot_w64(!(test_T < 0.0001f));
                dmask[k] |= live & ~keep;                  // T would fall below 1e-4: the pixel is finished, this splat is not composited
                const unsigned long long comp = live & keep;
                const float w = sel0_f(comp, alpha * Tr[k]);
                C0[k] += c.x * w; C1[k] += c.y * w; C2[k] += c.z * w;
                D[k] += b.z * w;
                O[k] += w;
                if (NTOUCH) touched_tot += (uint32_t)__popcll(comp & __builtin_amdgcn_ballot_w64(test_T > 0.5f));
                Tr[k] = sel_f(comp, test_T, Tr[k]);
                last[k] = sel_u(comp, (uint32_t)(base + j + 1), last[k]);
            }
            if (NTOUCH) {   // one atomic per (tile, splat)
                if (touched_tot && lane == 0) atomicAdd(n_touched + (size_t)v * d.G + (__float_as_uint(b.w)), (int)touched_tot);
            }
        }
        if ((dmask[0] & dmask[1] & dmask[2] & dmask[3]) == ~0ull) break;
    }

    const size_t P = (size_t)d.H * d.W;
    const GsrView &vw = views[v];
    float sq = 0.f;             // fused MSE: this lane's sum of squared errors
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!inside[k]) continue;
        const size_t pix = (size_t)(oy + (k >> 1) * 8) * d.W + (ox + (k & 1) * 8);
        ws.final_T[v * P + pix] = Tr[k];
        ws.n_contrib[v * P + pix] = last[k];
        const float r0 = C0[k] + Tr[k] * vw.bg[0], r1 = C1[k] + Tr[k] * vw.bg[1], r2 = C2[k] + Tr[k] * vw.bg[2];
        image[(v * 3 + 0) * P + pix] = r0;
        image[(v * 3 + 1) * P + pix] = r1;
        image[(v * 3 + 2) * P + pix] = r2;
        out_depth[v * P + pix] = D[k];
        out_opacity[v * P + pix] = O[k];
        if (mse_target) {       // (wave-uniform)
            const float e0 = r0 - mse_target[(v * 3 + 0) * P + pix], e1 = r1 - mse_target[(v * 3 + 1) * P + pix],
                        e2 = r2 - mse_target[(v * 3 + 2) * P + pix];
            sq += (e0 * e0 + e1 * e1) + e2 * e2;
            ws.loss_diff[(v * 3 + 0) * P + pix] = e0; ws.loss_diff[(v * 3 + 1) * P + pix] = e1; ws.loss_diff[(v * 3 + 2) * P + pix] = e2;
        }
    }
    if (!mse_target) return;
    const int V = d.B * d.Vt;
    sq = wave_sum_to_lane63(sq);
    uint32_t arrived = 0;
    if (lane == 63) {
        agent_publish(ws.loss_partial + t, sq);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the partial is at the coherence point before the ticket moves
        arrived = __hip_atomic_fetch_add(ws.loss_ticket + v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    arrived = (uint32_t)__builtin_amdgcn_readlane((int)arrived, 63);
    if (arrived != (uint32_t)T - 1u) return;
    float s1 = 0.f;                                             // last tile of view v: the view's partials, lane-strided then one fixed tree
    for (int i = lane; i < T; i += 64) s1 += agent_read(ws.loss_partial + (size_t)v * T + i);
    s1 = wave_sum_to_lane63(s1);
    arrived = 0;
    if (lane == 63) {
        agent_publish(ws.loss_partial + (size_t)V * T + v, s1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        arrived = __hip_atomic_fetch_add(ws.loss_ticket + V, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    arrived = (uint32_t)__builtin_amdgcn_readlane((int)arrived, 63);
    if (arrived != (uint32_t)V - 1u) return;
    float s2 = 0.f;                                             // last view: the V view sums
    for (int i = lane; i < V; i += 64) s2 += agent_read(ws.loss_partial + (size_t)V * T + i);
    s2 = wave_sum_to_lane63(s2);
    if (lane == 63) mse_loss[0] = mse_weight * (s2 / (float)((size_t)V * 3 * P));
}

// ------------------------------------------------------------------ host
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int layout(const GsrDims &d, long long cap, GsrLayout &L)
{
    if (d.B <= 0 || d.Vt <= 0 || d.G <= 0 || d.H <= 0 || d.W <= 0 || cap <= 0) return GSR_EINVAL;
    if (d.M < 0 || d.sh_degree < 0 || d.sh_degree > 4) return GSR_EINVAL;
    if (d.M > 0 && (d.sh_degree + 1) * (d.sh_degree + 1) > d.M) return GSR_EINVAL;
    if (d.M == 0 && d.sh_degree != 0) return GSR_EINVAL;
    if (cap > 0xffffffffll) return GSR_EINVAL;
    // 32-bit byte offsets into one view's records / gradient records: G < 89 M < 2^27, so a point_list word has room for the 4 mask bits
    if ((long long)d.G * (long long)sizeof(SplatRec) > 0xffffffffll) return GSR_EINVAL;
    static_assert((0xffffffffull / sizeof(SplatRec)) < GSR_ID_MASK, "ids must leave the quadrant-mask bits of a point_list word free");
    const size_t V = (size_t)d.B * d.Vt, T = (size_t)tiles_x(d.W) * tiles_y(d.H), P = (size_t)d.H * d.W;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    L.records = take(V * d.G * sizeof(SplatRec));
    L.tile_count = take(V * T * 4);
    L.tile_offset = take((V * T + 1) * 4);
    L.tile_cursor = take(V * T * 4);
    L.pairs = take((size_t)cap * 8);
    L.point_list = take((size_t)cap * 4);
    L.final_T = take(V * P * 4);
    L.n_contrib = take(V * P * 4);
    L.grad_rec = take(V * d.G * 12 * 4);
    L.status = take(GSR_STATUS_WORDS * 4);
    L.tile_order = take(V * T * 4);
    L.pairs_alt = take((size_t)cap * 8);
    L.loss_partial = take((V * T + V) * 4);
    L.loss_ticket = take((V + 1) * 4);
    L.loss_diff = take(V * 3 * P * 4);
    L.total = off;
    return GSR_OK;
}

Ptrs carve(void *base, const GsrLayout &L)
{
    char *p = static_cast<char *>(base);
    Ptrs w;
    w.records = reinterpret_cast<SplatRec *>(p + L.records);
    w.tile_count = reinterpret_cast<uint32_t *>(p + L.tile_count);
    w.tile_offset = reinterpret_cast<uint32_t *>(p + L.tile_offset);
    w.tile_cursor = reinterpret_cast<uint32_t *>(p + L.tile_cursor);
    w.pairs = reinterpret_cast<unsigned long long *>(p + L.pairs);
    w.point_list = reinterpret_cast<uint32_t *>(p + L.point_list);
    w.final_T = reinterpret_cast<float *>(p + L.final_T);
    w.n_contrib = reinterpret_cast<uint32_t *>(p + L.n_contrib);
    w.grad_rec = reinterpret_cast<float *>(p + L.grad_rec);
    w.status = reinterpret_cast<int32_t *>(p + L.status);
    w.tile_order = reinterpret_cast<uint32_t *>(p + L.tile_order);
    w.pairs_alt = reinterpret_cast<unsigned long long *>(p + L.pairs_alt);
    w.loss_partial = reinterpret_cast<float *>(p + L.loss_partial);
    w.loss_ticket = reinterpret_cast<uint32_t *>(p + L.loss_ticket);
    w.loss_diff = reinterpret_cast<float *>(p + L.loss_diff);
    return w;
}

int forward(const GsrDims &d, const GsrView *views, const float *means, const float *cov6, const float *opac,
            const float *shs, long long cap, void *workspace, size_t workspace_bytes, float *image, float *depth,
            float *opacity, int32_t *radii, int32_t *n_touched, int32_t *status, const GsrFused *fx, hipStream_t stream)
{
    GsrLayout L;
    int rc = layout(d, cap, L);
    if (rc != GSR_OK) return rc;
    if (!views || !means || !cov6 || !opac || !shs || !workspace || !image || !depth || !opacity || !radii || !status)
        return GSR_EINVAL;
    const bool ntouch = (d.flags & GSR_FLAG_NTOUCHED) != 0;
    if (ntouch && !n_touched) return GSR_EINVAL;
    if (workspace_bytes < L.total) return GSR_ENOSPACE;
    const float *mse_target = fx ? fx->mse_target : nullptr;
    if (mse_target && !fx->mse_loss) return GSR_EINVAL;
    Ptrs ws = carve(workspace, L);
    const bool persistent_counters = fx && fx->tile_count;
    if (persistent_counters) ws.tile_count = fx->tile_count;
    const int V = d.B * d.Vt, gx = tiles_x(d.W), gy = tiles_y(d.H), T = gx * gy;
    (void)hipGetLastError();  // drop stale (non-sticky) errors of earlier runtime calls, e.g. hipErrorNotReady polls
    const bool bin = (d.flags & GSR_FLAG_PHASE_BIN) != 0, render = (d.flags & GSR_FLAG_PHASE_RENDER) != 0;
    if (bin && render) return GSR_EINVAL;
    StageTimer tm(d.profile, true, stream, render);
    const dim3 gG((d.G + 255) / 256, d.B), gV((d.G + 255) / 256, V);
    // K1 / K3 bin through LDS histograms when a view's tiles fit (GSR_FLAG_BIN_BALLOT: the wave-aggregated global atomics of rounds 1 - 4, the
    // path of larger images, forced for A/B runs and the equivalence test)
    const int lds_tiles = (T <= LDS_TILES_MAX && !(d.flags & GSR_FLAG_BIN_BALLOT)) ? T : 0;
    if (render) goto render_phase;   // K1-K2 of this workspace were enqueued by the PHASE_BIN call

    if (!persistent_counters && !hip_ok(hipMemsetAsync(ws.tile_count, 0, (size_t)V * T * 4, stream))) return GSR_ELAUNCH;
    if (ntouch && !hip_ok(hipMemsetAsync(n_touched, 0, (size_t)V * d.G * 4, stream))) return GSR_ELAUNCH;

    tm.begin(GSR_STAGE_PREPROCESS);
#define GSR_LAUNCH_K1(DEG) hipLaunchKernelGGL(k_preprocess<DEG>, gG, dim3(256), lds_tiles * 4, stream, d, views, means, cov6, opac, shs, ws, radii, lds_tiles)
    switch (d.M > 0 ? d.sh_degree : -1) {
        case -1: GSR_LAUNCH_K1(-1); break;
        case 0: GSR_LAUNCH_K1(0); break;
        case 1: GSR_LAUNCH_K1(1); break;
        case 2: GSR_LAUNCH_K1(2); break;
        case 3: GSR_LAUNCH_K1(3); break;
        default: GSR_LAUNCH_K1(4); break;
    }
#undef GSR_LAUNCH_K1
    tm.end(GSR_STAGE_PREPROCESS); tm.begin(GSR_STAGE_SCAN);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, stream, V * T, cap, ws, status, persistent_counters ? 1 : 0, mse_target ? V + 1 : 0);
    tm.end(GSR_STAGE_SCAN);
    if (bin) return launch_status();  // status is final here: the host can size / retry before the heavy stages
render_phase:
    tm.begin(GSR_STAGE_SCATTER);
    if (lds_tiles) hipLaunchKernelGGL(k_scatter_lds, gV, dim3(256), lds_tiles * 8, stream, d, ws);
    else hipLaunchKernelGGL(k_scatter, gV, dim3(256), 0, stream, d, ws);
    tm.end(GSR_STAGE_SCATTER); tm.begin(GSR_STAGE_SORT);
    {
        // LDS budget of the per-tile sort: 1024 / 2048 / 4096 keys (8 / 16 / 32 KiB).  The host passes the longest list it
        // has seen for this shape (GSR_FLAG_SORT_KEYS_*); a smaller budget doubles the resident workgroups per CU, and a
        // list that exceeds it is still sorted correctly (in place in global memory, slower).
        const int sel = (d.flags >> GSR_FLAG_SORT_KEYS_SHIFT) & 3;
        const uint32_t lds_keys = sel == 1 ? 1024u : (sel == 2 ? 2048u : SORT_LDS_KEYS);
        hipLaunchKernelGGL(k_tile_sort, dim3(T, V), dim3(256), lds_keys * 8, stream, d, ws, lds_keys);
    }
    tm.end(GSR_STAGE_SORT); tm.begin(GSR_STAGE_COMPOSITE_FWD);
    {
        const float mse_weight = fx ? fx->mse_weight : 0.f;
        float *mse_loss = fx ? fx->mse_loss : nullptr;
        if (ntouch) hipLaunchKernelGGL(k_composite_fwd<true>, dim3(T, V), dim3(64), 0, stream, d, views, ws, image, depth, opacity, n_touched, mse_target, mse_weight, mse_loss);
        else hipLaunchKernelGGL(k_composite_fwd<false>, dim3(T, V), dim3(64), 0, stream, d, views, ws, image, depth, opacity, n_touched, mse_target, mse_weight, mse_loss);
    }
    tm.end(GSR_STAGE_COMPOSITE_FWD);
    return launch_status();
}

}  // namespace gsr
