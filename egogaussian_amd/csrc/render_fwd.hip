// render_fwd.hip -- front-to-back compositing of colour, depth and alpha for gfx950.
// Replaces the forward `render` stage of the upstream op called from
// /root/reference/gaussian_renderer/__init__.py:90-98.
//
// Mapping (wave64-first, not a 16x16-thread block port):
//   * a workgroup is one 16x16 tile = 4 waves; each WAVE owns one 8x8 pixel quadrant and runs on its
//     own -- there is no __syncthreads() anywhere, a wave whose 64 pixels are saturated simply leaves;
//   * the tile's sorted splat list is consumed in batches of 64: lane j fetches list entry j's packed
//     48-byte record (three dwordx4, L2-resident gather), tests the record's pixel bounding box against
//     the wave's quadrant, and a 64-bit ballot becomes the work list -- splats that cannot reach the
//     quadrant cost nothing (s_ff1 over the mask), which removes roughly half of the (pixel, splat)
//     evaluations of a whole-tile loop at 3DGS-typical footprints;
//   * per-pixel state is branch-free: predicates live in VCC and feed v_cndmask directly, a stopped pixel is
//     represented by a live transmittance of 0, and the loop exit is one s_cbranch on the `cont` compare;
//   * surviving records are parked in a wave-private 3 KiB LDS slice and re-read with a wave-uniform
//     address (hardware broadcast, conflict-free) -- the LDS is a register-file extension here, not a
//     cross-wave exchange;
//   * the next batch's ids and records are issued before the current batch is blended, so the gather
//     latency hides under VALU work.
// Tried and rejected (round 1, config C): feeding the per-splat record through the scalar path instead of LDS
// (v_readlane id -> s_load_dwordx8 + x2 into SGPRs, VALU reads SGPR operands, no LDS in the loop).  Compiler-scheduled:
// 100 us; hand-placed s_waitcnt with the next splat's s_load in flight during the blend: 86 us; LDS staging: 70 us.
// One s_load per (wave, splat) pays an L2 round trip each, while the LDS design gathers 64 records at once a batch ahead.
// Also rejected: one-wave workgroups with the occupancy capped through LDS padding, so that the dispatcher hands out the
// last quadrants dynamically (8 / 6 / 5 / 4 waves per SIMD: 70 / 76 / 79 / 81 us) -- latency hiding beats balance here.
// Arithmetic: alpha evaluation is shared with the backward (blend_common.h) so both make identical
// keep/skip decisions.
#include "egs_common.h"
#include "blend_common.h"
#include "tile_sort.h"            // SORT instantiations: the workgroup sorts its tile's bucket before it blends it
#include "backward_prologue.h"
#include "blend_instrument.h"      // measurement / ablation hooks: all empty in the product build

namespace {

#ifndef EGS_FWD_COST_BATCH      // cost model of the placement hint: 10 per blended splat + this per batch of 64 list entries scanned
#define EGS_FWD_COST_BATCH 25u
#endif
#ifndef EGS_FWD_LRPT1           // estimated splats left at which the forward's issue priority steps up (swept at config C)
#define EGS_FWD_LRPT1 16
#define EGS_FWD_LRPT2 48
#define EGS_FWD_LRPT3 128
#endif

// DA = false (ABI 4: out_depth == out_alpha == NULL): the caller reads the colour image only -- the training step's loss -- so the depth and
// alpha sums (two of the ~26 vector instructions per visit) and their two planes are left out.
// SORT (RA: the fast LDS ranking, tile_sort.h): the per-tile depth sort runs HERE, in the workgroup that is about to blend the tile -- a tile's
// bucket is private to it, so the sort needs no launch of its own (k_tile_sort, ~4.5 us of ramp and drain per frame); the sorted ids go to
// point_list as before (the backward reads them) and are read back through the CU's L1.  The sort's exchange buffer and bucket counters
// (18.5 KB) and the blend's staged records (12 KB) share the LDS.
template <bool DA, bool SORT, bool RA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_render_forward(
    int W, int H, int gx, int n_tiles, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ rec, const float* __restrict__ bg, float* __restrict__ out_color,
    float* __restrict__ out_depth, float* __restrict__ out_alpha, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ quad_work, uint32_t* __restrict__ quad_pairs,
    const uint32_t* __restrict__ order, uint32_t* __restrict__ cost_hint, const EgsSortArgs sortA) {
    constexpr int SM_BYTES = SORT ? (TS_SMALL_CAP * 8 + 2 * 512 * 4 + 64) : 4 * 64 * EGS_SPLAT_REC_F4 * 16;
    static_assert(SM_BYTES >= 4 * 64 * EGS_SPLAT_REC_F4 * 16, "the staged records fit");
    __shared__ __attribute__((aligned(16))) uint64_t smem64[SM_BYTES / 8];
    float4 (*lds)[64 * EGS_SPLAT_REC_F4] = reinterpret_cast<float4 (*)[64 * EGS_SPLAT_REC_F4]>(smem64);
    __shared__ uint32_t quad_claimed;
    if (SORT && sortA.zero_after)         // (what k_tile_sort's first launch also did: the fused count pass's chunk sums are consumed; every workgroup of the grid)
        for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < sortA.zero_after_n; k += gridDim.x * 256u) sortA.zero_after[k] = 0u;
    const unsigned lane = threadIdx.x & 63, wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave id, kept scalar
    // Placement.  Without a cost hint: workgroup b takes tile egs_tile_of_block(b) and wave w quadrant w.  With one (`order`: the
    // words the ordering job made of the hint, backward_prologue.h): the tile and the quadrant -> SIMD assignment come from there, so
    // that CUs and SIMDs carry equal work (a wave reads its SIMD from HW_ID and claims the quadrant through an LDS word, as in the
    // backward).  Placement decides when a quadrant is blended, never what it computes.
    int tile; unsigned q = wv;
    if (order) {
        const uint32_t order_word = order[blockIdx.x];               // 0xffffffff = padding workgroup
        if (order_word == 0xffffffffu) return;
        tile = (int)(order_word & ((order_word & EGS_ORDER_HAS_PERM) ? 0xffffu : 0xffffffffu));
        if (order_word & EGS_ORDER_HAS_PERM) {
            if (threadIdx.x == 0) quad_claimed = 0u;
            __syncthreads();
            uint32_t hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned want = (order_word >> (16 + 2 * ((hw >> 4) & 3u))) & 3u;
            if (lane == 0) {
                uint32_t before = atomicOr(&quad_claimed, 1u << want);
                while (before & (1u << want)) {                      // taken (two waves of the workgroup on one SIMD): any free one
                    want = (unsigned)__builtin_ctz(~before & 0xfu);
                    before = atomicOr(&quad_claimed, 1u << want);
                }
            }
            q = (unsigned)__builtin_amdgcn_readfirstlane((int)want);
        }
    } else {
        tile = egs_tile_of_block(blockIdx.x, n_tiles);
        if (tile < 0) return;
    }
    uint2 sorted_range = make_uint2(0u, 0u);
    if (SORT) {
        tile_sort_body<RA, 4, TS_SMALL_CAP, 0u>(sortA, tile, smem64, reinterpret_cast<uint32_t*>(smem64 + TS_SMALL_CAP),
                                                reinterpret_cast<uint32_t*>(smem64 + TS_SMALL_CAP) + 2 * 512, &sorted_range);
        __syncthreads();                                              // the ids are in point_list; the LDS is the blend's from here on
    }
    float4* my = lds[wv];
    const unsigned my_addr = (unsigned)(uintptr_t)my;             // LDS byte offset of the wave's slice (low half of the generic address)
    const int qx0 = (tile % gx) * EGS_TILE + (int)(q & 1) * 8, qy0 = (tile / gx) * EGS_TILE + (int)(q >> 1) * 8;
    if (qx0 >= W || qy0 >= H) {                                    // quadrant entirely outside the image
        if (lane == 0) { quad_work[tile * 4 + q] = 0; quad_pairs[tile * 4 + q] = 0; quad_pairs[(n_tiles + tile) * 4 + q] = 0; if (cost_hint) cost_hint[tile * 4 + q] = 0; }
        return;
    }
    const int px = qx0 + (int)(lane & 7), py = qy0 + (int)(lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t qx1 = (uint32_t)min(qx0 + 7, W - 1), qy1 = (uint32_t)min(qy0 + 7, H - 1);

    const uint2 range = SORT ? sorted_range : ranges[tile];
    const uint32_t n = range.y - range.x;
    const uint32_t* list = (SORT ? (const uint32_t*)sortA.point_list : point_list) + range.x;      // (SORT: not through the read-only argument -- this workgroup has just written them)

    // Tl: live transmittance, forced to 0 once the pixel has stopped (so later splats add nothing);
    // Tf: the value final_T reports.  Invariant while live: Tl == Tf >= 1e-4.
    float Tl = inside ? 1.f : 0.f, Tf = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dacc = 0.f, Aacc = 0.f;
    uint32_t last = 0;
    uint32_t visits = 0;                                            // wave-uniform: splats blended by this quadrant
    uint32_t pairs = 0;                                             // wave-uniform: (pixel, splat) pairs that contributed (scalar unit only)
    EGS_IF_MEASURE(uint32_t meas = 0; const uint64_t t_start = wall_clock64();)

    // software pipeline: ids two batches ahead, records one batch ahead
    uint32_t id_next = lane < n ? list[lane] : 0u;
    float4 r0, r1, r2;
    egs_load_rec(rec, id_next, lane < n, r0, r1, r2);
    id_next = 64 + lane < n ? list[64 + lane] : 0u;

    bool alive = true;                                             // wave-uniform: some pixel still live
    uint32_t scanned = 0;                                           // wave-uniform: batches of 64 list entries looked at
    for (uint32_t base = 0; alive && base < n; base += 64) {
        scanned++;
        EGS_LRPT(
        // Longest-remaining-work-first, as in the backward (render_bwd.hip) -- but here the work left is not known, so it is estimated
        // once per batch from the quadrant's own history: its least saturated pixel has come ln(Tmax) of the way to ln(1e-4) with the
        // splats blended so far, so about visits * (ln(1e-4) - ln Tmax) / ln Tmax are left, and never more than the rest of the list
        // would give at the same rate.  The wave's issue priority (4 levels) follows that estimate; it decides who issues first,
        // never a result.
        if (base) {
            float tmax = Tl;
            _Pragma("unroll") for (int d = 32; d >= 1; d >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, d, 64));
            const float lnT = __logf(fmaxf(tmax, 1e-30f));
            const float by_decay = lnT < -1e-4f ? (9.2103404f + lnT) / -lnT : 1e9f;
            const float by_list = (float)(n - base) / (float)base;
            const int left = (int)__builtin_amdgcn_readfirstlane((int)fminf((float)visits * fminf(fmaxf(by_decay, 0.f), by_list), 1e6f));
            const int want = left >= EGS_FWD_LRPT3 ? 3 : left >= EGS_FWD_LRPT2 ? 2 : left >= EGS_FWD_LRPT1 ? 1 : 0;
            if (want == 3) __builtin_amdgcn_s_setprio(3); else if (want == 2) __builtin_amdgcn_s_setprio(2); else if (want == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        } else {
            __builtin_amdgcn_s_setprio(3);
        })
        const float4 c0 = r0, c1 = r1, c2 = r2;
        const bool have = base + lane < n;
        // issue the next batch's gathers now
        egs_load_rec(rec, id_next, base + 64 + lane < n, r0, r1, r2);
        id_next = base + 128 + lane < n ? list[base + 128 + lane] : 0u;

        uint64_t mask = __ballot(have && egs_block_hits(c0, c1, c2, (uint32_t)qx0, qx1, (uint32_t)qy0, qy1));
        if (mask == 0ull) continue;
        my[lane * 3 + 0] = c0; my[lane * 3 + 1] = c1; my[lane * 3 + 2] = c2;
        __builtin_amdgcn_wave_barrier();
        // The record of the NEXT splat of the work list is fetched from the LDS slice while the current one is blended (two register
        // sets, the loop unrolled by two so that no copy sits between a fetch and its use): once most waves of a SIMD have left -- all
        // start together and a quadrant takes 10-70 us -- a lone wave would otherwise sit through the LDS round trip at every splat,
        // and the launch ends with exactly those waves.  The reads and their waits are written out (egs_lds_fetch / egs_lds_wait_*):
        // left to the compiler the fetch is either sunk next to its use or followed by a full lgkmcnt(0).  A fetch is issued on every
        // path (entry 0 again when the list is exhausted), so exactly one -- three reads -- is outstanding at each wait.
#define EGS_FWD_BLEND(J, S0, S1, S2) {                                                                                              \
            EGS_FWD_COUNT_VISIT                                                                                                     \
            float G;                                                                                                                \
            const float a = EGS_FWD_ALPHA(S0.x - pxf, S0.y - pyf, S0.z, S0.w, S1.x, S1.y, G);   /* 0 = skipped */                   \
            const float test = fmaf(-a, Tl, Tl);                  /* T (1 - alpha); == Tl when skipped, 0 when stopped */            \
            const bool cont = test >= 0.0001f;                    /* false: this splat stops the pixel (or already stopped) */       \
            const float w = cont ? a * Tl : 0.f;                                                                                    \
            C0 = fmaf(S1.z, w, C0); C1 = fmaf(S1.w, w, C1); C2 = fmaf(S2.x, w, C2);                                                  \
            if (DA) { Dacc = fmaf(S2.y, w, Dacc); Aacc += w; }                                                                      \
            Tf = cont ? test : Tf;                                                                                                  \
            Tl = cont ? test : 0.f;                                                                                                 \
            const bool used = w > 0.f;                                                                                              \
            EGS_FWD_LAST(used, J)                                                                                                   \
            EGS_FWD_COUNT_PAIRS(used)                                                                                               \
            alive = __ballot(cont) != 0ull;                       /* whole quadrant saturated -> leave */                           \
            EGS_FWD_MEAS(w) }
        unsigned ja, jb;
        egs_f4 a0, a1, b0, b1;
        egs_f2 a2, b2;
        ja = (unsigned)__builtin_ctzll(mask); mask &= mask - 1ull;
        egs_lds_fetch(my_addr + ja * 48u, a0, a1, a2);
        for (;;) {
            const bool more_b = mask != 0ull;                     // (wave-uniform)
            jb = more_b ? (unsigned)__builtin_ctzll(mask) : 0u; mask &= mask - 1ull;
            EGS_FWD_PREFETCH(egs_lds_fetch(my_addr + jb * 48u, b0, b1, b2); egs_lds_wait_older(a0, a1, a2);, b0 = a0; b1 = a1; b2 = a2;)
            EGS_FWD_BLEND(ja, a0, a1, a2)
            if (!more_b || !alive) break;
            const bool more_a = mask != 0ull;
            ja = more_a ? (unsigned)__builtin_ctzll(mask) : 0u; mask &= mask - 1ull;
            EGS_FWD_PREFETCH(egs_lds_fetch(my_addr + ja * 48u, a0, a1, a2); egs_lds_wait_older(b0, b1, b2);, )
            EGS_FWD_BLEND(jb, b0, b1, b2)
            if (!more_a || !alive) break;
        }
        egs_lds_wait_all(a0, a1, a2, b0, b1, b2);                  // nothing may still be landing in registers the code below reuses
#undef EGS_FWD_BLEND
        __builtin_amdgcn_wave_barrier();
    }
    const float T = Tf;
    {   // cost of this quadrant's wave in the backward, which replays the same splats up to the deepest pixel's last
        // contributor: ~10 time units per blended splat + ~18 per batch of 64 list entries scanned (measured per-SIMD
        // regression, tools/lane_use.py).  Used only to balance that launch.
        uint32_t wmax = last;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d, 64));
        if (lane == 0) {
            quad_work[tile * 4 + q] = 10u * visits + 18u * ((wmax + 63u) / 64u);
            quad_pairs[tile * 4 + q] = pairs; quad_pairs[(n_tiles + tile) * 4 + q] = visits;     // measurement only (bench.py: Q, visits)
            if (cost_hint) cost_hint[tile * 4 + q] = 10u * visits + EGS_FWD_COST_BATCH * scanned;  // what THIS kernel spent on the quadrant
        }
        EGS_IF_MEASURE(if (lane == 0) quad_work[tile * 4 + q] = meas;)
    }
    if (inside) {
        const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
        final_T[pix] = T; n_contrib[pix] = last;
        out_color[pix] = fmaf(T, bg[0], C0); out_color[HW + pix] = fmaf(T, bg[1], C1);
        out_color[2 * HW + pix] = fmaf(T, bg[2], C2);
        if (DA) { out_depth[pix] = Dacc; out_alpha[pix] = Aacc; }
        EGS_FWD_TIMELINE()
    }
}

}  // namespace

hipError_t egs_launch_render_forward(int W, int H, const float* bg, EgsGeomPtrs g, const uint32_t* point_list,
                                     EgsImgPtrs im, float* out_color, float* out_depth, float* out_alpha, int placed,
                                     const EgsSortArgs* sort, hipStream_t s) {
    const int gx = (W + EGS_TILE - 1) / EGS_TILE, gy = (H + EGS_TILE - 1) / EGS_TILE;
    const int n_tiles = gx * gy;
    if (n_tiles == 0) return hipSuccess;
    EgsSortArgs sa = {};
    const bool do_sort = sort && sort->table_scanned;
    if (do_sort) sa = *sort;
#define EGS_FWD_LAUNCH(DA, SO, RA) hipLaunchKernelGGL((k_render_forward<DA, SO, RA>), dim3(egs_blocks_for_tiles(n_tiles)), dim3(256), 0, s, W, H, gx, n_tiles, \
                       im.ranges, point_list, g.rec, bg, out_color, out_depth, out_alpha, im.final_T, im.n_contrib, im.quad_work, im.quad_pairs, \
                       placed ? im.fwd_order : (const uint32_t*)nullptr, im.fwd_cost, sa)
    const bool da = out_depth && out_alpha;
    if (!do_sort) { if (da) EGS_FWD_LAUNCH(true, false, false); else EGS_FWD_LAUNCH(false, false, false); }
    else if (sa.rank_atomic) { if (da) EGS_FWD_LAUNCH(true, true, true); else EGS_FWD_LAUNCH(false, true, true); }
    else { if (da) EGS_FWD_LAUNCH(true, true, false); else EGS_FWD_LAUNCH(false, true, false); }
#undef EGS_FWD_LAUNCH
    return hipGetLastError();
}
