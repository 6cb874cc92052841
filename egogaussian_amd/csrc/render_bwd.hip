// render_bwd.hip -- back-to-front replay of the compositing and per-splat gradient accumulation (gfx950).
// Replaces the backward `render` stage of the upstream op reached through loss.backward() at
// /root/reference/trainers/train_static.py:110.
//
// Same wave-per-8x8-quadrant mapping as render_fwd.hip (no workgroup barriers; bounding-box ballot as the
// work list; wave-private LDS slice read with a uniform address).  What is specific to the backward:
//   * the replay starts at the wave's maximum n_contrib, not at the end of the tile list, so saturated
//     tiles do not walk the occluded tail;
//   * the colour / depth / alpha channels share ONE running accumulator: with u_j = c_j . dL/dC + d_j dL/dD
//     + dL/dA the published per-channel recurrences collapse to U <- a_last u_last + (1 - a_last) U and
//     dL/dalpha_j = T_j (u_j - U_j) - T_final/(1-a_j) bg . dL/dC   (algebraically identical);
//   * the 10 per-splat partial sums (five moments of gd = dL/dalpha * G, opacity, rgb, depth) are reduced across the 64 lanes in three
//     stages priced with tools/ubench/xlane_rate.hip (cycles per SIMD at 8 waves: plain VALU 2.4, DPP add 6.8,
//     v_permlane{16,32}_swap 11.5, v_readlane 8, ds_bpermute 22):
//       1. v_permlane32_swap "transpose-and-add" folds the ten registers into five (lanes 0-31: even value, 32-63: odd);
//       2. the five registers go through a wave-private 1.25 KiB LDS slice (5 ds_write_b32), and lane 4v+g reads eight
//          consecutive partials of value v (2 ds_read_b128) and adds them -- the LDS pipe is otherwise nearly idle here;
//       3. two quad_perm DPP adds finish the sum in lanes 0, 4, ..., 36,
//     about 100 VALU-cycles instead of 196 for swaps + DPP rows alone (and 408 for ten DPP butterflies);
//   * those ten lanes issue ONE global_atomic_add_f32 instruction into the Gaussian's 48-byte accumulator line
//     (egs_common.h) instead of ten.
#include "egs_common.h"
#include "blend_common.h"
#include <algorithm>

namespace {

typedef unsigned uint2v __attribute__((ext_vector_type(2)));

// a' + b' where (a', b') = halves-swapped (a, b): lanes 0-31 <- a[l] + a[l+32], lanes 32-63 <- b[l-32] + b[l].
__device__ __forceinline__ float fold32(float a, float b) {
    const uint2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// rows (16 lanes): out = [a.r0+a.r1, b.r0+b.r1, a.r2+a.r3, b.r2+b.r3]
__device__ __forceinline__ float fold16(float a, float b) {
    const uint2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, true));
}
// all 16 lanes of every row end up holding that row's sum
__device__ __forceinline__ float row_sum(float v) {
    v = dpp_add<0xB1>(v);       // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);       // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);      // row_half_mirror
    v = dpp_add<0x140>(v);      // row_mirror
    return v;
}

// Work-aware placement of tiles for the backward blend.  All workgroups of that launch are resident at once (8 per CU),
// so its duration is the busiest CU's total; with tiles dealt in index order the busiest CU carries 1.2-1.3x the mean.
// Workgroup b runs on XCD b % 8 and, inside the XCD, on CU (b / 8) % 32 (tools/ubench/dispatch_map.hip; used for speed
// only -- any placement gives the same results).  One workgroup per XCD band ranks the band's tiles by the cost the forward
// recorded -- a counting sort over 1024 cost levels, O(tiles): the exact O(tiles^2) rank it replaces took 45 us at 1920x1080 --
// and deals them to the 32 CUs in snake order (rank r -> round r/32, CU r%32 or 31 - r%32).
// The same launch clears the gradient accumulator (workgroups 8..): two short kernels cost more than one.
// Tried and rejected (round 1, config C, per-wave timelines from tools/lane_use.py): (a) persistent waves pulling
// (tile, quadrant) tasks, sorted by cost, from one queue per XCD: perfectly balanced and 2.2x slower -- the four waves
// of a workgroup then work on unrelated tiles and stop sharing list and record lines in the CU's L1; (b) dealing each
// tile's quadrants to the CU's SIMDs by cost (a wave reads its SIMD from HW_ID): per-SIMD spread +-16% -> +-10%, but the
// CU-level spread (-12%/+8% of blended splats) then bounds the launch and the longer prologue cancels the 2 us gained.
// (c) running this prologue on a second stream right after the forward, so that it overlaps the loss kernels (fork / join
// captured into the hipGraph): the step got 3 % SLOWER -- the graph's cross-stream dependencies cost more than the 11 us hidden.
#define ORDER_MAX_BAND 8192
#define ORDER_LEVELS 1024
#define ORDER_BALANCE_MAX 256               // band size up to which every workgroup of the launch is resident at once (8 per CU x 32 CUs)
#define EGS_ORDER_HAS_PERM 0x01000000u      // tile_order word: bits 0-15 tile, 16-23 quadrant for the wave on SIMD 0..3 (two bits each), 24 = those are set
__global__ __launch_bounds__(1024) void k_backward_prologue(int n_tiles, const uint32_t* __restrict__ quad_work,
                                                             uint32_t* __restrict__ tile_order, float4* __restrict__ acc4, size_t n4,
                                                             int has_tick, EgsAdamTick tick) {
    if (blockIdx.x >= EGS_XCDS) {
        // an optimizer fused into this backward (egs_backward_adam): its once-per-step bookkeeping rides here, two launches ahead of
        // its reader, in a workgroup of its own (the first after the ordering ones) so that no zeroing waits for the pow() calls
        const unsigned first = EGS_XCDS + (has_tick ? 1u : 0u);
        if (blockIdx.x < first) { if (threadIdx.x < 64) egs_adam_tick(tick, threadIdx.x); return; }
        const size_t stride = (size_t)(gridDim.x - first) * 1024;
        for (size_t i = (size_t)(blockIdx.x - first) * 1024 + threadIdx.x; i < n4; i += stride) acc4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    // Order of one band: a counting sort on the cost quantised to ORDER_LEVELS levels (descending).  Ties land in arrival order --
    // the order only decides which workgroup blends which tile, never a result.
    __shared__ uint32_t work[ORDER_MAX_BAND];
    __shared__ uint32_t level_base[ORDER_LEVELS], level_fill[ORDER_LEVELS], wsum[16], wmax_s;
    __shared__ uint4 quad_cost[ORDER_BALANCE_MAX];                   // the four quadrant costs of every tile of a small band
    __shared__ uint16_t sorted_tile[ORDER_BALANCE_MAX];
    const int per = egs_tiles_per_xcd(n_tiles), x = blockIdx.x;
    const int t0 = x * per, n = max(0, min(per, n_tiles - t0));
    const int slots = ((per + 31) / 32) * 32;
    if (per > ORDER_MAX_BAND) {                                      // very large images: keep index order
        for (int sl = threadIdx.x; sl < per; sl += blockDim.x) tile_order[8 * sl + x] = sl < n ? (uint32_t)(t0 + sl) : 0xffffffffu;
        return;
    }
    if (threadIdx.x == 0) wmax_s = 1u;
    level_fill[threadIdx.x] = 0u;                                    // (ORDER_LEVELS == blockDim.x)
    __syncthreads();
    uint32_t mx = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint4 w4 = *reinterpret_cast<const uint4*>(quad_work + 4 * (size_t)(t0 + i));
        const uint32_t wsum_i = w4.x + w4.y + w4.z + w4.w;
        work[i] = wsum_i; mx = max(mx, wsum_i);
        if (per <= ORDER_BALANCE_MAX) quad_cost[i] = w4;
    }
    if (per > ORDER_BALANCE_MAX)
        for (int sl = threadIdx.x; sl < per; sl += blockDim.x) tile_order[8 * sl + x] = 0xffffffffu;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&wmax_s, mx);
    __syncthreads();
    const float to_level = (float)(ORDER_LEVELS - 1) / (float)wmax_s;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t lv = (uint32_t)(ORDER_LEVELS - 1) - min((uint32_t)((float)work[i] * to_level), (uint32_t)(ORDER_LEVELS - 1));
        work[i] = lv;                                                // 0 = most expensive
        atomicAdd(&level_fill[lv], 1u);
    }
    __syncthreads();
    {   // exclusive scan of the level counts (one level per thread)
        const uint32_t c = level_fill[threadIdx.x];
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if ((int)(threadIdx.x & 63) >= d) incl += o; }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t base = incl - c;
        for (unsigned k = 0; k < (threadIdx.x >> 6); k++) base += wsum[k];
        level_base[threadIdx.x] = base;
        level_fill[threadIdx.x] = 0u;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t lv = work[i];
        const int rank = (int)(level_base[lv] + atomicAdd(&level_fill[lv], 1u));
        if (per <= ORDER_BALANCE_MAX) { sorted_tile[rank] = (uint16_t)i; continue; }      // band-local index, most expensive first
        const int round = rank / 32, pos = rank % 32;
        int slot = round * 32 + ((round & 1) ? 31 - pos : pos);
        if (slot >= per) slot = round * 32 + pos;                    // last, partial round: no room to mirror
        if (slot >= per) slot = per - 1 - (slots - 1 - slot);        // (cannot happen when per is a multiple of 32)
        tile_order[8 * slot + x] = (uint32_t)(t0 + i);
    }
    if (per > ORDER_BALANCE_MAX) return;
    // Small band (every workgroup of the launch resident at once: the workgroup in slot 32 k + c of the band runs on CU c).  With the
    // waves' issue priority following the work they have left (k_render_backward), a SIMD ends when its total work is done (measured
    // correlation of blended splats per SIMD and SIMD end: 0.99), so what is left to balance is that total:
    //   * CUs: dealt in sorted rounds -- in every round the CU that carries the least so far takes the round's most expensive tile;
    //   * SIMDs: the four quadrant-waves of a workgroup land on the CU's four SIMDs, and which wave takes which quadrant is free, so
    //     the tile's most expensive quadrant goes to the SIMD of that CU that carries the least.  The choice rides in bits 16-23 of
    //     the tile_order word (two bits per SIMD = the quadrant its wave should take); the wave reads its SIMD from HW_ID.
    // One wave does it (lane c = CU c), from the costs cached in LDS.
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int c = (int)threadIdx.x;
    uint32_t load_cu = 0, ls0 = 0, ls1 = 0, ls2 = 0, ls3 = 0;
    const int rounds = (per + 31) / 32;
    for (int k = 0; k < rounds; k++) {
        const bool has_slot = c < 32 && 32 * k + c < per;
        const int m = min(32, n - 32 * k);                           // tiles of this round (uniform)
        // position of this CU among the CUs with a slot, lightest first: 32 v_readlane + compare on unique keys (a loop of __shfl
        // = ds_bpermute, each waited for, made this launch 13 us longer)
        const uint32_t key = has_slot ? (min(load_cu, 0x03ffffffu) << 5) | (uint32_t)c : 0xffffffffu;
        int rank = 0;
#pragma unroll
        for (int j = 0; j < 32; j++) rank += (uint32_t)__builtin_amdgcn_readlane((int)key, j) < key ? 1 : 0;
        uint32_t word = 0xffffffffu;
        if (has_slot && rank < m) {
            const int i = (int)sorted_tile[32 * k + rank];
            const uint4 w4 = quad_cost[i];
            // two four-element sorting networks on (value << 2 | index) keys: registers only (indexing a local array by a run-time
            // value would go through scratch memory, ~1 us per access)
            uint32_t a0 = (min(w4.x, 0x3fffffffu) << 2) | 0u, a1 = (min(w4.y, 0x3fffffffu) << 2) | 1u,
                     a2 = (min(w4.z, 0x3fffffffu) << 2) | 2u, a3 = (min(w4.w, 0x3fffffffu) << 2) | 3u;      // quadrants, to be sorted descending
            uint32_t b0 = (min(ls0, 0x3fffffffu) << 2) | 0u, b1 = (min(ls1, 0x3fffffffu) << 2) | 1u,
                     b2 = (min(ls2, 0x3fffffffu) << 2) | 2u, b3 = (min(ls3, 0x3fffffffu) << 2) | 3u;        // SIMDs, ascending
#define EGS_CS(lo, hi) { const uint32_t t_ = min(lo, hi); hi = max(lo, hi); lo = t_; }
            EGS_CS(a0, a1) EGS_CS(a2, a3) EGS_CS(a0, a2) EGS_CS(a1, a3) EGS_CS(a1, a2)          // a0 <= a1 <= a2 <= a3
            EGS_CS(b0, b1) EGS_CS(b2, b3) EGS_CS(b0, b2) EGS_CS(b1, b3) EGS_CS(b1, b2)          // b0 <= b1 <= b2 <= b3
#undef EGS_CS
            // the most expensive quadrant (a3) goes to the least loaded SIMD (b0), and so on
            const uint32_t qd[4] = { a3 & 3u, a2 & 3u, a1 & 3u, a0 & 3u }, cd[4] = { a3 >> 2, a2 >> 2, a1 >> 2, a0 >> 2 };
            const uint32_t sd[4] = { b0 & 3u, b1 & 3u, b2 & 3u, b3 & 3u };
            uint32_t perm = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {                            // (r is a compile-time index after unrolling)
                perm |= qd[r] << (2u * sd[r]);
                ls0 += sd[r] == 0u ? cd[r] : 0u; ls1 += sd[r] == 1u ? cd[r] : 0u; ls2 += sd[r] == 2u ? cd[r] : 0u; ls3 += sd[r] == 3u ? cd[r] : 0u;
            }
            load_cu += w4.x + w4.y + w4.z + w4.w;
            word = (uint32_t)(t0 + i) | (perm << 16) | EGS_ORDER_HAS_PERM;
        }
        if (has_slot) tile_order[8 * (32 * k + c) + x] = word;
    }
}

// 8 waves per SIMD (64 VGPRs, one spilled): every wave of a 960x540 frame is resident from the start (-2% vs 70 VGPRs / 7 waves)
// HAS_DA: upstream gradients on the depth and / or alpha outputs exist (the training loss uses colour only: three FMAs and a
// multiply less per (wave, splat) visit of a kernel that is 86 % VALU-busy).
template <bool HAS_DA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_render_backward(
    int W, int H, int gx, int n_tiles, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ rec, const float* __restrict__ bg, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
    const float* __restrict__ dL_dalpha, const uint32_t* __restrict__ tile_order, float* __restrict__ grad_acc,
    const uint32_t* __restrict__ quad_visits) {
    __shared__ float4 lds[4][64 * EGS_SPLAT_REC_F4];
    __shared__ __attribute__((aligned(16))) float red[4][5 * 64];
    __shared__ uint32_t quad_claimed;
    const uint32_t order_word = tile_order[blockIdx.x];              // 0xffffffff = padding workgroup
    if (order_word == 0xffffffffu) return;
    const int tile = (int)(order_word & ((order_word & EGS_ORDER_HAS_PERM) ? 0xffffu : 0xffffffffu));
    const unsigned lane = threadIdx.x & 63, wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave id, kept scalar
    // Which quadrant this wave blends: the one the prologue chose for the SIMD it happens to run on (HW_ID bits 4-5), claimed
    // through an LDS word so that the four waves take four different quadrants whatever the placement was (balance only -- any
    // assignment gives the same sums).
    unsigned q = wv;
    if (order_word & EGS_ORDER_HAS_PERM) {
        if (threadIdx.x == 0) quad_claimed = 0u;
        __syncthreads();
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned want = (order_word >> (16 + 2 * ((hw >> 4) & 3u))) & 3u;
        if (lane == 0) {
            uint32_t before = atomicOr(&quad_claimed, 1u << want);
            while (before & (1u << want)) {                          // taken (two waves of the workgroup on one SIMD): any free one
                want = (unsigned)__builtin_ctz(~before & 0xfu);
                before = atomicOr(&quad_claimed, 1u << want);
            }
        }
        q = (unsigned)__builtin_amdgcn_readfirstlane((int)want);
    }
    float4* my = lds[wv];
    float* myred = red[wv];
#if defined(EGS_MEASURE) && EGS_MEASURE == 4      // instrumentation build (tools/lane_use.py): per-wave timeline of the backward
    const uint64_t t_start = wall_clock64();
    uint32_t meas = 0;
#endif
    const int qx0 = (tile % gx) * EGS_TILE + (int)(q & 1) * 8, qy0 = (tile / gx) * EGS_TILE + (int)(q >> 1) * 8;
    if (qx0 >= W || qy0 >= H) return;
    const int px = qx0 + (int)(lane & 7), py = qy0 + (int)(lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t qx1 = (uint32_t)min(qx0 + 7, W - 1), qy1 = (uint32_t)min(qy0 + 7, H - 1);

    const uint2 range = ranges[tile];
    const uint32_t* list = point_list + range.x;

    float T_final = 0.f, g_r = 0.f, g_g = 0.f, g_b = 0.f, g_d = 0.f, g_a = 0.f;
    uint32_t last = 0;
    if (inside) {
        const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
        T_final = final_T[pix]; last = n_contrib[pix];
        g_r = dL_dcolor[pix]; g_g = dL_dcolor[HW + pix]; g_b = dL_dcolor[2 * HW + pix];
        if (HAS_DA && dL_ddepth) g_d = dL_ddepth[pix];
        if (HAS_DA && dL_dalpha) g_a = dL_dalpha[pix];
    }
    const float bg_term = -T_final * (bg[0] * g_r + bg[1] * g_g + bg[2] * g_b);
    uint32_t wmax = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d, 64));
    if (wmax == 0) return;

    // stage 2/3 of the reduction: lane 4v+g (v < 10) sums partials [8g, 8g+8) of value v; lane 4v publishes it.
    // After the permlane32 fold, value v lives in register row v/2, lanes (v%2)*32 .. +31.
    const unsigned rv = lane >> 2, rg = lane & 3;
    const float4* red_src = reinterpret_cast<const float4*>(myred + (rv < 10 ? (rv >> 1) * 64 + (rv & 1) * 32 + rg * 8 : 0));
    const int slot = (rg == 0 && rv < 10) ? (int)rv : -1;

    // S = the blended colour-gradient term of everything BEHIND the splat being processed (U of the header), kept "ready for the
    // next contributor": after a splat with (a, u) it becomes a u + (1 - a) S -- one state word and one select instead of three
    // (same values and roundings as U_next = fma(a_last, u_last - U, U); the select keeps a skipped splat's colour, NaN included,
    // out of the pixel's chain).
    float T = T_final, S = 0.f;
#if defined(EGS_ABL) && EGS_ABL == 5
    float abl_sink = 0.f;
#endif

    // Longest-remaining-work-first.  A SIMD issues its oldest wave first, and the tiles are dispatched most expensive first, so the
    // light, late waves used to wait for the others and then run down alone -- the last quarter of the launch had fewer than three
    // waves per SIMD (profiles/r2_blend_timelines.md).  The forward counted the splats every quadrant blends; this wave's issue
    // priority (s_setprio, 4 levels) follows the number it still has to replay, so the waves of a SIMD reach the end together.
    // Placement-like: it decides who issues first, never a result.  -DEGS_NO_LRPT builds without it.
#ifndef EGS_NO_LRPT
    int remaining = (int)quad_visits[tile * 4 + q];
    int prio_now = -1;
#endif
    const int nb = (int)((wmax + 63) / 64);
    int b = nb - 1;
    uint32_t id_next = (uint32_t)b * 64 + lane < wmax ? list[(uint32_t)b * 64 + lane] : 0u;
    float4 r0, r1, r2;
    egs_load_rec(rec, id_next, (uint32_t)b * 64 + lane < wmax, r0, r1, r2);
    uint32_t id_cur = id_next;
    id_next = b >= 1 ? list[(uint32_t)(b - 1) * 64 + lane] : 0u;

    for (; b >= 0; b--) {
        const uint32_t base = (uint32_t)b * 64;
        const float4 c0 = r0, c1 = r1, c2 = r2;
        const uint32_t my_id = id_cur;
        const bool have = base + lane < wmax;
        egs_load_rec(rec, id_next, b >= 1, r0, r1, r2);
        id_cur = id_next;
        id_next = b >= 2 ? list[(uint32_t)(b - 2) * 64 + lane] : 0u;

        uint64_t mask = __ballot(have && egs_block_hits(c0, c1, c2, (uint32_t)qx0, qx1, (uint32_t)qy0, qy1));
        if (mask == 0ull) continue;
        my[lane * 3 + 0] = c0; my[lane * 3 + 1] = c1; my[lane * 3 + 2] = c2;
        __builtin_amdgcn_wave_barrier();
        const uint32_t lastb = last > base ? last - base : 0u;       // this pixel uses entries j < lastb of the batch
        while (mask) {
            const int j = 63 - __builtin_clzll(mask);
            mask &= ~(1ull << j);
#ifndef EGS_NO_LRPT
            {
                const int want = remaining >= 64 ? 3 : remaining >= 24 ? 2 : remaining >= 8 ? 1 : 0;       // (thresholds swept at config C)
                if (want != prio_now) {
                    prio_now = want;
                    if (want == 3) __builtin_amdgcn_s_setprio(3); else if (want == 2) __builtin_amdgcn_s_setprio(2); else if (want == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                }
                remaining--;
            }
#endif
            const float4 s0 = my[j * 3 + 0], s1 = my[j * 3 + 1];
            const float2 s2 = *reinterpret_cast<const float2*>(&my[j * 3 + 2]);
            const float dx = s0.x - pxf, dy = s0.y - pyf;
            // log2 falloff with its intermediates kept: t = qa dx + qb dy, n = qc dy   (same arithmetic as egs_alpha)
            const float m = __fmul_rn(s0.z, dx);
            const float t = __fmaf_rn(s0.w, dy, m);
            const float nn = __fmul_rn(s1.x, dy);
            const float p = __fmaf_rn(nn, dy, __fmul_rn(t, dx));
            const float G = __builtin_amdgcn_exp2f(p);
            float a = fminf(0.99f, __fmul_rn(s1.y, G));
            a = p > 0.f ? 0.f : a;
            a = a < (1.0f / 255.0f) ? 0.f : a;
            a = ((uint32_t)j < lastb) ? a : 0.f;                        // 0 = this pixel does not use the splat
            const bool contrib = a > 0.f;
            if (__ballot(contrib) == 0ull) continue;
#if defined(EGS_MEASURE) && EGS_MEASURE == 4
            meas++;
#endif
            const float rcp = __builtin_amdgcn_rcpf(1.f - a);
            const float Tn = T * rcp;                                   // transmittance in front of this splat
            const float w = a * Tn;
            const float u = HAS_DA ? fmaf(s1.z, g_r, fmaf(s1.w, g_g, fmaf(s2.x, g_b, fmaf(s2.y, g_d, g_a))))
                                   : fmaf(s1.z, g_r, fmaf(s1.w, g_g, s2.x * g_b));
            float dLda = fmaf(bg_term, rcp, (u - S) * Tn);
            dLda = contrib ? dLda : 0.f;
            T = Tn; S = contrib ? fmaf(a, u - S, S) : S;

            // Per-splat sums published to the accumulator line are MOMENTS of gd = dL/dalpha * G over the pixels:
            //   v0 = sum gd dx, v1 = sum gd dy, v2 = sum gd dx^2, v3 = sum gd dx dy, v4 = sum gd dy^2,  v5 = sum gd
            // k_preprocess_backward turns them into d/d mean2D and d/d conic with the Gaussian's own opacity and conic
            // (they are linear in these moments), which keeps that algebra out of the per-pixel loop.
            const float gd = G * dLda;                                  // d/d opacity
            const float v0 = gd * dx, v1 = gd * dy;
            const float v2 = v0 * dx, v3 = v0 * dy, v4 = v1 * dy;
            const float v5 = gd;
            const float v6 = w * g_r, v7 = w * g_g, v8 = w * g_b, v9 = HAS_DA ? w * g_d : 0.f;
            (void)t; (void)m; (void)nn;

#if defined(EGS_ABL) && EGS_ABL == 5          /* ablation build (timing only): no cross-lane reduction, no atomic -- what the pair arithmetic alone costs */
            abl_sink += ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7)) + (v8 + v9);
            continue;
#endif
            // 64-lane sums of v0..v9 (see the header): swap-fold, LDS regroup, quad DPP
            myred[0 * 64 + lane] = fold32(v0, v1); myred[1 * 64 + lane] = fold32(v2, v3); myred[2 * 64 + lane] = fold32(v4, v5);
            myred[3 * 64 + lane] = fold32(v6, v7); myred[4 * 64 + lane] = fold32(v8, v9);
            const float4 pa = red_src[0], pb = red_src[1];
            float out = ((pa.x + pa.y) + (pa.z + pa.w)) + ((pb.x + pb.y) + (pb.z + pb.w));
            out = dpp_add<0xB1>(out);       // quad_perm [1,0,3,2]
            out = dpp_add<0x4E>(out);       // quad_perm [2,3,0,1]
            const uint32_t gid = (uint32_t)__builtin_amdgcn_readlane((int)my_id, j);
            if (slot >= 0) unsafeAtomicAdd(grad_acc + (gid * (uint32_t)EGS_GRAD_STRIDE + (uint32_t)slot), out);   // (48 P < 2^32 bytes: P < 89 M)
        }
        __builtin_amdgcn_wave_barrier();
    }
#if defined(EGS_ABL) && EGS_ABL == 5
    if (abl_sink == 12345.678f) grad_acc[lane] = abl_sink;
#endif
#if defined(EGS_MEASURE) && EGS_MEASURE == 4
    if (inside) {                                                    // n_contrib was consumed above: reuse it as the log
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        uint32_t* log = const_cast<uint32_t*>(n_contrib) + (size_t)py * W + px;
        if (lane == 0) *log = (uint32_t)t_start;
        if (lane == 1) *log = (uint32_t)wall_clock64();
        if (lane == 2) *log = ((xcc & 0xfu) << 16) | (hw & 0xffffu);
        if (lane == 3) *log = range.y - range.x;
        if (lane == 4) *log = meas;
        if (lane == 5) *log = wmax;
    }
#endif
}

}  // namespace

hipError_t egs_launch_render_backward(int W, int H, const float* bg, EgsGeomPtrs g, const uint32_t* point_list,
                                      EgsImgPtrs im, const float* dL_dcolor, const float* dL_ddepth,
                                      const float* dL_dalpha, float* grad_acc, size_t acc_floats, const EgsAdamTick* tick, hipStream_t s) {
    const int gx = (W + EGS_TILE - 1) / EGS_TILE, gy = (H + EGS_TILE - 1) / EGS_TILE;
    const int n_tiles = gx * gy;
    if (n_tiles == 0) {
        if (tick) { hipError_t e = egs_launch_adam_tick(*tick, s); if (e != hipSuccess) return e; }
        return egs_launch_zero_f4((float4*)grad_acc, acc_floats / 4, s);
    }
    const size_t n4 = acc_floats / 4;
    const unsigned zero_blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((n4 + 1023) / 1024, 1024));
    EgsAdamTick no_tick = {};
    hipLaunchKernelGGL(k_backward_prologue, dim3(EGS_XCDS + (tick ? 1 : 0) + zero_blocks), dim3(1024), 0, s, n_tiles, im.quad_work, im.tile_order,
                       (float4*)grad_acc, n4, tick ? 1 : 0, tick ? *tick : no_tick);
    if (dL_ddepth || dL_dalpha)
        hipLaunchKernelGGL(k_render_backward<true>, dim3(egs_blocks_for_tiles(n_tiles)), dim3(256), 0, s, W, H, gx, n_tiles,
                           im.ranges, point_list, g.rec, bg, im.final_T, im.n_contrib, dL_dcolor, dL_ddepth, dL_dalpha,
                           im.tile_order, grad_acc, im.quad_pairs + (size_t)4 * n_tiles);
    else
        hipLaunchKernelGGL(k_render_backward<false>, dim3(egs_blocks_for_tiles(n_tiles)), dim3(256), 0, s, W, H, gx, n_tiles,
                           im.ranges, point_list, g.rec, bg, im.final_T, im.n_contrib, dL_dcolor, dL_ddepth, dL_dalpha,
                           im.tile_order, grad_acc, im.quad_pairs + (size_t)4 * n_tiles);
    return hipGetLastError();
}
