// backward_prologue.h -- what has to happen between the forward blend and the backward blend of one frame, as a set of
// workgroup-sized JOBS that any launch can carry: the stand-alone k_backward_prologue (render_bwd.hip) or, as a side job of
// a launch that is running anyway at that point of a training step, the image loss's backward (loss.hip; egs_l1_ssim_backward_ex).
//   job 0 .. 7        order the tiles of XCD band x for the backward blend (below)
//   job 8 (optional)  the once-per-step bookkeeping of an optimizer fused into the backward (egs_adam_tick)
//   the rest          clear the gradient accumulator (grid-stride over the remaining jobs)
#pragma once
#include "egs_common.h"
#include "blend_common.h"

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

struct EgsPrologueArgs {
    int n_tiles; const uint32_t* quad_work; uint32_t* tile_order; float4* acc4; size_t n4; int has_tick; EgsAdamTick tick;
    // hot replica lines (egs_common.h): n4 covers the regular lines AND the replicas when block_hot == NULL; else the regular lines only
    // and of the replicas those in use are cleared: block_hot[b] lines of each of the EGS_HOT_REPLICAS copies of workgroup b's budget
    const uint32_t* block_hot; uint32_t hot_blocks, hot_slots; float* hot_base;
};
// Fills the accumulator part of the arguments for a model of P Gaussians whose backward scratch starts at `scratch`.
static inline void egs_prologue_acc(EgsPrologueArgs& a, float* scratch, size_t P, const uint32_t* block_hot) {
    a.acc4 = (float4*)scratch; a.block_hot = block_hot;
    a.n4 = (block_hot ? P * EGS_GRAD_STRIDE : egs_acc_floats(P)) / 4;
    a.hot_blocks = (uint32_t)((P + 255) / 256); a.hot_slots = (uint32_t)egs_hot_slots(P); a.hot_base = scratch + P * EGS_GRAD_STRIDE;
}
struct EgsOrderLds {                        // 12.6 KiB
    uint32_t level_base[ORDER_LEVELS], level_fill[ORDER_LEVELS], wsum[16], wmax;
    uint4 quad_cost[ORDER_BALANCE_MAX];     // the four quadrant costs of every tile of a small band
    uint16_t sorted_tile[ORDER_BALANCE_MAX];
};
// number of jobs a launch with NT threads per workgroup should carry
// `max_zero_jobs` > 0: at most that many zeroing workgroups (each then strides over a larger share) -- for a carrier whose own
// workgroups should all be resident from the start (loss.hip)
static inline unsigned egs_prologue_jobs(size_t n4, int has_tick, int NT, unsigned max_zero_jobs = 0) {
    const size_t per_block = (size_t)NT * 8;                         // ~8 float4 stores per thread, at most 1024 zeroing workgroups of 1024 threads' worth
    size_t z = (n4 + per_block - 1) / per_block;
    const size_t zmax = max_zero_jobs ? (size_t)max_zero_jobs : (size_t)1024 * 1024 / NT;
    if (z > zmax) z = zmax;
    if (z < 1) z = 1;
    return (unsigned)(EGS_XCDS + (has_tick ? 1 : 0) + z);
}

// Order of one band: a counting sort on the cost quantised to ORDER_LEVELS levels (descending).  Ties land in arrival order --
// the order only decides which workgroup blends which tile, never a result.  NT threads (a multiple of 64, at most 1024).
template <int NT>
__device__ __forceinline__ void egs_order_band(const EgsPrologueArgs& a, const int x, EgsOrderLds& L) {
    const uint32_t* __restrict__ quad_work = a.quad_work; uint32_t* __restrict__ tile_order = a.tile_order;
    const int n_tiles = a.n_tiles, tid = (int)threadIdx.x;
    const int per = egs_tiles_per_xcd(n_tiles);                      // slots of a band; its tiles: egs_band_tile(x, i), i < n
    const int n = egs_band_count(x, n_tiles);
    const int slots = ((per + 31) / 32) * 32;
    if (per > ORDER_MAX_BAND) {                                      // very large images: keep index order
        for (int sl = tid; sl < per; sl += NT) tile_order[8 * sl + x] = sl < n ? (uint32_t)egs_band_tile(x, sl, n_tiles) : 0xffffffffu;
        return;
    }
    const bool small = per <= ORDER_BALANCE_MAX;
    if (tid == 0) L.wmax = 1u;
    for (int l = tid; l < ORDER_LEVELS; l += NT) L.level_fill[l] = 0u;
    __syncthreads();
    uint32_t mx = 0;
    for (int i = tid; i < n; i += NT) {
        const uint4 w4 = *reinterpret_cast<const uint4*>(quad_work + 4 * (size_t)egs_band_tile(x, i, n_tiles));
        mx = max(mx, w4.x + w4.y + w4.z + w4.w);
        if (small) L.quad_cost[i] = w4;
    }
    if (!small)
        for (int sl = tid; sl < per; sl += NT) tile_order[8 * sl + x] = 0xffffffffu;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    if ((tid & 63) == 0) atomicMax(&L.wmax, mx);
    __syncthreads();
    const float to_level = (float)(ORDER_LEVELS - 1) / (float)L.wmax;
    // level of tile i, 0 = most expensive (evaluated twice per tile, from the same words: the band's costs stay in L2 / LDS)
    auto level_of = [&](int i) -> uint32_t {
        const uint4 w4 = small ? L.quad_cost[i] : *reinterpret_cast<const uint4*>(quad_work + 4 * (size_t)egs_band_tile(x, i, n_tiles));
        return (uint32_t)(ORDER_LEVELS - 1) - min((uint32_t)((float)(w4.x + w4.y + w4.z + w4.w) * to_level), (uint32_t)(ORDER_LEVELS - 1));
    };
    for (int i = tid; i < n; i += NT) atomicAdd(&L.level_fill[level_of(i)], 1u);
    __syncthreads();
    {   // exclusive scan of the level counts: ORDER_LEVELS / NT consecutive levels per thread
        constexpr int LP = ORDER_LEVELS / NT;
        uint32_t c[LP], sum = 0;
#pragma unroll
        for (int k = 0; k < LP; k++) { c[k] = L.level_fill[tid * LP + k]; sum += c[k]; }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if ((tid & 63) >= d) incl += o; }
        if ((tid & 63) == 63) L.wsum[tid >> 6] = incl;
        __syncthreads();
        uint32_t base = incl - sum;
        for (int k = 0; k < (tid >> 6); k++) base += L.wsum[k];
#pragma unroll
        for (int k = 0; k < LP; k++) { L.level_base[tid * LP + k] = base; base += c[k]; L.level_fill[tid * LP + k] = 0u; }
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
        const uint32_t lv = level_of(i);
        const int rank = (int)(L.level_base[lv] + atomicAdd(&L.level_fill[lv], 1u));
        if (small) { L.sorted_tile[rank] = (uint16_t)i; continue; }  // band-local index, most expensive first
        const int round = rank / 32, pos = rank % 32;
        int slot = round * 32 + ((round & 1) ? 31 - pos : pos);
        if (slot >= per) slot = round * 32 + pos;                    // last, partial round: no room to mirror
        if (slot >= per) slot = per - 1 - (slots - 1 - slot);        // (cannot happen when per is a multiple of 32)
        tile_order[8 * slot + x] = (uint32_t)egs_band_tile(x, i, n_tiles);
    }
    if (!small) return;
    // Small band (every workgroup of the launch resident at once: the workgroup in slot 32 k + c of the band runs on CU c).  With the
    // waves' issue priority following the work they have left (k_render_backward), a SIMD ends when its total work is done (measured
    // correlation of blended splats per SIMD and SIMD end: 0.99), so what is left to balance is that total:
    //   * CUs: dealt in sorted rounds -- in every round the CU that carries the least so far takes the round's most expensive tile;
    //   * SIMDs: the four quadrant-waves of a workgroup land on the CU's four SIMDs, and which wave takes which quadrant is free, so
    //     the tile's most expensive quadrant goes to the SIMD of that CU that carries the least.  The choice rides in bits 16-23 of
    //     the tile_order word (two bits per SIMD = the quadrant its wave should take); the wave reads its SIMD from HW_ID.
    // One wave does it (lane c = CU c), from the costs cached in LDS.
    __syncthreads();
    if (tid >= 64) return;
    const int c = tid;
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
            const int i = (int)L.sorted_tile[32 * k + rank];
            const uint4 w4 = L.quad_cost[i];
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
            word = (uint32_t)egs_band_tile(x, i, n_tiles) | (perm << 16) | EGS_ORDER_HAS_PERM;
        }
        if (has_slot) tile_order[8 * (32 * k + c) + x] = word;
    }
}

template <int NT>
__device__ __forceinline__ void egs_prologue_job(const EgsPrologueArgs& a, const unsigned job, const unsigned n_jobs, EgsOrderLds& L) {
#if defined(EGS_ABL_SIDE) && EGS_ABL_SIDE == 2      // timing ablations of the carried jobs (tools/loss_side_time.py): 2 = no ordering, 1 = no zeroing
    if (job < EGS_XCDS) return;
#endif
    if (job < EGS_XCDS) { egs_order_band<NT>(a, (int)job, L); return; }
    // the optimizer's bookkeeping has a workgroup of its own (the first after the ordering ones) so that no zeroing waits for its pow() calls
    const unsigned first = EGS_XCDS + (a.has_tick ? 1u : 0u);
    if (job < first) { if (threadIdx.x < 64) egs_adam_tick(a.tick, threadIdx.x); return; }
    const size_t stride = (size_t)(n_jobs - first) * NT;
#if defined(EGS_ABL_SIDE) && EGS_ABL_SIDE == 1
    return;
#endif
    for (size_t i = (size_t)(job - first) * NT + threadIdx.x; i < a.n4; i += stride) a.acc4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.block_hot) {   // one thread per (workgroup of k_preprocess, replica): a few hundred hot Gaussians per frame, mostly nothing to do
        const size_t pairs = (size_t)a.hot_blocks * EGS_HOT_REPLICAS;
        for (size_t i = (size_t)(job - first) * NT + threadIdx.x; i < pairs; i += stride) {
            const uint32_t b = (uint32_t)(i / EGS_HOT_REPLICAS), r = (uint32_t)(i % EGS_HOT_REPLICAS);
            const uint32_t cnt = min(a.block_hot[b], EGS_HOT_PER_BLOCK);
            float4* line = reinterpret_cast<float4*>(a.hot_base + ((size_t)r * a.hot_slots + (size_t)b * EGS_HOT_PER_BLOCK) * EGS_HOT_LINE);
            for (uint32_t k = 0; k < cnt; k++)
                for (uint32_t q = 0; q < EGS_GRAD_STRIDE / 4; q++) line[k * (EGS_HOT_LINE / 4) + q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// loss.hip: the image loss's backward, optionally carrying the jobs above (side != NULL)
// the same launch with the two terms weighed apart: w_l1_n / w_ssim_n = weights of mean|x - y| and of (1 - mean SSIM) before the division by
// the element count; upstream_ssim != NULL: each term times its own upstream scalar (egs_l1_ssim_pair_backward)
int egs_launch_l1_ssim_backward_w(int channels, int height, int width, const float* img, const float* gt, float w_l1_n, float w_ssim_n, float lambda_dssim,
                                  const float* upstream_grad, const float* upstream_ssim, const float* gate, const float* dm_dmu1, const float* dm_dexx,
                                  const float* dm_dexy, float* dL_dimg, const float* deferred_partial_sums, float* deferred_loss,
                                  float* loss_running_sum, const EgsPrologueArgs* side, hipStream_t stream);
int egs_launch_l1_ssim_backward(int channels, int height, int width, const float* img, const float* gt, float lambda_dssim,
                                const float* upstream_grad, const float* gate, const float* dm_dmu1, const float* dm_dexx,
                                const float* dm_dexy, float* dL_dimg, const float* deferred_partial_sums, float* deferred_loss,
                                float* loss_running_sum, const EgsPrologueArgs* side, hipStream_t stream);
int egs_launch_l1_ssim_forward(int channels, int height, int width, const float* img, const float* gt, float lambda_dssim,
                               float* partial_sums, float* dm_dmu1, float* dm_dexx, float* dm_dexy, float* loss, float* loss_running_sum,
                               const EgsPrologueArgs* side, hipStream_t stream);      // side != NULL: the launch carries the backward blend's preparation
