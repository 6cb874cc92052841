// binning.hip -- per-tile bucketing and depth ordering of the (Gaussian, tile) instances for gfx950.
// Replaces upstream's InclusiveSum / duplicateWithKeys / 64-bit SortPairs / identifyTileRanges stages of the op called
// from /root/reference/gaussian_renderer/__init__.py:90-98.
//
// Contract (bit-exact against oracle/raster_oracle.c): the instance list ends up ordered by
//     (tile id, float bits of depth, Gaussian index)
// which is what a stable sort of Gaussian-ordered (tile<<32 | depth) keys produces, and ranges[tile] = [start, end).
//
// How it is produced here is not a global 64-bit radix sort (six passes over 12-byte pairs).  The key is two
// independent pieces -- a tile id with a few thousand values and a depth -- so:
//   1. k_bin_count    each workgroup walks its Gaussians' tile rectangles and histograms tile ids in LDS
//                     (LDS atomics, one 4-byte counter per tile), then writes its row of the [tile][block] table;
//   2. scan           exclusive scan of that table = start of every (tile, block) slice, and of every tile;
//   3. k_bin_scatter  the same walk again; an LDS cursor per tile hands out slots; writes (depth<<32 | index) pairs
//                     bucketed by tile (order inside a bucket is arbitrary at this point);
//   4. k_tile_sort    ONE workgroup per tile sorts its bucket by the full 64-bit (depth, index) pair with an LSD
//                     radix sort whose keys live in registers and are exchanged through a single LDS buffer, and
//                     writes the Gaussian indices (point_list) and the tile's range.  Buckets larger than the
//                     register/LDS capacity take a global-memory path of the same algorithm.
// HBM traffic per instance: 8 B written + 8 B read + 4 B written (vs 6 x 24 B for the global sort), and the whole
// stage is 6 launches instead of ~35.
//
// Wave64 specifics: instance slots of 64 consecutive Gaussians are dealt to lanes by a 6-step in-wave search over the
// wave's exclusive offsets (so lanes do equal work however uneven the rectangles are); digit ranking inside a wave uses
// 8 ballots ("same digit" peer mask) + popcount-below-lane, with per-wave LDS counters touched only by each peer
// group's lowest lane.
#include "egs_common.h"
#include "blend_common.h"
#include "bin_walk.h"
#include <atomic>


namespace {

// ---------------------------------------------------------------------------------------------
// u32 scan: block-level reduce -> spine scan (recursive) -> block-level scan with carry-in.
// ---------------------------------------------------------------------------------------------
// Exclusive scan of one value per thread across a 256-thread block; returns the block total via *total.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* lds4, uint32_t* total) {
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v);
    if (lane == 63) lds4[w] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < EGS_SCAN_THREADS / 64; k++) { const uint32_t t = lds4[k]; if (k < (int)w) base += t; tot += t; }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(EGS_SCAN_THREADS) void k_scan_reduce(const uint32_t* __restrict__ in, size_t n,
                                                                   uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t lds4[4];
    const size_t base = (size_t)blockIdx.x * EGS_SCAN_EPB + (size_t)threadIdx.x * EGS_SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < EGS_SCAN_ITEMS; k++) if (base + k < n) s += in[base + k];
    uint32_t tot; block_excl_scan(s, lds4, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(EGS_SCAN_THREADS) void k_scan_single(const uint32_t* __restrict__ in,
                                                                   uint32_t* __restrict__ out, size_t n, int inclusive,
                                                                   uint64_t* __restrict__ total) {
    __shared__ uint32_t lds4[4];
    const size_t base = (size_t)threadIdx.x * EGS_SCAN_ITEMS;
    uint32_t v[EGS_SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < EGS_SCAN_ITEMS; k++) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
    uint32_t tot; uint32_t run = block_excl_scan(s, lds4, &tot);
#pragma unroll
    for (int k = 0; k < EGS_SCAN_ITEMS; k++) {
        const uint32_t ex = run; run += v[k];
        if (base + k < n) out[base + k] = inclusive ? run : ex;
    }
    if (total && threadIdx.x == 0) *total = tot;
}

__global__ __launch_bounds__(EGS_SCAN_THREADS) void k_scan_apply(const uint32_t* __restrict__ in,
                                                                  uint32_t* __restrict__ out, size_t n, int inclusive,
                                                                  const uint32_t* __restrict__ block_offsets,
                                                                  uint64_t* __restrict__ total) {
    __shared__ uint32_t lds4[4];
    const size_t base = (size_t)blockIdx.x * EGS_SCAN_EPB + (size_t)threadIdx.x * EGS_SCAN_ITEMS;
    uint32_t v[EGS_SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < EGS_SCAN_ITEMS; k++) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
    uint32_t tot; uint32_t run = block_excl_scan(s, lds4, &tot) + block_offsets[blockIdx.x];
#pragma unroll
    for (int k = 0; k < EGS_SCAN_ITEMS; k++) {
        const uint32_t ex = run; run += v[k];
        if (base + k < n) out[base + k] = inclusive ? run : ex;
    }
    if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == EGS_SCAN_THREADS - 1) *total = run;
}

// Two-kernel scan for a short spine: every block adds up the raw sums of the blocks before it (L2-resident, at most
// EGS_SCAN_SPINE_MAX words) instead of waiting for a third kernel to scan them.
#define EGS_SCAN_SPINE_MAX 4096
__global__ __launch_bounds__(EGS_SCAN_THREADS) void k_scan_apply_sum(const uint32_t* __restrict__ in,
                                                                      uint32_t* __restrict__ out, size_t n, int inclusive,
                                                                      const uint32_t* __restrict__ block_sums,
                                                                      uint64_t* __restrict__ total) {
    __shared__ uint32_t lds4[4];
    uint32_t before = 0;
    for (unsigned k = threadIdx.x; k < blockIdx.x; k += EGS_SCAN_THREADS) before += block_sums[k];
    const size_t base = (size_t)blockIdx.x * EGS_SCAN_EPB + (size_t)threadIdx.x * EGS_SCAN_ITEMS;
    uint32_t v[EGS_SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < EGS_SCAN_ITEMS; k++) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
    uint32_t carry; block_excl_scan(before, lds4, &carry);
    uint32_t tot; uint32_t run = block_excl_scan(s, lds4, &tot) + carry;
#pragma unroll
    for (int k = 0; k < EGS_SCAN_ITEMS; k++) {
        const uint32_t ex = run; run += v[k];
        if (base + k < n) out[base + k] = inclusive ? run : ex;
    }
    if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == EGS_SCAN_THREADS - 1) *total = run;
}

// Exclusive prefix, over the chunk index, of the chunk totals (the EGS_BIN_GROUPS partial accumulators of a chunk added up), written over
// row 0 of chunk_sum.  One workgroup; launched only for tables of many chunks (see k_table_scan).
#define EGS_CHUNK_PREFIX_MIN 4096
__global__ __launch_bounds__(1024) void k_chunk_prefix(uint32_t* __restrict__ chunk_sum, uint32_t n_chunks) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0u;
    __syncthreads();
    for (uint32_t k0 = 0; k0 < n_chunks; k0 += 1024) {
        const uint32_t k = k0 + threadIdx.x;
        uint32_t v = 0;
        if (k < n_chunks) {
#pragma unroll
            for (unsigned g = 0; g < EGS_BIN_GROUPS; g++) v += chunk_sum[(size_t)g * n_chunks + k];
        }
        const uint32_t incl = wave_incl_scan(v);
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t base = carry_s;
        for (unsigned j = 0; j < w; j++) base += wsum[j];
        if (k < n_chunks) chunk_sum[k] = base + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = base + incl;
        __syncthreads();
    }
}

// Exclusive scan, in place, of the bucketing's count table [n_tiles][stride] (columns >= nblocks hold nothing and count as zero):
// one kernel, because the sums of the 2048-entry chunks arrive with the table -- k_bin_count accumulated them (EGS_BIN_GROUPS partial
// accumulators per chunk).  Workgroup i adds up the chunks before it and scans its own.
// `prefixed`: k_chunk_prefix ran first and row 0 of chunk_sum holds, per chunk, the sum of all chunks before it (large images: adding
// up the chunks before it costs workgroup i 8 i loads -- 103 us of scan at 3840x2160, 8 100 chunks).
__global__ __launch_bounds__(EGS_SCAN_THREADS) void k_table_scan(uint32_t* __restrict__ table, size_t n, uint32_t stride, uint32_t nblocks,
                                                                  const uint32_t* __restrict__ chunk_sum, uint32_t n_chunks,
                                                                  uint64_t* __restrict__ total, int prefixed) {
    __shared__ uint32_t lds4[4];
    uint32_t before = 0;
    if (prefixed) {
        if (threadIdx.x == 0) before = chunk_sum[blockIdx.x];
    } else {
        for (unsigned k = threadIdx.x; k < blockIdx.x; k += EGS_SCAN_THREADS) {
#pragma unroll
            for (unsigned g = 0; g < EGS_BIN_GROUPS; g++) before += chunk_sum[(size_t)g * n_chunks + k];
        }
    }
    static_assert(EGS_SCAN_ITEMS == 8, "a thread's share is two 16-byte vectors");
    const size_t base = (size_t)blockIdx.x * EGS_SCAN_EPB + (size_t)threadIdx.x * EGS_SCAN_ITEMS;
    uint32_t v[EGS_SCAN_ITEMS], s = 0;
    const bool whole = base + EGS_SCAN_ITEMS <= n;                     // (n is a multiple of 4 and so is base: vectors never straddle the end)
    if (whole) {                                                       // 16-byte loads (eight predicated 4-byte loads ran the scan at 1.7 TB/s)
        const uint4 a = *reinterpret_cast<const uint4*>(table + base), b = *reinterpret_cast<const uint4*>(table + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
#pragma unroll
    for (int k = 0; k < EGS_SCAN_ITEMS; k++) {
        const size_t e = base + k;
        if (!whole) v[k] = e < n ? table[e] : 0u;
        if ((uint32_t)(e & (stride - 1)) >= nblocks) v[k] = 0u;       // columns beyond the last workgroup hold nothing
        s += v[k];
    }
    uint32_t carry; block_excl_scan(before, lds4, &carry);
    uint32_t tot; uint32_t run = block_excl_scan(s, lds4, &tot) + carry;
    uint32_t ex[EGS_SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < EGS_SCAN_ITEMS; k++) { ex[k] = run; run += v[k]; }
    if (whole) {
        *reinterpret_cast<uint4*>(table + base) = make_uint4(ex[0], ex[1], ex[2], ex[3]);
        *reinterpret_cast<uint4*>(table + base + 4) = make_uint4(ex[4], ex[5], ex[6], ex[7]);
    } else {
#pragma unroll
        for (int k = 0; k < EGS_SCAN_ITEMS; k++) if (base + k < n) table[base + k] = ex[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == EGS_SCAN_THREADS - 1) *total = run;
}

extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];


__global__ __launch_bounds__(EGS_BIN_THREADS) void k_bin_count(int P, int gpr, const uint32_t* __restrict__ tiles_touched,
                                                    const uint2* __restrict__ rect, const float4* __restrict__ rec, int gx,
                                                    int n_tiles, uint32_t nblocks, int cull, int use_map, int W, int H,
                                                    uint32_t* __restrict__ table, uint32_t stride, uint32_t* __restrict__ chunk_sum) {
    const unsigned bid = bin_logical_block(nblocks);
    if (bid >= nblocks) return;
    uint32_t* hist = dyn_lds;
    uint32_t* round_lds = dyn_lds + ((n_tiles + 3) & ~3);
    const bool need_depth = false; (void)need_depth;                 // (BIN_STAMP: only the count pass is stamped)
    BIN_STAMP(0);
    for (int t = threadIdx.x; t < n_tiles; t += EGS_BIN_THREADS) hist[t] = 0;            // (the first round's barrier orders this)
    for_each_instance(bid, nblocks, gpr, P, tiles_touched, rect, rec, gx, false, cull != 0, use_map != 0, W, H, round_lds,
                      [&](uint32_t tile, uint32_t, uint32_t) { atomicAdd(&hist[tile], 1u); });
    BIN_STAMP(3);
    __syncthreads();
    BIN_STAMP(4);
    bin_flush_counts(hist, n_tiles, bid, stride, table, chunk_sum, blockIdx.x);
    BIN_STAMP(5);
}

#ifdef EGS_BIN_TIMING
extern "C" int egs_debug_bin_stamps(unsigned long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(egs_bin_stamps), sizeof(egs_bin_stamps)); }
extern "C" int egs_debug_bin_unit_stamps(unsigned long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(egs_bin_unit_stamps), sizeof(egs_bin_unit_stamps)); }
#endif

__global__ __launch_bounds__(EGS_BIN_THREADS) void k_bin_scatter(int P, int gpr, const uint32_t* __restrict__ tiles_touched,
                                                      const uint2* __restrict__ rect, const float4* __restrict__ rec, int gx,
                                                      int n_tiles, uint32_t nblocks, int cull, int use_map, int W, int H,
                                                      const uint32_t* __restrict__ table_scanned, uint32_t stride,
                                                      uint32_t cap, uint64_t* __restrict__ pairs) {
    const unsigned bid = bin_logical_block(nblocks);
    if (bid >= nblocks) return;
    uint32_t* cursor = dyn_lds;
    uint32_t* round_lds = dyn_lds + ((n_tiles + 3) & ~3);
    for (int t = threadIdx.x; t < n_tiles; t += EGS_BIN_THREADS) cursor[t] = table_scanned[(size_t)t * stride + bid];
    for_each_instance(bid, nblocks, gpr, P, tiles_touched, rect, rec, gx, true, cull != 0, use_map != 0, W, H, round_lds, [&](uint32_t tile, uint32_t idx, uint32_t dbits) {
        const uint32_t pos = atomicAdd(&cursor[tile], 1u);
        if (pos < cap) pairs[pos] = ((uint64_t)dbits << 32) | idx;      // cap < R only in a speculative launch that will be redone
    });
}

}  // namespace

#include "tile_sort.h"

namespace {

__global__ void k_check_lds_atomic_order(uint32_t* __restrict__ violations) {
    __shared__ uint32_t cnt[4][TS_DIGITS];
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t bad = 0, x = 2654435761u * (threadIdx.x + 1u);
    for (int t = 0; t < 64; t++) {
        for (int i = lane; i < TS_DIGITS; i += 64) cnt[w][i] = 0;
        __builtin_amdgcn_wave_barrier();
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const uint32_t d = (x >> 8) % (t < 16 ? 2u : t < 32 ? 7u : t < 48 ? 37u : (uint32_t)TS_DIGITS);
        const uint32_t r = atomicAdd(&cnt[w][d], 1u);
        const uint64_t peers = digit_peers(d, true);
        if (r != (uint32_t)__popcll(peers & lanemask_lt())) bad++;
        __builtin_amdgcn_wave_barrier();
    }
    if (bad) atomicAdd(violations, bad);
}

// End of the bucketing chain (N_MIN == 0, the first of the two launches): the chunk sums a fused count pass accumulated into the caller's
// placement buffer have been consumed by k_table_scan -- cleared here, they are ZERO again when the next frame's k_preprocess_count starts
// adding (egs_common.h).
template <bool RANK_ATOMIC, int TS_WAVES, int TS_CAP, uint32_t N_MIN>
__global__ __launch_bounds__(64 * TS_WAVES) __attribute__((amdgpu_waves_per_eu(TS_WAVES == 4 ? 8 : 5, 8))) void k_tile_sort(EgsSortArgs A) {
    constexpr int TS_NB = TS_WAVES * 128;
    __shared__ uint64_t xbuf[TS_CAP];
    __shared__ uint32_t bkt[2 * TS_NB];
    __shared__ uint32_t lds8[2 * TS_WAVES];
    if (N_MIN == 0 && A.zero_after)
        for (uint32_t k = blockIdx.x * (uint32_t)(64 * TS_WAVES) + threadIdx.x; k < A.zero_after_n; k += gridDim.x * (uint32_t)(64 * TS_WAVES)) A.zero_after[k] = 0u;
    tile_sort_body<RANK_ATOMIC, TS_WAVES, TS_CAP, N_MIN>(A, (int)blockIdx.x, xbuf, bkt, lds8, nullptr);
}
}  // namespace

int egs_key_bits_for_tiles(int n_tiles) { int b = 0; while ((n_tiles >> b) != 0) b++; return 32 + b; }

// Scratch needed by egs_launch_scan_u32: one u32 per block at every level of the spine.
size_t egs_scan_scratch_elems(size_t n) {
    size_t tot = 0;
    while (n > EGS_SCAN_EPB) { n = (n + EGS_SCAN_EPB - 1) / EGS_SCAN_EPB; tot += (n + 63) & ~(size_t)63; }
    return tot + 64;
}

hipError_t egs_launch_scan_u32(const uint32_t* in, uint32_t* out, size_t n, int inclusive, uint32_t* scratch,
                               uint64_t* total, hipStream_t s) {
    if (n == 0) { return total ? egs_launch_zero_u32((uint32_t*)total, 2, s) : hipSuccess; }
    if (n <= EGS_SCAN_EPB) {
        hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(EGS_SCAN_THREADS), 0, s, in, out, n, inclusive, total);
        return hipGetLastError();
    }
    const size_t nb = (n + EGS_SCAN_EPB - 1) / EGS_SCAN_EPB;
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(EGS_SCAN_THREADS), 0, s, in, n, scratch);
    if (nb <= EGS_SCAN_SPINE_MAX) {
        hipLaunchKernelGGL(k_scan_apply_sum, dim3((unsigned)nb), dim3(EGS_SCAN_THREADS), 0, s, in, out, n, inclusive, scratch, total);
        return hipGetLastError();
    }
    hipError_t e = egs_launch_scan_u32(scratch, scratch, nb, 0, scratch + ((nb + 63) & ~(size_t)63), nullptr, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(EGS_SCAN_THREADS), 0, s, in, out, n, inclusive, scratch, total);
    return hipGetLastError();
}

#define EGS_DBG(s)                                                                   \
    do { if (debug) { hipError_t e_ = hipStreamSynchronize(s); if (e_ != hipSuccess) return e_; \
                      e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } } while (0)

// The count table has one column per bucketing workgroup and one row per tile; its scan and the strided column accesses grow
// with (tiles x workgroups), which at 2M Gaussians @ 3840x2160 (32 400 tiles) made the bucketing 2.3 ms with 1024 Gaussians per
// workgroup.  The workgroup count is therefore kept near 512 (two per CU) whatever P is: gpb = 256 x ceil(P / (256 x 512)) -- whole
// 256-Gaussian blocks; 1024 at 500k Gaussians, 256 at 100k (98 workgroups of 1024 Gaussians left most CUs idle while each walked
// four times the slots: BASELINE config 2's bucketing 28.9 us), 512 on the 253k-Gaussian trained scene.
#ifndef EGS_BIN_TARGET_BLOCKS
#define EGS_BIN_TARGET_BLOCKS 512
#endif
int egs_bin_gpb(int P) { const int k = (P + 256 * EGS_BIN_TARGET_BLOCKS - 1) / (256 * EGS_BIN_TARGET_BLOCKS); return 256 * (k > 1 ? k : 1); }      // whole 256-Gaussian blocks (bin_walk.h)
uint32_t egs_bin_blocks(int P) { const int g = egs_bin_gpb(P); return (uint32_t)((P + g - 1) / g); }

// Launch geometry of the bucketing kernels for a model of P Gaussians at W x H (also used by preprocess.hip's fused count pass).
EgsBinGeometry egs_bin_geometry(int P, int W, int H, int cull) {
    EgsBinGeometry q;
    q.gx = (W + EGS_TILE - 1) / EGS_TILE; q.n_tiles = q.gx * ((H + EGS_TILE - 1) / EGS_TILE);
    q.nblocks = egs_bin_blocks(P);
    // per-tile counters, then (16-byte aligned) the round's set-up block; fewer groups per round, then no culling, when
    // the counters leave too little of the 160 KiB (beyond ~28k tiles)
    const size_t counters = (size_t)((q.n_tiles + 3) & ~3) * sizeof(uint32_t), room = 160 * 1024 - counters;
    // the slot walk's owner map (16 KiB at 16 groups) is used where it costs neither groups per round nor the culling: up to ~21k tiles
    const bool use_map = bin_round_words(EGS_BIN_WAVES, cull != 0, true) * sizeof(uint32_t) <= room;
    auto fits = [&](int groups, bool with_cull) { return bin_round_words(groups, with_cull, use_map) * sizeof(uint32_t) <= room; };
    int gpr = EGS_BIN_WAVES;
    if (cull) {
        while (gpr > 4 && !fits(gpr, true)) gpr >>= 1;
        if (!fits(gpr, true)) { cull = 0; gpr = EGS_BIN_WAVES; }
    }
    if (!cull) while (gpr > 1 && !fits(gpr, false)) gpr >>= 1;
    // Two workgroups per CU (all ~489 resident at once, 32 waves per CU to hide the loads behind) beat one with twice the groups per
    // round: 1M Gaussians @ 1920x1080 (32 KiB of counters) count pass 90.6 -> 77.3 us with 8 groups per round; 4 lose again (84.7).
    {
        const bool c = cull != 0;
        if (gpr == EGS_BIN_WAVES && counters + bin_round_words(gpr, c, use_map) * sizeof(uint32_t) > 80 * 1024 &&
            counters + bin_round_words(gpr / 2, c, use_map) * sizeof(uint32_t) <= 80 * 1024) gpr /= 2;
    }
    q.gpr = gpr; q.cull = cull; q.use_map = use_map ? 1 : 0;
    q.lds = counters + bin_round_words(gpr, cull != 0, use_map) * sizeof(uint32_t);
    q.stride = egs_table_stride(q.nblocks); q.n_chunks = (uint32_t)egs_table_chunks((size_t)q.n_tiles, q.stride);
    return q;
}

// counted: the table and the chunk sums of this frame are in place already (preprocess.hip's k_preprocess_count walked the rectangles)
// sort_in_blend (may be NULL): filled with what the forward blend needs to sort its tiles itself -- then no sort launch is made here
// (table_scanned == NULL in it: not taken, e.g. a second sort instantiation would be needed)
hipError_t egs_launch_binning(int P, int64_t R64, int W, int H, EgsGeomPtrs g, EgsBinPtrs b, EgsImgPtrs im,
                              uint64_t* running_max, uint32_t* overflow_flag, int sums_zeroed, int counted, EgsSortArgs* sort_in_blend, hipStream_t s, int call_flags) {
    const int debug = call_flags & EGS_CALL_SYNC;
    if (sort_in_blend) sort_in_blend->table_scanned = nullptr;
    const EgsBinGeometry q = egs_bin_geometry(P, W, H, (call_flags & EGS_CALL_KEEP_ALL_INSTANCES) ? 0 : 1);
    const int gx = q.gx, n_tiles = q.n_tiles;
    if (R64 == 0 || P == 0) {
        if (overflow_flag) { hipError_t e = egs_launch_zero_u32(overflow_flag, 2, s); if (e != hipSuccess) return e; }
        return egs_launch_zero_u32((uint32_t*)im.ranges, 2 * (size_t)n_tiles, s);
    }
    const uint32_t R = (uint32_t)R64;
    const uint32_t nblocks = q.nblocks;
    const int cull = q.cull, gpr = q.gpr; const bool use_map = q.use_map != 0;
    const size_t lds = q.lds;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_bin_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)k_bin_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint32_t stride = q.stride, n_chunks = q.n_chunks;
    egs_prof_start(EGS_K_DUPLICATE, s);
    if (!counted) {
        if (!sums_zeroed) {
            hipError_t e0 = egs_launch_zero_u32(b.chunk_sum, (size_t)EGS_BIN_GROUPS * n_chunks, s);
            if (e0 != hipSuccess) return e0;
        }
        hipLaunchKernelGGL(k_bin_count, dim3(((nblocks + 7) / 8) * 8), dim3(EGS_BIN_THREADS), lds, s, P, gpr, g.offsets, g.rect, g.rec, gx, n_tiles, nblocks, cull, use_map ? 1 : 0, W, H,
                           b.table, stride, b.chunk_sum);
        EGS_DBG(s);
    }
    const int prefixed = n_chunks >= EGS_CHUNK_PREFIX_MIN;
    if (prefixed) hipLaunchKernelGGL(k_chunk_prefix, dim3(1), dim3(1024), 0, s, b.chunk_sum, n_chunks);
    hipLaunchKernelGGL(k_table_scan, dim3(n_chunks), dim3(EGS_SCAN_THREADS), 0, s, b.table, (size_t)n_tiles * stride, stride, nblocks, b.chunk_sum, n_chunks, b.total, prefixed);
    hipLaunchKernelGGL(k_bin_scatter, dim3(((nblocks + 7) / 8) * 8), dim3(EGS_BIN_THREADS), lds, s, P, gpr, g.offsets, g.rect, g.rec, gx, n_tiles, nblocks,
                       cull, use_map ? 1 : 0, W, H, b.table, stride, R, b.pairs);
    egs_prof_stop(EGS_K_DUPLICATE, s);
    EGS_DBG(s);
    int index_bits = 0; while (((unsigned)(P - 1) >> index_bits) != 0) index_bits++;
    // One-time device check of the LDS lane-order property the fast ranking relies on (see wave_digit_rank).
    static std::atomic<int> lds_rank_ok{-1};
    int fast = lds_rank_ok.load();
    if (fast < 0) {
        uint32_t* flag = b.flag;
        hipError_t e2 = egs_launch_zero_u32(flag, 1, s);
        if (e2 != hipSuccess) return e2;
        hipLaunchKernelGGL(k_check_lds_atomic_order, dim3(64), dim3(256), 0, s, flag);
        uint32_t bad = 1;
        e2 = hipMemcpyAsync(&bad, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
        if (e2 != hipSuccess) return e2;
        e2 = hipStreamSynchronize(s);
        if (e2 != hipSuccess) return e2;
        fast = bad == 0 ? 1 : 0;
        lds_rank_ok.store(fast);
    }
    if (call_flags & EGS_CALL_BALLOT_RANK) fast = 0;
    const int ip = (index_bits + TS_DBITS - 1) / TS_DBITS;
    // A launch costs ~4.5 us of GPU time even when every workgroup returns at once.  When the buffer holds on average at most 2048
    // instances per tile (R is the capacity: >= 1.25 x the rectangle count, itself ~1.5 x what survives culling) no tile is expected
    // to need the second instantiation, so it is not launched; a tile that does exceed 2048 is still sorted, by the first one's
    // global-memory path.
    const int solo = (uint64_t)R <= 2048ull * (uint64_t)n_tiles ? 1 : 0;
    EgsSortArgs sa = { n_tiles, stride, b.table, b.total, running_max, overflow_flag, solo, R, ip, b.pairs, b.scratch, b.point_list, im.ranges,
                       b.zero_after, b.zero_after_n };
    if (sort_in_blend && solo) {
        // the forward blend sorts every tile itself (render_fwd.hip SORT instantiations): no sort launch.  What the launch carries besides the
        // sort -- the overflow word, the running maximum, clearing the chunk sums -- travels in `sort_out` to that launch.
        sa.rank_atomic = fast;
        *sort_in_blend = sa;
        return hipGetLastError();                                     // (no EGS_K_SORT stage: the sort's time is part of the forward blend's)
    }
    egs_prof_start(EGS_K_SORT, s);
    if (sort_in_blend) sort_in_blend->table_scanned = nullptr;        // (not taken: the forward reads the lists these launches leave)
    EgsSortArgs sa2 = sa; sa2.solo = 0;
    if (fast) {
        hipLaunchKernelGGL((k_tile_sort<true, 4, TS_SMALL_CAP, 0u>), dim3(n_tiles), dim3(256), 0, s, sa);
        if (!solo) hipLaunchKernelGGL((k_tile_sort<true, 8, 4096, TS_SMALL_CAP + 1u>), dim3(n_tiles), dim3(512), 0, s, sa2);
    } else {
        hipLaunchKernelGGL((k_tile_sort<false, 4, TS_SMALL_CAP, 0u>), dim3(n_tiles), dim3(256), 0, s, sa);
        if (!solo) hipLaunchKernelGGL((k_tile_sort<false, 8, 4096, TS_SMALL_CAP + 1u>), dim3(n_tiles), dim3(512), 0, s, sa2);
    }
    egs_prof_stop(EGS_K_SORT, s);
    EGS_DBG(s);
    return hipGetLastError();
}
