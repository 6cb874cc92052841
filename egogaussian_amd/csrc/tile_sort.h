// tile_sort.h -- the per-tile depth sort of the bucketed (depth << 32 | index) pairs as device code shared by binning.hip (k_tile_sort) and
// render_fwd.hip (the forward blend sorting its own tile first).  Replaces upstream's 64-bit SortPairs + identifyTileRanges stages of the op
// called from /root/reference/gaussian_renderer/__init__.py:90-98.
#pragma once
#include "egs_common.h"
#include "bin_walk.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Per-tile sort of (depth<<32 | index) pairs.  LSD radix with 9-bit digits over only the bits that can differ:
//   depth   the tile's smallest depth word is subtracted first (positive floats order like their bit patterns), so a
//           tile whose depths span [zmin, zmax] needs ceil(bits(zmax - zmin) / 9) passes -- three for any range up to
//           2^27 ulps, which covers every scene with z in [0.2, 1e3]; four only beyond that
//   index   ceil(index_bits / 9) passes, run only when two entries of the tile share a depth (see below)
// Stable ranking as in a global radix pass, but the whole bucket belongs to one workgroup: wave w owns the contiguous
// quarter [w*chunk, (w+1)*chunk) in rounds of 64.
// ---------------------------------------------------------------------------------------------
// Two instantiations share the launch sequence: <4 waves, 1792 pairs> (18.3 KiB of LDS and 64 VGPRs: eight workgroups per CU, so a
// 960x540 frame's 2040 tiles are all resident at once) sorts every tile of up to 1792 instances; <8 waves, 4096 pairs> takes the
// larger ones and, beyond 4096, the depth-slab path.  Each workgroup returns at once if its tile belongs to the other.
#define TS_DBITS 9
#define TS_SMALL_CAP 1792           // pairs the four-wave instantiation sorts in registers: 14 KiB + 4 KiB of buckets = 18.7 KiB of LDS, EIGHT workgroups per CU
                                    // (2048 pairs were 20.3 KiB: seven per CU, 1792 of a 960x540 frame's 2040 tiles resident)
#define TS_DIGITS (1 << TS_DBITS)
#define TS_BUCKET_MAX 96u           // largest top-digit bucket the in-bucket comparison takes (see k_tile_sort)
#define TS_SLAB_BUCKET_MAX 512u     // the same for a tile sorted in depth slabs (what it falls back to is far slower than a long comparison loop)

__device__ __forceinline__ uint64_t digit_peers(uint32_t d, bool ok) {
    uint64_t peers = __ballot(ok);
#pragma unroll
    for (int b = 0; b < TS_DBITS; b++) {
        const uint64_t m = __ballot((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? m : ~m;
    }
    return peers;
}

// Digit `pass` of a pair: passes [0, index_passes) walk the index word, the rest walk (depth - dmin).
__device__ __forceinline__ uint32_t ts_digit(uint64_t kv, int pass, int index_passes, uint32_t dmin) {
    const uint32_t word = pass < index_passes ? (uint32_t)kv : (uint32_t)(kv >> 32) - dmin;
    const int sh = TS_DBITS * (pass < index_passes ? pass : pass - index_passes);
    return (word >> sh) & (TS_DIGITS - 1);
}

// Exclusive scan of one value per thread over the FIRST 256 threads of the workgroup (every thread must call).  The caller
// must pass a workgroup barrier before the next call (lds4 is reused); both users end with one.
__device__ __forceinline__ uint32_t scan_first_256(uint32_t v, uint32_t* lds4) {
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v);
    if (lane == 63 && w < 4) lds4[w] = incl;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) if (k < (int)w) base += lds4[k];
    return base + incl - v;
}

// Register path: a wave's share holds at most 1024 pairs, so two digit counters share one LDS word (16 bits each).
// After every wave has accumulated its counts: turn them into exclusive positions
//   pos[w][d] = (#keys with digit < d) + (#keys with digit d in waves < w).       thread t < 256 owns digits 2t, 2t+1.
template <int TS_WAVES>
__device__ __forceinline__ void digit_bases_packed(uint32_t (*cnt)[256], uint32_t* lds4) {
    const unsigned t = threadIdx.x;
    uint32_t c[TS_WAVES], s = 0;
    if (t < 256) {
#pragma unroll
        for (int k = 0; k < TS_WAVES; k++) { c[k] = cnt[k][t]; s += c[k]; }        // halves add independently (each total <= 4096)
    }
    const uint32_t lo = s & 0xffffu, hi = s >> 16;
    const uint32_t base = scan_first_256(lo + hi, lds4);
    if (t < 256) {
        uint32_t b = base | ((base + lo) << 16);                        // wave 0: digit 2t starts at base, digit 2t+1 after all of 2t
#pragma unroll
        for (int k = 0; k < TS_WAVES; k++) { cnt[k][t] = b; b += c[k]; }
    }
    __syncthreads();
}

// Oversize path: full-width counters, thread t < 256 owns digits 2t and 2t+1.
template <int TS_WAVES>
__device__ __forceinline__ void digit_bases_wide(uint32_t (*cnt)[TS_DIGITS], uint32_t* lds4) {
    const unsigned t = threadIdx.x;
    uint32_t a[TS_WAVES], b[TS_WAVES], sa = 0, sb = 0;
    if (t < 256) {
#pragma unroll
        for (int k = 0; k < TS_WAVES; k++) { a[k] = cnt[k][2 * t]; b[k] = cnt[k][2 * t + 1]; sa += a[k]; sb += b[k]; }
    }
    uint32_t ba = scan_first_256(sa + sb, lds4), bb = ba + sa;
    if (t < 256) {
#pragma unroll
        for (int k = 0; k < TS_WAVES; k++) { cnt[k][2 * t] = ba; cnt[k][2 * t + 1] = bb; ba += a[k]; bb += b[k]; }
    }
    __syncthreads();
}

// Stable rank of a key inside its wave's quarter, by digit.  Two implementations:
//   RANK_ATOMIC = true   one ds_add_rtn_u32 on the wave's counter of that digit.  This relies on the LDS resolving
//                        same-address lanes of one wave-instruction in increasing lane order, which is what gfx950 does
//                        (tools/ubench/lds_atomic_order.hip: 2.3e8 lane-operations, none out of order) but is not an
//                        architectural promise -- egs_launch_binning verifies it on the device once per process
//                        (k_check_lds_atomic_order) and otherwise uses
//   RANK_ATOMIC = false  ballots build the "same digit" peer mask; all lanes read the counter, the lowest peer bumps it.
// `PACKED`: the counter of digit d is the 16-bit half (d & 1) of word d >> 1.
template <bool RANK_ATOMIC, bool PACKED>
__device__ __forceinline__ uint32_t wave_digit_rank(uint32_t* cnt_w, uint32_t d, bool ok, unsigned lane, uint64_t lt) {
    uint32_t* word = PACKED ? cnt_w + (d >> 1) : cnt_w + d;
    const uint32_t sh = PACKED ? 16u * (d & 1u) : 0u;
    const uint32_t mask = PACKED ? 0xffffu : 0xffffffffu;
    if (RANK_ATOMIC) return ok ? (atomicAdd(word, 1u << sh) >> sh) & mask : 0u;
    const uint64_t peers = digit_peers(d, ok);
    const uint32_t start = (*word >> sh) & mask;                        // every lane reads before any leader adds (in-order LDS)
    if (ok && lane == (unsigned)__ffsll((unsigned long long)peers) - 1u) atomicAdd(word, (uint32_t)__popcll(peers) << sh);
    return start + (uint32_t)__popcll(peers & lt);
}


// min and max over the workgroup of one value per thread (both returned to every thread)
template <int TS_WAVES>
__device__ __forceinline__ void block_min_max(uint32_t& mn, uint32_t& mx, uint32_t* lds2w) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, d, 64));
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    }
    const unsigned w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { lds2w[w] = mn; lds2w[TS_WAVES + w] = mx; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TS_WAVES; k++) { mn = min(mn, lds2w[k]); mx = max(mx, lds2w[TS_WAVES + k]); }
    __syncthreads();
}

#ifdef EGS_BIN_TIMING
__device__ unsigned long long egs_sort_stamps[2048 * 8];
#define SORT_STAMP(ph) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && tile < 2048) egs_sort_stamps[tile * 8 + (ph)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int egs_debug_sort_stamps(unsigned long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(egs_sort_stamps), sizeof(egs_sort_stamps)); }
#else
#define SORT_STAMP(ph)
#endif
// The per-tile sort as a device function: k_tile_sort (binning.hip) calls it for tile blockIdx.x; the forward blend (render_fwd.hip) calls it for the
// tile it is about to blend, in the LDS it will stage records in afterwards (SORT instantiations: one launch less per frame).
// xbuf: TS_CAP uint64, bkt: 2 * TS_NB uint32, lds8: 2 * TS_WAVES uint32 of LDS.  *range_out = the tile's [start, end) (0, 0 when empty).
template <bool RANK_ATOMIC, int TS_WAVES, int TS_CAP, uint32_t N_MIN>
__device__ __forceinline__ void tile_sort_body(const EgsSortArgs& A, const int tile, uint64_t* __restrict__ xbuf, uint32_t* __restrict__ bkt, uint32_t* __restrict__ lds8,
                                               uint2* range_out) {
    const int n_tiles = A.n_tiles; const uint32_t stride = A.stride; const uint32_t* __restrict__ table_scanned = A.table_scanned;
    const uint64_t* __restrict__ total = A.total; uint64_t* __restrict__ running_max = A.running_max; uint32_t* __restrict__ overflow_flag = A.overflow_flag;
    const int solo = A.solo; const uint32_t R = A.R; const int index_passes = A.index_passes; uint64_t* __restrict__ pairs = A.pairs;
    uint64_t* __restrict__ scratch = A.scratch; uint32_t* __restrict__ point_list = A.point_list; uint2* __restrict__ ranges = A.ranges;
    constexpr int TS_THREADS = 64 * TS_WAVES, TS_ITEMS = TS_CAP / TS_THREADS;
    static_assert(TS_CAP * 2 >= TS_WAVES * TS_DIGITS, "the oversize path keeps its counters in the exchange buffer");
    constexpr int TS_NB = TS_WAVES * 128;                              // depth buckets of the one-pass path (512 for the four-wave instantiation)
    static_assert(2 * TS_NB >= TS_WAVES * 256 && TS_NB % 256 == 0, "the digit counters of the fallback passes live in the bucket arrays");
    uint32_t (*cnt)[256] = reinterpret_cast<uint32_t (*)[256]>(bkt);   // bkt: one-pass path: bucket cursors [TS_NB], bucket starts [TS_NB]; fallback: cnt[TS_WAVES][256]
    SORT_STAMP(0);
    const uint32_t beg = table_scanned[(size_t)tile * stride];
    const uint32_t end = tile + 1 < n_tiles ? table_scanned[(size_t)(tile + 1) * stride] : (uint32_t)*total;
    const uint32_t n = end > R ? 0u : end - beg;                     // end > capacity: speculative launch that overflowed
    if (N_MIN == 0) {                                                // the first of the two launches also publishes the bookkeeping
        if (running_max && tile == 0 && threadIdx.x == 0 && *total > *running_max) *running_max = *total;   // for hipGraph replays (api.hip)
        if (overflow_flag && tile == 0 && threadIdx.x == 0) {          // include/egs_raster.h: [0] this frame was clipped, [1] its instance count
            overflow_flag[0] = *total > (uint64_t)R ? 1u : 0u; overflow_flag[1] = (uint32_t)min(*total, (uint64_t)0xffffffffu);
        }
        if (threadIdx.x == 0) ranges[tile] = n ? make_uint2(beg, end) : make_uint2(0u, 0u);
        if (range_out) *range_out = n ? make_uint2(beg, end) : make_uint2(0u, 0u);
    }
    // `solo`: the second instantiation is not launched (no tile is expected beyond TS_CAP); one that is takes the global-memory path here
    if (n == 0 || (N_MIN == 0 ? (n > (uint32_t)TS_CAP && !solo) : n < N_MIN)) return;      // empty, or the other instantiation's tile
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t lt = lanemask_lt();
    const uint32_t chunk = ((n + TS_WAVES - 1) / TS_WAVES + 63) & ~63u;       // per-wave share, multiple of 64
    const uint32_t wbeg = w * chunk;

    if (n <= TS_CAP) {
        // ---- register path ----
        // Depth ties inside a tile are rare, so the bucket is first sorted on the depth digits only; if two
        // neighbours then share a depth the index digits are sorted and the depth digits redone (LSD order), which
        // restores the canonical (depth, index) order.  `phase` 0: depth only; 1: index then depth.
        uint64_t key[TS_ITEMS];
        uint32_t dmin = 0xffffffffu, dmax = 0u;
#pragma unroll
        for (int r = 0; r < TS_ITEMS; r++) {
            const uint32_t i = wbeg + r * 64 + lane;
            const bool ok = r * 64u < chunk && i < n;
            key[r] = ok ? pairs[beg + i] : ~0ull;
            if (ok) { const uint32_t dw = (uint32_t)(key[r] >> 32); dmin = min(dmin, dw); dmax = max(dmax, dw); }
        }
        SORT_STAMP(1);
        for (int k = threadIdx.x; k < TS_NB; k += TS_THREADS) bkt[k] = 0;     // (block_min_max's first barrier orders this before the counting)
        block_min_max<TS_WAVES>(dmin, dmax, lds8);
        SORT_STAMP(2);
        const int depth_passes = (32 - __clz((int)(dmax - dmin)) + TS_DBITS - 1) / TS_DBITS;     // 0 when every depth is equal
        const int npass = index_passes + depth_passes;
        if (dmax > dmin) {
            // ---- one pass: TS_NB buckets LINEAR in the depth value, then every key counts the smaller keys of its own bucket ----
            // bucket(z) = floor((z - zmin) * TS_NB / (zmax - zmin)) is non-decreasing in z (IEEE subtraction, multiplication and the
            // conversion are monotonic) and depths are positive floats, which order like their bit patterns: a valid first digit.
            // It spreads a tile's instances evenly whatever the exponent range -- the top nine bits of the pattern (rounds 1-2) put a
            // quarter of a [2, 10] depth range into a sixteenth of the buckets.  Where a key lands INSIDE its bucket does not matter:
            // its final position is the bucket's start + the number of smaller (depth, index) keys in the bucket, which also settles
            // depth ties.  So a bucket needs ONE shared counter -- no per-wave stable ranks, no per-wave digit bases.  A bucket of
            // more than TS_BUCKET_MAX keys (depths piled up) sends the tile to the digit-by-digit passes below.
            const float zmin = __uint_as_float(dmin), zscale = (float)TS_NB / (__uint_as_float(dmax) - zmin);
            auto bucket = [&](uint64_t kv) -> uint32_t {
                const float f = (__uint_as_float((uint32_t)(kv >> 32)) - zmin) * zscale;
                return (uint32_t)fminf(fmaxf(f, 0.f), (float)(TS_NB - 1));          // (NaN -> 0)
            };
            uint32_t dg[TS_ITEMS];
#pragma unroll
            for (int r = 0; r < TS_ITEMS; r++) {
                const uint32_t i = wbeg + r * 64 + lane;
                dg[r] = bucket(key[r]);
                if (r * 64u < chunk && i < n) atomicAdd(&bkt[dg[r]], 1u);
            }
            SORT_STAMP(3);
            __syncthreads();
            {   // exclusive scan of the bucket counts: the first 256 threads take TS_NB / 256 consecutive buckets each
                constexpr int PER = TS_NB / 256;
                uint32_t c[PER], sum = 0;
                if (threadIdx.x < 256) {
#pragma unroll
                    for (int k = 0; k < PER; k++) { c[k] = bkt[threadIdx.x * PER + k]; sum += c[k]; }
                }
                uint32_t base = scan_first_256(sum, lds8);
                if (threadIdx.x < 256) {
#pragma unroll
                    for (int k = 0; k < PER; k++) { bkt[threadIdx.x * PER + k] = base; bkt[TS_NB + threadIdx.x * PER + k] = base; base += c[k]; }
                }
                __syncthreads();
            }
            SORT_STAMP(4);
#pragma unroll
            for (int r = 0; r < TS_ITEMS; r++) {
                const uint32_t i = wbeg + r * 64 + lane;
                if (r * 64u < chunk && i < n) xbuf[atomicAdd(&bkt[dg[r]], 1u)] = key[r];
            }
            __syncthreads();
            SORT_STAMP(5);
            bool big = false;
#pragma unroll
            for (int r = 0; r < TS_ITEMS; r++) {
                const uint32_t i = wbeg + r * 64 + lane;
                if (r * 64u < chunk && i < n) {
                    const uint64_t k = xbuf[i];
                    const uint32_t d = bucket(k);
                    const uint32_t bs = bkt[TS_NB + d], be = d + 1 < TS_NB ? bkt[TS_NB + d + 1] : n;
                    if (be - bs > TS_BUCKET_MAX) { big = true; continue; }
                    uint32_t smaller = 0;
                    for (uint32_t q = bs; q < be; q++) smaller += xbuf[q] < k ? 1u : 0u;
                    point_list[beg + bs + smaller] = (uint32_t)k;
                }
            }
            SORT_STAMP(6);
            if (!__syncthreads_or(big ? 1 : 0)) return;
        } else {
            __syncthreads();
        }
        // Barriers per pass: 4.  Every wave clears ITS OWN counters (nobody else touches them between the barrier after the
        // scatter and the one after the ranking), so clearing needs no workgroup barrier.  (The bucket arrays become the counters.)
        for (int k = lane; k < 256; k += 64) cnt[w][k] = 0;
        for (int phase = depth_passes ? 0 : 1; phase < 2; phase++) {
            for (int p = phase == 0 ? index_passes : 0; p < npass; p++) {
                uint32_t rank[TS_ITEMS];
#pragma unroll
                for (int r = 0; r < TS_ITEMS; r++) {
                    rank[r] = 0;
                    if (r * 64u < chunk) {                              // wave-uniform
                        const uint32_t i = wbeg + r * 64 + lane;
                        // running count of this digit in the wave's quarter (LDS operations of one wave execute in
                        // order, so round r+1 sees round r's update)
                        rank[r] = wave_digit_rank<RANK_ATOMIC, true>(cnt[w], ts_digit(key[r], p, index_passes, dmin), i < n, lane, lt);
                    }
                }
                __syncthreads();
                digit_bases_packed<TS_WAVES>(cnt, lds8);
#pragma unroll
                for (int r = 0; r < TS_ITEMS; r++) {
                    const uint32_t i = wbeg + r * 64 + lane;
                    if (r * 64u < chunk && i < n) {
                        const uint32_t d = ts_digit(key[r], p, index_passes, dmin);
                        xbuf[((cnt[w][d >> 1] >> (16u * (d & 1u))) & 0xffffu) + rank[r]] = key[r];
                    }
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < TS_ITEMS; r++) {
                    const uint32_t i = wbeg + r * 64 + lane;
                    if (r * 64u < chunk && i < n) key[r] = xbuf[i];
                }
                for (int k = lane; k < 256; k += 64) cnt[w][k] = 0;     // own counters, for the next pass
                // xbuf stays intact until the next pass writes it (after three barriers), so it can be read below
            }
            if (phase == 1) break;
            // sorted by depth: any equal-depth neighbours?
            bool tie = false;
#pragma unroll
            for (int r = 0; r < TS_ITEMS; r++) {
                const uint32_t i = wbeg + r * 64 + lane;
                if (r * 64u < chunk && i + 1 < n) tie = tie || (uint32_t)(key[r] >> 32) == (uint32_t)(xbuf[i + 1] >> 32);
            }
            if (!__syncthreads_or(tie ? 1 : 0)) break;
        }
#pragma unroll
        for (int r = 0; r < TS_ITEMS; r++) {
            const uint32_t i = wbeg + r * 64 + lane;
            if (r * 64u < chunk && i < n) point_list[beg + i] = (uint32_t)key[r];
        }
        return;
    }

    // ---- a tile with more instances than the registers hold: DEPTH SLABS through the same LDS buffer ----
    // The one-pass idea again, with the keys streamed from global memory: histogram all n keys into the TS_NB linear depth buckets,
    // scan, then take runs of consecutive buckets holding at most TS_CAP keys ("slabs") one after another -- gather the slab's keys
    // into LDS in bucket order, rank every key inside its bucket by comparison, write its index.  Slabs are disjoint depth ranges in
    // increasing order, so the list comes out sorted; the keys are read once per slab (n^2 / TS_CAP reads in all, L2-resident).
    // Trained scenes put 5-10 k splats into their densest tiles; the digit-by-digit global-memory passes below (six passes with
    // agent-scope fences) took 2.2 ms per frame for S(500k) with every splat three times larger (profiles/r3_footprint_sweep.md).
    uint64_t* src = pairs + beg;
    uint32_t dmin = 0xffffffffu, dmax = 0u;
    for (uint32_t i = threadIdx.x; i < n; i += TS_THREADS) { const uint32_t dw = (uint32_t)(src[i] >> 32); dmin = min(dmin, dw); dmax = max(dmax, dw); }
    for (int k = threadIdx.x; k < TS_NB; k += TS_THREADS) bkt[k] = 0;
    block_min_max<TS_WAVES>(dmin, dmax, lds8);
    if (dmax > dmin) {
        const float zmin = __uint_as_float(dmin), zscale = (float)TS_NB / (__uint_as_float(dmax) - zmin);
        auto bucket = [&](uint64_t kv) -> uint32_t {
            const float f = (__uint_as_float((uint32_t)(kv >> 32)) - zmin) * zscale;
            return (uint32_t)fminf(fmaxf(f, 0.f), (float)(TS_NB - 1));
        };
        for (uint32_t i = threadIdx.x; i < n; i += TS_THREADS) atomicAdd(&bkt[bucket(src[i])], 1u);
        __syncthreads();
        bool too_big = false;                                          // a bucket the in-bucket comparison (or a slab) cannot take
        {
            constexpr int PER = TS_NB / 256;
            uint32_t c[PER], sum = 0;
            if (threadIdx.x < 256) {
#pragma unroll
                for (int k = 0; k < PER; k++) { c[k] = bkt[threadIdx.x * PER + k]; sum += c[k]; too_big = too_big || c[k] > TS_SLAB_BUCKET_MAX; }
            }
            uint32_t base = scan_first_256(sum, lds8);
            if (threadIdx.x < 256) {
#pragma unroll
                for (int k = 0; k < PER; k++) { bkt[threadIdx.x * PER + k] = base; bkt[TS_NB + threadIdx.x * PER + k] = base; base += c[k]; }
            }
        }
        if (!__syncthreads_or(too_big ? 1 : 0)) {
            auto bstart = [&](uint32_t d) -> uint32_t { return d < (uint32_t)TS_NB ? bkt[TS_NB + d] : n; };
            for (uint32_t d0 = 0; d0 < (uint32_t)TS_NB;) {             // (every quantity below is workgroup-uniform)
                const uint32_t s0 = bstart(d0);
                uint32_t lo = d0 + 1, hi = TS_NB;                      // largest d1 in [d0 + 1, TS_NB] with bstart(d1) - s0 <= TS_CAP (d0 + 1 qualifies)
                while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (bstart(mid) - s0 <= (uint32_t)TS_CAP) lo = mid; else hi = mid - 1; }
                const uint32_t d1 = lo, m = bstart(d1) - s0;
                if (m) {
                    for (uint32_t i = threadIdx.x; i < n; i += TS_THREADS) {
                        const uint64_t k = src[i];
                        const uint32_t d = bucket(k);
                        if (d >= d0 && d < d1) xbuf[atomicAdd(&bkt[d], 1u) - s0] = k;
                    }
                    __syncthreads();
                    for (uint32_t j = threadIdx.x; j < m; j += TS_THREADS) {
                        const uint64_t k = xbuf[j];
                        const uint32_t d = bucket(k);
                        const uint32_t bs = bstart(d) - s0, be = bstart(d + 1) - s0;
                        uint32_t smaller = 0;
                        for (uint32_t q = bs; q < be; q++) smaller += xbuf[q] < k ? 1u : 0u;
                        point_list[beg + s0 + bs + smaller] = (uint32_t)k;
                    }
                    __syncthreads();                                    // xbuf is free for the next slab
                }
                d0 = d1;
            }
            return;
        }
    }
    __syncthreads();

    // ---- last resort (depths piled up in one bucket, or all equal): same digits as the register path's passes, keys stay in global
    // memory (ping-pong with `scratch`), full (index, depth) order ----
    uint32_t (*wide)[TS_DIGITS] = reinterpret_cast<uint32_t (*)[TS_DIGITS]>(xbuf);     // xbuf is idle on this path
    uint64_t* dst = scratch + beg;
    const int depth_passes = (32 - __clz((int)(dmax - dmin)) + TS_DBITS - 1) / TS_DBITS;
    const int npass = index_passes + depth_passes;
    for (int p = 0; p < npass; p++) {
        for (int k = threadIdx.x; k < TS_WAVES * TS_DIGITS; k += TS_THREADS) reinterpret_cast<uint32_t*>(xbuf)[k] = 0;
        __syncthreads();
        for (uint32_t r0 = 0; r0 < chunk; r0 += 64) {                  // count
            const uint32_t i = wbeg + r0 + lane;
            if (i < n) atomicAdd(&wide[w][ts_digit(src[i], p, index_passes, dmin)], 1u);     // counting only: order irrelevant
        }
        __syncthreads();
        digit_bases_wide<TS_WAVES>(wide, lds8);
        const bool last = p == npass - 1;
        for (uint32_t r0 = 0; r0 < chunk; r0 += 64) {                  // rank and move (wide[w][d] is the running cursor)
            const uint32_t i = wbeg + r0 + lane;
            const bool ok = i < n;
            const uint64_t kv = ok ? src[i] : 0ull;
            const uint32_t pos = wave_digit_rank<RANK_ATOMIC, false>(wide[w], ts_digit(kv, p, index_passes, dmin), ok, lane, lt);
            if (ok) {
                if (last) point_list[beg + pos] = (uint32_t)kv;
                else dst[pos] = kv;
            }
        }
        // other waves of this workgroup read `dst` next pass: publish, then drop any stale L1 lines of it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        uint64_t* t = src; src = dst; dst = t;
    }
}

}  // namespace
