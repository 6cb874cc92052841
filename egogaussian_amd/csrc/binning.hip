// binning.hip -- instance generation and ordering for gfx950:
//   scan of tiles-touched  ->  (tile | depth) keys per (Gaussian, tile) instance  ->  stable LSD radix
//   sort on the low 32 + bits(tiles) key bits  ->  per-tile [start, end) ranges.
// Replaces upstream's InclusiveSum / duplicateWithKeys / SortPairs / identifyTileRanges stages of the op
// called from /root/reference/gaussian_renderer/__init__.py:90-98.
//
// The sort order is the canonical (tile, depth bits, Gaussian index) order: instances are generated in
// Gaussian-index order and every pass is stable, so ties keep index order -- bit-exact against
// oracle/raster_oracle.c::egso_sort_pairs.
//
// Wave64 specifics: digit ranking inside a wave uses 8 ballots (one per digit bit) to build the
// "same digit" peer mask and v_mbcnt-style popcounts below the lane; per-wave digit counters live in LDS
// and are touched only by each peer group's lowest lane, so there are no LDS atomics in the ranking.
#include "egs_common.h"

namespace {

__device__ __forceinline__ uint64_t lanemask_lt() {
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}
__device__ __forceinline__ unsigned lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// ---------------------------------------------------------------------------------------------
// u32 scan: block-level reduce -> spine scan (recursive) -> block-level scan with carry-in.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t n = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v += n;
    }
    return v;
}

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

// Scans n <= EGS_SCAN_EPB elements in one block (exclusive), in place allowed.
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

// ---------------------------------------------------------------------------------------------
// duplicate: one key/value per touched tile.  A wave owns 64 consecutive Gaussians and emits their
// instances cooperatively: output slot s of the wave's contiguous span is mapped back to its Gaussian
// by a 6-step search over the wave's exclusive offsets (held one per lane), so the 12-byte stores of a
// wave are consecutive instead of 64 separate strided runs.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_duplicate(int P, const float4* __restrict__ rec, const uint2* __restrict__ rect,
                                                    const uint32_t* __restrict__ offsets, int gx,
                                                    uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = lane_id();
    const int wave_first = i - (int)lane;
    if (wave_first >= P) return;
    const bool have = i < P;
    const uint32_t incl = have ? offsets[i] : 0u;
    uint32_t prev = __shfl_up(incl, 1, 64);
    if (lane == 0) prev = wave_first == 0 ? 0u : offsets[wave_first - 1];
    const uint32_t wave_base = __shfl(prev, 0, 64);
    const int last_lane = min(63, P - 1 - wave_first);
    const uint32_t wave_end = __shfl(incl, last_lane, 64);
    const uint32_t excl = have ? prev - wave_base : 0xffffffffu;     // start of my span, relative to the wave
    const uint32_t cnt = have ? incl - prev : 0u;
    uint2 rc = make_uint2(0u, 0u); uint32_t dbits = 0;
    if (cnt) { rc = rect[i]; dbits = __float_as_uint(rec[(size_t)i * EGS_SPLAT_REC_F4 + 2].y); }
    const uint32_t total = wave_end - wave_base;
    for (uint32_t s0 = 0; s0 < total; s0 += 64) {
        const uint32_t s = s0 + lane;
        // owner = last lane whose span starts at or before s (spans are sorted; empty spans share a start
        // with their successor, and the search lands on the last of them -- fix up by requiring cnt > 0 via
        // "start <= s" on the NEXT lane being false).
        int lo = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
            const int probe = lo + step;
            const uint32_t st = __shfl(excl, probe & 63, 64);
            if (probe < 64 && st <= s) lo = probe;
        }
        const uint32_t ost = __shfl(excl, lo, 64);
        const uint2 orc = make_uint2(__shfl(rc.x, lo, 64), __shfl(rc.y, lo, 64));
        const uint32_t odb = __shfl(dbits, lo, 64);
        if (s < total) {
            const uint32_t k = s - ost;
            const uint32_t x0 = orc.x & 0xffffu, x1 = orc.x >> 16, y0 = orc.y & 0xffffu;
            const uint32_t w = x1 - x0;
            const uint32_t ty = y0 + k / w, tx = x0 + k % w;
            const uint64_t key = ((uint64_t)(ty * (uint32_t)gx + tx) << 32) | odb;
            keys[wave_base + s] = key;
            vals[wave_base + s] = (uint32_t)(wave_first + lo);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// radix sort pass: histogram -> scan (generic scan above) -> stable scatter.
// Block b owns keys [b*KPB, (b+1)*KPB); wave w of the block owns a contiguous quarter, processed in
// ITEMS rounds of 64 consecutive keys, so the in-block order is (wave, round, lane) = input order.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EGS_SORT_THREADS) void k_sort_hist(const uint64_t* __restrict__ keys, uint32_t R, int shift,
                                                                 uint32_t nblocks, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[EGS_SORT_BINS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * EGS_SORT_KPB + w * (EGS_SORT_KPB / 4);
#pragma unroll
    for (int r = 0; r < EGS_SORT_ITEMS; r++) {
        const uint32_t idx = base + r * 64 + lane;
        if (idx < R) atomicAdd(&h[(uint32_t)(keys[idx] >> shift) & (EGS_SORT_BINS - 1)], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];      // digit-major for the scan
}

__global__ __launch_bounds__(EGS_SORT_THREADS) void k_sort_scatter(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint64_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, uint32_t R, int shift, uint32_t nblocks, const uint32_t* __restrict__ hist_scanned) {
    __shared__ uint32_t cnt[4][EGS_SORT_BINS];          // per-wave running digit counts
    __shared__ uint32_t gbase[EGS_SORT_BINS];           // global base of digit d for this block
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; k++) cnt[k][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * EGS_SORT_KPB + w * (EGS_SORT_KPB / 4);
    const uint64_t lt = lanemask_lt();
    uint64_t key[EGS_SORT_ITEMS]; uint32_t val[EGS_SORT_ITEMS]; uint32_t rank[EGS_SORT_ITEMS];
#pragma unroll
    for (int r = 0; r < EGS_SORT_ITEMS; r++) {
        const uint32_t idx = base + r * 64 + lane;
        const bool ok = idx < R;
        key[r] = ok ? keys_in[idx] : ~0ull;
        val[r] = ok ? vals_in[idx] : 0u;
    }
#pragma unroll
    for (int r = 0; r < EGS_SORT_ITEMS; r++) {
        const uint32_t idx = base + r * 64 + lane;
        const bool ok = idx < R;
        const uint32_t d = (uint32_t)(key[r] >> shift) & (EGS_SORT_BINS - 1);
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < EGS_SORT_BITS; b++) {
            const uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        // peers = lanes (valid) holding my digit.  Lowest peer reads-and-bumps the wave's counter.
        const unsigned leader = (unsigned)__ffsll((unsigned long long)peers) - 1u;
        uint32_t start = 0;
        if (ok && lane == leader) { start = cnt[w][d]; cnt[w][d] = start + (uint32_t)__popcll(peers); }
        start = __shfl(start, ok ? leader : lane, 64);
        rank[r] = start + (uint32_t)__popcll(peers & lt);
    }
    __syncthreads();
    {   // thread d: wave-exclusive bases for digit d and the block's global base
        const unsigned d = threadIdx.x;
        uint32_t run = hist_scanned[(size_t)d * nblocks + blockIdx.x];
        gbase[d] = run;
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t c = cnt[k][d]; cnt[k][d] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < EGS_SORT_ITEMS; r++) {
        const uint32_t idx = base + r * 64 + lane;
        if (idx < R) {
            const uint32_t d = (uint32_t)(key[r] >> shift) & (EGS_SORT_BINS - 1);
            const uint32_t pos = cnt[w][d] + rank[r];
            keys_out[pos] = key[r]; vals_out[pos] = val[r];
        }
    }
}

__global__ __launch_bounds__(256) void k_tile_ranges(uint32_t R, const uint64_t* __restrict__ keys, uint2* __restrict__ ranges) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const uint32_t t = (uint32_t)(keys[i] >> 32);
    if (i == 0) ranges[t].x = 0;
    else { const uint32_t p = (uint32_t)(keys[i - 1] >> 32); if (p != t) { ranges[p].y = i; ranges[t].x = i; } }
    if (i == R - 1) ranges[t].y = R;
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
    if (n == 0) { return total ? hipMemsetAsync(total, 0, sizeof(uint64_t), s) : hipSuccess; }
    if (n <= EGS_SCAN_EPB) {
        hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(EGS_SCAN_THREADS), 0, s, in, out, n, inclusive, total);
        return hipGetLastError();
    }
    const size_t nb = (n + EGS_SCAN_EPB - 1) / EGS_SCAN_EPB;
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(EGS_SCAN_THREADS), 0, s, in, n, scratch);
    hipError_t e = egs_launch_scan_u32(scratch, scratch, nb, 0, scratch + ((nb + 63) & ~(size_t)63), nullptr, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(EGS_SCAN_THREADS), 0, s, in, out, n, inclusive, scratch, total);
    return hipGetLastError();
}

#define EGS_DBG(s)                                                                   \
    do { if (debug) { hipError_t e_ = hipStreamSynchronize(s); if (e_ != hipSuccess) return e_; \
                      e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } } while (0)

hipError_t egs_launch_binning(int P, int64_t R64, int W, int H, EgsGeomPtrs g, EgsBinPtrs b, EgsImgPtrs im,
                              hipStream_t s, int debug) {
    const int gx = (W + EGS_TILE - 1) / EGS_TILE, gy = (H + EGS_TILE - 1) / EGS_TILE;
    hipError_t e = hipMemsetAsync(im.ranges, 0, sizeof(uint2) * (size_t)gx * gy, s);
    if (e != hipSuccess) return e;
    if (R64 == 0 || P == 0) return hipSuccess;
    const uint32_t R = (uint32_t)R64;
    egs_prof_start(EGS_K_DUPLICATE, s);
    hipLaunchKernelGGL(k_duplicate, dim3((P + 255) / 256), dim3(256), 0, s, P, g.rec, g.rect, g.offsets, gx, b.keys_a, b.vals_a);
    egs_prof_stop(EGS_K_DUPLICATE, s);
    EGS_DBG(s);
    const uint32_t nblocks = (R + EGS_SORT_KPB - 1) / EGS_SORT_KPB;
    uint64_t* kin = b.keys_a; uint64_t* kout = b.keys_b; uint32_t* vin = b.vals_a; uint32_t* vout = b.vals_b;
    egs_prof_start(EGS_K_SORT, s);
    for (int pass = 0; pass < b.passes; pass++) {
        const int shift = pass * EGS_SORT_BITS;
        hipLaunchKernelGGL(k_sort_hist, dim3(nblocks), dim3(EGS_SORT_THREADS), 0, s, kin, R, shift, nblocks, b.hist);
        e = egs_launch_scan_u32(b.hist, b.hist, (size_t)nblocks * EGS_SORT_BINS, 0, b.spine, nullptr, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_sort_scatter, dim3(nblocks), dim3(EGS_SORT_THREADS), 0, s, kin, vin, kout, vout, R, shift,
                           nblocks, b.hist);
        EGS_DBG(s);
        uint64_t* tk = kin; kin = kout; kout = tk; uint32_t* tv = vin; vin = vout; vout = tv;
    }
    egs_prof_stop(EGS_K_SORT, s);
    egs_prof_start(EGS_K_RANGES, s);
    hipLaunchKernelGGL(k_tile_ranges, dim3((R + 255) / 256), dim3(256), 0, s, R, kin, im.ranges);
    egs_prof_stop(EGS_K_RANGES, s);
    EGS_DBG(s);
    return hipGetLastError();
}
