// bin_walk.h -- the instance-slot walk of the tile bucketing (binning.hip: k_bin_count, k_bin_scatter) as device code that a second
// translation unit can carry: preprocess.hip's k_preprocess_count runs the COUNT walk right behind the projection of the same
// Gaussians, with their rectangles, boxes and conics still in registers (no set-up loads, no launch of its own).
// Replaces upstream's duplicateWithKeys stage of the op called from /root/reference/gaussian_renderer/__init__.py:90-98
// (SURVEY.md section 8a rows a-5, a-6).
//
// The Gaussians are cut into BLOCKS of 256 consecutive ones (four 64-Gaussian groups -- the unit k_preprocess's per-block instance
// and hot counts are defined on); block B belongs to workgroup B % nblocks, so a run of heavy groups (the clones and splits
// densification appends at the end of the arrays are all on screen and close to the camera that asked for them) is spread over all
// workgroups instead of landing in the last few.  A workgroup (16 waves) takes `gpr` of its groups per round:
//   set-up  wave w < gpr owns one group: the rectangles' tile counts are scanned into per-Gaussian span starts and parked in LDS
//           with the rectangles, the depth words and (when culling) the ellipse parameters (bin_park_group);
//   deal    the round's instance slots are cut into units of 64 consecutive slots of ONE group and unit u goes to wave
//           u % 16, whichever wave set the group up -- the waves of a workgroup finish within one unit of each other however
//           uneven the rectangles are.  Inside a unit slot s is mapped back to its Gaussian through the owner map (below).
// `body(tile, gaussian_index, depth_bits)` runs once per instance.  The order in which a workgroup's instances reach a
// tile's bucket is not defined (its waves share the LDS cursors); the per-tile sort orders by (depth, index), which is unique.
#pragma once
#include "egs_common.h"
#include "blend_common.h"

namespace {

__device__ __forceinline__ unsigned lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ uint64_t lanemask_lt() {
    const unsigned lane = lane_id();
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t n = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v += n;
    }
    return v;
}

// Workgroup b runs on XCD b % 8 (observed, used for speed only).  The bucketed array interleaves, inside every tile's
// region, the slices of consecutive table columns; giving each XCD a contiguous run of columns lets its private L2
// merge the 8-byte stores of neighbouring slices into full lines before they leave for HBM.  `first`: workgroups of the grid in
// front of the bucketing ones (a multiple of 8).
__device__ __forceinline__ unsigned bin_logical_block(unsigned nblocks, unsigned first = 0u) {
    const unsigned per = (nblocks + 7) / 8, b = blockIdx.x - first;
    return (b % 8) * per + b / 8;                                   // >= nblocks for the padding blocks of the grid
}

// Tile culling (`cull`): the rectangle is the reference's 3-sigma bounding square, so many of its tiles hold no pixel the
// splat can reach with alpha >= 1/255 (corners of elongated splats, faint splats).  Such an instance can never
// contribute -- the reference skips it at every pixel -- so dropping it here changes no output bit; it only shortens the
// sort and the lists the blend kernels scan (config C: 2.94M -> 1.87M instances).  The test is the exact, conservative
// ellipse-vs-block test the blend kernels apply per 8x8 quadrant (blend_common.h), on the whole 16x16 tile; each slot-lane
// reads its Gaussian's prepared ellipse parameters from the LDS block the set-up wave wrote.
#ifdef EGS_BIN_TIMING
// measurement builds (tools/bin_phases.py): every wave of every workgroup of k_bin_count stamps s_memtime at its phase boundaries
__device__ unsigned long long egs_bin_stamps[512 * 16 * 6];
__device__ unsigned long long egs_bin_unit_stamps[8 * 6];
#define BIN_USTAMP(ph) do { __builtin_amdgcn_sched_barrier(0); if (!need_depth && threadIdx.x == 0 && bid == 7 && ucount < 8) egs_bin_unit_stamps[ucount * 6 + (ph)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define BIN_STAMP(ph) do { __builtin_amdgcn_sched_barrier(0); if (!need_depth && (threadIdx.x & 63) == 0 && bid < 512) egs_bin_stamps[(bid * 16 + (threadIdx.x >> 6)) * 6 + (ph)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define BIN_STAMP(ph)
#define BIN_USTAMP(ph)
#endif
#define EGS_BIN_WAVES (EGS_BIN_THREADS / 64)
// LDS behind the per-tile counters, for a round of `gpr` groups (words): span starts, rectangles, depth words, 16 unit counts +
// 16 group totals, then (culling only, 16-byte aligned) two float4 per Gaussian.
// + per group the owner map of the slot walk (see bin_walk_round): 64 packed (span start << 6 | lane) words of the Gaussians that have
// tiles, a 4096-bit map of the slots at which a span starts and 64 prefix counts of it.
#define BIN_MAP_SLOTS 4096
__host__ __device__ inline size_t bin_round_words(int gpr, bool cull, bool map) {
    return (size_t)gpr * 64 * 4 + 32 + (cull ? (size_t)gpr * 64 * 8 : 0) + (map ? (size_t)gpr * 256 : 0);
}

// The round's LDS block, carved (bin_round_words).
struct BinRound {
    uint32_t* span;      // [gpr * 64] exclusive slot offset inside the group
    uint2* rcs;          // [gpr * 64] walked tile rectangle
    uint32_t* dbs;       // [gpr * 64] depth words
    uint32_t* units;     // [16] 64-slot units per group, [16] slots per group
    float4* stage;       // [gpr * 64][2] ellipse parameters (culling)
    uint32_t* packed;    // [gpr * 64] owner map: (span start << 6 | lane) of the r-th Gaussian that has tiles
    uint32_t* bm;        // [gpr][128] one bit per slot < BIN_MAP_SLOTS at which a span starts
    uint32_t* bmpre;     // [gpr][64] span starts before slot 64 u
};
__device__ __forceinline__ BinRound bin_round_carve(uint32_t* round_lds, int gpr, bool cull) {
    BinRound r;
    r.span = round_lds;
    r.rcs = reinterpret_cast<uint2*>(r.span + gpr * 64);
    r.dbs = r.span + gpr * 64 * 3;
    r.units = r.dbs + gpr * 64;
    r.stage = reinterpret_cast<float4*>(r.units + 32);
    r.packed = reinterpret_cast<uint32_t*>(r.stage + (cull ? (size_t)gpr * 64 * 2 : 0));
    r.bm = r.packed + gpr * 64;
    r.bmpre = r.bm + gpr * 128;
    return r;
}

// Group of 64 Gaussians that local group l (0, 1, 2, ... in the order a workgroup takes them) of workgroup `bid` stands for: block
// B = bid + nblocks * (l / 4), group 4 B + l % 4.
__device__ __forceinline__ unsigned bin_group_of(unsigned bid, unsigned nblocks, unsigned l) { return 4u * (bid + nblocks * (l >> 2)) + (l & 3u); }
// local groups of the busiest workgroup
__host__ __device__ inline unsigned bin_groups_per_block(unsigned P, unsigned nblocks) {
    const unsigned blocks256 = (P + 255u) / 256u;
    return 4u * ((blocks256 + nblocks - 1u) / nblocks);
}

// Set-up of one group by ONE wave (w < gpr): `cnt` tiles of the reference's rectangle `rc_l` (0: culled / absent), record words r0..r2.
// Owner map (which Gaussian of the group does slot s belong to?).  A 6-step binary search over the span starts by ds_bpermute was
// 1 000 of the 2 400 cycles a wave spends per 64-slot unit (tools/bin_phases.py); instead the set-up wave leaves, per group,
//   packed[r]   (span start << 6 | lane) of the r-th Gaussian that has tiles,
//   bm          one bit per slot < BIN_MAP_SLOTS at which a span starts,      bmpre[u] = span starts before slot 64 u,
// and a slot's owner is packed[bmpre[u] + popcount(bm[u] & bits up to the slot) - 1]: two LDS round trips, the first at a uniform
// address.  Units beyond the map (a group covering more than 4 096 tiles) keep the search, and so does everything when the per-tile
// counters of a large image leave no room for the map (`use_map`).
__device__ __forceinline__ void bin_park_group(const BinRound& L, unsigned w, unsigned lane, bool have, uint32_t cnt, uint2 rc_l,
                                               const float4& r0, const float4& r1, const float4& r2, bool need_depth, bool cull, bool use_map) {
    if (cull && cnt) {
        // Only the tiles of the record's alpha >= 1/255 box (egs_common.h) can pass the ellipse test: walk the box's tile
        // rectangle cut to the reference's instead of the reference's 3-sigma rectangle.  On a trained scene most splats are
        // faint -- the box is a fraction of the rectangle, or empty (opacity < 1/255: nothing to walk): 4.06 M rectangle slots
        // -> 1.24 M there.  `tiles_touched` and R stay the reference's; with culling off the full rectangle is walked.
        const uint32_t bx = __float_as_uint(r2.z), by = __float_as_uint(r2.w);
        const uint32_t px0 = bx & EGS_BOX_MASK, px1 = (bx >> 16) & EGS_BOX_MASK, py0 = by & EGS_BOX_MASK, py1 = (by >> 16) & EGS_BOX_MASK;
        const uint32_t x0 = max(rc_l.x & 0xffffu, px0 / EGS_TILE), x1 = min(rc_l.x >> 16, px1 / EGS_TILE + 1u);
        const uint32_t y0 = max(rc_l.y & 0xffffu, py0 / EGS_TILE), y1 = min(rc_l.y >> 16, py1 / EGS_TILE + 1u);
        const bool some = px0 <= px1 && py0 <= py1 && x0 < x1 && y0 < y1;
        cnt = some ? (x1 - x0) * (y1 - y0) : 0u;
        rc_l = make_uint2(x0 | (x1 << 16), y0 | (y1 << 16));
    }
    const uint32_t incl = wave_incl_scan(cnt);
    L.span[w * 64 + lane] = have ? incl - cnt : 0xffffffffu;         // invalid lanes sort to the end
    if (cnt) {
        L.rcs[w * 64 + lane] = rc_l;
        if (need_depth) L.dbs[w * 64 + lane] = __float_as_uint(r2.y);
        if (cull) { L.stage[2 * (w * 64 + lane)] = r0; L.stage[2 * (w * 64 + lane) + 1] = egs_ellipse_prep(r0.z, r0.w, r1.x, r1.y); }
    }
    if (lane == 63) { L.units[w] = (incl + 63u) >> 6; L.units[16 + w] = incl; }
    if (use_map) {   // owner map of this group (one wave: its LDS operations execute in order)
        const uint32_t excl_l = incl - cnt;
        const uint64_t nzm = __ballot(cnt != 0);
        if (cnt) L.packed[w * 64 + __popcll(nzm & lanemask_lt())] = (excl_l << 6) | lane;       // (a group has < 2^24 slots)
        L.bm[w * 128 + lane] = 0u; L.bm[w * 128 + 64 + lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (cnt && excl_l < BIN_MAP_SLOTS) atomicOr(&L.bm[w * 128 + (excl_l >> 5)], 1u << (excl_l & 31u));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        const uint32_t pc = (uint32_t)__popc(*(volatile uint32_t*)&L.bm[w * 128 + 2 * lane]) + (uint32_t)__popc(*(volatile uint32_t*)&L.bm[w * 128 + 2 * lane + 1]);
        L.bmpre[w * 64 + lane] = wave_incl_scan(pc) - pc;
    }
}

// The walk of one round (every wave; the set-up is behind a workgroup barrier): local groups g0 .. g0 + gpr - 1 of workgroup `bid`.
template <typename Body>
__device__ __forceinline__ void bin_walk_round(const BinRound& L, unsigned bid, unsigned nblocks, unsigned g0, int gpr, int gx, bool need_depth, bool cull,
                                               bool use_map, int W, int H, Body body) {
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t un = (int)lane < gpr ? L.units[lane] : 0u, tot = (int)lane < gpr ? L.units[16 + lane] : 0u;
    const uint32_t uincl = wave_incl_scan(un);
    const uint32_t n_units = __shfl(uincl, 63, 64);
    int ucount = -1; (void)ucount;
    for (uint32_t u = w; u < n_units; u += EGS_BIN_WAVES) {
        ucount++;
        BIN_USTAMP(0);
        const int k = __popcll(__ballot(uincl <= u && (int)lane < gpr));       // the group unit u falls in (uniform)
        const uint32_t s = ((u - (__shfl(uincl, k, 64) - __shfl(un, k, 64))) << 6) + lane;
        const uint32_t total = __shfl(tot, k, 64);
        const uint32_t uu = s >> 6;                                // unit inside the group (uniform)
        int lo; uint32_t ost;
        if (use_map && uu < BIN_MAP_SLOTS / 64) {
            const uint64_t B = *reinterpret_cast<const uint64_t*>(&L.bm[k * 128 + 2 * uu]);
            const uint32_t R = L.bmpre[k * 64 + uu] + (uint32_t)__popcll(B & (lanemask_lt() | (1ull << lane))) - 1u;
            const uint32_t pk = L.packed[k * 64 + (R & 63u)];      // (lanes past the group's last slot read a valid word and are masked below)
            lo = (int)(pk & 63u); ost = pk >> 6;
        } else {
            const uint32_t excl = L.span[k * 64 + lane];
            lo = 0;                                                  // last lane whose span starts at or before s
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                const int probe = lo + step;
                const uint32_t st = __shfl(excl, probe & 63, 64);
                if (probe < 64 && st <= s) lo = probe;
            }
            ost = __shfl(excl, lo, 64);                            // (all lanes take part: outside the branch)
        }
        BIN_USTAMP(1);
        if (s < total) {
            const uint32_t kk = s - ost;
            const uint2 orc = L.rcs[k * 64 + lo];
            const uint32_t x0 = orc.x & 0xffffu, x1 = orc.x >> 16, y0 = orc.y & 0xffffu;
            const uint32_t wd = x1 - x0;
            uint32_t row = (uint32_t)((float)kk * __builtin_amdgcn_rcpf((float)wd));      // k < 2^24: off by at most one
            uint32_t col = kk - row * wd;
            if ((int)col < 0) { row--; col += wd; }
            if (col >= wd) { row++; col -= wd; }
            const uint32_t ty = y0 + row, tx = x0 + col;
            bool keep = true;
            BIN_USTAMP(2);
            if (cull) {
                const float4 e0 = L.stage[2 * (k * 64 + lo)], e1 = L.stage[2 * (k * 64 + lo) + 1];   // (x, y, qa, qb), (qc, need, sy, sx)
                keep = egs_ellipse_hits_prepped(e0, e1, tx * EGS_TILE, min(tx * EGS_TILE + EGS_TILE - 1, (uint32_t)W - 1),
                                                ty * EGS_TILE, min(ty * EGS_TILE + EGS_TILE - 1, (uint32_t)H - 1));
            }
            BIN_USTAMP(3);
            if (keep) body(ty * (uint32_t)gx + tx, bin_group_of(bid, nblocks, g0 + (unsigned)k) * 64u + (unsigned)lo, need_depth ? L.dbs[k * 64 + lo] : 0u);
        }
        BIN_USTAMP(4);
    }
}

// Both passes of binning.hip: the set-up loads what k_preprocess left in the geometry buffer.
template <typename Body>
__device__ __forceinline__ void for_each_instance(unsigned bid, unsigned nblocks, int gpr, int P, const uint32_t* __restrict__ tiles_touched,
                                                  const uint2* __restrict__ rect, const float4* __restrict__ rec, int gx,
                                                  bool need_depth, bool cull, bool use_map, int W, int H, uint32_t* __restrict__ round_lds, Body body) {
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const BinRound L = bin_round_carve(round_lds, gpr, cull);
    const unsigned groups = ((unsigned)P + 63u) / 64u;
    const unsigned per_block = bin_groups_per_block((unsigned)P, nblocks);     // local groups of the busiest workgroup
    for (unsigned g0 = 0; g0 < per_block; g0 += (unsigned)gpr) {
        if (g0) __syncthreads();                                       // the previous round's readers are done
        if ((int)w < gpr) {
            const unsigned j = bin_group_of(bid, nblocks, g0 + w);
            const int i = (int)(j * 64u + lane);
            const bool have = g0 + w < per_block && j < groups && i < P;
            // Everything a Gaussian contributes is requested at once -- the rectangle and the record words do not wait for the tile
            // count to come back (a culled Gaussian's words are loaded for nothing; the set-up phase was two dependent round trips
            // of ~2 us each in a workgroup that does nothing else meanwhile, tools/bin_phases.py).
            const int il = have ? i : 0;
            const uint32_t cnt = have ? tiles_touched[i] : 0u;
            const uint2 rc_l = rect[il];
            float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
            if (cull) { r0 = rec[(size_t)il * EGS_SPLAT_REC_F4]; r1 = rec[(size_t)il * EGS_SPLAT_REC_F4 + 1]; r2 = rec[(size_t)il * EGS_SPLAT_REC_F4 + 2]; }
            else if (need_depth) r2.y = rec[(size_t)il * EGS_SPLAT_REC_F4 + 2].y;
            bin_park_group(L, w, lane, have, cnt, rc_l, r0, r1, r2, need_depth, cull, use_map);
        }
        BIN_STAMP(1);
        __syncthreads();
        BIN_STAMP(2);
        bin_walk_round(L, bid, nblocks, g0, gpr, gx, need_depth, cull, use_map, W, H, body);
    }
}

// The count pass's epilogue: the workgroup's row of the [tile][workgroup] table and its share of every scan chunk's sum (2048 entries =
// 2048 / stride whole rows), added to one of EGS_BIN_GROUPS partial accumulators: the scan then needs no reduction pass of its own (one
// launch less; the atomics return nothing).  `sums` must have been ZERO before the first workgroup of the launch got here.
__device__ __forceinline__ void bin_flush_counts(const uint32_t* hist, int n_tiles, unsigned bid, uint32_t stride, uint32_t* __restrict__ table,
                                                 uint32_t* __restrict__ chunk_sum, unsigned acc_group) {
    for (int t = threadIdx.x; t < n_tiles; t += EGS_BIN_THREADS) table[(size_t)t * stride + bid] = hist[t];    // tile-major
    const int rpc = 2048 / (int)stride, n_chunks = (n_tiles + rpc - 1) / rpc;
    uint32_t* sums = chunk_sum + (size_t)(acc_group % EGS_BIN_GROUPS) * n_chunks;
    for (int c = threadIdx.x; c < n_chunks; c += EGS_BIN_THREADS) {
        uint32_t sum = 0;
        for (int t = c * rpc; t < min((c + 1) * rpc, n_tiles); t++) sum += hist[t];
        if (sum) atomicAdd(&sums[c], sum);
    }
}

}  // namespace
