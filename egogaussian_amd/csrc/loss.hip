// loss.hip -- fused training image loss, forward and backward:
//     loss = (1 - lambda) * mean|x - y| + lambda * (1 - mean(SSIM_map(x, y)))
// as computed by the reference with ~6 depthwise 11x11 convolutions and ~25 elementwise kernels per step:
//     l1_loss, ssim           /root/reference/utils/loss_utils.py:57-107  (11x11 Gaussian window, sigma 1.5, zero padding,
//                                                                          C1 = 0.01^2, C2 = 0.03^2)
//     loss assembly, hand-mask gradient gate   /root/reference/trainers/train_static.py:91-95
// Forward: the five windowed moments (E[x], E[y], E[x^2], E[y^2], E[xy]) come from a separable 11-tap blur, the SSIM
// map value is summed per wave, and the three partial derivatives of the map (w.r.t. E[x], E[x^2], E[xy]) are stored.
// Backward: the same machinery blurs those three maps (the window is symmetric, so the adjoint of the blur is the blur)
// and adds the L1 term and the optional per-pixel gradient gate.  8 B in + 12 B out per pixel-channel forward,
// 20 B in + 4 B out backward (+ 10/SR halo rows and 10/54 halo columns re-read).
#include "egs_common.h"
#include "backward_prologue.h"
#include "loss_window.h"

// Mapping (wave64 streaming, no workgroup barriers): a wave owns a strip of SW = 54 output columns x SR output rows of
// one channel.  Lane L is image column  strip_x0 - 5 + L  (5 halo columns each side) and walks DOWN the rows:
//   * vertical 11-tap blur in registers: the last 11 rows of the per-pixel products (x, y, x^2, y^2, xy -- or the three
//     derivative maps in the backward) live in a register window that is rotated by full unrolling, never moved;
//   * horizontal 11-tap blur across lanes through a wave-private LDS row (one write, 11 broadcast-free reads per map);
//   * rows are read from HBM exactly once per strip (+10 halo rows per SR) with fully coalesced 256-byte accesses.
// The old tiling (16x16 outputs per 256-thread workgroup, 26x26 inputs staged in LDS, three __syncthreads) spent its time
// waiting: 3 rounds of latency-bound workgroups, 30 us + 25 us at 3x540x960; this one is bound by VALU/LDS issue.
#define HALO 5
#define SW 54                 // useful columns per wave (64 lanes - 2 * HALO)
#ifdef EGS_LOSS_SR
#define SR EGS_LOSS_SR        // (tuning builds)
#else
#define SR 15                 // output rows per wave (3 x 540 x 960: 1944 waves, just under 2 per SIMD)
#endif
#ifndef WPB
#define WPB 2                 // waves per workgroup (independent)
#endif
#ifndef EGS_LOSS_ABL
#define EGS_LOSS_ABL 0        // ablation builds (wrong results): 1 no map stores, 2 no horizontal blur, 4 no SSIM arithmetic, 8 no loads
#endif

namespace {

// Forward: the maps are blurred in PAIRS, (E[x], E[y]) and (E[x^2 + y^2], E[xy]).  A pair lives in a 64-bit register pair and every tap is one v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 -- the same IEEE
// operations per component in the same order as the scalar formulation, at half the instruction count -- and one ds_write_b64 /
// ds_read_b64 in the horizontal pass.  What bounds these kernels is the length of ONE wave's instruction stream: 1944 waves = 1.9 per
// SIMD, each a chain of dependent steps (s_memtime stamps, tools/loss_rows.py: ~1300 cycles per output row = vertical blur 200, LDS
// round trip + horizontal blur 400, SSIM arithmetic with its two IEEE divisions + stores 700), every instruction of any kind costs
// the wave >= 4 cycles and ~9 when it depends on the previous one.  Ablations at 3x540x960 (of 19.6 us): no map stores -2.9, no
// horizontal pass -4.2, no SSIM arithmetic -2.1, no loads +0.2; rows per wave 8 / 12 / 15 / 20: 18.5 / 17.7 / 18.3 / 19.9 us.
// Pairs + one 11-row buffer refilled in place + 32-bit indices: 2277 -> 1395 static VALU, 144 -> 124 VGPRs, 19.6 -> 18.0 us.  The same
// rewrite of the backward (1812 -> 1078 VALU) ran 2 us SLOWER and was dropped: its per-row chain is short, and the clamps and
// selects of the branch-free loads lengthen the scalar part of the stream.
typedef float v2f __attribute__((ext_vector_type(2)));
#ifdef EGS_LOSS_TIMING
// measurement builds: wave 0 of workgroup 0 stamps s_memtime around the phases of every row step (tools/loss_rows.py)
__device__ unsigned long long egs_loss_stamps[4 * 64];
__device__ __forceinline__ void loss_stamp(bool on, int row, int phase) {
    __builtin_amdgcn_sched_barrier(0);
    if (on && row >= 0 && row < 64) egs_loss_stamps[4 * row + phase] = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
}
#define LOSS_STAMP(row, phase) loss_stamp(c.stamp, row, phase)
#else
#define LOSS_STAMP(row, phase)
#endif
__device__ __forceinline__ v2f pk_fma(float w, v2f a, v2f c) { return __builtin_elementwise_fma((v2f)(w), a, c); }

// Horizontal blur of one pair per lane (= per column) through the wave's LDS row: out = sum_k w[k] v[lane - 5 + k].
// Lanes 0..4 and 59..63 read the row's zero padding and return values nobody uses.
__device__ __forceinline__ void hblur_issue(v2f v, v2f* row /* [80], lane L at [8 + L] */, unsigned lane) { row[8 + lane] = v; }
__device__ __forceinline__ v2f hblur_collect(const v2f* row, unsigned lane) {
    v2f r[11];
#pragma unroll
    for (int k = 0; k < 11; k++) r[k] = row[3 + lane + k];                     // r[k] = v[lane - 5 + k]
    v2f acc = (v2f)(kwin(0)) * (r[0] + r[10]);                   // the window is symmetric: 6 multiplies instead of 11
    acc = pk_fma(kwin(1), r[1] + r[9], acc); acc = pk_fma(kwin(2), r[2] + r[8], acc);
    acc = pk_fma(kwin(3), r[3] + r[7], acc); acc = pk_fma(kwin(4), r[4] + r[6], acc);
    return pk_fma(kwin(5), r[5], acc);
}
// Vertical blur over a window of the last 11 rows (slot = row % 11) when the newest row sits in slot NEWEST (compile-time):
// out = sum_k w[k] row[NEWEST + 1 + k].
template <int NEWEST>
__device__ __forceinline__ v2f vblur(const v2f (&w)[11]) {
    v2f acc = (v2f)(0.f);
#pragma unroll
    for (int k = 0; k < 11; k++) acc = pk_fma(kwin(k), w[(NEWEST + 1 + k) % 11], acc);
    return acc;
}
struct FwdCtx {
    int H, W, gx, y_first, y_end; size_t plane; bool col_ok, col_out; unsigned lane;
    float* dm_dmu1; float* dm_dexx; float* dm_dexy; v2f* rows;   // rows: [2][80] pairs
    float l1, sm; bool stamp; int row0;
};

// One input row enters (slot NEWEST); if 11 rows are in, the output row 5 above it leaves.
template <int NEWEST>
__device__ __forceinline__ void fwd_step(FwdCtx& c, v2f (&w01)[11], v2f (&w23)[11], int y_in, float u, float v) {
    if (!c.col_ok || y_in < 0 || y_in >= c.H) { u = 0.f; v = 0.f; }    // outside the image: the window's zero padding (the loads were clamped)
    // SSIM uses the two variances only as their sum, so E[x^2] and E[y^2] are blurred together: four maps, not five
    LOSS_STAMP(y_in - c.row0, 0);
    w01[NEWEST] = v2f{u, v}; w23[NEWEST] = v2f{fmaf(u, u, v * v), u * v};
    const int y_out = y_in - HALO;
    if (y_out < c.y_first || y_out >= c.y_end) return;                  // wave-uniform
    const v2f vb01 = vblur<NEWEST>(w01), vb23 = vblur<NEWEST>(w23);
    LOSS_STAMP(y_in - c.row0, 1);
    v2f hb01, hb23;
    if (EGS_LOSS_ABL & 2) { hb01 = vb01; hb23 = vb23; }
    else {
        // both rows are written, then all reads are issued together: one LDS round trip per image row
        hblur_issue(vb01, c.rows, c.lane); hblur_issue(vb23, c.rows + 80, c.lane);
        __builtin_amdgcn_wave_barrier();
        hb01 = hblur_collect(c.rows, c.lane); hb23 = hblur_collect(c.rows + 80, c.lane);
        __builtin_amdgcn_wave_barrier();
    }
    LOSS_STAMP(y_in - c.row0, 2);
    const float mu1 = hb01.x, mu2 = hb01.y, exx_eyy = hb23.x, exy = hb23.y;
    if (c.col_out) {
        const float C1 = 0.0001f, C2 = 0.0009f;
        const float s12 = exy - mu1 * mu2;
        const float A = 2.f * mu1 * mu2 + C1, B = 2.f * s12 + C2, D = mu1 * mu1 + mu2 * mu2 + C1, E = (exx_eyy - (D - C1)) + C2;
        const size_t p = c.plane + (size_t)y_out * c.W + c.gx;
        if (EGS_LOSS_ABL & 4) {
            c.dm_dmu1[p] = A; c.dm_dexx[p] = B; c.dm_dexy[p] = D + E; c.sm += A;
            return;
        }
        // 1 / (D E) as v_rcp_f32 + one Newton step (<= 1 ulp) and the second quotient as a product: the two IEEE division sequences were
        // ~20 dependent instructions of a wave whose instruction stream IS the launch (16.8 -> 15.8 us at 3x540x960); -DEGS_LOSS_IEEE_DIV: A/B
#ifdef EGS_LOSS_IEEE_DIV
        const float invDE = 1.f / (D * E);
#else
        const float de_ = D * E; float invDE = __builtin_amdgcn_rcpf(de_); invDE = fmaf(fmaf(-de_, invDE, 1.f), invDE, invDE);
#endif
        const float sm = A * B * invDE;
        // partial derivatives of the map holding the other windowed moments fixed
        if (EGS_LOSS_ABL & 1) { c.sm += (2.f * mu2 * (B - A)) * invDE - sm * (2.f * mu1 * (E - D)) * invDE + -sm / E + 2.f * A * invDE; }
        else {
            c.dm_dmu1[p] = (2.f * mu2 * (B - A)) * invDE - sm * (2.f * mu1 * (E - D)) * invDE;
#ifndef EGS_LOSS_IEEE_DIV
            c.dm_dexx[p] = -sm * (D * invDE);
#else
            c.dm_dexx[p] = -sm / E;
#endif
            c.dm_dexy[p] = 2.f * A * invDE;
        }
        constexpr int CENTRE = (NEWEST + 11 - HALO) % 11;               // the row that is leaving sits 5 slots behind the newest
        c.l1 += fabsf(w01[CENTRE].x - w01[CENTRE].y);
        c.sm += sm;
    }
    LOSS_STAMP(y_in - c.row0, 3);
}

// Workgroup b runs on XCD b % 8 (blend_common.h).  Neighbouring strips share their halo -- 10 of 64 columns, 10 of 25 rows -- and the
// 128-byte lines a misaligned 256-byte row segment straddles; dealt in index order they land on eight different L2s and every strip
// fetches its whole footprint from HBM (43.6 MB for 12.4 MB of image, profiles/r4_loss_traffic.md).  Logical workgroup
// (b % 8) * per + b / 8 gives each XCD one contiguous run of strips, so that its L2 serves the shared lines.  -> logical index, or
// `main` and beyond for the padding workgroups of the rounded-up grid.
__device__ __forceinline__ unsigned loss_logical_block(unsigned b, unsigned main) {
#ifdef EGS_LOSS_NO_REMAP                     // A/B switch (tools/loss_time.py): index order
    (void)main; return b;
#else
    const unsigned per = (main + 7u) / 8u;
    return (b & 7u) * per + (b >> 3);
#endif
}

// grid: 8 * ceil(C * ceil(strips_x * strips_y / WPB) / 8) (1-D, see loss_logical_block); a wave = one strip
// SIDE: the last `side_jobs` workgroups carry the preparation of the rasterizer's backward blend of the same frame (as k_l1_ssim_backward<true>
// does) -- for a training step whose blend computes the loss gradient itself (render_bwd.hip, LG) and therefore has no loss-backward launch.
template <bool SIDE>
__global__ __launch_bounds__(64 * WPB) void k_l1_ssim_forward(int H, int W, int strips_x, int strips_y, const float* __restrict__ img,
                                                               const float* __restrict__ gt, float* __restrict__ partial,
                                                               float* __restrict__ dm_dmu1, float* __restrict__ dm_dexx,
                                                               float* __restrict__ dm_dexy, unsigned per_plane, unsigned main_wgs,
                                                               unsigned side_jobs, EgsPrologueArgs side) {
    __shared__ v2f lds[WPB][2 * 80];
    if (SIDE) {
        __shared__ EgsOrderLds order_lds;
        if (blockIdx.x >= gridDim.x - side_jobs) { egs_prologue_job<64 * WPB>(side, blockIdx.x - (gridDim.x - side_jobs), side_jobs, order_lds); return; }
    }
    const unsigned lane = threadIdx.x & 63, wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave id, kept scalar
    const unsigned rel = loss_logical_block(blockIdx.x, main_wgs);
    if (rel >= main_wgs) return;
    const unsigned plane_z = rel / per_plane;
    const int strip = (int)(rel - plane_z * per_plane) * WPB + (int)wv;
    if (strip >= strips_x * strips_y) return;
    for (int k = lane; k < 2 * 80; k += 64) lds[wv][k] = (v2f)(0.f);   // the padding words stay zero
    __builtin_amdgcn_wave_barrier();
    FwdCtx c;
    c.H = H; c.W = W; c.lane = lane; c.dm_dmu1 = dm_dmu1; c.dm_dexx = dm_dexx; c.dm_dexy = dm_dexy;
    c.rows = lds[wv]; c.plane = (size_t)plane_z * H * W; c.l1 = 0.f; c.sm = 0.f;
    const int sx = strip % strips_x, sy = strip / strips_x;
    c.gx = sx * SW - HALO + (int)lane;
    c.col_ok = c.gx >= 0 && c.gx < W;
    c.col_out = c.col_ok && lane >= HALO && lane < HALO + SW;
    c.y_first = sy * SR; c.y_end = min(c.y_first + SR, H);
    c.stamp = false; c.row0 = c.y_first - HALO;
#ifdef EGS_LOSS_TIMING
    c.stamp = rel == EGS_LOSS_TIMING && wv == 0;
#endif
    v2f w01[11], w23[11];
#pragma unroll
    for (int k = 0; k < 11; k++) { w01[k] = (v2f)(0.f); w23[k] = (v2f)(0.f); }
    // Input rows y_first - 5 .. y_end + 4, eleven per trip so that every window slot index is a compile-time constant; the next
    // trip's 22 loads are in flight while this trip computes (a row-by-row load would expose a full memory latency per row:
    // 28 rows x ~1 us).  A row's address is a wave-uniform base (scalar registers, advanced by scalar adds) plus the lane's column,
    // clamped into the image -- lanes beyond the image edge load a valid word and zero it when it enters the window; rows beyond
    // the image are skipped by a uniform branch.  (Per-lane 64-bit addresses and selects cost ~6 VALU instructions per load.)
    const unsigned col = (unsigned)min(max(c.gx, 0), W - 1);
    const float* __restrict__ img_p = img + c.plane; const float* __restrict__ gt_p = gt + c.plane;
    auto load_row = [&](int y, float& u, float& v) {
        const unsigned ro = (unsigned)min(max(y, 0), H - 1) * (unsigned)W;   // wave-uniform; rows outside the image are zeroed when they enter the window
        if (EGS_LOSS_ABL & 8) { u = (float)(col & 255) * 0.003f; v = (float)(col & 127) * 0.007f; }
        else { u = img_p[ro + col]; v = gt_p[ro + col]; }                     // (one plane has fewer than 2^32 elements: check_dims)
    };
    // one buffer of eleven rows, refilled in place: as soon as row y0 + K has entered the window its registers receive row
    // y0 + 11 + K -- every load is eleven row-steps ahead of its use with 22 registers instead of 44 (4 waves per SIMD instead of 2)
    float cu[11], cv[11];
#pragma unroll
    for (int k = 0; k < 11; k++) load_row(c.y_first - HALO + k, cu[k], cv[k]);
    for (int y0 = c.y_first - HALO; y0 < c.y_end + HALO; y0 += 11) {
#define FSTEP(K) fwd_step<K>(c, w01, w23, y0 + K, cu[K], cv[K]); load_row(y0 + 11 + K, cu[K], cv[K])
        FSTEP(0); FSTEP(1); FSTEP(2); FSTEP(3); FSTEP(4); FSTEP(5); FSTEP(6); FSTEP(7); FSTEP(8); FSTEP(9); FSTEP(10);
#undef FSTEP
    }
    float l1 = c.l1, sm = c.sm;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { l1 += __shfl_xor(l1, d, 64); sm += __shfl_xor(sm, d, 64); }
    if (lane == 0) {
        const size_t b = (size_t)plane_z * strips_x * strips_y + strip;
        partial[2 * b] = l1; partial[2 * b + 1] = sm;
    }
}

// (backward kernel: scalar formulation)  Horizontal blur of NV values per lane (= per column) through the wave's LDS rows: out[m] = sum_k w[k] v[m][lane - 5 + k].
// All NV rows are written, then all reads are issued together: one LDS round trip per image row, not one per map.
// Lanes 0..4 and 59..63 read the rows' zero padding and return values nobody uses.
template <int NV>
__device__ __forceinline__ void hblur_s(const float (&v)[NV], float (&out)[NV], float* rows /* [NV][80], lane L at [8 + L] */, unsigned lane) {
#pragma unroll
    for (int m = 0; m < NV; m++) rows[m * 80 + 8 + lane] = v[m];
    __builtin_amdgcn_wave_barrier();
    float r[NV][11];
#pragma unroll
    for (int m = 0; m < NV; m++)
#pragma unroll
        for (int k = 0; k < 11; k++) r[m][k] = rows[m * 80 + 3 + lane + k];          // r[m][k] = v[m][lane - 5 + k]
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < NV; m++) {
        float acc = kwin(0) * (r[m][0] + r[m][10]);             // the window is symmetric: 6 multiplies instead of 11
        acc = fmaf(kwin(1), r[m][1] + r[m][9], acc); acc = fmaf(kwin(2), r[m][2] + r[m][8], acc);
        acc = fmaf(kwin(3), r[m][3] + r[m][7], acc); acc = fmaf(kwin(4), r[m][4] + r[m][6], acc);
        out[m] = fmaf(kwin(5), r[m][5], acc);
    }
}

template <int NV>
struct Window {                                          // the last 11 rows of NV per-pixel values, slot = row % 11
    float v[NV][11];
};

// Vertical blur over the window when the newest row sits in slot `newest` (compile-time): out = sum_k w[k] row[newest+1+k].
template <int NV, int NEWEST>
__device__ __forceinline__ void vblur_s(const Window<NV>& w, float (&out)[NV]) {
#pragma unroll
    for (int m = 0; m < NV; m++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) acc = fmaf(kwin(k), w.v[m][(NEWEST + 1 + k) % 11], acc);
        out[m] = acc;
    }
}

struct BwdCtx {
    int H, W, gx, y_first, y_end; size_t plane; bool col_ok, col_out; unsigned lane;
    const float* img; const float* gt; const float* m0; const float* m1; const float* m2; const float* gate; float* dimg; float* rows;
    float w_l1, w_ssim, up;
};

template <int NEWEST>
__device__ __forceinline__ void bwd_step(BwdCtx& c, Window<3>& w, int y_in, float a, float b, float d, float x, float y) {
    w.v[0][NEWEST] = a; w.v[1][NEWEST] = b; w.v[2][NEWEST] = d;
    const int y_out = y_in - HALO;
    if (y_out < c.y_first || y_out >= c.y_end) return;                  // wave-uniform
    float vb[3];
    vblur_s<3, NEWEST>(w, vb);
    float hbv[3];
    hblur_s<3>(vb, hbv, c.rows, c.lane);
    const float ba = hbv[0], bb = hbv[1], bd = hbv[2];
    if (c.col_out) {
        const size_t p = c.plane + (size_t)y_out * c.W + c.gx;
        const float diff = x - y;
        const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        float g = c.w_l1 * sgn - c.w_ssim * (ba + 2.f * x * bb + y * bd);     // d loss / d x ; loss uses (1 - mean SSIM)
        g *= c.up;
        if (c.gate) g *= c.gate[(size_t)y_out * c.W + c.gx];
        c.dimg[p] = g;
    }
}

// SIDE: the launch also carries the jobs that prepare the rasterizer's backward blend of the same frame (backward_prologue.h) in
// its LAST `side_jobs` workgroups -- tile ordering on one CU per XCD, the fused optimizer's bookkeeping, clearing the gradient
// accumulator with a few dozen workgroups that stride over it.  (Round 2 put ~1 500 short zeroing workgroups FIRST: they took every
// resident slot -- 200 VGPRs: four workgroups per CU -- and the strips, whose latency chains are the launch, started when those
// retired: 22.3 us in the step against 18.2 us for the loss backward alone; now 20.7.  Letting every strip wave clear a share with a
// dozen stores of its own instead was slower still, 23.3 us: the stores sit in front of the strip's first loads.)  They have nothing to do with the loss; they ride here because this launch sits between the two blends of a training
// step and leaves most of the machine's issue slots and all of its HBM bandwidth unused, while a launch of their own costs 12 us.
// grid: 1-D = per channel plane ceil(strips / WPB) strip workgroups (+ 1 for the deferred loss value), then the side jobs
template <bool SIDE>
__global__ __launch_bounds__(64 * WPB) void k_l1_ssim_backward(int H, int W, int strips_x, int strips_y, const float* __restrict__ img,
                                                                const float* __restrict__ gt, float w_l1, float w_ssim,
                                                                const float* __restrict__ upstream, const float* __restrict__ upstream_ssim,
                                                                const float* __restrict__ gate,
                                                                const float* __restrict__ dm_dmu1, const float* __restrict__ dm_dexx,
                                                                const float* __restrict__ dm_dexy, float* __restrict__ dimg,
                                                                const float* __restrict__ fin_partial, size_t fin_n, float fin_lambda,
                                                                float* __restrict__ fin_loss, float* __restrict__ fin_running,
                                                                unsigned per_plane, unsigned main_wgs, unsigned side_jobs, EgsPrologueArgs side) {
    __shared__ __attribute__((aligned(8))) float lds[WPB][3 * 80];      // per wave: [80] pairs (two maps), then [80] floats (the third)
    if (SIDE) {
        __shared__ EgsOrderLds order_lds;
        if (blockIdx.x >= gridDim.x - side_jobs) { egs_prologue_job<64 * WPB>(side, blockIdx.x - (gridDim.x - side_jobs), side_jobs, order_lds); return; }
    }
    const unsigned lane = threadIdx.x & 63, wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave id, kept scalar
    const unsigned rel = loss_logical_block(blockIdx.x, main_wgs);    // (the strip workgroups come first in the grid, rounded up to a multiple of 8)
    if (rel >= main_wgs) return;
    const unsigned plane_z = rel / per_plane, bx = rel - plane_z * per_plane;
    const int strip = (int)bx * WPB + (int)wv;
    if (fin_partial && bx == per_plane - 1) {                          // deferred loss value: one extra workgroup per channel plane, no strip
        if (plane_z == 0 && wv == 0) wave_finish_loss(fin_n, fin_partial, w_l1, w_ssim, fin_lambda, fin_loss, fin_running, lane);
        return;                                                        // (every resident wave ends with the kernel: a wave with a strip has no slack)
    }
    if (strip >= strips_x * strips_y) return;
    for (int k = lane; k < 3 * 80; k += 64) lds[wv][k] = 0.f;
    __builtin_amdgcn_wave_barrier();
    BwdCtx c;
    c.H = H; c.W = W; c.lane = lane; c.img = img; c.gt = gt; c.m0 = dm_dmu1; c.m1 = dm_dexx; c.m2 = dm_dexy; c.gate = gate; c.dimg = dimg;
    c.rows = lds[wv]; c.plane = (size_t)plane_z * H * W; c.w_l1 = w_l1; c.w_ssim = w_ssim; c.up = upstream[0];
    if (upstream_ssim) { c.w_l1 *= c.up; c.w_ssim *= upstream_ssim[0]; c.up = 1.f; }      // the two terms with an upstream scalar each (egs_l1_ssim_pair_backward)
    const int sx = strip % strips_x, sy = strip / strips_x;
    c.gx = sx * SW - HALO + (int)lane;
    c.col_ok = c.gx >= 0 && c.gx < W;
    c.col_out = c.col_ok && lane >= HALO && lane < HALO + SW;
    c.y_first = sy * SR; c.y_end = min(c.y_first + SR, H);
    Window<3> w;
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int k = 0; k < 11; k++) w.v[m][k] = 0.f;
    // per trip: the 3 maps of 11 input rows and (x, y) of the 11 rows that leave (5 above), prefetched one trip ahead
    struct Rows { float a[11], b[11], d[11], x[11], y[11]; };
    auto load_rows = [&](int y0, Rows& r) {
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const int yi = y0 + k, yo = yi - HALO;
            const bool ok = c.col_ok && yi >= 0 && yi < H && yi < c.y_end + HALO;
            const size_t p = c.plane + (size_t)(ok ? yi : 0) * W + (ok ? c.gx : 0);
            r.a[k] = ok ? dm_dmu1[p] : 0.f; r.b[k] = ok ? dm_dexx[p] : 0.f; r.d[k] = ok ? dm_dexy[p] : 0.f;
            const bool oo = c.col_out && yo >= c.y_first && yo < c.y_end;
            const size_t q = c.plane + (size_t)(oo ? yo : 0) * W + (oo ? c.gx : 0);
            r.x[k] = oo ? img[q] : 0.f; r.y[k] = oo ? gt[q] : 0.f;
        }
    };
    Rows cur, nxt;
    load_rows(c.y_first - HALO, cur);
    for (int y0 = c.y_first - HALO; y0 < c.y_end + HALO; y0 += 11) {
        load_rows(y0 + 11, nxt);
#define BSTEP(K) bwd_step<K>(c, w, y0 + K, cur.a[K], cur.b[K], cur.d[K], cur.x[K], cur.y[K])
        BSTEP(0); BSTEP(1); BSTEP(2); BSTEP(3); BSTEP(4); BSTEP(5); BSTEP(6); BSTEP(7); BSTEP(8); BSTEP(9); BSTEP(10);
#undef BSTEP
        cur = nxt;
    }
}

// Adds up the per-block partial sums and assembles the scalar loss (one workgroup; deterministic order).
__global__ __launch_bounds__(1024) void k_l1_ssim_finish(size_t nblocks, const float* __restrict__ partial, float w_l1, float w_ssim,
                                                          float lambda, float* __restrict__ loss, float* __restrict__ running_sum) {
    __shared__ float red[2][16];
    float a = 0.f, b = 0.f;
    for (size_t i = threadIdx.x; i < nblocks; i += 1024) { const float2 v = reinterpret_cast<const float2*>(partial)[i]; a += v.x; b += v.y; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ta = 0.f, tb = 0.f;
        for (int k = 0; k < 16; k++) { ta += red[0][k]; tb += red[1][k]; }
        const float v = w_l1 * ta + lambda - w_ssim * tb;                     // (1-l) mean|x-y| + l (1 - mean SSIM)
        loss[0] = v;
        if (running_sum) running_sum[0] += v;
    }
}

// the two means as two values (egs_l1_ssim_pair_forward)
__global__ __launch_bounds__(1024) void k_l1_ssim_finish_pair(size_t nblocks, const float* __restrict__ partial, float inv_n, float* __restrict__ l1_out,
                                                               float* __restrict__ ssim_out) {
    __shared__ float red[2][16];
    float a = 0.f, b = 0.f;
    for (size_t i = threadIdx.x; i < nblocks; i += 1024) { const float2 v = reinterpret_cast<const float2*>(partial)[i]; a += v.x; b += v.y; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ta = 0.f, tb = 0.f;
        for (int k = 0; k < 16; k++) { ta += red[0][k]; tb += red[1][k]; }
        l1_out[0] = inv_n * ta; ssim_out[0] = inv_n * tb;
    }
}

}  // namespace

int egs_launch_l1_ssim_backward_w(int channels, int height, int width, const float* img, const float* gt, float w_l1_n, float w_ssim_n, float lambda_dssim,
                                const float* upstream_grad, const float* upstream_ssim, const float* gate, const float* dm_dmu1, const float* dm_dexx,
                                const float* dm_dexy, float* dL_dimg, const float* deferred_partial_sums, float* deferred_loss,
                                float* loss_running_sum, const EgsPrologueArgs* side, hipStream_t stream);
int egs_launch_l1_ssim_forward(int channels, int height, int width, const float* img, const float* gt, float lambda_dssim,
                               float* partial_sums, float* dm_dmu1, float* dm_dexx, float* dm_dexy, float* loss, float* loss_running_sum,
                               const EgsPrologueArgs* side, hipStream_t stream);
int egs_launch_l1_ssim_backward(int channels, int height, int width, const float* img, const float* gt, float lambda_dssim,
                                const float* upstream_grad, const float* gate, const float* dm_dmu1, const float* dm_dexx,
                                const float* dm_dexy, float* dL_dimg, const float* deferred_partial_sums, float* deferred_loss,
                                float* loss_running_sum, const EgsPrologueArgs* side, hipStream_t stream) {
    return egs_launch_l1_ssim_backward_w(channels, height, width, img, gt, 1.f - lambda_dssim, lambda_dssim, lambda_dssim, upstream_grad, nullptr, gate, dm_dmu1,
                                     dm_dexx, dm_dexy, dL_dimg, deferred_partial_sums, deferred_loss, loss_running_sum, side, stream);
}
// w_l1_n, w_ssim_n: the weights of mean|x - y| and of (1 - mean SSIM) BEFORE the division by the element count
int egs_launch_l1_ssim_backward_w(int channels, int height, int width, const float* img, const float* gt, float w_l1_n, float w_ssim_n, float lambda_dssim,
                                const float* upstream_grad, const float* upstream_ssim, const float* gate, const float* dm_dmu1, const float* dm_dexx,
                                const float* dm_dexy, float* dL_dimg, const float* deferred_partial_sums, float* deferred_loss,
                                float* loss_running_sum, const EgsPrologueArgs* side, hipStream_t stream) {
    if (channels <= 0 || height <= 0 || width <= 0 || !img || !gt || !upstream_grad || !dm_dmu1 || !dm_dexx || !dm_dexy || !dL_dimg)
        return EGS_ERR_ARG;
    const float n = (float)channels * (float)height * (float)width;
    const int strips_x = (width + SW - 1) / SW, strips_y = (height + SR - 1) / SR;
    const unsigned per_plane = (unsigned)((strips_x * strips_y + WPB - 1) / WPB + (deferred_partial_sums ? 1 : 0));
    // zeroing workgroups: what is left of the 1024 resident slots (4 per CU) next to the strips, the ordering jobs and the tick; 32 at least
    const unsigned main_wgs = per_plane * (unsigned)channels;
    const unsigned main_pad = ((main_wgs + 7u) / 8u) * 8u;            // loss_logical_block: eight equal runs (the grid of strip workgroups)
    // (counted against the PADDED grid: one workgroup over the 1024 slots waits ~15 us for a strip to end and then clears its share --
    // the launch took 24.4 instead of 17.9 us, tools/loss_side_time.py)
    const unsigned spare = main_pad + EGS_XCDS + 1 + 32 <= 1024 ? 1024 - main_pad - EGS_XCDS - 1 : 32;
    const unsigned side_jobs = side ? egs_prologue_jobs(side->n4, side->has_tick, 64 * WPB, spare) : 0u;
    EgsPrologueArgs none = {};
#define LB_ARGS height, width, strips_x, strips_y, img, gt, w_l1_n / n, w_ssim_n / n, upstream_grad, upstream_ssim, gate, dm_dmu1, dm_dexx, \
                dm_dexy, dL_dimg, deferred_partial_sums, (size_t)strips_x * strips_y * channels, lambda_dssim, deferred_loss,            \
                deferred_partial_sums ? loss_running_sum : nullptr, per_plane, main_wgs, side_jobs
    if (side) hipLaunchKernelGGL(k_l1_ssim_backward<true>, dim3(side_jobs + main_pad), dim3(64 * WPB), 0, stream, LB_ARGS, *side);
    else hipLaunchKernelGGL(k_l1_ssim_backward<false>, dim3(main_pad), dim3(64 * WPB), 0, stream, LB_ARGS, none);
#undef LB_ARGS
    return (int)hipGetLastError();
}

#ifdef EGS_LOSS_TIMING
extern "C" int egs_debug_loss_stamps(unsigned long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(egs_loss_stamps), sizeof(unsigned long long) * 4 * 64); }
#endif

extern "C" {

size_t egs_l1_ssim_partial_count(int channels, int height, int width) {
    return (size_t)channels * ((height + SR - 1) / SR) * ((width + SW - 1) / SW) * 2;
}

int egs_l1_ssim_forward(int channels, int height, int width, const float* img, const float* gt, float lambda_dssim,
                        float* partial_sums, float* dm_dmu1, float* dm_dexx, float* dm_dexy, float* loss, float* loss_running_sum,
                        void* stream) {
    return egs_launch_l1_ssim_forward(channels, height, width, img, gt, lambda_dssim, partial_sums, dm_dmu1, dm_dexx, dm_dexy, loss, loss_running_sum,
                                      nullptr, (hipStream_t)stream);
}

}  // extern "C"

int egs_launch_l1_ssim_forward(int channels, int height, int width, const float* img, const float* gt, float lambda_dssim,
                               float* partial_sums, float* dm_dmu1, float* dm_dexx, float* dm_dexy, float* loss, float* loss_running_sum,
                               const EgsPrologueArgs* side, hipStream_t stream) {
    if (channels <= 0 || height <= 0 || width <= 0 || !img || !gt || !partial_sums || !dm_dmu1 || !dm_dexx || !dm_dexy)
        return EGS_ERR_ARG;
    const int strips_x = (width + SW - 1) / SW, strips_y = (height + SR - 1) / SR;
    const unsigned per_plane = (unsigned)((strips_x * strips_y + WPB - 1) / WPB), main_wgs = per_plane * (unsigned)channels;
    const unsigned main_pad = ((main_wgs + 7u) / 8u) * 8u;
    EgsPrologueArgs none = {};
    if (side) {
        // 124 VGPRs: eight 2-wave workgroups per CU, 2 048 resident slots; the zeroing workgroups take what the strips leave, 32 at least
        const unsigned spare = main_pad + EGS_XCDS + 1 + 32 <= 2048 ? 2048 - main_pad - EGS_XCDS - 1 : 32;
        const unsigned side_jobs = egs_prologue_jobs(side->n4, side->has_tick, 64 * WPB, spare);
        hipLaunchKernelGGL(k_l1_ssim_forward<true>, dim3(main_pad + side_jobs), dim3(64 * WPB), 0, stream, height, width, strips_x, strips_y, img, gt,
                           partial_sums, dm_dmu1, dm_dexx, dm_dexy, per_plane, main_wgs, side_jobs, *side);
    } else
    hipLaunchKernelGGL(k_l1_ssim_forward<false>, dim3(main_pad), dim3(64 * WPB), 0, stream, height, width, strips_x, strips_y, img, gt,
                       partial_sums, dm_dmu1, dm_dexx, dm_dexy, per_plane, main_wgs, 0u, none);
    const float n = (float)channels * (float)height * (float)width;
    if (loss)                                       // loss == NULL: the value is assembled by egs_l1_ssim_backward (deferred)
        hipLaunchKernelGGL(k_l1_ssim_finish, dim3(1), dim3(1024), 0, stream, (size_t)strips_x * strips_y * channels, partial_sums,
                           (1.f - lambda_dssim) / n, lambda_dssim / n, lambda_dssim, loss, loss_running_sum);
    return (int)hipGetLastError();
}

extern "C" {

int egs_l1_ssim_pair_forward(int channels, int height, int width, const float* img, const float* gt, float* partial_sums, float* dm_dmu1,
                             float* dm_dexx, float* dm_dexy, float* l1_out, float* ssim_out, void* stream) {
    if (!l1_out || !ssim_out) return EGS_ERR_ARG;
    const int rc = egs_l1_ssim_forward(channels, height, width, img, gt, 0.f, partial_sums, dm_dmu1, dm_dexx, dm_dexy, nullptr, nullptr, stream);
    if (rc) return rc;
    const int strips_x = (width + SW - 1) / SW, strips_y = (height + SR - 1) / SR;
    hipLaunchKernelGGL(k_l1_ssim_finish_pair, dim3(1), dim3(1024), 0, (hipStream_t)stream, (size_t)strips_x * strips_y * channels, partial_sums,
                       1.f / ((float)channels * (float)height * (float)width), l1_out, ssim_out);
    return (int)hipGetLastError();
}

int egs_l1_ssim_backward(int channels, int height, int width, const float* img, const float* gt, float lambda_dssim,
                         const float* upstream_grad, const float* gate, const float* dm_dmu1, const float* dm_dexx,
                         const float* dm_dexy, float* dL_dimg, const float* deferred_partial_sums, float* deferred_loss,
                         float* loss_running_sum, void* stream) {
    return egs_launch_l1_ssim_backward(channels, height, width, img, gt, lambda_dssim, upstream_grad, gate, dm_dmu1, dm_dexx, dm_dexy, dL_dimg,
                                       deferred_partial_sums, deferred_loss, loss_running_sum, nullptr, (hipStream_t)stream);
}

}  // extern "C"
