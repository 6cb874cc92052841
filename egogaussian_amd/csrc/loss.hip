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
#define WPB 2                 // waves per workgroup (independent)

namespace {

// gaussian(11, 1.5) normalised, as float32 (utils/loss_utils.py:66-68)
#define KW0 1.028380124e-03f
#define KW1 7.598758209e-03f
#define KW2 3.600077331e-02f
#define KW3 1.093606874e-01f
#define KW4 2.130055279e-01f
#define KW5 2.660117149e-01f
__device__ __forceinline__ constexpr float kwin(int k) {
    return k == 0 || k == 10 ? KW0 : k == 1 || k == 9 ? KW1 : k == 2 || k == 8 ? KW2 : k == 3 || k == 7 ? KW3 : k == 4 || k == 6 ? KW4 : KW5;
}

// Horizontal blur of NV values per lane (= per column) through the wave's LDS rows: out[m] = sum_k w[k] v[m][lane - 5 + k].
// All NV rows are written, then all reads are issued together: one LDS round trip per image row, not one per map.
// Lanes 0..4 and 59..63 read the rows' zero padding and return values nobody uses.
template <int NV>
__device__ __forceinline__ void hblur(const float (&v)[NV], float (&out)[NV], float* rows /* [NV][80], lane L at [8 + L] */, unsigned lane) {
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
__device__ __forceinline__ void vblur(const Window<NV>& w, float (&out)[NV]) {
#pragma unroll
    for (int m = 0; m < NV; m++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) acc = fmaf(kwin(k), w.v[m][(NEWEST + 1 + k) % 11], acc);
        out[m] = acc;
    }
}

struct FwdCtx {
    int H, W, gx, y_first, y_end; size_t plane; bool col_ok, col_out; unsigned lane;
    const float* img; const float* gt; float* dm_dmu1; float* dm_dexx; float* dm_dexy; float* rows;   // rows: [4][80] floats
    float l1, sm;
};

// One input row enters (slot NEWEST); if 11 rows are in, the output row 5 above it leaves.
template <int NEWEST>
__device__ __forceinline__ void fwd_step(FwdCtx& c, Window<4>& w, int y_in, float u, float v) {
    // SSIM uses the two variances only as their sum, so E[x^2] and E[y^2] are blurred together: four maps, not five
    w.v[0][NEWEST] = u; w.v[1][NEWEST] = v; w.v[2][NEWEST] = fmaf(u, u, v * v); w.v[3][NEWEST] = u * v;
    const int y_out = y_in - HALO;
    if (y_out < c.y_first || y_out >= c.y_end) return;                  // wave-uniform
    float vb[4];
    vblur<4, NEWEST>(w, vb);
    float hbv[4];
    hblur<4>(vb, hbv, c.rows, c.lane);
    const float mu1 = hbv[0], mu2 = hbv[1], exx_eyy = hbv[2], exy = hbv[3];
    if (c.col_out) {
        const float C1 = 0.0001f, C2 = 0.0009f;
        const float s12 = exy - mu1 * mu2;
        const float A = 2.f * mu1 * mu2 + C1, B = 2.f * s12 + C2, D = mu1 * mu1 + mu2 * mu2 + C1, E = (exx_eyy - (D - C1)) + C2;
        const float invDE = 1.f / (D * E);
        const float sm = A * B * invDE;
        // partial derivatives of the map holding the other windowed moments fixed
        const size_t p = c.plane + (size_t)y_out * c.W + c.gx;
        c.dm_dmu1[p] = (2.f * mu2 * (B - A)) * invDE - sm * (2.f * mu1 * (E - D)) * invDE;
        c.dm_dexx[p] = -sm / E;
        c.dm_dexy[p] = 2.f * A * invDE;
        constexpr int CENTRE = (NEWEST + 11 - HALO) % 11;               // the row that is leaving sits 5 slots behind the newest
        c.l1 += fabsf(w.v[0][CENTRE] - w.v[1][CENTRE]);
        c.sm += sm;
    }
}

// grid: (ceil(strips_x * strips_y / WPB), 1, C); a wave = one strip
__global__ __launch_bounds__(64 * WPB) void k_l1_ssim_forward(int H, int W, int strips_x, int strips_y, const float* __restrict__ img,
                                                               const float* __restrict__ gt, float* __restrict__ partial,
                                                               float* __restrict__ dm_dmu1, float* __restrict__ dm_dexx,
                                                               float* __restrict__ dm_dexy) {
    __shared__ float lds[WPB][4 * 80];
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int strip = blockIdx.x * WPB + (int)wv;
    if (strip >= strips_x * strips_y) return;
    for (int k = lane; k < 4 * 80; k += 64) lds[wv][k] = 0.f;           // the padding words stay zero
    __builtin_amdgcn_wave_barrier();
    FwdCtx c;
    c.H = H; c.W = W; c.lane = lane; c.img = img; c.gt = gt; c.dm_dmu1 = dm_dmu1; c.dm_dexx = dm_dexx; c.dm_dexy = dm_dexy;
    c.rows = lds[wv]; c.plane = (size_t)blockIdx.z * H * W; c.l1 = 0.f; c.sm = 0.f;
    const int sx = strip % strips_x, sy = strip / strips_x;
    c.gx = sx * SW - HALO + (int)lane;
    c.col_ok = c.gx >= 0 && c.gx < W;
    c.col_out = c.col_ok && lane >= HALO && lane < HALO + SW;
    c.y_first = sy * SR; c.y_end = min(c.y_first + SR, H);
    Window<4> w;
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int k = 0; k < 11; k++) w.v[m][k] = 0.f;
    // input rows y_first - 5 .. y_end + 4, eleven per trip so that every window slot index is a compile-time constant;
    // the next trip's 22 loads are in flight while this trip computes (a row-by-row load would expose a full memory
    // latency per row: 28 rows x ~1 us)
    auto load_rows = [&](int y0, float (&u)[11], float (&v)[11]) {
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const int y = y0 + k;
            const bool ok = c.col_ok && y >= 0 && y < H && y < c.y_end + HALO;
            const size_t p = c.plane + (size_t)(ok ? y : 0) * W + (ok ? c.gx : 0);
            u[k] = ok ? img[p] : 0.f; v[k] = ok ? gt[p] : 0.f;
        }
    };
    float cu[11], cv[11], nu[11], nv[11];
    load_rows(c.y_first - HALO, cu, cv);
    for (int y0 = c.y_first - HALO; y0 < c.y_end + HALO; y0 += 11) {
        load_rows(y0 + 11, nu, nv);
        fwd_step<0>(c, w, y0, cu[0], cv[0]);         fwd_step<1>(c, w, y0 + 1, cu[1], cv[1]); fwd_step<2>(c, w, y0 + 2, cu[2], cv[2]);
        fwd_step<3>(c, w, y0 + 3, cu[3], cv[3]);     fwd_step<4>(c, w, y0 + 4, cu[4], cv[4]); fwd_step<5>(c, w, y0 + 5, cu[5], cv[5]);
        fwd_step<6>(c, w, y0 + 6, cu[6], cv[6]);     fwd_step<7>(c, w, y0 + 7, cu[7], cv[7]); fwd_step<8>(c, w, y0 + 8, cu[8], cv[8]);
        fwd_step<9>(c, w, y0 + 9, cu[9], cv[9]);     fwd_step<10>(c, w, y0 + 10, cu[10], cv[10]);
#pragma unroll
        for (int k = 0; k < 11; k++) { cu[k] = nu[k]; cv[k] = nv[k]; }
    }
    float l1 = c.l1, sm = c.sm;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { l1 += __shfl_xor(l1, d, 64); sm += __shfl_xor(sm, d, 64); }
    if (lane == 0) {
        const size_t b = (size_t)blockIdx.z * strips_x * strips_y + strip;
        partial[2 * b] = l1; partial[2 * b + 1] = sm;
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
    vblur<3, NEWEST>(w, vb);
    float hbv[3];
    hblur<3>(vb, hbv, c.rows, c.lane);
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

// The scalar loss from the per-strip partial sums, by ONE wave (fixed order, so the value is deterministic): used by the backward kernel when the caller deferred
// the loss value to it (egs_l1_ssim_forward with loss == NULL) -- a training step replayed from a graph reads the value only after
// the backward anyway, and every launch it does not make is ~4.5 us of GPU time.
__device__ __forceinline__ void wave_finish_loss(size_t nblocks, const float* __restrict__ partial, float w_l1, float w_ssim, float lambda,
                                                 float* __restrict__ loss, float* __restrict__ running_sum, unsigned lane) {
    float a = 0.f, b = 0.f;
    for (size_t i0 = 0; i0 < nblocks; i0 += 64 * 8) {                  // eight loads in flight per lane
        float2 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { const size_t i = i0 + (size_t)k * 64 + lane; v[k] = i < nblocks ? reinterpret_cast<const float2*>(partial)[i] : make_float2(0.f, 0.f); }
#pragma unroll
        for (int k = 0; k < 8; k++) { a += v[k].x; b += v[k].y; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    if (lane == 0) {
        const float v = w_l1 * a + lambda - w_ssim * b;
        if (loss) loss[0] = v;
        if (running_sum) running_sum[0] += v;
    }
}

// SIDE: the launch also carries the jobs that prepare the rasterizer's backward blend of the same frame (backward_prologue.h) in
// its first `side_jobs` workgroups -- tile ordering on one CU per XCD, the fused optimizer's bookkeeping, clearing the gradient
// accumulator.  They have nothing to do with the loss; they ride here because this launch sits between the two blends of a training
// step and leaves most of the machine's issue slots and all of its HBM bandwidth unused, while a launch of their own costs 12 us.
// grid: 1-D = side jobs, then per channel plane ceil(strips / WPB) strip workgroups (+ 1 for the deferred loss value)
template <bool SIDE>
__global__ __launch_bounds__(64 * WPB) void k_l1_ssim_backward(int H, int W, int strips_x, int strips_y, const float* __restrict__ img,
                                                                const float* __restrict__ gt, float w_l1, float w_ssim,
                                                                const float* __restrict__ upstream, const float* __restrict__ gate,
                                                                const float* __restrict__ dm_dmu1, const float* __restrict__ dm_dexx,
                                                                const float* __restrict__ dm_dexy, float* __restrict__ dimg,
                                                                const float* __restrict__ fin_partial, size_t fin_n, float fin_lambda,
                                                                float* __restrict__ fin_loss, float* __restrict__ fin_running,
                                                                unsigned per_plane, unsigned side_jobs, EgsPrologueArgs side) {
    __shared__ float lds[WPB][3 * 80];
    if (SIDE) {
        __shared__ EgsOrderLds order_lds;
        if (blockIdx.x < side_jobs) { egs_prologue_job<64 * WPB>(side, blockIdx.x, side_jobs, order_lds); return; }
    }
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned rel = blockIdx.x - (SIDE ? side_jobs : 0u);
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

}  // namespace

int egs_launch_l1_ssim_backward(int channels, int height, int width, const float* img, const float* gt, float lambda_dssim,
                                const float* upstream_grad, const float* gate, const float* dm_dmu1, const float* dm_dexx,
                                const float* dm_dexy, float* dL_dimg, const float* deferred_partial_sums, float* deferred_loss,
                                float* loss_running_sum, const EgsPrologueArgs* side, hipStream_t stream) {
    if (channels <= 0 || height <= 0 || width <= 0 || !img || !gt || !upstream_grad || !dm_dmu1 || !dm_dexx || !dm_dexy || !dL_dimg)
        return EGS_ERR_ARG;
    const float n = (float)channels * (float)height * (float)width;
    const int strips_x = (width + SW - 1) / SW, strips_y = (height + SR - 1) / SR;
    const unsigned per_plane = (unsigned)((strips_x * strips_y + WPB - 1) / WPB + (deferred_partial_sums ? 1 : 0));
    const unsigned side_jobs = side ? egs_prologue_jobs(side->n4, side->has_tick, 64 * WPB) : 0u;
    EgsPrologueArgs none = {};
#define LB_ARGS height, width, strips_x, strips_y, img, gt, (1.f - lambda_dssim) / n, lambda_dssim / n, upstream_grad, gate, dm_dmu1, dm_dexx, \
                dm_dexy, dL_dimg, deferred_partial_sums, (size_t)strips_x * strips_y * channels, lambda_dssim, deferred_loss,            \
                deferred_partial_sums ? loss_running_sum : nullptr, per_plane, side_jobs
    if (side) hipLaunchKernelGGL(k_l1_ssim_backward<true>, dim3(side_jobs + per_plane * (unsigned)channels), dim3(64 * WPB), 0, stream, LB_ARGS, *side);
    else hipLaunchKernelGGL(k_l1_ssim_backward<false>, dim3(per_plane * (unsigned)channels), dim3(64 * WPB), 0, stream, LB_ARGS, none);
#undef LB_ARGS
    return (int)hipGetLastError();
}

extern "C" {

size_t egs_l1_ssim_partial_count(int channels, int height, int width) {
    return (size_t)channels * ((height + SR - 1) / SR) * ((width + SW - 1) / SW) * 2;
}

int egs_l1_ssim_forward(int channels, int height, int width, const float* img, const float* gt, float lambda_dssim,
                        float* partial_sums, float* dm_dmu1, float* dm_dexx, float* dm_dexy, float* loss, float* loss_running_sum,
                        void* stream) {
    if (channels <= 0 || height <= 0 || width <= 0 || !img || !gt || !partial_sums || !dm_dmu1 || !dm_dexx || !dm_dexy)
        return EGS_ERR_ARG;
    const int strips_x = (width + SW - 1) / SW, strips_y = (height + SR - 1) / SR;
    dim3 grid((strips_x * strips_y + WPB - 1) / WPB, 1, channels);
    hipLaunchKernelGGL(k_l1_ssim_forward, grid, dim3(64 * WPB), 0, (hipStream_t)stream, height, width, strips_x, strips_y, img, gt,
                       partial_sums, dm_dmu1, dm_dexx, dm_dexy);
    const float n = (float)channels * (float)height * (float)width;
    if (loss)                                       // loss == NULL: the value is assembled by egs_l1_ssim_backward (deferred)
        hipLaunchKernelGGL(k_l1_ssim_finish, dim3(1), dim3(1024), 0, (hipStream_t)stream, (size_t)strips_x * strips_y * channels, partial_sums,
                           (1.f - lambda_dssim) / n, lambda_dssim / n, lambda_dssim, loss, loss_running_sum);
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
