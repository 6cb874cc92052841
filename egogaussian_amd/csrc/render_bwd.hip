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
#include "backward_prologue.h"
#include "blend_instrument.h"      // measurement / ablation hooks: all empty in the product build
#include "loss_window.h"
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

__global__ __launch_bounds__(1024) void k_backward_prologue(EgsPrologueArgs a) {
    __shared__ EgsOrderLds L;
    egs_prologue_job<1024>(a, blockIdx.x, gridDim.x, L);
}

// 8 waves per SIMD (64 VGPRs, one spilled): every wave of a 960x540 frame is resident from the start (-2% vs 70 VGPRs / 7 waves)
// MODE 2: upstream gradients on the depth and / or alpha outputs exist; MODE 1: colour only (the training loss: three FMAs and a
// multiply less per (wave, splat) visit of a kernel that is 86 % VALU-busy).
// MODE 0 (ABI 4, egs_backward grad_mask == EGS_GRAD_COLORS): only dL/dcolors_precomp is wanted -- the reference's label call
// (/root/reference/gaussian_renderer/render_helper.py:38-54 detaches every geometric input).  dL/dcolour_c = sum over pixels of
// w dL/dC_c with w = alpha T: no dL/dalpha recurrence, no background term, no moments -- three sums per (wave, splat) instead of ten.
// LG (loss gradient inside the blend): dL/dC of the tile's 256 pixels is not loaded but computed here from what the image loss's FORWARD
// left -- its three derivative maps, the image and the ground truth -- with k_l1_ssim_backward's arithmetic, operation for operation
// (loss.hip bwd_step / vblur_s / hblur_s: vertical 11-tap blur of the maps first, then the horizontal one, zero padding outside the image),
// so that the training step needs no loss-backward launch at all.  Threads 0..233 = (channel, map, window column): each loads its column of
// the tile's 26-row window and leaves 16 vertically blurred values in LDS ([channel][map][16][27] floats, 15.2 KiB, in the space the loop
// below uses for the staged records and the reduction); after ONE barrier every lane blurs horizontally at its own pixel.
#ifdef EGS_LG_CHECK
__device__ unsigned egs_lg_mismatch;
#endif
struct EgsLossGrad { const float* img; const float* gt; const float* m0; const float* m1; const float* m2; const float* gate;
                     const float* up; const float* up_ssim; float w_l1, w_ssim;
                     const float* fin_partial; size_t fin_n; float fin_lambda; float* fin_loss; float* fin_running; };
#define LG_COLS 27
template <int MODE, bool LG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_render_backward(
    int W, int H, int gx, int n_tiles, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ rec, const float* __restrict__ bg, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
    const float* __restrict__ dL_dalpha, const uint32_t* __restrict__ tile_order, float* __restrict__ grad_acc,
    const uint32_t* __restrict__ quad_visits, const uint32_t hot_base /* first float of the hot replica lines inside grad_acc */,
    const uint32_t hot_slots /* lines per replica = ceil(P / 256) * EGS_HOT_PER_BLOCK */, const EgsLossGrad lg) {
    constexpr bool HAS_DA = MODE == 2;
    constexpr int NV = MODE == 0 ? 3 : 10;                            // sums per (wave, splat)
    __shared__ float4 smem[4 * 64 * EGS_SPLAT_REC_F4 + 4 * 5 * 16];      // the staged records of four waves, then their reduction slices
    float4 (*lds)[64 * EGS_SPLAT_REC_F4] = reinterpret_cast<float4 (*)[64 * EGS_SPLAT_REC_F4]>(smem);
    float (*red)[5 * 64] = reinterpret_cast<float (*)[5 * 64]>(smem + 4 * 64 * EGS_SPLAT_REC_F4);
    static_assert(sizeof(smem) >= 9 * 16 * LG_COLS * sizeof(float), "the loss-gradient prologue's blurred maps fit the loop's LDS");
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
    EGS_BWD_MEASURE(const uint64_t t_start = wall_clock64(); uint32_t meas = 0;)
    const int qx0 = (tile % gx) * EGS_TILE + (int)(q & 1) * 8, qy0 = (tile / gx) * EGS_TILE + (int)(q >> 1) * 8;
    const int px = qx0 + (int)(lane & 7), py = qy0 + (int)(lane >> 3);
    const bool inside = px < W && py < H;
    float lg_r = 0.f, lg_g = 0.f, lg_b = 0.f;
    if (LG) {
        // the loss value the forward deferred: one wave of the launch adds up the per-strip partial sums (loss_window.h), as k_l1_ssim_backward did
        if (lg.fin_partial && blockIdx.x == 0 && wv == 0) wave_finish_loss(lg.fin_n, lg.fin_partial, lg.w_l1, lg.w_ssim, lg.fin_lambda, lg.fin_loss, lg.fin_running, lane);
        const int tx0 = (tile % gx) * EGS_TILE, ty0 = (tile / gx) * EGS_TILE;
        float* vb = reinterpret_cast<float*>(smem);                  // [9][16][LG_COLS]
        const size_t HWp = (size_t)H * W;
        if (!(order_word & EGS_ORDER_HAS_PERM)) __syncthreads();      // (nothing else has touched the LDS yet; keeps the barrier count uniform)
        if (threadIdx.x < 234) {
            const unsigned cm = threadIdx.x / 26u, col = threadIdx.x - cm * 26u, ch = cm / 3u, mp_i = cm - ch * 3u;
            const float* __restrict__ mp = (mp_i == 0 ? lg.m0 : mp_i == 1 ? lg.m1 : lg.m2) + ch * HWp;
            const int gxc = tx0 - 5 + (int)col;
            const bool colok = gxc >= 0 && gxc < W;
            float wv_[26];
#pragma unroll
            for (int r = 0; r < 26; r++) {
                const int gy_ = ty0 - 5 + r;
                const bool ok = colok && gy_ >= 0 && gy_ < H;
                wv_[r] = ok ? mp[(size_t)gy_ * W + gxc] : 0.f;
            }
            float* dst = vb + (size_t)cm * 16 * LG_COLS + col;
#pragma unroll
            for (int o = 0; o < 16; o++) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 11; k++) acc = fmaf(kwin(k), wv_[o + k], acc);
                dst[o * LG_COLS] = acc;
            }
        }
        __syncthreads();
        // horizontal blur at the lane's own pixel, then the gradient of 0.8 L1 + 0.2 (1 - SSIM) (bwd_step of loss.hip, operation for operation)
        if (inside) {
            const int lx = px - tx0, ly = py - ty0;
            const float up0 = lg.up[0];
            float w_l1 = lg.w_l1, w_ssim = lg.w_ssim, up = up0;
            if (lg.up_ssim) { w_l1 *= up0; w_ssim *= lg.up_ssim[0]; up = 1.f; }
            const float gate = lg.gate ? lg.gate[(size_t)py * W + px] : 1.f;
            float gch[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                float bl[3];
#pragma unroll
                for (int m = 0; m < 3; m++) {
                    const float* r = vb + (size_t)(ch * 3 + m) * 16 * LG_COLS + ly * LG_COLS + lx;      // r[k] = column lx - 5 + k of the window
                    float acc = kwin(0) * (r[0] + r[10]);
                    acc = fmaf(kwin(1), r[1] + r[9], acc); acc = fmaf(kwin(2), r[2] + r[8], acc);
                    acc = fmaf(kwin(3), r[3] + r[7], acc); acc = fmaf(kwin(4), r[4] + r[6], acc);
                    bl[m] = fmaf(kwin(5), r[5], acc);
                }
                const size_t q = ch * HWp + (size_t)py * W + px;
                const float x = lg.img[q], y = lg.gt[q];
                const float diff = x - y;
                const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
                float g = w_l1 * sgn - w_ssim * (bl[0] + 2.f * x * bl[1] + y * bl[2]);
                g *= up;
                if (lg.gate) g *= gate;
                gch[ch] = g;
            }
            lg_r = gch[0]; lg_g = gch[1]; lg_b = gch[2];
        }
        __syncthreads();                                              // the loop below reuses this LDS for its staged records
    }
    if (qx0 >= W || qy0 >= H) return;

    const float pxf = (float)px, pyf = (float)py;
    const uint32_t qx1 = (uint32_t)min(qx0 + 7, W - 1), qy1 = (uint32_t)min(qy0 + 7, H - 1);

    const uint2 range = ranges[tile];
    const uint32_t* list = point_list + range.x;

    float T_final = 0.f, g_r = 0.f, g_g = 0.f, g_b = 0.f, g_d = 0.f, g_a = 0.f;
    uint32_t last = 0;
    if (inside) {
        const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
        T_final = final_T[pix]; last = n_contrib[pix];
        if (LG) { g_r = lg_r; g_g = lg_g; g_b = lg_b; }
        else { g_r = dL_dcolor[pix]; g_g = dL_dcolor[HW + pix]; g_b = dL_dcolor[2 * HW + pix]; }
#ifdef EGS_LG_CHECK                     // debug build: the computed gradient against the one k_l1_ssim_backward wrote (bit for bit)
        if (LG && (__float_as_uint(lg_r) != __float_as_uint(dL_dcolor[pix]) || __float_as_uint(lg_g) != __float_as_uint(dL_dcolor[HW + pix]) ||
                   __float_as_uint(lg_b) != __float_as_uint(dL_dcolor[2 * HW + pix]))) atomicAdd(&egs_lg_mismatch, 1u);
#endif
        if (HAS_DA && dL_ddepth) g_d = dL_ddepth[pix];
        if (HAS_DA && dL_dalpha) g_a = dL_dalpha[pix];
    }
    const float bg_term = MODE == 0 ? 0.f : -T_final * (bg[0] * g_r + bg[1] * g_g + bg[2] * g_b);
    uint32_t wmax = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d, 64));
    if (wmax == 0) return;

    // stage 2/3 of the reduction: lane 4v+g (v < 10) sums partials [8g, 8g+8) of value v; lane 4v publishes it.
    // After the permlane32 fold, value v lives in register row v/2, lanes (v%2)*32 .. +31.
    const unsigned rv = lane >> 2, rg = lane & 3;
    const float4* red_src = reinterpret_cast<const float4*>(myred + (rv < (unsigned)NV ? (rv >> 1) * 64 + (rv & 1) * 32 + rg * 8 : 0));
    const int slot = (rg == 0 && rv < (unsigned)NV) ? (int)rv + (MODE == 0 ? 6 : 0) : -1;      // (MODE 0: the three colour slots of the line)

    // S = the blended colour-gradient term of everything BEHIND the splat being processed (U of the header), kept "ready for the
    // next contributor": after a splat with (a, u) it becomes a u + (1 - a) S -- one state word and one select instead of three
    // (same values and roundings as U_next = fma(a_last, u_last - U, U); the select keeps a skipped splat's colour, NaN included,
    // out of the pixel's chain).
    float T = T_final, S = 0.f;
    EGS_BWD_ABL5(float abl_sink = 0.f;)

    // Longest-remaining-work-first.  A SIMD issues its oldest wave first, and the tiles are dispatched most expensive first, so the
    // light, late waves used to wait for the others and then run down alone -- the last quarter of the launch had fewer than three
    // waves per SIMD (profiles/r2_blend_timelines.md).  The forward counted the splats every quadrant blends; this wave's issue
    // priority (s_setprio, 4 levels) follows the number it still has to replay, so the waves of a SIMD reach the end together.
    // Placement-like: it decides who issues first, never a result.  -DEGS_NO_LRPT builds without it.
    EGS_LRPT(int remaining = (int)quad_visits[tile * 4 + q]; int prio_now = -1;)
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
        // entries whose Gaussian is hot (egs_common.h): their sums go to one of the Gaussian's replica lines instead of its own
        const uint32_t my_code = egs_hot_code(__float_as_uint(c2.z), __float_as_uint(c2.w));
#ifdef EGS_NO_HOT                          // A/B switch: every splat accumulates into its own line
        const uint64_t hot_mask = 0ull; (void)my_code;
#else
        const uint64_t hot_mask = __ballot(have && my_code != 0u);
#endif
        const uint32_t lastb = last > base ? last - base : 0u;       // this pixel uses entries j < lastb of the batch
        while (mask) {
            const int j = 63 - __builtin_clzll(mask);
            mask &= ~(1ull << j);
            EGS_LRPT({
                const int want = remaining >= 64 ? 3 : remaining >= 24 ? 2 : remaining >= 8 ? 1 : 0;       // (thresholds swept at config C)
                if (want != prio_now) {
                    prio_now = want;
                    if (want == 3) __builtin_amdgcn_s_setprio(3); else if (want == 2) __builtin_amdgcn_s_setprio(2); else if (want == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                }
                remaining--;
            })
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
            EGS_BWD_MEASURE(meas++;)
            const float rcp = __builtin_amdgcn_rcpf(1.f - a);
            const float Tn = T * rcp;                                   // transmittance in front of this splat
            const float w = a * Tn;
            if (MODE == 0) {
                T = Tn;
                const float v6 = w * g_r, v7 = w * g_g, v8 = w * g_b;
                (void)S; (void)bg_term;
                myred[0 * 64 + lane] = fold32(v6, v7); myred[1 * 64 + lane] = fold32(v8, 0.f);
            } else {
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

            EGS_BWD_ABL5(abl_sink += ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7)) + (v8 + v9); continue;)
            // 64-lane sums of v0..v9 (see the header): swap-fold, LDS regroup, quad DPP
            myred[0 * 64 + lane] = fold32(v0, v1); myred[1 * 64 + lane] = fold32(v2, v3); myred[2 * 64 + lane] = fold32(v4, v5);
            myred[3 * 64 + lane] = fold32(v6, v7); myred[4 * 64 + lane] = fold32(v8, v9);
            }
            const float4 pa = red_src[0], pb = red_src[1];
            float out = ((pa.x + pa.y) + (pa.z + pa.w)) + ((pb.x + pb.y) + (pb.z + pb.w));
            out = dpp_add<0xB1>(out);       // quad_perm [1,0,3,2]
            out = dpp_add<0x4E>(out);       // quad_perm [2,3,0,1]
            const uint32_t gid = (uint32_t)__builtin_amdgcn_readlane((int)my_id, j);
            EGS_BWD_ABL7(my[j * 3 + 2])
            uint32_t line = gid * (uint32_t)EGS_GRAD_STRIDE;            // first float of the accumulator line (48 P < 2^32 bytes: P < 89 M)
            if ((hot_mask >> j) & 1ull) {                               // (wave-uniform, rare)
                const uint32_t code = (uint32_t)__builtin_amdgcn_readlane((int)my_code, j);
                // replica = the XCD this workgroup runs on (b % 8); the replicas of a Gaussian lie hot_slots lines apart (egs_common.h)
                line = hot_base + ((blockIdx.x % EGS_HOT_REPLICAS) * hot_slots + (gid >> 8) * EGS_HOT_PER_BLOCK + code - 1u) * EGS_HOT_LINE;
            }
            if (slot >= 0) EGS_BWD_ACCUM(grad_acc + (line + (uint32_t)slot), out);
        }
        __builtin_amdgcn_wave_barrier();
    }
    EGS_BWD_ABL5(if (abl_sink == 12345.678f) grad_acc[lane] = abl_sink;)
    EGS_BWD_TIMELINE()
}

}  // namespace

#ifdef EGS_LG_CHECK
EgsLossGradHost egs_debug_lossgrad = {};       // (debug build only, tools/dev/lg_check.py: the product library keeps no such state)
#endif
namespace {
__global__ __launch_bounds__(64) void k_loss_finish(size_t n, const float* __restrict__ partial, float w_l1, float w_ssim, float lambda, float* loss, float* running) {
    wave_finish_loss(n, partial, w_l1, w_ssim, lambda, loss, running, threadIdx.x);
}
}  // namespace
hipError_t egs_launch_loss_finish(const EgsLossGradHost& lg, int W, int H, hipStream_t s) {
    if (!lg.fin_partial) return hipSuccess;
    const float n = (float)W * (float)H * 3.f;
    hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(64), 0, s, lg.fin_n, lg.fin_partial, lg.w_l1_n / n, lg.w_ssim_n / n, lg.fin_lambda, lg.fin_loss, lg.fin_running);
    return hipGetLastError();
}
#ifdef EGS_LG_CHECK
extern "C" unsigned egs_debug_lg_mismatches() { unsigned v = 0; (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(egs_lg_mismatch), sizeof(v)); return v; }
#endif
// What the blend below needs in place: tile order, cleared accumulator, (fused optimizer) bookkeeping -- as a launch of its own
// (egs_l1_ssim_backward_ex can carry the same jobs instead).
hipError_t egs_launch_backward_prologue(int P, int W, int H, EgsImgPtrs im, float* grad_acc, const uint32_t* block_hot, const EgsAdamTick* tick, hipStream_t s) {
    const int n_tiles = ((W + EGS_TILE - 1) / EGS_TILE) * ((H + EGS_TILE - 1) / EGS_TILE);
    if (n_tiles == 0) {
        if (tick) { hipError_t e = egs_launch_adam_tick(*tick, s); if (e != hipSuccess) return e; }
        return egs_launch_zero_f4((float4*)grad_acc, egs_acc_floats((size_t)P) / 4, s);
    }
    EgsPrologueArgs pa = {};
    pa.n_tiles = n_tiles; pa.quad_work = im.quad_work; pa.tile_order = im.tile_order;
    egs_prologue_acc(pa, grad_acc, (size_t)P, block_hot);
    pa.has_tick = tick ? 1 : 0; if (tick) pa.tick = *tick;
    hipLaunchKernelGGL(k_backward_prologue, dim3(egs_prologue_jobs(pa.n4, pa.has_tick, 1024)), dim3(1024), 0, s, pa);
    return hipGetLastError();
}

hipError_t egs_launch_render_backward(int P, int W, int H, const float* bg, EgsGeomPtrs g, const uint32_t* point_list,
                                      EgsImgPtrs im, const float* dL_dcolor, const float* dL_ddepth,
                                      const float* dL_dalpha, float* grad_acc, int colors_only, const EgsLossGradHost* lg, hipStream_t s) {
#ifdef EGS_LG_CHECK
    if (!lg && egs_debug_lossgrad.img) lg = &egs_debug_lossgrad;      // (experiment hook: egs_debug_set_lossgrad)
#endif
    const int gx = (W + EGS_TILE - 1) / EGS_TILE, gy = (H + EGS_TILE - 1) / EGS_TILE;
    const int n_tiles = gx * gy;
    if (n_tiles == 0) return hipSuccess;
    EgsLossGrad lgk = {};
    if (lg) { lgk.img = lg->img; lgk.gt = lg->gt; lgk.m0 = lg->dm_dmu1; lgk.m1 = lg->dm_dexx; lgk.m2 = lg->dm_dexy; lgk.gate = lg->gate;
              lgk.up = lg->upstream; lgk.up_ssim = lg->upstream_ssim; lgk.w_l1 = lg->w_l1_n / ((float)W * (float)H * 3.f); lgk.w_ssim = lg->w_ssim_n / ((float)W * (float)H * 3.f);
              lgk.fin_partial = lg->fin_partial; lgk.fin_n = lg->fin_n; lgk.fin_lambda = lg->fin_lambda; lgk.fin_loss = lg->fin_loss; lgk.fin_running = lg->fin_running; }
#define EGS_BWD_LAUNCH(MODE, LGF) hipLaunchKernelGGL((k_render_backward<MODE, LGF>), dim3(egs_blocks_for_tiles(n_tiles)), dim3(256), 0, s, W, H, gx, n_tiles, \
                           im.ranges, point_list, g.rec, bg, im.final_T, im.n_contrib, dL_dcolor, dL_ddepth, dL_dalpha, \
                           im.tile_order, grad_acc, im.quad_pairs + (size_t)4 * n_tiles, (uint32_t)((size_t)P * EGS_GRAD_STRIDE), (uint32_t)egs_hot_slots((size_t)P), lgk)
    if (colors_only) EGS_BWD_LAUNCH(0, false);
    else if (dL_ddepth || dL_dalpha) EGS_BWD_LAUNCH(2, false);
    else if (lg) EGS_BWD_LAUNCH(1, true);
    else EGS_BWD_LAUNCH(1, false);
#undef EGS_BWD_LAUNCH
    return hipGetLastError();
}

// MODE 0's second half: dL/dcolors_precomp[i] = the three colour sums of Gaussian i's accumulator line (+ its replica lines if it is hot).
namespace {
__global__ __launch_bounds__(256) void k_colors_from_acc(int P, const float* __restrict__ grad_acc, const float* __restrict__ hot_acc, size_t hot_slots,
                                                         const uint8_t* __restrict__ clamped, const int32_t* __restrict__ radii,
                                                         float* __restrict__ dcolors) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    if (radii[i] <= 0) {                                              // never blended; its `clamped` byte (the hot code) is not even written
        dcolors[3 * (size_t)i] = 0.f; dcolors[3 * (size_t)i + 1] = 0.f; dcolors[3 * (size_t)i + 2] = 0.f;
        return;
    }
    const float4* ga = reinterpret_cast<const float4*>(grad_acc + (size_t)i * EGS_GRAD_STRIDE);
    const float4 a1 = ga[1], a2 = ga[2];
    float r = a1.z, g = a1.w, b = a2.x;
    const uint32_t code = (uint32_t)clamped[i] >> 3;
    if (code) {
        const float* hl = hot_acc + ((size_t)(i >> 8) * EGS_HOT_PER_BLOCK + (code - 1u)) * EGS_HOT_LINE;
        for (unsigned rp = 0; rp < EGS_HOT_REPLICAS; rp++) {
            const float4* h = reinterpret_cast<const float4*>(hl + (size_t)rp * hot_slots * EGS_HOT_LINE);
            const float4 h1 = h[1], h2 = h[2];
            r += h1.z; g += h1.w; b += h2.x;
        }
    }
    dcolors[3 * (size_t)i] = r; dcolors[3 * (size_t)i + 1] = g; dcolors[3 * (size_t)i + 2] = b;
}
}  // namespace
hipError_t egs_launch_colors_from_acc(int P, const float* grad_acc, const uint8_t* clamped, const int32_t* radii, float* dcolors, hipStream_t s) {
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_colors_from_acc, dim3((P + 255) / 256), dim3(256), 0, s, P, grad_acc, grad_acc + (size_t)P * EGS_GRAD_STRIDE,
                       egs_hot_slots((size_t)P), clamped, radii, dcolors);
    return hipGetLastError();
}
