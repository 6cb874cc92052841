// preprocess.hip -- per-Gaussian stages of the rasterizer for gfx950:
//   k_preprocess           3D->2D projection, EWA covariance, conic, radius, tile rect, SH->RGB
//   k_preprocess_backward  conic/mean2D/colour/depth gradients -> mean3D, cov3D, SH, scale, quaternion
//   k_mark_visible         near-plane test
// Replaces the per-Gaussian stages of the upstream op called from
// /root/reference/gaussian_renderer/__init__.py:90-98.
//
// One lane per Gaussian; a wave reads 64 consecutive xyz / cov6 / SH rows, i.e. contiguous 768 B /
// 1.5 KB / 768 B spans, so the array-of-rows inputs coalesce without staging.  The forward result is
// packed into one 48-byte record per Gaussian (egs_common.h) so the blend kernels gather three
// dwordx4 per splat instead of touching five arrays.
//
// This translation unit is compiled with -ffp-contract=off and every expression follows the order of
// oracle/raster_oracle.c, so radii, tile rectangles and depth bits are bit-identical to the oracle.
#include "egs_common.h"
#include <stdlib.h>
#include "bin_walk.h"              // the count pass of the tile bucketing, carried by k_preprocess_count
#include "backward_prologue.h"

namespace {

__device__ __constant__ float kC0 = 0.28209479177387814f;
__device__ __constant__ float kC1 = 0.4886025119029199f;
__device__ __constant__ float kC2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                         -1.0925484305920792f, 0.5462742152960396f };
__device__ __constant__ float kC3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                         0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                         -0.5900435899266435f };

__device__ __forceinline__ void xform43(const float* p, const float* m, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
__device__ __forceinline__ void xform44(const float* p, const float* m, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

__device__ __forceinline__ void quat_to_rot(const float* q, float* R) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

// cov3D = (R S)(R S)^T, six unique entries (00,01,02,11,12,22); quaternion used as given.
// Mrot != NULL: the object rotation of the `fine_all` call shape, L <- M L, with the operation order of cov3d.hip (k_cov3d_forward), so
// that the covariance -- hence radii, rectangles, sort order -- is the one the stand-alone producer would have handed over.
__device__ __forceinline__ void cov3d_from_scale_rot(const float* s, float mod, const float* q, float* c6, const float* Mrot = nullptr) {
    float R[9]; quat_to_rot(q, R);
    float sc[3] = { mod * s[0], mod * s[1], mod * s[2] };
    float L[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) L[3 * i + k] = R[3 * i + k] * sc[k];
    if (Mrot) {
        float L2[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) L2[3 * i + j] = Mrot[3 * i] * L[j] + Mrot[3 * i + 1] * L[3 + j] + Mrot[3 * i + 2] * L[6 + j];
#pragma unroll
        for (int k = 0; k < 9; k++) L[k] = L2[k];
    }
    c6[0] = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
    c6[1] = L[0] * L[3] + L[1] * L[4] + L[2] * L[5];
    c6[2] = L[0] * L[6] + L[1] * L[7] + L[2] * L[8];
    c6[3] = L[3] * L[3] + L[4] * L[4] + L[5] * L[5];
    c6[4] = L[3] * L[6] + L[4] * L[7] + L[5] * L[8];
    c6[5] = L[6] * L[6] + L[7] * L[7] + L[8] * L[8];
}

// Raw-parameter mode (EGS_ACT_*, egs_common.h): the model's activations -- exp on the scales, normalisation of the
// quaternion, sigmoid on the opacity (/root/reference/scene/gaussian_model.py:36-44) -- applied in place of separate
// PyTorch launches.  `qinv` returns 1/|q_raw| for the backward.
__device__ __forceinline__ void activate_scale_rot(int act, float* s, float* q, float& qinv) {
    if (act & EGS_ACT_LOG_SCALES) { s[0] = expf(s[0]); s[1] = expf(s[1]); s[2] = expf(s[2]); }
    qinv = 1.f;
    if (act & EGS_ACT_RAW_QUATS) {
        qinv = 1.f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        q[0] *= qinv; q[1] *= qinv; q[2] *= qinv; q[3] *= qinv;
    }
}

// Everything the forward and backward share: camera-space point, clamped Jacobian rows (J R), Sigma*m.
struct Ewa {
    float t[3], tx, ty, txtz, tytz, limx, limy, fx, fy;
    float m0[3], m1[3], S0[3], S1[3], a, b, c;
};
__device__ __forceinline__ void ewa_project(const float* p, const float* c6, const float* V, int W, int H,
                                            float tanfovx, float tanfovy, Ewa& e) {
    xform43(p, V, e.t);
    e.fx = (float)W / (2.f * tanfovx); e.fy = (float)H / (2.f * tanfovy);
    e.limx = 1.3f * tanfovx; e.limy = 1.3f * tanfovy;
    e.txtz = e.t[0] / e.t[2]; e.tytz = e.t[1] / e.t[2];
    e.tx = fminf(e.limx, fmaxf(-e.limx, e.txtz)) * e.t[2];
    e.ty = fminf(e.limy, fmaxf(-e.limy, e.tytz)) * e.t[2];
    float j00 = e.fx / e.t[2], j02 = -(e.fx * e.tx) / (e.t[2] * e.t[2]);
    float j11 = e.fy / e.t[2], j12 = -(e.fy * e.ty) / (e.t[2] * e.t[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        e.m0[k] = j00 * V[4 * k + 0] + j02 * V[4 * k + 2];
        e.m1[k] = j11 * V[4 * k + 1] + j12 * V[4 * k + 2];
    }
    e.S0[0] = c6[0] * e.m0[0] + c6[1] * e.m0[1] + c6[2] * e.m0[2];
    e.S0[1] = c6[1] * e.m0[0] + c6[3] * e.m0[1] + c6[4] * e.m0[2];
    e.S0[2] = c6[2] * e.m0[0] + c6[4] * e.m0[1] + c6[5] * e.m0[2];
    e.S1[0] = c6[0] * e.m1[0] + c6[1] * e.m1[1] + c6[2] * e.m1[2];
    e.S1[1] = c6[1] * e.m1[0] + c6[3] * e.m1[1] + c6[4] * e.m1[2];
    e.S1[2] = c6[2] * e.m1[0] + c6[4] * e.m1[1] + c6[5] * e.m1[2];
    e.a = e.m0[0] * e.S0[0] + e.m0[1] * e.S0[1] + e.m0[2] * e.S0[2] + 0.3f;
    e.b = e.m0[0] * e.S1[0] + e.m0[1] * e.S1[1] + e.m0[2] * e.S1[2];
    e.c = e.m1[0] * e.S1[0] + e.m1[1] * e.S1[1] + e.m1[2] * e.S1[2] + 0.3f;
}

__device__ __forceinline__ float sh_channel(int deg, const float* sh, int ch, float x, float y, float z) {
#define SH(k) sh[(k) * 3 + ch]
    float v = kC0 * SH(0);
    if (deg > 0) {
        v = v - kC1 * y * SH(1) + kC1 * z * SH(2) - kC1 * x * SH(3);
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            v = v + kC2[0] * xy * SH(4) + kC2[1] * yz * SH(5) + kC2[2] * (2.f * zz - xx - yy) * SH(6) +
                kC2[3] * xz * SH(7) + kC2[4] * (xx - yy) * SH(8);
            if (deg > 2) {
                v = v + kC3[0] * y * (3.f * xx - yy) * SH(9) + kC3[1] * xy * z * SH(10) +
                    kC3[2] * y * (4.f * zz - xx - yy) * SH(11) + kC3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * SH(12) +
                    kC3[4] * x * (4.f * zz - xx - yy) * SH(13) + kC3[5] * z * (xx - yy) * SH(14) +
                    kC3[6] * x * (xx - 3.f * yy) * SH(15);
            }
        }
    }
#undef SH
    return v + 0.5f;
}

struct PreOut { uint2 rect; float4 r0, r1, r2; };
// Returns the number of tiles the Gaussian touches (0 = culled).
__device__ __forceinline__ uint32_t preprocess_one(
    int i, int D, int M, const float* __restrict__ means3D, const float* __restrict__ shs,
    const float* __restrict__ colors, const float* __restrict__ opac, const float* __restrict__ scales, float mod,
    const float* __restrict__ rots, const float* __restrict__ cov3D_in, int act, const float* __restrict__ V,
    const float* __restrict__ PM, const float* __restrict__ campos, int W, int H, float tanfovx, float tanfovy,
    int32_t* __restrict__ radii, float4* __restrict__ rec, uint2* __restrict__ rect_out,
    uint32_t* __restrict__ tiles_touched, uint8_t* __restrict__ clamped_out, uint8_t* __restrict__ visible, const EgsObjRot rot,
    bool& hot, PreOut* out = nullptr /*the rectangle and the record as written (left alone when culled, i.e. on a zero return): k_preprocess_count walks them*/) {
    const int gx = (W + EGS_TILE - 1) / EGS_TILE, gy = (H + EGS_TILE - 1) / EGS_TILE;
    radii[i] = 0; tiles_touched[i] = 0; visible[i] = 0;

    const float p[3] = { means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2] };
    float c6[6];
    if (cov3D_in) {
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = cov3D_in[6 * (size_t)i + k];
    } else {
        float s[3] = { scales[3 * i], scales[3 * i + 1], scales[3 * i + 2] };
        float q[4] = { rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3] };
        float qinv; activate_scale_rot(act, s, q, qinv);
        cov3d_from_scale_rot(s, mod, q, c6, (rot.M && (!rot.sel || rot.sel[i])) ? rot.M : nullptr);
    }
    Ewa e; ewa_project(p, c6, V, W, H, tanfovx, tanfovy, e);
    if (e.t[2] <= 0.2f) return 0u;                                // near-plane cull
    float hom[4]; xform44(p, PM, hom);
    const float pw = 1.f / (hom[3] + 0.0000001f);
    const float ndc_x = hom[0] * pw, ndc_y = hom[1] * pw;

    const float det = e.a * e.c - e.b * e.b;
    if (det == 0.f) return 0u;
    const float det_inv = 1.f / det;
    const float conA = e.c * det_inv, conB = -e.b * det_inv, conC = e.a * det_inv;
    const float mid = 0.5f * (e.a + e.c);
    const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
    const float lam1 = mid + disc, lam2 = mid - disc;
    const int rad = (int)ceilf(3.f * sqrtf(fmaxf(lam1, lam2)));
    const float px = ((ndc_x + 1.f) * (float)W - 1.f) * 0.5f;
    const float py = ((ndc_y + 1.f) * (float)H - 1.f) * 0.5f;
    const int rx0 = min(gx, max(0, (int)((px - (float)rad) / (float)EGS_TILE)));
    const int ry0 = min(gy, max(0, (int)((py - (float)rad) / (float)EGS_TILE)));
    const int rx1 = min(gx, max(0, (int)((px + (float)rad + (float)(EGS_TILE - 1)) / (float)EGS_TILE)));
    const int ry1 = min(gy, max(0, (int)((py + (float)rad + (float)(EGS_TILE - 1)) / (float)EGS_TILE)));
    if ((rx1 - rx0) * (ry1 - ry0) == 0) return 0u;

    float rgb[3]; uint32_t cl = 0;
    if (colors) {
        rgb[0] = colors[3 * i]; rgb[1] = colors[3 * i + 1]; rgb[2] = colors[3 * i + 2];
    } else if (!shs) {
        rgb[0] = rgb[1] = rgb[2] = 0.f;                            // k_sh_forward fills the colour (and the clamp flags) in afterwards
    } else {
        float dir[3] = { p[0] - campos[0], p[1] - campos[1], p[2] - campos[2] };
        const float inv = 1.f / sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
        dir[0] *= inv; dir[1] *= inv; dir[2] *= inv;
        const float* sh = shs + (size_t)i * M * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float v = sh_channel(D, sh, ch, dir[0], dir[1], dir[2]);
            if (v < 0.f) cl |= 1u << ch;
            rgb[ch] = fmaxf(v, 0.f);
        }
    }
    const float o = (act & EGS_ACT_LOGIT_OPACITY) ? 1.f / (1.f + expf(-opac[i])) : opac[i];

    // Conservative pixel bounding box of {alpha >= 1/255}: the ellipse 0.5 d^T Q d <= tau with
    // tau = ln(255 o).  Not part of the published algorithm -- it only lets the blend kernels skip
    // (8x8 pixel block, splat) pairs that provably contribute nothing, so results are unchanged.
    uint32_t bbx = 1u, bby = 1u;                                   // x0 = 1 > x1 = 0: empty
    {
        const float lo = 255.f * o;
        if (!(lo < 0.999f)) {
            const float tau2 = 2.f * (logf(fmaxf(lo, 1.f)) + 0.01f);
            const float detq = conA * conC - conB * conB;
            float ex = 1e9f, ey = 1e9f;
            if (detq > 0.f && conA * conC < 1e4f * detq) {
                ex = fmaxf(sqrtf(tau2 * conC / detq), sqrtf(tau2 * e.a)) * 1.002f + 0.01f;
                ey = fmaxf(sqrtf(tau2 * conA / detq), sqrtf(tau2 * e.c)) * 1.002f + 0.01f;
            }
            const float fx0 = fmaxf(floorf(px - ex), 0.f), fx1 = fminf(ceilf(px + ex), (float)(W - 1));
            const float fy0 = fmaxf(floorf(py - ey), 0.f), fy1 = fminf(ceilf(py + ey), (float)(H - 1));
            if (fx0 <= fx1 && fy0 <= fy1) {
                bbx = (uint32_t)fx0 | ((uint32_t)fx1 << 16);
                bby = (uint32_t)fy0 | ((uint32_t)fy1 << 16);
                // a box over this many tiles: its accumulator line would be hit by every quadrant-wave of all of them (egs_common.h)
                hot = (((uint32_t)fx1 >> 4) - ((uint32_t)fx0 >> 4) + 1u) * (((uint32_t)fy1 >> 4) - ((uint32_t)fy0 >> 4) + 1u) >= EGS_HOT_MIN_TILES;
            } else if (!(ex == ex) || !(ey == ey) || !(px == px) || !(py == py)) {   // NaN: never cull
                bbx = 0u | ((uint32_t)(W - 1) << 16); bby = 0u | ((uint32_t)(H - 1) << 16);
            }
        }
    }

    radii[i] = rad; visible[i] = 1;                                  // radii > 0 as one byte (a torch.bool view for the caller)
    tiles_touched[i] = (uint32_t)((rx1 - rx0) * (ry1 - ry0));
    rect_out[i] = make_uint2((uint32_t)rx0 | ((uint32_t)rx1 << 16), (uint32_t)ry0 | ((uint32_t)ry1 << 16));
    clamped_out[i] = (uint8_t)cl;
    float4* r = rec + (size_t)i * EGS_SPLAT_REC_F4;
    const float4 q0 = make_float4(px, py, (-0.5f * EGS_LOG2E) * conA, -EGS_LOG2E * conB);
    const float4 q1 = make_float4((-0.5f * EGS_LOG2E) * conC, o, rgb[0], rgb[1]);
    const float4 q2 = make_float4(rgb[2], e.t[2], __uint_as_float(bbx), __uint_as_float(bby));
    r[0] = q0; r[1] = q1; r[2] = q2;
    if (out) { out->rect = make_uint2((uint32_t)rx0 | ((uint32_t)rx1 << 16), (uint32_t)ry0 | ((uint32_t)ry1 << 16)); out->r0 = q0; out->r1 = q1; out->r2 = q2; }
    return (uint32_t)((rx1 - rx0) * (ry1 - ry0));
}

// PLACE: the first eight workgroups do not preprocess anything -- each orders one XCD band of the forward blend's tiles by the
// costs the image buffer holds (backward_prologue.h), a job that needs doing before that blend and fits under this launch.
template <bool PLACE>
__global__ __launch_bounds__(256) void k_preprocess(
    int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ shs,
    const float* __restrict__ colors, const float* __restrict__ opac, const float* __restrict__ scales, float mod,
    const float* __restrict__ rots, const float* __restrict__ cov3D_in, int act, const float* __restrict__ V,
    const float* __restrict__ PM, const float* __restrict__ campos, int W, int H, float tanfovx, float tanfovy,
    int32_t* __restrict__ radii, float4* __restrict__ rec, uint2* __restrict__ rect_out,
    uint32_t* __restrict__ tiles_touched, uint8_t* __restrict__ clamped_out, uint8_t* __restrict__ visible,
    uint32_t* __restrict__ block_sums, uint32_t* __restrict__ block_hot, uint32_t* __restrict__ zero_words, size_t zero_n,
    const int32_t* __restrict__ active_count, EgsPrologueArgs place, EgsObjRot rot) {
    __shared__ uint32_t wsum[4], whot[4];
    if (PLACE) {
        __shared__ EgsOrderLds order_lds;
        if (blockIdx.x < EGS_XCDS) { egs_order_band<256>(place, (int)blockIdx.x, order_lds); return; }
    }
    const unsigned bid = blockIdx.x - (PLACE ? EGS_XCDS : 0u), nblk = gridDim.x - (PLACE ? EGS_XCDS : 0u);
    const int i = (int)(bid * blockDim.x + threadIdx.x);
    for (size_t z = (size_t)i; z < zero_n; z += (size_t)nblk * blockDim.x) zero_words[z] = 0u;   // on the side (egs_common.h)
    uint32_t my_tiles = 0;
    // capacity-sized models (include/egs_raster.h): rows at and beyond the device-side live count are culled whatever they hold
    const int live = active_count ? min(P, max(*active_count, 0)) : P;
    if (i >= live && i < P) { radii[i] = 0; tiles_touched[i] = 0; visible[i] = 0; }
    bool hot = false;
    if (i < live) my_tiles = preprocess_one(i, D, M, means3D, shs, colors, opac, scales, mod, rots, cov3D_in, act, V, PM, campos, W, H,
                                         tanfovx, tanfovy, radii, rec, rect_out, tiles_touched, clamped_out, visible, rot, hot);
    hot = hot && my_tiles != 0;
    const uint64_t hot_wave = __ballot(hot);
    // per-block instance count; the host adds the block sums to get R (no contended atomic, deterministic)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) my_tiles += (uint32_t)__shfl_xor((int)my_tiles, d, 64);
    if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = my_tiles; whot[threadIdx.x >> 6] = (uint32_t)__popcll(hot_wave); }
    __syncthreads();
    if (threadIdx.x == 0) { block_sums[bid] = wsum[0] + wsum[1] + wsum[2] + wsum[3]; block_hot[bid] = min(whot[0] + whot[1] + whot[2] + whot[3], EGS_HOT_PER_BLOCK); }
    if (hot) {   // rank among the workgroup's hot Gaussians -> the code that names this one's replica lines (egs_common.h); few per frame
        const unsigned w = threadIdx.x >> 6, lane = threadIdx.x & 63;
        uint32_t rank = (uint32_t)__popcll(hot_wave & (lane ? (~0ull >> (64 - lane)) : 0ull));
        for (unsigned k = 0; k < w; k++) rank += whot[k];
        if (rank < EGS_HOT_PER_BLOCK) {
            float4* r2 = rec + (size_t)i * EGS_SPLAT_REC_F4 + 2;
            uint32_t bbx = __float_as_uint(r2->z), bby = __float_as_uint(r2->w);
            egs_hot_code_set(bbx, bby, rank + 1u);
            r2->z = __uint_as_float(bbx); r2->w = __uint_as_float(bby);
            clamped_out[i] = (uint8_t)((clamped_out[i] & EGS_CLAMP_MASK) | ((rank + 1u) << 3));   // where the backward looks for it
        }
    }
}

// k_preprocess and the COUNT pass of the tile bucketing (binning.hip k_bin_count) in one launch.  The count pass's workgroup is a chain
// of latencies -- launch, set-up loads (~4 us with every workgroup asking at once), walk, flush -- of which the first two only fetch
// what k_preprocess had in registers a launch earlier (profiles/r5_bucketing_experiments.md); here wave w of a 16-wave workgroup
// projects one 64-Gaussian group of the round and parks rectangle, box and conic in the walk's LDS block directly.  The walk, the
// [tile][workgroup] table and the chunk sums are bin_walk.h's, so the lists stay what k_bin_scatter + k_tile_sort made of them before.
// Blocks of 256 Gaussians are dealt to workgroups (bin_walk.h): the per-block instance and hot counts are those of k_preprocess
// (block B = Gaussians 256 B .. 256 B + 255; waves 4 m .. 4 m + 3 of a round hold one block).  Needs gpr % 4 == 0 (egs_can_fuse_count).
// PLACE: the first eight workgroups order the forward blend's tiles instead (as in k_preprocess), in the dynamic LDS block.
struct EgsPreArgs {
    int P, D, M; const float* means3D; const float* shs; const float* colors; const float* opac; const float* scales; float mod;
    const float* rots; const float* cov3D_in; int act; const float* V; const float* PM; const float* campos; int W, H; float tanfovx, tanfovy;
    int32_t* radii; float4* rec; uint2* rect_out; uint32_t* tiles_touched; uint8_t* clamped_out; uint8_t* visible;
    uint32_t* block_sums; uint32_t* block_hot; const int32_t* active_count;
};
struct EgsCountArgs { int gpr, gx, n_tiles; uint32_t nblocks; int cull, use_map; uint32_t* table; uint32_t stride; uint32_t* chunk_sum; };
extern __shared__ __attribute__((aligned(16))) uint32_t pc_dyn_lds[];
// ONE_ROUND: every workgroup's groups fit one round (no loop: what the projection loads is dead before the walk starts -- inside a loop the
// camera matrices and the argument pointers stayed live through it and the kernel needed 105 VGPRs; 64 keep the 16-wave workgroups two per CU).
template <bool PLACE, bool ONE_ROUND>
__global__ __launch_bounds__(EGS_BIN_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_preprocess_count(EgsPreArgs a, EgsCountArgs c, EgsPrologueArgs place, EgsObjRot rot) {
    __shared__ uint32_t wsum[EGS_BIN_WAVES], whot[EGS_BIN_WAVES];
    if (PLACE) {
        if (blockIdx.x < EGS_XCDS) { egs_order_band<EGS_BIN_THREADS>(place, (int)blockIdx.x, *reinterpret_cast<EgsOrderLds*>(pc_dyn_lds)); return; }
    }
    const unsigned bid = bin_logical_block(c.nblocks, PLACE ? EGS_XCDS : 0u);
    if (bid >= c.nblocks) return;
    uint32_t* hist = pc_dyn_lds;
    const BinRound L = bin_round_carve(pc_dyn_lds + ((c.n_tiles + 3) & ~3), c.gpr, c.cull != 0);
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < c.n_tiles; t += EGS_BIN_THREADS) hist[t] = 0;            // (the first round's barrier orders this)
    const int P = a.P;
    const int live = a.active_count ? min(P, max(*a.active_count, 0)) : P;                  // capacity-sized models: rows beyond the live count are culled
    const unsigned groups = ((unsigned)P + 63u) / 64u, per_block = bin_groups_per_block((unsigned)P, c.nblocks);
    unsigned g0 = 0;
    do {
        if (!ONE_ROUND && g0) __syncthreads();                         // the previous round's readers (walk, block sums, hot ranks) are done
        bool hot = false; uint64_t hot_wave = 0ull; int i = -1;
        if ((int)w < c.gpr) {
            const unsigned j = bin_group_of(bid, c.nblocks, g0 + w);
            const bool have = g0 + w < per_block && j < groups;
            i = have ? (int)(j * 64u + lane) : -1;
            uint32_t my_tiles = 0; PreOut o;
            if (i >= live && i < P) { a.radii[i] = 0; a.tiles_touched[i] = 0; a.visible[i] = 0; }
            const float* V = a.V; const float* PM = a.PM; const float* campos = a.campos;
            if (!ONE_ROUND) asm volatile("" : "+s"(V), "+s"(PM), "+s"(campos));     // (re-read every round instead of held in 35 scalar registers through the walk)
            if (i >= 0 && i < live)
                my_tiles = preprocess_one(i, a.D, a.M, a.means3D, a.shs, a.colors, a.opac, a.scales, a.mod, a.rots, a.cov3D_in, a.act, V, PM, campos,
                                          a.W, a.H, a.tanfovx, a.tanfovy, a.radii, a.rec, a.rect_out, a.tiles_touched, a.clamped_out, a.visible, rot, hot, &o);
            hot = hot && my_tiles != 0;
            hot_wave = __ballot(hot);
            if (my_tiles == 0) { o.rect = make_uint2(0u, 0u); o.r0 = o.r1 = o.r2 = make_float4(0.f, 0.f, 0.f, 0.f); }      // (culled: preprocess_one left `o` alone)
            bin_park_group(L, w, lane, i >= 0 && i < P, my_tiles, o.rect, o.r0, o.r1, o.r2, false, c.cull != 0, c.use_map != 0);
            uint32_t sum = my_tiles;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) sum += (uint32_t)__shfl_xor((int)sum, d, 64);
            if (lane == 0) { wsum[w] = sum; whot[w] = (uint32_t)__popcll(hot_wave); }
        }
        __syncthreads();
        if ((int)w < c.gpr && i >= 0) {
            // the 256-Gaussian block of this wave: waves (w & ~3) .. (w & ~3) + 3 of the round (all present: gpr is a multiple of four)
            const unsigned w0 = w & ~3u, blk = (unsigned)i >> 8;
            if ((w & 3u) == 0 && lane == 0) {
                a.block_sums[blk] = wsum[w0] + wsum[w0 + 1] + wsum[w0 + 2] + wsum[w0 + 3];
                a.block_hot[blk] = min(whot[w0] + whot[w0 + 1] + whot[w0 + 2] + whot[w0 + 3], EGS_HOT_PER_BLOCK);
            }
            if (hot) {   // rank among the block's hot Gaussians -> the code that names this one's replica lines (egs_common.h); few per frame
                uint32_t rank = (uint32_t)__popcll(hot_wave & (lane ? (~0ull >> (64 - lane)) : 0ull));
                for (unsigned k = w0; k < w; k++) rank += whot[k];
                if (rank < EGS_HOT_PER_BLOCK) {
                    float4* r2 = a.rec + (size_t)i * EGS_SPLAT_REC_F4 + 2;
                    uint32_t bbx = __float_as_uint(r2->z), bby = __float_as_uint(r2->w);
                    egs_hot_code_set(bbx, bby, rank + 1u);
                    r2->z = __uint_as_float(bbx); r2->w = __uint_as_float(bby);
                    a.clamped_out[i] = (uint8_t)((a.clamped_out[i] & EGS_CLAMP_MASK) | ((rank + 1u) << 3));   // where the backward looks for it
                }
            }
        }
        bin_walk_round(L, bid, c.nblocks, g0, c.gpr, c.gx, false, c.cull != 0, c.use_map != 0, a.W, a.H,
                       [&](uint32_t tile, uint32_t, uint32_t) { atomicAdd(&hist[tile], 1u); });
        g0 += (unsigned)c.gpr;
    } while (!ONE_ROUND && g0 < per_block);
    __syncthreads();
    bin_flush_counts(hist, c.n_tiles, bid, c.stride, c.table, c.chunk_sum, blockIdx.x);
}

// stage offsets (floats) of the leaves a fused optimizer owns: rows of the workgroup's 256 Gaussians, leaf after leaf
#define ST_MEANS 0
#define ST_OPAC 768
#define ST_SCALES 1024
#define ST_ROTS 1792
#define ST_SH 2816
template <bool SINK>
__device__ __forceinline__ void pp_bwd_one(
    const int i, int D, int M, const float* __restrict__ means3D, const float* __restrict__ shs,
    const float* __restrict__ scales, float mod, const float* __restrict__ rots, const float* __restrict__ cov3D_in, int act,
    const float* __restrict__ V, const float* __restrict__ PM, const float* __restrict__ campos, int W, int H,
    float tanfovx, float tanfovy, const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped,
    const float4* __restrict__ rec, const float* __restrict__ grad_acc, const float* __restrict__ hot_acc, const size_t hot_slots, float* __restrict__ dmeans2D, float* __restrict__ dcolors,
    float* __restrict__ dopac, float* __restrict__ dmeans3D, float* __restrict__ dcov3D, float* __restrict__ dsh,
    float* __restrict__ dscales, float* __restrict__ drots,
    float* __restrict__ stat_grad_accum, float* __restrict__ stat_denom, float* __restrict__ stat_max_radii,
    const uint32_t* __restrict__ skip_flag, float* __restrict__ stage, const unsigned fused, const EgsObjRot rot) {
    // SINK: the gradients of the leaves in `fused` (bit = EGS_SINK_*) also go to `stage` (LDS), from where the workgroup applies
    // Adam to its 256 rows; their dX arrays may then be NULL (nothing written)
    const unsigned tid = threadIdx.x;
    const bool vis = radii[i] > 0;
    float acc[EGS_GRAD_STRIDE];
    {
        const float4* ga = reinterpret_cast<const float4*>(grad_acc + (size_t)i * EGS_GRAD_STRIDE);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float4 v = vis ? ga[k] : make_float4(0.f, 0.f, 0.f, 0.f);
            acc[4 * k] = v.x; acc[4 * k + 1] = v.y; acc[4 * k + 2] = v.z; acc[4 * k + 3] = v.w;
        }
    }
    // acc[0..4] are moments of gd = dL/dalpha * G (egs_common.h); with the opacity o and the conic (A, B, C) of this Gaussian
    //   dL/dmean2D = -o (A Sx + B Sy, C Sy + B Sx)   and   dL/dconic = -0.5 o (Sxx, Sxy, Syy)  (xy slot: half the derivative)
    // The published op reports dL/dmean2D in NDC units (x 0.5 W, 0.5 H).
    float gmx = 0.f, gmy = 0.f;
    if (vis) {
        const float4* r = rec + (size_t)i * EGS_SPLAT_REC_F4;
        const float4 r0 = r[0], r1 = r[1];
        const uint32_t code = (uint32_t)clamped[i] >> 3;
        if (code) {   // a hot Gaussian: the blend spread its sums over EGS_HOT_REPLICAS lines behind the regular ones (egs_common.h)
            const float* hl = hot_acc + ((size_t)(i >> 8) * EGS_HOT_PER_BLOCK + (code - 1u)) * EGS_HOT_LINE;
            for (unsigned rp = 0; rp < EGS_HOT_REPLICAS; rp++) {
                const float4* ga = reinterpret_cast<const float4*>(hl + (size_t)rp * hot_slots * EGS_HOT_LINE);
#pragma unroll
                for (int k = 0; k < 3; k++) { const float4 v = ga[k]; acc[4 * k] += v.x; acc[4 * k + 1] += v.y; acc[4 * k + 2] += v.z; acc[4 * k + 3] += v.w; }
            }
        }
        const float cA = r0.z * (-2.f * EGS_LN2), cB = r0.w * (-EGS_LN2), cC = r1.x * (-2.f * EGS_LN2);
        const float o = r1.y;
        gmx = -o * (cA * acc[0] + cB * acc[1]); gmy = -o * (cC * acc[1] + cB * acc[0]);
        acc[2] *= -0.5f * o; acc[3] *= -0.5f * o; acc[4] *= -0.5f * o;
    }
    acc[0] = gmx * (0.5f * (float)W); acc[1] = gmy * (0.5f * (float)H);
    dmeans2D[3 * i] = acc[0]; dmeans2D[3 * i + 1] = acc[1]; dmeans2D[3 * i + 2] = 0.f;
    if (stat_grad_accum && vis && !(skip_flag && *skip_flag)) {       // (skip_flag: this frame overflowed its instance capacity)
        // the trainer's per-iteration densification statistics, where their inputs are produced
        // (/root/reference/scene/gaussian_model.py:735-737 add_densification_stats, trainers/train_static.py:125 max_radii2D)
        stat_grad_accum[i] += sqrtf(acc[0] * acc[0] + acc[1] * acc[1]);
        stat_denom[i] += 1.f;
        if (stat_max_radii) stat_max_radii[i] = fmaxf(stat_max_radii[i], (float)radii[i]);
    }
    if (dcolors) { dcolors[3 * i] = acc[6]; dcolors[3 * i + 1] = acc[7]; dcolors[3 * i + 2] = acc[8]; }
    {   // logit opacities: chain through the sigmoid with the activated value the forward parked in the record
        const float o = rec[(size_t)i * EGS_SPLAT_REC_F4 + 1].y;
        const float go = (vis && (act & EGS_ACT_LOGIT_OPACITY)) ? acc[5] * (o * (1.f - o)) : acc[5];
        if (dopac) dopac[i] = go;
        if (SINK && (fused & (1u << EGS_SINK_OPACITY))) stage[ST_OPAC + tid] = go;
    }
    float gmean[3] = { 0.f, 0.f, 0.f }, g6[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    if (!vis) {
#pragma unroll
        for (int k = 0; k < 3; k++) if (dmeans3D) dmeans3D[3 * i + k] = 0.f;
#pragma unroll
        for (int k = 0; k < 6; k++) if (dcov3D) dcov3D[6 * (size_t)i + k] = 0.f;
        if (dsh) for (int k = 0; k < M * 3; k++) dsh[(size_t)i * M * 3 + k] = 0.f;
        if (dscales) { dscales[3 * i] = 0.f; dscales[3 * i + 1] = 0.f; dscales[3 * i + 2] = 0.f; }
        if (drots) { drots[4 * i] = 0.f; drots[4 * i + 1] = 0.f; drots[4 * i + 2] = 0.f; drots[4 * i + 3] = 0.f; }
        if (SINK) {                                                   // a culled Gaussian still takes its Adam step, with a zero gradient
#pragma unroll
            for (int k = 0; k < 3; k++) { stage[ST_MEANS + 3 * tid + k] = 0.f; stage[ST_SCALES + 3 * tid + k] = 0.f; stage[ST_SH + 3 * tid + k] = 0.f; }
            *reinterpret_cast<float4*>(stage + ST_ROTS + 4 * tid) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const float p[3] = { means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2] };
    float c6[6], s[3] = { 0.f, 0.f, 0.f }, q[4] = { 1.f, 0.f, 0.f, 0.f }, qinv = 1.f;
    if (cov3D_in) {
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = cov3D_in[6 * (size_t)i + k];
    } else {
        s[0] = scales[3 * i]; s[1] = scales[3 * i + 1]; s[2] = scales[3 * i + 2];
        q[0] = rots[4 * i]; q[1] = rots[4 * i + 1]; q[2] = rots[4 * i + 2]; q[3] = rots[4 * i + 3];
        activate_scale_rot(act, s, q, qinv);
        cov3d_from_scale_rot(s, mod, q, c6, (rot.M && (!rot.sel || rot.sel[i])) ? rot.M : nullptr);
    }
    Ewa e; ewa_project(p, c6, V, W, H, tanfovx, tanfovy, e);

    // conic (xx, xy/2, yy) -> cov2D (a,b,c); the published backward regularises 1/det^2 by 1e-7.
    // These three lines are dL/dcov2D = -cov2D^-1 (dL/dconic) cov2D^-1 written out, and the lines after them multiply the result by the
    // covariance again (gm0 = 2 S0 dL_da + S1 dL_db ...): the product only cancels back to "cov2D^-1 x gradient" if the three numbers keep
    // that structure.  In float32 they do not for an elongated splat (a c within 2 % of b^2: every term below is ~60 x the sum, the rounding
    // of each lands in the next lines' cancellation once more) -- dL/dmeans3D of the one recorded parity miss was 7e-4 of the array's
    // maximum off with EXACT blend sums, float32 oracle included (tools/dev/chain_precision.py).  Evaluated in float64 from the same float32
    // (a, b, c) they are the exact formula of a covariance 1e-7 away, and the row lands 2e-6 from the float64 oracle.  ~25 double
    // operations per Gaussian in a kernel that streams 300 B per Gaussian.
    const double gA = (double)acc[2], gB = (double)acc[3], gC = (double)acc[4];
    const double ea = (double)e.a, eb = (double)e.b, ec = (double)e.c;
    const double denom = ea * ec - eb * eb;
    const double d2inv = 1.0 / (denom * denom + 0.0000001);
    float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
    if (d2inv != 0.0) {
        dL_da = (float)(d2inv * (-ec * ec * gA + 2.0 * eb * ec * gB + (denom - ea * ec) * gC));
        dL_dc = (float)(d2inv * (-ea * ea * gC + 2.0 * ea * eb * gB + (denom - ea * ec) * gA));
        dL_db = (float)(d2inv * 2.0 * (eb * ec * gA - (denom + 2.0 * eb * eb) * gB + ea * eb * gC));
        const float* m0 = e.m0; const float* m1 = e.m1;
        g6[0] = m0[0] * m0[0] * dL_da + m0[0] * m1[0] * dL_db + m1[0] * m1[0] * dL_dc;
        g6[3] = m0[1] * m0[1] * dL_da + m0[1] * m1[1] * dL_db + m1[1] * m1[1] * dL_dc;
        g6[5] = m0[2] * m0[2] * dL_da + m0[2] * m1[2] * dL_db + m1[2] * m1[2] * dL_dc;
        g6[1] = 2.f * m0[0] * m0[1] * dL_da + (m0[0] * m1[1] + m0[1] * m1[0]) * dL_db + 2.f * m1[0] * m1[1] * dL_dc;
        g6[2] = 2.f * m0[0] * m0[2] * dL_da + (m0[0] * m1[2] + m0[2] * m1[0]) * dL_db + 2.f * m1[0] * m1[2] * dL_dc;
        g6[4] = 2.f * m0[2] * m0[1] * dL_da + (m0[1] * m1[2] + m0[2] * m1[1]) * dL_db + 2.f * m1[1] * m1[2] * dL_dc;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) if (dcov3D) dcov3D[6 * (size_t)i + k] = g6[k];     // NULL: only the scale / rotation gradients are wanted

    // cov2D -> rows of (J R) -> J -> camera-space point -> mean
    float gJ00 = 0.f, gJ02 = 0.f, gJ11 = 0.f, gJ12 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float gm0 = 2.f * e.S0[k] * dL_da + e.S1[k] * dL_db;
        const float gm1 = 2.f * e.S1[k] * dL_dc + e.S0[k] * dL_db;
        gJ00 += V[4 * k + 0] * gm0; gJ02 += V[4 * k + 2] * gm0;
        gJ11 += V[4 * k + 1] * gm1; gJ12 += V[4 * k + 2] * gm1;
    }
    const float xmask = (e.txtz < -e.limx || e.txtz > e.limx) ? 0.f : 1.f;
    const float ymask = (e.tytz < -e.limy || e.tytz > e.limy) ? 0.f : 1.f;
    const float tz = 1.f / e.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
    const float gtx = xmask * -e.fx * tz2 * gJ02;
    const float gty = ymask * -e.fy * tz2 * gJ12;
    const float gtz = -e.fx * tz2 * gJ00 - e.fy * tz2 * gJ11 + (2.f * e.fx * e.tx) * tz3 * gJ02 +
                      (2.f * e.fy * e.ty) * tz3 * gJ12;
#pragma unroll
    for (int k = 0; k < 3; k++) gmean[k] += V[4 * k + 0] * gtx + V[4 * k + 1] * gty + V[4 * k + 2] * gtz;

    // screen-space mean (NDC-scaled) through the projective divide
    float hom[4]; xform44(p, PM, hom);
    const float mw = 1.f / (hom[3] + 0.0000001f);
    const float mul1 = hom[0] * mw * mw, mul2 = hom[1] * mw * mw;
#pragma unroll
    for (int k = 0; k < 3; k++)
        gmean[k] += (PM[4 * k + 0] * mw - PM[4 * k + 3] * mul1) * acc[0] + (PM[4 * k + 1] * mw - PM[4 * k + 3] * mul2) * acc[1];

    // depth = view.z
    const float mul3 = V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14];
#pragma unroll
    for (int k = 0; k < 3; k++) gmean[k] += (V[4 * k + 2] - V[4 * k + 3] * mul3) * acc[9];

    // colour -> SH coefficients and view direction
    if (shs) {
        const float d0[3] = { p[0] - campos[0], p[1] - campos[1], p[2] - campos[2] };
        const float inv = 1.f / sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
        const float x = d0[0] * inv, y = d0[1] * inv, z = d0[2] * inv;
        const float* sh = shs + (size_t)i * M * 3;
        // (fused SH leaf: M == 1, checked by the caller; without a dsh array the three values go straight to the stage)
        float* gsh = dsh ? dsh + (size_t)i * M * 3 : stage + ST_SH + 3 * tid;
        const uint32_t cl = clamped[i] & EGS_CLAMP_MASK;
        float gdir[3] = { 0.f, 0.f, 0.f };
        for (int ch = 0; ch < 3; ch++) {
            const float g = ((cl >> ch) & 1u) ? 0.f : acc[6 + ch];
#define SH(k) sh[(k) * 3 + ch]
#define GSH(k) gsh[(k) * 3 + ch]
            float dx_ = 0.f, dy_ = 0.f, dz_ = 0.f;
            GSH(0) = kC0 * g;
            if (D > 0) {
                GSH(1) = -kC1 * y * g; GSH(2) = kC1 * z * g; GSH(3) = -kC1 * x * g;
                dx_ = -kC1 * SH(3); dy_ = -kC1 * SH(1); dz_ = kC1 * SH(2);
                if (D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    GSH(4) = kC2[0] * xy * g; GSH(5) = kC2[1] * yz * g; GSH(6) = kC2[2] * (2.f * zz - xx - yy) * g;
                    GSH(7) = kC2[3] * xz * g; GSH(8) = kC2[4] * (xx - yy) * g;
                    dx_ += kC2[0] * y * SH(4) + kC2[2] * 2.f * -x * SH(6) + kC2[3] * z * SH(7) + kC2[4] * 2.f * x * SH(8);
                    dy_ += kC2[0] * x * SH(4) + kC2[1] * z * SH(5) + kC2[2] * 2.f * -y * SH(6) + kC2[4] * 2.f * -y * SH(8);
                    dz_ += kC2[1] * y * SH(5) + kC2[2] * 4.f * z * SH(6) + kC2[3] * x * SH(7);
                    if (D > 2) {
                        GSH(9) = kC3[0] * y * (3.f * xx - yy) * g; GSH(10) = kC3[1] * xy * z * g;
                        GSH(11) = kC3[2] * y * (4.f * zz - xx - yy) * g;
                        GSH(12) = kC3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * g;
                        GSH(13) = kC3[4] * x * (4.f * zz - xx - yy) * g; GSH(14) = kC3[5] * z * (xx - yy) * g;
                        GSH(15) = kC3[6] * x * (xx - 3.f * yy) * g;
                        dx_ += kC3[0] * SH(9) * 6.f * xy + kC3[1] * SH(10) * yz + kC3[2] * SH(11) * -2.f * xy +
                               kC3[3] * SH(12) * -6.f * xz + kC3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                               kC3[5] * SH(14) * 2.f * xz + kC3[6] * SH(15) * 3.f * (xx - yy);
                        dy_ += kC3[0] * SH(9) * 3.f * (xx - yy) + kC3[1] * SH(10) * xz +
                               kC3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + kC3[3] * SH(12) * -6.f * yz +
                               kC3[4] * SH(13) * -2.f * xy + kC3[5] * SH(14) * -2.f * yz + kC3[6] * SH(15) * -6.f * xy;
                        dz_ += kC3[1] * SH(10) * xy + kC3[2] * SH(11) * 8.f * yz + kC3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) +
                               kC3[4] * SH(13) * 8.f * xz + kC3[5] * SH(14) * (xx - yy);
                    }
                }
            }
            for (int k = (D + 1) * (D + 1); k < M; k++) GSH(k) = 0.f;     // coefficients above the active degree
#undef SH
#undef GSH
            gdir[0] += dx_ * g; gdir[1] += dy_ * g; gdir[2] += dz_ * g;
        }
        if (SINK && dsh && (fused & (1u << EGS_SINK_SH))) { stage[ST_SH + 3 * tid] = gsh[0]; stage[ST_SH + 3 * tid + 1] = gsh[1]; stage[ST_SH + 3 * tid + 2] = gsh[2]; }
        const float dot = x * gdir[0] + y * gdir[1] + z * gdir[2];
        gmean[0] += (gdir[0] - x * dot) * inv; gmean[1] += (gdir[1] - y * dot) * inv; gmean[2] += (gdir[2] - z * dot) * inv;
    }
    if (dmeans3D) { dmeans3D[3 * i] = gmean[0]; dmeans3D[3 * i + 1] = gmean[1]; dmeans3D[3 * i + 2] = gmean[2]; }
    if (SINK && (fused & (1u << EGS_SINK_MEANS3D))) { stage[ST_MEANS + 3 * tid] = gmean[0]; stage[ST_MEANS + 3 * tid + 1] = gmean[1]; stage[ST_MEANS + 3 * tid + 2] = gmean[2]; }

    // cov3D -> scale, quaternion (only when the forward built cov3D itself)
    if (!cov3D_in) {
        float Rm[9]; quat_to_rot(q, Rm);
        const float sc[3] = { mod * s[0], mod * s[1], mod * s[2] };
        const float Gs[9] = { g6[0], 0.5f * g6[1], 0.5f * g6[2], 0.5f * g6[1], g6[3], 0.5f * g6[4],
                              0.5f * g6[2], 0.5f * g6[4], g6[5] };
        float L[9], gL[9], gR[9];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int k = 0; k < 3; k++) L[3 * a + k] = Rm[3 * a + k] * sc[k];
        // object rotation (egs_object_rotation; operation for operation k_cov3d_backward of cov3d.hip): L = M L0, dL/dL0 = M^T dL/dL, and
        // the reference's duplicated-index multiplier on row 0 (covariance.py)
        const bool moved = rot.M && (!rot.sel || rot.sel[i]);
        float Mm[9];
        if (moved) {
            float L0[9];
#pragma unroll
            for (int k = 0; k < 9; k++) { Mm[k] = rot.M[k]; L0[k] = L[k]; }
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) L[3 * a + b] = Mm[3 * a] * L0[b] + Mm[3 * a + 1] * L0[3 + b] + Mm[3 * a + 2] * L0[6 + b];
        }
        const float mult = (moved && i == 0) ? (rot.mult_dev ? rot.mult_dev[0] : rot.mult) : 1.f;
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                gL[3 * a + k] = (Gs[3 * a] * L[k] + Gs[3 * a + 1] * L[3 + k] + Gs[3 * a + 2] * L[6 + k]) * (2.f * mult);
        if (moved) {
            float g0[9];
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) g0[3 * a + b] = Mm[a] * gL[b] + Mm[3 + a] * gL[3 + b] + Mm[6 + a] * gL[6 + b];
#pragma unroll
            for (int k = 0; k < 9; k++) gL[k] = g0[k];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float ds = mod * (gL[k] * Rm[k] + gL[3 + k] * Rm[3 + k] + gL[6 + k] * Rm[6 + k]);
            const float dsk = (act & EGS_ACT_LOG_SCALES) ? ds * s[k] : ds;                  // d/d log-scale = d/d scale * scale
            if (dscales) dscales[3 * i + k] = dsk;
            if (SINK && (fused & (1u << EGS_SINK_SCALES))) stage[ST_SCALES + 3 * tid + k] = dsk;
#pragma unroll
            for (int a = 0; a < 3; a++) gR[3 * a + k] = gL[3 * a + k] * sc[k];
        }
        const float r = q[0], qx = q[1], qy = q[2], qz = q[3];
        float gq[4];
        gq[0] = 2.f * (-qz * gR[1] + qy * gR[2] + qz * gR[3] - qx * gR[5] - qy * gR[6] + qx * gR[7]);
        gq[1] = 2.f * (qy * gR[1] + qz * gR[2] + qy * gR[3] - 2.f * qx * gR[4] - r * gR[5] + qz * gR[6] + r * gR[7] - 2.f * qx * gR[8]);
        gq[2] = 2.f * (-2.f * qy * gR[0] + qx * gR[1] + r * gR[2] + qx * gR[3] + qz * gR[5] - r * gR[6] + qz * gR[7] - 2.f * qy * gR[8]);
        gq[3] = 2.f * (-2.f * qz * gR[0] - r * gR[1] + qx * gR[2] + r * gR[3] - 2.f * qz * gR[4] + qy * gR[5] + qx * gR[6] + qy * gR[7]);
        if (act & EGS_ACT_RAW_QUATS) {                                  // back through q = q_raw / |q_raw|
            const float dot = q[0] * gq[0] + q[1] * gq[1] + q[2] * gq[2] + q[3] * gq[3];
#pragma unroll
            for (int k = 0; k < 4; k++) gq[k] = (gq[k] - q[k] * dot) * qinv;
        }
        if (drots) { drots[4 * i + 0] = gq[0]; drots[4 * i + 1] = gq[1]; drots[4 * i + 2] = gq[2]; drots[4 * i + 3] = gq[3]; }
        if (SINK && (fused & (1u << EGS_SINK_ROTATIONS))) *reinterpret_cast<float4*>(stage + ST_ROTS + 4 * tid) = make_float4(gq[0], gq[1], gq[2], gq[3]);
    }
}

// One lane per Gaussian.  SINK (the optimizer fused into the backward, include/egs_raster.h egs_backward_adam): instead of (or besides)
// writing the gradients of the leaves it owns, the workgroup parks them in LDS in array order and then takes the Adam step of ITS 256
// rows of every such leaf with the stand-alone kernel's arithmetic (egs_adam1) and access pattern (float4, fully coalesced:
// rows 256 b .. 256 b + 255 of a [P, k] array are one contiguous span).  The gradients never reach HBM and the parameters are not
// read a second time by another launch: 16 B in + 12 B out per element become 12 + 12, and the step has one launch less.
// 896 float4 tasks per workgroup (64 x (3 + 1 + 3 + 4 + 3)), task -> leaf boundaries fall on wave boundaries.
template <bool SINK>
__global__ __launch_bounds__(256) void k_preprocess_backward(
    int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ shs,
    const float* __restrict__ scales, float mod, const float* __restrict__ rots, const float* __restrict__ cov3D_in, int act,
    const float* __restrict__ V, const float* __restrict__ PM, const float* __restrict__ campos, int W, int H,
    float tanfovx, float tanfovy, const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped,
    const float4* __restrict__ rec, const float* __restrict__ grad_acc, float* __restrict__ dmeans2D, float* __restrict__ dcolors,
    float* __restrict__ dopac, float* __restrict__ dmeans3D, float* __restrict__ dcov3D, float* __restrict__ dsh,
    float* __restrict__ dscales, float* __restrict__ drots,
    float* __restrict__ stat_grad_accum, float* __restrict__ stat_denom, float* __restrict__ stat_max_radii,
    const uint32_t* __restrict__ skip_flag, EgsSink sink, EgsObjRot rot) {
    __shared__ __attribute__((aligned(16))) float stage[SINK ? 4 * EGS_SINK_TASKS : 4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned fused = 0;
    if (SINK) {
#pragma unroll
        for (int l = 0; l < EGS_SINK_PP_LEAVES; l++) fused |= sink.leaf[l].p ? (1u << l) : 0u;
    }
    if (i < P)
        pp_bwd_one<SINK>(i, D, M, means3D, shs, scales, mod, rots, cov3D_in, act, V, PM, campos, W, H, tanfovx, tanfovy, radii, clamped, rec,
                         grad_acc, grad_acc + (size_t)P * EGS_GRAD_STRIDE, egs_hot_slots((size_t)P), dmeans2D, dcolors, dopac, dmeans3D, dcov3D, dsh, dscales, drots, stat_grad_accum, stat_denom,
                         stat_max_radii, skip_flag, stage, fused, rot);
    if (!SINK) return;
    __syncthreads();
    if (sink.skip && *sink.skip) return;                            // the frame overflowed its instance capacity: no step (and none was counted)
    const int rows = sink.active_rows ? min(P, max(*sink.active_rows, 0)) : P;      // capacity-sized model: live rows only
    const int row0 = blockIdx.x * 256;
    const int live = min(256, rows - row0);                          // rows of this workgroup that take the step
    if (live <= 0) return;
    constexpr int RF[EGS_SINK_PP_LEAVES] = { 3, 1, 3, 4, 3 };
    constexpr int T0[EGS_SINK_PP_LEAVES + 1] = { 0, 192, 256, 448, 704, 896 };
    // all loads of the thread's (up to four) tasks first, then the arithmetic, then the stores
    float4 Pq[4], Mq[4], Vq[4]; int cnt[4]; float* pp[4]; float* pm[4]; float* pv[4]; float ss[4], ib[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int task = (int)threadIdx.x + 256 * k;
        cnt[k] = 0;
#pragma unroll
        for (int l = 0; l < EGS_SINK_PP_LEAVES; l++) {
            if (task < T0[l] || task >= T0[l + 1] || !sink.leaf[l].p) continue;
            const int e = 4 * (task - T0[l]);                         // first element of the task inside the workgroup's span of leaf l
            const size_t off = (size_t)row0 * RF[l] + e;
            pp[k] = sink.leaf[l].p + off; pm[k] = sink.leaf[l].m + off; pv[k] = sink.leaf[l].v + off;
            cnt[k] = max(0, min(4, live * RF[l] - e));
            ss[k] = sink.coef[2 * l]; ib[k] = sink.coef[2 * l + 1];
        }
        if (cnt[k] == 4 && ((((size_t)pp[k]) | ((size_t)pm[k]) | ((size_t)pv[k])) & 15) == 0) {
            Pq[k] = *reinterpret_cast<const float4*>(pp[k]); Mq[k] = *reinterpret_cast<const float4*>(pm[k]); Vq[k] = *reinterpret_cast<const float4*>(pv[k]);
        } else if (cnt[k] > 0) {
            float t[12];
#pragma unroll
            for (int j = 0; j < 4; j++) { const bool ok = j < cnt[k]; t[j] = ok ? pp[k][j] : 0.f; t[4 + j] = ok ? pm[k][j] : 0.f; t[8 + j] = ok ? pv[k][j] : 0.f; }
            Pq[k] = make_float4(t[0], t[1], t[2], t[3]); Mq[k] = make_float4(t[4], t[5], t[6], t[7]); Vq[k] = make_float4(t[8], t[9], t[10], t[11]);
            cnt[k] = -cnt[k];                                          // negative: element-wise stores
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (cnt[k] == 0) continue;
        const float4 G = *reinterpret_cast<const float4*>(stage + 4 * ((int)threadIdx.x + 256 * k));
        float4 Pn = Pq[k], Mn = Mq[k], Vn = Vq[k];
        egs_adam1(Pn.x, G.x, Mn.x, Vn.x, sink.b1, sink.b2, sink.eps, ss[k], ib[k]); egs_adam1(Pn.y, G.y, Mn.y, Vn.y, sink.b1, sink.b2, sink.eps, ss[k], ib[k]);
        egs_adam1(Pn.z, G.z, Mn.z, Vn.z, sink.b1, sink.b2, sink.eps, ss[k], ib[k]); egs_adam1(Pn.w, G.w, Mn.w, Vn.w, sink.b1, sink.b2, sink.eps, ss[k], ib[k]);
        if (cnt[k] == 4) {
            *reinterpret_cast<float4*>(pp[k]) = Pn; *reinterpret_cast<float4*>(pm[k]) = Mn; *reinterpret_cast<float4*>(pv[k]) = Vn;
        } else {
            const float a[4] = { Pn.x, Pn.y, Pn.z, Pn.w }, b_[4] = { Mn.x, Mn.y, Mn.z, Mn.w }, c_[4] = { Vn.x, Vn.y, Vn.z, Vn.w };
#pragma unroll
            for (int j = 0; j < 4; j++) if (j < -cnt[k]) { pp[k][j] = a[j]; pm[k][j] = b_[j]; pv[k][j] = c_[j]; }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Spherical harmonics with more than one coefficient (or given as separate DC / rest arrays).  A Gaussian's row is 12 M
// bytes; one lane walking its own row touches a different cache line per load, and the M * 3 gradient stores of the backward
// are 4-byte writes 12 M bytes apart.  Here a wave owns 64 consecutive Gaussians and moves their rows as ONE contiguous block:
// float4 loads / stores in lane order (full lines), transposed through an LDS tile of 64 x (3 M + 1) words whose odd row
// stride makes the per-lane row accesses conflict-free.  Rows of invisible Gaussians and coefficients above the active degree
// are not loaded; their gradient is written as zeros by the same coalesced stores.
//   k_sh_forward   after k_preprocess: colour + clamp flags into the record it left open
//   k_sh_backward  after k_preprocess_backward: reads dL/dcolour, writes dL/dSH, adds the view-direction term to dL/dmean3D
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sh_row_col(int e, int rw, float inv_rw, int& r, int& c) {
    r = (int)(((float)e + 0.5f) * inv_rw);                             // e < 64 * 48: exact
    c = e - r * rw;
}

// Rows [i0, i0 + nvalid) of a [P, rw] float array -> tile[row * ld + col0 + col]; elements of invisible rows or with col >= need are
// left alone (never read).
__device__ __forceinline__ void sh_load_rows(float* tile, int ld, int col0, const float* __restrict__ src, int rw, int i0, int nvalid,
                                             uint64_t vmask, int need, unsigned lane) {
    if (need <= 0) return;
    const float inv_rw = 1.f / (float)rw;
    const float* base = src + (size_t)i0 * rw;
    const int n = nvalid * rw;
    if ((((uintptr_t)base) & 15) == 0 && (n & 3) == 0 && rw <= 48) {
        // all (up to 12) loads of the lane are issued before the first one is used: one HBM round trip per wave, not twelve
        const float4* b4 = reinterpret_cast<const float4*>(base);
        float4 v[12]; bool want[12];
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const int q = (int)lane + 64 * k;
            int r0, c0, r3, c3;
            sh_row_col(4 * q, rw, inv_rw, r0, c0); sh_row_col(4 * q + 3, rw, inv_rw, r3, c3);
            want[k] = 4 * q < n && ((((vmask >> r0) & 1) && c0 < need) || (r3 != r0 && ((vmask >> r3) & 1)) ||
                                    (r3 - r0 > 1 && ((vmask >> (r0 + 1)) & 1)));
            if (want[k]) v[k] = b4[q];
        }
#pragma unroll
        for (int k = 0; k < 12; k++) {
            if (!want[k]) continue;
            const int q = (int)lane + 64 * k;
            const float vv[4] = { v[k].x, v[k].y, v[k].z, v[k].w };
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int r, c; sh_row_col(4 * q + j, rw, inv_rw, r, c);
                tile[r * ld + col0 + c] = vv[j];
            }
        }
    } else {
        for (int e = lane; e < n; e += 64) {
            int r, c; sh_row_col(e, rw, inv_rw, r, c);
            if (((vmask >> r) & 1) && c < need) tile[r * ld + col0 + c] = base[e];
        }
    }
}

// tile[row * ld + col0 + col] -> rows [i0, i0 + nvalid) of a [P, rw] float array, every element.
__device__ __forceinline__ void sh_store_rows(const float* tile, int ld, int col0, float* __restrict__ dst, int rw, int i0, int nvalid,
                                              unsigned lane) {
    const float inv_rw = 1.f / (float)rw;
    float* base = dst + (size_t)i0 * rw;
    const int n = nvalid * rw;
    if ((((uintptr_t)base) & 15) == 0 && (n & 3) == 0) {
        float4* b4 = reinterpret_cast<float4*>(base);
        for (int q = lane; 4 * q < n; q += 64) {
            float vv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int r, c; sh_row_col(4 * q + j, rw, inv_rw, r, c);
                vv[j] = tile[r * ld + col0 + c];
            }
            b4[q] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        }
    } else {
        for (int e = lane; e < n; e += 64) {
            int r, c; sh_row_col(e, rw, inv_rw, r, c);
            base[e] = tile[r * ld + col0 + c];
        }
    }
}

extern __shared__ __attribute__((aligned(16))) float sh_tile[];

// sh_rest == nullptr: sh_a is [P, M, 3]; otherwise sh_a is the DC block [P, 1, 3] and sh_rest [P, M - 1, 3].
__global__ __launch_bounds__(64) void k_sh_forward(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ campos,
                                                   const float* __restrict__ sh_a, const float* __restrict__ sh_rest,
                                                   const uint8_t* __restrict__ visible, float4* __restrict__ rec, uint8_t* __restrict__ clamped) {
    const unsigned lane = threadIdx.x;
    const int i0 = blockIdx.x * 64, i = i0 + (int)lane;
    const bool vis = i < P && visible[i] != 0;
    const uint64_t vmask = __ballot(vis);
    if (!vmask) return;
    const int rw = 3 * M, ld = rw + 1, nvalid = min(64, P - i0), need = 3 * (D + 1) * (D + 1);
    if (sh_rest) {
        sh_load_rows(sh_tile, ld, 0, sh_a, 3, i0, nvalid, vmask, 3, lane);
        sh_load_rows(sh_tile, ld, 3, sh_rest, rw - 3, i0, nvalid, vmask, need - 3, lane);
    } else {
        sh_load_rows(sh_tile, ld, 0, sh_a, rw, i0, nvalid, vmask, need, lane);
    }
    __syncthreads();                                                   // one wave: orders the LDS writes before the row reads
    if (!vis) return;
    const float p[3] = { means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2] };
    float dir[3] = { p[0] - campos[0], p[1] - campos[1], p[2] - campos[2] };
    const float inv = 1.f / sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    dir[0] *= inv; dir[1] *= inv; dir[2] *= inv;
    const float* sh = sh_tile + lane * ld;
    float rgb[3]; uint32_t cl = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float v = sh_channel(D, sh, ch, dir[0], dir[1], dir[2]);
        if (v < 0.f) cl |= 1u << ch;
        rgb[ch] = fmaxf(v, 0.f);
    }
    clamped[i] = (uint8_t)((clamped[i] & ~EGS_CLAMP_MASK) | cl);      // (bits 3-5: the HOT code k_preprocess left)
    float* r = reinterpret_cast<float*>(rec + (size_t)i * EGS_SPLAT_REC_F4);
    r[6] = rgb[0]; r[7] = rgb[1]; r[8] = rgb[2];
}

__global__ __launch_bounds__(64) void k_sh_backward(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ campos,
                                                    const float* __restrict__ sh_a, const float* __restrict__ sh_rest,
                                                    const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped,
                                                    const float* __restrict__ dcolors, float* __restrict__ dsh_a, float* __restrict__ dsh_rest,
                                                    float* __restrict__ dmeans3D) {
    const unsigned lane = threadIdx.x;
    const int i0 = blockIdx.x * 64, i = i0 + (int)lane;
    const bool vis = i < P && radii[i] > 0;
    const uint64_t vmask = __ballot(vis);
    const int rw = 3 * M, ld = rw + 1, nvalid = min(64, P - i0), need = 3 * (D + 1) * (D + 1);
    float* row = sh_tile + lane * ld;
    if (vmask && D > 0) {                                              // the coefficients only enter through the view-direction term
        if (sh_rest) sh_load_rows(sh_tile, ld, 3, sh_rest, rw - 3, i0, nvalid, vmask, need - 3, lane);
        else sh_load_rows(sh_tile, ld, 0, sh_a, rw, i0, nvalid, vmask, need, lane);
        __syncthreads();
    }
    if (vis) {
        const float p[3] = { means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2] };
        const float d0[3] = { p[0] - campos[0], p[1] - campos[1], p[2] - campos[2] };
        const float inv = 1.f / sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
        const float x = d0[0] * inv, y = d0[1] * inv, z = d0[2] * inv;
        const uint32_t cl = clamped[i] & EGS_CLAMP_MASK;
        float gdir[3] = { 0.f, 0.f, 0.f };
        for (int ch = 0; ch < 3; ch++) {
            const float g = ((cl >> ch) & 1u) ? 0.f : dcolors[3 * i + ch];
#define SH(k) row[(k) * 3 + ch]
            float dx_ = 0.f, dy_ = 0.f, dz_ = 0.f;                    // reads of this channel's coefficients first, then its gradient in place
            if (D > 0) {
                dx_ = -kC1 * SH(3); dy_ = -kC1 * SH(1); dz_ = kC1 * SH(2);
                if (D > 1) {
                    dx_ += kC2[0] * y * SH(4) + kC2[2] * 2.f * -x * SH(6) + kC2[3] * z * SH(7) + kC2[4] * 2.f * x * SH(8);
                    dy_ += kC2[0] * x * SH(4) + kC2[1] * z * SH(5) + kC2[2] * 2.f * -y * SH(6) + kC2[4] * 2.f * -y * SH(8);
                    dz_ += kC2[1] * y * SH(5) + kC2[2] * 4.f * z * SH(6) + kC2[3] * x * SH(7);
                    if (D > 2) {
                        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        dx_ += kC3[0] * SH(9) * 6.f * xy + kC3[1] * SH(10) * yz + kC3[2] * SH(11) * -2.f * xy +
                               kC3[3] * SH(12) * -6.f * xz + kC3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                               kC3[5] * SH(14) * 2.f * xz + kC3[6] * SH(15) * 3.f * (xx - yy);
                        dy_ += kC3[0] * SH(9) * 3.f * (xx - yy) + kC3[1] * SH(10) * xz +
                               kC3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + kC3[3] * SH(12) * -6.f * yz +
                               kC3[4] * SH(13) * -2.f * xy + kC3[5] * SH(14) * -2.f * yz + kC3[6] * SH(15) * -6.f * xy;
                        dz_ += kC3[1] * SH(10) * xy + kC3[2] * SH(11) * 8.f * yz + kC3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) +
                               kC3[4] * SH(13) * 8.f * xz + kC3[5] * SH(14) * (xx - yy);
                    }
                }
            }
            gdir[0] += dx_ * g; gdir[1] += dy_ * g; gdir[2] += dz_ * g;
            SH(0) = kC0 * g;
            if (D > 0) {
                SH(1) = -kC1 * y * g; SH(2) = kC1 * z * g; SH(3) = -kC1 * x * g;
                if (D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    SH(4) = kC2[0] * xy * g; SH(5) = kC2[1] * yz * g; SH(6) = kC2[2] * (2.f * zz - xx - yy) * g;
                    SH(7) = kC2[3] * xz * g; SH(8) = kC2[4] * (xx - yy) * g;
                    if (D > 2) {
                        SH(9) = kC3[0] * y * (3.f * xx - yy) * g; SH(10) = kC3[1] * xy * z * g;
                        SH(11) = kC3[2] * y * (4.f * zz - xx - yy) * g;
                        SH(12) = kC3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * g;
                        SH(13) = kC3[4] * x * (4.f * zz - xx - yy) * g; SH(14) = kC3[5] * z * (xx - yy) * g;
                        SH(15) = kC3[6] * x * (xx - 3.f * yy) * g;
                    }
                }
            }
            for (int k = (D + 1) * (D + 1); k < M; k++) SH(k) = 0.f;      // coefficients above the active degree
#undef SH
        }
        if (D > 0) {
            const float dot = x * gdir[0] + y * gdir[1] + z * gdir[2];
            dmeans3D[3 * i] += (gdir[0] - x * dot) * inv; dmeans3D[3 * i + 1] += (gdir[1] - y * dot) * inv;
            dmeans3D[3 * i + 2] += (gdir[2] - z * dot) * inv;
        }
    } else {
        for (int k = 0; k < rw; k++) row[k] = 0.f;
    }
    __syncthreads();
    if (dsh_rest) {
        sh_store_rows(sh_tile, ld, 0, dsh_a, 3, i0, nvalid, lane);
        sh_store_rows(sh_tile, ld, 3, dsh_rest, rw - 3, i0, nvalid, lane);
    } else {
        sh_store_rows(sh_tile, ld, 0, dsh_a, rw, i0, nvalid, lane);
    }
}

// ---- M = 16 (the reference's max_sh_degree = 3), 16-byte aligned arrays: the same two kernels with the index arithmetic of the
// transposition removed.
//   CAT   [P,16,3]: float4 q of the block goes to LDS slot q + q / 12, i.e. rows of 13 float4 -- an odd stride, so the twelve
//         ds_read_b128 / ds_write_b128 a lane does on its own row are conflict-free.
//   SPLIT [P,1,3] + [P,15,3]: both blocks are copied linearly; rows of 3 and 45 words are odd strides already (scalar row accesses).
// A lane keeps its row (48 coefficients, then 48 gradients) in registers.
#define SH16_CAT 1
#define SH16_SPLIT 2
#define SH16_REST_WORDS (64 * 45)

// gather this block's rows into LDS; `need` = words of a row's head that will be read (3 (D+1)^2)
template <int MODE>
__device__ __forceinline__ void sh16_stage_in(float* tile, const float* __restrict__ sh_a, const float* __restrict__ sh_rest, int i0, int nvalid,
                                              uint64_t vmask, int need, unsigned lane) {
    if (MODE == SH16_CAT) {
        const float4* b4 = reinterpret_cast<const float4*>(sh_a + (size_t)i0 * 48);
        float4* t4 = reinterpret_cast<float4*>(tile);
        // (twelve named registers, not an array: the compiler parks an array of conditionally loaded float4 in scratch, which
        // serialises the loads)
#define SH16_LD(k) float4 v##k = make_float4(0.f, 0.f, 0.f, 0.f); bool w##k;                                   \
        { const unsigned q = lane + 64u * k, row = q / 12u, j = q - row * 12u;                                 \
          w##k = (int)row < nvalid && ((vmask >> row) & 1) && (int)(4u * j) < need; if (w##k) v##k = b4[q]; }
#define SH16_PUT(k) { const unsigned q = lane + 64u * k; if (w##k) t4[q + q / 12u] = v##k; }
        SH16_LD(0) SH16_LD(1) SH16_LD(2) SH16_LD(3) SH16_LD(4) SH16_LD(5) SH16_LD(6) SH16_LD(7) SH16_LD(8) SH16_LD(9) SH16_LD(10) SH16_LD(11)
        SH16_PUT(0) SH16_PUT(1) SH16_PUT(2) SH16_PUT(3) SH16_PUT(4) SH16_PUT(5) SH16_PUT(6) SH16_PUT(7) SH16_PUT(8) SH16_PUT(9) SH16_PUT(10) SH16_PUT(11)
#undef SH16_LD
#undef SH16_PUT
    } else {
        float* tdc = tile + SH16_REST_WORDS;
        const int n_dc = nvalid * 3, n_rest = nvalid * 45, need_rest = need - 3;
        const float* bdc = sh_a + (size_t)i0 * 3;
        const float* brest = sh_rest + (size_t)i0 * 45;
        float4 vd = make_float4(0.f, 0.f, 0.f, 0.f); bool wd = false;
        {
            const int e = 4 * (int)lane;
            wd = e + 3 < n_dc && ((vmask >> (e / 3)) | (vmask >> ((e + 3) / 3))) & 1;
            if (wd) vd = reinterpret_cast<const float4*>(bdc)[lane];
        }
#define SH16_LD(k) float4 v##k = make_float4(0.f, 0.f, 0.f, 0.f); bool w##k;                                   \
        { const int e = 4 * ((int)lane + 64 * k), r0 = e / 45, r1 = (e + 3) / 45;                              \
          w##k = need_rest > 0 && e + 3 < n_rest &&                                                            \
                 ((((vmask >> r0) & 1) && e - 45 * r0 < need_rest) || (r1 != r0 && ((vmask >> r1) & 1)));      \
          if (w##k) v##k = reinterpret_cast<const float4*>(brest)[lane + 64 * k]; }
#define SH16_PUT(k) if (w##k) reinterpret_cast<float4*>(tile)[lane + 64 * k] = v##k;
        SH16_LD(0) SH16_LD(1) SH16_LD(2) SH16_LD(3) SH16_LD(4) SH16_LD(5) SH16_LD(6) SH16_LD(7) SH16_LD(8) SH16_LD(9) SH16_LD(10) SH16_LD(11)
        if (wd) reinterpret_cast<float4*>(tdc)[lane] = vd;
        SH16_PUT(0) SH16_PUT(1) SH16_PUT(2) SH16_PUT(3) SH16_PUT(4) SH16_PUT(5) SH16_PUT(6) SH16_PUT(7) SH16_PUT(8) SH16_PUT(9) SH16_PUT(10) SH16_PUT(11)
#undef SH16_LD
#undef SH16_PUT
        if (lane < 3) {                                                // ragged last block: the words after the last whole float4
            const int ed = (n_dc & ~3) + (int)lane, er = (n_rest & ~3) + (int)lane;
            if (ed < n_dc) tdc[ed] = bdc[ed];
            if (er < n_rest && need_rest > 0) tile[er] = brest[er];
        }
    }
}

template <int MODE>
__device__ __forceinline__ void sh16_row_to_regs(const float* tile, unsigned lane, int need, float (&c)[48]) {
    if (MODE == SH16_CAT) {
        const float4* t4 = reinterpret_cast<const float4*>(tile) + lane * 13u;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (4 * j < need) t = t4[j];
            c[4 * j] = t.x; c[4 * j + 1] = t.y; c[4 * j + 2] = t.z; c[4 * j + 3] = t.w;
        }
    } else {
        const float* tdc = tile + SH16_REST_WORDS + lane * 3u;
        const float* tr = tile + lane * 45u;
        c[0] = tdc[0]; c[1] = tdc[1]; c[2] = tdc[2];
#pragma unroll
        for (int t = 0; t < 45; t++) c[3 + t] = t < need - 3 ? tr[t] : 0.f;
    }
}

template <int MODE>
__device__ __forceinline__ void sh16_regs_to_row(float* tile, unsigned lane, const float (&g)[48]) {
    if (MODE == SH16_CAT) {
        float4* t4 = reinterpret_cast<float4*>(tile) + lane * 13u;
#pragma unroll
        for (int j = 0; j < 12; j++) t4[j] = make_float4(g[4 * j], g[4 * j + 1], g[4 * j + 2], g[4 * j + 3]);
    } else {
        float* tdc = tile + SH16_REST_WORDS + lane * 3u;
        float* tr = tile + lane * 45u;
        tdc[0] = g[0]; tdc[1] = g[1]; tdc[2] = g[2];
#pragma unroll
        for (int t = 0; t < 45; t++) tr[t] = g[3 + t];
    }
}

// LDS rows -> this block's rows of the gradient arrays, every word, in lane order
template <int MODE>
__device__ __forceinline__ void sh16_stage_out(const float* tile, float* __restrict__ dsh_a, float* __restrict__ dsh_rest, int i0, int nvalid,
                                               unsigned lane) {
    if (MODE == SH16_CAT) {
        float4* b4 = reinterpret_cast<float4*>(dsh_a + (size_t)i0 * 48);
        const float4* t4 = reinterpret_cast<const float4*>(tile);
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const unsigned q = lane + 64u * k, row = q / 12u;
            if ((int)row < nvalid) b4[q] = t4[q + row];
        }
    } else {
        const float* tdc = tile + SH16_REST_WORDS;
        const int n_dc = nvalid * 3, n_rest = nvalid * 45;
        float* bdc = dsh_a + (size_t)i0 * 3;
        float* brest = dsh_rest + (size_t)i0 * 45;
        if (4 * (int)lane + 3 < n_dc) reinterpret_cast<float4*>(bdc)[lane] = reinterpret_cast<const float4*>(tdc)[lane];
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const int q = (int)lane + 64 * k;
            if (4 * q + 3 < n_rest) reinterpret_cast<float4*>(brest)[q] = reinterpret_cast<const float4*>(tile)[q];
        }
        if (lane < 3) {
            const int ed = (n_dc & ~3) + (int)lane, er = (n_rest & ~3) + (int)lane;
            if (ed < n_dc) bdc[ed] = tdc[ed];
            if (er < n_rest) brest[er] = tile[er];
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(64) void k_sh16_forward(int P, int D, const float* __restrict__ means3D, const float* __restrict__ campos,
                                                     const float* __restrict__ sh_a, const float* __restrict__ sh_rest,
                                                     const uint8_t* __restrict__ visible, float4* __restrict__ rec, uint8_t* __restrict__ clamped) {
    const unsigned lane = threadIdx.x;
    const int i0 = blockIdx.x * 64, i = i0 + (int)lane;
    const bool inb = i < P;
    const bool vis = inb && visible[i] != 0;
    float p[3] = { 0.f, 0.f, 1.f };
    if (inb) { p[0] = means3D[3 * i]; p[1] = means3D[3 * i + 1]; p[2] = means3D[3 * i + 2]; }   // (in flight together with the rows)
    const uint64_t vmask = __ballot(vis);
    if (!vmask) return;
    const int need = 3 * (D + 1) * (D + 1);
    sh16_stage_in<MODE>(sh_tile, sh_a, sh_rest, i0, min(64, P - i0), vmask, need, lane);
    __syncthreads();
    if (!vis) return;
    float c[48];
    sh16_row_to_regs<MODE>(sh_tile, lane, need, c);
    float dir[3] = { p[0] - campos[0], p[1] - campos[1], p[2] - campos[2] };
    const float inv = 1.f / sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    dir[0] *= inv; dir[1] *= inv; dir[2] *= inv;
    float rgb[3]; uint32_t cl = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float v = sh_channel(D, c, ch, dir[0], dir[1], dir[2]);
        if (v < 0.f) cl |= 1u << ch;
        rgb[ch] = fmaxf(v, 0.f);
    }
    clamped[i] = (uint8_t)((clamped[i] & ~EGS_CLAMP_MASK) | cl);      // (bits 3-5: the HOT code k_preprocess left)
    float* r = reinterpret_cast<float*>(rec + (size_t)i * EGS_SPLAT_REC_F4);
    r[6] = rgb[0]; r[7] = rgb[1]; r[8] = rgb[2];
}

// One span of `n` floats (n4 = n / 4 float4 + a tail) of a leaf stepped with the gradients in `t` (LDS, same order): the stand-alone
// optimizer kernel's arithmetic on the store pattern of sh16_stage_out.
__device__ __forceinline__ void sh16_adam_span(const float* t, const EgsSinkLeaf& f, size_t first, int n, float ss, float ib, float b1, float b2,
                                               float eps, unsigned lane) {
    float* __restrict__ p = f.p + first; float* __restrict__ m = f.m + first; float* __restrict__ v = f.v + first;
    const bool vec = ((((size_t)p) | ((size_t)m) | ((size_t)v)) & 15) == 0;
    for (int q0 = 0; 4 * q0 < n; q0 += 4 * 64) {                      // four float4 per lane and round: twelve loads in flight
        float4 P4[4], M4[4], V4[4]; bool on[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int q = q0 + (int)lane + 64 * k;
            on[k] = vec && 4 * q + 3 < n;
            if (on[k]) { P4[k] = reinterpret_cast<const float4*>(p)[q]; M4[k] = reinterpret_cast<const float4*>(m)[q]; V4[k] = reinterpret_cast<const float4*>(v)[q]; }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (!on[k]) continue;
            const int q = q0 + (int)lane + 64 * k;
            const float4 G = reinterpret_cast<const float4*>(t)[q];
            egs_adam1(P4[k].x, G.x, M4[k].x, V4[k].x, b1, b2, eps, ss, ib); egs_adam1(P4[k].y, G.y, M4[k].y, V4[k].y, b1, b2, eps, ss, ib);
            egs_adam1(P4[k].z, G.z, M4[k].z, V4[k].z, b1, b2, eps, ss, ib); egs_adam1(P4[k].w, G.w, M4[k].w, V4[k].w, b1, b2, eps, ss, ib);
            reinterpret_cast<float4*>(p)[q] = P4[k]; reinterpret_cast<float4*>(m)[q] = M4[k]; reinterpret_cast<float4*>(v)[q] = V4[k];
        }
    }
    const int done = vec ? (n & ~3) : 0;                              // the tail (and everything, if a base is not 16-byte aligned) element by element
    for (int e = done + (int)lane; e < n; e += 64) {
        float P1 = p[e], M1 = m[e], V1 = v[e];
        egs_adam1(P1, t[e], M1, V1, b1, b2, eps, ss, ib);
        p[e] = P1; m[e] = M1; v[e] = V1;
    }
}

// SINK (split arrays only; include/egs_raster.h egs_backward_adam): the launch also takes the Adam step of the leaves whose gradient it
// finishes -- features_dc, features_rest and, because the view-direction term lands here, the positions -- for its 64 rows.
template <int MODE, bool SINK>
__global__ __launch_bounds__(64) void k_sh16_backward(int P, int D, const float* __restrict__ means3D, const float* __restrict__ campos,
                                                      const float* __restrict__ sh_a, const float* __restrict__ sh_rest,
                                                      const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped,
                                                      const float* __restrict__ dcolors, float* __restrict__ dsh_a, float* __restrict__ dsh_rest,
                                                      float* __restrict__ dmeans3D, EgsSink sink) {
    const unsigned lane = threadIdx.x;
    const int i0 = blockIdx.x * 64, i = i0 + (int)lane;
    const bool inb = i < P;
    const bool vis = inb && radii[i] > 0;
    float p[3] = { 0.f, 0.f, 1.f }, dc[3] = { 0.f, 0.f, 0.f }, gm[3] = { 0.f, 0.f, 0.f };
    uint32_t cl = 0;
    if (inb) {                                                         // everything the row math needs besides the rows, issued up front
        p[0] = means3D[3 * i]; p[1] = means3D[3 * i + 1]; p[2] = means3D[3 * i + 2];
        dc[0] = dcolors[3 * i]; dc[1] = dcolors[3 * i + 1]; dc[2] = dcolors[3 * i + 2];
        cl = clamped[i] & EGS_CLAMP_MASK;
        if (D > 0 || (SINK && sink.leaf[EGS_SINK_MEANS3D].p)) { gm[0] = dmeans3D[3 * i]; gm[1] = dmeans3D[3 * i + 1]; gm[2] = dmeans3D[3 * i + 2]; }
    }
    const uint64_t vmask = __ballot(vis);
    const int nvalid = min(64, P - i0), need = 3 * (D + 1) * (D + 1);
    float c[48], g[48];
#pragma unroll
    for (int k = 0; k < 48; k++) { c[k] = 0.f; g[k] = 0.f; }
    if (vmask && D > 0) {                                              // the coefficients only enter through the view-direction term
        sh16_stage_in<MODE>(sh_tile, sh_a, sh_rest, i0, nvalid, vmask, need, lane);
        __syncthreads();
        if (vis) sh16_row_to_regs<MODE>(sh_tile, lane, need, c);
    }
    if (vis) {
        const float d0[3] = { p[0] - campos[0], p[1] - campos[1], p[2] - campos[2] };
        const float inv = 1.f / sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
        const float x = d0[0] * inv, y = d0[1] * inv, z = d0[2] * inv;
        float gdir[3] = { 0.f, 0.f, 0.f };
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float gg = ((cl >> ch) & 1u) ? 0.f : dc[ch];
#define SH(k) c[(k) * 3 + ch]
#define GSH(k) g[(k) * 3 + ch]
            float dx_ = 0.f, dy_ = 0.f, dz_ = 0.f;
            GSH(0) = kC0 * gg;
            if (D > 0) {
                GSH(1) = -kC1 * y * gg; GSH(2) = kC1 * z * gg; GSH(3) = -kC1 * x * gg;
                dx_ = -kC1 * SH(3); dy_ = -kC1 * SH(1); dz_ = kC1 * SH(2);
                if (D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    GSH(4) = kC2[0] * xy * gg; GSH(5) = kC2[1] * yz * gg; GSH(6) = kC2[2] * (2.f * zz - xx - yy) * gg;
                    GSH(7) = kC2[3] * xz * gg; GSH(8) = kC2[4] * (xx - yy) * gg;
                    dx_ += kC2[0] * y * SH(4) + kC2[2] * 2.f * -x * SH(6) + kC2[3] * z * SH(7) + kC2[4] * 2.f * x * SH(8);
                    dy_ += kC2[0] * x * SH(4) + kC2[1] * z * SH(5) + kC2[2] * 2.f * -y * SH(6) + kC2[4] * 2.f * -y * SH(8);
                    dz_ += kC2[1] * y * SH(5) + kC2[2] * 4.f * z * SH(6) + kC2[3] * x * SH(7);
                    if (D > 2) {
                        GSH(9) = kC3[0] * y * (3.f * xx - yy) * gg; GSH(10) = kC3[1] * xy * z * gg;
                        GSH(11) = kC3[2] * y * (4.f * zz - xx - yy) * gg;
                        GSH(12) = kC3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * gg;
                        GSH(13) = kC3[4] * x * (4.f * zz - xx - yy) * gg; GSH(14) = kC3[5] * z * (xx - yy) * gg;
                        GSH(15) = kC3[6] * x * (xx - 3.f * yy) * gg;
                        dx_ += kC3[0] * SH(9) * 6.f * xy + kC3[1] * SH(10) * yz + kC3[2] * SH(11) * -2.f * xy +
                               kC3[3] * SH(12) * -6.f * xz + kC3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                               kC3[5] * SH(14) * 2.f * xz + kC3[6] * SH(15) * 3.f * (xx - yy);
                        dy_ += kC3[0] * SH(9) * 3.f * (xx - yy) + kC3[1] * SH(10) * xz +
                               kC3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + kC3[3] * SH(12) * -6.f * yz +
                               kC3[4] * SH(13) * -2.f * xy + kC3[5] * SH(14) * -2.f * yz + kC3[6] * SH(15) * -6.f * xy;
                        dz_ += kC3[1] * SH(10) * xy + kC3[2] * SH(11) * 8.f * yz + kC3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) +
                               kC3[4] * SH(13) * 8.f * xz + kC3[5] * SH(14) * (xx - yy);
                    }
                }
            }
#undef SH
#undef GSH
            gdir[0] += dx_ * gg; gdir[1] += dy_ * gg; gdir[2] += dz_ * gg;
        }
        if (D > 0) {
            const float dot = x * gdir[0] + y * gdir[1] + z * gdir[2];
            gm[0] += (gdir[0] - x * dot) * inv; gm[1] += (gdir[1] - y * dot) * inv; gm[2] += (gdir[2] - z * dot) * inv;
            dmeans3D[3 * i] = gm[0]; dmeans3D[3 * i + 1] = gm[1]; dmeans3D[3 * i + 2] = gm[2];
        }
    }
    sh16_regs_to_row<MODE>(sh_tile, lane, g);                          // (a lane touches only its own row: no barrier since the reads above)
    __syncthreads();
    if (!SINK || dsh_a) sh16_stage_out<MODE>(sh_tile, dsh_a, dsh_rest, i0, nvalid, lane);
    if (SINK && MODE == SH16_SPLIT) {
        if (sink.skip && *sink.skip) return;                          // overflowed frame: no step
        const int rows = sink.active_rows ? min(P, max(*sink.active_rows, 0)) : P;
        const int live = min(64, rows - i0);
        if (live <= 0) return;
        if (sink.leaf[EGS_SINK_SH_REST].p)
            sh16_adam_span(sh_tile, sink.leaf[EGS_SINK_SH_REST], (size_t)i0 * 45, live * 45, sink.coef[2 * EGS_SINK_SH_REST],
                           sink.coef[2 * EGS_SINK_SH_REST + 1], sink.b1, sink.b2, sink.eps, lane);
        if (sink.leaf[EGS_SINK_SH].p)
            sh16_adam_span(sh_tile + SH16_REST_WORDS, sink.leaf[EGS_SINK_SH], (size_t)i0 * 3, live * 3, sink.coef[2 * EGS_SINK_SH],
                           sink.coef[2 * EGS_SINK_SH + 1], sink.b1, sink.b2, sink.eps, lane);
        if (sink.leaf[EGS_SINK_MEANS3D].p && (int)lane < live) {       // three floats per lane (gm is zero for a culled row, as its gradient is)
            const EgsSinkLeaf& f = sink.leaf[EGS_SINK_MEANS3D];
            const float ss = sink.coef[2 * EGS_SINK_MEANS3D], ib = sink.coef[2 * EGS_SINK_MEANS3D + 1];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float P1 = f.p[3 * i + k], M1 = f.m[3 * i + k], V1 = f.v[3 * i + k];
                egs_adam1(P1, gm[k], M1, V1, sink.b1, sink.b2, sink.eps, ss, ib);
                f.p[3 * i + k] = P1; f.m[3 * i + k] = M1; f.v[3 * i + k] = V1;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_mark_visible(int P, const float* __restrict__ means3D,
                                                       const float* __restrict__ V, uint8_t* __restrict__ present) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float p[3] = { means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2] };
    float t[3]; xform43(p, V, t);
    present[i] = t[2] > 0.2f ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_zero_f4(float4* __restrict__ p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

}  // namespace

namespace {
__global__ void k_zero_u32(uint32_t* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
}  // namespace
hipError_t egs_launch_zero_u32(uint32_t* p, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_zero_u32, dim3(blocks), dim3(256), 0, s, p, n);
    return hipGetLastError();
}

hipError_t egs_launch_zero_f4(float4* p, size_t n4, hipStream_t s) {
    if (n4 == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_zero_f4, dim3(blocks), dim3(256), 0, s, p, n4);
    return hipGetLastError();
}

hipError_t egs_launch_preprocess(int P, int D, int M, const float* means3D, const float* shs, const float* colors,
                                 const float* opac, const float* scales, float mod, const float* rots, int act,
                                 const float* cov3D, EgsCamera cam, int32_t* radii, EgsGeomPtrs g, uint32_t* zero_words, size_t zero_n,
                                 const int32_t* active_count, const EgsImgPtrs* place, EgsObjRot rot, hipStream_t s) {
    if (P == 0) return hipSuccess;
    if (!zero_words) zero_n = 0;
    EgsPrologueArgs pa = {};
#define PP_ARGS P, D, M, means3D, shs, colors, opac, scales, mod, rots, cov3D, act, cam.view, cam.proj, cam.campos, cam.W, cam.H, cam.tanfovx, \
                cam.tanfovy, radii, g.rec, g.rect, g.offsets, g.clamped, g.visible, g.scan_scratch, g.block_hot, zero_words, zero_n, active_count
    if (place) {
        pa.n_tiles = ((cam.W + EGS_TILE - 1) / EGS_TILE) * ((cam.H + EGS_TILE - 1) / EGS_TILE);
        pa.quad_work = place->fwd_cost; pa.tile_order = place->fwd_order;
        hipLaunchKernelGGL(k_preprocess<true>, dim3((P + 255) / 256 + EGS_XCDS), dim3(256), 0, s, PP_ARGS, pa, rot);
    } else {
        hipLaunchKernelGGL(k_preprocess<false>, dim3((P + 255) / 256), dim3(256), 0, s, PP_ARGS, pa, rot);
    }
#undef PP_ARGS
    return hipGetLastError();
}

bool egs_can_fuse_count(int P, int W, int H, int cull) {
    if (P <= 0) return false;
    const EgsBinGeometry q = egs_bin_geometry(P, W, H, cull);
    // one round only: the looped instantiation holds the projection's inputs through the walk and spills (1M @ 1080p, four rounds of eight
    // groups: 116 + 151 us against 34 + 214 with the separate count pass) -- EGS_FUSE_MULTI_ROUND=1 lets it run all the same (measurements)
    static const bool multi = getenv("EGS_FUSE_MULTI_ROUND") != nullptr;
    if (!multi && bin_groups_per_block((unsigned)P, q.nblocks) > (unsigned)q.gpr) return false;
    return q.gpr >= 4 && q.gpr % 4 == 0 && q.lds >= sizeof(EgsOrderLds);
}
hipError_t egs_launch_preprocess_count(int P, int D, int M, const float* means3D, const float* shs, const float* colors,
                                       const float* opac, const float* scales, float mod, const float* rots, int act,
                                       const float* cov3D, EgsCamera cam, int32_t* radii, EgsGeomPtrs g, EgsBinPtrs b,
                                       const int32_t* active_count, const EgsImgPtrs* place, EgsObjRot rot, int cull, hipStream_t s) {
    const EgsBinGeometry q = egs_bin_geometry(P, cam.W, cam.H, cull);
    EgsPreArgs a = { P, D, M, means3D, shs, colors, opac, scales, mod, rots, cov3D, act, cam.view, cam.proj, cam.campos, cam.W, cam.H, cam.tanfovx, cam.tanfovy,
                     radii, g.rec, g.rect, g.offsets, g.clamped, g.visible, g.scan_scratch, g.block_hot, active_count };
    EgsCountArgs c = { q.gpr, q.gx, q.n_tiles, q.nblocks, q.cull, q.use_map, b.table, q.stride, b.chunk_sum };
    EgsPrologueArgs pa = {};
    const unsigned grid = ((q.nblocks + 7) / 8) * 8;
    const bool one = bin_groups_per_block((unsigned)P, q.nblocks) <= (unsigned)q.gpr;
    if (place) { pa.n_tiles = q.n_tiles; pa.quad_work = place->fwd_cost; pa.tile_order = place->fwd_order; }
#define PC_LAUNCH(PL, ONE) do {                                                                                                       \
        if (q.lds > 64 * 1024) {                                                                                                      \
            hipError_t e = hipFuncSetAttribute((const void*)k_preprocess_count<PL, ONE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q.lds); \
            if (e != hipSuccess) return e;                                                                                            \
        }                                                                                                                             \
        hipLaunchKernelGGL((k_preprocess_count<PL, ONE>), dim3(grid + (PL ? EGS_XCDS : 0)), dim3(EGS_BIN_THREADS), q.lds, s, a, c, pa, rot); } while (0)
    if (place) { if (one) PC_LAUNCH(true, true); else PC_LAUNCH(true, false); }
    else { if (one) PC_LAUNCH(false, true); else PC_LAUNCH(false, false); }
#undef PC_LAUNCH
    return hipGetLastError();
}

hipError_t egs_launch_preprocess_backward(int P, int D, int M, const float* means3D, const float* shs,
                                          const float* scales, float mod, const float* rots, const float* cov3D, int act,
                                          EgsCamera cam, const int32_t* radii, EgsGeomPtrs g, const float* grad_acc,
                                          int colors_given, float* dmeans2D, float* dcolors, float* dopac,
                                          float* dmeans3D, float* dcov3D, float* dsh, float* dscales, float* drots,
                                          float* stat_grad_accum, float* stat_denom, float* stat_max_radii, const uint32_t* skip_flag,
                                          const EgsSink* sink, EgsObjRot rot, hipStream_t s) {
    if (P == 0) return hipSuccess;
    EgsSink none = {};
#define PPB_ARGS P, D, M, means3D, colors_given ? nullptr : shs, scales, mod, rots, cov3D, act, cam.view, cam.proj, cam.campos, cam.W, \
                 cam.H, cam.tanfovx, cam.tanfovy, radii, g.clamped, g.rec, grad_acc, dmeans2D, dcolors, dopac, dmeans3D, \
                 dcov3D, colors_given ? nullptr : dsh, cov3D ? nullptr : dscales, cov3D ? nullptr : drots, \
                 stat_grad_accum, stat_denom, stat_max_radii, skip_flag
    if (sink) hipLaunchKernelGGL(k_preprocess_backward<true>, dim3((P + 255) / 256), dim3(256), 0, s, PPB_ARGS, *sink, rot);
    else hipLaunchKernelGGL(k_preprocess_backward<false>, dim3((P + 255) / 256), dim3(256), 0, s, PPB_ARGS, none, rot);
#undef PPB_ARGS
    return hipGetLastError();
}

bool egs_sh_backward_can_sink(int M, const float* sh_a, const float* sh_rest) {
    return M == 16 && sh_rest && ((((uintptr_t)sh_a) | ((uintptr_t)sh_rest)) & 15) == 0;
}
static bool sh16_fast(int M, const void* a, const void* b, const void* c, const void* d) {
    return M == 16 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) == 0;
}

hipError_t egs_launch_sh_forward(int P, int D, int M, const float* means3D, const float* sh_a, const float* sh_rest, EgsCamera cam,
                                 EgsGeomPtrs g, hipStream_t s) {
    if (P == 0) return hipSuccess;
    const dim3 grid((P + 63) / 64), block(64);
    if (sh16_fast(M, sh_a, sh_rest, nullptr, nullptr)) {
        if (sh_rest) hipLaunchKernelGGL(k_sh16_forward<SH16_SPLIT>, grid, block, (SH16_REST_WORDS + 192) * sizeof(float), s, P, D, means3D, cam.campos,
                                        sh_a, sh_rest, g.visible, g.rec, g.clamped);
        else hipLaunchKernelGGL(k_sh16_forward<SH16_CAT>, grid, block, 64 * 13 * sizeof(float4), s, P, D, means3D, cam.campos, sh_a, sh_rest,
                                g.visible, g.rec, g.clamped);
        return hipGetLastError();
    }
    const size_t lds = (size_t)64 * (3 * M + 1) * sizeof(float);
    hipLaunchKernelGGL(k_sh_forward, grid, block, lds, s, P, D, M, means3D, cam.campos, sh_a, sh_rest, g.visible, g.rec, g.clamped);
    return hipGetLastError();
}

hipError_t egs_launch_sh_backward(int P, int D, int M, const float* means3D, const float* sh_a, const float* sh_rest, EgsCamera cam,
                                  const int32_t* radii, EgsGeomPtrs g, const float* dcolors, float* dsh_a, float* dsh_rest,
                                  float* dmeans3D, const EgsSink* sink, hipStream_t s) {
    if (P == 0) return hipSuccess;
    const dim3 grid((P + 63) / 64), block(64);
    EgsSink none = {};
    if (sink) {                                                       // (the caller checked egs_sh_backward_can_sink)
        hipLaunchKernelGGL((k_sh16_backward<SH16_SPLIT, true>), grid, block, (SH16_REST_WORDS + 192) * sizeof(float), s, P, D, means3D, cam.campos,
                           sh_a, sh_rest, radii, g.clamped, dcolors, dsh_a, dsh_rest, dmeans3D, *sink);
        return hipGetLastError();
    }
    if (sh16_fast(M, sh_a, sh_rest, dsh_a, dsh_rest)) {
        if (sh_rest) hipLaunchKernelGGL((k_sh16_backward<SH16_SPLIT, false>), grid, block, (SH16_REST_WORDS + 192) * sizeof(float), s, P, D, means3D, cam.campos,
                                        sh_a, sh_rest, radii, g.clamped, dcolors, dsh_a, dsh_rest, dmeans3D, none);
        else hipLaunchKernelGGL((k_sh16_backward<SH16_CAT, false>), grid, block, 64 * 13 * sizeof(float4), s, P, D, means3D, cam.campos, sh_a, sh_rest,
                                radii, g.clamped, dcolors, dsh_a, dsh_rest, dmeans3D, none);
        return hipGetLastError();
    }
    const size_t lds = (size_t)64 * (3 * M + 1) * sizeof(float);
    hipLaunchKernelGGL(k_sh_backward, grid, block, lds, s, P, D, M, means3D, cam.campos, sh_a, sh_rest, radii, g.clamped,
                       dcolors, dsh_a, dsh_rest, dmeans3D);
    return hipGetLastError();
}

hipError_t egs_launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s) {
    if (P == 0) return hipSuccess;
    hipLaunchKernelGGL(k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
    return hipGetLastError();
}
