/*
 * oracle/raster_oracle.c -- CPU restatement of the tile-based differentiable 3D Gaussian
 * rasterizer that EgoGaussian's render path calls.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path (egogaussian_amd/) never does.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in an un-vendored third-party submodule,
 * ashawkey/diff-gaussian-rasterization @ 8829d14f814fccdaf840b7b0f3021a616583c0a1 (pinned only
 * in prose at /root/reference/README.md:26, URL in /root/reference/.gitmodules:1-3).  Its source
 * is absent from /root/reference and the reference ships no tests or golden vectors for it.
 * This file restates the published algorithm (Kerbl et al. 2023, "3D Gaussian Splatting",
 * sections 4-6 and appendix; depth/alpha channels as in the fork named above) and is anchored on
 * the reference's own call sites:
 *     /root/reference/gaussian_renderer/__init__.py:38-53,90-98   (training call, SH + cov3D_precomp)
 *     /root/reference/gaussian_renderer/render_helper.py:15-28,61-63 (label call, colours + scale/rot)
 * and on the reference's importable Python for the sub-steps it does contain:
 *     cov3D from scale/quaternion  /root/reference/utils/general_utils.py:110-156
 *     SH basis and +0.5 / clamp    /root/reference/utils/sh_utils.py:57-112,
 *                                  /root/reference/gaussian_renderer/__init__.py:83
 *     matrix conventions           /root/reference/scene/cameras.py:67-70 (row-vector, transposed)
 * Gradients are pinned by torch.autograd through oracle/raster_torch.py (tests/test_oracle_*.py).
 *
 * Parity-critical constants: near cull view.z <= 0.2; 1/(w + 1e-7);
 * frustum clamp 1.3*tanfov; +0.3 px^2 low-pass; radius = ceil(3*sqrt(lambda_max)) with
 * sqrt(max(0.1, mid^2 - det)); 16x16 tiles; key = tile<<32 | float_bits(depth); stable sort;
 * alpha = min(0.99, o*exp(power)); skip alpha < 1/255; stop when T*(1-alpha) < 1e-4.
 *
 * Build: see oracle/Makefile.  -DEGSO_REAL=double gives the fp64 variant used to check the
 * analytic backward against autograd at tight tolerance.  All floating-point expressions are
 * written one operation per statement group so that, compiled with -ffp-contract=off, the fp32
 * variant is operation-for-operation identical to the HIP preprocess kernel (bit-exact radii,
 * tile rectangles, depth bits -> bit-exact sort keys).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef EGSO_REAL
#define EGSO_REAL float
#endif
typedef EGSO_REAL real;

/* The published backward divides by (det^2 + 1e-7); the fp64 verification build sets this to 0 so
 * the analytic gradient can be compared with autograd (which divides by det^2) at 1e-9. */
#ifndef EGSO_DENOM_EPS
#define EGSO_DENOM_EPS 0.0000001
#endif
#define TILE_X 16
#define TILE_Y 16
#define R_(x) ((real)(x))

static const real SH_C0 = R_(0.28209479177387814);
static const real SH_C1 = R_(0.4886025119029199);
static const real SH_C2[5] = { R_(1.0925484305920792), R_(-1.0925484305920792), R_(0.31539156525252005),
                               R_(-1.0925484305920792), R_(0.5462742152960396) };
static const real SH_C3[7] = { R_(-0.5900435899266435), R_(2.890611442640554), R_(-0.4570457994644658),
                               R_(0.3731763325901154), R_(-0.4570457994644658), R_(1.445305721320277),
                               R_(-0.5900435899266435) };

static real r_sqrt(real x) { return sizeof(real) == 4 ? (real)sqrtf((float)x) : (real)sqrt((double)x); }
static real r_exp(real x)  { return sizeof(real) == 4 ? (real)expf((float)x)  : (real)exp((double)x); }
static real r_ceil(real x) { return sizeof(real) == 4 ? (real)ceilf((float)x) : (real)ceil((double)x); }
static real r_max(real a, real b) { return a > b ? a : b; }
static real r_min(real a, real b) { return a < b ? a : b; }
static int  i_max(int a, int b) { return a > b ? a : b; }
static int  i_min(int a, int b) { return a < b ? a : b; }

int egso_real_bytes(void) { return (int)sizeof(real); }

/* Row-vector convention: the reference stores matrices transposed (scene/cameras.py:67-69), so a
 * point is transformed as out_j = sum_i p_i * m[4*i + j] + m[12 + j]. */
static void xform43(const real* p, const real* m, real* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void xform44(const real* p, const real* m, real* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* Tile rectangle touched by a splat of integer radius r centred at pixel-space p. */
static void tile_rect(real px, real py, int r, int gx, int gy, int* rect) {
    rect[0] = i_min(gx, i_max(0, (int)((px - (real)r) / (real)TILE_X)));
    rect[1] = i_min(gy, i_max(0, (int)((py - (real)r) / (real)TILE_Y)));
    rect[2] = i_min(gx, i_max(0, (int)((px + (real)r + (real)(TILE_X - 1)) / (real)TILE_X)));
    rect[3] = i_min(gy, i_max(0, (int)((py + (real)r + (real)(TILE_Y - 1)) / (real)TILE_Y)));
}

/* cov3D = (R S)(R S)^T from an (un-normalised) quaternion (w,x,y,z) and scales; six unique
 * entries in the order (00,01,02,11,12,22) -- utils/general_utils.py:110-156. */
static void cov3d_from_scale_rot(const real* s, real mod, const real* q, real* c6) {
    real r = q[0], x = q[1], y = q[2], z = q[3];
    real R[9] = { R_(1) - R_(2) * (y * y + z * z), R_(2) * (x * y - r * z), R_(2) * (x * z + r * y),
                  R_(2) * (x * y + r * z), R_(1) - R_(2) * (x * x + z * z), R_(2) * (y * z - r * x),
                  R_(2) * (x * z - r * y), R_(2) * (y * z + r * x), R_(1) - R_(2) * (x * x + y * y) };
    real sc[3] = { mod * s[0], mod * s[1], mod * s[2] };
    real L[9];
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) L[3 * i + k] = R[3 * i + k] * sc[k];
    int n = 0;
    for (int i = 0; i < 3; i++) for (int j = i; j < 3; j++)
        c6[n++] = L[3 * i] * L[3 * j] + L[3 * i + 1] * L[3 * j + 1] + L[3 * i + 2] * L[3 * j + 2];
}

/* SH -> RGB, degrees 0..3; sh is [M][3]; dir is the unit view direction (utils/sh_utils.py:57-112). */
static void sh_to_rgb(int deg, const real* sh, const real* d, real* rgb) {
    real x = d[0], y = d[1], z = d[2];
    for (int c = 0; c < 3; c++) {
#define SH(k) sh[(k) * 3 + c]
        real v = SH_C0 * SH(0);
        if (deg > 0) {
            v = v - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                v = v + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) + SH_C2[2] * (R_(2) * zz - xx - yy) * SH(6) +
                    SH_C2[3] * xz * SH(7) + SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    v = v + SH_C3[0] * y * (R_(3) * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                        SH_C3[2] * y * (R_(4) * zz - xx - yy) * SH(11) +
                        SH_C3[3] * z * (R_(2) * zz - R_(3) * xx - R_(3) * yy) * SH(12) +
                        SH_C3[4] * x * (R_(4) * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                        SH_C3[6] * x * (xx - R_(3) * yy) * SH(15);
                }
            }
        }
#undef SH
        rgb[c] = v + R_(0.5);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Stage 1: per-Gaussian preprocess.  Outputs are zero for culled Gaussians (radii == 0).       */
/* Any of scales/rotations (with cov3D_precomp), shs (with colors_precomp) may be NULL.         */
int egso_preprocess(int P, int D, int M, const real* means3D, const real* scales, real scale_modifier,
                    const real* rotations, const real* opacities, const real* shs, const real* colors_precomp,
                    const real* cov3D_precomp, const real* viewmatrix, const real* projmatrix, const real* campos,
                    int W, int H, real tanfovx, real tanfovy,
                    /* out */ int32_t* radii, real* xy, real* depths, real* cov3D, real* rgb, real* conic_opacity,
                    uint32_t* tiles_touched, uint8_t* clamped, int32_t* rects) {
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
    const real focal_x = (real)W / (R_(2) * tanfovx), focal_y = (real)H / (R_(2) * tanfovy);
    for (int i = 0; i < P; i++) {
        radii[i] = 0; tiles_touched[i] = 0;
        xy[2 * i] = xy[2 * i + 1] = 0; depths[i] = 0;
        for (int k = 0; k < 6; k++) cov3D[6 * i + k] = 0;
        for (int k = 0; k < 3; k++) { rgb[3 * i + k] = 0; clamped[3 * i + k] = 0; }
        for (int k = 0; k < 4; k++) { conic_opacity[4 * i + k] = 0; rects[4 * i + k] = 0; }

        const real* p = means3D + 3 * i;
        real t[3]; xform43(p, viewmatrix, t);
        if (t[2] <= R_(0.2)) continue;                       /* near-plane cull */
        real hom[4]; xform44(p, projmatrix, hom);
        real pw = R_(1) / (hom[3] + R_(0.0000001));
        real ndc_x = hom[0] * pw, ndc_y = hom[1] * pw;

        real c6[6];
        if (cov3D_precomp) memcpy(c6, cov3D_precomp + 6 * i, sizeof(c6));
        else cov3d_from_scale_rot(scales + 3 * i, scale_modifier, rotations + 4 * i, c6);

        /* EWA projection: cov2D = (J R) Sigma (J R)^T, J the perspective Jacobian at the clamped
         * camera-space point, R the rotation part of the world->view matrix. */
        real limx = R_(1.3) * tanfovx, limy = R_(1.3) * tanfovy;
        real txtz = t[0] / t[2], tytz = t[1] / t[2];
        real tx = r_min(limx, r_max(-limx, txtz)) * t[2];
        real ty = r_min(limy, r_max(-limy, tytz)) * t[2];
        real j00 = focal_x / t[2], j02 = -(focal_x * tx) / (t[2] * t[2]);
        real j11 = focal_y / t[2], j12 = -(focal_y * ty) / (t[2] * t[2]);
        /* R_jk = viewmatrix[4*k + j] */
        real m0[3], m1[3];
        for (int k = 0; k < 3; k++) {
            m0[k] = j00 * viewmatrix[4 * k + 0] + j02 * viewmatrix[4 * k + 2];
            m1[k] = j11 * viewmatrix[4 * k + 1] + j12 * viewmatrix[4 * k + 2];
        }
        real S0[3] = { c6[0] * m0[0] + c6[1] * m0[1] + c6[2] * m0[2], c6[1] * m0[0] + c6[3] * m0[1] + c6[4] * m0[2],
                       c6[2] * m0[0] + c6[4] * m0[1] + c6[5] * m0[2] };
        real S1[3] = { c6[0] * m1[0] + c6[1] * m1[1] + c6[2] * m1[2], c6[1] * m1[0] + c6[3] * m1[1] + c6[4] * m1[2],
                       c6[2] * m1[0] + c6[4] * m1[1] + c6[5] * m1[2] };
        real a = m0[0] * S0[0] + m0[1] * S0[1] + m0[2] * S0[2] + R_(0.3);
        real b = m0[0] * S1[0] + m0[1] * S1[1] + m0[2] * S1[2];
        real c = m1[0] * S1[0] + m1[1] * S1[1] + m1[2] * S1[2] + R_(0.3);

        real det = a * c - b * b;
        if (det == R_(0)) continue;
        real det_inv = R_(1) / det;
        real conA = c * det_inv, conB = -b * det_inv, conC = a * det_inv;
        real mid = R_(0.5) * (a + c);
        real disc = r_sqrt(r_max(R_(0.1), mid * mid - det));
        real lam1 = mid + disc, lam2 = mid - disc;
        int rad = (int)r_ceil(R_(3) * r_sqrt(r_max(lam1, lam2)));
        real px = ((ndc_x + R_(1)) * (real)W - R_(1)) * R_(0.5);
        real py = ((ndc_y + R_(1)) * (real)H - R_(1)) * R_(0.5);
        int rect[4]; tile_rect(px, py, rad, gx, gy, rect);
        if ((rect[2] - rect[0]) * (rect[3] - rect[1]) == 0) continue;

        if (colors_precomp) {
            for (int k = 0; k < 3; k++) rgb[3 * i + k] = colors_precomp[3 * i + k];
        } else {
            real dir[3] = { p[0] - campos[0], p[1] - campos[1], p[2] - campos[2] };
            real inv = R_(1) / r_sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
            dir[0] *= inv; dir[1] *= inv; dir[2] *= inv;
            real col[3]; sh_to_rgb(D, shs + (size_t)i * M * 3, dir, col);
            for (int k = 0; k < 3; k++) { clamped[3 * i + k] = col[k] < 0; rgb[3 * i + k] = r_max(col[k], R_(0)); }
        }
        depths[i] = t[2];
        radii[i] = rad;
        xy[2 * i] = px; xy[2 * i + 1] = py;
        for (int k = 0; k < 6; k++) cov3D[6 * i + k] = c6[k];
        conic_opacity[4 * i + 0] = conA; conic_opacity[4 * i + 1] = conB; conic_opacity[4 * i + 2] = conC;
        conic_opacity[4 * i + 3] = opacities[i];
        tiles_touched[i] = (uint32_t)((rect[2] - rect[0]) * (rect[3] - rect[1]));
        for (int k = 0; k < 4; k++) rects[4 * i + k] = rect[k];
    }
    return 0;
}

/* Stage 2: inclusive scan of tiles_touched.  Returns R = number of (Gaussian, tile) instances. */
int64_t egso_scan(int P, const uint32_t* tiles_touched, uint32_t* offsets) {
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += tiles_touched[i]; offsets[i] = acc; }
    return (int64_t)acc;
}

/* Stage 3: one (key, value) per touched tile, tile rows outer / columns inner;
 * key = tile_id << 32 | bits(float depth), value = Gaussian index. */
int egso_duplicate(int P, const real* depths, const uint32_t* offsets, const int32_t* radii, const int32_t* rects,
                   int W, uint64_t* keys, uint32_t* vals) {
    const int gx = (W + TILE_X - 1) / TILE_X;
    for (int i = 0; i < P; i++) {
        if (radii[i] <= 0) continue;
        uint32_t off = i == 0 ? 0 : offsets[i - 1];
        float df = (float)depths[i]; uint32_t dbits; memcpy(&dbits, &df, 4);
        for (int y = rects[4 * i + 1]; y < rects[4 * i + 3]; y++)
            for (int x = rects[4 * i + 0]; x < rects[4 * i + 2]; x++) {
                keys[off] = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dbits;
                vals[off] = (uint32_t)i; off++;
            }
    }
    return 0;
}

/* Number of low key bits that need sorting: 32 depth bits + bits to hold any tile id < n_tiles. */
int egso_key_bits(int n_tiles) { int b = 0; while ((n_tiles >> b) != 0) b++; return 32 + b; }

/* Stage 4: stable LSD radix sort (8-bit digits) of the pairs on the low `bits` bits. */
int egso_sort_pairs(int64_t R, const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out,
                    uint32_t* vals_out, int bits) {
    if (R == 0) return 0;
    uint64_t* ka = (uint64_t*)malloc(sizeof(uint64_t) * R); uint64_t* kb = (uint64_t*)malloc(sizeof(uint64_t) * R);
    uint32_t* va = (uint32_t*)malloc(sizeof(uint32_t) * R); uint32_t* vb = (uint32_t*)malloc(sizeof(uint32_t) * R);
    if (!ka || !kb || !va || !vb) { free(ka); free(kb); free(va); free(vb); return 1; }
    memcpy(ka, keys_in, sizeof(uint64_t) * R); memcpy(va, vals_in, sizeof(uint32_t) * R);
    for (int shift = 0; shift < bits; shift += 8) {
        int64_t cnt[257]; memset(cnt, 0, sizeof(cnt));
        for (int64_t i = 0; i < R; i++) cnt[((ka[i] >> shift) & 0xff) + 1]++;
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (int64_t i = 0; i < R; i++) { int64_t o = cnt[(ka[i] >> shift) & 0xff]++; kb[o] = ka[i]; vb[o] = va[i]; }
        uint64_t* tk = ka; ka = kb; kb = tk; uint32_t* tv = va; va = vb; vb = tv;
    }
    memcpy(keys_out, ka, sizeof(uint64_t) * R); memcpy(vals_out, va, sizeof(uint32_t) * R);
    free(ka); free(kb); free(va); free(vb);
    return 0;
}

/* Stage 5: ranges[tile] = [first, last+1) in the sorted list; empty tiles stay (0,0). */
int egso_tile_ranges(int64_t R, const uint64_t* keys_sorted, int n_tiles, uint32_t* ranges) {
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)n_tiles);
    for (int64_t i = 0; i < R; i++) {
        uint32_t t = (uint32_t)(keys_sorted[i] >> 32);
        if (i == 0 || t != (uint32_t)(keys_sorted[i - 1] >> 32)) ranges[2 * t] = (uint32_t)i;
        if (i == R - 1 || t != (uint32_t)(keys_sorted[i + 1] >> 32)) ranges[2 * t + 1] = (uint32_t)(i + 1);
    }
    return 0;
}

/* Stage 6: per-pixel front-to-back compositing of colour, depth and alpha. */
int egso_render_forward(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const real* xy,
                        const real* rgb, const real* depths, const real* conic_opacity, const real* bg,
                        /* out */ real* out_color, real* out_depth, real* out_alpha, real* final_T,
                        uint32_t* n_contrib, int nthreads) {
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx0 = (tile % gx) * TILE_X, ty0 = (tile / gx) * TILE_Y;
        const uint32_t beg = ranges[2 * tile], end = ranges[2 * tile + 1];
        for (int ly = 0; ly < TILE_Y; ly++) for (int lx = 0; lx < TILE_X; lx++) {
            const int px = tx0 + lx, py = ty0 + ly;
            if (px >= W || py >= H) continue;
            const size_t pix = (size_t)py * W + px;
            real T = R_(1), C[3] = { 0, 0, 0 }, Dacc = 0, Aacc = 0;
            uint32_t contributor = 0, last = 0;
            for (uint32_t k = beg; k < end; k++) {
                contributor++;
                const uint32_t g = point_list[k];
                real dx = xy[2 * g] - (real)px, dy = xy[2 * g + 1] - (real)py;
                const real* co = conic_opacity + 4 * g;
                real power = R_(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > R_(0)) continue;
                real alpha = r_min(R_(0.99), co[3] * r_exp(power));
                if (alpha < R_(1) / R_(255)) continue;
                real test_T = T * (R_(1) - alpha);
                if (test_T < R_(0.0001)) break;
                real w = alpha * T;
                C[0] += rgb[3 * g] * w; C[1] += rgb[3 * g + 1] * w; C[2] += rgb[3 * g + 2] * w;
                Dacc += depths[g] * w; Aacc += w;
                T = test_T; last = contributor;
            }
            final_T[pix] = T; n_contrib[pix] = last;
            for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pix] = C[ch] + T * bg[ch];
            out_depth[pix] = Dacc; out_alpha[pix] = Aacc;
        }
    }
    return 0;
}

static void atomic_add(real* p, real v) {
#pragma omp atomic
    *p += v;
}

/* Backward stage 1: back-to-front replay.  dL_dmean2D is scaled to NDC units (x 0.5*W, 0.5*H),
 * which is what the reference's densification statistic consumes
 * (/root/reference/scene/gaussian_model.py:735-740).  dL_dconic is [P][4] with (xx, xy, -, yy);
 * the xy slot holds half the true derivative (the symmetric entry is counted once). */
int egso_render_backward(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const real* xy,
                         const real* rgb, const real* depths, const real* conic_opacity, const real* bg,
                         const real* final_T, const uint32_t* n_contrib, const real* dL_dpix, const real* dL_dpix_depth,
                         const real* dL_dpix_alpha,
                         /* out, pre-zeroed */ real* dL_dmean2D /*[P][3]*/, real* dL_dconic /*[P][4]*/,
                         real* dL_dopacity, real* dL_dcolor, real* dL_ddepth, int nthreads) {
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
    const real ddelx_dx = R_(0.5) * (real)W, ddely_dy = R_(0.5) * (real)H;
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx0 = (tile % gx) * TILE_X, ty0 = (tile / gx) * TILE_Y;
        const uint32_t beg = ranges[2 * tile];
        for (int ly = 0; ly < TILE_Y; ly++) for (int lx = 0; lx < TILE_X; lx++) {
            const int px = tx0 + lx, py = ty0 + ly;
            if (px >= W || py >= H) continue;
            const size_t pix = (size_t)py * W + px;
            const real T_final = final_T[pix];
            real T = T_final;
            const real gpix[3] = { dL_dpix[pix], dL_dpix[(size_t)H * W + pix], dL_dpix[(size_t)2 * H * W + pix] };
            const real gdep = dL_dpix_depth[pix], galp = dL_dpix_alpha[pix];
            const real bg_dot = bg[0] * gpix[0] + bg[1] * gpix[1] + bg[2] * gpix[2];
            real acc_c[3] = { 0, 0, 0 }, last_c[3] = { 0, 0, 0 }, acc_d = 0, last_d = 0, acc_a = 0, last_alpha = 0;
            for (int64_t k = (int64_t)n_contrib[pix] - 1; k >= 0; k--) {
                const uint32_t g = point_list[beg + k];
                real dx = xy[2 * g] - (real)px, dy = xy[2 * g + 1] - (real)py;
                const real* co = conic_opacity + 4 * g;
                real power = R_(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > R_(0)) continue;
                real G = r_exp(power);
                real alpha = r_min(R_(0.99), co[3] * G);
                if (alpha < R_(1) / R_(255)) continue;
                T = T / (R_(1) - alpha);
                const real w = alpha * T;
                real dL_dalpha = 0;
                for (int ch = 0; ch < 3; ch++) {
                    const real c = rgb[3 * g + ch];
                    acc_c[ch] = last_alpha * last_c[ch] + (R_(1) - last_alpha) * acc_c[ch];
                    last_c[ch] = c;
                    dL_dalpha += (c - acc_c[ch]) * gpix[ch];
                    atomic_add(&dL_dcolor[3 * g + ch], w * gpix[ch]);
                }
                const real cd = depths[g];
                acc_d = last_alpha * last_d + (R_(1) - last_alpha) * acc_d; last_d = cd;
                dL_dalpha += (cd - acc_d) * gdep;
                atomic_add(&dL_ddepth[g], w * gdep);
                acc_a = last_alpha + (R_(1) - last_alpha) * acc_a;
                dL_dalpha += (R_(1) - acc_a) * galp;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (R_(1) - alpha)) * bg_dot;
                const real dL_dG = co[3] * dL_dalpha;
                const real gdx = G * dx, gdy = G * dy;
                const real dG_ddelx = -gdx * co[0] - gdy * co[1];
                const real dG_ddely = -gdy * co[2] - gdx * co[1];
                atomic_add(&dL_dmean2D[3 * g + 0], dL_dG * dG_ddelx * ddelx_dx);
                atomic_add(&dL_dmean2D[3 * g + 1], dL_dG * dG_ddely * ddely_dy);
                atomic_add(&dL_dconic[4 * g + 0], R_(-0.5) * gdx * dx * dL_dG);
                atomic_add(&dL_dconic[4 * g + 1], R_(-0.5) * gdx * dy * dL_dG);
                atomic_add(&dL_dconic[4 * g + 3], R_(-0.5) * gdy * dy * dL_dG);
                atomic_add(&dL_dopacity[g], G * dL_dalpha);
            }
        }
    }
    return 0;
}

/* Backward stage 2: conic -> cov2D -> cov3D and mean (through the Jacobian), projective divide,
 * depth, SH, and optionally cov3D -> (scale, quaternion). */
int egso_preprocess_backward(int P, int D, int M, const real* means3D, const int32_t* radii, const real* shs,
                             const uint8_t* clamped, const real* scales, const real* rotations, real scale_modifier,
                             const real* cov3D /* as used in forward */, const real* viewmatrix, const real* projmatrix,
                             const real* campos, int W, int H, real tanfovx, real tanfovy,
                             const real* dL_dmean2D, const real* dL_dconic, const real* dL_dcolor_in,
                             const real* dL_ddepth,
                             /* out */ real* dL_dmeans3D, real* dL_dcov3D, real* dL_dsh, real* dL_dscale, real* dL_drot) {
    const real focal_x = (real)W / (R_(2) * tanfovx), focal_y = (real)H / (R_(2) * tanfovy);
    for (int i = 0; i < P; i++) {
        for (int k = 0; k < 3; k++) dL_dmeans3D[3 * i + k] = 0;
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * i + k] = 0;
        if (dL_dsh) for (int k = 0; k < M * 3; k++) dL_dsh[(size_t)i * M * 3 + k] = 0;
        if (dL_dscale) for (int k = 0; k < 3; k++) dL_dscale[3 * i + k] = 0;
        if (dL_drot) for (int k = 0; k < 4; k++) dL_drot[4 * i + k] = 0;
        if (!(radii[i] > 0)) continue;
        const real* p = means3D + 3 * i;
        const real* c6 = cov3D + 6 * i;
        real gmean[3] = { 0, 0, 0 };

        /* --- conic -> cov2D (a,b,c) --- */
        real t[3]; xform43(p, viewmatrix, t);
        real limx = R_(1.3) * tanfovx, limy = R_(1.3) * tanfovy;
        real txtz = t[0] / t[2], tytz = t[1] / t[2];
        real tx = r_min(limx, r_max(-limx, txtz)) * t[2];
        real ty = r_min(limy, r_max(-limy, tytz)) * t[2];
        real xmask = (txtz < -limx || txtz > limx) ? R_(0) : R_(1);
        real ymask = (tytz < -limy || tytz > limy) ? R_(0) : R_(1);
        real j00 = focal_x / t[2], j02 = -(focal_x * tx) / (t[2] * t[2]);
        real j11 = focal_y / t[2], j12 = -(focal_y * ty) / (t[2] * t[2]);
        real m0[3], m1[3];
        for (int k = 0; k < 3; k++) {
            m0[k] = j00 * viewmatrix[4 * k + 0] + j02 * viewmatrix[4 * k + 2];
            m1[k] = j11 * viewmatrix[4 * k + 1] + j12 * viewmatrix[4 * k + 2];
        }
        real S0[3] = { c6[0] * m0[0] + c6[1] * m0[1] + c6[2] * m0[2], c6[1] * m0[0] + c6[3] * m0[1] + c6[4] * m0[2],
                       c6[2] * m0[0] + c6[4] * m0[1] + c6[5] * m0[2] };
        real S1[3] = { c6[0] * m1[0] + c6[1] * m1[1] + c6[2] * m1[2], c6[1] * m1[0] + c6[3] * m1[1] + c6[4] * m1[2],
                       c6[2] * m1[0] + c6[4] * m1[1] + c6[5] * m1[2] };
        real a = m0[0] * S0[0] + m0[1] * S0[1] + m0[2] * S0[2] + R_(0.3);
        real b = m0[0] * S1[0] + m0[1] * S1[1] + m0[2] * S1[2];
        real c = m1[0] * S1[0] + m1[1] * S1[1] + m1[2] * S1[2] + R_(0.3);
        real gA = dL_dconic[4 * i + 0], gB = dL_dconic[4 * i + 1], gC = dL_dconic[4 * i + 3];
        real denom = a * c - b * b;
        real d2inv = R_(1) / (denom * denom + R_(EGSO_DENOM_EPS));
        real dL_da = 0, dL_db = 0, dL_dc = 0;
        if (d2inv != R_(0)) {
            dL_da = d2inv * (-c * c * gA + R_(2) * b * c * gB + (denom - a * c) * gC);
            dL_dc = d2inv * (-a * a * gC + R_(2) * a * b * gB + (denom - a * c) * gA);
            dL_db = d2inv * R_(2) * (b * c * gA - (denom + R_(2) * b * b) * gB + a * b * gC);
            /* cov2D -> stored cov3D entries (off-diagonals appear twice in the symmetric matrix) */
            dL_dcov3D[6 * i + 0] = m0[0] * m0[0] * dL_da + m0[0] * m1[0] * dL_db + m1[0] * m1[0] * dL_dc;
            dL_dcov3D[6 * i + 3] = m0[1] * m0[1] * dL_da + m0[1] * m1[1] * dL_db + m1[1] * m1[1] * dL_dc;
            dL_dcov3D[6 * i + 5] = m0[2] * m0[2] * dL_da + m0[2] * m1[2] * dL_db + m1[2] * m1[2] * dL_dc;
            dL_dcov3D[6 * i + 1] = R_(2) * m0[0] * m0[1] * dL_da + (m0[0] * m1[1] + m0[1] * m1[0]) * dL_db + R_(2) * m1[0] * m1[1] * dL_dc;
            dL_dcov3D[6 * i + 2] = R_(2) * m0[0] * m0[2] * dL_da + (m0[0] * m1[2] + m0[2] * m1[0]) * dL_db + R_(2) * m1[0] * m1[2] * dL_dc;
            dL_dcov3D[6 * i + 4] = R_(2) * m0[2] * m0[1] * dL_da + (m0[1] * m1[2] + m0[2] * m1[1]) * dL_db + R_(2) * m1[1] * m1[2] * dL_dc;
        }
        /* cov2D -> (J R) rows -> J -> camera-space point -> mean */
        real gm0[3], gm1[3];
        for (int k = 0; k < 3; k++) {
            gm0[k] = R_(2) * S0[k] * dL_da + S1[k] * dL_db;
            gm1[k] = R_(2) * S1[k] * dL_dc + S0[k] * dL_db;
        }
        real gJ00 = 0, gJ02 = 0, gJ11 = 0, gJ12 = 0;
        for (int k = 0; k < 3; k++) {
            gJ00 += viewmatrix[4 * k + 0] * gm0[k]; gJ02 += viewmatrix[4 * k + 2] * gm0[k];
            gJ11 += viewmatrix[4 * k + 1] * gm1[k]; gJ12 += viewmatrix[4 * k + 2] * gm1[k];
        }
        real tz = R_(1) / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        real gtx = xmask * -focal_x * tz2 * gJ02;
        real gty = ymask * -focal_y * tz2 * gJ12;
        real gtz = -focal_x * tz2 * gJ00 - focal_y * tz2 * gJ11 + (R_(2) * focal_x * tx) * tz3 * gJ02 +
                   (R_(2) * focal_y * ty) * tz3 * gJ12;
        for (int k = 0; k < 3; k++)
            gmean[k] += viewmatrix[4 * k + 0] * gtx + viewmatrix[4 * k + 1] * gty + viewmatrix[4 * k + 2] * gtz;

        /* --- screen-space mean (NDC-scaled) -> mean through the projective divide --- */
        real hom[4]; xform44(p, projmatrix, hom);
        real mw = R_(1) / (hom[3] + R_(0.0000001));
        real mul1 = hom[0] * mw * mw, mul2 = hom[1] * mw * mw;
        real g2x = dL_dmean2D[3 * i + 0], g2y = dL_dmean2D[3 * i + 1];
        for (int k = 0; k < 3; k++)
            gmean[k] += (projmatrix[4 * k + 0] * mw - projmatrix[4 * k + 3] * mul1) * g2x +
                        (projmatrix[4 * k + 1] * mw - projmatrix[4 * k + 3] * mul2) * g2y;

        /* --- depth = view.z (perspective row of the view matrix is (0,0,0,1) in practice) --- */
        real mul3 = viewmatrix[2] * p[0] + viewmatrix[6] * p[1] + viewmatrix[10] * p[2] + viewmatrix[14];
        for (int k = 0; k < 3; k++)
            gmean[k] += (viewmatrix[4 * k + 2] - viewmatrix[4 * k + 3] * mul3) * dL_ddepth[i];

        /* --- colour -> SH coefficients and view direction --- */
        if (shs) {
            real dir0[3] = { p[0] - campos[0], p[1] - campos[1], p[2] - campos[2] };
            real len2 = dir0[0] * dir0[0] + dir0[1] * dir0[1] + dir0[2] * dir0[2];
            real inv = R_(1) / r_sqrt(len2);
            real x = dir0[0] * inv, y = dir0[1] * inv, z = dir0[2] * inv;
            const real* sh = shs + (size_t)i * M * 3;
            real* gsh = dL_dsh + (size_t)i * M * 3;
            real gdir[3] = { 0, 0, 0 };
            for (int ch = 0; ch < 3; ch++) {
                real g = clamped[3 * i + ch] ? R_(0) : dL_dcolor_in[3 * i + ch];
#define SH(k) sh[(k) * 3 + ch]
#define GSH(k) gsh[(k) * 3 + ch]
                real dx_ = 0, dy_ = 0, dz_ = 0;   /* d colour / d (x,y,z) */
                GSH(0) = SH_C0 * g;
                if (D > 0) {
                    GSH(1) = -SH_C1 * y * g; GSH(2) = SH_C1 * z * g; GSH(3) = -SH_C1 * x * g;
                    dx_ = -SH_C1 * SH(3); dy_ = -SH_C1 * SH(1); dz_ = SH_C1 * SH(2);
                    if (D > 1) {
                        real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        GSH(4) = SH_C2[0] * xy * g; GSH(5) = SH_C2[1] * yz * g; GSH(6) = SH_C2[2] * (R_(2) * zz - xx - yy) * g;
                        GSH(7) = SH_C2[3] * xz * g; GSH(8) = SH_C2[4] * (xx - yy) * g;
                        dx_ += SH_C2[0] * y * SH(4) + SH_C2[2] * R_(2) * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * R_(2) * x * SH(8);
                        dy_ += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * R_(2) * -y * SH(6) + SH_C2[4] * R_(2) * -y * SH(8);
                        dz_ += SH_C2[1] * y * SH(5) + SH_C2[2] * R_(2) * R_(2) * z * SH(6) + SH_C2[3] * x * SH(7);
                        if (D > 2) {
                            GSH(9) = SH_C3[0] * y * (R_(3) * xx - yy) * g; GSH(10) = SH_C3[1] * xy * z * g;
                            GSH(11) = SH_C3[2] * y * (R_(4) * zz - xx - yy) * g;
                            GSH(12) = SH_C3[3] * z * (R_(2) * zz - R_(3) * xx - R_(3) * yy) * g;
                            GSH(13) = SH_C3[4] * x * (R_(4) * zz - xx - yy) * g; GSH(14) = SH_C3[5] * z * (xx - yy) * g;
                            GSH(15) = SH_C3[6] * x * (xx - R_(3) * yy) * g;
                            dx_ += SH_C3[0] * SH(9) * R_(3) * R_(2) * xy + SH_C3[1] * SH(10) * yz + SH_C3[2] * SH(11) * -R_(2) * xy +
                                   SH_C3[3] * SH(12) * -R_(3) * R_(2) * xz + SH_C3[4] * SH(13) * (-R_(3) * xx + R_(4) * zz - yy) +
                                   SH_C3[5] * SH(14) * R_(2) * xz + SH_C3[6] * SH(15) * R_(3) * (xx - yy);
                            dy_ += SH_C3[0] * SH(9) * R_(3) * (xx - yy) + SH_C3[1] * SH(10) * xz +
                                   SH_C3[2] * SH(11) * (-R_(3) * yy + R_(4) * zz - xx) + SH_C3[3] * SH(12) * -R_(3) * R_(2) * yz +
                                   SH_C3[4] * SH(13) * -R_(2) * xy + SH_C3[5] * SH(14) * -R_(2) * yz + SH_C3[6] * SH(15) * -R_(3) * R_(2) * xy;
                            dz_ += SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * R_(4) * R_(2) * yz +
                                   SH_C3[3] * SH(12) * R_(3) * (R_(2) * zz - xx - yy) + SH_C3[4] * SH(13) * R_(4) * R_(2) * xz +
                                   SH_C3[5] * SH(14) * (xx - yy);
                        }
                    }
                }
#undef SH
#undef GSH
                gdir[0] += dx_ * g; gdir[1] += dy_ * g; gdir[2] += dz_ * g;
            }
            /* back through dir = dir0 / |dir0| */
            real dot = x * gdir[0] + y * gdir[1] + z * gdir[2];
            gmean[0] += (gdir[0] - x * dot) * inv; gmean[1] += (gdir[1] - y * dot) * inv; gmean[2] += (gdir[2] - z * dot) * inv;
        }
        for (int k = 0; k < 3; k++) dL_dmeans3D[3 * i + k] = gmean[k];

        /* --- cov3D -> scale, quaternion (only when the kernel built cov3D itself) --- */
        if (scales && rotations && dL_dscale && dL_drot) {
            const real* q = rotations + 4 * i; const real* s = scales + 3 * i;
            real r = q[0], qx = q[1], qy = q[2], qz = q[3];
            real Rm[9] = { R_(1) - R_(2) * (qy * qy + qz * qz), R_(2) * (qx * qy - r * qz), R_(2) * (qx * qz + r * qy),
                           R_(2) * (qx * qy + r * qz), R_(1) - R_(2) * (qx * qx + qz * qz), R_(2) * (qy * qz - r * qx),
                           R_(2) * (qx * qz - r * qy), R_(2) * (qy * qz + r * qx), R_(1) - R_(2) * (qx * qx + qy * qy) };
            real sc[3] = { scale_modifier * s[0], scale_modifier * s[1], scale_modifier * s[2] };
            const real* g6 = dL_dcov3D + 6 * i;
            /* symmetric dL/dSigma with off-diagonals split evenly */
            real Gs[9] = { g6[0], R_(0.5) * g6[1], R_(0.5) * g6[2], R_(0.5) * g6[1], g6[3], R_(0.5) * g6[4],
                           R_(0.5) * g6[2], R_(0.5) * g6[4], g6[5] };
            real L[9], gL[9];
            for (int a_ = 0; a_ < 3; a_++) for (int k = 0; k < 3; k++) L[3 * a_ + k] = Rm[3 * a_ + k] * sc[k];
            for (int a_ = 0; a_ < 3; a_++) for (int k = 0; k < 3; k++)
                gL[3 * a_ + k] = R_(2) * (Gs[3 * a_] * L[k] + Gs[3 * a_ + 1] * L[3 + k] + Gs[3 * a_ + 2] * L[6 + k]);
            real gR[9];
            for (int k = 0; k < 3; k++) {
                dL_dscale[3 * i + k] = scale_modifier * (gL[k] * Rm[k] + gL[3 + k] * Rm[3 + k] + gL[6 + k] * Rm[6 + k]);
                for (int a_ = 0; a_ < 3; a_++) gR[3 * a_ + k] = gL[3 * a_ + k] * sc[k];
            }
            dL_drot[4 * i + 0] = R_(2) * (-qz * gR[1] + qy * gR[2] + qz * gR[3] - qx * gR[5] - qy * gR[6] + qx * gR[7]);
            dL_drot[4 * i + 1] = R_(2) * (qy * gR[1] + qz * gR[2] + qy * gR[3] - R_(2) * qx * gR[4] - r * gR[5] + qz * gR[6] + r * gR[7] - R_(2) * qx * gR[8]);
            dL_drot[4 * i + 2] = R_(2) * (-R_(2) * qy * gR[0] + qx * gR[1] + r * gR[2] + qx * gR[3] + qz * gR[5] - r * gR[6] + qz * gR[7] - R_(2) * qy * gR[8]);
            dL_drot[4 * i + 3] = R_(2) * (-R_(2) * qz * gR[0] - r * gR[1] + qx * gR[2] + r * gR[3] - R_(2) * qz * gR[4] + qy * gR[5] + qx * gR[6] + qy * gR[7]);
        }
    }
    return 0;
}

/* mark_visible: the frustum test alone (near plane). */
int egso_mark_visible(int P, const real* means3D, const real* viewmatrix, uint8_t* present) {
    for (int i = 0; i < P; i++) { real t[3]; xform43(means3D + 3 * i, viewmatrix, t); present[i] = t[2] > R_(0.2); }
    return 0;
}
