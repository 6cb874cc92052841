// blend_common.h -- pieces shared by the forward and backward blend kernels (gfx950).
#pragma once
#include "egs_common.h"

// Tiles are handed to workgroups so that each XCD (workgroup b runs on XCD b % 8 on MI355X -- used for
// L2 affinity only, never for correctness) owns contiguous runs of tile rows: neighbouring tiles
// share most of their splats, so a run keeps its splat records in that XCD's private 4 MiB L2.
// EGS_BAND_SEGMENTS runs per XCD: the image is cut into 8 x EGS_BAND_SEGMENTS segments of consecutive tiles and segment g belongs
// to XCD g % 8 -- one segment each (= one band) keeps the most records private, several spread a scene whose cost varies down the
// image over the XCDs (a launch ends with its busiest XCD: profiles/r4_trained_scene.md).  Band-local index i of XCD x <-> tile:
#define EGS_XCDS 8
#ifndef EGS_BAND_SEGMENTS
#define EGS_BAND_SEGMENTS 4
#endif
__host__ __device__ __forceinline__ int egs_band_seg(int n_tiles) { return (n_tiles + EGS_XCDS * EGS_BAND_SEGMENTS - 1) / (EGS_XCDS * EGS_BAND_SEGMENTS); }
__host__ __device__ __forceinline__ int egs_tiles_per_xcd(int n_tiles) { return egs_band_seg(n_tiles) * EGS_BAND_SEGMENTS; }      // slots of a band
__host__ __forceinline__ int egs_blocks_for_tiles(int n_tiles) { return egs_tiles_per_xcd(n_tiles) * EGS_XCDS; }
__host__ __device__ __forceinline__ int egs_band_tile(int x, int i, int n_tiles) {     // -1 beyond the band's last tile (valid i form a prefix)
    const int seg = egs_band_seg(n_tiles), sg = i / seg;
    const int t = (sg * EGS_XCDS + x) * seg + (i - sg * seg);
    return (sg < EGS_BAND_SEGMENTS && t < n_tiles) ? t : -1;
}
__host__ __device__ __forceinline__ int egs_band_count(int x, int n_tiles) {          // tiles of band x
    const int seg = egs_band_seg(n_tiles);
    int n = 0;
    for (int sg = 0; sg < EGS_BAND_SEGMENTS; sg++) n += max(0, min(seg, n_tiles - (sg * EGS_XCDS + x) * seg));
    return n;
}
__device__ __forceinline__ int egs_tile_of_block(unsigned b, int n_tiles) {
    return (b / EGS_XCDS) < (unsigned)egs_tiles_per_xcd(n_tiles) ? egs_band_tile((int)(b % EGS_XCDS), (int)(b / EGS_XCDS), n_tiles) : -1;
}

__device__ __forceinline__ void egs_load_rec(const float4* __restrict__ rec, uint32_t id, bool ok, float4& a,
                                             float4& b, float4& c) {
    if (ok) {
        const float4* r = rec + (size_t)id * EGS_SPLAT_REC_F4;
        a = r[0]; b = r[1]; c = r[2];
    } else {
        a = b = make_float4(0.f, 0.f, 0.f, 0.f);
        c = make_float4(0.f, 0.f, __uint_as_float(1u), __uint_as_float(1u));     // empty bbox
    }
}

// Can the splat reach the pixel block [qx0,qx1] x [qy0,qy1] at all?  Two stages, both conservative (never a false "no"):
//   1. the record's pixel bounding box (c2.z = x0 | x1<<16, c2.w = y0 | y1<<16, 15-bit fields) must intersect the block;
//   2. exact: max over the block of the (concave) log2 falloff  qa dx^2 + qb dx dy + qc dy^2  must reach
//      log2(1/(255 o)) -- below that, alpha < 1/255 for every pixel of the block.  If the centre is outside the
//      block the maximum sits on one of the (at most two) edges facing the centre, at the clamped 1-D optimum.
// Runs once per staged splat (one lane each), i.e. 1/64 of an instruction per candidate per wave, and removes the
// (wave, splat) visits that a bounding box cannot: diagonal / elongated splats and block corners.  The ellipse stage alone
// is also what the tile bucketing applies to whole 16x16 tiles (binning.hip).
__device__ __forceinline__ bool egs_ellipse_hits(float cx, float cy, float qa, float qb, float qc, float opacity, uint32_t qx0,
                                                 uint32_t qx1, uint32_t qy0, uint32_t qy1) {
    const float lx = (float)qx0 - cx, hx = (float)qx1 - cx, ly = (float)qy0 - cy, hy = (float)qy1 - cy;
    const float dxe = fminf(fmaxf(0.f, lx), hx), dye = fminf(fmaxf(0.f, ly), hy);      // nearest block point to the centre
    if (dxe == 0.f && dye == 0.f) return true;                                           // centre inside the block
    // Only a proper ellipse (negative-definite form: qa, qc < 0 and qb^2 < 4 qa qc, with a margin) has its block maximum at the
    // clamped edge optimum computed below.  The reference rejects det == 0 only, so an indefinite conic is reachable (a
    // cov3D_precomp that is not positive semi-definite, extreme scales): such a splat is never culled.
    if (!(qa < 0.f && qc < 0.f && qb * qb < 3.99f * qa * qc)) return true;
    const float dys = fminf(fmaxf(-0.5f * qb * dxe * __builtin_amdgcn_rcpf(qc), ly), hy);     // best dy on the line dx = dxe
    const float dxs = fminf(fmaxf(-0.5f * qb * dye * __builtin_amdgcn_rcpf(qa), lx), hx);     // best dx on the line dy = dye
    const float q1 = qa * dxe * dxe + qb * dxe * dys + qc * dys * dys;
    const float q2 = qa * dxs * dxs + qb * dxs * dye + qc * dye * dye;
    const float need = -__builtin_amdgcn_logf(255.f * opacity) - 0.03f;                  // log2(1/(255 o)), with a rounding margin
    return !(fmaxf(q1, q2) < need);                                                      // NaN anywhere -> keep
}

// The same test with the per-splat quantities hoisted (tile bucketing: computed once per splat, used for every tile of
// its rectangle):  need = log2(1/(255 o)) - margin,  sy = -qb/(2 qc),  sx = -qb/(2 qa).
__device__ __forceinline__ float4 egs_ellipse_prep(float qa, float qb, float qc, float opacity) {
    return make_float4(qc, -__builtin_amdgcn_logf(255.f * opacity) - 0.03f, -0.5f * qb * __builtin_amdgcn_rcpf(qc),
                       -0.5f * qb * __builtin_amdgcn_rcpf(qa));
}
__device__ __forceinline__ bool egs_ellipse_hits_prepped(const float4& e0 /*x, y, qa, qb*/, const float4& e1 /*qc, need, sy, sx*/,
                                                         uint32_t qx0, uint32_t qx1, uint32_t qy0, uint32_t qy1) {
    const float qa = e0.z, qb = e0.w, qc = e1.x;
    const float lx = (float)qx0 - e0.x, hx = (float)qx1 - e0.x, ly = (float)qy0 - e0.y, hy = (float)qy1 - e0.y;
    const float dxe = fminf(fmaxf(0.f, lx), hx), dye = fminf(fmaxf(0.f, ly), hy);
    if (dxe == 0.f && dye == 0.f) return true;
    if (!(qa < 0.f && qc < 0.f && qb * qb < 3.99f * qa * qc)) return true;                // not a proper ellipse: keep (see egs_ellipse_hits)
    const float dys = fminf(fmaxf(e1.z * dxe, ly), hy);
    const float dxs = fminf(fmaxf(e1.w * dye, lx), hx);
    const float q1 = qa * dxe * dxe + qb * dxe * dys + qc * dys * dys;
    const float q2 = qa * dxs * dxs + qb * dxs * dye + qc * dye * dye;
    return !(fmaxf(q1, q2) < e1.y);
}

__device__ __forceinline__ bool egs_block_hits(const float4& c0, const float4& c1, const float4& c2, uint32_t qx0,
                                               uint32_t qx1, uint32_t qy0, uint32_t qy1) {
    const uint32_t bx = __float_as_uint(c2.z), by = __float_as_uint(c2.w);
    const uint32_t x0 = bx & EGS_BOX_MASK, x1 = (bx >> 16) & EGS_BOX_MASK, y0 = by & EGS_BOX_MASK, y1 = (by >> 16) & EGS_BOX_MASK;     // (the other bits: egs_hot_code)
    if (!(x0 <= qx1 && x1 >= qx0 && y0 <= qy1 && y1 >= qy0)) return false;
    return egs_ellipse_hits(c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, qx0, qx1, qy0, qy1);
}

// log2 of the Gaussian falloff from the pre-scaled conic (egs_common.h): qa dx^2 + qb dx dy + qc dy^2.
// Explicit fma / mul so the forward and the backward evaluate bit-identical values.
__device__ __forceinline__ float egs_log2_falloff(float dx, float dy, float qa, float qb, float qc) {
    const float t = __fmaf_rn(qb, dy, __fmul_rn(qa, dx));
    return __fmaf_rn(__fmul_rn(qc, dy), dy, __fmul_rn(t, dx));
}

__device__ __forceinline__ float egs_alpha_noexp(float dx, float dy, float qa, float qb, float qc, float o, float& G) {   // ablation builds only
    const float p = egs_log2_falloff(dx, dy, qa, qb, qc);
    G = p * 0.001f + 1.f;
    float a = fminf(0.99f, __fmul_rn(o, G));
    a = p > 0.f ? 0.f : a;
    a = a < (1.0f / 255.0f) ? 0.f : a;
    return a;
}
// alpha = min(0.99, o * G) with the published skip rules folded in: returns 0 when the pair is skipped
// (power > 0 or alpha < 1/255), so "alpha > 0" means "kept".  G = exp(power) is returned for the backward.
// Both predicates stay in VCC -> v_cndmask form (no scalar mask round trip).
__device__ __forceinline__ float egs_alpha(float dx, float dy, float qa, float qb, float qc, float o, float& G) {
    const float p = egs_log2_falloff(dx, dy, qa, qb, qc);
    G = __builtin_amdgcn_exp2f(p);
    float a = fminf(0.99f, __fmul_rn(o, G));
    a = p > 0.f ? 0.f : a;
    a = a < (1.0f / 255.0f) ? 0.f : a;
    return a;
}

// ---- hand-placed LDS reads of one staged splat record (10 floats at `addr`, an LDS byte offset) and the waits that go with them.
// The compiler does not see these reads as asynchronous, so every register set passes through a wait (tied to it by "+v") before
// its first use, and egs_lds_wait_all stands before the registers can be reused for anything else.
typedef float egs_f4 __attribute__((ext_vector_type(4)));
typedef float egs_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void egs_lds_fetch(unsigned addr, egs_f4& r0, egs_f4& r1, egs_f2& r2) {
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b64 %2, %3 offset:32"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "v"(addr) : "memory");
}
// everything but the newest fetch (three reads) has landed
__device__ __forceinline__ void egs_lds_wait_older(egs_f4& r0, egs_f4& r1, egs_f2& r2) {
    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(r0), "+v"(r1), "+v"(r2) : : "memory");
}
__device__ __forceinline__ void egs_lds_wait_all(egs_f4& a0, egs_f4& a1, egs_f2& a2, egs_f4& b0, egs_f4& b1, egs_f2& b2) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b0), "+v"(b1), "+v"(b2) : : "memory");
}
