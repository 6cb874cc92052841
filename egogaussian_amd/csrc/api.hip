// api.hip -- the C ABI of libegs_raster.so (declared in include/egs_raster.h): argument checks, carving
// of the three caller-owned byte buffers, and the launch sequence on the caller's stream.
// Replaces the pybind `_C` entry points of the upstream extension bound at
// /root/reference/gaussian_renderer/__init__.py:14.
#include "egs_common.h"
#include "backward_prologue.h"
#include <string.h>
#include <vector>
#include <mutex>
#include <unordered_map>
#include <stdlib.h>

namespace {

struct GeomLayout { egs_geom_layout o; size_t bytes; };
struct BinLayout { egs_binning_layout o; size_t bytes; };
struct ImgLayout { egs_image_layout o; size_t bytes; };

GeomLayout geom_layout(int P) {
    GeomLayout L; size_t off = 0; const size_t n = (size_t)(P > 0 ? P : 0);
    L.o.rec = off;          off = egs_align(off + n * sizeof(float4) * EGS_SPLAT_REC_F4);
    L.o.rect = off;         off = egs_align(off + n * sizeof(uint2));
    L.o.offsets = off;      off = egs_align(off + n * sizeof(uint32_t));
    L.o.clamped = off;      off = egs_align(off + n);
    L.o.visible = off;      off = egs_align(off + n);
    L.o.scan_scratch = off; off = egs_align(off + (2 * ((n + 255) / 256) + 64) * sizeof(uint32_t)); // per-block instance counts, 64 spare words, per-block hot counts
    L.o.total = off;        off = egs_align(off + sizeof(uint64_t));
    L.bytes = off; return L;
}
BinLayout bin_layout(int P, int64_t R, int W, int H) {
    BinLayout L; size_t off = 0; const size_t n = (size_t)(R > 0 ? R : 0);
    const size_t gx = (W + EGS_TILE - 1) / EGS_TILE, gy = (H + EGS_TILE - 1) / EGS_TILE;
    const size_t nblocks = egs_bin_blocks(P > 0 ? P : 0), stride = egs_table_stride((uint32_t)nblocks), tab = gx * gy * stride;
    const size_t sums = EGS_BIN_GROUPS * egs_table_chunks(gx * gy, (uint32_t)stride);
    L.o.key_bits = egs_key_bits_for_tiles((int)(gx * gy));
    L.o.bin_blocks = (int)nblocks;
    L.o.table_stride = (int)stride;
    int ib = 0; while (P > 1 && (((unsigned)(P - 1)) >> ib) != 0) ib++;
    L.o.index_passes = (ib + 8) / 9;
    L.o.point_list = off; off = egs_align(off + n * sizeof(uint32_t));          // first: the backward needs nothing else
    L.o.pairs = off;      off = egs_align(off + n * sizeof(uint64_t));
    L.o.scratch = off;    off = egs_align(off + n * sizeof(uint64_t));
    L.o.table = off;      off = egs_align(off + (n ? tab : 0) * sizeof(uint32_t));
    L.o.spine = off;      off = egs_align(off + (n ? sums + 64 + 4 : 0) * sizeof(uint32_t));
    L.o.total = L.o.spine + ((sums + 32 + 1) & ~(size_t)1) * sizeof(uint32_t);
    L.bytes = off; return L;
}
ImgLayout img_layout(int W, int H) {
    ImgLayout L; size_t off = 0;
    const size_t gx = (W + EGS_TILE - 1) / EGS_TILE, gy = (H + EGS_TILE - 1) / EGS_TILE, px = (size_t)W * H;
    L.o.ranges = off;    off = egs_align(off + gx * gy * sizeof(uint2));
    L.o.final_T = off;   off = egs_align(off + px * sizeof(float));
    L.o.n_contrib = off; off = egs_align(off + px * sizeof(uint32_t));
    L.o.quad_work = off; off = egs_align(off + gx * gy * 4 * sizeof(uint32_t));
    L.o.tile_order = off; off = egs_align(off + (size_t)egs_blocks_for_tiles((int)(gx * gy)) * sizeof(uint32_t));       // EGS_XCDS bands of egs_tiles_per_xcd() slots
    L.o.quad_pairs = off; off = egs_align(off + gx * gy * 8 * sizeof(uint32_t));                 // pairs [tiles][4], then visits [tiles][4]
    L.bytes = off; return L;
}
EgsGeomPtrs geom_ptrs(void* buf, int P) {
    const GeomLayout L = geom_layout(P); char* b = (char*)buf; EgsGeomPtrs g;
    g.rec = (float4*)(b + L.o.rec); g.rect = (uint2*)(b + L.o.rect); g.offsets = (uint32_t*)(b + L.o.offsets);
    g.clamped = (uint8_t*)(b + L.o.clamped); g.visible = (uint8_t*)(b + L.o.visible); g.scan_scratch = (uint32_t*)(b + L.o.scan_scratch);
    g.total = (uint64_t*)(b + L.o.total);
    g.block_hot = g.scan_scratch + ((size_t)(P > 0 ? P : 0) + 255) / 256 + 64;
    return g;
}
EgsBinPtrs bin_ptrs(void* buf, int P, int64_t R, int W, int H) {
    const BinLayout L = bin_layout(P, R, W, H); char* b = (char*)buf; EgsBinPtrs p;
    p.pairs = (uint64_t*)(b + L.o.pairs); p.scratch = (uint64_t*)(b + L.o.scratch);
    p.point_list = (uint32_t*)(b + L.o.point_list); p.table = (uint32_t*)(b + L.o.table); p.chunk_sum = (uint32_t*)(b + L.o.spine);
    p.total = (uint64_t*)(b + L.o.total);
    p.flag = (uint32_t*)p.total - 32;                                 // (the 32 words before `total`)
    p.zero_after = nullptr; p.zero_after_n = 0;
    return p;
}
EgsImgPtrs img_ptrs(void* buf, int W, int H) {
    const ImgLayout L = img_layout(W, H); char* b = (char*)buf; EgsImgPtrs p;
    p.ranges = (uint2*)(b + L.o.ranges); p.final_T = (float*)(b + L.o.final_T);
    p.n_contrib = (uint32_t*)(b + L.o.n_contrib); p.quad_work = (uint32_t*)(b + L.o.quad_work);
    p.tile_order = (uint32_t*)(b + L.o.tile_order); p.quad_pairs = (uint32_t*)(b + L.o.quad_pairs);
    p.fwd_cost = nullptr; p.fwd_order = nullptr; return p;                // (set from the caller's placement buffer, forward_impl)
}

int check_dims(int P, int W, int H) {
    if (P < 0 || W <= 0 || H <= 0) return EGS_ERR_ARG;
    if (W > 32767 || H > 32767) return EGS_ERR_RANGE;                    // 15-bit pixel coordinates in the record's box (egs_common.h)
    if ((size_t)((W + EGS_TILE - 1) / EGS_TILE) * (size_t)((H + EGS_TILE - 1) / EGS_TILE) > EGS_MAX_TILES) return EGS_ERR_RANGE;
    return 0;
}
// The three opaque buffers hold 16-byte vectors at 256-byte-aligned offsets (float4 records, uint4 table words, 64-bit pairs): their
// base addresses must be 256-byte aligned (include/egs_raster.h, Conventions).  NULL passes: emptiness is checked where it matters.
static inline bool misaligned(const void* a, const void* b = nullptr, const void* c = nullptr) {
    return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 255u) != 0;
}
int check_modes(const float* shs, const float* colors, const float* scales, const float* rots, const float* cov, int act) {
    if (act & ~(EGS_ACT_LOG_SCALES | EGS_ACT_RAW_QUATS | EGS_ACT_LOGIT_OPACITY)) return EGS_ERR_MODE;
    if ((act & (EGS_ACT_LOG_SCALES | EGS_ACT_RAW_QUATS)) && cov != nullptr) return EGS_ERR_MODE;   // nothing to activate: the covariance is given
    if ((shs != nullptr) == (colors != nullptr)) return EGS_ERR_MODE;
    const bool sr = scales != nullptr && rots != nullptr;
    if ((scales != nullptr) != (rots != nullptr)) return EGS_ERR_MODE;
    if (sr == (cov != nullptr)) return EGS_ERR_MODE;
    return 0;
}

#define EGS_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)
#define EGS_SYNC_IF_DEBUG(s) do { if (debug & EGS_CALL_SYNC) { EGS_TRY(hipStreamSynchronize(s)); EGS_TRY(hipGetLastError()); } } while (0)

// ---- stage timing pool -----------------------------------------------------------------------------
struct ProfRec { int stage; hipEvent_t a, b; bool closed; };
ProfRec* g_prof = nullptr; int g_prof_cap = 0, g_prof_n = 0; int g_prof_open[EGS_K_COUNT];

}  // namespace

void egs_prof_start(int stage, hipStream_t s) {
    if (!g_prof || g_prof_n >= g_prof_cap) return;
    ProfRec& r = g_prof[g_prof_n];
    r.stage = stage; r.closed = false;
    if (hipEventRecord(r.a, s) != hipSuccess) return;
    g_prof_open[stage] = g_prof_n++;
}
void egs_prof_stop(int stage, hipStream_t s) {
    if (!g_prof) return;
    const int i = g_prof_open[stage];
    if (i < 0) return;
    if (hipEventRecord(g_prof[i].b, s) == hipSuccess) g_prof[i].closed = true;
    g_prof_open[stage] = -1;
}

extern "C" {

int egs_profile_begin(int max_records) {
    if (g_prof || max_records <= 0) return EGS_ERR_ARG;
    g_prof = new ProfRec[max_records];
    for (int i = 0; i < max_records; i++) {
        if (hipEventCreate(&g_prof[i].a) != hipSuccess || hipEventCreate(&g_prof[i].b) != hipSuccess) return EGS_ERR_NO_DEVICE;
        g_prof[i].closed = false;
    }
    for (int k = 0; k < EGS_K_COUNT; k++) g_prof_open[k] = -1;
    g_prof_cap = max_records; g_prof_n = 0;
    return 0;
}
int egs_profile_end(double* total_ms, int* launches) {
    if (!g_prof || !total_ms || !launches) return EGS_ERR_ARG;
    for (int k = 0; k < EGS_K_COUNT; k++) { total_ms[k] = 0.0; launches[k] = 0; }
    for (int i = 0; i < g_prof_n; i++) {
        if (!g_prof[i].closed) continue;
        float ms = 0.f;
        if (hipEventSynchronize(g_prof[i].b) == hipSuccess && hipEventElapsedTime(&ms, g_prof[i].a, g_prof[i].b) == hipSuccess) {
            total_ms[g_prof[i].stage] += ms; launches[g_prof[i].stage]++;
        }
    }
    for (int i = 0; i < g_prof_cap; i++) { (void)hipEventDestroy(g_prof[i].a); (void)hipEventDestroy(g_prof[i].b); }
    delete[] g_prof; g_prof = nullptr; g_prof_cap = g_prof_n = 0;
    return 0;
}
const char* egs_profile_stage_name(int stage) {
    static const char* n[EGS_K_COUNT] = { "preprocess", "scan", "tile_bucket", "tile_sort", "tile_ranges", "render_forward",
                                          "render_backward", "preprocess_backward" };
    return stage >= 0 && stage < EGS_K_COUNT ? n[stage] : "?";
}

int egs_abi_version(void) { return EGS_ABI_VERSION; }
#ifndef EGS_SOURCE_HASH
#define EGS_SOURCE_HASH "unknown"
#endif
const char* egs_source_hash(void) { return EGS_SOURCE_HASH; }

#ifdef EGS_LG_CHECK
// experiment hook: the next backward blends compute the image-loss gradient themselves (all NULL: off)
int egs_debug_set_lossgrad(const float* img, const float* gt, const float* m0, const float* m1, const float* m2, const float* gate,
                           const float* upstream, float lambda_dssim) {
    egs_debug_lossgrad = EgsLossGradHost{ img, gt, m0, m1, m2, gate, upstream, nullptr, 1.f - lambda_dssim, lambda_dssim };
    return 0;
}
#endif
// Per-call flags (include/egs_raster.h EGS_CALL_*, ABI 6): what used to be process-wide switches (egs_debug_set_tile_culling / _fused_count /
// _sort_in_blend / egs_debug_force_ballot_rank through ABI 5) travels in the last argument of the call it applies to; the library keeps none of it.
static inline bool fused_count_on(int flags) { return !(flags & EGS_CALL_SEPARATE_COUNT); }
static inline bool sort_in_blend_on(int flags) { return !(flags & (EGS_CALL_SEPARATE_SORT | EGS_CALL_SYNC)); }     // (SYNC checks after every launch: keep them apart)
static inline int cull_on(int flags) { return (flags & EGS_CALL_KEEP_ALL_INSTANCES) ? 0 : 1; }
int egs_forward_fuses_count(int P, int width, int height, int flags) { return (fused_count_on(flags) && egs_can_fuse_count(P, width, height, cull_on(flags))) ? 1 : 0; }

const char* egs_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case EGS_ERR_ARG: return "egs: invalid argument (null pointer or negative size)";
        case EGS_ERR_MODE: return "egs: provide exactly one of {shs, colors_precomp} and exactly one of {cov3D_precomp, (scales, rotations)}";
        case EGS_ERR_RANGE: return "egs: size outside the supported range";
        case EGS_ERR_NO_DEVICE: return "egs: no usable HIP device";
        case EGS_RETRY_LARGER: return "egs: binning capacity too small for this frame; call egs_forward_render with a buffer for R";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "egs: unknown error";
    }
}

int egs_device_info(char* name, int name_len, char* arch, int arch_len, int* compute_units) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return EGS_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    EGS_TRY(hipGetDeviceProperties(&prop, dev));
    if (name && name_len > 0) { strncpy(name, prop.name, (size_t)name_len - 1); name[name_len - 1] = 0; }
    if (arch && arch_len > 0) { strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1); arch[arch_len - 1] = 0; }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    return 0;
}

size_t egs_geom_bytes(int P) { return geom_layout(P).bytes; }
size_t egs_binning_bytes(int P, int64_t R, int width, int height) { return bin_layout(P, R, width, height).bytes; }
size_t egs_image_bytes(int width, int height) { return img_layout(width, height).bytes; }
size_t egs_backward_scratch_bytes(int P) { return egs_align(egs_acc_floats((size_t)(P > 0 ? P : 0)) * sizeof(float)); }

int egs_get_geom_layout(int P, egs_geom_layout* out) { if (!out || P < 0) return EGS_ERR_ARG; *out = geom_layout(P).o; return 0; }
int egs_get_binning_layout(int P, int64_t R, int width, int height, egs_binning_layout* out) {
    if (!out || P < 0 || R < 0 || width <= 0 || height <= 0) return EGS_ERR_ARG;
    *out = bin_layout(P, R, width, height).o; return 0;
}
// placement buffer: [4 n_tiles] quadrant costs, [egs_blocks_for_tiles] the forward's tile order, then (256-byte aligned, ABI 5) the chunk sums
// of a fused count pass: EGS_BIN_GROUPS x n_tiles words (a table row never spans more than one scan chunk: n_chunks <= n_tiles)
static size_t placement_sums_offset(size_t nt) { return egs_align((nt * 4 + (size_t)egs_blocks_for_tiles((int)nt)) * sizeof(uint32_t)); }
static size_t placement_sums_words(size_t nt) { return (size_t)EGS_BIN_GROUPS * nt; }
size_t egs_placement_bytes(int width, int height) {
    if (width <= 0 || height <= 0) return 0;
    const size_t nt = (size_t)((width + EGS_TILE - 1) / EGS_TILE) * (size_t)((height + EGS_TILE - 1) / EGS_TILE);
    return egs_align(placement_sums_offset(nt) + placement_sums_words(nt) * sizeof(uint32_t));
}
// The sums region of a placement buffer must be ZERO when a fused forward starts: egs_placement_init() once per (buffer, image size), the
// chain itself leaves it zero again (include/egs_raster.h).  The library keeps no record of buffers (ABI 6; through ABI 5 a process-wide
// map of "clean" addresses stood here, which a freed-and-reallocated address could fool).
int egs_placement_init(void* placement, int width, int height, void* stream) {
    if (!placement || check_dims(0, width, height)) return EGS_ERR_ARG;
    const size_t nt = (size_t)((width + EGS_TILE - 1) / EGS_TILE) * (size_t)((height + EGS_TILE - 1) / EGS_TILE);
    EGS_TRY(egs_launch_zero_u32((uint32_t*)((char*)placement + placement_sums_offset(nt)), placement_sums_words(nt), (hipStream_t)stream));
    return 0;
}
// A chain that fails between the fused count pass and the launch that clears the sums would leave them dirty for the next frame: the failing
// call clears them itself on its way out (best effort; the caller gets the error code either way).
struct PlacementClearOnError {
    void* p = nullptr; int w = 0, h = 0; hipStream_t s = nullptr;
    ~PlacementClearOnError() { if (p) (void)egs_placement_init(p, w, h, (void*)s); }
};
static void placement_sums(void* placement, int width, int height, uint32_t** sums, uint32_t* words) {
    const size_t nt = (size_t)((width + EGS_TILE - 1) / EGS_TILE) * (size_t)((height + EGS_TILE - 1) / EGS_TILE);
    *sums = (uint32_t*)((char*)placement + placement_sums_offset(nt)); *words = (uint32_t)placement_sums_words(nt);
}
int egs_order_words(int width, int height) {
    if (width <= 0 || height <= 0) return 0;
    return egs_blocks_for_tiles(((width + EGS_TILE - 1) / EGS_TILE) * ((height + EGS_TILE - 1) / EGS_TILE));
}
static void placement_ptrs(void* placement, int width, int height, EgsImgPtrs& im) {
    const size_t nt = (size_t)((width + EGS_TILE - 1) / EGS_TILE) * (size_t)((height + EGS_TILE - 1) / EGS_TILE);
    im.fwd_cost = (uint32_t*)placement; im.fwd_order = placement ? (uint32_t*)placement + nt * 4 : nullptr;
}

int egs_get_image_layout(int width, int height, egs_image_layout* out) {
    if (!out || width <= 0 || height <= 0) return EGS_ERR_ARG;
    *out = img_layout(width, height).o; return 0;
}

// egs_object_rotation (HOST struct) -> kernel argument; needs the scales + rotations mode
static int obj_rot_args(const egs_object_rotation* rot, const float* scales, EgsObjRot& r) {
    r = EgsObjRot{ nullptr, nullptr, 1.f, nullptr };
    if (!rot || !rot->M9) return 0;
    if (!scales) return EGS_ERR_MODE;
    r.M = rot->M9; r.sel = rot->selected; r.mult = rot->row0_grad_mult; r.mult_dev = rot->row0_grad_mult_dev;
    return 0;
}

int egs_forward_geometry(int P, int sh_degree, int sh_coeffs, const float* means3D, const float* shs, const float* shs_rest,
                         const float* colors_precomp, const float* opacities, const float* scales,
                         float scale_modifier, const float* rotations, const float* cov3D_precomp, int activation_flags,
                         const float* viewmatrix, const float* projmatrix, const float* campos, int width, int height,
                         float tan_fovx, float tan_fovy, int prefiltered, int32_t* radii, void* geom_buffer,
                         int64_t* num_rendered, const int32_t* active_count, const egs_object_rotation* rot, void* stream, int debug) {
    (void)prefiltered;
    int rc = check_dims(P, width, height); if (rc) return rc;
    if (!num_rendered) return EGS_ERR_ARG;
    *num_rendered = 0;
    if (P == 0) return 0;
    if (!means3D || !opacities || !viewmatrix || !projmatrix || !campos || !radii || !geom_buffer) return EGS_ERR_ARG;
    if (misaligned(geom_buffer)) return EGS_ERR_ARG;
    rc = check_modes(shs, colors_precomp, scales, rotations, cov3D_precomp, activation_flags); if (rc) return rc;
    if (shs && (sh_degree < 0 || sh_degree > EGS_MAX_SH_DEGREE || sh_coeffs < (sh_degree + 1) * (sh_degree + 1))) return EGS_ERR_RANGE;
    if (shs_rest && (!shs || sh_coeffs < 2)) return EGS_ERR_MODE;
    hipStream_t s = (hipStream_t)stream;
    EgsObjRot orot; rc = obj_rot_args(rot, scales, orot); if (rc) return rc;
    EgsGeomPtrs g = geom_ptrs(geom_buffer, P);
    EgsCamera cam = { viewmatrix, projmatrix, campos, width, height, tan_fovx, tan_fovy };
    egs_prof_start(EGS_K_PREPROCESS, s);
    const bool sh_apart = shs && (sh_coeffs > 1 || shs_rest);         // rows of 12 M bytes: the wave-tiled kernel (preprocess.hip)
    EGS_TRY(egs_launch_preprocess(P, sh_degree, sh_coeffs, means3D, sh_apart ? nullptr : shs, colors_precomp, opacities, scales, scale_modifier,
                                  rotations, activation_flags, cov3D_precomp, cam, radii, g, nullptr, 0, active_count, nullptr, orot, s));
    if (sh_apart) EGS_TRY(egs_launch_sh_forward(P, sh_degree, sh_coeffs, means3D, shs, shs_rest, cam, g, s));
    egs_prof_stop(EGS_K_PREPROCESS, s);
    EGS_SYNC_IF_DEBUG(s);
    // R = sum of the per-block instance counts (a few KB device->host; the only host wait of the forward)
    const size_t nb = ((size_t)P + 255) / 256;
    std::vector<uint32_t> sums(nb);
    EGS_TRY(hipMemcpyAsync(sums.data(), g.scan_scratch, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    EGS_TRY(hipStreamSynchronize(s));
    uint64_t R = 0;
    for (size_t k = 0; k < nb; k++) R += sums[k];
    if (R >= (1ull << 31)) return EGS_ERR_RANGE;
    *num_rendered = (int64_t)R;
    return 0;
}

// One-call forward: preprocess, then -- without waiting for R -- the whole binning + blend chain is enqueued against
// the caller's capacity guess while the host waits only for the copy of the per-block instance counts.  The GPU never
// idles on the host round trip.  If the guess was too small nothing valid was produced: the true R is returned and the
// caller finishes with egs_forward_render on a buffer of the right size.
static int forward_impl(int wait_for_count, int P, int sh_degree, int sh_coeffs, const float* means3D, const float* shs, const float* shs_rest, const float* colors_precomp,
                const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                const float* cov3D_precomp, int activation_flags, const float* viewmatrix, const float* projmatrix, const float* campos,
                const float* background, int width, int height, float tan_fovx, float tan_fovy, int prefiltered,
                int32_t* radii, void* geom_buffer, int64_t capacity, void* binning_buffer, void* image_buffer,
                float* out_color, float* out_depth, float* out_alpha, uint32_t* pinned_host_counts, uint64_t* running_max,
                int64_t* num_rendered, const int32_t* active_count, uint32_t* overflow_flag, void* placement, const egs_object_rotation* rot,
                void* stream, int debug) {
    (void)prefiltered;
    int rc = check_dims(P, width, height); if (rc) return rc;
    if (!num_rendered || capacity < 0 || capacity >= (1ll << 31)) return EGS_ERR_ARG;
    *num_rendered = 0;
    if (!background || !image_buffer || !out_color || ((out_depth != nullptr) != (out_alpha != nullptr))) return EGS_ERR_ARG;     // (depth and alpha: both or neither, ABI 4)
    if (misaligned(geom_buffer, binning_buffer, image_buffer)) return EGS_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (P == 0) return egs_forward_render(0, 0, background, width, height, geom_buffer, binning_buffer, image_buffer, out_color,
                                          out_depth, out_alpha, stream, debug);
    if (!means3D || !opacities || !viewmatrix || !projmatrix || !campos || !radii || !geom_buffer) return EGS_ERR_ARG;
    if (wait_for_count && !pinned_host_counts) return EGS_ERR_ARG;
    if (capacity > 0 && !binning_buffer) return EGS_ERR_ARG;
    if (!wait_for_count && capacity <= 0) return EGS_ERR_ARG;
    rc = check_modes(shs, colors_precomp, scales, rotations, cov3D_precomp, activation_flags); if (rc) return rc;
    if (shs && (sh_degree < 0 || sh_degree > EGS_MAX_SH_DEGREE || sh_coeffs < (sh_degree + 1) * (sh_degree + 1))) return EGS_ERR_RANGE;
    if (shs_rest && (!shs || sh_coeffs < 2)) return EGS_ERR_MODE;
    EgsObjRot orot; rc = obj_rot_args(rot, scales, orot); if (rc) return rc;
    static thread_local hipEvent_t ev = nullptr;
    if (wait_for_count && !ev) EGS_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    EgsGeomPtrs g = geom_ptrs(geom_buffer, P);
    EgsCamera cam = { viewmatrix, projmatrix, campos, width, height, tan_fovx, tan_fovy };
    egs_prof_start(EGS_K_PREPROCESS, s);
    const bool sh_apart = shs && (sh_coeffs > 1 || shs_rest);         // rows of 12 M bytes: the wave-tiled kernel (preprocess.hip)
    // the speculative bucketing that follows needs its chunk sums cleared: the preprocess launch does that on the side (a launch of
    // its own costs ~4.5 us of GPU time whatever it does)
    EgsBinPtrs b_spec = {};
    size_t n_sums = 0;
    if (capacity > 0) {
        b_spec = bin_ptrs(binning_buffer, P, capacity, width, height);
        const size_t nt = (size_t)((width + EGS_TILE - 1) / EGS_TILE) * (size_t)((height + EGS_TILE - 1) / EGS_TILE);
        n_sums = EGS_BIN_GROUPS * egs_table_chunks(nt, egs_table_stride(egs_bin_blocks(P)));
    }
    // ... and carries the placement of the forward blend's tiles (backward_prologue.h), computed from the costs the image buffer holds
    EgsImgPtrs im_spec = img_ptrs(image_buffer, width, height);
    placement_ptrs(placement, width, height, im_spec);
    // With a persistent placement buffer the count pass of the bucketing rides in the preprocess launch (k_preprocess_count) and adds its
    // chunk sums into that buffer's sums region, which is zero between frames (egs_common.h EgsBinPtrs); egs_debug_set_fused_count: A/B switch
    const bool fuse = capacity > 0 && placement && fused_count_on(debug) && egs_can_fuse_count(P, width, height, cull_on(debug));
    PlacementClearOnError dirty_guard;
    if (fuse) {
        dirty_guard.p = placement; dirty_guard.w = width; dirty_guard.h = height; dirty_guard.s = s;      // (disarmed once the bucketing chain is enqueued)
        placement_sums(placement, width, height, &b_spec.chunk_sum, &b_spec.zero_after_n);
        b_spec.zero_after = b_spec.chunk_sum;
        EGS_TRY(egs_launch_preprocess_count(P, sh_degree, sh_coeffs, means3D, sh_apart ? nullptr : shs, colors_precomp, opacities, scales, scale_modifier,
                                            rotations, activation_flags, cov3D_precomp, cam, radii, g, b_spec, active_count, &im_spec, orot, cull_on(debug), s));
    } else
    EGS_TRY(egs_launch_preprocess(P, sh_degree, sh_coeffs, means3D, sh_apart ? nullptr : shs, colors_precomp, opacities, scales, scale_modifier,
                                  rotations, activation_flags, cov3D_precomp, cam, radii, g, capacity > 0 ? b_spec.chunk_sum : nullptr, n_sums,
                                  active_count, (capacity > 0 && placement) ? &im_spec : nullptr, orot, s));
    if (sh_apart) EGS_TRY(egs_launch_sh_forward(P, sh_degree, sh_coeffs, means3D, shs, shs_rest, cam, g, s));
    egs_prof_stop(EGS_K_PREPROCESS, s);
    const size_t nb = ((size_t)P + 255) / 256;
    if (pinned_host_counts) EGS_TRY(hipMemcpyAsync(pinned_host_counts, g.scan_scratch, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (wait_for_count) EGS_TRY(hipEventRecord(ev, s));
    if (capacity > 0) {                                              // speculative: sized by the caller's guess
        EgsBinPtrs b = b_spec;
        EgsImgPtrs im = im_spec;
        // the per-tile sort inside the forward blend's launch (render_fwd.hip SORT): egs_launch_binning then makes no sort launch and says what the
        // blend needs; with the sums of a fused count pass to clear, a failure before the blend is enqueued leaves them dirty (dirty_guard)
        EgsSortArgs sort_args = {};
        EGS_TRY(egs_launch_binning(P, capacity, width, height, g, b, im, running_max, overflow_flag, 1, fuse ? 1 : 0,
                                   !(debug & EGS_CALL_SEPARATE_SORT) ? &sort_args : nullptr, s, debug & ~EGS_CALL_SYNC));      // (speculative chain: never synchronised mid-way)
        egs_prof_start(EGS_K_RENDER_FWD, s);
        EGS_TRY(egs_launch_render_forward(width, height, background, g, b.point_list, im, out_color, out_depth, out_alpha, placement ? 1 : 0, &sort_args, s));
        dirty_guard.p = nullptr;
        egs_prof_stop(EGS_K_RENDER_FWD, s);
        // a caller that checks LATER (egs_forward_enqueue outside a graph: the eager loop without a host wait per forward) also gets the
        // overflow word -- [0] clipped, [1] instances bucketed -- behind the counts, once the chain that writes it has run
        if (!wait_for_count && pinned_host_counts && overflow_flag)
            EGS_TRY(hipMemcpyAsync(pinned_host_counts + nb, overflow_flag, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    }
    if (!wait_for_count) { *num_rendered = -1; return 0; }              // graph-capturable: no host wait at all
    EGS_TRY(hipEventSynchronize(ev));
    uint64_t R = 0;
    for (size_t k = 0; k < nb; k++) R += pinned_host_counts[k];
    if (R >= (1ull << 31)) return EGS_ERR_RANGE;
    *num_rendered = (int64_t)R;
    if ((int64_t)R > capacity || capacity == 0) return EGS_RETRY_LARGER;       // caller: egs_forward_render with a buffer for R
    EGS_SYNC_IF_DEBUG(s);
    return 0;
}

int egs_forward(int P, int sh_degree, int sh_coeffs, const float* means3D, const float* shs, const float* shs_rest, const float* colors_precomp,
                const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                const float* cov3D_precomp, int activation_flags, const float* viewmatrix, const float* projmatrix, const float* campos,
                const float* background, int width, int height, float tan_fovx, float tan_fovy, int prefiltered,
                int32_t* radii, void* geom_buffer, int64_t capacity, void* binning_buffer, void* image_buffer,
                float* out_color, float* out_depth, float* out_alpha, uint32_t* pinned_host_counts, int64_t* num_rendered,
                const int32_t* active_count, void* placement, const egs_object_rotation* rot, void* stream, int debug) {
    return forward_impl(1, P, sh_degree, sh_coeffs, means3D, shs, shs_rest, colors_precomp, opacities, scales, scale_modifier, rotations,
                        cov3D_precomp, activation_flags, viewmatrix, projmatrix, campos, background, width, height, tan_fovx, tan_fovy, prefiltered,
                        radii, geom_buffer, capacity, binning_buffer, image_buffer, out_color, out_depth, out_alpha,
                        pinned_host_counts, nullptr, num_rendered, active_count, nullptr, placement, rot, stream, debug);
}

// Same chain with NO host wait: everything is only enqueued, so the call can be captured into a hipGraph.  Overflow of
// `capacity` is detected afterwards, either from pinned_host_counts (optional copy of the per-workgroup rectangle counts;
// egs_sum_counts after synchronising) or from `running_max` (optional device uint64 that the chain raises to the number
// of instances it bucketed whenever that is larger -- one word a caller can read after any number of replays).
int egs_forward_enqueue(int P, int sh_degree, int sh_coeffs, const float* means3D, const float* shs, const float* shs_rest, const float* colors_precomp,
                        const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                        const float* cov3D_precomp, int activation_flags, const float* viewmatrix, const float* projmatrix, const float* campos,
                        const float* background, int width, int height, float tan_fovx, float tan_fovy, int prefiltered,
                        int32_t* radii, void* geom_buffer, int64_t capacity, void* binning_buffer, void* image_buffer,
                        float* out_color, float* out_depth, float* out_alpha, uint32_t* pinned_host_counts, uint64_t* running_max,
                        const int32_t* active_count, uint32_t* overflow_flag, void* placement, const egs_object_rotation* rot, void* stream, int flags) {
    int64_t unused = 0;
    return forward_impl(0, P, sh_degree, sh_coeffs, means3D, shs, shs_rest, colors_precomp, opacities, scales, scale_modifier, rotations,
                        cov3D_precomp, activation_flags, viewmatrix, projmatrix, campos, background, width, height, tan_fovx, tan_fovy, prefiltered,
                        radii, geom_buffer, capacity, binning_buffer, image_buffer, out_color, out_depth, out_alpha,
                        pinned_host_counts, running_max, &unused, active_count, overflow_flag, placement, rot, stream, flags & ~EGS_CALL_SYNC);      // (nothing may wait: capturable)
}

int64_t egs_sum_counts(int P, const uint32_t* pinned_host_counts) {
    if (P <= 0 || !pinned_host_counts) return 0;
    uint64_t R = 0;
    for (size_t k = 0, nb = ((size_t)P + 255) / 256; k < nb; k++) R += pinned_host_counts[k];
    return (int64_t)R;
}

int egs_forward_render(int P, int64_t R, const float* background, int width, int height, const void* geom_buffer,
                       void* binning_buffer, void* image_buffer, float* out_color, float* out_depth, float* out_alpha,
                       void* stream, int debug) {
    int rc = check_dims(P, width, height); if (rc) return rc;
    if (R < 0 || R >= (1ll << 31)) return EGS_ERR_RANGE;
    if (!background || !image_buffer || !out_color || ((out_depth != nullptr) != (out_alpha != nullptr))) return EGS_ERR_ARG;     // (depth and alpha: both or neither, ABI 4)
    if (P > 0 && !geom_buffer) return EGS_ERR_ARG;
    if (R > 0 && !binning_buffer) return EGS_ERR_ARG;
    if (misaligned(geom_buffer, binning_buffer, image_buffer)) return EGS_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    EgsGeomPtrs g = geom_ptrs(const_cast<void*>(geom_buffer), P);
    EgsBinPtrs b = bin_ptrs(binning_buffer, P, R, width, height);
    EgsImgPtrs im = img_ptrs(image_buffer, width, height);
    EgsSortArgs sort_args = {};
    EGS_TRY(egs_launch_binning(P, R, width, height, g, b, im, nullptr, nullptr, 0, 0, sort_in_blend_on(debug) ? &sort_args : nullptr, s, debug));
    const uint32_t* point_list = b.point_list;
    egs_prof_start(EGS_K_RENDER_FWD, s);
    EGS_TRY(egs_launch_render_forward(width, height, background, g, point_list, im, out_color, out_depth, out_alpha, 0, &sort_args, s));
    egs_prof_stop(EGS_K_RENDER_FWD, s);
    EGS_SYNC_IF_DEBUG(s);
    return 0;
}

// egs_adam_sink (HOST struct of the C ABI) -> what the kernels take by value
static void sink_to_kernel_args(const egs_adam_sink* sink, const uint32_t* skip_flag, EgsSink& ks, EgsAdamTick& tick) {
    for (int l = 0; l < EGS_SINK_LEAVES; l++) {
        if (!sink->leaf[l].param) continue;
        ks.leaf[l].p = sink->leaf[l].param; ks.leaf[l].m = sink->leaf[l].exp_avg; ks.leaf[l].v = sink->leaf[l].exp_avg_sq;
        tick.step[l] = sink->leaf[l].step; tick.lr[l] = sink->leaf[l].lr;
    }
    ks.coef = sink->coef; ks.active_rows = sink->active_rows; ks.skip = skip_flag; ks.b1 = sink->beta1; ks.b2 = sink->beta2; ks.eps = sink->eps;
    tick.coef = sink->coef; tick.skip = skip_flag; tick.b1 = sink->beta1; tick.b2 = sink->beta2;
}

static int backward_impl(int P, int sh_degree, int sh_coeffs, int64_t R, const float* background, const float* means3D,
                 const float* shs, const float* shs_rest, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, int activation_flags, const float* viewmatrix, const float* projmatrix,
                 const float* campos, int width, int height, float tan_fovx, float tan_fovy, const int32_t* radii,
                 const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
                 const float* dL_dout_color, const float* dL_dout_depth, const float* dL_dout_alpha, float* dL_dmeans2D,
                 float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dsh_rest,
                 float* dL_dscales, float* dL_drotations, float* stat_grad_accum, float* stat_denom, float* stat_max_radii,
                 const uint32_t* skip_flag, const egs_adam_sink* sink, int prologue_done, const egs_object_rotation* rot, int grad_mask, void* scratch,
                 void* stream, int debug, const egs_loss_grad* loss_grad = nullptr) {
    int rc = check_dims(P, width, height); if (rc) return rc;
    EgsObjRot orot; rc = obj_rot_args(rot, scales, orot); if (rc) return rc;
    // the image loss's gradient computed by the blend itself (egs_backward_lossgrad): colour gradients only, three channels
    EgsLossGradHost lgh = {}; const EgsLossGradHost* lgp = nullptr;
    if (loss_grad) {
        const egs_loss_grad& q = *loss_grad;
        if (!q.image || !q.gt || !q.dm_dmu1 || !q.dm_dexx || !q.dm_dexy || !q.upstream_grad) return EGS_ERR_ARG;
        if (dL_dout_depth || dL_dout_alpha || (grad_mask == EGS_GRAD_COLORS && colors_precomp && !sink && !stat_grad_accum)) return EGS_ERR_MODE;
        lgh = EgsLossGradHost{ q.image, q.gt, q.dm_dmu1, q.dm_dexx, q.dm_dexy, q.gate, q.upstream_grad, nullptr, 1.f - q.lambda_dssim, q.lambda_dssim,
                               q.deferred_partial_sums, q.deferred_partial_sums ? egs_l1_ssim_partial_count(3, height, width) / 2 : 0, q.lambda_dssim,
                               q.deferred_loss, q.deferred_partial_sums ? q.loss_running_sum : nullptr };
        lgp = &lgh;
        if (!dL_dout_color) dL_dout_color = q.image;                 // (never read: the checks below want a pointer)
    }
    if (P == 0) { if (lgp) EGS_TRY(egs_launch_loss_finish(lgh, width, height, (hipStream_t)stream)); return 0; }
    if (R < 0 || R >= (1ll << 31)) return EGS_ERR_RANGE;
    if (grad_mask & ~EGS_GRAD_MASK_BITS) return EGS_ERR_ARG;
    // Only the precomputed colours' gradient is wanted (the reference's label call): a blend that sums w dL/dC alone, no preprocess backward
    const bool colors_only = grad_mask == EGS_GRAD_COLORS && colors_precomp && !sink && !stat_grad_accum;
    if (!background || !means3D || !viewmatrix || !projmatrix || !campos || !radii || !geom_buffer || !image_buffer ||
        !dL_dout_color || (!dL_dmeans2D && !colors_only) || !scratch)
        return EGS_ERR_ARG;
    if (misaligned(geom_buffer, binning_buffer, image_buffer) || ((uintptr_t)scratch & 15u)) return EGS_ERR_ARG;
    if (colors_only) {
        if (!dL_dcolors || (R > 0 && !binning_buffer)) return EGS_ERR_ARG;
        rc = check_modes(shs, colors_precomp, scales, rotations, cov3D_precomp, activation_flags); if (rc) return rc;
        hipStream_t s = (hipStream_t)stream;
        EgsGeomPtrs g = geom_ptrs(const_cast<void*>(geom_buffer), P);
        float* grad_acc = (float*)scratch;
        if (R == 0) return (int)egs_launch_zero_u32((uint32_t*)dL_dcolors, (size_t)P * 3, s);
        EgsBinPtrs b = bin_ptrs(const_cast<void*>(binning_buffer), P, R, width, height);
        EgsImgPtrs im = img_ptrs(const_cast<void*>(image_buffer), width, height);
        if (!prologue_done) EGS_TRY(egs_launch_backward_prologue(P, width, height, im, grad_acc, g.block_hot, nullptr, s));
        egs_prof_start(EGS_K_RENDER_BWD, s);
        EGS_TRY(egs_launch_render_backward(P, width, height, background, g, b.point_list, im, dL_dout_color, nullptr, nullptr, grad_acc, 1, nullptr, s));
        egs_prof_stop(EGS_K_RENDER_BWD, s);
        EGS_SYNC_IF_DEBUG(s);
        egs_prof_start(EGS_K_PREPROCESS_BWD, s);
        EGS_TRY(egs_launch_colors_from_acc(P, grad_acc, g.clamped, radii, dL_dcolors, s));
        egs_prof_stop(EGS_K_PREPROCESS_BWD, s);
        EGS_SYNC_IF_DEBUG(s);
        return 0;
    }
    // leaves a fused optimizer owns: their gradient arrays are optional
    const bool sh_apart = shs && (sh_coeffs > 1 || shs_rest);
    bool own[EGS_SINK_LEAVES] = { false, false, false, false, false, false };
    const bool sh_sink_ok = sh_apart && egs_sh_backward_can_sink(sh_coeffs, shs, shs_rest);      // the split M = 16 kernel steps dc / rest / positions
    if (sink) {
        if (!sink->coef) return EGS_ERR_ARG;
        for (int l = 0; l < EGS_SINK_LEAVES; l++) {
            const egs_adam_leaf& f = sink->leaf[l];
            own[l] = f.param != nullptr;
            if (own[l] && (!f.exp_avg || !f.exp_avg_sq || !f.lr || !f.step)) return EGS_ERR_ARG;
        }
        if ((own[EGS_SINK_SCALES] || own[EGS_SINK_ROTATIONS]) && cov3D_precomp) return EGS_ERR_MODE;
        if (own[EGS_SINK_SH] && (!shs || (sh_apart && !sh_sink_ok))) return EGS_ERR_MODE;
        if (own[EGS_SINK_SH_REST] && (!sh_sink_ok || sink->leaf[EGS_SINK_SH_REST].param != shs_rest)) return EGS_ERR_MODE;
        if (own[EGS_SINK_MEANS3D] && sh_apart && (!sh_sink_ok || !dL_dmeans3D)) return EGS_ERR_MODE;
        if (sh_apart && (own[EGS_SINK_SH] || own[EGS_SINK_SH_REST]) && ((dL_dsh != nullptr) != (dL_dsh_rest != nullptr))) return EGS_ERR_ARG;
        if ((own[EGS_SINK_MEANS3D] && sink->leaf[EGS_SINK_MEANS3D].param != means3D) || (own[EGS_SINK_SCALES] && sink->leaf[EGS_SINK_SCALES].param != scales) ||
            (own[EGS_SINK_ROTATIONS] && sink->leaf[EGS_SINK_ROTATIONS].param != rotations) || (own[EGS_SINK_SH] && sink->leaf[EGS_SINK_SH].param != shs))
            return EGS_ERR_ARG;
    }
    if ((!dL_dcolors && (colors_precomp || sh_apart)) || (!dL_dopacity && !own[EGS_SINK_OPACITY]) || (!dL_dmeans3D && !own[EGS_SINK_MEANS3D]))
        return EGS_ERR_ARG;
    if (R > 0 && !binning_buffer) return EGS_ERR_ARG;
    rc = check_modes(shs, colors_precomp, scales, rotations, cov3D_precomp, activation_flags); if (rc) return rc;
    if (shs && !dL_dsh && !own[EGS_SINK_SH]) return EGS_ERR_ARG;
    if (sh_apart && !dL_dsh && !(own[EGS_SINK_SH] && (own[EGS_SINK_SH_REST] || !shs_rest))) return EGS_ERR_ARG;
    if (shs_rest && (!shs || sh_coeffs < 2)) return EGS_ERR_MODE;
    if (!shs_rest && dL_dsh_rest) return EGS_ERR_ARG;
    if (shs_rest && !dL_dsh_rest && !own[EGS_SINK_SH_REST]) return EGS_ERR_ARG;
    if ((stat_grad_accum != nullptr) != (stat_denom != nullptr) || (stat_max_radii && !stat_grad_accum)) return EGS_ERR_ARG;
    if (!cov3D_precomp && ((!dL_dscales && !own[EGS_SINK_SCALES]) || (!dL_drotations && !own[EGS_SINK_ROTATIONS]))) return EGS_ERR_ARG;
    if (cov3D_precomp && !dL_dcov3D) return EGS_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    EgsGeomPtrs g = geom_ptrs(const_cast<void*>(geom_buffer), P);
    EgsBinPtrs b = bin_ptrs(const_cast<void*>(binning_buffer), P, R, width, height);
    EgsImgPtrs im = img_ptrs(const_cast<void*>(image_buffer), width, height);
    float* grad_acc = (float*)scratch;
    EgsSink ks = {}, ks_sh = {}; EgsAdamTick tick = {};
    bool pp_sinks = false, sh_sinks = false;
    if (sink) {
        sink_to_kernel_args(sink, skip_flag, ks, tick);
        ks_sh = ks;
        for (int l = 0; l < EGS_SINK_LEAVES; l++) {                    // who finishes which gradient: the spherical-harmonics launch, or the one before it
            const bool by_sh = sh_apart && (l == EGS_SINK_SH || l == EGS_SINK_SH_REST || l == EGS_SINK_MEANS3D);
            if (by_sh) ks.leaf[l] = EgsSinkLeaf{ nullptr, nullptr, nullptr }; else ks_sh.leaf[l] = EgsSinkLeaf{ nullptr, nullptr, nullptr };
            pp_sinks = pp_sinks || ks.leaf[l].p; sh_sinks = sh_sinks || ks_sh.leaf[l].p;
        }
    }
    // the accumulator is cleared by a kernel, not a memset node (see egs_launch_zero_f4): fused into the blend's prologue
    if (R == 0 && !prologue_done) {
        EGS_TRY(egs_launch_zero_f4((float4*)grad_acc, egs_acc_floats((size_t)P) / 4, s));
        if (sink) EGS_TRY(egs_launch_adam_tick(tick, s));
    }
    if (R == 0 && lgp) EGS_TRY(egs_launch_loss_finish(lgh, width, height, s));
    if (R > 0) {
        const uint32_t* point_list = b.point_list;
        if (!prologue_done) EGS_TRY(egs_launch_backward_prologue(P, width, height, im, grad_acc, g.block_hot, sink ? &tick : nullptr, s));
        egs_prof_start(EGS_K_RENDER_BWD, s);                         // (the stage is the blend kernel alone)
        EGS_TRY(egs_launch_render_backward(P, width, height, background, g, point_list, im, dL_dout_color, dL_dout_depth, dL_dout_alpha, grad_acc, 0, lgp, s));
        egs_prof_stop(EGS_K_RENDER_BWD, s);
        EGS_SYNC_IF_DEBUG(s);
    }
    egs_prof_start(EGS_K_PREPROCESS_BWD, s);
    EgsCamera cam = { viewmatrix, projmatrix, campos, width, height, tan_fovx, tan_fovy };
    EGS_TRY(egs_launch_preprocess_backward(P, sh_degree, sh_coeffs, means3D, sh_apart ? nullptr : shs, scales, scale_modifier, rotations,
                                           cov3D_precomp, activation_flags, cam, radii, g, grad_acc, colors_precomp != nullptr, dL_dmeans2D,
                                           dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, sh_apart ? nullptr : dL_dsh, dL_dscales,
                                           dL_drotations, stat_grad_accum, stat_denom, stat_max_radii, skip_flag, pp_sinks ? &ks : nullptr, orot, s));
    if (sh_apart) EGS_TRY(egs_launch_sh_backward(P, sh_degree, sh_coeffs, means3D, shs, shs_rest, cam, radii, g, dL_dcolors, dL_dsh,
                                                 dL_dsh_rest, dL_dmeans3D, sh_sinks ? &ks_sh : nullptr, s));
    egs_prof_stop(EGS_K_PREPROCESS_BWD, s);
    EGS_SYNC_IF_DEBUG(s);
    return 0;
}

int egs_backward(int P, int sh_degree, int sh_coeffs, int64_t R, const float* background, const float* means3D,
                 const float* shs, const float* shs_rest, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, int activation_flags, const float* viewmatrix, const float* projmatrix,
                 const float* campos, int width, int height, float tan_fovx, float tan_fovy, const int32_t* radii,
                 const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
                 const float* dL_dout_color, const float* dL_dout_depth, const float* dL_dout_alpha, float* dL_dmeans2D,
                 float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dsh_rest,
                 float* dL_dscales, float* dL_drotations, float* stat_grad_accum, float* stat_denom, float* stat_max_radii,
                 const uint32_t* skip_flag, int grad_mask, void* scratch, void* stream, int debug) {
    if (grad_mask == EGS_GRAD_COLORS && colors_precomp && !stat_grad_accum) { if (!dL_dcolors) return P == 0 ? 0 : EGS_ERR_ARG; }
    else if (!dL_dcolors || !dL_dopacity || !dL_dmeans3D) return P == 0 ? 0 : EGS_ERR_ARG;
    return backward_impl(P, sh_degree, sh_coeffs, R, background, means3D, shs, shs_rest, colors_precomp, scales, scale_modifier, rotations,
                         cov3D_precomp, activation_flags, viewmatrix, projmatrix, campos, width, height, tan_fovx, tan_fovy, radii, geom_buffer,
                         binning_buffer, image_buffer, dL_dout_color, dL_dout_depth, dL_dout_alpha, dL_dmeans2D, dL_dcolors, dL_dopacity,
                         dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dsh_rest, dL_dscales, dL_drotations, stat_grad_accum, stat_denom, stat_max_radii,
                         skip_flag, nullptr, 0, nullptr, grad_mask, scratch, stream, debug);
}

int egs_backward_adam(int P, int sh_degree, int sh_coeffs, int64_t R, const float* background, const float* means3D,
                      const float* shs, const float* shs_rest, const float* colors_precomp, const float* scales, float scale_modifier,
                      const float* rotations, const float* cov3D_precomp, int activation_flags, const float* viewmatrix, const float* projmatrix,
                      const float* campos, int width, int height, float tan_fovx, float tan_fovy, const int32_t* radii,
                      const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
                      const float* dL_dout_color, const float* dL_dout_depth, const float* dL_dout_alpha, float* dL_dmeans2D,
                      float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dsh_rest,
                      float* dL_dscales, float* dL_drotations, float* stat_grad_accum, float* stat_denom, float* stat_max_radii,
                      const uint32_t* skip_flag, const egs_adam_sink* sink, int prologue_done, const egs_object_rotation* rot, int grad_mask, void* scratch,
                      void* stream, int debug) {
    return backward_impl(P, sh_degree, sh_coeffs, R, background, means3D, shs, shs_rest, colors_precomp, scales, scale_modifier, rotations,
                         cov3D_precomp, activation_flags, viewmatrix, projmatrix, campos, width, height, tan_fovx, tan_fovy, radii, geom_buffer,
                         binning_buffer, image_buffer, dL_dout_color, dL_dout_depth, dL_dout_alpha, dL_dmeans2D, dL_dcolors, dL_dopacity,
                         dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dsh_rest, dL_dscales, dL_drotations, stat_grad_accum, stat_denom, stat_max_radii,
                         skip_flag, sink, prologue_done, rot, grad_mask, scratch, stream, debug);
}

int egs_backward_lossgrad(int P, int sh_degree, int sh_coeffs, int64_t R, const float* background, const float* means3D,
                          const float* shs, const float* shs_rest, const float* colors_precomp, const float* scales, float scale_modifier,
                          const float* rotations, const float* cov3D_precomp, int activation_flags, const float* viewmatrix, const float* projmatrix,
                          const float* campos, int width, int height, float tan_fovx, float tan_fovy, const int32_t* radii,
                          const void* geom_buffer, const void* binning_buffer, const void* image_buffer, const egs_loss_grad* loss_grad,
                          float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dsh_rest,
                          float* dL_dscales, float* dL_drotations, float* stat_grad_accum, float* stat_denom, float* stat_max_radii,
                          const uint32_t* skip_flag, const egs_adam_sink* sink, int prologue_done, const egs_object_rotation* rot, int grad_mask, void* scratch,
                          void* stream, int debug) {
    if (!loss_grad) return EGS_ERR_ARG;
    return backward_impl(P, sh_degree, sh_coeffs, R, background, means3D, shs, shs_rest, colors_precomp, scales, scale_modifier, rotations,
                         cov3D_precomp, activation_flags, viewmatrix, projmatrix, campos, width, height, tan_fovx, tan_fovy, radii, geom_buffer,
                         binning_buffer, image_buffer, nullptr, nullptr, nullptr, dL_dmeans2D, dL_dcolors, dL_dopacity,
                         dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dsh_rest, dL_dscales, dL_drotations, stat_grad_accum, stat_denom, stat_max_radii,
                         skip_flag, sink, prologue_done, rot, grad_mask, scratch, stream, debug, loss_grad);
}

// egs_backward_prologue (HOST struct) -> the side jobs a loss backward launch carries
static int prologue_args(const egs_backward_prologue* side, EgsPrologueArgs& pa) {
    int rc = check_dims(side->P, side->width, side->height); if (rc) return rc;
    if (side->P <= 0 || !side->image_buffer || !side->scratch) return EGS_ERR_ARG;
    EgsImgPtrs im = img_ptrs(side->image_buffer, side->width, side->height);
    pa.n_tiles = ((side->width + EGS_TILE - 1) / EGS_TILE) * ((side->height + EGS_TILE - 1) / EGS_TILE);
    pa.quad_work = im.quad_work; pa.tile_order = im.tile_order;
    if (side->geom_buffer && misaligned(side->geom_buffer)) return EGS_ERR_ARG;
    egs_prologue_acc(pa, (float*)side->scratch, (size_t)side->P,
                     side->geom_buffer ? geom_ptrs(const_cast<void*>(side->geom_buffer), side->P).block_hot : nullptr);
    if (side->sink) {
        if (!side->sink->coef) return EGS_ERR_ARG;
        EgsSink ks = {};
        sink_to_kernel_args(side->sink, side->skip_flag, ks, pa.tick);
        pa.has_tick = 1;
    }
    return 0;
}

int egs_l1_ssim_backward_ex(int channels, int height, int width, const float* img, const float* gt, float lambda_dssim,
                            const float* upstream_grad, const float* gate, const float* dm_dmu1, const float* dm_dexx,
                            const float* dm_dexy, float* dL_dimg, const float* deferred_partial_sums, float* deferred_loss,
                            float* loss_running_sum, const egs_backward_prologue* side, void* stream) {
    if (!side)
        return egs_launch_l1_ssim_backward(channels, height, width, img, gt, lambda_dssim, upstream_grad, gate, dm_dmu1, dm_dexx, dm_dexy, dL_dimg,
                                           deferred_partial_sums, deferred_loss, loss_running_sum, nullptr, (hipStream_t)stream);
    EgsPrologueArgs pa = {};
    int rc = prologue_args(side, pa); if (rc) return rc;
    return egs_launch_l1_ssim_backward(channels, height, width, img, gt, lambda_dssim, upstream_grad, gate, dm_dmu1, dm_dexx, dm_dexy, dL_dimg,
                                       deferred_partial_sums, deferred_loss, loss_running_sum, &pa, (hipStream_t)stream);
}

int egs_l1_ssim_forward_ex(int channels, int height, int width, const float* img, const float* gt, float lambda_dssim,
                           float* partial_sums, float* dm_dmu1, float* dm_dexx, float* dm_dexy, float* loss, float* loss_running_sum,
                           const egs_backward_prologue* side, void* stream) {
    EgsPrologueArgs pa = {};
    if (side) { int rc = prologue_args(side, pa); if (rc) return rc; }
    return egs_launch_l1_ssim_forward(channels, height, width, img, gt, lambda_dssim, partial_sums, dm_dmu1, dm_dexx, dm_dexy, loss, loss_running_sum,
                                      side ? &pa : nullptr, (hipStream_t)stream);
}

int egs_l1_ssim_pair_backward(int channels, int height, int width, const float* img, const float* gt, const float* upstream_l1,
                              const float* upstream_ssim, const float* gate, const float* dm_dmu1, const float* dm_dexx, const float* dm_dexy,
                              float* dL_dimg, const egs_backward_prologue* side, void* stream) {
    if (!upstream_l1 || !upstream_ssim) return EGS_ERR_ARG;
    EgsPrologueArgs pa = {};
    if (side) { int rc = prologue_args(side, pa); if (rc) return rc; }
    // d(mean SSIM) = -d(1 - mean SSIM): weight -1 on the kernel's (1 - SSIM) term, each term times its own upstream scalar (read on the device)
    return egs_launch_l1_ssim_backward_w(channels, height, width, img, gt, 1.f, -1.f, 0.f, upstream_l1, upstream_ssim, gate, dm_dmu1, dm_dexx, dm_dexy, dL_dimg,
                                         nullptr, nullptr, nullptr, side ? &pa : nullptr, (hipStream_t)stream);
}

int egs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream) {
    (void)projmatrix;
    if (P < 0) return EGS_ERR_ARG;
    if (P == 0) return 0;
    if (!means3D || !viewmatrix || !present) return EGS_ERR_ARG;
    EGS_TRY(egs_launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream));
    return 0;
}

}  // extern "C"
