// egs_common.h -- shared declarations of the gfx950 rasterizer library (internal; the public C ABI
// is include/egs_raster.h).  Wave = 64 lanes everywhere; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/egs_raster.h"

#define EGS_WAVE 64
#define EGS_SPLAT_REC_F4 3          // float4 per packed splat record (48 B)

// Packed per-Gaussian record produced by preprocess and gathered by the blend kernels:
//   f4[0] = (x, y, qa, qb)   f4[1] = (qc, opacity, red, green)
//   f4[2] = (blue, depth, bits(bbox_x = x0 | x1<<16 | hot bits), bits(bbox_y = y0 | y1<<16 | hot bit))
//           box coordinates are 15 bits (image sides <= 32767); bits 15 and 31 of bbox_x and bit 15 of bbox_y hold the HOT code
// (qa, qb, qc) = (-0.5 A, -B, -0.5 C) * log2(e): the conic pre-scaled so that
//   log2(G) = qa dx^2 + qb dx dy + qc dy^2   feeds v_exp_f32 directly (5 VALU instead of 8).
// The blend loop reads f4[0], f4[1] and the first half of f4[2] (ds_read_b128 x2 + ds_read_b64).
// bbox = conservative pixel bounding box of {alpha >= 1/255}; x0 > x1 marks "never contributes".
#define EGS_LOG2E 1.4426950408889634f
#define EGS_LN2   0.6931471805599453f

// Per-Gaussian accumulator written by the blend backward (one 48-B line per Gaussian).  With gd = dL/dalpha * G per
// (pixel, splat) pair and d = splat centre - pixel:
//   [0]=sum gd dx  [1]=sum gd dy  [2]=sum gd dx^2  [3]=sum gd dx dy  [4]=sum gd dy^2  [5]=sum gd = dL/dopacity
//   [6..8]=dL/dcolor rgb  [9]=dL/ddepth
// k_preprocess_backward converts the five moments into dL/dmean2D and dL/dconic with the Gaussian's own opacity and conic, the
// conic -> cov2D step in float64 (preprocess.hip: the one place where float32 lost a recorded parity case).  Slots 10 and 11 of the
// 48-byte line are spare.  (Round 6 tried "FAR" splats here -- a per-pixel cancelled form of dL/dmean2D for splats whose pixels all lie
// far from the centre, accumulated into the two spare slots: 7 more vector instructions per visit of the backward blend and no change
// in any gradient of the off-screen family, whose loss turned out to sit in the chain behind the sums; experiments/far_mean2d.patch.)
#define EGS_GRAD_STRIDE 12

// Hot Gaussians.  A splat whose alpha >= 1/255 box covers EGS_HOT_MIN_TILES tiles or more is visited by hundreds to thousands of
// quadrant-waves, all of which add into ITS ONE accumulator line -- and the waves of every tile reach it at about the same point of
// their lists.  Atomics on one line retire one after the other (~4 ns each), and the queue behind a hot line holds up everybody's:
// on a trained scene (a few hundred screen-filling splats) that was 52 of the backward blend's 217 us (profiles/r4_trained_scene.md).
// Such a Gaussian therefore gets EGS_HOT_REPLICAS accumulator lines, the wave picks one by its tile, and k_preprocess_backward adds
// them up.  Which lines: k_preprocess ranks the hot Gaussians of its 256-Gaussian workgroup (no global counter) and leaves
// code = rank + 1 (1 .. EGS_HOT_PER_BLOCK; 0 = not hot, also for those beyond the workgroup's budget) in the record's spare bits
// (and in bits 3-5 of the Gaussian's `clamped` byte, where k_preprocess_backward finds it without another load);
// the lines of Gaussian i with code c are hot_acc[r * slots + (i / 256) * EGS_HOT_PER_BLOCK + c - 1] (lines of EGS_HOT_LINE floats, slots =
// ceil(P / 256) * EGS_HOT_PER_BLOCK), r = the XCD the blending workgroup runs on, right behind the P regular lines in the backward's
// scratch.  Replica-major: a Gaussian's replicas lie hundreds of KB apart, i.e. in different memory channels -- side by side they
// relieved the line but not the channel that serves it.  Sums only move between lines: every gradient is the same sum of the same terms.
#ifndef EGS_HOT_MIN_TILES
#define EGS_HOT_MIN_TILES 256u
#endif
#define EGS_HOT_PER_BLOCK 7u
#define EGS_HOT_REPLICAS 8u
#define EGS_HOT_LINE 16u            // floats between replica lines (64 B: two per 128-byte line)
#define EGS_BOX_MASK 0x7fffu
__host__ __device__ __forceinline__ uint32_t egs_hot_code(uint32_t bbx, uint32_t bby) {
    return ((bbx >> 15) & 1u) | ((bbx >> 30) & 2u) | ((bby >> 13) & 4u);
}
__host__ __device__ __forceinline__ void egs_hot_code_set(uint32_t& bbx, uint32_t& bby, uint32_t code) {
    bbx |= ((code & 1u) << 15) | ((code & 2u) << 30); bby |= (code & 4u) << 13;
}
__host__ __device__ static inline size_t egs_hot_slots(size_t P) { return ((P + 255) / 256) * EGS_HOT_PER_BLOCK; }
static inline size_t egs_hot_floats(size_t P) { return egs_hot_slots(P) * EGS_HOT_REPLICAS * EGS_HOT_LINE; }
static inline size_t egs_acc_floats(size_t P) { return P * EGS_GRAD_STRIDE + egs_hot_floats(P); }    // regular lines, then the hot replicas

struct EgsGeomPtrs {
    float4* rec; uint2* rect; uint32_t* offsets; uint8_t* clamped; uint8_t* visible; uint32_t* scan_scratch; uint64_t* total;
    uint32_t* block_hot;    // [ceil(P / 256)] hot Gaussians of every 256-Gaussian workgroup of k_preprocess (<= EGS_HOT_PER_BLOCK): which replica lines are in use
};
#define EGS_CLAMP_MASK 7u           // `clamped` byte: bits 0-2 colour channel clamped at zero, bits 3-5 the HOT code (below)
struct EgsBinPtrs {
    uint64_t* pairs;        // [R] (depth<<32 | index), bucketed by tile
    uint64_t* scratch;      // [R] ping-pong space for oversize buckets
    uint32_t* point_list;   // [R] sorted Gaussian indices
    uint32_t* table;        // [n_tiles][table_stride] per-(tile, bucketing workgroup) instance counts, scanned in place (columns >= bin_blocks unused)
    uint32_t* chunk_sum;    // [EGS_BIN_GROUPS][chunks] sums of the table's 2048-entry scan chunks, accumulated by k_bin_count; ZERO before it
    uint32_t* flag;         // [32] scratch words
    uint64_t* total;        // [1] number of instances found by the scan (== R)
    // Fused count pass (preprocess.hip k_preprocess_count): chunk_sum then points into the caller's PERSISTENT placement buffer, whose sums
    // region is zero between frames -- the first per-tile sort launch of a frame clears these words again (NULL: nothing to clear)
    uint32_t* zero_after; uint32_t zero_after_n;
};
#define EGS_BIN_GROUPS 8            // partial accumulators per scan chunk (a same-address atomic chain is bin_blocks / 8 long)
static inline uint32_t egs_table_stride(uint32_t bin_blocks) { uint32_t s = 4; while (s < bin_blocks) s <<= 1; return s; }   // power of two <= 2048
static inline size_t egs_table_chunks(size_t n_tiles, uint32_t stride) { const size_t rpc = 2048 / stride; return (n_tiles + rpc - 1) / rpc; }
struct EgsImgPtrs { uint2* ranges; float* final_T; uint32_t* n_contrib; uint32_t* quad_work; uint32_t* tile_order; uint32_t* quad_pairs;
                    uint32_t* fwd_cost; uint32_t* fwd_order; };

static inline size_t egs_align(size_t x) { return (x + 255) & ~(size_t)255; }

// binning geometry (binning.hip)
#ifdef EGS_BIN_GPB_OVERRIDE
#define EGS_BIN_GPB EGS_BIN_GPB_OVERRIDE
#else
#define EGS_BIN_GPB 1024                                       // Gaussians per bucketing workgroup at config C (see egs_bin_gpb: sized by P)
#endif
#ifdef EGS_BIN_THREADS_OVERRIDE
#define EGS_BIN_THREADS EGS_BIN_THREADS_OVERRIDE
#else
#define EGS_BIN_THREADS 1024                                   // 16 waves, 64 Gaussians each
#endif
#define EGS_MAX_TILES 36864                                    // one 4-byte LDS counter per tile must fit in 160 KiB
uint32_t egs_bin_blocks(int P);
int egs_bin_gpb(int P);                                          // Gaussians per bucketing workgroup for a model of P Gaussians
struct EgsBinGeometry { uint32_t nblocks, stride, n_chunks; int gpr, cull, use_map, n_tiles, gx; size_t lds; };
EgsBinGeometry egs_bin_geometry(int P, int W, int H, int cull);  // launch geometry of the bucketing kernels (binning.hip); cull: tile culling wanted (EGS_CALL_KEEP_ALL_INSTANCES clear)
#define EGS_SCAN_THREADS 256
#define EGS_SCAN_ITEMS 8
#define EGS_SCAN_EPB (EGS_SCAN_THREADS * EGS_SCAN_ITEMS)      // elements per block

int egs_key_bits_for_tiles(int n_tiles);
size_t egs_scan_scratch_elems(size_t n);

// ---- Adam, shared by the stand-alone optimizer launch (adam.hip) and the optimizer fused into the preprocess backward ----------
// torch.optim.Adam with weight_decay = 0, amsgrad = False, maximize = False; ss = lr / (1 - b1^t), ib = 1 / sqrt(1 - b2^t)
__device__ __forceinline__ void egs_adam1(float& p, float g, float& m, float& v, float b1, float b2, float eps, float ss, float ib) {
    m = fmaf(1.f - b1, g - m, m);
    v = fmaf(1.f - b2, g * g, b2 * v);
    p -= ss * (m / (sqrtf(v) * ib + eps));
}
// Leaves a fused optimizer can own (include/egs_raster.h: EGS_SINK_*), in the order their rows are staged: floats per row and
// the float4 task at which each leaf's share of a 256-Gaussian workgroup starts (64 * row floats tasks each).
#define EGS_SINK_LEAVES 6          // the sixth (EGS_SINK_SH_REST) is stepped by k_sh16_backward only
#define EGS_SINK_PP_LEAVES 5       // leaves k_preprocess_backward can step
#define EGS_SINK_TASKS 896
struct EgsSinkLeaf { float* p; float* m; float* v; };
struct EgsSink {                       // kernel argument of k_preprocess_backward<true>
    EgsSinkLeaf leaf[EGS_SINK_LEAVES]; const float* coef; const int32_t* active_rows; const uint32_t* skip; float b1, b2, eps;
};
struct EgsAdamTick {                   // one thread per backward: advances state["step"] and derives the two coefficients of every leaf
    float* step[EGS_SINK_LEAVES]; const float* lr[EGS_SINK_LEAVES]; float* coef; const uint32_t* skip; float b1, b2;
};
// Called by threads 0 .. 2 * EGS_SINK_LEAVES - 1 of ONE workgroup: thread 2 l derives the step size of leaf l (and advances its
// state["step"]), thread 2 l + 1 the second-moment correction -- the double-precision pow() calls run side by side (one thread doing
// all ten made its launch 4 us longer).
__device__ __forceinline__ void egs_adam_tick(const EgsAdamTick& t, unsigned lane) {
    if (lane >= 2 * EGS_SINK_LEAVES) return;
    if (t.skip && *t.skip) return;                                   // overflowed frame: no step is taken, none is counted
#pragma unroll
    for (int l = 0; l < EGS_SINK_LEAVES; l++) {
        if ((lane >> 1) != (unsigned)l || !t.step[l]) continue;
        // the same double-precision expressions as k_adam / the host use: fused and stand-alone steps are bit-identical
        const double st = (double)t.step[l][0] + 1.0;                // (both threads of the pair read the step before either writes: see below)
        const double bc = 1.0 - pow((double)((lane & 1) ? t.b2 : t.b1), st);
        const float out = (lane & 1) ? (float)(1.0 / sqrt(bc)) : (float)((double)t.lr[l][0] / bc);
        t.coef[2 * l + (lane & 1)] = out;
        __builtin_amdgcn_wave_barrier();                             // the pair sits in one wave: its loads of step[l] precede this store in program order
        if (!(lane & 1)) t.step[l][0] = (float)st;
    }
}

// The `fine_all` call shape inside the preprocess kernels (include/egs_raster.h egs_object_rotation): rows with sel[i] != 0 (all rows if
// sel == NULL) get cov3D = (M R S)(M R S)^T.  M == NULL: nothing is moved.
struct EgsObjRot { const float* M; const uint8_t* sel; float mult; const float* mult_dev; };

// ---- launchers (host side, one per translation unit) -------------------------------------------
struct EgsCamera {
    const float* view; const float* proj; const float* campos;
    int W, H; float tanfovx, tanfovy;
};
// `act`: activation flags of the raw-parameter mode (include/egs_raster.h: EGS_ACT_*)
// zero_words / zero_n: an unrelated region this launch also clears (the bucketing's chunk sums of the same frame; may be NULL)
hipError_t egs_launch_preprocess(int P, int D, int M, const float* means3D, const float* shs, const float* colors,
                                 const float* opac, const float* scales, float mod, const float* rots, int act,
                                 const float* cov3D, EgsCamera cam, int32_t* radii, EgsGeomPtrs g, uint32_t* zero_words, size_t zero_n,
                                 const int32_t* active_count, const EgsImgPtrs* place /*NULL, or: also order im.fwd_cost into im.fwd_order*/,
                                 EgsObjRot rot, hipStream_t s);
hipError_t egs_launch_preprocess_backward(int P, int D, int M, const float* means3D, const float* shs,
                                          const float* scales, float mod, const float* rots, const float* cov3D, int act,
                                          EgsCamera cam, const int32_t* radii, EgsGeomPtrs g, const float* grad_acc,
                                          int colors_given, float* dmeans2D, float* dcolors, float* dopac,
                                          float* dmeans3D, float* dcov3D, float* dsh, float* dscales, float* drots,
                                          float* stat_grad_accum, float* stat_denom, float* stat_max_radii, const uint32_t* skip_flag,
                                          const EgsSink* sink /*NULL: gradients only*/, EgsObjRot rot, hipStream_t s);
// Spherical harmonics as separate launches (M > 1 coefficients, or DC / rest given as two arrays: sh_rest != NULL).  The
// preprocess launchers are then called with shs = NULL: the forward leaves the record's colour open, the backward leaves
// dL/dSH and the view-direction part of dL/dmean3D to egs_launch_sh_backward (which must run after it).
hipError_t egs_launch_sh_forward(int P, int D, int M, const float* means3D, const float* sh_a, const float* sh_rest, EgsCamera cam,
                                 EgsGeomPtrs g, hipStream_t s);
// sink (may be NULL; only with egs_sh_backward_can_sink): the Adam step of features_dc / features_rest / positions taken by this launch
hipError_t egs_launch_sh_backward(int P, int D, int M, const float* means3D, const float* sh_a, const float* sh_rest, EgsCamera cam,
                                  const int32_t* radii, EgsGeomPtrs g, const float* dcolors, float* dsh_a, float* dsh_rest,
                                  float* dmeans3D, const EgsSink* sink, hipStream_t s);
bool egs_sh_backward_can_sink(int M, const float* sh_a, const float* sh_rest);      // the M = 16 split-array kernel will run
hipError_t egs_launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s);

// exclusive/inclusive u32 scan of n elements; scratch holds egs_scan_scratch_elems(n) u32; optionally
// writes the grand total (u64) to *total.
hipError_t egs_launch_scan_u32(const uint32_t* in, uint32_t* out, size_t n, int inclusive, uint32_t* scratch,
                               uint64_t* total, hipStream_t s);
// sums_zeroed: b.chunk_sum was cleared by this frame's preprocess launch (else a zero-fill launch comes first)
// What the per-tile sort needs (tile_sort.h): kernel argument of k_tile_sort, and of the forward blend when it sorts its tiles itself
struct EgsSortArgs {
    int n_tiles; uint32_t stride; const uint32_t* table_scanned; const uint64_t* total; uint64_t* running_max; uint32_t* overflow_flag; int solo;
    uint32_t R /* capacity */; int index_passes; uint64_t* pairs; uint64_t* scratch; uint32_t* point_list; uint2* ranges;
    uint32_t* zero_after; uint32_t zero_after_n; int rank_atomic /* launcher -> launcher: the LDS lane-order property holds (wave_digit_rank) */;
};
// counted: b.table and b.chunk_sum were filled by egs_launch_preprocess_count (the count pass is not launched)
// sort_in_blend (may be NULL): see binning.hip
hipError_t egs_launch_binning(int P, int64_t R, int W, int H, EgsGeomPtrs g, EgsBinPtrs b, EgsImgPtrs im,
                              uint64_t* running_max, uint32_t* overflow_flag, int sums_zeroed, int counted, EgsSortArgs* sort_in_blend, hipStream_t s, int call_flags /* EGS_CALL_* */);
// k_preprocess + the count pass of the tile bucketing in ONE launch (preprocess.hip); b.chunk_sum must be ZERO (see EgsBinPtrs).
// -> false when the launch geometry does not allow it (fewer than four groups per round: very large images)
bool egs_can_fuse_count(int P, int W, int H, int cull);
hipError_t egs_launch_preprocess_count(int P, int D, int M, const float* means3D, const float* shs, const float* colors,
                                       const float* opac, const float* scales, float mod, const float* rots, int act,
                                       const float* cov3D, EgsCamera cam, int32_t* radii, EgsGeomPtrs g, EgsBinPtrs b,
                                       const int32_t* active_count, const EgsImgPtrs* place, EgsObjRot rot, int cull, hipStream_t s);
// placed: im.fwd_order holds this frame's placement (the preprocess launch carried the ordering job); else the static mapping
// sort (may be NULL, or table_scanned == NULL in it): the blend sorts every tile's bucket itself first (egs_launch_binning made no sort launch)
hipError_t egs_launch_render_forward(int W, int H, const float* bg, EgsGeomPtrs g, const uint32_t* point_list,
                                     EgsImgPtrs im, float* out_color, float* out_depth, float* out_alpha, int placed,
                                     const EgsSortArgs* sort, hipStream_t s);
// The backward blend, and what it needs in place first (tile order, cleared accumulator; `tick`, may be NULL: the per-step bookkeeping
// of an optimizer fused into this backward) as a launch of its own -- or carried by egs_l1_ssim_backward_ex (backward_prologue.h).
// block_hot (may be NULL: every replica line is cleared): the per-workgroup hot counts of the frame's preprocess -- only the replica lines in use are cleared
hipError_t egs_launch_backward_prologue(int P, int W, int H, EgsImgPtrs im, float* grad_acc, const uint32_t* block_hot, const EgsAdamTick* tick, hipStream_t s);
// The image loss's gradient computed inside the backward blend (render_bwd.hip k_render_backward<1, true>): what the loss forward left.
// w_l1_n, w_ssim_n: the weights of mean|x - y| and of (1 - mean SSIM) before the division by the element count (loss.hip).
// fin_*: the loss VALUE the forward deferred (egs_l1_ssim_forward with loss == NULL) is assembled by one wave of the blend launch.
struct EgsLossGradHost { const float* img; const float* gt; const float* dm_dmu1; const float* dm_dexx; const float* dm_dexy; const float* gate;
                         const float* upstream; const float* upstream_ssim; float w_l1_n, w_ssim_n;
                         const float* fin_partial; size_t fin_n; float fin_lambda; float* fin_loss; float* fin_running; };
#ifdef EGS_LG_CHECK
extern EgsLossGradHost egs_debug_lossgrad;
#endif
hipError_t egs_launch_loss_finish(const EgsLossGradHost& lg, int W, int H, hipStream_t s);     // the deferred value alone (a frame with no instance)
// lg (may be NULL): the blend computes dL/dcolour itself (k_render_backward<1, true>); dL_dcolor, dL_ddepth, dL_dalpha are then not read
hipError_t egs_launch_render_backward(int P, int W, int H, const float* bg, EgsGeomPtrs g, const uint32_t* point_list,
                                      EgsImgPtrs im, const float* dL_dcolor, const float* dL_ddepth,
                                      const float* dL_dalpha, float* grad_acc, int colors_only, const EgsLossGradHost* lg, hipStream_t s);
// colors_only: the blend left only the colour sums (k_render_backward<0>); this turns them into dL/dcolors_precomp [P,3]
hipError_t egs_launch_colors_from_acc(int P, const float* grad_acc, const uint8_t* clamped, const int32_t* radii, float* dcolors, hipStream_t s);
hipError_t egs_launch_adam_tick(const EgsAdamTick& tick, hipStream_t s);       // the same bookkeeping as a launch of its own (frames with no instance)

// Zero-fill by a kernel.  hipMemsetAsync is avoided inside the per-step chain: captured into a hipGraph it becomes a memset
// node, and on ROCm 7.2 replays of the training-step graph intermittently saw stale accumulator contents with it.
hipError_t egs_launch_zero_f4(float4* p, size_t n4, hipStream_t s);
hipError_t egs_launch_zero_u32(uint32_t* p, size_t n, hipStream_t s);

// optional stage timing (api.hip); no-ops unless egs_profile_begin() was called
void egs_prof_start(int stage, hipStream_t s);
void egs_prof_stop(int stage, hipStream_t s);
