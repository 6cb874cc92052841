/*
 * egs_raster.h -- C ABI of libegs_raster.so, the MI355X (gfx950) differentiable 3D Gaussian
 * tile rasterizer behind EgoGaussian's gaussian_renderer.render() path.
 *
 * What each entry point replaces.  The reference binds the CUDA extension
 * `diff_gaussian_rasterization._C` (an un-vendored submodule: ashawkey/diff-gaussian-rasterization
 * @ 8829d14f, /root/reference/README.md:26, /root/reference/.gitmodules:1-3) through the Python
 * surface imported at
 *     /root/reference/gaussian_renderer/__init__.py:14      (GaussianRasterizationSettings, GaussianRasterizer)
 *     /root/reference/gaussian_renderer/render_helper.py:3
 * and invoked at
 *     /root/reference/gaussian_renderer/__init__.py:90-98   (training call: shs + cov3D_precomp)
 *     /root/reference/gaussian_renderer/render_helper.py:61-63 (label call: colors_precomp + scales/rotations)
 *   egs_forward_geometry + egs_forward_render  <->  _C.rasterize_gaussians          (forward)
 *   egs_backward                                <->  _C.rasterize_gaussians_backward (backward)
 *   egs_mark_visible                            <->  _C.mark_visible
 *   egs_*_bytes / egs_*_layout                  <->  the three resizable byte buffers
 *                                                    (geomBuffer, binningBuffer, imgBuffer) the
 *                                                    upstream op allocates through callbacks
 * The forward is split in two because the number of (Gaussian, tile) instances R is data dependent:
 * egs_forward_geometry returns R on the host, the caller sizes the binning buffer, then calls
 * egs_forward_render.
 *
 * Conventions
 *   - All pointers are DEVICE pointers unless marked HOST.  fp32, contiguous, row-major.
 *   - The library never allocates or frees device memory; every buffer is owned by the caller and must
 *     stay alive until the work enqueued on `stream` has completed.
 *   - The opaque geometry / binning / image buffers must start at 256-byte-aligned addresses (hipMalloc and the PyTorch allocator
 *     give at least that); a misaligned one is refused with EGS_ERR_ARG.  Their layouts place every array at a 256-byte offset.
 *   - Process-wide state, all of it: (1) the optional profiling pool (egs_profile_begin / egs_profile_end), off by
 *     default; (2) the result of the one-time device check behind the tile sort's ranker (an atomic int: a property of
 *     the hardware, the same for every caller); (3) one hipEvent per calling thread inside egs_forward (thread_local).
 *     Nothing else is kept between calls -- no switches, no registry of buffers (ABI 6: what were process-wide debug
 *     switches through ABI 5 are bits of the call's own flags word, EGS_CALL_* below; a placement buffer is initialised
 *     by its owner, egs_placement_init).  Concurrent calls on different streams / threads with different settings are safe.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it.  egs_forward_geometry is
 *     the only call that waits on the stream (one 8-byte device->host read of R).
 *   - Matrices use the reference's row-vector layout (/root/reference/scene/cameras.py:67-69):
 *     viewmatrix = W2V^T, projmatrix = W2V^T * P^T, 16 floats each.
 *   - Optional inputs are NULL when absent: exactly one of {shs, colors_precomp} and exactly one of
 *     {cov3D_precomp, (scales, rotations)} must be non-NULL.
 *   - Return value: 0 on success; EGS_ERR_* (negative) for argument errors; a positive hipError_t if
 *     the HIP runtime reported one.  Nothing throws across this boundary.
 *   - The last argument of every compute entry point (`debug`; `flags` on egs_forward_enqueue) is a set of EGS_CALL_* bits.  Bit 0
 *     (EGS_CALL_SYNC -- what debug != 0 meant through ABI 5) synchronises the stream and checks for errors after every kernel.
 */
#ifndef EGS_RASTER_H
#define EGS_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGS_ABI_VERSION 6          /* 6: per-call flags (EGS_CALL_*) instead of the process-wide egs_debug_set_* switches, egs_forward_enqueue takes a flags word,
                                         egs_placement_init is the ONLY thing that clears a placement buffer's sums region (no registry of addresses);
                                         5: the placement buffer grew a sums region (egs_placement_bytes) that must be ZERO when the buffer is first handed over:
                                         egs_placement_init;  4: grad_mask on the backwards, egs_l1_ssim_pair_*, out_depth / out_alpha may both be NULL on every
                                         forward (colour only) */
#define EGS_TILE 16                 /* tile edge in pixels; part of the parity contract */

/* Per-call flags (ABI 6).  None of them changes an output value: colour, depth, alpha, radii, final transmittance and every gradient are
 * bit-identical whatever the bits; they select which launches produce them (and, for KEEP_ALL_INSTANCES, what the internal lists hold). */
#define EGS_CALL_SYNC               0x01   /* synchronise the stream and check for errors after every kernel (debugging) */
#define EGS_CALL_KEEP_ALL_INSTANCES 0x02   /* no tile culling.  By default an instance (splat, tile) of the reference's 3-sigma rectangle is dropped
                                              when no pixel of the tile can receive alpha >= 1/255 from the splat (exact, conservative ellipse-vs-tile
                                              test): the reference skips such an instance at every pixel, so only the internal lists (binning buffer,
                                              per-pixel contributor positions) get shorter; `num_rendered` stays the reference's count.  With this bit
                                              every instance is kept and the lists are comparable bit for bit with the reference algorithm's (parity
                                              tests).  Give the SAME bit to the forward and to nothing else: the backward reads the lists as they are. */
#define EGS_CALL_SEPARATE_COUNT     0x04   /* the count pass of the tile bucketing as a launch of its own even when a placement buffer would let it
                                              ride in the preprocess launch (A/B measurements, tests) */
#define EGS_CALL_SEPARATE_SORT      0x08   /* the per-tile sort as launch(es) of its own instead of inside the forward blend's launch */
#define EGS_CALL_BALLOT_RANK        0x10   /* the per-tile sort ranks keys with the ballot-based fallback instead of the LDS atomic whose lane-order
                                              behaviour is verified on the device once per process (test hook: covers the fallback) */
#define EGS_MAX_SH_DEGREE 3

#define EGS_ERR_ARG        (-1)     /* NULL / inconsistent arguments */
#define EGS_ERR_MODE       (-2)     /* colour or covariance mode not "exactly one of" */
#define EGS_ERR_RANGE      (-3)     /* size outside supported range (image side > 32767 px or > 36864 tiles, R >= 2^31, degree > 3) */
#define EGS_ERR_NO_DEVICE  (-4)     /* no HIP device / wrong architecture */
#define EGS_RETRY_LARGER   (-100)   /* egs_forward only: capacity guess too small, nothing rendered; *num_rendered holds R */

int         egs_abi_version(void);
const char* egs_source_hash(void);      /* hash of the sources this library was built from (csrc/Makefile); "unknown" if built by hand */
const char* egs_error_string(int code);
/* Name of the device the current HIP context runs on and its gcnArchName; returns 0 or an error. */
int         egs_device_info(char* name, int name_len, char* arch, int arch_len, int* compute_units);

/* ---- buffer sizes (bytes) -------------------------------------------------------------------- */
size_t egs_geom_bytes(int P);
size_t egs_binning_bytes(int P, int64_t R, int width, int height);
size_t egs_image_bytes(int width, int height);
size_t egs_backward_scratch_bytes(int P);
/* words of a tile-order array (image layout `tile_order`; second part of the placement buffer, behind 4 words per tile): one per
 * workgroup of a blend launch -- 8 XCD bands of equal slot counts, 0xffffffff = padding workgroup */
int    egs_order_words(int width, int height);

/* ---- buffer layouts, for tests and tools: byte offsets of the named sub-arrays ----------------- */
typedef struct egs_geom_layout {
    size_t rec;            /* float4[P][3]  packed splat record (csrc/egs_common.h): (x, y, qa, qb | qc, opacity, red, green |
                              blue, depth, bits(bbox_x = x0 | x1<<16), bits(bbox_y = y0 | y1<<16)) with (qa, qb, qc) =
                              (-conA/2, -conB, -conC/2) * log2(e), the conic pre-scaled for v_exp_f32 */
    size_t rect;           /* uint32[P][2]  tile rect: (x0 | x1<<16, y0 | y1<<16) */
    size_t offsets;        /* uint32[P]     tiles touched by each Gaussian (0 = culled) */
    size_t clamped;        /* uint8[P]      bit c set <=> colour channel c was clamped at 0 */
    size_t visible;        /* uint8[P]      1 <=> radii > 0 (the renderer's visibility_filter without a compare kernel) */
    size_t scan_scratch;   /* uint32[ceil(P/256)] per-workgroup instance counts (their sum is R) */
    size_t total;          /* reserved */
} egs_geom_layout;
typedef struct egs_binning_layout {
    size_t point_list;     /* uint32[R]  Gaussian indices ordered by (tile, depth bits, index); ALWAYS at offset 0 */
    size_t pairs;          /* uint64[R]  (float bits of depth << 32 | Gaussian index), bucketed by tile */
    size_t scratch;        /* uint64[R]  ping-pong space for buckets too large for the in-register sort */
    size_t table;          /* uint32[tiles][table_stride] per-(tile, bucketing workgroup) counts, exclusive-scanned in place; columns >= bin_blocks unused */
    size_t spine;          /* uint32[..] sums of the table's 2048-entry scan chunks (8 partial accumulators each), scratch words */
    size_t total;          /* uint64[1]  instance count found by the bucketing scan (== R; may exceed a speculative capacity) */
    int    bin_blocks;     /* workgroups of the bucketing kernels: ceil(P / gpb), gpb = 1024 * ceil(P / 524288) */
    int    key_bits;       /* significant bits of the canonical (tile<<32 | depth) key: 32 + bits(tile count) */
    int    index_passes;   /* 9-bit radix passes on the Gaussian index inside the per-tile sort, run only for tiles with depth ties
                             (+ up to 4 on depth: ceil(bits(zmax - zmin of the tile) / 9)) */
    int    table_stride;   /* row length of `table`: the power of two >= max(bin_blocks, 4) */
} egs_binning_layout;
typedef struct egs_image_layout {
    size_t ranges;         /* uint32[tiles][2] */
    size_t final_T;        /* float[H*W] */
    size_t n_contrib;      /* uint32[H*W] */
    size_t quad_work;      /* uint32[tiles][4] estimated backward cost of every 8x8 quadrant (blended splats and list batches), written by the forward */
    size_t tile_order;     /* uint32[8*ceil(tiles/8)] tile handled by each workgroup of the backward blend */
    size_t quad_pairs;     /* uint32[tiles][4] (pixel, splat) pairs every 8x8 quadrant blended (alpha >= 1/255, before saturation) and,
                              in the upper array uint32[tiles][4] right after it, the (wave, splat) visits it made -- measurement only
                              (bench.py: pair throughput Q/s) */
} egs_image_layout;
int egs_get_geom_layout(int P, egs_geom_layout* out);
int egs_get_binning_layout(int P, int64_t R, int width, int height, egs_binning_layout* out);
int egs_get_image_layout(int width, int height, egs_image_layout* out);

/* ---- forward, part 1: per-Gaussian geometry + instance count  (upstream: preprocess + InclusiveSum) ------ */
/* Raw-parameter mode.  The reference activates its parameters with three PyTorch ops before every render
 * (/root/reference/scene/gaussian_model.py:36-44: scaling exp, rotation normalize, opacity sigmoid).  With these flags the
 * forward applies them inside the preprocess kernel and the backward returns the gradients w.r.t. the RAW tensors:
 *   EGS_ACT_LOG_SCALES     `scales` holds log-scales            (only with scales + rotations, not with cov3D_precomp)
 *   EGS_ACT_RAW_QUATS      `rotations` is not normalised        (likewise)
 *   EGS_ACT_LOGIT_OPACITY  `opacities` holds logits
 * Pass the SAME flags to the forward and to egs_backward. */
/* Split spherical harmonics.  The reference keeps the colour coefficients as two parameters, _features_dc [P,1,3] and
 * _features_rest [P,M-1,3], and concatenates them before every render (/root/reference/scene/gaussian_model.py:157-160
 * get_features); autograd splits the gradient again.  With shs_rest != NULL, `shs` is the DC block [P,1,3] and `shs_rest` the
 * remaining [P,M-1,3] (sh_coeffs is still M >= 2); egs_backward then writes dL_dsh [P,1,3] and dL_dsh_rest [P,M-1,3]. */
#define EGS_ACT_LOG_SCALES 1
#define EGS_ACT_RAW_QUATS 2
#define EGS_ACT_LOGIT_OPACITY 4
/* Object rotation inside the rasterizer (ABI 2 addition).  The `fine_all` trainer renders with the covariance of the object's Gaussians
 * rotated by the accumulated object motion: render(..., rot_cov=True, accum_R, which_object)
 * (/root/reference/trainers/fine_all.py:88-93, /root/reference/scene/gaussian_model.py:46-63: L = R S, L <- M L for the selected rows,
 * Sigma = L L^T), which upstream can only be given as cov3D_precomp.  With `rot` the forward builds that covariance itself from
 * scales + rotations (any activation flags) and the backward chains through M: no [P,6] covariance in HBM either way, no producer
 * launches, and the raw parameters stay eligible for egs_backward_adam.  M is a constant of the loss (no gradient is produced for
 * it; a trainable object rotation goes through cov3D_precomp / egs_cov3d_*).  Same arithmetic, operation for operation, as
 * egs_cov3d_forward / egs_cov3d_backward.  NULL = no rotation.  Pass the same to the forward and to egs_backward_adam. */
typedef struct egs_object_rotation {
    const float* M9;                  /* device float[9], row-major 3x3 */
    const uint8_t* selected;          /* device uint8[P]: rows to rotate; NULL = every row */
    float row0_grad_mult;             /* gradient multiplier of row 0 when it is rotated: the reference's [N,1]-index quirk
                                         (egogaussian_amd/covariance.py), 1 = none */
    const float* row0_grad_mult_dev;  /* device float[1] used instead of row0_grad_mult, or NULL */
} egs_object_rotation;
int egs_forward_geometry(
    int P, int sh_degree, int sh_coeffs /* M: coefficients per channel in `shs` */,
    const float* means3D /*[P,3]*/, const float* shs /*[P,M,3] or NULL*/, const float* shs_rest /*see below; normally NULL*/,
    const float* colors_precomp /*[P,3] or NULL*/,
    const float* opacities /*[P]*/, const float* scales /*[P,3] or NULL*/, float scale_modifier,
    const float* rotations /*[P,4] or NULL*/, const float* cov3D_precomp /*[P,6] or NULL*/, int activation_flags /*EGS_ACT_*, 0 = none*/,
    const float* viewmatrix /*[16]*/, const float* projmatrix /*[16]*/, const float* campos /*[3]*/,
    int width, int height, float tan_fovx, float tan_fovy, int prefiltered,
    int32_t* radii /*[P] out*/, void* geom_buffer, int64_t* num_rendered /*HOST out: R*/,
    const int32_t* active_count /*device int32[1] or NULL; see "capacity-sized models" below*/,
    const egs_object_rotation* rot /*HOST or NULL*/, void* stream, int debug);

/* ---- forward, part 2: bucket instances by tile, sort each tile by (depth, index), blend
 *      (upstream: duplicateWithKeys, SortPairs, identifyTileRanges, render) ------------------------ */
/* R here (and in egs_backward) is the instance count the binning buffer was laid out for: the exact count from
 * egs_forward_geometry, or the `capacity` given to egs_forward. */
int egs_forward_render(
    int P, int64_t R, const float* background /*[3]*/, int width, int height,
    const void* geom_buffer, void* binning_buffer, void* image_buffer,
    float* out_color /*[3,H,W]*/, float* out_depth /*[1,H,W]*/, float* out_alpha /*[1,H,W]*/,
    void* stream, int debug);

/* Capacity-sized models (ABI 2).  A trainer that densifies / prunes every 100 iterations
 * (/root/reference/scene/gaussian_model.py:565-586,678-709) can keep its arrays at a fixed CAPACITY of P rows and the number
 * of live Gaussians in a device word: with active_count != NULL rows i >= *active_count are treated as culled (radii 0, no
 * instances, zero gradients), whatever their contents.  P, every launch size and every buffer layout then stay fixed while
 * the model grows and shrinks, so a hipGraph captured around the step survives densification.  NULL: all P rows are live.
 *
 * Overflow word (ABI 2).  egs_forward_enqueue cannot report a too-small `capacity` to the host; with overflow_flag != NULL
 * (device uint32[2]) the chain WRITES [0] = 1 when this frame needed more than `capacity` instances (its image is then
 * clipped and invalid), else 0, and [1] = the number of instances the frame bucketed.  Passed on as `skip_flag` to egs_backward and egs_adam_step_capturable it makes the
 * rest of a captured training step a no-op for that frame: no densification statistics, no parameter, moment or step-count
 * update -- every optimizer step follows a complete render, as in the reference (trainers/train_static.py:110-138). */

/* ---- forward in ONE call, without a GPU bubble.  Same work as egs_forward_geometry + egs_forward_render, but the
 *      binning and blend kernels are enqueued against `capacity` (the caller's guess of R, e.g. 1.25 x the largest R seen
 *      so far; layout = egs_binning_bytes(P, capacity, ...)) before R is known, and the host waits only for the copy of
 *      the per-workgroup instance counts into `pinned_host_counts` (page-locked host memory, >= ceil(P/256) uint32).
 *      Returns 0 with *num_rendered = R when R <= capacity.  Returns EGS_RETRY_LARGER with *num_rendered = R when the guess
 *      was too small (or capacity == 0): nothing valid was rendered; allocate a binning buffer for >= R instances and call
 *      egs_forward_render(P, that_size, ...).  Keeps one hipEvent per calling thread. */
int egs_forward(
    int P, int sh_degree, int sh_coeffs,
    const float* means3D, const float* shs, const float* shs_rest, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp, int activation_flags,
    const float* viewmatrix, const float* projmatrix, const float* campos, const float* background,
    int width, int height, float tan_fovx, float tan_fovy, int prefiltered,
    int32_t* radii /*[P] out*/, void* geom_buffer, int64_t capacity, void* binning_buffer, void* image_buffer,
    float* out_color, float* out_depth, float* out_alpha, uint32_t* pinned_host_counts /*HOST, page-locked*/,
    int64_t* num_rendered /*HOST out*/, const int32_t* active_count /*device int32[1] or NULL*/,
    void* placement /*egs_placement_bytes(width, height) or NULL; see below*/, const egs_object_rotation* rot /*HOST or NULL*/,
    void* stream, int debug);

/* Placement buffer (ABI 2 addition).  The forward blend runs one workgroup per tile, all resident at once, so the launch lasts as
 * long as its busiest SIMD; which tiles share a CU / SIMD is free.  `placement` is caller-owned device memory that PERSISTS between
 * calls: every forward leaves there what its blend spent on each 8x8 quadrant, and the next forward given the same buffer deals its
 * tiles to CUs and its quadrants to SIMDs by those numbers (an ordering job carried by the preprocess launch).  It pays when
 * consecutive calls render similar frames -- a video or an orbit in order, the static buffers of a replayed hipGraph -- and is neutral
 * otherwise.  One buffer per concurrently running forward (it is read and written by the call).  NULL: the static tile mapping.
 * ABI 5: given a placement buffer, the forward also folds the count pass of its tile bucketing into the preprocess launch (one launch and
 * one round of set-up loads less per frame); that pass accumulates per-chunk instance sums in the tail of the buffer, which the chain
 * itself clears again before it ends.  Those words must therefore be ZERO when a forward starts: the owner calls egs_placement_init()
 * (one small launch on `stream`) ONCE per buffer and image size -- after allocating it, and again before using it for another image size
 * (the region's offset follows the tile count) -- and does not use the memory for anything else between calls.  The library keeps no
 * record of buffers (ABI 6): a forward that finds stale sums counts wrongly.  A forward that fails half way clears the words itself before
 * it returns its error.  The cost and tile-order words in front of the sums may hold anything -- they decide when a quadrant is blended,
 * never what is computed.  Results are identical with and without a placement buffer. */
size_t egs_placement_bytes(int width, int height);
int egs_placement_init(void* placement, int width, int height, void* stream);

/* ---- the same chain with NO host wait, for hipGraph capture of a whole training step: everything is only enqueued
 *      (capacity must be > 0).  A frame that needs more than `capacity` instances is invalid (its kernels were clipped to
 *      the capacity) and must be redone with a larger buffer; the caller finds out after synchronising, from either of
 *        pinned_host_counts   optional (NULL: no copy): the per-workgroup rectangle counts, R = egs_sum_counts(P, ..); when
 *                             `overflow_flag` is given as well the buffer holds ceil(P/256) + 2 words and the two overflow words
 *                             ([0] this frame was clipped, [1] instances bucketed) are copied behind the counts at the end of the
 *                             chain -- what an EAGER loop needs to check the frame at its next call instead of waiting now
 *        running_max          optional device uint64: raised by the chain to the number of instances it bucketed whenever
 *                             that is larger, so one word tells whether ANY replay so far overflowed. */
int egs_forward_enqueue(
    int P, int sh_degree, int sh_coeffs,
    const float* means3D, const float* shs, const float* shs_rest, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp, int activation_flags,
    const float* viewmatrix, const float* projmatrix, const float* campos, const float* background,
    int width, int height, float tan_fovx, float tan_fovy, int prefiltered,
    int32_t* radii, void* geom_buffer, int64_t capacity, void* binning_buffer, void* image_buffer,
    float* out_color, float* out_depth, float* out_alpha, uint32_t* pinned_host_counts, uint64_t* running_max,
    const int32_t* active_count /*device int32[1] or NULL*/, uint32_t* overflow_flag /*device uint32[2] out or NULL*/,
    void* placement /*or NULL*/, const egs_object_rotation* rot /*HOST or NULL*/, void* stream, int flags /*EGS_CALL_*; EGS_CALL_SYNC is ignored: nothing may wait*/);
int64_t egs_sum_counts(int P, const uint32_t* pinned_host_counts /*HOST*/);

/* ---- backward  (upstream: render backward + computeCov2D backward + preprocess backward) ------- */
/* grad_mask (ABI 4): which inputs' gradients the caller is going to read -- autograd's ctx.needs_input_grad.  EGS_GRAD_ALL (0) = all of
 * them, the behaviour of ABI <= 3.  The library uses it to skip work; the one combination it acts on today is colors_precomp given and
 * grad_mask == EGS_GRAD_COLORS -- the reference's label call, which detaches every geometric input
 * (/root/reference/gaussian_renderer/render_helper.py:38-54; 30 000 times in /root/reference/trainers/train_static.py:105-109): the
 * blend then accumulates w * dL/dC only (no dL/dalpha recurrence, no moments, no background term), the preprocess backward is not
 * launched, ONLY dL_dcolors is written and every other dL_d* output may be NULL.  Any other mask: everything is computed and every
 * output must be given as before. */
#define EGS_GRAD_ALL        0
#define EGS_GRAD_MEANS3D    1
#define EGS_GRAD_MEANS2D    2
#define EGS_GRAD_SH         4
#define EGS_GRAD_COLORS     8
#define EGS_GRAD_OPACITY    16
#define EGS_GRAD_SCALES     32
#define EGS_GRAD_ROTATIONS  64
#define EGS_GRAD_COV3D      128
#define EGS_GRAD_MASK_BITS  255
int egs_backward(
    int P, int sh_degree, int sh_coeffs, int64_t R,
    const float* background, const float* means3D, const float* shs, const float* shs_rest, const float* colors_precomp,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp, int activation_flags,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    int width, int height, float tan_fovx, float tan_fovy,
    const int32_t* radii, const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
    const float* dL_dout_color /*[3,H,W]*/, const float* dL_dout_depth /*[1,H,W] or NULL*/,
    const float* dL_dout_alpha /*[1,H,W] or NULL*/,
    float* dL_dmeans2D /*[P,3] out, NDC-scaled (x 0.5*W, 0.5*H), z = 0*/,
    float* dL_dcolors /*[P,3] out*/, float* dL_dopacity /*[P] out*/, float* dL_dmeans3D /*[P,3] out*/,
    float* dL_dcov3D /*[P,6] out; may be NULL with scales + rotations*/, float* dL_dsh /*[P,M,3] out or NULL*/,
    float* dL_dsh_rest /*[P,M-1,3] out with shs_rest (dL_dsh is then [P,1,3]), else NULL*/,
    float* dL_dscales /*[P,3] out or NULL*/, float* dL_drotations /*[P,4] out or NULL*/,
    /* Optional (all NULL = off): the trainer's per-iteration densification statistics updated IN PLACE by the same launch that
     * produces dL_dmeans2D -- for Gaussians with radii > 0: grad_accum[i] += |dL_dmeans2D[i, :2]|, denom[i] += 1
     * (/root/reference/scene/gaussian_model.py:735-737) and max_radii[i] = max(max_radii[i], radii[i])
     * (/root/reference/trainers/train_static.py:125).  Same arithmetic as egs_densify_stats, one launch less per iteration. */
    float* stat_grad_accum /*[P] in/out or NULL*/, float* stat_denom /*[P] in/out or NULL*/, float* stat_max_radii /*[P] in/out or NULL*/,
    const uint32_t* skip_flag /*device uint32[1] or NULL: non-zero = leave the three statistics untouched (overflowed frame)*/,
    int grad_mask /*EGS_GRAD_* bits, 0 = all (see above)*/,
    void* scratch /* egs_backward_scratch_bytes(P) */, void* stream, int debug);

/* ---- backward with the optimizer inside (ABI 2 addition; no upstream counterpart: upstream returns the gradients to autograd
 *      and torch.optim.Adam reads them back, /root/reference/trainers/train_static.py:97,137).
 * egs_backward_adam is egs_backward plus, for every LEAF the sink owns, the Adam step of that leaf taken by the kernel that
 * produces its gradient: the gradient never reaches HBM, the parameter is not read again by an optimizer launch, and the training
 * step has one launch less.  Arithmetic and results are those of egs_adam_step_capturable, bit for bit.
 *   - A leaf is one of the five per-Gaussian inputs below; `param` must be the SAME device array the call receives as that input
 *     (for EGS_SINK_OPACITY: the array the forward received as `opacities`; the backward itself reads the activated value from the
 *     geometry buffer).  It is updated IN PLACE after its last read; exp_avg / exp_avg_sq have its shape, lr / step are device float[1]:
 *     torch's state["step"], advanced by one per call.  param == NULL: that leaf is not fused.
 *   - The dL_d* output of a fused leaf may be NULL (nothing written); if given, the gradient is written as well.
 *   - Conditions (else EGS_ERR_MODE): EGS_SINK_SCALES / EGS_SINK_ROTATIONS need scales + rotations (not cov3D_precomp).
 *     Colour: either `shs` with sh_coeffs == 1 and no shs_rest (EGS_SINK_SH, and EGS_SINK_MEANS3D is then stepped by the same
 *     kernel as the others), or split spherical harmonics with sh_coeffs == 16 and 16-byte aligned arrays: EGS_SINK_SH (the DC
 *     block), EGS_SINK_SH_REST and EGS_SINK_MEANS3D are then stepped by the spherical-harmonics launch that finishes their gradients
 *     (dL_dmeans3D must be given: it carries the partial gradient between the two launches); with `colors_precomp` only
 *     EGS_SINK_MEANS3D of the three.  Any other colour layout: those three leaves cannot be fused.
 *   - skip_flag set: no step is taken and none is counted.  active_rows: rows >= *active_rows are left alone (capacity-sized
 *     models), as in egs_adam_step_capturable.
 *   - coef: device float[12] scratch owned by the caller, written and read by this call only.
 *   - CALLER-CHECKED PRECONDITION (the library cannot see the caller's computation graph): the gradient this backward produces for a
 *     fused leaf must be the leaf's WHOLE gradient for this optimizer step -- this rasterizer call is the only consumer of the
 *     parameter in the loss.  A second path (an entropy term on the opacities, colours computed from the positions outside the
 *     library, a second render of the same model in the same iteration) would have its share applied with stale moments or lost.
 *     The Python host side enforces it where it can (optim.FusedAdam.make_sink leaves the positions out when colors_precomp requires
 *     a gradient; FusedAdam.step raises when a fused leaf arrives with a .grad from another path); a direct C caller owns the check. */
#define EGS_SINK_MEANS3D   0    /* [P,3] */
#define EGS_SINK_OPACITY   1    /* [P,1] */
#define EGS_SINK_SCALES    2    /* [P,3] */
#define EGS_SINK_ROTATIONS 3    /* [P,4] */
#define EGS_SINK_SH        4    /* [P,1,3]: `shs` when it has one coefficient, or the DC block of split spherical harmonics */
#define EGS_SINK_SH_REST   5    /* [P,15,3]: `shs_rest` of split spherical harmonics with sh_coeffs == 16 */
typedef struct egs_adam_leaf {
    float* param; float* exp_avg; float* exp_avg_sq;
    const float* lr;     /* device float[1] */
    float* step;         /* device float[1], in/out */
} egs_adam_leaf;
typedef struct egs_adam_sink {
    egs_adam_leaf leaf[6];       /* indexed by EGS_SINK_* */
    float beta1, beta2, eps;
    float* coef;                 /* device float[12] scratch */
    const int32_t* active_rows;  /* device int32[1] or NULL */
} egs_adam_sink;
int egs_backward_adam(
    int P, int sh_degree, int sh_coeffs, int64_t R,
    const float* background, const float* means3D, const float* shs, const float* shs_rest, const float* colors_precomp,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp, int activation_flags,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    int width, int height, float tan_fovx, float tan_fovy,
    const int32_t* radii, const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
    const float* dL_dout_color, const float* dL_dout_depth, const float* dL_dout_alpha,
    float* dL_dmeans2D, float* dL_dcolors /*may be NULL with shs*/, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
    float* dL_dsh_rest, float* dL_dscales, float* dL_drotations,
    float* stat_grad_accum, float* stat_denom, float* stat_max_radii, const uint32_t* skip_flag,
    const egs_adam_sink* sink /*HOST; NULL = no leaf is fused*/,
    int prologue_done /*non-zero: egs_l1_ssim_backward_ex carried this frame's egs_backward_prologue (same scratch, same sink)*/,
    const egs_object_rotation* rot /*HOST or NULL: as given to the forward*/,
    int grad_mask /*as egs_backward; the colours-only path is taken only with sink == NULL*/,
    void* scratch, void* stream, int debug);

/* ---- frustum test only  (upstream: markVisible) -------------------------------------------------- */
int egs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present /*[P] out, 0/1*/, void* stream);

/* ==== "next" rows of SURVEY.md section 8f: the callers either side of the rasterizer ======================= */

/* ---- f-1: fused 3D covariance producer.  Replaces build_scaling_rotation / strip_symmetric
 *      (/root/reference/utils/general_utils.py:110-156) and the covariance activations
 *      (/root/reference/scene/gaussian_model.py:29-33,46-63):
 *      q <- q/|q|; L = R(q) diag(scale_modifier * scaling); rows with selected[i] != 0 (all rows if selected == NULL)
 *      get L <- M L when M9 != NULL; cov6 = unique entries of L L^T in the order (00,01,02,11,12,22).
 *      scaling_is_log != 0: `scaling` holds the raw parameters and the scaling activation exp() (gaussian_model.py:36) is
 *      applied by the kernel; the backward then returns the gradient w.r.t. the raw parameters.
 *      opacity_raw != NULL: the same launch also applies the opacity activation, opacity[i] = sigmoid(opacity_raw[i])
 *      (gaussian_model.py:40), and the backward turns dL/dopacity into dL/dopacity_raw -- two elementwise launches less per step. */
int egs_cov3d_forward(int N, const float* scaling /*[N,3]*/, int scaling_is_log, float scale_modifier, const float* rotation /*[N,4] raw*/,
                      const float* M9 /*[9] device, row-major, or NULL*/, const uint8_t* selected /*[N] or NULL*/,
                      float* cov6 /*[N,6] out*/, const float* opacity_raw /*[N] or NULL*/, float* opacity /*[N] out or NULL*/,
                      void* stream);
/* row0_grad_mult reproduces the reference's duplicated-index gradient on Gaussian 0 (egogaussian_amd/covariance.py);
 * pass 1.0 otherwise; row0_grad_mult_dev (device float[1], may be NULL) overrides it with a value computed on the device, so
 * that the caller needs no host read of the selection count.  dL_dM9 (device [9], may be NULL) is written by the callee; when it is requested, dM_scratch must
 * provide egs_cov3d_dm_scratch_floats(N) floats (per-workgroup partial sums, no atomics). */
size_t egs_cov3d_dm_scratch_floats(int N);
int egs_cov3d_backward(int N, const float* scaling, int scaling_is_log, float scale_modifier, const float* rotation, const float* M9,
                       const uint8_t* selected, float row0_grad_mult, const float* row0_grad_mult_dev, const float* dL_dcov6 /*[N,6]*/,
                       float* dL_dscaling /*[N,3] out*/, float* dL_drotation /*[N,4] out*/, float* dL_dM9, float* dM_scratch,
                       const float* opacity /*[N] forward output or NULL*/, const float* dL_dopacity /*[N]*/,
                       float* dL_dopacity_raw /*[N] out*/, void* stream);

/* ---- f-3: fused image loss (1 - lambda) * L1 + lambda * (1 - SSIM), 11x11 Gaussian window sigma 1.5, zero padding.
 *      Replaces l1_loss + ssim (/root/reference/utils/loss_utils.py:57-107) as combined at
 *      /root/reference/trainers/train_static.py:92-95.  The forward writes the scalar loss (device float[1]), using
 *      egs_l1_ssim_partial_count floats of scratch for per-block partial sums, and three derivative maps [C,H,W] the
 *      backward consumes.  `gate` (optional, [H,W]) multiplies the image gradient
 *      per pixel -- the hand-mask hook of train_static.py:91.
 *      `loss_running_sum` (optional device float[1]): the loss value is also ADDED to it -- a trainer's logging sum without a
 *      launch or a host read per iteration.
 *      Deferred value: with loss == NULL the forward does not launch the kernel that assembles the scalar; pass the same
 *      partial_sums to egs_l1_ssim_backward as `deferred_partial_sums` (+ `deferred_loss`, `loss_running_sum`) and one wave of
 *      the backward kernel assembles it.  For a training step replayed from a hipGraph, which reads the value only after the
 *      backward anyway, that is one launch (~4.5 us of GPU time) less. */
size_t egs_l1_ssim_partial_count(int channels, int height, int width);
int egs_l1_ssim_forward(int channels, int height, int width, const float* img /*[C,H,W]*/, const float* gt /*[C,H,W]*/,
                        float lambda_dssim, float* partial_sums /*scratch*/, float* dm_dmu1, float* dm_dexx, float* dm_dexy,
                        float* loss /*device [1] out, or NULL: deferred*/, float* loss_running_sum /*device [1] in/out or NULL*/,
                        void* stream);
/* ABI 4: the two terms as two values, for a loop that combines them itself -- the reference's trainers call l1_loss(x, gt) and
 * ssim(x, gt) (/root/reference/utils/loss_utils.py:57-58,79-107) and weigh them in Python (trainers/train_static.py:92-95).  ONE forward
 * launch (+ a one-workgroup reduction) yields mean|x - gt| and mean SSIM; ONE backward launch takes an upstream scalar for each, read on
 * the device (autograd hands them over as tensors).  Same kernels and maps as egs_l1_ssim_forward / _backward; the backward is declared
 * behind egs_backward_prologue below (it can carry a rasterizer backward's preparation like egs_l1_ssim_backward_ex). */
int egs_l1_ssim_pair_forward(int channels, int height, int width, const float* img, const float* gt, float* partial_sums, float* dm_dmu1,
                             float* dm_dexx, float* dm_dexy, float* l1_out /*[1]*/, float* ssim_out /*[1]*/, void* stream);
int egs_l1_ssim_backward(int channels, int height, int width, const float* img, const float* gt, float lambda_dssim,
                         const float* upstream_grad /*device [1]*/, const float* gate /*[H,W] or NULL*/,
                         const float* dm_dmu1, const float* dm_dexx, const float* dm_dexy, float* dL_dimg /*[C,H,W] out*/,
                         const float* deferred_partial_sums /*NULL unless the forward deferred the value*/,
                         float* deferred_loss /*device [1] out or NULL*/, float* loss_running_sum /*device [1] in/out or NULL*/,
                         void* stream);

/* The same launch carrying, in extra workgroups, what a rasterizer backward of the same frame needs done before its blend
 * kernel: ordering the tiles by the cost the forward recorded, clearing the gradient accumulator (`scratch`) and, with a sink, the
 * fused optimizer's per-step bookkeeping.  In a training step this launch sits between the two blends and leaves most of the
 * machine idle, while that preparation as a launch of its own (which egs_backward makes otherwise) costs ~12 us.  The
 * egs_backward_adam call that follows on the same stream is told so with prologue_done = 1 and must receive the same scratch,
 * image buffer, sink and skip_flag.  side == NULL: egs_l1_ssim_backward. */
typedef struct egs_backward_prologue {
    int P, width, height;            /* of the rasterizer call */
    void* image_buffer;              /* of that call's forward */
    void* scratch;                   /* egs_backward_scratch_bytes(P): the backward's scratch */
    const egs_adam_sink* sink;       /* HOST, or NULL */
    const uint32_t* skip_flag;       /* device uint32[1] or NULL */
    const void* geom_buffer;         /* ABI 3: of that call's forward, or NULL.  With it only the accumulator lines in use are cleared
                                      * (the replica lines of the frame's hot Gaussians, csrc/egs_common.h); NULL: all of them */
} egs_backward_prologue;
int egs_l1_ssim_pair_backward(int channels, int height, int width, const float* img, const float* gt, const float* upstream_l1 /*[1]*/,
                              const float* upstream_ssim /*[1]*/, const float* gate /*[H,W] or NULL*/, const float* dm_dmu1, const float* dm_dexx,
                              const float* dm_dexy, float* dL_dimg, const egs_backward_prologue* side /*HOST or NULL*/, void* stream);
int egs_l1_ssim_backward_ex(int channels, int height, int width, const float* img, const float* gt, float lambda_dssim,
                            const float* upstream_grad, const float* gate, const float* dm_dmu1, const float* dm_dexx,
                            const float* dm_dexy, float* dL_dimg, const float* deferred_partial_sums, float* deferred_loss,
                            float* loss_running_sum, const egs_backward_prologue* side /*HOST or NULL*/, void* stream);

/* ---- ABI 5: a training step WITHOUT a loss-backward launch.  The rasterizer's backward blend runs one workgroup per 16x16 tile and starts
 * by loading dL/dC of its 256 pixels; that gradient is a function of what the loss FORWARD left (its three derivative maps, the image, the
 * ground truth) in an 11x11 neighbourhood of the pixel, so the workgroup can compute it itself -- with k_l1_ssim_backward's arithmetic
 * operation for operation: the values are bit-identical to the ones that kernel would have written (tests/test_gpu_fused.py) -- and the
 * step loses a launch whose own work was ~15 us (config C: 3 350 -> ~3 450 it/s).  What that launch also carried moves:
 *   egs_l1_ssim_forward_ex     the forward, carrying the backward blend's preparation (egs_backward_prologue) as egs_l1_ssim_backward_ex did
 *   egs_backward_lossgrad      egs_backward_adam with an egs_loss_grad in place of the three dL_dout_* arrays (colour loss only: the depth
 *                              and alpha outputs carry no gradient); sink may be NULL (no optimizer inside); with deferred_partial_sums the
 *                              loss VALUE the forward deferred is assembled by one wave of the blend launch. */
typedef struct egs_loss_grad {       /* HOST struct */
    const float* image;              /* [3,H,W] the rendered image the loss was taken on */
    const float* gt;                 /* [3,H,W] */
    const float* dm_dmu1; const float* dm_dexx; const float* dm_dexy;   /* [3,H,W] each, as written by egs_l1_ssim_forward */
    const float* gate;               /* [H,W] or NULL: per-pixel factor on the image gradient (the trainers' 1 - hand_mask hook) */
    const float* upstream_grad;      /* device float[1]: dL/d(loss) */
    float lambda_dssim;
    const float* deferred_partial_sums;   /* NULL unless the forward deferred the value (loss == NULL there) */
    float* deferred_loss;            /* device float[1] out, or NULL */
    float* loss_running_sum;         /* device float[1] in/out, or NULL */
} egs_loss_grad;
int egs_l1_ssim_forward_ex(int channels, int height, int width, const float* img, const float* gt, float lambda_dssim,
                           float* partial_sums, float* dm_dmu1, float* dm_dexx, float* dm_dexy, float* loss /*or NULL: deferred*/,
                           float* loss_running_sum, const egs_backward_prologue* side /*HOST or NULL*/, void* stream);
int egs_backward_lossgrad(int P, int sh_degree, int sh_coeffs, int64_t R, const float* background, const float* means3D,
                          const float* shs, const float* shs_rest, const float* colors_precomp, const float* scales, float scale_modifier,
                          const float* rotations, const float* cov3D_precomp, int activation_flags, const float* viewmatrix, const float* projmatrix,
                          const float* campos, int width, int height, float tan_fovx, float tan_fovy, const int32_t* radii,
                          const void* geom_buffer, const void* binning_buffer, const void* image_buffer, const egs_loss_grad* loss_grad /*HOST*/,
                          float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dsh_rest,
                          float* dL_dscales, float* dL_drotations, float* stat_grad_accum, float* stat_denom, float* stat_max_radii,
                          const uint32_t* skip_flag, const egs_adam_sink* sink /*HOST or NULL*/, int prologue_done, const egs_object_rotation* rot /*HOST or NULL*/,
                          int grad_mask, void* scratch, void* stream, int debug);

/* ---- f-4 (optimizer part): multi-tensor Adam step in one launch.  Same update as torch.optim.Adam(weight_decay=0,
 *      amsgrad=False), which the reference builds at /root/reference/scene/gaussian_model.py:198 and steps at
 *      /root/reference/trainers/train_static.py:137.  All array arguments are HOST arrays of length n_tensors holding
 *      device pointers / sizes / per-tensor learning rate and 1-based step count. */
int egs_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                  float* const* exp_avg_sq, const int64_t* numels, const float* lrs, const int64_t* steps,
                  float beta1, float beta2, float eps, void* stream);
/* hipGraph-capturable variant: `lr_dev[t]` (device float[1]) is tensor t's learning rate, read by the kernel; the step number
 * comes from `counters[t]` (device uint32[egs_adam_workgroups(numels[t])], one array per tensor, never shared), every word of
 * which must hold the number of steps tensor t has taken; the launch adds one to each, so a captured iteration needs no
 * other kernel to keep count.  `step_dev[t]` (device float[1]) is WRITTEN with the number of the step
 * just taken (torch's state["step"]). */
int64_t egs_adam_workgroups(int64_t numel);
/* ABI 2: `skip_flag` (device uint32[1] or NULL) non-zero makes the launch a no-op (parameters, moments, counters and step_dev
 * untouched) -- the overflow word of egs_forward_enqueue.  `active_rows` (device int32[1] or NULL) with `row_floats` (HOST
 * int32[n_tensors], 0 = whole tensor) limits tensor t to its first *active_rows * row_floats[t] elements: the live rows of a
 * capacity-sized model. */
int egs_adam_step_capturable(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                             float* const* exp_avg_sq, const int64_t* numels, float* const* step_dev /*HOST array of device ptrs*/,
                             const float* const* lr_dev /*HOST array of device ptrs*/, uint32_t* const* counters /*HOST array of device ptrs*/,
                             float beta1, float beta2, float eps, const uint32_t* skip_flag, const int32_t* active_rows,
                             const int32_t* row_floats /*HOST or NULL*/, void* stream);

/* ---- f-4 (densify / prune part): the bookkeeping of /root/reference/scene/gaussian_model.py:506-709,735-740 on the device.
 *      egs_densify_stats         per-iteration statistics in one pass: for every visible Gaussian (visible[i] != 0, or
 *                                radii[i] > 0 when `visible` is NULL)  grad_accum += |viewspace_grad.xy|, denom += 1 and, when
 *                                radii and max_radii2D are given, max_radii2D = max(max_radii2D, radii)
 *                                (add_densification_stats :735-740 + trainers/train_static.py:125).
 *      egs_densify_plan          evaluates densify_and_clone / densify_and_split / the final prune of densify_and_prune
 *                                (:588-709) for every Gaussian and compacts the results: src_index[k], kind[k] (0 kept original,
 *                                1 clone, 2 / 3 first / second split child) for the k-th Gaussian of the new model, in the
 *                                reference's order; split_rank[i] = rank of source i among the split ones (-1: not split);
 *                                totals (device uint64[4]) = kept originals, kept clones, kept children PER COPY, split sources.
 *                                New size = totals[0] + totals[1] + 2 totals[2].  Everything is enqueued; read totals after
 *                                synchronising.  EGS_ERR_MODE for prune_prev_gen == 0 without a curr_gen (the reference raises).
 *      egs_prune_plan            the same plan for prune_points(mask) (:536-563): kept = !mask.
 *      egs_gather_rows_f32       out[k][:] = in[src_index[k]][:]; zero_new_rows != 0 writes zeros for kind != 0 (Adam moments).
 *      egs_gather_i32            same for the int32 side arrays; clone_value_on: clones take clone_value (curr_gen).
 *      egs_split_children        rows of kind 2 / 3: xyz = parent + R(q) (exp(scaling) * z), scaling = log(exp(scaling) / 1.6);
 *                                z = the 2 * n_split standard-normal draws, first children first (torch.normal at :621). */
int egs_densify_stats(int P, const float* viewspace_grad /*[P,3]*/, const uint8_t* visible /*[P] or NULL*/,
                      const int32_t* radii /*[P] or NULL*/, float* grad_accum /*[P]*/, float* denom /*[P]*/,
                      float* max_radii2D /*[P] or NULL*/, void* stream);
size_t egs_densify_plan_scratch_bytes(int P);
int egs_densify_plan(int P, const float* grad_accum, const float* denom, const float* scaling_raw /*[P,3]*/,
                     const float* opacity_raw /*[P]*/, const float* max_radii2D /*[P]*/, const int32_t* generation /*[P]*/,
                     const int32_t* is_object /*[P]*/, float max_grad, float min_opacity, float percent_dense, float extent,
                     float max_screen_size /* <= 0: criterion off */, int clone, int split, int has_curr_gen, int curr_gen,
                     int prune_prev_gen, int has_which_object, int which_object, void* scratch,
                     int32_t* src_index /*[3P] out*/, uint8_t* kind /*[3P] out*/, int32_t* split_rank /*[P] out*/,
                     uint64_t* totals /*device [4] out*/, void* stream);
int egs_prune_plan(int P, const uint8_t* prune_mask /*[P]*/, void* scratch, int32_t* src_index, uint8_t* kind, int32_t* split_rank,
                   uint64_t* totals, void* stream);
int egs_gather_rows_f32(int64_t rows, int row_floats, const int32_t* src_index, const uint8_t* kind, int zero_new_rows,
                        const float* in, float* out, void* stream);
int egs_gather_i32(int64_t rows, const int32_t* src_index, const uint8_t* kind, int clone_value_on, int clone_value,
                   const int32_t* in, int32_t* out, void* stream);
int egs_split_children(int64_t rows, const int32_t* src_index, const uint8_t* kind, const int32_t* split_rank, int n_split,
                       const float* z /*[2 n_split, 3]*/, const float* xyz_old, const float* scaling_old, const float* rotation_old,
                       float* xyz_new, float* scaling_new, void* stream);

/* ---- f-2: mean squared distance of every point to its 3 nearest neighbours (self excluded by index).
 *      Replaces simple_knn._C.distCUDA2 (un-vendored submodule, /root/reference/.gitmodules:4-6), imported at
 *      /root/reference/scene/gaussian_model.py:21 and called at :301.  Exact (all pairs). */
int egs_knn3_mean_dist2(int N, const float* points /*[N,3]*/, float* mean_dist2 /*[N] out*/, void* stream);
/* The same statistic, bit for bit, through a uniform-grid search (counting sort of the points by cell, then rings of cells around
 * every query until the third neighbour is provably inside): O(N) for clouds of roughly even density instead of O(N^2) -- the
 * all-pairs call takes 0.4 s at 1 M points.  Points with a NaN / inf coordinate are nobody's neighbour and get +inf, in both
 * calls.  `scratch`: egs_knn3_grid_scratch_bytes(N) bytes of device memory. */
size_t egs_knn3_grid_scratch_bytes(int N);
int egs_knn3_grid(int N, const float* points /*[N,3]*/, float* mean_dist2 /*[N] out*/, void* scratch, void* stream);

/* Whether a forward of this size, given a placement buffer and these EGS_CALL_* flags, folds the count pass of its bucketing into the
 * preprocess launch (1) or not (0: very large images, or EGS_CALL_SEPARATE_COUNT). */
int egs_forward_fuses_count(int P, int width, int height, int flags);

/* ---- optional per-stage timing with HIP events on the caller's stream (bench / profiling aid) ------
 * The only process-wide state in the library; off by default.  egs_profile_begin allocates an event pool and
 * turns recording on; every stage launched afterwards is bracketed by two events on its stream;
 * egs_profile_end waits for the events, returns per-stage total milliseconds and launch counts, frees the pool. */
enum {
    EGS_K_PREPROCESS = 0, EGS_K_SCAN, EGS_K_DUPLICATE, EGS_K_SORT, EGS_K_RANGES, EGS_K_RENDER_FWD,
    EGS_K_RENDER_BWD, EGS_K_PREPROCESS_BWD, EGS_K_COUNT
};
int         egs_profile_begin(int max_records);
int         egs_profile_end(double* total_ms /*HOST [EGS_K_COUNT]*/, int* launches /*HOST [EGS_K_COUNT]*/);
const char* egs_profile_stage_name(int stage);

#ifdef __cplusplus
}
#endif
#endif /* EGS_RASTER_H */
