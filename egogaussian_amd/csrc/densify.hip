// densify.hip -- densification bookkeeping on the device (densify / prune part).
// What the reference does with boolean-mask indexing, torch.cat and repeated nn.Parameter re-creation
// (/root/reference/scene/gaussian_model.py:506-709, :735-740; trainers/train_static.py:125-127) is restated as:
//   k_densify_stats      the per-iteration statistics (gradient-norm accumulator, visit count, largest screen radius) in one
//                        pass over the Gaussians -- no index tensors, no host round trip for the mask size
//   k_densify_flags      every selection rule of densify_and_clone / densify_and_split / the final prune, evaluated once per
//                        source Gaussian for the up to four results it can produce (itself, a clone, two split children)
//   scans + k_densify_emit   stream compaction: the results' (source, kind) in EXACTLY the order the reference's
//                        cat / mask sequence leaves them: [kept originals | kept clones | kept first children | kept second children]
//   k_gather_rows_f32 / k_gather_i32 / k_split_children      build every array of the new model from the plan
// Semantics follow the reference code path as it actually runs, including: NaN gradient means (0/0) count as 0; the
// statistics (and with them max_radii2D) are zeroed by densification_postfix before the final prune whenever a clone or
// split step ran; clones take `curr_gen` when it is given, split children always inherit (see oracle/densify_torch.py).
#include "egs_common.h"
#include <math.h>

namespace {

__global__ __launch_bounds__(256) void k_densify_stats(int P, const float* __restrict__ vs_grad, const uint8_t* __restrict__ visible,
                                                        const int32_t* __restrict__ radii, float* __restrict__ accum,
                                                        float* __restrict__ denom, float* __restrict__ max_radii) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const bool vis = visible ? visible[i] != 0 : radii[i] > 0;
    if (!vis) return;
    const float gx = vs_grad[3 * i], gy = vs_grad[3 * i + 1];
    accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.f;
    if (radii && max_radii) max_radii[i] = fmaxf(max_radii[i], (float)radii[i]);
}

struct DensifyRules {
    float max_grad, min_opacity, dense_extent /* percent_dense * extent */, big_extent /* 0.1 * extent */, max_screen_size;
    int use_screen_size, clone, split, has_curr_gen, curr_gen, prune_prev_gen, has_object, which_object, stats_reset;
};

__global__ __launch_bounds__(256) void k_densify_flags(int P, const float* __restrict__ accum, const float* __restrict__ denom,
                                                        const float* __restrict__ scaling_raw, const float* __restrict__ opacity_raw,
                                                        const float* __restrict__ max_radii, const int32_t* __restrict__ generation,
                                                        const int32_t* __restrict__ is_object, DensifyRules r,
                                                        uint32_t* __restrict__ keep_orig, uint32_t* __restrict__ keep_clone,
                                                        uint32_t* __restrict__ keep_child, uint32_t* __restrict__ is_split) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float g = accum[i] / denom[i];
    if (g != g) g = 0.f;                                              // grads[grads.isnan()] = 0
    const float smax = fmaxf(fmaxf(expf(scaling_raw[3 * i]), expf(scaling_raw[3 * i + 1])), expf(scaling_raw[3 * i + 2]));
    const bool obj_ok = !r.has_object || is_object[i] == r.which_object;
    const bool C = r.clone && fabsf(g) >= r.max_grad && smax <= r.dense_extent && obj_ok;
    const bool S = r.split && g >= r.max_grad && smax > r.dense_extent && obj_ok;
    const float opacity = 1.f / (1.f + expf(-opacity_raw[i]));
    const bool faint = opacity < r.min_opacity;
    const int gen = generation[i];
    const bool big_vs = r.use_screen_size && !r.stats_reset && max_radii[i] > r.max_screen_size;
    const bool big_ws = r.use_screen_size && smax > r.big_extent;
    const bool big_ws_child = r.use_screen_size && expf(logf(smax / 1.6f)) > r.big_extent;    // children carry log(scale / 1.6)
    const bool gen_orig = r.prune_prev_gen || gen == r.curr_gen;
    const bool gen_clone = r.prune_prev_gen || (r.has_curr_gen ? r.curr_gen : gen) == r.curr_gen;
    const bool prune_orig = (faint || big_vs || big_ws) && gen_orig;
    const bool prune_clone = (faint || big_ws) && gen_clone;
    const bool prune_child = (faint || big_ws_child) && gen_orig;
    keep_orig[i] = (!S && !prune_orig) ? 1u : 0u;
    keep_clone[i] = (C && !prune_clone) ? 1u : 0u;
    keep_child[i] = (S && !prune_child) ? 1u : 0u;
    is_split[i] = S ? 1u : 0u;
}

// Plain mask -> flags (prune_points): keep_orig = !mask, nothing else.
__global__ __launch_bounds__(256) void k_mask_flags(int P, const uint8_t* __restrict__ prune_mask, uint32_t* __restrict__ keep_orig,
                                                     uint32_t* __restrict__ keep_clone, uint32_t* __restrict__ keep_child,
                                                     uint32_t* __restrict__ is_split) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    keep_orig[i] = prune_mask[i] ? 0u : 1u; keep_clone[i] = 0u; keep_child[i] = 0u; is_split[i] = 0u;
}

// flags: values before the scans; off_*: their exclusive scans; totals[0..3] = (originals, clones, children, splits)
__global__ __launch_bounds__(256) void k_densify_emit(int P, const uint32_t* __restrict__ flags /* [4][P] */, const uint32_t* __restrict__ offs /* [4][P] */,
                                                       const uint64_t* __restrict__ totals, int32_t* __restrict__ src, uint8_t* __restrict__ kind,
                                                       int32_t* __restrict__ split_rank /* [P] */) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const uint32_t nO = (uint32_t)totals[0], nC = (uint32_t)totals[1], nK = (uint32_t)totals[2];
    if (flags[i]) { const uint32_t p = offs[i]; src[p] = i; kind[p] = 0; }
    if (flags[(size_t)P + i]) { const uint32_t p = nO + offs[(size_t)P + i]; src[p] = i; kind[p] = 1; }
    if (flags[2 * (size_t)P + i]) {
        const uint32_t p = nO + nC + offs[2 * (size_t)P + i];
        src[p] = i; kind[p] = 2; src[p + nK] = i; kind[p + nK] = 3;
    }
    split_rank[i] = flags[3 * (size_t)P + i] ? (int32_t)offs[3 * (size_t)P + i] : -1;
}

// out[row][0..D) = in[src[row]][0..D); mode 1 (optimizer moments): zero for every row that is not a kept original.
__global__ __launch_bounds__(256) void k_gather_rows_f32(size_t n_elems, int D, const int32_t* __restrict__ src, const uint8_t* __restrict__ kind,
                                                          int zero_new, const float* __restrict__ in, float* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_elems) return;
    const size_t row = e / (size_t)D;
    const int col = (int)(e - row * (size_t)D);
    out[e] = (zero_new && kind[row] != 0) ? 0.f : in[(size_t)src[row] * D + col];
}

__global__ __launch_bounds__(256) void k_gather_i32(int n, const int32_t* __restrict__ src, const uint8_t* __restrict__ kind, int clone_value_on,
                                                     int clone_value, const int32_t* __restrict__ in, int32_t* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    out[e] = (clone_value_on && kind[e] == 1) ? clone_value : in[src[e]];
}

// Rows of kind 2 / 3 (first / second split child): position = parent + R(q) (scale * z), raw scale = log(scale / 1.6).
// z holds the 2 * n_split standard-normal draws in the reference's order: all first children, then all second children.
__global__ __launch_bounds__(256) void k_split_children(int n, const int32_t* __restrict__ src, const uint8_t* __restrict__ kind,
                                                         const int32_t* __restrict__ split_rank, int n_split, const float* __restrict__ z,
                                                         const float* __restrict__ xyz_old, const float* __restrict__ scaling_old,
                                                         const float* __restrict__ rot_old, float* __restrict__ xyz_new,
                                                         float* __restrict__ scaling_new) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n || kind[e] < 2) return;
    const int s = src[e];
    const float* zz = z + 3 * ((size_t)(kind[e] - 2) * n_split + split_rank[s]);
    const float sc[3] = { expf(scaling_old[3 * s]), expf(scaling_old[3 * s + 1]), expf(scaling_old[3 * s + 2]) };
    const float v[3] = { sc[0] * zz[0], sc[1] * zz[1], sc[2] * zz[2] };
    float q[4] = { rot_old[4 * s], rot_old[4 * s + 1], rot_old[4 * s + 2], rot_old[4 * s + 3] };
    const float inv = 1.f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float r = q[0] * inv, x = q[1] * inv, y = q[2] * inv, w = q[3] * inv;
    const float R[9] = { 1.f - 2.f * (y * y + w * w), 2.f * (x * y - r * w), 2.f * (x * w + r * y),
                         2.f * (x * y + r * w), 1.f - 2.f * (x * x + w * w), 2.f * (y * w - r * x),
                         2.f * (x * w - r * y), 2.f * (y * w + r * x), 1.f - 2.f * (x * x + y * y) };
#pragma unroll
    for (int a = 0; a < 3; a++) {
        xyz_new[3 * e + a] = (R[3 * a] * v[0] + R[3 * a + 1] * v[1] + R[3 * a + 2] * v[2]) + xyz_old[3 * s + a];
        scaling_new[3 * e + a] = logf(sc[a] / 1.6f);
    }
}

}  // namespace

extern "C" {

int egs_densify_stats(int P, const float* viewspace_grad, const uint8_t* visible, const int32_t* radii, float* grad_accum,
                      float* denom, float* max_radii2D, void* stream) {
    if (P < 0) return EGS_ERR_ARG;
    if (P == 0) return 0;
    if (!viewspace_grad || !grad_accum || !denom || (!visible && !radii)) return EGS_ERR_ARG;
    hipLaunchKernelGGL(k_densify_stats, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, viewspace_grad, visible, radii,
                       grad_accum, denom, max_radii2D);
    return (int)hipGetLastError();
}

size_t egs_densify_plan_scratch_bytes(int P) {
    const size_t n = (size_t)(P > 0 ? P : 0);
    return egs_align(8 * n * sizeof(uint32_t)) + egs_align((egs_scan_scratch_elems(n) + 64) * sizeof(uint32_t));
}

// Everything is enqueued; the caller reads totals[4] after synchronising and then sizes the new model.
static int plan_common(int P, uint32_t* flags, void* scratch, int32_t* src_index, uint8_t* kind, int32_t* split_rank, uint64_t* totals,
                       hipStream_t s) {
    uint32_t* offs = flags + 4 * (size_t)P;
    uint32_t* spine = (uint32_t*)((char*)scratch + egs_align(8 * (size_t)P * sizeof(uint32_t)));
    for (int k = 0; k < 4; k++) {
        hipError_t e = egs_launch_scan_u32(flags + (size_t)k * P, offs + (size_t)k * P, (size_t)P, 0, spine, totals + k, s);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k_densify_emit, dim3((P + 255) / 256), dim3(256), 0, s, P, flags, offs, totals, src_index, kind, split_rank);
    return (int)hipGetLastError();
}

int egs_densify_plan(int P, const float* grad_accum, const float* denom, const float* scaling_raw, const float* opacity_raw,
                     const float* max_radii2D, const int32_t* generation, const int32_t* is_object, float max_grad, float min_opacity,
                     float percent_dense, float extent, float max_screen_size /* <= 0: criterion off */, int clone, int split,
                     int has_curr_gen, int curr_gen, int prune_prev_gen, int has_which_object, int which_object, void* scratch,
                     int32_t* src_index /*[3P]*/, uint8_t* kind /*[3P]*/, int32_t* split_rank /*[P]*/, uint64_t* totals /*device [4]*/,
                     void* stream) {
    if (P <= 0 || !grad_accum || !denom || !scaling_raw || !opacity_raw || !max_radii2D || !generation || !is_object || !scratch ||
        !src_index || !kind || !split_rank || !totals)
        return EGS_ERR_ARG;
    if (!prune_prev_gen && !has_curr_gen) return EGS_ERR_MODE;
    DensifyRules r;
    r.max_grad = max_grad; r.min_opacity = min_opacity; r.dense_extent = percent_dense * extent; r.big_extent = 0.1f * extent;
    r.max_screen_size = max_screen_size; r.use_screen_size = max_screen_size > 0.f; r.clone = clone != 0; r.split = split != 0;
    r.has_curr_gen = has_curr_gen != 0; r.curr_gen = curr_gen; r.prune_prev_gen = prune_prev_gen != 0;
    r.has_object = has_which_object != 0; r.which_object = which_object; r.stats_reset = (clone || split) ? 1 : 0;
    uint32_t* flags = (uint32_t*)scratch;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_densify_flags, dim3((P + 255) / 256), dim3(256), 0, s, P, grad_accum, denom, scaling_raw, opacity_raw, max_radii2D,
                       generation, is_object, r, flags, flags + (size_t)P, flags + 2 * (size_t)P, flags + 3 * (size_t)P);
    return plan_common(P, flags, scratch, src_index, kind, split_rank, totals, s);
}

int egs_prune_plan(int P, const uint8_t* prune_mask, void* scratch, int32_t* src_index, uint8_t* kind, int32_t* split_rank,
                   uint64_t* totals, void* stream) {
    if (P <= 0 || !prune_mask || !scratch || !src_index || !kind || !split_rank || !totals) return EGS_ERR_ARG;
    uint32_t* flags = (uint32_t*)scratch;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_mask_flags, dim3((P + 255) / 256), dim3(256), 0, s, P, prune_mask, flags, flags + (size_t)P, flags + 2 * (size_t)P,
                       flags + 3 * (size_t)P);
    return plan_common(P, flags, scratch, src_index, kind, split_rank, totals, s);
}

int egs_gather_rows_f32(int64_t rows, int row_floats, const int32_t* src_index, const uint8_t* kind, int zero_new_rows, const float* in,
                        float* out, void* stream) {
    if (rows < 0 || row_floats < 0) return EGS_ERR_ARG;
    const size_t n = (size_t)rows * (size_t)row_floats;
    if (n == 0) return 0;
    if (!src_index || !kind || !in || !out) return EGS_ERR_ARG;
    hipLaunchKernelGGL(k_gather_rows_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, row_floats, src_index, kind,
                       zero_new_rows, in, out);
    return (int)hipGetLastError();
}

int egs_gather_i32(int64_t rows, const int32_t* src_index, const uint8_t* kind, int clone_value_on, int clone_value, const int32_t* in,
                   int32_t* out, void* stream) {
    if (rows < 0) return EGS_ERR_ARG;
    if (rows == 0) return 0;
    if (!src_index || !kind || !in || !out) return EGS_ERR_ARG;
    hipLaunchKernelGGL(k_gather_i32, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int)rows, src_index, kind,
                       clone_value_on, clone_value, in, out);
    return (int)hipGetLastError();
}

int egs_split_children(int64_t rows, const int32_t* src_index, const uint8_t* kind, const int32_t* split_rank, int n_split, const float* z,
                       const float* xyz_old, const float* scaling_old, const float* rotation_old, float* xyz_new, float* scaling_new,
                       void* stream) {
    if (rows < 0 || n_split < 0) return EGS_ERR_ARG;
    if (rows == 0 || n_split == 0) return 0;
    if (!src_index || !kind || !split_rank || !z || !xyz_old || !scaling_old || !rotation_old || !xyz_new || !scaling_new) return EGS_ERR_ARG;
    hipLaunchKernelGGL(k_split_children, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int)rows, src_index, kind,
                       split_rank, n_split, z, xyz_old, scaling_old, rotation_old, xyz_new, scaling_new);
    return (int)hipGetLastError();
}

}  // extern "C"
