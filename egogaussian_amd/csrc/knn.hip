// knn.hip -- mean squared distance to the 3 nearest neighbours of every point.
// Replaces `simple_knn._C.distCUDA2`, an un-vendored CUDA submodule (/root/reference/.gitmodules:4-6) that the reference
// imports at /root/reference/scene/gaussian_model.py:21 and calls once per scene initialisation at :301 to seed the
// Gaussian scales (`clamp_min(distCUDA2(points), 1e-7)` then `log(sqrt(.))`, :301-312).
// The result is the exact 3-NN statistic, so any exact search matches it.  Upstream walks a Morton-ordered box
// hierarchy; on MI355X the whole all-pairs problem is cheaper than building one for a one-off call: 10^10 pair
// evaluations (N = 100k) take ~2 ms of the 157 TFLOP/s vector rate.  Each lane owns one query and keeps its three best
// squared distances in registers; candidate points stream through LDS in tiles read with a wave-uniform address
// (broadcast), 6 VALU + a 3-deep insertion per pair.
#include "egs_common.h"
#include <float.h>

#define KNN_TILE 1024

namespace {

__global__ __launch_bounds__(256) void k_knn3_mean_dist2(int N, const float* __restrict__ pts, float* __restrict__ out) {
    __shared__ float4 tile[KNN_TILE];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool have = i < N;
    const float qx = have ? pts[3 * (size_t)i] : 0.f, qy = have ? pts[3 * (size_t)i + 1] : 0.f, qz = have ? pts[3 * (size_t)i + 2] : 0.f;
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;                 // b0 <= b1 <= b2
    for (int base = 0; base < N; base += KNN_TILE) {
        __syncthreads();
        for (int t = threadIdx.x; t < KNN_TILE; t += 256) {
            const int j = base + t;
            tile[t] = j < N ? make_float4(pts[3 * (size_t)j], pts[3 * (size_t)j + 1], pts[3 * (size_t)j + 2], 0.f)
                            : make_float4(FLT_MAX, FLT_MAX, FLT_MAX, 0.f);
        }
        __syncthreads();
        const int cnt = min(KNN_TILE, N - base);
#pragma unroll 4
        for (int t = 0; t < cnt; t++) {
            const float4 p = tile[t];
            const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
            float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            d = (base + t == i || !(d <= FLT_MAX)) ? FLT_MAX : d;    // a point is not its own neighbour (by index); NaN / inf points are nobody's
            // insert d into the sorted triple
            const float n2 = fminf(b2, fmaxf(b1, d));
            const float n1 = fminf(b1, fmaxf(b0, d));
            b0 = fminf(b0, d); b1 = n1; b2 = n2;
        }
    }
    if (have) out[i] = (b0 + b1 + b2) / 3.0f;
}

// ---------------------------------------------------------------------------------------------------------------------
// Uniform-grid search for large clouds (the all-pairs kernel is O(N^2): 4 ms at 100k points, 0.4 s at 1 M, 40 s at 10 M).
// Same result, bit for bit: the three smallest values of the same fmaf expression, whatever the order they are met in.
//   1. bounding box of the finite points -> cell size so that a cell holds ~4 points, grid dimensions capped by `max_cells`;
//   2. counting sort of the points by cell (histogram with atomics, exclusive scan, fill through per-cell cursors);
//   3. one lane per point, in cell order (a wave's 64 queries share their candidate cells): scan the (2r+1)^3 cube of cells around
//      the query's cell, r = 1, 2, ...; every point outside the cube is farther than r cells, so the triple is final as soon as
//      its largest member is within (r * cell)^2 (with a 1e-4 slack for a point that rounding put one cell over).
// Upstream walks a Morton-ordered box hierarchy to the same end.
struct KnnGrid { float minx, miny, minz, cell, inv_cell; int nx, ny, nz; };

__device__ __forceinline__ bool knn_finite(float x, float y, float z) { return fabsf(x) <= FLT_MAX && fabsf(y) <= FLT_MAX && fabsf(z) <= FLT_MAX; }

__global__ __launch_bounds__(256) void k_knn_bounds(int N, const float* __restrict__ pts, float* __restrict__ partial) {
    __shared__ float red[6][4];
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)N; i += (size_t)gridDim.x * 256) {
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        if (!knn_finite(x, y, z)) continue;
        lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
        hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], d, 64)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], d, 64)); }
        if ((threadIdx.x & 63) == 0) { red[k][threadIdx.x >> 6] = lo[k]; red[3 + k][threadIdx.x >> 6] = hi[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const float* r = red[threadIdx.x];
        partial[blockIdx.x * 6 + threadIdx.x] = threadIdx.x < 3 ? fminf(fminf(r[0], r[1]), fminf(r[2], r[3])) : fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3]));
    }
}

__global__ void k_knn_grid_setup(int N, int n_partials, const float* __restrict__ partial, int max_cells, KnnGrid* __restrict__ g) {
    if (threadIdx.x != 0) return;
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int b = 0; b < n_partials; b++)
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], partial[b * 6 + k]); hi[k] = fmaxf(hi[k], partial[b * 6 + 3 + k]); }
    float ext[3];
    for (int k = 0; k < 3; k++) { if (!(lo[k] <= hi[k])) { lo[k] = 0.f; hi[k] = 0.f; } ext[k] = hi[k] - lo[k]; }
    const float longest = fmaxf(fmaxf(ext[0], ext[1]), fmaxf(ext[2], 1e-30f));
    for (int k = 0; k < 3; k++) ext[k] = fmaxf(ext[k], 1e-6f * longest);            // flat clouds: one thin layer of cells
    // ~4 points per cell; enlarge the cell until the grid fits max_cells
    float cell = cbrtf(ext[0] * ext[1] * ext[2] * 4.f / (float)(N > 0 ? N : 1));
    cell = fmaxf(cell, longest / 1024.f);
    int nx, ny, nz;
    for (;;) {
        nx = (int)fminf(ceilf(ext[0] / cell), 1024.f) + 1; ny = (int)fminf(ceilf(ext[1] / cell), 1024.f) + 1; nz = (int)fminf(ceilf(ext[2] / cell), 1024.f) + 1;
        if ((long long)nx * ny * nz <= (long long)max_cells) break;
        cell *= 1.26f;
    }
    g->minx = lo[0]; g->miny = lo[1]; g->minz = lo[2]; g->cell = cell; g->inv_cell = 1.f / cell; g->nx = nx; g->ny = ny; g->nz = nz;
}

__device__ __forceinline__ void knn_cell_of(const KnnGrid& g, float x, float y, float z, int& cx, int& cy, int& cz) {
    cx = min(max((int)floorf((x - g.minx) * g.inv_cell), 0), g.nx - 1);
    cy = min(max((int)floorf((y - g.miny) * g.inv_cell), 0), g.ny - 1);
    cz = min(max((int)floorf((z - g.minz) * g.inv_cell), 0), g.nz - 1);
}

__global__ __launch_bounds__(256) void k_knn_cell_count(int N, const float* __restrict__ pts, const KnnGrid* __restrict__ gp,
                                                        uint32_t* __restrict__ cell_of_point, uint32_t* __restrict__ count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const KnnGrid g = *gp;
    const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
    uint32_t c = 0xffffffffu;                                          // non-finite points are in no cell
    if (knn_finite(x, y, z)) { int cx, cy, cz; knn_cell_of(g, x, y, z, cx, cy, cz); c = (uint32_t)((cz * g.ny + cy) * g.nx + cx); atomicAdd(&count[c], 1u); }
    cell_of_point[i] = c;
}

__global__ __launch_bounds__(256) void k_knn_cell_fill(int N, const float* __restrict__ pts, const uint32_t* __restrict__ cell_of_point,
                                                       const uint32_t* __restrict__ start, uint32_t* __restrict__ fill, float4* __restrict__ sorted) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t c = cell_of_point[i];
    if (c == 0xffffffffu) return;
    const uint32_t pos = start[c] + atomicAdd(&fill[c], 1u);
    sorted[pos] = make_float4(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], __uint_as_float((uint32_t)i));
}

__global__ __launch_bounds__(256) void k_knn_query(int N, const float* __restrict__ pts, const uint32_t* __restrict__ cell_of_point,
                                                   const KnnGrid* __restrict__ gp, const uint32_t* __restrict__ start, const uint64_t* __restrict__ n_sorted,
                                                   const float4* __restrict__ sorted, float* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int placed = (int)*n_sorted;                                 // finite points, in cell order
    if (t >= N) return;
    if (t >= placed) {                                                 // the tail threads hand the non-finite points their answer
        return;
    }
    const KnnGrid g = *gp;
    const float4 q = sorted[t];
    const uint32_t self = __float_as_uint(q.w);
    int cx, cy, cz; knn_cell_of(g, q.x, q.y, q.z, cx, cy, cz);
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;                    // the three smallest distances so far: kept across radii
    const int rmax = max(max(g.nx, g.ny), g.nz);
    auto scan = [&](uint32_t s, uint32_t e) {
        for (uint32_t k = s; k < e; k++) {
            const float4 p = sorted[k];
            const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
            float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            d = (__float_as_uint(p.w) == self || !(d <= FLT_MAX)) ? FLT_MAX : d;
            const float n2 = fminf(b2, fmaxf(b1, d));
            const float n1 = fminf(b1, fmaxf(b0, d));
            b0 = fminf(b0, d); b1 = n1; b2 = n2;
        }
    };
    // Radius r visits only the SHELL of cells at Chebyshev distance r from the query's cell (r = 0: the cell itself); an isolated
    // point therefore costs O(r^3) cell visits in total, not O(r^4).
    for (int r = 0;; r++) {
        const int z0 = max(cz - r, 0), z1 = min(cz + r, g.nz - 1), y0 = max(cy - r, 0), y1 = min(cy + r, g.ny - 1);
        const int x0 = max(cx - r, 0), x1 = min(cx + r, g.nx - 1);
        for (int z = z0; z <= z1; z++)
            for (int y = y0; y <= y1; y++) {
                const uint32_t row = (uint32_t)((z * g.ny + y) * g.nx);
                if (abs(z - cz) == r || abs(y - cy) == r) {
                    scan(start[row + x0], start[row + x1 + 1]);                // a face row: cells x0..x1 are contiguous in cell order
                } else {                                                       // an inner row: its two end cells
                    if (cx - r >= 0) scan(start[row + cx - r], start[row + cx - r + 1]);
                    if (cx + r <= g.nx - 1) scan(start[row + cx + r], start[row + cx + r + 1]);
                }
            }
        // Every point nearer than (r - 2 e) cells is inside the cube just completed, e = the rounding error of a cell coordinate:
        // floorf((x - min) * inv_cell) is off by at most ~index * 2^-22 <= 2.5e-4 cells at index 1024 (the longest axis has at most
        // 1025 cells), for the query's own cell and for the neighbour's.  1e-3 covers both and the product below.
        const float reach = ((float)r - 1e-3f) * g.cell;
        if ((r >= 1 && b2 <= reach * reach) || r >= rmax) break;
    }
    out[self] = (b0 + b1 + b2) / 3.0f;
}

__global__ __launch_bounds__(256) void k_knn_nonfinite(int N, const uint32_t* __restrict__ cell_of_point, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N && cell_of_point[i] == 0xffffffffu) out[i] = (FLT_MAX + FLT_MAX + FLT_MAX) / 3.0f;      // = inf, what the all-pairs kernel gives
}

}  // namespace

static int knn_max_cells(int N) { long long m = (long long)N / 2 + 64; return (int)(m > (1 << 26) ? (1 << 26) : m); }
static size_t knn_al(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t egs_knn3_grid_scratch_bytes(int N) {
    if (N <= 0) return 0;
    const size_t mc = (size_t)knn_max_cells(N);
    return knn_al((size_t)N * 4) + 2 * knn_al((mc + 1) * 4) + knn_al(mc * 4) + knn_al((size_t)N * 16) + knn_al(1024 * 6 * 4) + knn_al(64) +
           knn_al(egs_scan_scratch_elems(mc + 1) * 4) + 256;
}

extern "C" int egs_knn3_grid(int N, const float* points, float* mean_dist2, void* scratch, void* stream) {
    if (N < 0) return EGS_ERR_ARG;
    if (N == 0) return 0;
    if (!points || !mean_dist2 || !scratch) return EGS_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int mc = knn_max_cells(N);
    char* b = (char*)scratch;
    uint32_t* cell_of_point = (uint32_t*)b; b += knn_al((size_t)N * 4);
    uint32_t* count = (uint32_t*)b;         b += knn_al(((size_t)mc + 1) * 4);
    uint32_t* start = (uint32_t*)b;         b += knn_al(((size_t)mc + 1) * 4);
    uint32_t* fill = (uint32_t*)b;          b += knn_al((size_t)mc * 4);
    float4* sorted = (float4*)b;            b += knn_al((size_t)N * 16);
    float* partial = (float*)b;             b += knn_al(1024 * 6 * 4);
    KnnGrid* grid = (KnnGrid*)b;            b += knn_al(64);
    uint32_t* scan_scratch = (uint32_t*)b;  b += knn_al(egs_scan_scratch_elems((size_t)mc + 1) * 4);
    uint64_t* total = (uint64_t*)b;
    const int nb = (N + 255) / 256, nbb = nb < 1024 ? nb : 1024;
    hipError_t e;
    if ((e = egs_launch_zero_u32(count, (size_t)mc + 1, s)) != hipSuccess) return (int)e;
    if ((e = egs_launch_zero_u32(fill, (size_t)mc, s)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_knn_bounds, dim3(nbb), dim3(256), 0, s, N, points, partial);
    hipLaunchKernelGGL(k_knn_grid_setup, dim3(1), dim3(64), 0, s, N, nbb, partial, mc, grid);
    hipLaunchKernelGGL(k_knn_cell_count, dim3(nb), dim3(256), 0, s, N, points, grid, cell_of_point, count);
    if ((e = egs_launch_scan_u32(count, start, (size_t)mc + 1, 0, scan_scratch, total, s)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_knn_cell_fill, dim3(nb), dim3(256), 0, s, N, points, cell_of_point, start, fill, sorted);
    hipLaunchKernelGGL(k_knn_query, dim3(nb), dim3(256), 0, s, N, points, cell_of_point, grid, start, total, sorted, mean_dist2);
    hipLaunchKernelGGL(k_knn_nonfinite, dim3(nb), dim3(256), 0, s, N, cell_of_point, mean_dist2);
    return (int)hipGetLastError();
}

extern "C" int egs_knn3_mean_dist2(int N, const float* points, float* mean_dist2, void* stream) {
    if (N < 0) return EGS_ERR_ARG;
    if (N == 0) return 0;
    if (!points || !mean_dist2) return EGS_ERR_ARG;
    hipLaunchKernelGGL(k_knn3_mean_dist2, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, points, mean_dist2);
    return (int)hipGetLastError();
}
