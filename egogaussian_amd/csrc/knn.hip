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
            d = (base + t == i) ? FLT_MAX : d;                       // a point is not its own neighbour (by index)
            // insert d into the sorted triple
            const float n2 = fminf(b2, fmaxf(b1, d));
            const float n1 = fminf(b1, fmaxf(b0, d));
            b0 = fminf(b0, d); b1 = n1; b2 = n2;
        }
    }
    if (have) out[i] = (b0 + b1 + b2) / 3.0f;
}

}  // namespace

extern "C" int egs_knn3_mean_dist2(int N, const float* points, float* mean_dist2, void* stream) {
    if (N < 0) return EGS_ERR_ARG;
    if (N == 0) return 0;
    if (!points || !mean_dist2) return EGS_ERR_ARG;
    hipLaunchKernelGGL(k_knn3_mean_dist2, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, points, mean_dist2);
    return (int)hipGetLastError();
}
