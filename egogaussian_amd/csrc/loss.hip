// loss.hip -- fused training image loss, forward and backward:
//     loss = (1 - lambda) * mean|x - y| + lambda * (1 - mean(SSIM_map(x, y)))
// as computed by the reference with ~6 depthwise 11x11 convolutions and ~25 elementwise kernels per step:
//     l1_loss, ssim           /root/reference/utils/loss_utils.py:57-107  (11x11 Gaussian window, sigma 1.5, zero padding,
//                                                                          C1 = 0.01^2, C2 = 0.03^2)
//     loss assembly, hand-mask gradient gate   /root/reference/trainers/train_static.py:91-95
// Forward: one pass per (channel, 16x16 tile): x and y tiles with a 5-pixel halo are staged in LDS once, the five
// windowed moments (E[x], E[y], E[x^2], E[y^2], E[xy]) are produced by a separable 11-tap blur inside LDS, the SSIM map
// value is reduced per block, and the three partial derivatives of the map (w.r.t. E[x], E[x^2], E[xy]) are stored.
// Backward: the same tiling blurs those three maps (the window is symmetric, so the adjoint of the blur is the blur)
// and adds the L1 term and the optional per-pixel gradient gate.  HBM-bound: 8 B in + 12 B out per pixel-channel
// forward, 20 B in + 4 B out backward.
#include "egs_common.h"

#define LT 16                 // output tile edge
#define HALO 5
#define LW (LT + 2 * HALO)    // 26

namespace {

// gaussian(11, 1.5) normalised, as float32 (utils/loss_utils.py:66-68)
__device__ __constant__ float kWin[11] = { 1.028380124e-03f, 7.598758209e-03f, 3.600077331e-02f, 1.093606874e-01f,
                                           2.130055279e-01f, 2.660117149e-01f, 2.130055279e-01f, 1.093606874e-01f,
                                           3.600077331e-02f, 7.598758209e-03f, 1.028380124e-03f };

__device__ __forceinline__ float block_sum_256(float v, float* lds4) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// grid: (tiles_x, tiles_y, C); block 256 = 16x16 outputs
__global__ __launch_bounds__(256) void k_l1_ssim_forward(int H, int W, const float* __restrict__ img,
                                                          const float* __restrict__ gt, float* __restrict__ partial,
                                                          float* __restrict__ dm_dmu1, float* __restrict__ dm_dexx,
                                                          float* __restrict__ dm_dexy) {
    __shared__ float sx[LW][LW + 1], sy[LW][LW + 1];
    __shared__ float hb[5][LW][LT + 1];          // horizontally blurred x, y, xx, yy, xy for 26 rows x 16 columns
    __shared__ float red[8];
    const int c = blockIdx.z, x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
    const size_t plane = (size_t)c * H * W;
    for (int t = threadIdx.x; t < LW * LW; t += 256) {
        const int ly = t / LW, lx = t % LW, gy = y0 + ly - HALO, gx = x0 + lx - HALO;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        sx[ly][lx] = in ? img[plane + (size_t)gy * W + gx] : 0.f;
        sy[ly][lx] = in ? gt[plane + (size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < LW * LT; t += 256) {
        const int ly = t / LT, lx = t % LT;
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = kWin[k], u = sx[ly][lx + k], v = sy[ly][lx + k];
            a = fmaf(w, u, a); b = fmaf(w, v, b); aa = fmaf(w, u * u, aa); bb = fmaf(w, v * v, bb); ab = fmaf(w, u * v, ab);
        }
        hb[0][ly][lx] = a; hb[1][ly][lx] = b; hb[2][ly][lx] = aa; hb[3][ly][lx] = bb; hb[4][ly][lx] = ab;
    }
    __syncthreads();
    const int lx = threadIdx.x % LT, ly = threadIdx.x / LT, gx = x0 + lx, gy = y0 + ly;
    float mu1 = 0.f, mu2 = 0.f, exx = 0.f, eyy = 0.f, exy = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = kWin[k];
        mu1 = fmaf(w, hb[0][ly + k][lx], mu1); mu2 = fmaf(w, hb[1][ly + k][lx], mu2);
        exx = fmaf(w, hb[2][ly + k][lx], exx); eyy = fmaf(w, hb[3][ly + k][lx], eyy); exy = fmaf(w, hb[4][ly + k][lx], exy);
    }
    const bool in = gx < W && gy < H;
    float l1 = 0.f, sm = 0.f;
    if (in) {
        const float C1 = 0.0001f, C2 = 0.0009f;
        const float s1 = exx - mu1 * mu1, s2 = eyy - mu2 * mu2, s12 = exy - mu1 * mu2;
        const float A = 2.f * mu1 * mu2 + C1, B = 2.f * s12 + C2, D = mu1 * mu1 + mu2 * mu2 + C1, E = s1 + s2 + C2;
        const float invDE = 1.f / (D * E);
        sm = A * B * invDE;
        // partial derivatives of the map holding the other windowed moments fixed
        const float dmu1 = (2.f * mu2 * (B - A)) * invDE - sm * (2.f * mu1 * (E - D)) * invDE;
        const float dexx = -sm / E;
        const float dexy = 2.f * A * invDE;
        const size_t p = plane + (size_t)gy * W + gx;
        dm_dmu1[p] = dmu1; dm_dexx[p] = dexx; dm_dexy[p] = dexy;
        l1 = fabsf(sx[ly + HALO][lx + HALO] - sy[ly + HALO][lx + HALO]);
    }
    const float tl1 = block_sum_256(l1, red);
    __syncthreads();
    const float tsm = block_sum_256(sm, red + 4);
    if (threadIdx.x == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * b] = tl1; partial[2 * b + 1] = tsm;
    }
}

__global__ __launch_bounds__(256) void k_l1_ssim_backward(int H, int W, const float* __restrict__ img,
                                                           const float* __restrict__ gt, float w_l1, float w_ssim,
                                                           const float* __restrict__ upstream, const float* __restrict__ gate,
                                                           const float* __restrict__ dm_dmu1, const float* __restrict__ dm_dexx,
                                                           const float* __restrict__ dm_dexy, float* __restrict__ dimg) {
    __shared__ float s[3][LW][LW + 1];
    __shared__ float hb[3][LW][LT + 1];
    const int c = blockIdx.z, x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
    const size_t plane = (size_t)c * H * W;
    for (int t = threadIdx.x; t < LW * LW; t += 256) {
        const int ly = t / LW, lx = t % LW, gy = y0 + ly - HALO, gx = x0 + lx - HALO;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t p = plane + (size_t)gy * W + gx;
        s[0][ly][lx] = in ? dm_dmu1[p] : 0.f; s[1][ly][lx] = in ? dm_dexx[p] : 0.f; s[2][ly][lx] = in ? dm_dexy[p] : 0.f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < LW * LT; t += 256) {
        const int ly = t / LT, lx = t % LT;
        float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = kWin[k];
            a = fmaf(w, s[0][ly][lx + k], a); b = fmaf(w, s[1][ly][lx + k], b); d = fmaf(w, s[2][ly][lx + k], d);
        }
        hb[0][ly][lx] = a; hb[1][ly][lx] = b; hb[2][ly][lx] = d;
    }
    __syncthreads();
    const int lx = threadIdx.x % LT, ly = threadIdx.x / LT, gx = x0 + lx, gy = y0 + ly;
    if (gx >= W || gy >= H) return;
    float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = kWin[k];
        a = fmaf(w, hb[0][ly + k][lx], a); b = fmaf(w, hb[1][ly + k][lx], b); d = fmaf(w, hb[2][ly + k][lx], d);
    }
    const size_t p = plane + (size_t)gy * W + gx;
    const float x = img[p], y = gt[p];
    const float diff = x - y;
    const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
    float g = w_l1 * sgn - w_ssim * (a + 2.f * x * b + y * d);     // d loss / d x ; loss uses (1 - mean SSIM)
    g *= upstream[0];
    if (gate) g *= gate[(size_t)gy * W + gx];
    dimg[p] = g;
}

// Adds up the per-block partial sums and assembles the scalar loss (one workgroup; deterministic order).
__global__ __launch_bounds__(1024) void k_l1_ssim_finish(size_t nblocks, const float* __restrict__ partial, float w_l1, float w_ssim,
                                                          float lambda, float* __restrict__ loss) {
    __shared__ float red[2][16];
    float a = 0.f, b = 0.f;
    for (size_t i = threadIdx.x; i < nblocks; i += 1024) { const float2 v = reinterpret_cast<const float2*>(partial)[i]; a += v.x; b += v.y; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ta = 0.f, tb = 0.f;
        for (int k = 0; k < 16; k++) { ta += red[0][k]; tb += red[1][k]; }
        loss[0] = w_l1 * ta + lambda - w_ssim * tb;                           // (1-l) mean|x-y| + l (1 - mean SSIM)
    }
}

}  // namespace

extern "C" {

size_t egs_l1_ssim_partial_count(int channels, int height, int width) {
    return (size_t)channels * ((height + LT - 1) / LT) * ((width + LT - 1) / LT) * 2;
}

int egs_l1_ssim_forward(int channels, int height, int width, const float* img, const float* gt, float lambda_dssim,
                        float* partial_sums, float* dm_dmu1, float* dm_dexx, float* dm_dexy, float* loss, void* stream) {
    if (channels <= 0 || height <= 0 || width <= 0 || !img || !gt || !partial_sums || !dm_dmu1 || !dm_dexx || !dm_dexy || !loss)
        return EGS_ERR_ARG;
    dim3 grid((width + LT - 1) / LT, (height + LT - 1) / LT, channels);
    hipLaunchKernelGGL(k_l1_ssim_forward, grid, dim3(256), 0, (hipStream_t)stream, height, width, img, gt, partial_sums,
                       dm_dmu1, dm_dexx, dm_dexy);
    const float n = (float)channels * (float)height * (float)width;
    hipLaunchKernelGGL(k_l1_ssim_finish, dim3(1), dim3(1024), 0, (hipStream_t)stream, (size_t)grid.x * grid.y * grid.z, partial_sums,
                       (1.f - lambda_dssim) / n, lambda_dssim / n, lambda_dssim, loss);
    return (int)hipGetLastError();
}

int egs_l1_ssim_backward(int channels, int height, int width, const float* img, const float* gt, float lambda_dssim,
                         const float* upstream_grad, const float* gate, const float* dm_dmu1, const float* dm_dexx,
                         const float* dm_dexy, float* dL_dimg, void* stream) {
    if (channels <= 0 || height <= 0 || width <= 0 || !img || !gt || !upstream_grad || !dm_dmu1 || !dm_dexx || !dm_dexy || !dL_dimg)
        return EGS_ERR_ARG;
    const float n = (float)channels * (float)height * (float)width;
    dim3 grid((width + LT - 1) / LT, (height + LT - 1) / LT, channels);
    hipLaunchKernelGGL(k_l1_ssim_backward, grid, dim3(256), 0, (hipStream_t)stream, height, width, img, gt,
                       (1.f - lambda_dssim) / n, lambda_dssim / n, upstream_grad, gate, dm_dmu1, dm_dexx, dm_dexy, dL_dimg);
    return (int)hipGetLastError();
}

}  // extern "C"
