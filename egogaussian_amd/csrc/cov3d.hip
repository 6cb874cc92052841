// cov3d.hip -- fused 3D covariance producer, forward and backward.
// Replaces the chain of PyTorch ops the reference runs before EVERY rasterizer call because it forces
// pipe.compute_cov3D_python = True (/root/reference/train.py:49):
//   build_rotation / build_scaling_rotation / strip_symmetric     /root/reference/utils/general_utils.py:110-156
//   covariance activation and its object-rotated variant          /root/reference/scene/gaussian_model.py:29-33,46-63
// i.e.  q <- q/|q| ;  L = R(q) diag(mod * s) ;  [L <- M L for selected rows] ;  Sigma = L L^T ;  out = 6 unique entries.
// In PyTorch that is ~15 kernels forward and ~40 backward over [N,3,3] temporaries (and, with the reference's own
// `L @ L.transpose`, a 500k-batch 3x3 GEMM that rocBLAS runs for milliseconds).  Here: one streaming kernel each way,
// 28 B in / 24 B out per Gaussian forward, 52 B in / 28 B out backward; HBM-bound.
#include "egs_common.h"

namespace {

__device__ __forceinline__ void rot_from_unit_quat(const float* q, float* R) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

__device__ __forceinline__ void mat3_mul(const float* A, const float* B, float* C) {      // C = A B
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// `log_scaling`: the scaling tensor holds the RAW parameters (log of the scales, /root/reference/scene/gaussian_model.py:36
// scaling_activation = torch.exp); the exponential and its derivative are then applied here instead of by two more
// elementwise kernels over [N,3] per step.
__global__ __launch_bounds__(256) void k_cov3d_forward(int N, const float* __restrict__ scaling, int log_scaling, float mod,
                                                        const float* __restrict__ rotation, const float* __restrict__ M,
                                                        const uint8_t* __restrict__ sel, float* __restrict__ cov,
                                                        const float* __restrict__ opacity_raw, float* __restrict__ opacity) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (opacity_raw) opacity[i] = 1.f / (1.f + expf(-opacity_raw[i]));           // opacity_activation = torch.sigmoid (gaussian_model.py:40)
    float q[4] = { rotation[4 * i], rotation[4 * i + 1], rotation[4 * i + 2], rotation[4 * i + 3] };
    const float inv = 1.f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
    float R[9]; rot_from_unit_quat(q, R);
    float s3[3] = { scaling[3 * i], scaling[3 * i + 1], scaling[3 * i + 2] };
    if (log_scaling) { s3[0] = expf(s3[0]); s3[1] = expf(s3[1]); s3[2] = expf(s3[2]); }
    const float sc[3] = { mod * s3[0], mod * s3[1], mod * s3[2] };
    float L[9];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int k = 0; k < 3; k++) L[3 * a + k] = R[3 * a + k] * sc[k];
    if (M && (!sel || sel[i])) {
        float Mm[9], L2[9];
#pragma unroll
        for (int k = 0; k < 9; k++) Mm[k] = M[k];
        mat3_mul(Mm, L, L2);
#pragma unroll
        for (int k = 0; k < 9; k++) L[k] = L2[k];
    }
    float* o = cov + 6 * (size_t)i;
    o[0] = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
    o[1] = L[0] * L[3] + L[1] * L[4] + L[2] * L[5];
    o[2] = L[0] * L[6] + L[1] * L[7] + L[2] * L[8];
    o[3] = L[3] * L[3] + L[4] * L[4] + L[5] * L[5];
    o[4] = L[3] * L[6] + L[4] * L[7] + L[5] * L[8];
    o[5] = L[6] * L[6] + L[7] * L[7] + L[8] * L[8];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ __launch_bounds__(256) void k_cov3d_backward(int N, const float* __restrict__ scaling, int log_scaling, float mod,
                                                         const float* __restrict__ rotation, const float* __restrict__ M,
                                                         const uint8_t* __restrict__ sel, float row0_mult,
                                                         const float* __restrict__ row0_mult_dev,
                                                         const float* __restrict__ dcov, float* __restrict__ dscaling,
                                                         float* __restrict__ drotation, float* __restrict__ dM_partial,
                                                         const float* __restrict__ opacity, const float* __restrict__ dopacity,
                                                         float* __restrict__ dopacity_raw) {
    __shared__ float wsum[4][9];
    if (opacity && blockIdx.x * blockDim.x + threadIdx.x < (unsigned)N) {
        const int j = blockIdx.x * blockDim.x + threadIdx.x;
        const float o = opacity[j];
        dopacity_raw[j] = dopacity[j] * ((1.f - o) * o);                        // sigmoid backward: grad * (1 - y) * y
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float gM[9] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    if (i < N) {
        const float q0[4] = { rotation[4 * i], rotation[4 * i + 1], rotation[4 * i + 2], rotation[4 * i + 3] };
        const float inv = 1.f / sqrtf(q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3]);
        const float q[4] = { q0[0] * inv, q0[1] * inv, q0[2] * inv, q0[3] * inv };
        float R[9]; rot_from_unit_quat(q, R);
        float s3[3] = { scaling[3 * i], scaling[3 * i + 1], scaling[3 * i + 2] };
        if (log_scaling) { s3[0] = expf(s3[0]); s3[1] = expf(s3[1]); s3[2] = expf(s3[2]); }
        const float sc[3] = { mod * s3[0], mod * s3[1], mod * s3[2] };
        float L0[9], L[9];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int k = 0; k < 3; k++) L0[3 * a + k] = R[3 * a + k] * sc[k];
        const bool moved = M && (!sel || sel[i]);
        float Mm[9];
        if (moved) {
#pragma unroll
            for (int k = 0; k < 9; k++) Mm[k] = M[k];
            mat3_mul(Mm, L0, L);
        } else {
#pragma unroll
            for (int k = 0; k < 9; k++) L[k] = L0[k];
        }
        const float* g6 = dcov + 6 * (size_t)i;
        const float Gs[9] = { g6[0], 0.5f * g6[1], 0.5f * g6[2], 0.5f * g6[1], g6[3], 0.5f * g6[4], 0.5f * g6[2], 0.5f * g6[4], g6[5] };
        float gL[9];                                                   // dL/dL = 2 Gs L
        mat3_mul(Gs, L, gL);
        const float mult = (moved && i == 0) ? (row0_mult_dev ? row0_mult_dev[0] : row0_mult) : 1.f;   // the reference's duplicated-index gradient (covariance.py)
#pragma unroll
        for (int k = 0; k < 9; k++) gL[k] *= 2.f * mult;
        float gL0[9];
        if (moved) {
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    gM[3 * a + b] = gL[3 * a] * L0[3 * b] + gL[3 * a + 1] * L0[3 * b + 1] + gL[3 * a + 2] * L0[3 * b + 2];   // gL L0^T
                    gL0[3 * a + b] = Mm[a] * gL[b] + Mm[3 + a] * gL[3 + b] + Mm[6 + a] * gL[6 + b];                          // M^T gL
                }
        } else {
#pragma unroll
            for (int k = 0; k < 9; k++) gL0[k] = gL[k];
        }
        float gR[9];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float ds = mod * (gL0[k] * R[k] + gL0[3 + k] * R[3 + k] + gL0[6 + k] * R[6 + k]);
            dscaling[3 * i + k] = log_scaling ? ds * s3[k] : ds;           // d/d raw = d/d scale * exp(raw)
#pragma unroll
            for (int a = 0; a < 3; a++) gR[3 * a + k] = gL0[3 * a + k] * sc[k];
        }
        const float r = q[0], x = q[1], y = q[2], z = q[3];
        float gq[4];
        gq[0] = 2.f * (-z * gR[1] + y * gR[2] + z * gR[3] - x * gR[5] - y * gR[6] + x * gR[7]);
        gq[1] = 2.f * (y * gR[1] + z * gR[2] + y * gR[3] - 2.f * x * gR[4] - r * gR[5] + z * gR[6] + r * gR[7] - 2.f * x * gR[8]);
        gq[2] = 2.f * (-2.f * y * gR[0] + x * gR[1] + r * gR[2] + x * gR[3] + z * gR[5] - r * gR[6] + z * gR[7] - 2.f * y * gR[8]);
        gq[3] = 2.f * (-2.f * z * gR[0] - r * gR[1] + x * gR[2] + r * gR[3] - 2.f * z * gR[4] + y * gR[5] + x * gR[6] + y * gR[7]);
        const float dot = q[0] * gq[0] + q[1] * gq[1] + q[2] * gq[2] + q[3] * gq[3];      // back through q/|q|
#pragma unroll
        for (int k = 0; k < 4; k++) drotation[4 * i + k] = (gq[k] - q[k] * dot) * inv;
    }
    // d/dM is a sum over every Gaussian.  One float atomic per wave and component (70k device-scope atomics on nine words
    // of one cache line at 500k Gaussians) serialises at the memory side: 896 us for this kernel instead of 9.  Workgroup
    // partial sums are written out plainly and added up by k_cov3d_dm_finish.
    if (dM_partial) {                                                      // workgroup-uniform branch
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const float s = wave_sum(gM[k]);
            if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6][k] = s;
        }
        __syncthreads();
        if (threadIdx.x < 9) dM_partial[(size_t)blockIdx.x * 9 + threadIdx.x] = (wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + (wsum[2][threadIdx.x] + wsum[3][threadIdx.x]);
    }
}

__global__ __launch_bounds__(1024) void k_cov3d_dm_finish(int nblocks, const float* __restrict__ partial, float* __restrict__ dM) {
    __shared__ float red[16];
    for (int k = 0; k < 9; k++) {
        float v = 0.f;
        for (int b = threadIdx.x; b < nblocks; b += 1024) v += partial[(size_t)b * 9 + k];
        v = wave_sum(v);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) { float t = 0.f; for (int w = 0; w < 16; w++) t += red[w]; dM[k] = t; }
        __syncthreads();
    }
}

}  // namespace

extern "C" {

int egs_cov3d_forward(int N, const float* scaling, int scaling_is_log, float scale_modifier, const float* rotation, const float* M9,
                      const uint8_t* selected, float* cov6, const float* opacity_raw, float* opacity, void* stream) {
    if (N < 0) return EGS_ERR_ARG;
    if (N == 0) return 0;
    if (!scaling || !rotation || !cov6 || (opacity_raw && !opacity)) return EGS_ERR_ARG;
    hipLaunchKernelGGL(k_cov3d_forward, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, scaling, scaling_is_log, scale_modifier,
                       rotation, M9, selected, cov6, opacity_raw, opacity);
    return (int)hipGetLastError();
}

size_t egs_cov3d_dm_scratch_floats(int N) { return N > 0 ? (size_t)((N + 255) / 256) * 9 : 0; }

int egs_cov3d_backward(int N, const float* scaling, int scaling_is_log, float scale_modifier, const float* rotation, const float* M9,
                       const uint8_t* selected, float row0_grad_mult, const float* row0_grad_mult_dev, const float* dL_dcov6, float* dL_dscaling,
                       float* dL_drotation, float* dL_dM9, float* dM_scratch, const float* opacity, const float* dL_dopacity,
                       float* dL_dopacity_raw, void* stream) {
    if (N < 0) return EGS_ERR_ARG;
    if (dL_dM9 && N == 0) { hipError_t e = egs_launch_zero_u32((uint32_t*)dL_dM9, 9, (hipStream_t)stream); if (e != hipSuccess) return (int)e; }
    if (N == 0) return 0;
    if (!scaling || !rotation || !dL_dcov6 || !dL_dscaling || !dL_drotation) return EGS_ERR_ARG;
    if (M9 && dL_dM9 && !dM_scratch) return EGS_ERR_ARG;
    if (opacity && (!dL_dopacity || !dL_dopacity_raw)) return EGS_ERR_ARG;
    hipLaunchKernelGGL(k_cov3d_backward, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, scaling, scaling_is_log, scale_modifier,
                       rotation, M9, selected, row0_grad_mult, row0_grad_mult_dev, dL_dcov6, dL_dscaling, dL_drotation, (M9 && dL_dM9) ? dM_scratch : nullptr,
                       opacity, dL_dopacity, dL_dopacity_raw);
    if (dL_dM9) {
        if (M9) hipLaunchKernelGGL(k_cov3d_dm_finish, dim3(1), dim3(1024), 0, (hipStream_t)stream, (N + 255) / 256, dM_scratch, dL_dM9);
        else { hipError_t e = egs_launch_zero_u32((uint32_t*)dL_dM9, 9, (hipStream_t)stream); if (e != hipSuccess) return (int)e; }
    }
    return (int)hipGetLastError();
}

}  // extern "C"
