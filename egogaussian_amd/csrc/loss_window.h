// loss_window.h -- the SSIM window of the image loss (loss.hip; also the loss gradient computed inside the backward blend, render_bwd.hip).
#pragma once
// gaussian(11, 1.5) normalised, as float32 (/root/reference/utils/loss_utils.py:66-68)
#define KW0 1.028380124e-03f
#define KW1 7.598758209e-03f
#define KW2 3.600077331e-02f
#define KW3 1.093606874e-01f
#define KW4 2.130055279e-01f
#define KW5 2.660117149e-01f
__device__ __forceinline__ constexpr float kwin(int k) {
    return k == 0 || k == 10 ? KW0 : k == 1 || k == 9 ? KW1 : k == 2 || k == 8 ? KW2 : k == 3 || k == 7 ? KW3 : k == 4 || k == 6 ? KW4 : KW5;
}

// The scalar loss from the per-strip partial sums, by ONE wave (fixed order, so the value is deterministic): used by the backward kernel when the caller deferred
// the loss value to it (egs_l1_ssim_forward with loss == NULL) -- a training step replayed from a graph reads the value only after
// the backward anyway, and every launch it does not make is ~4.5 us of GPU time.
__device__ __forceinline__ void wave_finish_loss(size_t nblocks, const float* __restrict__ partial, float w_l1, float w_ssim, float lambda,
                                                 float* __restrict__ loss, float* __restrict__ running_sum, unsigned lane) {
    float a = 0.f, b = 0.f;
    for (size_t i0 = 0; i0 < nblocks; i0 += 64 * 8) {                  // eight loads in flight per lane
        float2 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { const size_t i = i0 + (size_t)k * 64 + lane; v[k] = i < nblocks ? reinterpret_cast<const float2*>(partial)[i] : make_float2(0.f, 0.f); }
#pragma unroll
        for (int k = 0; k < 8; k++) { a += v[k].x; b += v[k].y; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    if (lane == 0) {
        const float v = w_l1 * a + lambda - w_ssim * b;
        if (loss) loss[0] = v;
        if (running_sum) running_sum[0] += v;
    }
}

