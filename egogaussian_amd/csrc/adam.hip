// adam.hip -- one-launch multi-tensor Adam step (optimizer part).
// The reference steps five parameter groups with torch.optim.Adam(l, lr=0.0, eps=1e-15)
// (/root/reference/scene/gaussian_model.py:180-198, step at /root/reference/trainers/train_static.py:137): per group
// PyTorch issues its own kernels (5 x ~45 us on MI355X even with fused=True at 500k Gaussians).  Here every tensor of
// every group is updated by ONE streaming kernel: 16 B in (p, g, m, v) + 12 B out per element, float4-vectorised.
// Semantics = torch.optim.Adam with weight_decay = 0, amsgrad = False, maximize = False:
//   m <- m + (1 - b1)(g - m);  v <- b2 v + (1 - b2) g^2;  p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "egs_common.h"
#include <math.h>

#define ADAM_MAX_TENSORS 16
#define ADAM_EPB 4096            // elements per workgroup (256 threads x 4 float4)

namespace {

struct AdamArgs {
    float* p[ADAM_MAX_TENSORS]; const float* g[ADAM_MAX_TENSORS]; float* m[ADAM_MAX_TENSORS]; float* v[ADAM_MAX_TENSORS];
    long long numel[ADAM_MAX_TENSORS]; unsigned block_start[ADAM_MAX_TENSORS + 1];
    float step_size[ADAM_MAX_TENSORS], inv_bc2_sqrt[ADAM_MAX_TENSORS];
    float* step_dev[ADAM_MAX_TENSORS]; const float* lr_dev[ADAM_MAX_TENSORS];         // capturable variant: step out / lr in, on the device
    unsigned* counter[ADAM_MAX_TENSORS];                                               // capturable variant: launches x workgroups so far
    int row_floats[ADAM_MAX_TENSORS];                                                  // capacity-sized models: floats per row (0 = whole tensor)
    const unsigned* skip; const int* active_rows;                                      // device words (may be NULL), include/egs_raster.h
    int n; float b1, b2, eps;
};

__global__ void k_adam_tick(EgsAdamTick t) { egs_adam_tick(t, threadIdx.x); }

__global__ __launch_bounds__(256) void k_adam(AdamArgs a) {
    if (a.skip && *a.skip) return;                                   // the frame behind these gradients overflowed its instance capacity: no step at all
    int t = 0;
#pragma unroll 1
    while (t + 1 < a.n && blockIdx.x >= a.block_start[t + 1]) t++;
    const long long base = (long long)(blockIdx.x - a.block_start[t]) * ADAM_EPB;
    long long n = a.numel[t];
    if (a.active_rows && a.row_floats[t]) n = min(n, (long long)max(*a.active_rows, 0) * a.row_floats[t]);      // live rows only (every workgroup still counts its step)
    float* __restrict__ p = a.p[t]; const float* __restrict__ g = a.g[t]; float* __restrict__ m = a.m[t]; float* __restrict__ v = a.v[t];
    float ss = a.step_size[t], ib = a.inv_bc2_sqrt[t];
    if (a.counter[t]) {                                              // workgroup-uniform: hipGraph replays see the current step / lr
        // The step number is kept without any cross-workgroup traffic: every workgroup of tensor t owns one word of
        // counter[t] holding the steps it has taken, reads it and writes it back plus one.  (Anything shared costs: ~1.7k
        // device-scope atomics per launch on a handful of words -- a last-workgroup ticket, or a fire-and-forget
        // launches-x-workgroups counter -- serialise at ~20 ns each and added 7 us to this 33 us kernel.)
        // The bias corrections use the same double-precision expressions as the host computes for the non-capturable
        // launch, so both variants take bit-identical steps; one thread per workgroup does the two pow() calls.
        __shared__ float s_ss, s_ib;
        if (threadIdx.x == 0) {
            unsigned* mine = a.counter[t] + (blockIdx.x - a.block_start[t]);
            const unsigned taken = *mine;
            *mine = taken + 1u;
            const double st = (double)(taken + 1u);
            s_ss = (float)((double)a.lr_dev[t][0] / (1.0 - pow((double)a.b1, st)));
            s_ib = (float)(1.0 / sqrt(1.0 - pow((double)a.b2, st)));
            if (blockIdx.x == a.block_start[t]) a.step_dev[t][0] = (float)st;        // torch's state["step"], for the host to read
        }
        __syncthreads();
        ss = s_ss; ib = s_ib;
    }
    const bool vec = (((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long i = base + ((long long)k * 256 + threadIdx.x) * 4;
        if (i >= n) break;
        if (vec && i + 4 <= n) {
            float4 P = *reinterpret_cast<float4*>(p + i), M = *reinterpret_cast<float4*>(m + i), V = *reinterpret_cast<float4*>(v + i);
            const float4 G = *reinterpret_cast<const float4*>(g + i);
            egs_adam1(P.x, G.x, M.x, V.x, a.b1, a.b2, a.eps, ss, ib); egs_adam1(P.y, G.y, M.y, V.y, a.b1, a.b2, a.eps, ss, ib);
            egs_adam1(P.z, G.z, M.z, V.z, a.b1, a.b2, a.eps, ss, ib); egs_adam1(P.w, G.w, M.w, V.w, a.b1, a.b2, a.eps, ss, ib);
            *reinterpret_cast<float4*>(p + i) = P; *reinterpret_cast<float4*>(m + i) = M; *reinterpret_cast<float4*>(v + i) = V;
        } else {
            for (long long j = i; j < i + 4 && j < n; j++) {
                float P = p[j], M = m[j], V = v[j];
                egs_adam1(P, g[j], M, V, a.b1, a.b2, a.eps, ss, ib);
                p[j] = P; m[j] = M; v[j] = V;
            }
        }
    }
}

}  // namespace

hipError_t egs_launch_adam_tick(const EgsAdamTick& tick, hipStream_t s) {
    hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(64), 0, s, tick);
    return hipGetLastError();
}

static int adam_impl(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                     float* const* exp_avg_sq, const int64_t* numels, const float* lrs, const int64_t* steps,
                     float* const* step_dev, const float* const* lr_dev, unsigned* const* counters, float beta1, float beta2, float eps,
                     const uint32_t* skip_flag, const int32_t* active_rows, const int32_t* row_floats, void* stream) {
    if (n_tensors < 0) return EGS_ERR_ARG;
    const bool dev = step_dev != nullptr;
    if (n_tensors && (!params || !grads || !exp_avg || !exp_avg_sq || !numels)) return EGS_ERR_ARG;
    if (n_tensors && (dev ? (!lr_dev || !counters) : (!lrs || !steps))) return EGS_ERR_ARG;
    for (int t0 = 0; t0 < n_tensors; t0 += ADAM_MAX_TENSORS) {
        AdamArgs a; a.n = 0; a.b1 = beta1; a.b2 = beta2; a.eps = eps;
        a.skip = skip_flag; a.active_rows = (active_rows && row_floats) ? active_rows : nullptr;
        unsigned blocks = 0;
        for (int t = t0; t < n_tensors && a.n < ADAM_MAX_TENSORS; t++) {
            if (numels[t] <= 0) continue;
            if (!params[t] || !grads[t] || !exp_avg[t] || !exp_avg_sq[t]) return EGS_ERR_ARG;
            if (dev ? (!step_dev[t] || !lr_dev[t] || !counters[t]) : steps[t] < 1) return EGS_ERR_ARG;
            const int k = a.n++;
            a.step_dev[k] = dev ? step_dev[t] : nullptr; a.lr_dev[k] = dev ? lr_dev[t] : nullptr; a.counter[k] = dev ? counters[t] : nullptr;
            a.p[k] = params[t]; a.g[k] = grads[t]; a.m[k] = exp_avg[t]; a.v[k] = exp_avg_sq[t]; a.numel[k] = numels[t];
            a.row_floats[k] = (active_rows && row_floats) ? row_floats[t] : 0;
            a.block_start[k] = blocks;
            blocks += (unsigned)((numels[t] + ADAM_EPB - 1) / ADAM_EPB);
            if (!dev) {
                const double bc1 = 1.0 - pow((double)beta1, (double)steps[t]), bc2 = 1.0 - pow((double)beta2, (double)steps[t]);
                a.step_size[k] = (float)((double)lrs[t] / bc1);
                a.inv_bc2_sqrt[k] = (float)(1.0 / sqrt(bc2));
            } else { a.step_size[k] = 0.f; a.inv_bc2_sqrt[k] = 0.f; }
        }
        a.block_start[a.n] = blocks;
        if (blocks == 0) continue;
        hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

extern "C" int egs_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                             float* const* exp_avg_sq, const int64_t* numels, const float* lrs, const int64_t* steps,
                             float beta1, float beta2, float eps, void* stream) {
    return adam_impl(n_tensors, params, grads, exp_avg, exp_avg_sq, numels, lrs, steps, nullptr, nullptr, nullptr, beta1, beta2, eps, nullptr, nullptr,
                     nullptr, stream);
}

// hipGraph-capturable variant: the learning rate of every tensor is read from a device scalar, and the step number from
// per-workgroup device counters the launch itself advances (see k_adam): counters[t] is uint32[egs_adam_workgroups(numel_t)],
// every word = the number of steps tensor t has taken, on entry and again on exit.  step_dev[t] (float[1]) is
// WRITTEN with the step just taken.  A captured step therefore keeps counting across replays with no other kernel and no
// workgroup waiting on another, and the host may edit the learning rates between replays.
extern "C" int64_t egs_adam_workgroups(int64_t numel) { return numel <= 0 ? 0 : (numel + ADAM_EPB - 1) / ADAM_EPB; }

extern "C" int egs_adam_step_capturable(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                                        float* const* exp_avg_sq, const int64_t* numels, float* const* step_dev,
                                        const float* const* lr_dev, uint32_t* const* counters, float beta1, float beta2, float eps,
                                        const uint32_t* skip_flag, const int32_t* active_rows, const int32_t* row_floats, void* stream) {
    if (n_tensors && !step_dev) return EGS_ERR_ARG;
    if (row_floats) for (int t = 0; t < n_tensors; t++) if (row_floats[t] < 0) return EGS_ERR_ARG;
    return adam_impl(n_tensors, params, grads, exp_avg, exp_avg_sq, numels, nullptr, nullptr, step_dev, lr_dev, counters, beta1, beta2, eps,
                     skip_flag, active_rows, row_floats, stream);
}
