// Micro-benchmark: VALU issue rate on gfx950 for plain / packed fp32 FMA, v_exp_f32, v_cndmask, ds_read broadcast.
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));
#define ITERS 4096
template <int MODE>
__global__ void k(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    const float b = seed * 0.5f, c = seed * 0.25f;
    const float2v bb = {b, b}, cc = {c, c};
    __shared__ float4 lds[64];
    lds[threadIdx.x & 63] = make_float4(a0, a1, a2, a3);
    __syncthreads();
    for (int i = 0; i < ITERS; i++) {
        if (MODE == 0) {   // 8 independent plain FMAs
            a0 = fmaf(a0, b, c); a1 = fmaf(a1, b, c); a2 = fmaf(a2, b, c); a3 = fmaf(a3, b, c);
            a4 = fmaf(a4, b, c); a5 = fmaf(a5, b, c); a6 = fmaf(a6, b, c); a7 = fmaf(a7, b, c);
        } else if (MODE == 1) {   // 8 independent packed FMAs
            p0 = __builtin_elementwise_fma(p0, bb, cc); p1 = __builtin_elementwise_fma(p1, bb, cc);
            p2 = __builtin_elementwise_fma(p2, bb, cc); p3 = __builtin_elementwise_fma(p3, bb, cc);
            p4 = __builtin_elementwise_fma(p4, bb, cc); p5 = __builtin_elementwise_fma(p5, bb, cc);
            p6 = __builtin_elementwise_fma(p6, bb, cc); p7 = __builtin_elementwise_fma(p7, bb, cc);
        } else if (MODE == 2) {   // dependent chain of 8 plain FMAs
            a0 = fmaf(a0, b, c); a0 = fmaf(a0, b, c); a0 = fmaf(a0, b, c); a0 = fmaf(a0, b, c);
            a0 = fmaf(a0, b, c); a0 = fmaf(a0, b, c); a0 = fmaf(a0, b, c); a0 = fmaf(a0, b, c);
        } else if (MODE == 3) {   // 8 independent v_exp_f32
            a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
            a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
        } else if (MODE == 4) {   // 4 x (cmp + cndmask)
            a0 = a0 > b ? c : a0 + 1.f; a1 = a1 > b ? c : a1 + 1.f; a2 = a2 > b ? c : a2 + 1.f; a3 = a3 > b ? c : a3 + 1.f;
        } else if (MODE == 5) {   // uniform-address ds_read_b128 x2 + 8 FMAs
            const float4 s = lds[i & 63], t = lds[(i + 7) & 63];
            a0 = fmaf(a0, s.x, t.x); a1 = fmaf(a1, s.y, t.y); a2 = fmaf(a2, s.z, t.z); a3 = fmaf(a3, s.w, t.w);
            a4 = fmaf(a4, s.x, t.y); a5 = fmaf(a5, s.y, t.z); a6 = fmaf(a6, s.z, t.w); a7 = fmaf(a7, s.w, t.x);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y +
        p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}
template <int MODE> double run(int waves_per_simd, float* d) {
    // one block of 256 threads = 4 waves = 1 wave per SIMD of a CU; waves_per_simd blocks per CU; 256 CUs
    dim3 grid(256 * waves_per_simd), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    const char* names[] = {"8 indep v_fma_f32", "8 indep v_pk_fma_f32", "8 dependent v_fma_f32", "8 indep v_exp_f32", "4x(v_cmp+v_cndmask+add)", "2 ds_read_b128(uniform)+8 fma"};
    const int ninstr[] = {8, 8, 8, 8, 12, 8};
    for (int w : {1, 2, 4, 8}) {
        double ms[6] = {run<0>(w, d), run<1>(w, d), run<2>(w, d), run<3>(w, d), run<4>(w, d), run<5>(w, d)};
        for (int m = 0; m < 6; m++) {
            // cycles per VALU instruction per SIMD, assuming 2.4 GHz: time * f / (ITERS * ninstr * waves_per_simd)
            double cyc = ms[m] * 1e-3 * 2.4e9 / ((double)ITERS * ninstr[m] * w);
            printf("waves/SIMD %d  %-30s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", w, names[m], ms[m], cyc);
        }
    }
    return 0;
}
