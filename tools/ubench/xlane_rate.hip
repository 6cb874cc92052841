// Issue cost of the cross-lane instructions the backward blend's reduction uses (8 waves/SIMD, 8 independent chains).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
#define ITERS 4096
template <int MODE>
__global__ void k(float* out, float seed) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x + i;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) a[i] = a[i] + 1.0f;                                                   // plain add
            else if (MODE == 1) a[i] = a[i] + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(a[i]), 0xB1, 0xf, 0xf, true));   // quad_perm dpp add
            else if (MODE == 2) a[i] = a[i] + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(a[i]), 0x140, 0xf, 0xf, true));  // row_mirror dpp add
            else if (MODE == 3) { uint2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[i]), __float_as_uint(a[(i + 1) & 7]), false, false); a[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
            else if (MODE == 4) { uint2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(a[(i + 1) & 7]), false, false); a[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
            else if (MODE == 5) a[i] = a[i] + __shfl_xor(a[i], 16, 64);                           // ds_bpermute path
            else if (MODE == 6) a[i] = a[i] + __builtin_amdgcn_rcpf(a[i]);                        // rcp + add
            else if (MODE == 7) a[i] = a[i] + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(a[i]), it & 63));
        }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> double run(float* d) {
    dim3 grid(256 * 8), block(256);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, 1.0001f); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, 1.0001f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    const char* names[] = {"v_add_f32", "v_add_f32_dpp quad_perm", "v_add_f32_dpp row_mirror", "permlane32_swap + 1 add", "permlane16_swap + 1 add", "shfl_xor (ds_bpermute) + add", "v_rcp_f32 + add", "v_readlane + add"};
    double ms[8] = {run<0>(d), run<1>(d), run<2>(d), run<3>(d), run<4>(d), run<5>(d), run<6>(d), run<7>(d)};
    for (int m = 0; m < 8; m++)
        printf("%-32s %8.3f ms -> %.2f cycles per group per SIMD (2.4 GHz, 8 waves/SIMD)\n", names[m], ms[m], ms[m] * 1e-3 * 2.4e9 / ((double)ITERS * 8 * 8));
    return 0;
}
