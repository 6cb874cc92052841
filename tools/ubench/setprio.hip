// Micro-benchmark: does s_setprio change which waves a SIMD issues first?  Eight waves per SIMD run the same dependent-FMA loop; the
// waves of the odd workgroups raise their priority to 3.  Each wave logs its end time; if priority counts, the prio-3 waves end first.
// hipcc --offload-arch=gfx950 -O3 setprio.hip -o setprio && ./setprio
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* t_end, int* prio_of, int use_prio, float seed) {
    const int hi = (blockIdx.x / 256) & 1;                 // block b -> XCD b % 8, CU (b / 8) % 32: blocks 256 apart share a CU; alternate groups
    if (use_prio && hi) __builtin_amdgcn_s_setprio(3);
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    const float b = seed * 0.5f, c = seed * 0.25f;
    for (int i = 0; i < 20000; i++) {                       // 4 independent chains: a wave alone fills part of the issue slots only
        a0 = fmaf(a0, b, c); a1 = fmaf(a1, b, c); a2 = fmaf(a2, b, c); a3 = fmaf(a3, b, c);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
    if ((threadIdx.x & 63) == 0) { t_end[blockIdx.x * 4 + threadIdx.x / 64] = wall_clock64(); prio_of[blockIdx.x * 4 + threadIdx.x / 64] = hi; }
}
int main() {
    const int blocks = 256 * 8;                             // 8 blocks (32 waves) per CU: 8 waves per SIMD
    float* out; unsigned long long* te; int* pr;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&te, blocks * 4 * 8); hipMalloc(&pr, blocks * 4 * 4);
    for (int use = 0; use < 2; use++) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, te, pr, use, 1.0001f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> t(blocks * 4); std::vector<int> p(blocks * 4);
        hipMemcpy(t.data(), te, t.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(p.data(), pr, p.size() * 4, hipMemcpyDeviceToHost);
        unsigned long long t0 = *std::min_element(t.begin(), t.end());
        double s[2] = {0, 0}, mx[2] = {0, 0}; int n[2] = {0, 0};
        for (size_t i = 0; i < t.size(); i++) { double d = (double)(t[i] - t0); s[p[i]] += d; n[p[i]]++; mx[p[i]] = std::max(mx[p[i]], d); }
        printf("s_setprio %s: mean end (10 ns ticks after the first wave ended)  group A (prio 0) %.0f  group B (%s) %.0f   max %.0f / %.0f\n",
               use ? "ON " : "off", s[0] / n[0], use ? "prio 3" : "prio 0", s[1] / n[1], mx[0], mx[1]);
    }
    return 0;
}
