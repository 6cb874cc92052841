// Does ds_add_rtn_u32 resolve same-address conflicts of ONE wave in increasing lane order on gfx950?
// For each trial every lane picks a digit (random, heavy conflicts), does atomicAdd(&cnt[digit], 1) on a per-wave LDS
// table and we check that among lanes with equal digit the returned values increase with lane id.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint32_t* digits, int trials, int nbins, uint32_t* violations, uint32_t* checked) {
    __shared__ uint32_t cnt[16][256];
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t bad = 0, n = 0;
    for (int t = 0; t < trials; t++) {
        for (int i = lane; i < 256; i += 64) cnt[w][i] = 0;
        __builtin_amdgcn_wave_barrier();
        const uint32_t d = digits[((size_t)(blockIdx.x * 16 + w) * trials + t) * 64 + lane] % nbins;
        const uint32_t r = atomicAdd(&cnt[w][d], 1u);
        // expected stable rank = number of lower lanes with the same digit
        uint32_t expect = 0;
        for (int l = 0; l < 64; l++) { const uint32_t dl = __shfl(d, l, 64); if (l < (int)lane && dl == d) expect++; }
        if (r != expect) bad++;
        n++;
        __builtin_amdgcn_wave_barrier();
    }
    atomicAdd(violations, bad); atomicAdd(checked, n);
}
int main() {
    const int blocks = 512, trials = 64;
    size_t n = (size_t)blocks * 16 * trials * 64;
    uint32_t* h = (uint32_t*)malloc(n * 4);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s >> 20); }
    uint32_t *d, *v, *c; hipMalloc(&d, n * 4); hipMalloc(&v, 4); hipMalloc(&c, 4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    for (int nbins : {1, 2, 3, 7, 16, 64, 256}) {
        hipMemset(v, 0, 4); hipMemset(c, 0, 4);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, d, trials, nbins, v, c);
        uint32_t hv, hc; hipMemcpy(&hv, v, 4, hipMemcpyDeviceToHost); hipMemcpy(&hc, c, 4, hipMemcpyDeviceToHost);
        printf("bins %3d: %u lane-ops checked, %u out of lane order\n", nbins, hc, hv);
    }
    return 0;
}
