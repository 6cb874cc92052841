// Checks the DPP inclusive max-scan used by the tile bucketing (binning.hip) against a serial scan.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_max_step(uint32_t v) {
    const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
    return max(v, t);
}
__device__ __forceinline__ uint32_t wave_incl_max_scan(uint32_t v) {
    v = dpp_max_step<0x111, 0xf>(v); v = dpp_max_step<0x112, 0xf>(v);
    v = dpp_max_step<0x114, 0xf>(v); v = dpp_max_step<0x118, 0xf>(v);
    v = dpp_max_step<0x142, 0xa>(v);
    v = dpp_max_step<0x143, 0xc>(v);
    return v;
}
__global__ void k(const uint32_t* in, uint32_t* out) { out[threadIdx.x] = wave_incl_max_scan(in[threadIdx.x]); }
int main() {
    uint32_t *di, *dout; hipMalloc(&di, 256); hipMalloc(&dout, 256);
    int bad = 0;
    uint32_t x = 12345;
    for (int trial = 0; trial < 200; trial++) {
        std::vector<uint32_t> h(64), o(64);
        for (int i = 0; i < 64; i++) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 20) % 5 == 0) ? (uint32_t)(i + 1) : 0u; }
        hipMemcpy(di, h.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
        hipMemcpy(o.data(), dout, 256, hipMemcpyDeviceToHost);
        uint32_t run = 0;
        for (int i = 0; i < 64; i++) { run = run > h[i] ? run : h[i]; if (o[i] != run) { if (bad < 5) printf("trial %d lane %d: got %u want %u\n", trial, i, o[i], run); bad++; } }
    }
    printf("dpp max-scan mismatches: %d\n", bad);
    return bad != 0;
}
