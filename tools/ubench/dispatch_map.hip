// Where do the workgroups of a 2048-block, 256-thread, 12 KiB-LDS launch land?  Records XCC id and HW_ID per block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(uint32_t* out, float* sink) {
    __shared__ float lds[3072];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float a = lds[(threadIdx.x * 7) & 255];
    for (int i = 0; i < 20000; i++) a = fmaf(a, 1.0000001f, 0.5f);      // keep the block resident for a while
    if (threadIdx.x == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    }
    if (a == 12345.f) sink[0] = a;
}
int main() {
    const int nb = 2048;
    uint32_t* d; float* s; hipMalloc(&d, nb * 8); hipMalloc(&s, 4);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, s);
    std::vector<uint32_t> h(nb * 2); hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    printf("block: xcc se sh cu simd wave\n");
    std::map<uint32_t, int> per_cu; 
    for (int b = 0; b < nb; b++) {
        uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, simd = (hw >> 4) & 3, wave = hw & 0xf;
        if (b < 48 || (b % 8 == 0 && b < 600)) printf("%4d: %u %u %u %2u %u %u\n", b, xcc, se, sh, cu, simd, wave);
        per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
    }
    printf("distinct CUs used: %zu\n", per_cu.size());
    std::map<int,int> hist; for (auto& kv : per_cu) hist[kv.second]++;
    for (auto& kv : hist) printf("  %d CUs host %d blocks\n", kv.second, kv.first);
    return 0;
}
