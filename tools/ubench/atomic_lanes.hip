// Which lanes of a wave carry the float atomics of one accumulator line -- does it matter?  Every wave adds into pseudo-random 48-byte
// lines (12 floats, as the backward blend's grad_acc), nine or ten floats of a line per wave-instruction, from different lane patterns:
//   0  lanes 4 v  (v < 10): one lane per quad, quads 0-9 (rounds 1-5 of the backward blend)
//   1  lanes 8 r  (r < 7), 56, 60: one lane in every second quad (the eight-row reduction's natural output)
//   2  lanes 0-8 consecutive
//   3  lane 0 only (one float)
//   4  lanes 0, 4, ..., 28 + 25 with the slots of the swap variant (non-monotonic)
//   5  lanes 16 r (r < 4) + ...: four lanes, one per 16-lane quarter
//   6  pattern 0 with plain stores instead of atomics
//   7  lanes 4 v (v < 10) but slots reversed (9 - v)
//   8  TWO lines per instruction: lanes 4 v carry line A, lanes 4 v + 1 line B (20 lanes) -- per LINE half the instructions of pattern 0
//   9  FOUR lines per instruction: lanes 4 v + k carry line k (40 lanes)
// with K VALU-only filler iterations between two atomics (K = 0: back to back).  Output: ns per wave-instruction chip-wide.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int PAT>
__global__ __launch_bounds__(256) void k(float* acc, uint32_t lines, int iters, int filler, float* sink) {
    const unsigned lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int slot = -1;
    if (PAT == 0 || PAT == 6) { if ((lane & 3) == 0 && lane < 40) slot = lane >> 2; }
    else if (PAT == 1) { if (lane < 56) { if ((lane & 7) == 0) slot = lane >> 3; } else if ((lane & 3) == 0) slot = 7 + ((lane - 56) >> 2); }
    else if (PAT == 2) { if (lane < 9) slot = lane; }
    else if (PAT == 3) { if (lane == 0) slot = 0; }
    else if (PAT == 4) { if (lane < 32) { if ((lane & 7) == 0) slot = lane >> 3; else if ((lane & 7) == 4) slot = lane == 28 ? 8 : 4 + (lane >> 3); else if (lane == 25) slot = 7; } }
    else if (PAT == 5) { if ((lane & 15) == 0) slot = lane >> 4; }
    else if (PAT == 7) { if ((lane & 3) == 0 && lane < 40) slot = 9 - (lane >> 2); }
    else if (PAT == 8) { if ((lane & 2) == 0 && lane < 40) slot = lane >> 2; }
    else if (PAT == 9) { if (lane < 40) slot = lane >> 2; }
    float f = (float)lane * 0.001f;
    for (int it = 0; it < iters; it++) {
        uint32_t line = hash(wave * 4096u + it) % lines;
        if (PAT == 8 || PAT == 9) line = hash(wave * 4096u + it + 77777u * (lane & 3)) % lines;
        for (int q = 0; q < filler; q++) f = fmaf(f, 1.0001f, 0.5f);
        if (slot >= 0) {
            if (PAT == 6) acc[(size_t)line * 12 + slot] = f;
            else unsafeAtomicAdd(acc + (size_t)line * 12 + slot, f);
        }
    }
    if (f == 12345.f) sink[0] = f;
}
template <int PAT>
double run(float* acc, uint32_t lines, int iters, int filler, float* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 2048;                        // 8 workgroups of 4 waves per CU: every slot taken, as the blend
    hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, acc, lines, iters, filler, sink);
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, acc, lines, iters, filler, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e6 / 5 / ((double)blocks * 4 * iters);      // ns per wave-instruction, chip-wide
}
int main() {
    const uint32_t lines = 500000;
    float *acc, *sink; hipMalloc(&acc, (size_t)lines * 48 + 4096); hipMalloc(&sink, 4); hipMemset(acc, 0, (size_t)lines * 48);
    printf("ns per atomic wave-instruction chip-wide (8192 waves); requests per ns = 1 / that\n%8s", "filler");
    for (int p = 0; p < 10; p++) printf(" %8s%d", "pat", p);
    printf("\n");
    for (int filler : {0}) {
        printf("%8d", filler);
        const int iters = 120;
        printf(" %9.4f", run<0>(acc, lines, iters, filler, sink)); printf(" %9.4f", run<1>(acc, lines, iters, filler, sink));
        printf(" %9.4f", run<2>(acc, lines, iters, filler, sink)); printf(" %9.4f", run<3>(acc, lines, iters, filler, sink));
        printf(" %9.4f", run<4>(acc, lines, iters, filler, sink)); printf(" %9.4f", run<5>(acc, lines, iters, filler, sink));
        printf(" %9.4f", run<6>(acc, lines, iters, filler, sink)); printf(" %9.4f", run<7>(acc, lines, iters, filler, sink));
        printf(" %9.4f", run<8>(acc, lines, iters, filler, sink)); printf(" %9.4f", run<9>(acc, lines, iters, filler, sink));
        printf("\n");
    }
    return 0;
}
