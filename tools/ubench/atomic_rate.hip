// How fast are scattered device-scope global atomics on gfx950?  Decides whether the tile bucketing may count with
// fire-and-forget atomics (per-tile counters, 2-D difference array of the rectangles) instead of a [tile][workgroup] table.
//   mode 0  fire-and-forget atomicAdd(u32) to a pseudo-random word of a table of M words (return value unused)
//   mode 1  the same, return value used (a slot reservation)
//   mode 2  plain scattered 4-byte stores (the floor)
// N threads x K operations each; addresses: hash(thread, k) % M.  Shards: the table is replicated S times, shard = blockIdx % S.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ void k(uint32_t* table, uint32_t M, int K, int S, uint32_t* sink) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t* tab = table + (size_t)(blockIdx.x % S) * M;
    uint32_t acc = 0;
    for (int k = 0; k < K; k++) {
        const uint32_t a = hash(t * 16u + k) % M;
        if (MODE == 0) atomicAdd(&tab[a], 1u);
        else if (MODE == 1) acc += atomicAdd(&tab[a], 1u);
        else tab[a] = t;
    }
    if (MODE == 1 && acc == 0xdeadbeefu) sink[0] = acc;
}
template <int MODE>
float run(uint32_t* table, uint32_t M, int N, int K, int S, uint32_t* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k<MODE>, dim3(N / 256), dim3(256), 0, 0, table, M, K, S, sink);
    hipEventRecord(a);
    const int reps = 10;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k<MODE>, dim3(N / 256), dim3(256), 0, 0, table, M, K, S, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}
int main() {
    uint32_t *table, *sink; hipMalloc(&table, (size_t)8 * (1 << 20) * 4); hipMalloc(&sink, 4);
    hipMemset(table, 0, (size_t)8 * (1 << 20) * 4);
    const int N = 512 * 1024;
    printf("%8s %3s %3s | %10s %10s %10s   (us per launch; N = %d threads)\n", "M", "K", "S", "add", "add_rtn", "store", N);
    for (uint32_t M : {2048u, 8192u, 32768u, 1u << 20})
        for (int K : {1, 4})
            for (int S : {1, 8}) {
                const float t0 = run<0>(table, M, N, K, S, sink), t1 = run<1>(table, M, N, K, S, sink), t2 = run<2>(table, M, N, K, S, sink);
                printf("%8u %3d %3d | %10.1f %10.1f %10.1f   -> %.0f / %.0f M atomics per ms\n", M, K, S, t0, t1, t2,
                       (double)N * K / t0 / 1000.0, (double)N * K / t1 / 1000.0);
            }
    return 0;
}
