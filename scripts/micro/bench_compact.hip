// Microbenchmark of the LDS pool compaction primitives (qk_device.h): one wave per workgroup, R repetitions.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I quake_amd/csrc scripts/micro/bench_compact.hip -o /tmp/bench_compact && /tmp/bench_compact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <climits>
#include "qk_device.h"

template <int MAXCH, int SEL>
__global__ __launch_bounds__(64) void k_bench(int n, int k, int reps, uint32_t *sink) {
    __shared__ int64_t pid[64 * MAXCH];
    __shared__ uint32_t pord[64 * MAXCH];
    const int lane = threadIdx.x;
    uint32_t acc = 0;
    for (int r = 0; r < reps; r++) {
        for (int e = lane; e < n; e += 64) {
            uint32_t h = (uint32_t)(e * 2654435761u) ^ (uint32_t)(r * 40503u) ^ (blockIdx.x * 97u);
            h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
            pord[e] = h | 1u;
            pid[e] = e + 1000 * r;
        }
        __builtin_amdgcn_wave_barrier();
        int nn;
        if (SEL) {
            uint32_t kth;
            nn = select_pool<MAXCH>(pord, pid, n, k, lane, kth);
            acc += kth;
        } else {
            nn = compact_pool<MAXCH>(pord, pid, n, k, lane);
            acc += pord[k - 1];
        }
        acc += nn;
    }
    if (lane == 0) sink[blockIdx.x] = acc;
}

template <int MAXCH, int SEL>
void run(int n, int k) {
    uint32_t *sink;
    hipMalloc(&sink, 4096 * 4);
    const int reps = 200, grid = 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_bench<MAXCH, SEL>), dim3(grid), dim3(64), 0, 0, n, k, 2, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_bench<MAXCH, SEL>), dim3(grid), dim3(64), 0, 0, n, k, reps, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s<%d> n=%d k=%d : %.2f us per call (1024 waves = 4 per CU)\n", SEL ? "select_pool " : "compact_pool", MAXCH, n, k, ms * 1e3 / reps);
    hipFree(sink);
}

int main() {
    run<1, 0>(38, 10); run<1, 0>(64, 10);
    run<2, 0>(128, 100); run<2, 1>(128, 100);
    run<4, 0>(164, 100); run<4, 1>(164, 100); run<4, 0>(192, 100); run<4, 1>(192, 100);
    run<8, 0>(384, 100); run<8, 1>(384, 100); run<8, 0>(512, 448); run<8, 1>(512, 448);
    return 0;
}
