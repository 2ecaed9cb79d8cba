// Microbenchmark: cost of LDS atomics under same-address conflicts (development aid, not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;

// mode: 0 = non-returning add, 1 = returning add; pattern p: number of distinct addresses the 64 lanes of a wave
// hit (1 = all the same word, 64 = all different)
template <int RET, int ITER>
__global__ void __launch_bounds__(512) k(u32 *out, int distinct, int shared_across_waves, int moving) {
    __shared__ u32 bins[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) bins[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const u32 a = (u32)(lane % distinct) * 33u % 4096u + (shared_across_waves ? 0u : (u32)wid * 7u);
    u32 acc = 0;
    for (int it = 0; it < ITER; ++it) {
        if (RET) acc += atomicAdd(&bins[(a + it * moving) & 4095u], 1u);
        else atomicAdd(&bins[(a + it * moving) & 4095u], 1u);
    }
    __syncthreads();
    if (acc == 0xFFFFFFFFu || threadIdx.x == 0) out[blockIdx.x] = acc + bins[threadIdx.x];
}

int main() {
    u32 *out;
    hipMalloc(&out, 4096 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int ITER = 256;
    for (int moving : {64, 0})
    for (int ret = 0; ret < 2; ++ret)
        for (int shared = 1; shared < 2; ++shared)
            for (int distinct : {1, 4, 8, 16, 64}) {
                float best = 1e9;
                for (int rep = 0; rep < 5; ++rep) {
                    hipEventRecord(a);
                    if (ret) hipLaunchKernelGGL((k<1, ITER>), dim3(256), dim3(512), 0, 0, out, distinct, shared, moving);
                    else hipLaunchKernelGGL((k<0, ITER>), dim3(256), dim3(512), 0, 0, out, distinct, shared, moving);
                    hipEventRecord(b);
                    hipEventSynchronize(b);
                    float ms;
                    hipEventElapsedTime(&ms, a, b);
                    if (ms < best) best = ms;
                }
                // per CU: 8 waves x ITER wave-instructions
                const double cyc = best * 1e-3 * 2.4e9 / (8.0 * ITER);
                printf("moving=%2d ret=%d shared_across_waves=%d distinct=%2d  %.1f us  ~%.1f cycles per wave-instruction (8 waves/CU)\n",
                       moving, ret, shared, distinct, best * 1e3, cyc);
            }
    return 0;
}
