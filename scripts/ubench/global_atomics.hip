// Microbenchmark: what a counting hash table in device memory costs (development aid, not part of the library).
// n_ops keys drawn from `distinct` values are inserted into an open-addressed table of S slots (64-bit key CAS,
// 32-bit count add): the decision between "count the distinct values of the tail" and "move the tail through
// radix levels" for the run-length rank index hangs on this rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef unsigned long long u64;

__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
constexpr u64 EMPTY = ~0ull;

// mode 0: CAS always, then add; 1: load first, CAS only when the slot is not ours; 2: plain coalesced store of the
// key (what compaction would cost); 3: add only (slot known); 4: one 64-bit add carrying key-tag and count
__global__ void __launch_bounds__(256) k_insert(u64 *tk, u32 *tc, u32 smask, u64 n_ops, u64 distinct, int mode, u64 *sink) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_ops; i += (u64)gridDim.x * blockDim.x) {
        const u64 v = mix(mix(i) % distinct + 12345u);  // the key
        if (mode == 2) { sink[i] = v; continue; }
        u32 s = (u32)(mix(v) >> 20) & smask;
        if (mode == 3) { atomicAdd(&tc[s], 1u); continue; }
        if (mode == 4) { atomicAdd(&tk[s], 1ull); continue; }
        for (int probe = 0; probe < 64; ++probe) {
            u64 cur = mode == 1 ? __hip_atomic_load(&tk[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : EMPTY;
            if (mode == 0 || cur == EMPTY) cur = atomicCAS(&tk[s], EMPTY, v), cur = cur == EMPTY ? v : cur;
            if (cur == v) { atomicAdd(&tc[s], 1u); break; }
            s = (s + 1) & smask;
        }
    }
}

int main(int argc, char **argv) {
    const u64 n_ops = argc > 1 ? strtoull(argv[1], 0, 10) : 2000000ull;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    u64 *sink;
    hipMalloc(&sink, n_ops * 8);
    for (int lg : {20, 22, 24}) {
        const u32 S = 1u << lg;
        u64 *tk; u32 *tc;
        hipMalloc(&tk, (size_t)S * 8);
        hipMalloc(&tc, (size_t)S * 4);
        for (u64 distinct : {240ull, 20000ull, 171000ull, n_ops}) {
            if (distinct * 2 > S) continue;
            for (int mode = 0; mode < 5; ++mode) {
                for (int grid : {512, 2048}) {
                    float best = 1e9;
                    for (int rep = 0; rep < 4; ++rep) {
                        hipMemset(tk, 0xFF, (size_t)S * 8);
                        hipMemset(tc, 0, (size_t)S * 4);
                        hipDeviceSynchronize();
                        hipEventRecord(a);
                        hipLaunchKernelGGL(k_insert, dim3(grid), dim3(256), 0, 0, tk, tc, S - 1, n_ops, distinct, mode, sink);
                        hipEventRecord(b);
                        hipEventSynchronize(b);
                        float ms;
                        hipEventElapsedTime(&ms, a, b);
                        if (ms < best) best = ms;
                    }
                    printf("slots=2^%d distinct=%8llu mode=%d grid=%4d  %8.1f us  %.2f Gops/s\n", lg,
                           (unsigned long long)distinct, mode, grid, best * 1e3, n_ops / (best * 1e-3) * 1e-9);
                }
            }
        }
        hipFree(tk);
        hipFree(tc);
    }
    return 0;
}
