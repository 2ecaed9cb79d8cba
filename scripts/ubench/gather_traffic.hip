// Microbenchmark: what FETCH_SIZE reports for random 16-byte gathers against a known byte count (development aid).
// MI355X_MICROARCH.md calibrates the counter for wide coalesced streaming reads only (it reports half the bytes);
// k_sample's reads are random 16-byte records.  Run under `rocprofv3 --pmc FETCH_SIZE`:
//   k_stream   every lane reads 16 consecutive bytes, the wave 1 KiB: TABLE bytes in all
//   k_gather2  as k_gather, plus the record 64 bytes away in the same 128-byte line
//   k_gather   every lane reads one random aligned 16-byte record of the table, n_gathers of them: the bytes a
//              lane asks for are n_gathers * 16; the 64-byte sectors / 128-byte lines they touch are 4 x / 8 x that
//              when every gather misses, less when lines are still in an L2 (the table is 10 x an XCD's L2)
// The table is read once by k_stream first so that the Infinity Cache holds what it can (as the sampler's table is).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef unsigned long long u64;

__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

__global__ void __launch_bounds__(256) k_stream(const uint4 *t, u64 n16, u32 *out) {
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) {
        const uint4 v = t[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void __launch_bounds__(256) k_gather(const uint4 *t, u64 n16, u64 n_gathers, u64 salt, u32 *out) {
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_gathers; i += (u64)gridDim.x * blockDim.x) {
        const uint4 v = t[mix(i + salt) % n16];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// the same gathers, each followed by the record 64 bytes away in the same 128-byte line (index ^ 4): twice the bytes
// asked for; if a miss brings the whole 128-byte line and the counter tallies it at 64 B, FETCH_SIZE stays where
// k_gather's was; if HBM is read in 64-byte sectors, it doubles
__global__ void __launch_bounds__(256) k_gather2(const uint4 *t, u64 n16, u64 n_gathers, u64 salt, u32 *out) {
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_gathers; i += (u64)gridDim.x * blockDim.x) {
        const u64 j = mix(i + salt) % n16;
        const uint4 v = t[j], w = t[j ^ 4];
        acc ^= v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y ^ w.z ^ w.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char **argv) {
    const u64 table_bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 40ull) << 20;
    const u64 n_gathers = argc > 2 ? strtoull(argv[2], 0, 10) : 100000000ull;
    const u64 n16 = table_bytes / 16;
    uint4 *t; u32 *out;
    hipMalloc(&t, table_bytes);
    hipMalloc(&out, 64);
    hipMemset(t, 1, table_bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (const uint4 *)t, n16, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("k_stream  %llu bytes  %.1f us\n", (unsigned long long)table_bytes, ms * 1e3);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_gather, dim3(4096), dim3(256), 0, 0, (const uint4 *)t, n16, n_gathers, (u64)rep * 7919u, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("k_gather  %llu gathers = %llu bytes asked for  %.1f us\n", (unsigned long long)n_gathers,
               (unsigned long long)(n_gathers * 16), ms * 1e3);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_gather2, dim3(4096), dim3(256), 0, 0, (const uint4 *)t, n16, n_gathers, (u64)rep * 7919u + 13u, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("k_gather2 %llu pairs in one 128-byte line = %llu bytes asked for  %.1f us\n", (unsigned long long)n_gathers,
               (unsigned long long)(n_gathers * 32), ms * 1e3);
    }
    return 0;
}
