// Microbenchmark: what one Philox4x32-10 block costs a wave on gfx950, by the form of its 32x32->64 multiplies
// (development aid).  k_sample's loop produces one block per iteration; the verdict on round 2 asked for the price
// of v_mad_u64_u32 against v_mul_hi_u32 + v_mul_lo_u32 in cycles per block.
//   form 0: (u64)M * c as the compiler lowers it          form 1: __umulhi(M, c) and M * c
//   form 2: v_mad_u64_u32 by inline asm                   form 3: no multiply at all (xor/add only: the rest of a block)
// Every SIMD runs `waves` waves of ITER dependent blocks; cycles per block and wave = time * clock / (ITER * waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef unsigned long long u64;
constexpr u32 M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;

template <int FORM>
__device__ __forceinline__ void mul(u32 m, u32 c, u32 &hi, u32 &lo) {
    if (FORM == 0) {
        const u64 p = (u64)m * c;
        hi = (u32)(p >> 32);
        lo = (u32)p;
    } else if (FORM == 1) {
        hi = __umulhi(m, c);
        lo = m * c;
    } else if (FORM == 2) {
        u64 p;
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(m), "v"(c) : "vcc");
        hi = (u32)(p >> 32);
        lo = (u32)p;
    } else {
        hi = m ^ c;
        lo = m + c;
    }
}

template <int FORM, int ITER>
__global__ void __launch_bounds__(256) k(u32 *out, u32 k0, u32 k1) {
    u32 c0 = threadIdx.x, c1 = blockIdx.x, c2 = 1, c3 = 2;
    for (int it = 0; it < ITER; ++it) {
        u32 a0 = k0, a1 = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            u32 hi0, lo0, hi1, lo1;
            mul<FORM>(M0, c0, hi0, lo0);
            mul<FORM>(M1, c2, hi1, lo1);
            const u32 n0 = hi1 ^ c1 ^ a0, n2 = hi0 ^ c3 ^ a1;
            c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
            a0 += W0; a1 += W1;
        }
    }
    if ((c0 ^ c1 ^ c2 ^ c3) == 0x12345u) out[0] = c0;
}

template <int FORM>
void run(const char *name, u32 *out, int wg_per_cu) {
    constexpr int ITER = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<FORM, ITER>), dim3(256 * wg_per_cu), dim3(256), 0, 0, out, 123u, 456u);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    // one workgroup = 4 waves = one wave per SIMD of its CU: wg_per_cu waves per SIMD
    const double cyc = best * 1e-3 * 2.4e9 / ((double)ITER * wg_per_cu);
    printf("%-28s waves/SIMD=%d  %8.1f us  %.0f cycles per block and wave at 2.4 GHz\n", name, wg_per_cu, best * 1e3, cyc);
}

int main() {
    u32 *out;
    hipMalloc(&out, 64);
    for (int w : {1, 2, 4, 8}) {
        run<0>("(u64)M*c (compiler)", out, w);
        run<1>("__umulhi + mul", out, w);
        run<2>("v_mad_u64_u32 (asm)", out, w);
        run<3>("no multiplies", out, w);
    }
    return 0;
}
