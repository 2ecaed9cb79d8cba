// Reference point only (not part of the product): rocPRIM's radix_sort_keys on the same key shape.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>
#include <cstdint>
#include <cstring>
#include <cmath>
int main(int argc, char **argv) {
    size_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    std::vector<uint64_t> h(n);
    std::mt19937_64 g(1);
    std::normal_distribution<double> nd(40.0, 8.0);
    for (auto &x : h) { double v = std::fabs(nd(g)) + 4.0; memcpy(&x, &v, 8); }
    uint64_t *in, *out; void *tmp = nullptr; size_t tb = 0;
    hipMalloc(&in, n * 8); hipMalloc(&out, n * 8);
    hipMemcpy(in, h.data(), n * 8, hipMemcpyHostToDevice);
    for (int bits : {64, 56}) {
        rocprim::radix_sort_keys(nullptr, tb, in, out, n, 0, bits);
        hipMalloc(&tmp, tb);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int i = 0; i < 3; ++i) rocprim::radix_sort_keys(tmp, tb, in, out, n, 0, bits);
        hipEventRecord(a);
        for (int i = 0; i < 10; ++i) rocprim::radix_sort_keys(tmp, tb, in, out, n, 0, bits);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("rocprim radix_sort_keys n=%zu bits=%d: %.4f ms per sort (tmp %zu bytes)\n", n, bits, ms / 10, tb);
        hipFree(tmp); tmp = nullptr;
    }
    return 0;
}
