// micro-benchmark: rocPRIM radix_sort_pairs on voxel keys (decides hash-vs-sort design)
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
template <typename K> void run(size_t n, int bits)
{
    std::vector<K> h(n);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (K)(s & ((bits >= 64) ? ~0ull : ((1ull << bits) - 1))); }
    K *k0, *k1; unsigned *v0, *v1;
    hipMalloc(&k0, n * sizeof(K)); hipMalloc(&k1, n * sizeof(K)); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
    hipMemcpy(k0, h.data(), n * sizeof(K), hipMemcpyHostToDevice);
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, k0, k1, v0, v1, n, 0, bits);
    void *t; hipMalloc(&t, tmp);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; w++) (void)rocprim::radix_sort_pairs(t, tmp, k0, k1, v0, v1, n, 0, bits);
    hipEventRecord(a);
    for (int w = 0; w < 20; w++) (void)rocprim::radix_sort_pairs(t, tmp, k0, k1, v0, v1, n, 0, bits);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("n=%zu keybytes=%zu bits=%d: %.1f us/sort  (%.2f Gkeys/s) tmp=%zu\n", n, sizeof(K), bits, ms * 1000 / 20, n / (ms / 20 * 1e-3) / 1e9, tmp);
    hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(t);
}
int main()
{
    run<unsigned>(1000000, 25); run<unsigned>(1000000, 32); run<unsigned long long>(1000000, 40);
    run<unsigned long long>(1000000, 63); run<unsigned>(8000000, 31); run<unsigned>(64000000, 31);
    run<unsigned>(1000000, 20); run<unsigned>(1000000, 16);
    return 0;
}
