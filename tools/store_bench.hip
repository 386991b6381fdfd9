// micro-benchmark: streaming 16-byte stores (plain vs nt, grid-stride vs block-contiguous) at 320 MB and 3 GB
#include <cstdio>
#include <hip/hip_runtime.h>
typedef float vec4 __attribute__((ext_vector_type(4)));
template <bool NT> __global__ __launch_bounds__(256) void k_stride(vec4 *p, size_t n)
{
    size_t stride = (size_t)gridDim.x * blockDim.x;
    vec4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store(v, &p[i]); else p[i] = v;
    }
}
template <bool NT> __global__ __launch_bounds__(256) void k_chunk(vec4 *p, size_t n, size_t per_block)
{
    size_t b0 = (size_t)blockIdx.x * per_block, b1 = b0 + per_block < n ? b0 + per_block : n;
    vec4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = b0 + threadIdx.x; i < b1; i += 256) {
        if (NT) __builtin_nontemporal_store(v, &p[i]); else p[i] = v;
    }
}
template <class F> float timeit(F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f();
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main()
{
    for (size_t bytes : {(size_t)320 << 20, (size_t)3 << 30}) {
        vec4 *p; hipMalloc(&p, bytes); size_t n = bytes / 16;
        for (int blocks : {2048, 8192, 65536}) {
            float t1 = timeit([&] { k_stride<false><<<blocks, 256>>>(p, n); });
            float t2 = timeit([&] { k_stride<true><<<blocks, 256>>>(p, n); });
            size_t pb = (n + blocks - 1) / blocks;
            float t3 = timeit([&] { k_chunk<false><<<blocks, 256>>>(p, n, pb); });
            float t4 = timeit([&] { k_chunk<true><<<blocks, 256>>>(p, n, pb); });
            printf("%5zu MB blocks=%6d  stride plain %.2f nt %.2f | chunk plain %.2f nt %.2f TB/s\n", bytes >> 20, blocks,
                   bytes / t1 / 1e9, bytes / t2 / 1e9, bytes / t3 / 1e9, bytes / t4 / 1e9);
        }
        float t5 = timeit([&] { hipMemsetAsync(p, 0, bytes, 0); });
        printf("%5zu MB hipMemsetAsync %.2f TB/s\n", bytes >> 20, bytes / t5 / 1e9);
        hipFree(p);
    }
    return 0;
}
