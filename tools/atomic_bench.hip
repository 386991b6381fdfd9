// micro-benchmark: scattered 64-bit atomicAdd throughput by memory scope, with and without confining every
// workgroup's addresses to the table region of the XCD it runs on (HW_REG_XCC_ID)
#include <cstdio>
#include <cstdint>
#include <hip/hip_runtime.h>
typedef unsigned long long u64;
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xf; }
__device__ __forceinline__ u64 mix64(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

__global__ void k_xcc(uint32_t *out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

// SCOPE: 0 agent, 1 workgroup.  REGION: addresses confined to [xcc * slots/8, (xcc+1) * slots/8)
template <int SCOPE, bool REGION, bool RET>
__global__ __launch_bounds__(256) void k_atomic(u64 *tab, u64 slots, int per_thread, u64 *sink)
{
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 region = slots / 8, base = REGION ? xcc_id() * region : 0, span = REGION ? region : slots;
    u64 acc = 0;
    for (int k = 0; k < per_thread; k++) {
        const u64 h = base + mix64(gid * 131 + k) % span;
        if (SCOPE == 0) {
            if (RET) acc += __hip_atomic_fetch_add(&tab[h], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(&tab[h], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (RET) acc += __hip_atomic_fetch_add(&tab[h], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(&tab[h], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (RET && acc == 0x1234567) *sink = acc;
}
// OP: 0 add64, 1 cas64 (claim an empty slot, as the voxel hash does), 2 add32, 3 cas32, 4 or64
template <int OP>
__global__ __launch_bounds__(256) void k_ops(u64 *tab, u64 slots, int per_thread, u64 *sink)
{
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 acc = 0;
    for (int k = 0; k < per_thread; k++) {
        const u64 h = mix64(gid * 131 + k) % slots;
        if (OP == 0) acc += atomicAdd(&tab[h], 1ull);
        if (OP == 1) acc += atomicCAS(&tab[h], 0ull, gid + 1);
        if (OP == 2) acc += atomicAdd(reinterpret_cast<unsigned int *>(tab) + h, 1u);
        if (OP == 3) acc += atomicCAS(reinterpret_cast<unsigned int *>(tab) + h, 0u, (unsigned int)gid + 1);
        if (OP == 4) acc += atomicOr(&tab[h], 1ull << (gid & 63));
    }
    if (acc == 0x1234567) *sink = acc;
}
// the voxel hash's probe: one coherent 8-byte load, then one CAS on the same slot (2 requests per thread)
__global__ __launch_bounds__(256) void k_probe(u64 *tab, u64 slots, u64 *sink)
{
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 h = mix64(gid * 131) % slots;
    const u64 cur = __hip_atomic_load(&tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 old = atomicCAS(&tab[h], cur, cur + 1);
    if (old == 0x1234567) *sink = old;
}
// scattered plain accesses: OP 0 = 4-byte store, 1 = 16-byte store, 2 = 16-byte load, 3 = 8-byte load
template <int OP>
__global__ __launch_bounds__(256) void k_scat(u64 *tab, u64 slots, u64 *sink)
{
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 h = mix64(gid * 131) % slots;
    if (OP == 0) reinterpret_cast<unsigned int *>(tab)[h * 2] = (unsigned int)gid;
    if (OP == 1) reinterpret_cast<uint4 *>(tab)[h / 2] = make_uint4((unsigned)gid, 1, 2, 3);
    if (OP == 2) { uint4 v = reinterpret_cast<const uint4 *>(tab)[h / 2]; if (v.x == 0x1234567u) *sink = v.y; }
    if (OP == 3) { u64 v = tab[h]; if (v == 0x1234567ull) *sink = v; }
}
__global__ void k_sum(const u64 *tab, u64 slots, u64 *out)
{
    u64 s = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (u64)gridDim.x * blockDim.x) s += tab[i];
    atomicAdd(out, s);
}
template <class F> float timeit(F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipEventRecord(a);
    for (int i = 0; i < 5; i++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main()
{
    uint32_t *x; hipMalloc(&x, 4096 * 4); k_xcc<<<4096, 64>>>(x);
    uint32_t hx[4096]; hipMemcpy(hx, x, sizeof hx, hipMemcpyDeviceToHost);
    int mism = 0; int hist[16] = {0};
    for (int i = 0; i < 4096; i++) { hist[hx[i]]++; if (hx[i] != (uint32_t)(i % 8)) mism++; }
    printf("xcc ids of the first 16 workgroups:"); for (int i = 0; i < 16; i++) printf(" %u", hx[i]);
    printf("\nblocks with xcc != blockIdx %% 8: %d of 4096; histogram:", mism); for (int i = 0; i < 8; i++) printf(" %d", hist[i]); printf("\n");
    u64 *sink; hipMalloc(&sink, 8);
    for (u64 mb : {16ull, 32ull, 256ull}) {
        const u64 slots = mb * 1024 * 1024 / 8;
        u64 *tab; hipMalloc(&tab, slots * 8);
        const int blocks = 4096, per = 4;     // ~4.2 M atomics per launch
        const double total = (double)blocks * 256 * per;
        auto run = [&](const char *name, auto kern) {
            hipMemset(tab, 0, slots * 8); hipMemset(sink, 0, 8);
            float ms = timeit([&] { kern<<<blocks, 256>>>(tab, slots, per, sink); });
            u64 *d; hipMalloc(&d, 8); hipMemset(d, 0, 8); k_sum<<<1024, 256>>>(tab, slots, d);
            u64 h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); hipFree(d);
            printf("%4llu MB %-34s %7.2f G atomics/s   (sum check %s)\n", mb, name, total / ms / 1e6,
                   h == (u64)(total * 6) ? "ok" : "MISMATCH");
        };
        run("agent, whole table, no return", k_atomic<0, false, false>);
        run("agent, whole table, returning", k_atomic<0, false, true>);
        run("agent, xcc region, returning", k_atomic<0, true, true>);
        run("workgroup, xcc region, no return", k_atomic<1, true, false>);
        run("workgroup, xcc region, returning", k_atomic<1, true, true>);
        auto run2 = [&](const char *name, auto kern) {
            hipMemset(tab, 0, slots * 8);
            float ms = timeit([&] { kern<<<blocks, 256>>>(tab, slots, per, sink); });
            printf("%4llu MB %-34s %7.2f G atomics/s\n", mb, name, total / ms / 1e6);
        };
        run2("returning add 64", k_ops<0>);
        run2("returning cas 64", k_ops<1>);
        run2("returning add 32", k_ops<2>);
        run2("returning cas 32", k_ops<3>);
        run2("returning or 64", k_ops<4>);
        for (int per1 : {1, 2}) {
            const int blocks1 = 4096 * 4 / per1;
            hipMemset(tab, 0, slots * 8);
            float ms = timeit([&] { k_ops<1><<<blocks1, 256>>>(tab, slots, per1, sink); });
            printf("%4llu MB cas 64, %d per thread, %d blocks      %7.2f G atomics/s\n", mb, per1, blocks1, (double)blocks1 * 256 * per1 / ms / 1e6);
        }
        {
            const int blocksS = 16384;
            float m0 = timeit([&] { k_scat<0><<<blocksS, 256>>>(tab, slots, sink); });
            float m1 = timeit([&] { k_scat<1><<<blocksS, 256>>>(tab, slots, sink); });
            float m2 = timeit([&] { k_scat<2><<<blocksS, 256>>>(tab, slots, sink); });
            float m3 = timeit([&] { k_scat<3><<<blocksS, 256>>>(tab, slots, sink); });
            const double tot = (double)blocksS * 256 / 1e6;
            printf("%4llu MB scattered: 4-B store %.1f, 16-B store %.1f, 16-B load %.1f, 8-B load %.1f G/s\n", mb, tot / m0, tot / m1,
                   tot / m2, tot / m3);
        }
        {
            const int blocks1 = 16384;
            hipMemset(tab, 0, slots * 8);
            float ms = timeit([&] { k_probe<<<blocks1, 256>>>(tab, slots, sink); });
            printf("%4llu MB coherent load + cas per thread        %7.2f G requests/s\n", mb, 2.0 * blocks1 * 256 / ms / 1e6);
        }
        hipFree(tab);
    }
    return 0;
}
