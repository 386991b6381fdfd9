// micro-benchmark for the HBM-bound fill of voxels[V,P,4] (k_fill_c4) at config 5's size (3 GB, beyond the Infinity Cache):
// what do plain / nontemporal stores reach on this box for different launch shapes, and what do the gathers cost?
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/fill_bench tools/fill_bench.hip
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime.h>
typedef float vec4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 plain, 1 nt
__global__ __launch_bounds__(256) void k_store_stride(vec4 *p, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const vec4 v = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (MODE) __builtin_nontemporal_store(v, &p[i]); else p[i] = v;
    }
}
// each block owns one contiguous chunk
template <int MODE>
__global__ __launch_bounds__(256) void k_store_chunk(vec4 *p, size_t n, size_t per_block)
{
    const size_t b0 = (size_t)blockIdx.x * per_block, b1 = b0 + per_block < n ? b0 + per_block : n;
    const vec4 v = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = b0 + threadIdx.x; i < b1; i += 256) {
        if (MODE) __builtin_nontemporal_store(v, &p[i]); else p[i] = v;
    }
}
// 4 stores per lane per iteration (independent), stride layout
template <int MODE>
__global__ __launch_bounds__(256) void k_store_x4(vec4 *p, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    const vec4 v = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x * 4 + threadIdx.x; i < n; i += stride) {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (i + k * 256 < n) { if (MODE) __builtin_nontemporal_store(v, &p[i + k * 256]); else p[i + k * 256] = v; }
    }
}
// the fill itself: row r -> voxel r >> 5, slot r & 31; vinfo {.., .., base, count}
template <int MODE>
__global__ __launch_bounds__(256) void k_fill(const vec4 *__restrict__ staged, const uint4 *__restrict__ vinfo, size_t rows, vec4 *out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
        const size_t v = r >> 5;
        const uint32_t k = (uint32_t)(r & 31);
        const uint4 vi = vinfo[v];
        vec4 val = {0.f, 0.f, 0.f, 0.f};
        if (k < vi.w) val = staged[vi.z + k];
        if (MODE) __builtin_nontemporal_store(val, &out[r]); else out[r] = val;
    }
}
// fill, one voxel per half-wave but the record is loaded once per 32 lanes through a readlane-style broadcast:
// lane l of a wavefront loads vinfo[v0 + (l & 1)]... (here: lanes 0 and 32 load, shfl to the rest)
template <int MODE>
__global__ __launch_bounds__(256) void k_fill_bcast(const vec4 *__restrict__ staged, const uint4 *__restrict__ vinfo, size_t rows, vec4 *out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
        const size_t v = r >> 5;
        const uint32_t k = (uint32_t)(r & 31);
        uint32_t base = 0, cnt = 0;
        if ((lane & 31) == 0) { const uint4 vi = vinfo[v]; base = vi.z; cnt = vi.w; }
        base = __shfl(base, lane & 32, 64);
        cnt = __shfl(cnt, lane & 32, 64);
        vec4 val = {0.f, 0.f, 0.f, 0.f};
        if (k < cnt) val = staged[base + k];
        if (MODE) __builtin_nontemporal_store(val, &out[r]); else out[r] = val;
    }
}
// group-per-wavefront fill (the library's k_fill_c4): VAR 0 = ds_bpermute of {count, base, first row}; 1 = records staged in
// LDS (one b128 + one b64 read per step); 2 = v_readlane (P == 32: the voxel is uniform per half-wavefront)
template <int G, int VAR>
__global__ __launch_bounds__(256) void k_fill_group(const vec4 *__restrict__ staged, const uint4 *__restrict__ vinfo, size_t V, vec4 *voxels)
{
    constexpr uint32_t P = 32;
    __shared__ vec4 lfirst[4][64];
    __shared__ uint2 lrec[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t group = (size_t)blockIdx.x * 4 + w;
    const size_t v0 = group * G;
    if (v0 >= V) return;
    const uint32_t nv = V - v0 < G ? (uint32_t)(V - v0) : (uint32_t)G;
    uint32_t base = 0, cnt = 0;
    vec4 first = {0.f, 0.f, 0.f, 0.f};
    if ((uint32_t)lane < nv) {
        const uint4 vi = vinfo[v0 + lane];
        base = vi.z; cnt = vi.w;
        if (cnt > 0) first = staged[base];
    }
    if (VAR == 1) { lfirst[w][lane] = first; lrec[w][lane] = make_uint2(base, cnt); __builtin_amdgcn_wave_barrier(); }
    const uint32_t nrows = nv * P;
    vec4 *out = voxels + v0 * P;
    const vec4 zero = {0.f, 0.f, 0.f, 0.f};
    for (uint32_t q0 = 0; q0 < nrows; q0 += 4 * 64) {
        vec4 val[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t q = q0 + u * 64 + lane;
            const uint32_t qq = q < nrows ? q : 0u;
            const uint32_t j = qq >> 5, slot = qq & 31;
            uint32_t c, b; vec4 f;
            if (VAR == 0) {
                c = (uint32_t)__shfl((int)cnt, (int)j, 64); b = (uint32_t)__shfl((int)base, (int)j, 64);
                f.x = __shfl(first.x, (int)j, 64); f.y = __shfl(first.y, (int)j, 64); f.z = __shfl(first.z, (int)j, 64); f.w = __shfl(first.w, (int)j, 64);
            } else if (VAR == 1) {
                const uint2 r = lrec[w][j]; b = r.x; c = r.y; f = lfirst[w][j];
            } else {
                const uint32_t ja = (q0 + u * 64) >> 5, jb = ja + 1 < 64 ? ja + 1 : 63;      // uniform
                const bool hi = lane >= 32;
                const uint32_t c0 = __builtin_amdgcn_readlane(cnt, ja), c1 = __builtin_amdgcn_readlane(cnt, jb);
                const uint32_t b0 = __builtin_amdgcn_readlane(base, ja), b1 = __builtin_amdgcn_readlane(base, jb);
                c = hi ? c1 : c0; b = hi ? b1 : b0;
                f.x = hi ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, first.x), jb)) : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, first.x), ja));
                f.y = hi ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, first.y), jb)) : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, first.y), ja));
                f.z = hi ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, first.z), jb)) : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, first.z), ja));
                f.w = hi ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, first.w), jb)) : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, first.w), ja));
                if (q >= nrows) c = 0;
            }
            val[u] = (slot == 0 && c > 0) ? f : zero;
            if (slot > 0 && slot < c) val[u] = staged[b + slot];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t q = q0 + u * 64 + lane;
            if (q < nrows) __builtin_nontemporal_store(val[u], &out[q]);
        }
    }
}
// row-per-lane STORE order (grid-stride: the chip writes one compact moving window) with the reads of the group form: a
// wavefront prefetches the records + first rows of its next 32 steps (64 voxels: lane l -> step l >> 1, half l & 1) in two
// instructions, then runs the 32 steps from registers (ds_bpermute)
__global__ __launch_bounds__(256) void k_fill_pref(const vec4 *__restrict__ staged, const uint4 *__restrict__ vinfo, size_t V, vec4 *out)
{
    const int lane = threadIdx.x & 63;
    const size_t W = (size_t)gridDim.x * 4, gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nsteps = (V + 1) / 2;
    const vec4 zero = {0.f, 0.f, 0.f, 0.f};
    const uint32_t slot = lane & 31;
    for (size_t s0 = gw; s0 < nsteps; s0 += 32 * W) {
        const size_t vox = 2 * (s0 + (size_t)(lane >> 1) * W) + (lane & 1);
        uint32_t base = 0, cnt = 0;
        vec4 first = zero;
        if (vox < V) {
            const uint4 vi = vinfo[vox];
            base = vi.z; cnt = vi.w;
            if (cnt > 0) first = staged[base];
        }
        for (int k0 = 0; k0 < 32; k0 += 4) {
            if (s0 + (size_t)k0 * W >= nsteps) break;
            vec4 val[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int src = 2 * (k0 + u) + (lane >> 5);
                const uint32_t c = (uint32_t)__shfl((int)cnt, src, 64), b = (uint32_t)__shfl((int)base, src, 64);
                vec4 f;
                f.x = __shfl(first.x, src, 64); f.y = __shfl(first.y, src, 64); f.z = __shfl(first.z, src, 64); f.w = __shfl(first.w, src, 64);
                val[u] = (slot == 0 && c > 0) ? f : zero;
                if (slot > 0 && slot < c) val[u] = staged[b + slot];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const size_t s = s0 + (size_t)(k0 + u) * W;
                const size_t r = 64 * s + lane;
                if (s < nsteps && r < V * 32) __builtin_nontemporal_store(val[u], &out[r]);
            }
        }
    }
}

template <int G, int VAR> float run_group(const vec4 *staged, const uint4 *vinfo, size_t V, vec4 *out)
{
    const unsigned blocks = (unsigned)(((V + G - 1) / G + 3) / 4);
    return timeit([&] { k_fill_group<G, VAR><<<blocks, 256>>>(staged, vinfo, V, out); });
}

// fill in 4-row units: lane handles rows 4q..4q+3 of a voxel?  (64 B per lane contiguous: 4 stores of 16 B at stride 16 B
// -> a wavefront's store instruction covers 64 x 16 B at stride 64 B: NOT contiguous; kept for comparison)
vec4 *g_flush = nullptr;                 // 1 GB scratch: plain stores through it evict the Infinity Cache between runs
bool g_do_flush = false;
template <class F> float timeit(F f, int it = 5)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f();
    float tot = 0;
    for (int i = 0; i < it; i++) {
        if (g_do_flush) k_store_stride<0><<<8192, 256>>>(g_flush, ((size_t)1 << 30) / 16);
        (void)hipEventRecord(a);
        f();
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); tot += ms;
    }
    return tot / it;
}
// row-per-lane, grid-stride, FOUR rows per lane in flight: 4 record loads, then 4 row gathers, then 4 nt stores
template <int MODE>
__global__ __launch_bounds__(256) void k_fill_x4(const vec4 *__restrict__ staged, const uint4 *__restrict__ vinfo, size_t rows, vec4 *out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const vec4 zero = {0.f, 0.f, 0.f, 0.f};
    for (size_t r0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r0 < rows; r0 += 4 * stride) {
        uint4 vi[4];
        vec4 val[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const size_t r = r0 + u * stride; vi[u] = vinfo[(r < rows ? r : r0) >> 5]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t r = r0 + u * stride;
            const uint32_t k = (uint32_t)(r & 31);
            val[u] = zero;
            if (r < rows && k < vi[u].w) val[u] = staged[vi[u].z + k];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t r = r0 + u * stride;
            if (r < rows) { if (MODE) __builtin_nontemporal_store(val[u], &out[r]); else out[r] = val[u]; }
        }
    }
}
int main()
{
    const size_t V = 5866485, P = 32, rows = V * P, N = 8000000;
    vec4 *out; hipMalloc(&out, rows * 16);
    vec4 *staged; hipMalloc(&staged, (N + 64) * 16); hipMemset(staged, 0, (N + 64) * 16);
    std::vector<uint4> vi(V);
    // config 5's count distribution, roughly: 80 % of the voxels hold 1 point, the rest 2..8 (sum ~ N)
    uint32_t base = 0; srand(1);
    for (size_t v = 0; v < V; v++) {
        uint32_t c = (rand() % 100 < 80) ? 1 : 2 + rand() % 4;
        if (base + c > N) c = 1;
        if (base + c > N) { base = 0; }
        vi[v] = make_uint4(0, 0, base, c); base += c;
    }
    uint4 *vinfo; hipMalloc(&vinfo, V * 16); hipMemcpy(vinfo, vi.data(), V * 16, hipMemcpyHostToDevice);
    // segments in RANDOM order across the staged buffer (what first-seen order does to bucket-ordered segments)
    std::vector<uint4> vr = vi;
    for (size_t v = V - 1; v > 0; v--) { size_t j = ((size_t)rand() * 32768 + rand()) % (v + 1); uint32_t z = vr[v].z, w = vr[v].w; vr[v].z = vr[j].z; vr[v].w = vr[j].w; vr[j].z = z; vr[j].w = w; }
    uint4 *vinfo_r; hipMalloc(&vinfo_r, V * 16); hipMemcpy(vinfo_r, vr.data(), V * 16, hipMemcpyHostToDevice);
    (void)hipMalloc(&g_flush, (size_t)1 << 30);
    const double GB = rows * 16 / 1e9;
    printf("output %.2f GB\n", GB);
    for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
        float a = timeit([&] { k_store_stride<0><<<blocks, 256>>>(out, rows); });
        float b = timeit([&] { k_store_stride<1><<<blocks, 256>>>(out, rows); });
        size_t pb = (rows + blocks - 1) / blocks;
        float c = timeit([&] { k_store_chunk<0><<<blocks, 256>>>(out, rows, pb); });
        float d = timeit([&] { k_store_chunk<1><<<blocks, 256>>>(out, rows, pb); });
        float e = timeit([&] { k_store_x4<0><<<blocks, 256>>>(out, rows); });
        float f = timeit([&] { k_store_x4<1><<<blocks, 256>>>(out, rows); });
        printf("blocks %6d  stride plain %.2f nt %.2f | chunk plain %.2f nt %.2f | x4 plain %.2f nt %.2f TB/s\n", blocks, GB / a, GB / b,
               GB / c, GB / d, GB / e, GB / f);
    }
    float m = timeit([&] { hipMemsetAsync(out, 0, rows * 16, 0); });
    printf("hipMemsetAsync %.2f TB/s\n", GB / m);
    for (int blocks : {2048, 8192, 16384}) {
        float a = timeit([&] { k_fill<0><<<blocks, 256>>>(staged, vinfo, rows, out); });
        float b = timeit([&] { k_fill<1><<<blocks, 256>>>(staged, vinfo, rows, out); });
        float c = timeit([&] { k_fill<1><<<blocks, 256>>>(staged, vinfo_r, rows, out); });
        float d = timeit([&] { k_fill_bcast<1><<<blocks, 256>>>(staged, vinfo_r, rows, out); });
        float e = timeit([&] { k_fill_bcast<0><<<blocks, 256>>>(staged, vinfo_r, rows, out); });
        printf("blocks %6d  fill seq-segments plain %.2f nt %.2f | random segments nt %.2f | bcast nt %.2f plain %.2f TB/s (output bytes only)\n",
               blocks, GB / a, GB / b, GB / c, GB / d, GB / e);
    }
    for (int flush = 0; flush < 2; flush++) {
        g_do_flush = flush;
        printf("---- Infinity Cache %s between runs\n", flush ? "FLUSHED (1 GB of plain stores)" : "left as is");
        for (size_t VV : {(size_t)585563, V}) {
            const double gb = VV * P * 16 / 1e9;
            printf("V = %zu (%.2f GB): row-per-lane nt", VV, gb);
            for (int blocks : {8192, 65536}) { float a = timeit([&] { k_fill<1><<<blocks, 256>>>(staged, vinfo_r, VV * P, out); }); printf(" [%d blocks] %.2f", blocks, gb / a); }
            printf(" | x4");
            for (int blocks : {2048, 8192, 16384}) { float a = timeit([&] { k_fill_x4<1><<<blocks, 256>>>(staged, vinfo_r, VV * P, out); }); printf(" [%d blocks] %.2f", blocks, gb / a); }
            printf(" | pref");
            for (int blocks : {1024, 2048, 4096, 8192, 16384}) { float a = timeit([&] { k_fill_pref<<<blocks, 256>>>(staged, vinfo_r, VV, out); }); printf(" [%d] %.2f", blocks, gb / a); }
            printf(" TB/s\n");
#define ROW(G) printf("   group G=%2d: bpermute %.2f  lds %.2f  readlane %.2f TB/s\n", G, gb / run_group<G, 0>(staged, vinfo_r, VV, out), \
                          gb / run_group<G, 1>(staged, vinfo_r, VV, out), gb / run_group<G, 2>(staged, vinfo_r, VV, out));
            ROW(16) ROW(64)
        }
    }
    return 0;
}
