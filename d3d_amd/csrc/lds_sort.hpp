// lds_sort.hpp -- key order of the descending argsort and the workgroup-wide LDS sort shared by sort.hip (sample sort) and
// box.hip (the small-set NMS sorts its scores inside its first kernel).
#pragma once
#include "common.hpp"

// ---------------------------------------------------------------- key order
// The order both paths implement is torch's (nms.cpp:103 argsort(descending = true)): descending by VALUE, -0 == +0, every NaN
// equal to every other and greater than any number, ties in ascending index (stable).  Keys are mapped to unsigned integers
// whose ASCENDING order is that order, so the two paths cannot differ on special values.
template <typename K> struct KeyBits;
template <> struct KeyBits<float> {
    typedef uint32_t U;
    static __host__ __device__ __forceinline__ U desc(float x)
    {
        U b = __builtin_bit_cast(U, x);
        if (x != x) return 0u;                               // NaN: first
        if (b == 0x80000000u) b = 0u;                        // -0 -> +0
        const U asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
        return ~asc;
    }
};
template <> struct KeyBits<double> {
    typedef unsigned long long U;
    static __host__ __device__ __forceinline__ U desc(double x)
    {
        U b = __builtin_bit_cast(U, x);
        if (x != x) return 0ull;
        if (b == 0x8000000000000000ull) b = 0ull;
        const U asc = (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
        return ~asc;
    }
};
template <> struct KeyBits<int32_t> {                        // (internal: the sweep broad phase's x keys)
    typedef uint32_t U;
    static __host__ __device__ __forceinline__ U desc(int32_t x) { return ~((uint32_t)x ^ 0x80000000u); }
};
template <typename K> struct DescKey {
    typedef typename KeyBits<K>::U U;
    __host__ __device__ __forceinline__ U operator()(const K &x) const { return KeyBits<K>::desc(x); }
};

template <typename U> __device__ __forceinline__ bool comp_less(U da, uint32_t ia, U db, uint32_t ib)
{
    return da < db || (da == db && ia < ib);
}
__device__ __forceinline__ uint32_t shfl_xor_u(uint32_t v, int m) { return (uint32_t)__shfl_xor((int)v, m, kWave); }
__device__ __forceinline__ unsigned long long shfl_xor_u(unsigned long long v, int m)
{
    return ((unsigned long long)shfl_xor_u((uint32_t)(v >> 32), m) << 32) | shfl_xor_u((uint32_t)v, m);
}

// ascending sort of npad composites (power of two, 64 .. cap; entries are unique -- padding included) held in (d0, i0);
// (d1, i1) is the second buffer.  Returns through *rd, *ri the buffer that holds the result.  Whole workgroup; EPT = entries
// per thread (npad <= EPT * blockDim.x): with 2, a thread's two sorting networks / binary searches are interleaved, so the
// second entry rides in the latency shadow of the first (2048 entries cost ~1.2x of 1024, not 2x).
template <int EPT, typename U>
__device__ __forceinline__ void sort_lds(U *d0, uint32_t *i0, U *d1, uint32_t *i1, int npad, U **rd, uint32_t **ri)
{
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    {                                                                    // runs of 64 in registers
        U d[EPT];
        uint32_t i[EPT];
        bool on[EPT];
#pragma unroll
        for (int u = 0; u < EPT; u++) {
            const int c = wave + u * nwaves;
            on[u] = c < npad / kWave;                                    // (wave-uniform)
            d[u] = on[u] ? d0[c * kWave + lane] : (U)0;
            i[u] = on[u] ? i0[c * kWave + lane] : 0u;
        }
#pragma unroll
        for (int k = 2; k <= kWave; k <<= 1)
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
                U od[EPT];
                uint32_t oi[EPT];
#pragma unroll
                for (int u = 0; u < EPT; u++) { od[u] = shfl_xor_u(d[u], j); oi[u] = shfl_xor_u(i[u], j); }
#pragma unroll
                for (int u = 0; u < EPT; u++)
                    if (comp_less(od[u], oi[u], d[u], i[u]) == keep_min) { d[u] = od[u]; i[u] = oi[u]; }
            }
#pragma unroll
        for (int u = 0; u < EPT; u++)
            if (on[u]) { const int c = wave + u * nwaves; d0[c * kWave + lane] = d[u]; i0[c * kWave + lane] = i[u]; }
    }
    __syncthreads();
    U *sd = d0, *dd = d1;
    uint32_t *si = i0, *di = i1;
    for (int L = kWave; L < npad; L <<= 1) {
        U d[EPT];
        uint32_t i[EPT];
        int lo[EPT], e[EPT];
        const U *bd[EPT];
        const uint32_t *bi[EPT];
#pragma unroll
        for (int u = 0; u < EPT; u++) {
            e[u] = threadIdx.x + u * blockDim.x;
            if (e[u] >= npad) e[u] = threadIdx.x;                        // (idle slot: repeats entry 0's work, writes nothing)
            const int run = e[u] / L;
            bd[u] = sd + (run ^ 1) * L;
            bi[u] = si + (run ^ 1) * L;
            d[u] = sd[e[u]];
            i[u] = si[e[u]];
            lo[u] = 0;                                                   // number of sibling entries below (d, i)
        }
        for (int step = L >> 1; step > 0; step >>= 1) {
            // the searches are LDS-bandwidth bound (random 8-byte reads): the index is only fetched on equal keys
            U xd[EPT];
#pragma unroll
            for (int u = 0; u < EPT; u++) xd[u] = bd[u][lo[u] + step - 1];
#pragma unroll
            for (int u = 0; u < EPT; u++) {
                bool less = xd[u] < d[u];
                if (xd[u] == d[u]) less = bi[u][lo[u] + step - 1] < i[u];
                if (less) lo[u] += step;
            }
        }
#pragma unroll
        for (int u = 0; u < EPT; u++) {
            if (comp_less(bd[u][lo[u]], bi[u][lo[u]], d[u], i[u])) lo[u]++;      // (lo <= L - 1 here)
            if ((int)(threadIdx.x + u * blockDim.x) < npad) {
                const int run = e[u] / L, dst = (run & ~1) * L + (e[u] - run * L) + lo[u];
                dd[dst] = d[u];
                di[dst] = i[u];
            }
        }
        __syncthreads();
        U *td = sd; sd = dd; dd = td;
        uint32_t *ti = si; si = di; di = ti;
    }
    *rd = sd;
    *ri = si;
}

