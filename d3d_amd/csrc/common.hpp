// common.hpp -- shared host/device helpers for libd3d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "d3d_hip.h"

extern int g_d3d_last_hip_error;

#define D3D_HIP_CHECK(expr)                                        \
    do {                                                           \
        hipError_t e_ = (expr);                                    \
        if (e_ != hipSuccess) {                                    \
            g_d3d_last_hip_error = (int)e_;                        \
            return D3D_ERR_HIP;                                    \
        }                                                          \
    } while (0)
#define D3D_LAUNCH_CHECK() D3D_HIP_CHECK(hipGetLastError())

// opt-in per-kernel timing (api.hip); NAME is a string literal shared by all launches of a kernel
extern int g_d3d_prof_on;
void d3d_prof_pre(const char *name, hipStream_t st);
void d3d_prof_post(const char *name, hipStream_t st);
#define D3D_LAUNCH(NAME, kernel, grid, block, lds, st, ...)                   \
    do {                                                                      \
        if (g_d3d_prof_on) d3d_prof_pre(NAME, st);                            \
        hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);        \
        if (g_d3d_prof_on) d3d_prof_post(NAME, st);                           \
        D3D_LAUNCH_CHECK();                                                   \
    } while (0)

static inline size_t d3d_align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int64_t d3d_divup(int64_t a, int64_t b) { return (a + b - 1) / b; }

// bump allocator over the caller's workspace
struct WsCarver {
    char *base;
    size_t off, cap;
    WsCarver(void *p, size_t bytes) : base((char *)p), off(0), cap(bytes) {}
    template <typename T> T *take(size_t count)
    {
        size_t b = d3d_align_up(count * sizeof(T));
        T *r = (T *)(base ? base + off : nullptr);
        off += b;
        return r;
    }
    bool ok() const { return off <= cap; }
};

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kScanBlock = 256;      // threads per scan block (4 waves)
constexpr int kScanItems = 4;        // consecutive items per thread
constexpr int kScanTile = kScanBlock * kScanItems;

// ---------------------------------------------------------------- wave / block scans (u64)
// Inclusive scan over the wavefront on the DPP path of the vector ALUs: row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes,
// then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 -- six steps of two moves and a 64-bit add, no LDS
// traffic.  (__shfl_up is a ds_bpermute per 32-bit half and step: twelve dependent round trips through the LDS pipe, which the
// scans of k_bucket_index -- three per workgroup, on the pipe its hash table saturates -- paid in full.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long d3d_dpp_u64(unsigned long long v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROW_MASK, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xf, true);
    return ((unsigned long long)hi << 32) | lo;           // lanes without a source (row start, masked rows) read 0
}
__device__ __forceinline__ unsigned long long wave_incl_scan_u64(unsigned long long v)
{
    v += d3d_dpp_u64<0x111, 0xf>(v);                      // row_shr:1
    v += d3d_dpp_u64<0x112, 0xf>(v);                      // row_shr:2
    v += d3d_dpp_u64<0x114, 0xf>(v);                      // row_shr:4
    v += d3d_dpp_u64<0x118, 0xf>(v);                      // row_shr:8
    v += d3d_dpp_u64<0x142, 0xa>(v);                      // row_bcast:15 -> rows 1, 3
    v += d3d_dpp_u64<0x143, 0xc>(v);                      // row_bcast:31 -> rows 2, 3
    return v;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t d3d_dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    v += d3d_dpp_u32<0x111, 0xf>(v);
    v += d3d_dpp_u32<0x112, 0xf>(v);
    v += d3d_dpp_u32<0x114, 0xf>(v);
    v += d3d_dpp_u32<0x118, 0xf>(v);
    v += d3d_dpp_u32<0x142, 0xa>(v);
    v += d3d_dpp_u32<0x143, 0xc>(v);
    return v;
}
// sum over the wavefront, the same in every lane (wave-uniform: lane 63 of the scan)
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    const unsigned long long incl = wave_incl_scan_u64(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)incl, kWave - 1);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(incl >> 32), kWave - 1);
    return ((unsigned long long)hi << 32) | lo;
}

// Reductions over the wavefront on the same DPP steps; the result is valid in LANE 63 only.  A lane without a source adds
// nothing: 0.0 for the sum, its own value for max / min.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double d3d_dpp_f64(double v)
{
    const unsigned long long b = d3d_dpp_u64<CTRL, ROW_MASK>((unsigned long long)__double_as_longlong(v));
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ double wave_sum_f64_lane63(double v)
{
    v += d3d_dpp_f64<0x111, 0xf>(v);
    v += d3d_dpp_f64<0x112, 0xf>(v);
    v += d3d_dpp_f64<0x114, 0xf>(v);
    v += d3d_dpp_f64<0x118, 0xf>(v);
    v += d3d_dpp_f64<0x142, 0xa>(v);
    v += d3d_dpp_f64<0x143, 0xc>(v);
    return v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float d3d_dpp_f32_self(float v)
{
    const int b = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(b, b, CTRL, ROW_MASK, 0xf, false));
}
// IS_MAX: std::max(acc, x) = acc < x ? x : acc; else std::min(acc, x) = x < acc ? x : acc  (voxelize.cpp:146, 150)
template <bool IS_MAX>
__device__ __forceinline__ float wave_extreme_f32_lane63(float v)
{
#define D3D_EXT_STEP(CTRL, MASK)                                             \
    {                                                                        \
        const float t = d3d_dpp_f32_self<CTRL, MASK>(v);                     \
        v = IS_MAX ? (v < t ? t : v) : (t < v ? t : v);                      \
    }
    D3D_EXT_STEP(0x111, 0xf) D3D_EXT_STEP(0x112, 0xf) D3D_EXT_STEP(0x114, 0xf) D3D_EXT_STEP(0x118, 0xf)
    D3D_EXT_STEP(0x142, 0xa) D3D_EXT_STEP(0x143, 0xc)
#undef D3D_EXT_STEP
    return v;
}

// ---------------------------------------------------------------- decoupled look-back (single-pass scans across workgroups)
// One status word per tile: {state : 2 | value : 62}; state 0 = nothing yet, 1 = the tile's own total, 2 = the inclusive
// prefix up to and including the tile (status[] zeroed by an earlier kernel).  A tile publishes its total at once, then walks
// back 64 predecessors per step, adding totals until it meets an inclusive prefix -- in the steady state one or two steps,
// whatever the number of tiles (summing ALL predecessors, as chained_prefix in box.hip does for its <= 1024 chunks, grows
// with the tile number).  Tiles are numbered by a TICKET taken when the workgroup starts (lookback_ticket), not by
// blockIdx: every predecessor of a waiting tile took its ticket earlier, so it is running and publishes before it waits --
// progress wherever and in whatever order the workgroups are dispatched.  The value travels inside the word, so relaxed
// agent-scope accesses suffice.  Values are sums of up to two packed fields below 2^31 / 2^30 (no carry between them).
constexpr unsigned long long kLbAgg = 1ull << 62, kLbIncl = 2ull << 62, kLbMask = (1ull << 62) - 1;

__device__ __forceinline__ unsigned int lookback_ticket(unsigned int *ticket, unsigned int *sid /* LDS */)
{
    if (threadIdx.x == 0) *sid = atomicAdd(ticket, 1u);
    __syncthreads();
    return *sid;
}

// Bound of every cross-workgroup poll of this library (look-back predecessors, entries another wavefront publishes): TIME, not a
// poll count -- wall_clock64() ticks at 100 MHz whatever the shader clock does, and a predecessor that is merely slow (several
// processes sharing the GPU, as the 8 virtual ranks of the sharded tests do) must not be mistaken for one that never comes.  What
// never comes: a status word that was never cleared, a workspace shared by two calls in flight.  After kPollSeconds the wavefront
// traps: on ROCm that raises an HSA exception and the runtime ABORTS THE PROCESS (it does not come back as a HIP error on the
// stream) -- the alternative is a GPU hung for good, which takes every other process on it down as well.
constexpr unsigned long long kPollTicks = 8ull * 100000000ull;      // 8 s
__device__ __forceinline__ void poll_or_trap(unsigned long long &t0)
{
    const unsigned long long now = wall_clock64();
    if (t0 == 0) t0 = now;
    else if (now - t0 > kPollTicks) __builtin_trap();
    __builtin_amdgcn_s_sleep(1);
}

// all 64 lanes of ONE wavefront call it; returns (in every lane) the sum of the totals of tiles 0 .. tile - 1
__device__ __forceinline__ unsigned long long lookback_exclusive(unsigned long long *status, unsigned int tile, unsigned long long total)
{
    const int lane = threadIdx.x & (kWave - 1);
    if (tile == 0) {
        if (lane == 0) __hip_atomic_store(&status[0], kLbIncl | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0;
    }
    if (lane == 0) __hip_atomic_store(&status[tile], kLbAgg | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long excl = 0;
    long long base = (long long)tile - 1;
    unsigned long long waited = 0;
    for (;;) {
        const long long j = base - lane;
        const unsigned long long s = j >= 0 ? __hip_atomic_load(&status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kLbIncl;
        const unsigned long long incl = __ballot((s >> 62) == 2ull), empty = __ballot((s >> 62) == 0ull);
        const int p = incl ? __ffsll((long long)incl) - 1 : kWave - 1;         // the window: lanes 0 .. p
        const unsigned long long window = p == kWave - 1 ? ~0ull : ((2ull << p) - 1ull);
        if (empty & window) {                                                   // a predecessor has not published yet
            // (every predecessor holds an earlier ticket, i.e. is running: the wait is short; bounded by time, see poll_or_trap)
            poll_or_trap(waited);
            continue;
        }
        excl += wave_sum_u64(lane <= p ? (s & kLbMask) : 0ull);
        if (incl) break;
        base -= kWave;
    }
    if (lane == 0) __hip_atomic_store(&status[tile], kLbIncl | ((total + excl) & kLbMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

// exclusive scan over the block; *total = block sum.  smem: BLOCK/64 u64 entries.
template <int BLOCK>
__device__ __forceinline__ unsigned long long block_excl_scan_u64(unsigned long long v, unsigned long long *total,
                                                                  unsigned long long *smem)
{
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    unsigned long long incl = wave_incl_scan_u64(v);
    if (lane == kWave - 1) smem[w] = incl;
    __syncthreads();
    unsigned long long woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / kWave; k++) {
        unsigned long long x = smem[k];
        if (k < w) woff += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return woff + incl - v;
}

// ---------------------------------------------------------------- generic 3-kernel scan
// F provides:  u64 value(int64 i)   (count pass; may record side data)
//              u64 value2(int64 i)  (apply pass; must return the same value)
//              void apply(int64 i, u64 v, u64 excl)
// Values are "packed pairs": two u32 sums in one u64 (hi, lo) -- both stay < 2^32.
template <class F>
__global__ __launch_bounds__(kScanBlock) void k_scan_count(F f, int64_t n, unsigned long long *bsum)
{
    __shared__ unsigned long long smem[kScanBlock / kWave];
    // item layout of a tile: wavefront w owns [w * 64 * kScanItems, ...), row k of it = 64 consecutive items, one per
    // lane -> every access of f is coalesced (consecutive items per THREAD would stride the lanes by kScanItems)
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)(threadIdx.x >> 6) * (kWave * kScanItems) + (threadIdx.x & (kWave - 1));
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        int64_t i = base + (int64_t)k * kWave;
        if (i < n) s += f.value(i);
    }
    // block reduce via wave shuffles
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) s += __shfl_down(s, d, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) smem[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
#pragma unroll
        for (int k = 0; k < kScanBlock / kWave; k++) t += smem[k];
        bsum[blockIdx.x] = t;
    }
}

// single block: exclusive scan of bsum[nb] in place; writes clamped totals to counts.
// counts[idx_hi] = min(total_hi, cap_hi) (if idx_hi >= 0); counts[idx_lo] = total_lo (if idx_lo >= 0)
// Optional early read-back (see d3d_voxelize_3d_dense_notify): host[0..4) = first_counts, host[5..9) = counts, then the flag
// host[4] = 1 -- the sizes of the compaction's outputs reach the host while k_scan_apply is still writing them.
static __global__ __launch_bounds__(1024) void k_scan_bsum(unsigned long long *bsum, int64_t nb, int64_t *counts, int idx_hi,
                                                     int idx_lo, unsigned long long cap_hi, int64_t *host = nullptr,
                                                     const int64_t *first_counts = nullptr, int adopt_voxels = 0)
{
    __shared__ unsigned long long smem[1024 / kWave];
    unsigned long long carry = 0;
    // 4 consecutive block sums per thread and step (a 136 MB bitmap has 16.6 k of them: 5 steps instead of 17)
    for (int64_t c0 = 0; c0 < nb; c0 += 4096) {
        const int64_t i = c0 + (int64_t)threadIdx.x * 4;
        unsigned long long v[4], mine = 0, tot;
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k] = i + k < nb ? bsum[i + k] : 0ull; mine += v[k]; }
        unsigned long long ex = carry + block_excl_scan_u64<1024>(mine, &tot, smem);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i + k < nb) bsum[i + k] = ex;
            ex += v[k];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) {
        unsigned long long hi = carry >> 32, lo = carry & 0xffffffffull;
        if (idx_hi >= 0) counts[idx_hi] = (int64_t)(hi < cap_hi ? hi : cap_hi);
        if (idx_lo >= 0) counts[idx_lo] = (int64_t)lo;
        if (adopt_voxels) {          // fused sparse + filter: the voxel filter ran inside the index, its count lives there
            counts[D3D_COUNT_VOXELS] = first_counts[D3D_COUNT_VOXELS];
            counts[D3D_COUNT_STATUS] = 0;
            counts[D3D_COUNT_AUX] = 0;
        }
        if (host) {
            for (int k = 0; k < D3D_NUM_COUNTS; k++) {
                host[k] = first_counts ? first_counts[k] : 0;
                host[D3D_NUM_COUNTS + 1 + k] = counts[k];
            }
            __threadfence_system();
            __hip_atomic_store(&host[D3D_NUM_COUNTS], (int64_t)1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <class F>
__global__ __launch_bounds__(kScanBlock) void k_scan_apply(F f, int64_t n, const unsigned long long *bsum_excl)
{
    __shared__ unsigned long long smem[kScanBlock / kWave];
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)w * (kWave * kScanItems) + lane;   // see k_scan_count
    unsigned long long v[kScanItems], ex[kScanItems], carry = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        int64_t i = base + (int64_t)k * kWave;
        v[k] = i < n ? f.value2(i) : 0ull;
        const unsigned long long incl = wave_incl_scan_u64(v[k]);
        ex[k] = carry + incl - v[k];
        carry += __shfl(incl, kWave - 1, kWave);        // row total
    }
    if (lane == 0) smem[w] = carry;                      // wavefront total
    __syncthreads();
    unsigned long long woff = bsum_excl[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanBlock / kWave; k++)
        if (k < w) woff += smem[k];
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        int64_t i = base + (int64_t)k * kWave;
        if (i < n) f.apply(i, v[k], woff + ex[k]);
    }
}

// host driver for the trio.  bsum must hold ceil(n / kScanTile) u64.
template <class F>
static inline int d3d_run_scan(F f, int64_t n, unsigned long long *bsum, int64_t *counts, int idx_hi, int idx_lo,
                               unsigned long long cap_hi, hipStream_t st, int64_t *host = nullptr,
                               const int64_t *first_counts = nullptr, int adopt_voxels = 0)
{
    int64_t nb = d3d_divup(n, kScanTile);
    if (nb > 0) {
        D3D_LAUNCH(F::kName, k_scan_count<F>, dim3((unsigned)nb), dim3(kScanBlock), 0, st, f, n, bsum);
    }
    D3D_LAUNCH("k_scan_bsum", k_scan_bsum, dim3(1), dim3(1024), 0, st, bsum, nb, counts, idx_hi, idx_lo, cap_hi, host, first_counts,
               adopt_voxels);
    if (nb > 0) {
        D3D_LAUNCH(F::kName2, k_scan_apply<F>, dim3((unsigned)nb), dim3(kScanBlock), 0, st, f, n, bsum);
    }
    return D3D_OK;
}
