// voxel.hip -- point cloud -> voxels on MI355X (gfx950).  Replaces the sequential
// std::unordered_map loops of the reference (d3d/voxel/voxelize.cpp).  Two index paths build the same intermediate
// form (one record per voxel in first-seen order, the points' rows in per-voxel segments in point order):
//
// BINNED (default, up to 16 M points; "binned index" section below)
//   partition  the points are partitioned by hash(cell) into buckets of ~512: tile histograms in LDS, a scan, ONE
//              scattered 8-byte store per point {cell, index}
//   bucket     one workgroup per bucket does everything per point in LDS: cell -> slot (CAS), count, first index,
//              segment of indices, rank in point order (by counting, early exit at max_points); the rows are gathered
//              behind those phases and written next to their rank; reduction of the overflow voxels
//   number     the reference numbers voxels by first occurrence (voxelize.cpp:119,317): voxel id = number of "first
//              points" before the voxel's own first point = a prefix count over point indices, merged with the
//              per-voxel outputs, whose stores are then coalesced by voxel id
//
// HASH TABLE (any input: C != 4, more points, huge grids, the retry after a bucket overflow)
//   insert   one point per lane, coalesced float4 loads, open-addressing hash table in HBM.  A slot is ONE
//            64-bit word {count | cell key | first point index}: the first arrival claims + counts + records
//            itself with a single CAS, later arrivals need a single atomicAdd (which also returns their
//            arrival position and the current `first`, so the rare "I am smaller" fix-up costs nothing
//            in the common case).  Scattered 8-byte requests are the scarce resource (~26 G/s, whatever the
//            operation, scope or table size: tools/atomic_bench.hip).  Dense contract: key = linear cell of the
//            grid; sparse contract: linear cell inside the frame's bounding box, which k_bbox finds on the device.
//            Two-word slots (63-bit keys) when a field of the word overflows.
//   number   two coalesced sweeps over the table: flag[first] = 1, prefix-popcount of the flags = voxel id; the same
//            sweeps allocate every voxel's index segment (scan of the counts in slot order).  No sort.
//   rank     scatter indices by arrival position, then rank = number of smaller indices in the segment
//            (early exit at max_points): exact point order without sorting or atomic chains; the points
//            themselves are staged next to their rank (C == 4).
//
// SHARED
//   meta     one lane per voxel: coords, count, pmask row and the reduction over the staged rows,
//            sequentially in point order -> bit-exact MEAN; overflow voxels in fp64.
//   fill     streaming write of voxels[V,P,C] from the staged rows (the HBM-roofline kernel).
//   filter   (sparse contract) scans over voxels and points; TRIM ranks the overflow voxels' points densely (or takes
//            the ranking from the binned index).
//
// Build: hipcc --offload-arch=gfx950 -ffp-contract=off (IEEE div, no FMA contraction:
// voxel coordinates must round exactly like the reference's CPU code).
#include "common.hpp"
#include <stdlib.h>
#include <limits.h>
#include <algorithm>
#include <utility>

namespace {

typedef unsigned long long u64;

constexpr u64 kEmpty = ~0ull;
constexpr uint32_t kInf = 0xffffffffu;        // "no first point yet" / empty chain cell (filter)
constexpr uint32_t kSingleVoxel = 0xfffffffeu; // firstmap: first (and only) point of a voxel without a record (k_bucket_index -> k_emit)
constexpr uint32_t kNoVoxel = 0xffffffffu;    // voxel dropped by max_voxels
constexpr uint32_t kNoSlot = 0xffffffffu;     // pslot: point not in any voxel
constexpr uint32_t kNoBase = 0xffffffffu;     // aux.base of a dropped voxel
constexpr int kReduceSum = 4;                 // internal: MEAN without the division (sharded partials)
constexpr int kSweepTile = 1024;              // slots per block in the table sweeps
constexpr int kFlagTile = 256 * 64;           // flags per block in k_flagpack


__device__ __forceinline__ u64 mix64(u64 h)
{
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return h;
}

// ------------------------------------------------------------------ coordinate keys
// dense contract: idx = int((p - lo) / size), 0 <= idx < shape   (voxelize.cpp:100-101)
struct DenseKey {
    static constexpr bool kBox = false;
    typedef uint32_t bin_key_t;                         // binned index: cells < 2^32 - 1 (host-checked)
    static __device__ __forceinline__ uint32_t bin_hash(u64 key)
    {
        uint32_t h = (uint32_t)key;                     // murmur3 finaliser
        h ^= h >> 16; h *= 0x85ebca6bu;
        h ^= h >> 13; h *= 0xc2b2ae35u;
        h ^= h >> 16;
        return h;
    }
    float lo[3], size[3];
    int shape[3];
    __device__ __forceinline__ bool make(const float *p, u64 &key, uint32_t &status) const
    {
        (void)status;
        int c[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            float q = (p[d] - lo[d]) / size[d];
            // x86 cvttss2si semantics of the reference build: NaN / overflow -> INT_MIN -> out of range
            if (!(q > -2147483904.0f && q < 2147483648.0f)) return false;
            int idx = (int)q;   // truncation toward zero
            if (idx < 0 || idx >= shape[d]) return false;
            c[d] = idx;
        }
        key = ((u64)c[0] * (unsigned)shape[1] + (unsigned)c[1]) * (unsigned)shape[2] + (unsigned)c[2];
        return true;
    }
    __device__ __forceinline__ void decode(u64 key, long long *c) const
    {
        c[2] = (long long)(key % (unsigned)shape[2]); key /= (unsigned)shape[2];
        c[1] = (long long)(key % (unsigned)shape[1]);
        c[0] = (long long)(key / (unsigned)shape[1]);
    }
};

// sparse contract: coord = floor(p / size), unbounded (voxelize.cpp:309); 3 x 21-bit packing.  Field value 0 is the
// reference's INT_MIN: what its (int)floor(..) yields on x86 (cvttss2si) for NaN, +-inf and |q| >= 2^31 -- real frames
// carry NaN no-return points, and the reference files them under such a voxel (which any coordinate-bound filter then
// drops).  Finite coordinates are held for (-2^20, 2^20); beyond that: COORD_OVERFLOW (or dropped, see `tolerant`).
struct SparseKey {
    static constexpr bool kBox = false;
    typedef u64 bin_key_t;
    static __device__ __forceinline__ uint32_t bin_hash(u64 key) { return (uint32_t)mix64(key); }
    float size[3];
    // a point outside the key range (non-finite, or |floor(p/size)| >= 2^20): tolerant = it simply belongs to no voxel
    // (the fused sparse + filter call, whose coordinate bounds lie inside the key range: the reference gives such a point
    // a far-away voxel, voxelize.cpp:309, that its filter then drops, :376-384); otherwise COORD_OVERFLOW is raised
    bool tolerant = false;
    __device__ __forceinline__ bool make(const float *p, u64 &key, uint32_t &status) const
    {
        u64 k = 0;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            float q = floorf(p[d] / size[d]);
            unsigned field;
            if (!(q >= -2147483648.0f && q < 2147483648.0f)) field = 0u;          // NaN / inf / beyond int: INT_MIN
            else if (!(q > -1048576.0f && q < 1048576.0f)) { if (!tolerant) status |= D3D_VOXEL_STATUS_COORD_OVERFLOW; return false; }
            else field = (unsigned)((int)q + 1048576);
            k = (k << 21) | (u64)field;
        }
        key = k;
        return true;
    }
    static __device__ __forceinline__ long long field_coord(u64 f) { return f ? (long long)f - 1048576 : -2147483648ll; }
    __device__ __forceinline__ void decode(u64 key, long long *c) const
    {
        c[2] = field_coord(key & 0x1fffff);
        c[1] = field_coord((key >> 21) & 0x1fffff);
        c[0] = field_coord((key >> 42) & 0x1fffff);
    }
};


// sparse contract FUSED with the voxel filter (d3d_voxelize_3d_sparse_filter, round 5): the filter's coordinate bounds
// (voxelize.cpp:369-379: lo <= floor(p / size) < hi on every axis) are known before the first point is read, a voxel
// outside them is dropped whatever else holds, and a point can only belong to the voxel of its own cell -- so a point outside
// the bounds belongs to no kept voxel and need not be indexed at all.  Inside, the cell linearised over the bounds box is a
// 32-bit key (host-checked: fewer than 2^32 - 1 cells): the entries, LDS tables and kernels of the DENSE contract's index
// apply (8-byte entries instead of 16, four workgroups per CU instead of two).  NaN / inf / beyond-int coordinates -- the
// reference's INT_MIN -- are outside every such box (the host requires lo > INT_MIN).
struct BoundKey {
    typedef uint32_t bin_key_t;
    static __device__ __forceinline__ uint32_t bin_hash(u64 key) { return DenseKey::bin_hash(key); }
    float size[3];
    long long lo[3];
    unsigned ext[3];                                    // hi - lo
    __device__ __forceinline__ bool make(const float *p, u64 &key, uint32_t &status) const
    {
        (void)status;
        unsigned c[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const float q = floorf(p[d] / size[d]);     // voxelize.cpp:309
            if (!(q >= -2147483648.0f && q < 2147483648.0f)) return false;
            const long long rel = (long long)(int)q - lo[d];
            if (rel < 0 || rel >= (long long)ext[d]) return false;
            c[d] = (unsigned)rel;
        }
        key = ((u64)c[0] * ext[1] + c[1]) * ext[2] + c[2];
        return true;
    }
    __device__ __forceinline__ void decode(u64 key, long long *c) const
    {
        c[2] = (long long)(key % ext[2]) + lo[2]; key /= ext[2];
        c[1] = (long long)(key % ext[1]) + lo[1];
        c[0] = (long long)(key / ext[1]) + lo[0];
    }
};

// sparse contract, one-word slots: the same coordinates, linearised inside the bounding box of the frame's voxels
// (found on the device by k_bbox, so there is no host round trip): key = ((x-x0) * Ry + (y-y0)) * Rz + (z-z0) needs
// log2(Rx Ry Rz) bits -- 25 for a KITTI frame at 0.1 m -- instead of 63, which leaves room for count and first index
// in the same 64-bit word (TabPacked) and halves the requests per point of the insertion.
struct BoxParams {
    int mn[3], mx[3];
    // key width: keys < cells <= 2^kb - 1, never all ones (wave-uniform, a handful of scalar instructions)
    __device__ __forceinline__ int key_bits() const
    {
        if (mn[0] > mx[0]) return 1;                    // no valid point
        const u64 cells = (u64)(mx[0] - mn[0] + 1) * (u64)(mx[1] - mn[1] + 1) * (u64)(mx[2] - mn[2] + 1);   // < 2^63
        return 64 - __builtin_clzll(cells);
    }
};
struct BoxKey {
    static constexpr bool kBox = true;
    float size[3];
    BoxParams *prm;
    int kb_max;               // widest key that still leaves 8 count bits
    bool tolerant = false;    // see SparseKey
    __device__ __forceinline__ bool coord(const float *p, int *c, uint32_t &status) const
    {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            float q = floorf(p[d] / size[d]);
            // non-finite / beyond int: the reference's INT_MIN voxel has no place in a bounding box -> general slots
            if (!(q >= -2147483648.0f && q < 2147483648.0f)) { status |= D3D_VOXEL_STATUS_PACK_OVERFLOW; return false; }
            if (!(q > -1048576.0f && q < 1048576.0f)) { if (!tolerant) status |= D3D_VOXEL_STATUS_COORD_OVERFLOW; return false; }
            c[d] = (int)q;
        }
        return true;
    }
    __device__ __forceinline__ bool make(const float *p, u64 &key, uint32_t &status) const
    {
        int c[3];
        if (!coord(p, c, status)) return false;
        if (prm->key_bits() > kb_max) { status |= D3D_VOXEL_STATUS_PACK_OVERFLOW; return false; }   // the caller retries
        const u64 ry = (u64)(prm->mx[1] - prm->mn[1] + 1), rz = (u64)(prm->mx[2] - prm->mn[2] + 1);
        key = ((u64)(c[0] - prm->mn[0]) * ry + (u64)(c[1] - prm->mn[1])) * rz + (u64)(c[2] - prm->mn[2]);
        return true;
    }
    __device__ __forceinline__ void decode(u64 key, long long *c) const
    {
        const u64 ry = (u64)(prm->mx[1] - prm->mn[1] + 1), rz = (u64)(prm->mx[2] - prm->mn[2] + 1);
        c[2] = (long long)(key % rz) + prm->mn[2]; key /= rz;
        c[1] = (long long)(key % ry) + prm->mn[1];
        c[0] = (long long)(key / ry) + prm->mn[0];
    }
};

// ------------------------------------------------------------------ hash tables
struct SlotInfo { bool occupied; u64 key; uint32_t first, cnt; };

// Packed table: one u64 per slot = [count : cb][key : kb][first : ib], cb = 64 - kb - ib (>= 8).
// A count that reaches 2^cb wraps out of the top of the word (nothing else is corrupted); the thread that
// causes it raises D3D_VOXEL_STATUS_PACK_OVERFLOW and the caller repeats the call with the plain table.
struct TabPacked {
    u64 *w;
    int ib, kb;
    uint32_t *fmin = nullptr;        // [cap] indices below the claimer's (kInf when none)
    const BoxParams *box = nullptr;  // BoxKey: the key width is found on the device (clamped so that shifts stay defined)
    __device__ __forceinline__ int key_bits() const
    {
        if (!box) return kb;
        const int b = box->key_bits();
        return b < 56 - ib ? b : 56 - ib;
    }
    __device__ __forceinline__ bool insert(u64 key, uint32_t i, u64 mask, uint32_t &slot, uint32_t &arrival,
                                           uint32_t &status) const
    {
        const int kb = key_bits();
        const int cs = ib + kb;
        const u64 one = 1ull << cs, imask = (1ull << ib) - 1, kmask = (1ull << kb) - 1;
        u64 h = mix64(key) & mask;
        for (u64 probe = 0; probe <= mask; probe++) {
            u64 cur = __hip_atomic_load(&w[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == kEmpty) {
                u64 old = atomicCAS(&w[h], kEmpty, one | (key << ib) | i);    // claim + count + first, at once
                if (old == kEmpty) { slot = (uint32_t)h; arrival = 0; return true; }
                cur = old;
            }
            if (((cur >> ib) & kmask) == key) {
                u64 old = atomicAdd(&w[h], one);
                arrival = (uint32_t)(old >> cs);
                if (arrival == (uint32_t)((1ull << (64 - cs)) - 1)) status |= D3D_VOXEL_STATUS_PACK_OVERFLOW;
                // an earlier point arrived later than the one that claimed the slot: the word keeps the claimer's index,
                // lower ones go to a side array with ONE atomicMin (it cannot fail; a CAS on the word is retried whenever
                // another point of a busy voxel gets counted in between).  first = min of the two (read()).
                if (i < (uint32_t)(old & imask)) atomicMin(&fmin[h], i);
                slot = (uint32_t)h;
                return true;
            }
            h = (h + 1) & mask;
        }
        status |= D3D_VOXEL_STATUS_TABLE_FULL;
        return false;
    }
    __device__ __forceinline__ SlotInfo read(u64 s) const
    {
        const int kb = key_bits();
        const u64 v = w[s];
        SlotInfo r;
        r.occupied = v != kEmpty;
        const uint32_t f2 = fmin[s];
        r.first = (uint32_t)(v & ((1ull << ib) - 1));
        if (f2 < r.first) r.first = f2;
        r.key = (v >> ib) & ((1ull << kb) - 1);
        r.cnt = (uint32_t)(v >> (ib + kb));
        return r;
    }
    __device__ __forceinline__ void clear(u64 s) const { w[s] = kEmpty; fmin[s] = kInf; }
};

// Plain table: key word + {first, count} word.  Any key < 2^64-1, any n < 2^31.
struct TabPlain {
    u64 *key;
    uint2 *fc;     // x = first, y = count
    __device__ __forceinline__ bool insert(u64 k, uint32_t i, u64 mask, uint32_t &slot, uint32_t &arrival,
                                           uint32_t &status) const
    {
        u64 h = mix64(k) & mask;
        for (u64 probe = 0; probe <= mask; probe++) {
            u64 cur = __hip_atomic_load(&key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == kEmpty) {
                u64 old = atomicCAS(&key[h], kEmpty, k);
                cur = (old == kEmpty) ? k : old;
            }
            if (cur == k) {
                arrival = atomicAdd(&fc[h].y, 1u);
                // `first` only decreases, so a stale read can only cause a redundant atomicMin
                if (__hip_atomic_load(&fc[h].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > i) atomicMin(&fc[h].x, i);
                slot = (uint32_t)h;
                return true;
            }
            h = (h + 1) & mask;
        }
        status |= D3D_VOXEL_STATUS_TABLE_FULL;
        return false;
    }
    __device__ __forceinline__ SlotInfo read(u64 s) const
    {
        SlotInfo r;
        r.key = key[s];
        r.occupied = r.key != kEmpty;
        uint2 v = fc[s];
        r.first = v.x;
        r.cnt = v.y;
        return r;
    }
    __device__ __forceinline__ void clear(u64 s) const
    {
        key[s] = kEmpty;
        fc[s] = make_uint2(kInf, 0u);
    }
};

// ------------------------------------------------------------------ kernels: table build
// Sparse contract, ANY int32 coordinates (D3D_VOXEL_WIDE_KEYS; round 5): the reference keys a voxel by the three ints
// (int)floor(p / size) (voxelize.cpp:309-313), whatever their size -- 1 mm voxels a kilometre from the origin are beyond the
// 3 x 21 bits of SparseKey.  Here the table compares all 96 bits: slot = {w1 = 1 | x : 32 | y >> 1 : 31, w2 = 1 | y & 1 | z : 32,
// {first, count}}.  A point claims with one CAS on w1 and then publishes w2; a point that finds ITS w1 reads w2 (waiting, when
// the claimer -- another wavefront, or a lane of this one that has already issued the store -- has not published yet) and
// moves on to the next slot when the third coordinate differs.  The voxels' coordinates are not decoded from a key afterwards
// but recomputed from their first points (k_wide_coords).  A fallback for frames SparseKey refuses: not a fast path.
struct WideKey {
    static constexpr bool kBox = false;
    float size[3];
    static __device__ __forceinline__ int coord(float p, float sz)
    {
        const float q = floorf(p / sz);
        return (q >= -2147483648.0f && q < 2147483648.0f) ? (int)q : INT_MIN;       // x86 cvttss2si of the reference build
    }
    __device__ __forceinline__ bool make(const float *p, u64 &key, uint32_t &status) const
    {
        (void)status;
        const u64 a = (u64)(uint32_t)coord(p[0], size[0]), b = (u64)(uint32_t)coord(p[1], size[1]), c = (u64)(uint32_t)coord(p[2], size[2]);
        key = mix64((a << 32 | b) ^ mix64(c + 0x9e3779b97f4a7c15ull));      // where the probe starts; the table compares the coordinates
        return true;
    }
    __device__ __forceinline__ void decode(u64, long long *c) const { c[0] = c[1] = c[2] = 0; }     // (never used: k_wide_coords)
};
struct TabWide {
    u64 *w1, *w2;
    uint2 *fc;        // x = first, y = count
    const float *points;
    int c;
    float size[3];
    __device__ __forceinline__ bool insert(u64 k, uint32_t i, u64 mask, uint32_t &slot, uint32_t &arrival, uint32_t &status) const
    {
        const float *p = points + (size_t)i * c;
        const uint32_t x = (uint32_t)WideKey::coord(p[0], size[0]), y = (uint32_t)WideKey::coord(p[1], size[1]),
                       z = (uint32_t)WideKey::coord(p[2], size[2]);
        const u64 v1 = (1ull << 63) | ((u64)x << 31) | (u64)(y >> 1), v2 = (1ull << 63) | ((u64)(y & 1u) << 32) | (u64)z;
        u64 h = k & mask;
        for (u64 probe = 0; probe <= mask; probe++) {
            u64 cur = __hip_atomic_load(&w1[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == 0ull) {
                const u64 old = atomicCAS(&w1[h], 0ull, v1);
                if (old == 0ull) __hip_atomic_store(&w2[h], v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cur = old == 0ull ? v1 : old;
            }
            if (cur == v1) {                            // same x and y: the slot's z decides
                u64 b;
                unsigned long long waited = 0;
                while ((b = __hip_atomic_load(&w2[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0ull) poll_or_trap(waited);
                if (b == v2) {
                    arrival = atomicAdd(&fc[h].y, 1u);
                    if (__hip_atomic_load(&fc[h].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > i) atomicMin(&fc[h].x, i);
                    slot = (uint32_t)h;
                    return true;
                }
            }
            h = (h + 1) & mask;
        }
        status |= D3D_VOXEL_STATUS_TABLE_FULL;
        return false;
    }
    __device__ __forceinline__ SlotInfo read(u64 s) const
    {
        SlotInfo r;
        r.key = w1[s];
        r.occupied = r.key != 0ull;
        const uint2 v = fc[s];
        r.first = v.x;
        r.cnt = v.y;
        return r;
    }
    __device__ __forceinline__ void clear(u64 s) const
    {
        w1[s] = 0ull;
        w2[s] = 0ull;
        fc[s] = make_uint2(kInf, 0u);
    }
};

template <class Tab>
__global__ void k_init(Tab tab, int64_t cap, unsigned char *flags, int64_t nflags16, int64_t *counts, uint32_t *big_count,
                       BoxParams *box = nullptr, const float4 *warm = nullptr, int64_t nwarm = 0)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < cap; i += stride) tab.clear((u64)i);
    uint4 z = make_uint4(0, 0, 0, 0);
    for (int64_t i = t0; i < nflags16; i += stride) reinterpret_cast<uint4 *>(flags)[i] = z;
    if (t0 < D3D_NUM_COUNTS) counts[t0] = 0;
    if (t0 == 0) *big_count = 0;
    // pull the point tensor into the Infinity Cache while the table is being cleared: k_insert is a chain of dependent
    // requests per point (load -> probe -> atomic) and runs 20 % faster when its first link does not come from HBM
    // (measured: 100 -> 80 us at 1 M points, for 2 us more here)
    float acc = 0.f;
    for (int64_t i = t0; i < nwarm; i += stride) { const float4 v = warm[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1.2345e-30f) flags[0] = 1;               // never true; keeps the loads
    if (box && t0 == 0) {
        for (int d = 0; d < 3; d++) { box->mn[d] = INT_MAX; box->mx[d] = INT_MIN; }
        big_count[32] = 0;                              // k_bbox's ticket
    }
}

// bounding box of the frame's voxel coordinates (grid-stride)
template <bool VEC4>
__global__ __launch_bounds__(256) void k_bbox(BoxKey kf, const float *__restrict__ points, int64_t n, int c, int64_t *counts,
                                              int *partial, unsigned int *ticket)
{
    __shared__ int smn[4][3], smx[4][3];
    int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
    uint32_t status = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {
        float p[4][3];
#pragma unroll
        for (int u = 0; u < 4; u++) {                      // four independent loads in flight
            const int64_t i = i0 + u * stride < n ? i0 + u * stride : i0;
            if (VEC4) {
                float4 v = reinterpret_cast<const float4 *>(points)[i];
                p[u][0] = v.x; p[u][1] = v.y; p[u][2] = v.z;
            } else {
                const float *src = points + i * c;
                p[u][0] = src[0]; p[u][1] = src[1]; p[u][2] = src[2];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int q[3];
            if (kf.coord(p[u], q, status)) {
#pragma unroll
                for (int d = 0; d < 3; d++) { mn[d] = q[d] < mn[d] ? q[d] : mn[d]; mx[d] = q[d] > mx[d] ? q[d] : mx[d]; }
            }
        }
    }
#pragma unroll
    for (int d = 0; d < 3; d++)
        for (int o = kWave / 2; o > 0; o >>= 1) {
            const int a = __shfl_xor(mn[d], o, kWave), b = __shfl_xor(mx[d], o, kWave);
            mn[d] = a < mn[d] ? a : mn[d];
            mx[d] = b > mx[d] ? b : mx[d];
        }
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    if (lane == 0)
        for (int d = 0; d < 3; d++) { smn[w][d] = mn[d]; smx[w][d] = mx[d]; }
    __syncthreads();
    // per-workgroup partial boxes, then the LAST workgroup to finish (one ticket atomic each) folds them: atomics on one
    // cache line are serialised at ~7 ns each, and with atomicMin/Max on the box itself every workgroup of the first
    // wave issues all six of them (measured: 47 us for 489 workgroups)
    __shared__ bool last;
    if (threadIdx.x == 0) {
        int *mine = partial + (size_t)blockIdx.x * 6;
        for (int d = 0; d < 3; d++) {
            int a = smn[0][d], b = smx[0][d];
            for (int k = 1; k < 4; k++) { a = smn[k][d] < a ? smn[k][d] : a; b = smx[k][d] > b ? smx[k][d] : b; }
            __hip_atomic_store(&mine[d], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&mine[3 + d], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last) {
        __threadfence();
        int a[3] = {INT_MAX, INT_MAX, INT_MAX}, b[3] = {INT_MIN, INT_MIN, INT_MIN};
        for (unsigned int k = threadIdx.x; k < gridDim.x; k += blockDim.x)
            for (int d = 0; d < 3; d++) {
                const int x = __hip_atomic_load(&partial[(size_t)k * 6 + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int y = __hip_atomic_load(&partial[(size_t)k * 6 + 3 + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a[d] = x < a[d] ? x : a[d];
                b[d] = y > b[d] ? y : b[d];
            }
#pragma unroll
        for (int d = 0; d < 3; d++)
            for (int o = kWave / 2; o > 0; o >>= 1) {
                const int x = __shfl_xor(a[d], o, kWave), y = __shfl_xor(b[d], o, kWave);
                a[d] = x < a[d] ? x : a[d];
                b[d] = y > b[d] ? y : b[d];
            }
        __syncthreads();
        if (lane == 0)
            for (int d = 0; d < 3; d++) { smn[w][d] = a[d]; smx[w][d] = b[d]; }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int d = 0; d < 3; d++) {
                int x = smn[0][d], y = smx[0][d];
                for (int k = 1; k < 4; k++) { x = smn[k][d] < x ? smn[k][d] : x; y = smx[k][d] > y ? smx[k][d] : y; }
                kf.prm->mn[d] = x;
                kf.prm->mx[d] = y;
            }
    }
    if (status) atomicOr(reinterpret_cast<u64 *>(&counts[D3D_COUNT_STATUS]), (u64)status);
}

template <class Key, class Tab, bool VEC4>
__global__ __launch_bounds__(256) void k_insert(Key kf, Tab tab, const float *__restrict__ points, int64_t n, int c,
                                                u64 mask, uint32_t *pslot, uint32_t *parr, int64_t *counts)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t slot = kNoSlot, status = 0, arrival = 0;
    float p[3];
    if (VEC4) {
        float4 v = reinterpret_cast<const float4 *>(points)[i];
        p[0] = v.x; p[1] = v.y; p[2] = v.z;
    } else {
        const float *src = points + i * c;
        p[0] = src[0]; p[1] = src[1]; p[2] = src[2];
    }
    u64 key;
    if (kf.make(p, key, status)) tab.insert(key, (uint32_t)i, mask, slot, arrival, status);
    pslot[i] = slot;
    if (parr) parr[i] = arrival;
    if (status) atomicOr(reinterpret_cast<u64 *>(&counts[D3D_COUNT_STATUS]), (u64)status);
}

// ------------------------------------------------------------------ kernels: numbering by table sweeps
// sweep 1: flag the first point of every voxel; per-block sums of the counts (list space)
template <class Tab>
__global__ __launch_bounds__(256) void k_sweep1(Tab tab, int64_t cap, unsigned char *flags, uint32_t *bsumA)
{
    __shared__ u64 smem[256 / kWave];
    const int64_t s0 = (int64_t)blockIdx.x * kSweepTile + (int64_t)threadIdx.x * 4;
    u64 sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int64_t s = s0 + k;
        if (s < cap) {
            SlotInfo si = tab.read((u64)s);
            if (si.occupied) { flags[si.first] = 1; sum += si.cnt; }
        }
    }
    u64 tot;
    (void)block_excl_scan_u64<256>(sum, &tot, smem);
    if (threadIdx.x == 0) bsumA[blockIdx.x] = (uint32_t)tot;
}

// 64 flag bytes -> one bit word; word popcounts scanned inside the block (fwpre), block totals -> bsumF
__global__ __launch_bounds__(256) void k_flagpack(const unsigned char *__restrict__ flags, int64_t nwords, u64 *fwords,
                                                  uint32_t *fwpre, uint32_t *bsumF)
{
    __shared__ u64 smem[256 / kWave];
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    u64 bits = 0;
    if (t < nwords) {
        const uint4 *src = reinterpret_cast<const uint4 *>(flags + t * 64);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint4 v = src[q];
            const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // bytes are 0/1: gather the low bit of each of the 4 bytes
                uint32_t x = wv[j];
                u64 nib = (x & 1u) | ((x >> 7) & 2u) | ((x >> 14) & 4u) | ((x >> 21) & 8u);
                bits |= nib << (q * 16 + j * 4);
            }
        }
        fwords[t] = bits;
    }
    u64 tot;
    u64 ex = block_excl_scan_u64<256>((u64)__popcll(bits), &tot, smem);
    if (t < nwords) fwpre[t] = (uint32_t)ex;
    if (threadIdx.x == 0) bsumF[blockIdx.x] = (uint32_t)tot;
}

// single block: exclusive scans of the two small block-sum arrays; totals -> counts
__global__ __launch_bounds__(1024) void k_scan2(uint32_t *bsumF, int64_t nF, uint32_t *bsumA, int64_t nA, int64_t *counts,
                                                u64 max_voxels)
{
    __shared__ u64 smem[1024 / kWave];
    for (int pass = 0; pass < 2; pass++) {
        uint32_t *a = pass == 0 ? bsumF : bsumA;
        const int64_t m = pass == 0 ? nF : nA;
        u64 carry = 0;
        for (int64_t c0 = 0; c0 < m; c0 += 1024) {
            const int64_t i = c0 + threadIdx.x;
            u64 v = i < m ? a[i] : 0ull, tot;
            u64 ex = block_excl_scan_u64<1024>(v, &tot, smem);
            if (i < m) a[i] = (uint32_t)(carry + ex);
            carry += tot;
        }
        if (threadIdx.x == 0) {
            if (pass == 0) counts[D3D_COUNT_VOXELS] = (int64_t)(carry < max_voxels ? carry : max_voxels);
            else counts[D3D_COUNT_AUX] = (int64_t)carry;
        }
        __syncthreads();
    }
}

struct NumberOut {
    u64 *aux;             // [cap] {base : lo32, count : hi32}; base == kNoBase -> voxel dropped
    uint32_t *vidarr;     // [cap] voxel id per slot (optional)
    uint4 *vinfo;         // [V] {key lo, key hi, segment base, count}: ONE scattered 16-byte store per voxel;
                          // coords / npoints are produced from it later in voxel order (coalesced)
    int64_t *first_out;   // [V] optional
    int64_t index_offset;
    uint32_t max_voxels;
};

// sweep 2: voxel id = rank of `first` among the flags; segment base = exclusive scan of counts in slot order
template <class Tab>
__global__ __launch_bounds__(256) void k_sweep2(Tab tab, int64_t cap, const u64 *__restrict__ fwords,
                                                const uint32_t *__restrict__ fwpre, const uint32_t *__restrict__ bsumF,
                                                const uint32_t *__restrict__ bsumA, NumberOut o)
{
    __shared__ u64 smem[256 / kWave];
    const int64_t s0 = (int64_t)blockIdx.x * kSweepTile + (int64_t)threadIdx.x * 4;
    SlotInfo si[4];
    u64 sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int64_t s = s0 + k;
        si[k].occupied = false;
        if (s < cap) si[k] = tab.read((u64)s);
        if (si[k].occupied) sum += si[k].cnt;
    }
    u64 tot;
    u64 base = block_excl_scan_u64<256>(sum, &tot, smem) + bsumA[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (!si[k].occupied) continue;
        const int64_t s = s0 + k;
        const uint32_t f = si[k].first, wi = f >> 6;
        const uint32_t vid = bsumF[f / kFlagTile] + fwpre[wi] + (uint32_t)__popcll(fwords[wi] & ((1ull << (f & 63)) - 1));
        if (vid < o.max_voxels) {                       // voxelize.cpp:116-117: later voxels are never created
            o.aux[s] = base | ((u64)si[k].cnt << 32);
            if (o.vidarr) o.vidarr[s] = vid;
            o.vinfo[vid] = make_uint4((uint32_t)si[k].key, (uint32_t)(si[k].key >> 32), (uint32_t)base, si[k].cnt);
            if (o.first_out) o.first_out[vid] = o.index_offset + f;
        } else {
            o.aux[s] = (u64)kNoBase | ((u64)si[k].cnt << 32);
            if (o.vidarr) o.vidarr[s] = kNoVoxel;
        }
        base += si[k].cnt;
    }
}

__global__ __launch_bounds__(256) void k_map(const uint32_t *__restrict__ vidarr, const uint32_t *__restrict__ pslot,
                                             int64_t n, int64_t *mapping)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t ps = pslot[i];
    long long m = -1;
    if (ps != kNoSlot) {
        const uint32_t vid = vidarr[ps];
        if (vid != kNoVoxel) m = (long long)vid;
    }
    mapping[i] = m;
}

// ------------------------------------------------------------------ kernels: ranking
//   k_scatter  every point drops its index at unsorted[base + arrival]  (arrival order is arbitrary)
//   k_select   every point counts the indices smaller than its own in its voxel's (contiguous, L2-hot)
//              segment; that count IS its rank in point order.  It stops as soon as max_points smaller
//              ones were seen (the point is then not among the first max_points, voxelize.cpp:128-134),
//              so a voxel of c points costs O(c * max_points) loads when arrival order is roughly
//              index order.  Ranks < max_points land in sorted[base + rank].
__global__ __launch_bounds__(256) void k_scatter(int64_t n, const u64 *__restrict__ aux, uint32_t *__restrict__ pslot,
                                                 uint32_t *__restrict__ parr, uint32_t *unsorted, uint32_t *sorted,
                                                 const float4 *__restrict__ points4, float4 *staged)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t ps = pslot[i];
    uint32_t todo = 0, base = 0;
    if (ps != kNoSlot) {
        const u64 a = aux[ps];                 // the one random access of this pass (8 bytes)
        base = (uint32_t)a;
        const uint32_t cnt = (uint32_t)(a >> 32);
        if (base != kNoBase) {
            if (cnt == 1) { if (staged) staged[base] = points4[i]; else sorted[base] = (uint32_t)i; }
            else { unsorted[base + parr[i]] = (uint32_t)i; todo = cnt; }
        }
    }
    // hand (segment length, segment base) to k_select through the per-point arrays: coalesced there
    pslot[i] = todo;
    parr[i] = base;
}

// When `staged` is given (C == 4) the point's row is copied next to its rank as well, so that the fill and
// reduction kernels read contiguous rows instead of chasing sorted[] -> points[] (one coalesced read here
// replaces two dependent random gathers there).
__global__ __launch_bounds__(256) void k_select(int64_t n, const uint32_t *__restrict__ pcnt,
                                                const uint32_t *__restrict__ pbase,
                                                const uint32_t *__restrict__ unsorted, uint32_t *sorted,
                                                uint32_t max_points, const float4 *__restrict__ points4, float4 *staged)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t cnt = pcnt[i];
    if (cnt == 0) return;
    const uint32_t base = pbase[i];
    const uint32_t *seg = unsorted + base;
    const uint32_t me = (uint32_t)i;
    uint32_t rank = 0, k = 0;
    for (; k + 4 <= cnt && rank < max_points; k += 4) {   // 4 independent loads per exit test
        uint32_t a0 = seg[k], a1 = seg[k + 1], a2 = seg[k + 2], a3 = seg[k + 3];
        rank += (a0 < me) + (a1 < me) + (a2 < me) + (a3 < me);
    }
    for (; k < cnt && rank < max_points; k++) rank += seg[k] < me;
    if (rank < max_points) {
        // the index list is only read when the rows are not staged (C != 4): one scattered request per point less
        if (staged) staged[base + rank] = points4[i];
        else sorted[base + rank] = me;
    }
}

// pmask[V,P] bytes: pmask[v,k] = k < min(npoints[v], P).  (The reference leaves the
// False entries uninitialised, voxelize.cpp:58; we define them.)  16 bytes per lane.
__global__ __launch_bounds__(256) void k_pmask(const int64_t *__restrict__ counts, const int32_t *__restrict__ npoints,
                                               uint32_t max_points, uint8_t *pmask)
{
    const int64_t total = counts[D3D_COUNT_VOXELS] * (int64_t)max_points;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q * 16 < total; q += stride) {
        const int64_t b0 = q * 16;
        uint32_t w[4] = {0, 0, 0, 0};
        int64_t v = b0 / max_points;
        uint32_t k = (uint32_t)(b0 - v * max_points);
        uint32_t cnt = (uint32_t)npoints[v];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (b0 + j < total) {
                if (k < cnt) w[j >> 2] |= 1u << ((j & 3) * 8);
                if (++k == max_points) { k = 0; v++; if (b0 + j + 1 < total) cnt = (uint32_t)npoints[v]; }
            }
        }
        if (b0 + 16 <= total) reinterpret_cast<uint4 *>(pmask)[q] = make_uint4(w[0], w[1], w[2], w[3]);
        else
            for (int j = 0; b0 + j < total; j++) pmask[b0 + j] = (uint8_t)((w[j >> 2] >> ((j & 3) * 8)) & 0xff);
    }
}

// aggregates[V,C]: one (voxel, channel) per lane.  Voxels with cnt <= P are reduced
// sequentially in point order from their sorted list -> bit-identical to the reference's
// loop (voxelize.cpp:137-164).
__global__ __launch_bounds__(256) void k_aggregate(const float *__restrict__ points, int c,
                                                   const int64_t *__restrict__ counts,
                                                   const int32_t *__restrict__ npoints,
                                                   const uint32_t *__restrict__ voff, const uint32_t *__restrict__ list,
                                                   const uint32_t *__restrict__ unsorted, uint32_t max_points,
                                                   int reduction, float *agg)
{
    const int64_t total = counts[D3D_COUNT_VOXELS] * (int64_t)c;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int lane = threadIdx.x & (kWave - 1);
    const bool is_sum = reduction == D3D_REDUCE_MEAN || reduction == kReduceSum;
    // `wbase` is wave-uniform so that the whole wavefront stays in the loop for the cooperative part
    for (int64_t wbase = (int64_t)blockIdx.x * blockDim.x + threadIdx.x - lane; wbase < total; wbase += stride) {
        const int64_t t = wbase + lane;
        const bool valid = t < total;
        const int64_t v = valid ? t / c : 0;
        const int d = (int)(t - v * c);
        const uint32_t cnt = valid ? (uint32_t)npoints[v] : 0u;
        if (valid && cnt <= max_points) {
            const uint32_t base = voff[v];
            float acc = is_sum ? 0.0f : (reduction == D3D_REDUCE_MAX ? -INFINITY : INFINITY);
            for (uint32_t k = 0; k < cnt; k++) {
                float x = points[(int64_t)list[base + k] * c + d];
                if (is_sum) acc += x;
                else if (reduction == D3D_REDUCE_MAX) acc = acc < x ? x : acc;   // std::max(acc, x)
                else acc = x < acc ? x : acc;                                      // std::min(acc, x)
            }
            if (reduction == D3D_REDUCE_MEAN) acc = acc / (float)(int32_t)cnt;   // voxelize.cpp:164 (float / int)
            agg[t] = acc;
        }
        // Overflow voxels: every point counts (voxelize.cpp:137-157) but only max_points are ranked, so the
        // wavefront walks the arrival-ordered segment together, 64 entries per step.  MEAN accumulates in fp64
        // (insensitive to the arrival order to ~1e-16 => reproducible; differs from the reference's fp32
        // running sum by rounding only).
        unsigned long long big = __ballot(valid && cnt > max_points);
        while (big) {
            const int l = __builtin_ctzll(big);
            big &= big - 1;
            const int64_t vv = __shfl((long long)v, l, kWave);
            const int dd = __shfl(d, l, kWave);
            const uint32_t cc = __shfl(cnt, l, kWave);
            const uint32_t *seg = unsorted + voff[vv];
            double sum = 0.0;
            float ext = reduction == D3D_REDUCE_MAX ? -INFINITY : INFINITY;
            for (uint32_t k = lane; k < cc; k += kWave) {
                float x = points[(int64_t)seg[k] * c + dd];
                if (is_sum) sum += (double)x;
                else if (reduction == D3D_REDUCE_MAX) ext = ext < x ? x : ext;
                else ext = x < ext ? x : ext;
            }
#pragma unroll
            for (int o = kWave / 2; o > 0; o >>= 1) {
                double s2 = __shfl_xor(sum, o, kWave);
                float e2 = __shfl_xor(ext, o, kWave);
                sum += s2;
                if (reduction == D3D_REDUCE_MAX) ext = ext < e2 ? e2 : ext;
                else ext = e2 < ext ? e2 : ext;
            }
            if (lane == l)
                agg[t] = reduction == D3D_REDUCE_MEAN ? (float)sum / (float)(int32_t)cc : (is_sum ? (float)sum : ext);
        }
    }
}


// ------------------------------------------------------------------ kernels: outputs
// voxels[V, P, 4] -- the HBM-roofline kernel of the dense contract: 16 P bytes written per voxel, ~1.4 rows of 16 bytes
// read.  One wavefront owns a GROUP of G consecutive voxels (G = 16 / 32 / 64, chosen by the host so that a launch has
// >= 32 k wavefronts), i.e. G * P rows = one contiguous stretch of the output:
//   1. lane l < G loads the record of voxel v0 + l (one coalesced request) and gathers that voxel's FIRST row -- G
//      independent 16-byte gathers in ONE instruction.  ~80 % of a LiDAR frame's voxels hold a single point, so this one
//      instruction covers most of what the kernel reads;
//   2. the stretch is written 64 rows (1 KiB) per step with `global_store_dwordx4 nt`; the lane that holds row `slot` of
//      voxel j takes count / base / first row from lane j by `ds_bpermute` (no memory), and only rows 1.. of the multi-point
//      voxels are gathered inside the loop, four steps' worth in flight before their four stores.
// For frames whose intermediate arrays have left the Infinity Cache by the time the fill starts (more than ~2 M points).
// The row-per-lane form below (record load -> dependent row gather -> store, one chain per wavefront and KiB) then runs
// at 3.5 TB/s on config 5's 3 GB output -- latency-bound: 8192 resident wavefronts x 1 KiB per ~2.5 us chain -- against
// 6.5 TB/s for bare nt stores on the same box; this form reaches 5.5 TB/s there (tools/fill_bench.hip reproduces both with
// the cache flushed between runs: 3.2-3.3 vs 4.7-5.1 TB/s; profiles/r02_fill_bench.txt).
template <int G>
__global__ __launch_bounds__(256) void k_fill_c4(const float4 *__restrict__ staged, const int64_t *__restrict__ counts,
                                                 const uint4 *__restrict__ vinfo, uint32_t P, int pshift /* log2 P or -1 */,
                                                 float4 *voxels)
{
    typedef float vec4 __attribute__((ext_vector_type(4)));
    const int64_t V = counts[D3D_COUNT_VOXELS];
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t group = (int64_t)blockIdx.x * (256 / kWave) + (threadIdx.x >> 6);
    const int64_t v0 = group * G;
    if (v0 >= V) return;
    const uint32_t nv = V - v0 < G ? (uint32_t)(V - v0) : (uint32_t)G;        // voxels of this group
    uint32_t base = 0, cnt = 0;
    vec4 first = {0.f, 0.f, 0.f, 0.f};
    if ((uint32_t)lane < nv) {
        const uint4 vi = vinfo[v0 + lane];
        base = vi.z;
        cnt = vi.w;
        if (cnt > 0) first = *reinterpret_cast<const vec4 *>(&staged[base]);
    }
    const uint32_t nrows = nv * P;                                             // <= 64 P
    vec4 *out = reinterpret_cast<vec4 *>(voxels) + v0 * (int64_t)P;
    for (uint32_t q0 = 0; q0 < nrows; q0 += 4 * kWave) {
        vec4 val[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t q = q0 + u * kWave + lane;
            // rows past the group's end compute with voxel 0's data and are not stored
            const uint32_t qq = q < nrows ? q : 0u;
            const uint32_t j = pshift >= 0 ? (qq >> pshift) : qq / P;
            const uint32_t slot = qq - j * P;
            const uint32_t c = (uint32_t)__shfl((int)cnt, (int)j, kWave);
            const uint32_t b = (uint32_t)__shfl((int)base, (int)j, kWave);
            vec4 f;
            f.x = __shfl(first.x, (int)j, kWave); f.y = __shfl(first.y, (int)j, kWave);
            f.z = __shfl(first.z, (int)j, kWave); f.w = __shfl(first.w, (int)j, kWave);
            const vec4 zero = {0.f, 0.f, 0.f, 0.f};
            val[u] = (slot == 0 && c > 0) ? f : zero;
            if (slot > 0 && slot < c) val[u] = *reinterpret_cast<const vec4 *>(&staged[b + slot]);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t q = q0 + u * kWave + lane;
            if (q < nrows) __builtin_nontemporal_store(val[u], &out[q]);
        }
    }
}

// The same output, one 16-byte row per lane, grid-stride: the whole chip writes ONE compact moving window (the DRAM-friendliest
// store order: 6.2-6.5 TB/s on a 300 MB output) -- the faster form while the frame's staged rows and records are still
// cache-resident when the fill starts (config 2: 47-49 us for 324 MB; the group form 56 us, its 9 k-37 k short-lived
// wavefronts leave a tail).
__global__ __launch_bounds__(256) void k_fill_c4_rows(const float4 *__restrict__ staged, const int64_t *__restrict__ counts,
                                                      const uint4 *__restrict__ vinfo, uint32_t max_points, float4 *voxels)
{
    typedef float vec4 __attribute__((ext_vector_type(4)));
    const int64_t rows = counts[D3D_COUNT_VOXELS] * (int64_t)max_points;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // row -> (voxel, slot): a shift when max_points is a power of two, a 32-bit division while the row index fits, the
    // 64-bit division (~100 instructions per row: a third of this kernel's issue slots at config 2) only beyond that
    const int sh = (max_points & (max_points - 1)) == 0 ? __builtin_ctz(max_points) : -1;
    const bool small = rows < (1ll << 32);
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
        const int64_t v = sh >= 0 ? (r >> sh) : (small ? (int64_t)((uint32_t)r / max_points) : r / max_points);
        const uint32_t k = (uint32_t)(r - v * max_points);
        const uint4 vi = vinfo[v];
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < vi.w) val = staged[vi.z + k];
        vec4 x = {val.x, val.y, val.z, val.w};
        __builtin_nontemporal_store(x, reinterpret_cast<vec4 *>(&voxels[r]));
    }
}

// generic C: one row of C floats per lane (one index load and one division per row; a wavefront's rows are contiguous
// in memory, so its C strided store instructions fill whole cache lines between them).  Writing one dword per lane
// contiguously instead (source index of a float's row fetched by shuffle) was slower: 136 vs 124 us at C = 5.
__global__ __launch_bounds__(256) void k_fill_generic(const float *__restrict__ points, int c,
                                                      const int64_t *__restrict__ counts,
                                                      const uint4 *__restrict__ vinfo,
                                                      const uint32_t *__restrict__ sorted, uint32_t max_points,
                                                      float *voxels)
{
    const int64_t rows = counts[D3D_COUNT_VOXELS] * (int64_t)max_points;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
        const int64_t v = r / max_points;
        const uint32_t k = (uint32_t)(r - v * max_points);
        const uint4 vi = vinfo[v];
        float *dst = voxels + r * c;
        if (k < vi.w) {
            const float *src = points + (int64_t)sorted[vi.z + k] * c;
            for (int d = 0; d < c; d++) dst[d] = src[d];
        } else {
            for (int d = 0; d < c; d++) dst[d] = 0.f;
        }
    }
}

// generic C up to 16: a wavefront gathers the 64 rows it owns into LDS (lane = row: C contiguous floats of its source
// point, or zeros) and writes the 64 C floats out linearly, 16 bytes per lane -- coalesced nontemporal stores for any C
// (the row-per-lane kernel above issues C strided 4-byte stores per row: 124 us at C = 5, this one 109 us).
constexpr int kFillLdsMaxC = 16;
constexpr int64_t kFillRowsMaxPoints = 2 << 20;     // frames up to this many points take k_fill_c4_rows (see there)
__global__ __launch_bounds__(256) void k_fill_generic_lds(const float *__restrict__ points, int c,
                                                          const int64_t *__restrict__ counts,
                                                          const uint4 *__restrict__ vinfo,
                                                          const uint32_t *__restrict__ sorted, uint32_t max_points,
                                                          float *voxels)
{
    typedef float vec4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float stage[4][kWave * kFillLdsMaxC];
    const int64_t rows = counts[D3D_COUNT_VOXELS] * (int64_t)max_points;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int lane = threadIdx.x & (kWave - 1);
    float *buf = stage[threadIdx.x >> 6];
    for (int64_t r0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x - lane; r0 < rows; r0 += stride) {   // wave-uniform
        const int64_t r = r0 + lane;
        const float *src = nullptr;
        if (r < rows) {
            const int64_t v = r / max_points;
            const uint32_t k = (uint32_t)(r - v * max_points);
            const uint4 vi = vinfo[v];
            if (k < vi.w) src = points + (int64_t)sorted[vi.z + k] * c;
        }
        for (int d = 0; d < c; d++) buf[lane * c + d] = src ? src[d] : 0.f;
        __builtin_amdgcn_wave_barrier();              // LDS ops of one wavefront complete in order
        const int nrows = rows - r0 < kWave ? (int)(rows - r0) : kWave;
        const int nfl = nrows * c;                    // floats of this chunk; r0 * c * 4 bytes is a multiple of 16 (r0 % 64 == 0)
        float *dst = voxels + r0 * c;
        for (int j = lane * 4; j < nfl; j += kWave * 4) {
            if (j + 4 <= nfl) {
                const vec4 x = *reinterpret_cast<const vec4 *>(&buf[j]);
                __builtin_nontemporal_store(x, reinterpret_cast<vec4 *>(&dst[j]));
            } else {
                for (int t = j; t < nfl; t++) dst[t] = buf[t];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Per-voxel outputs in voxel order, one lane per voxel: coords, npoints, (C == 4:) the reduction as one float4
// and (P % 16 == 0:) the pmask row.  Voxels with <= P points are reduced sequentially in point order from their
// sorted list -> bit-identical to the reference's loop (voxelize.cpp:137-164).  Overflow voxels: every point
// counts (voxelize.cpp:137-157) but only P are ranked, so the wavefront walks the arrival-ordered segment
// together; MEAN accumulates in fp64 (insensitive to the arrival order to ~1e-16 => reproducible).
template <class Key, bool AGG4>
__device__ __forceinline__ void meta_voxel(const Key &kf, int64_t v, const uint4 vi, const float4 *__restrict__ staged,
                                           uint32_t P, int reduction, int64_t *coords, int32_t *npoints, uint32_t *voff,
                                           unsigned char *pmask, float4 *agg, uint32_t *big_list, uint32_t *big_count,
                                           int64_t *keys_out, uint32_t npoints_clamp = 0xffffffffu,
                                           const long long *coord_sub = nullptr /* VoxelGenerator's offset (voxel/__init__.py:103) */)
{
    const bool is_sum = reduction == D3D_REDUCE_MEAN || reduction == kReduceSum;
    if (coords) {                                  // (the sharded voxelizer's local pass only needs the cell keys)
        long long cc[3];
        kf.decode(((u64)vi.y << 32) | vi.x, cc);
        if (coord_sub) { cc[0] -= coord_sub[0]; cc[1] -= coord_sub[1]; cc[2] -= coord_sub[2]; }
        // 24 bytes as 16 + 8 (rows alternate their 16-byte alignment): two store requests instead of three
        long long *cp = reinterpret_cast<long long *>(coords) + v * 3;
        if ((reinterpret_cast<uintptr_t>(coords) & 15) == 0) {
            if ((v & 1) == 0) { *reinterpret_cast<longlong2 *>(cp) = make_longlong2(cc[0], cc[1]); cp[2] = cc[2]; }
            else { cp[0] = cc[0]; *reinterpret_cast<longlong2 *>(cp + 1) = make_longlong2(cc[1], cc[2]); }
        } else { cp[0] = cc[0]; cp[1] = cc[1]; cp[2] = cc[2]; }
    }
    npoints[v] = (int32_t)(vi.w < npoints_clamp ? vi.w : npoints_clamp);
    if (voff) voff[v] = vi.z;
    if (keys_out) keys_out[v] = (int64_t)(((u64)vi.y << 32) | vi.x);
    if (pmask) {                                   // P % 16 == 0, 16-byte aligned (host-checked)
        for (uint32_t k0 = 0; k0 < P; k0 += 16) {
            uint32_t w4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t b = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) b |= ((k0 + q * 4 + j) < vi.w ? 1u : 0u) << (8 * j);
                w4[q] = b;
            }
            *reinterpret_cast<uint4 *>(pmask + v * P + k0) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
    }
    if (AGG4) {
        const uint32_t cnt = vi.w, base = vi.z;
        if (cnt <= P) {
            float a0, a1, a2, a3;
            a0 = a1 = a2 = a3 = is_sum ? 0.0f : (reduction == D3D_REDUCE_MAX ? -INFINITY : INFINITY);
            // most voxels of a LiDAR frame hold ONE point: its row alone first (one gather), the rest 4 rows per step
            // (independent loads); the accumulation stays strictly in point order
            auto acc = [&](const float4 x) {
                if (is_sum) { a0 += x.x; a1 += x.y; a2 += x.z; a3 += x.w; }
                else if (reduction == D3D_REDUCE_MAX) {      // std::max(acc, x) = acc < x ? x : acc
                    a0 = a0 < x.x ? x.x : a0; a1 = a1 < x.y ? x.y : a1; a2 = a2 < x.z ? x.z : a2; a3 = a3 < x.w ? x.w : a3;
                } else {
                    a0 = x.x < a0 ? x.x : a0; a1 = x.y < a1 ? x.y : a1; a2 = x.z < a2 ? x.z : a2; a3 = x.w < a3 ? x.w : a3;
                }
            };
            if (cnt > 0) acc(staged[base]);
            for (uint32_t k = 1; k < cnt; k += 4) {
                const float4 *row = staged + base + k;         // contiguous rows, loads independent
                const float4 xs[4] = {row[0], k + 1 < cnt ? row[1] : row[0], k + 2 < cnt ? row[2] : row[0],
                                      k + 3 < cnt ? row[3] : row[0]};
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (k + j < cnt) acc(xs[j]);
            }
            if (reduction == D3D_REDUCE_MEAN) {              // voxelize.cpp:164 (float / int)
                const float d = (float)(int32_t)cnt;
                a0 = a0 / d; a1 = a1 / d; a2 = a2 / d; a3 = a3 / d;
            }
            agg[v] = make_float4(a0, a1, a2, a3);
        }
        // overflow voxels (every point counts, voxelize.cpp:137-157, but only P are ranked): hash path -> work list,
        // reduced one-wavefront-per-voxel by k_overflow_reduce; binned path -> k_bucket_index left the result in the
        // unused row P of the voxel's segment
        if (cnt > P) {
            if (big_list) big_list[atomicAdd(big_count, 1u)] = (uint32_t)v;
            else agg[v] = staged[base + P];
        }
    }
}

template <class Key, bool AGG4>
__global__ __launch_bounds__(256) void k_meta(Key kf, const float4 *__restrict__ points,
                                              const int64_t *__restrict__ counts, const uint4 *__restrict__ vinfo,
                                              const float4 *__restrict__ staged, const uint32_t *__restrict__ unsorted,
                                              uint32_t P, int reduction, int64_t *coords, int32_t *npoints,
                                              uint32_t *voff, unsigned char *pmask, float4 *agg, uint32_t *big_list,
                                              uint32_t *big_count, int64_t *keys_out = nullptr, int64_t status_row = -1)
{
    (void)points; (void)unsorted;
    const int64_t V = counts[D3D_COUNT_VOXELS];
    // sharded voxelizer: the status bits travel with the key list (row `status_row`, negative = not a cell)
    if (keys_out && status_row >= 0 && blockIdx.x == 0 && threadIdx.x == 0) keys_out[status_row] = -1 - counts[D3D_COUNT_STATUS];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += stride)
        meta_voxel<Key, AGG4>(kf, v, vinfo[v], staged, P, reduction, coords, npoints, voff, pmask, agg, big_list, big_count,
                              keys_out);
}

// One wavefront per overflow voxel: walk its arrival-ordered segment 64 entries per step.  MEAN accumulates in
// fp64 (insensitive to the arrival order to ~1e-16 => reproducible run to run; differs from the reference's fp32
// running sum by rounding only).
__global__ __launch_bounds__(256) void k_overflow_reduce(const float4 *__restrict__ points, const uint4 *__restrict__ vinfo,
                                                         const uint32_t *__restrict__ unsorted,
                                                         const uint32_t *__restrict__ big_list,
                                                         const uint32_t *__restrict__ big_count, int reduction, float4 *agg)
{
    const int lane = threadIdx.x & (kWave - 1);
    const bool is_sum = reduction == D3D_REDUCE_MEAN || reduction == kReduceSum;
    const uint32_t total = *big_count;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; t < total; t += nwaves) {
        const uint32_t v = big_list[t];
        const uint4 vi = vinfo[v];
        const uint32_t cc = vi.w;
        const uint32_t *seg = unsorted + vi.z;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        float e0, e1, e2, e3;
        e0 = e1 = e2 = e3 = reduction == D3D_REDUCE_MAX ? -INFINITY : INFINITY;
        for (uint32_t q = lane; q < cc; q += kWave) {
            const float4 x = points[seg[q]];
            if (is_sum) { s0 += x.x; s1 += x.y; s2 += x.z; s3 += x.w; }
            else if (reduction == D3D_REDUCE_MAX) {
                e0 = e0 < x.x ? x.x : e0; e1 = e1 < x.y ? x.y : e1; e2 = e2 < x.z ? x.z : e2; e3 = e3 < x.w ? x.w : e3;
            } else {
                e0 = x.x < e0 ? x.x : e0; e1 = x.y < e1 ? x.y : e1; e2 = x.z < e2 ? x.z : e2; e3 = x.w < e3 ? x.w : e3;
            }
        }
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            if (is_sum) {
                s0 += __shfl_xor(s0, o, kWave); s1 += __shfl_xor(s1, o, kWave);
                s2 += __shfl_xor(s2, o, kWave); s3 += __shfl_xor(s3, o, kWave);
            } else {
                const float t0 = __shfl_xor(e0, o, kWave), t1 = __shfl_xor(e1, o, kWave);
                const float t2 = __shfl_xor(e2, o, kWave), t3 = __shfl_xor(e3, o, kWave);
                if (reduction == D3D_REDUCE_MAX) {
                    e0 = e0 < t0 ? t0 : e0; e1 = e1 < t1 ? t1 : e1; e2 = e2 < t2 ? t2 : e2; e3 = e3 < t3 ? t3 : e3;
                } else {
                    e0 = t0 < e0 ? t0 : e0; e1 = t1 < e1 ? t1 : e1; e2 = t2 < e2 ? t2 : e2; e3 = t3 < e3 ? t3 : e3;
                }
            }
        }
        if (lane == 0) {
            if (is_sum) {
                const float d = reduction == D3D_REDUCE_MEAN ? (float)(int32_t)cc : 1.0f;
                agg[v] = make_float4((float)s0 / d, (float)s1 / d, (float)s2 / d, (float)s3 / d);
            } else agg[v] = make_float4(e0, e1, e2, e3);
        }
    }
}

// ------------------------------------------------------------------ binned index (dense contract, C == 4)
// The hash path above pays ~4 scattered HBM requests per point (probe, atomic, arrival list, staged row) at 25-80 G/s.
// Here the points are first PARTITIONED by hash(cell) into buckets of ~512 (tile histogram -> row scan -> scatter of the
// 16-byte rows: the only scattered global traffic per point), then one workgroup per bucket does all the per-point
// work -- voxel lookup, counting, first index, rank in point order -- in LDS and writes the rows next to their rank.
// Numbering by first occurrence stays a prefix count over point indices; it is merged with the per-voxel outputs
// (k_meta_first), whose stores are then coalesced by voxel id.
// Diagnostic build only (make PHASE_CLOCKS=1; tools/phase_clocks.py): wavefront 0 of every workgroup adds the cycles between
// its phase marks to g_phase[kernel][phase]; the product build compiles none of it.
#ifdef D3D_PHASE_CLOCKS
__device__ unsigned long long g_phase[4][16];
#define D3D_PHASE_DECL unsigned long long ph_t_ = __builtin_readcyclecounter()
#define D3D_PHASE(K, P)                                                                                  \
    do {                                                                                                 \
        const unsigned long long now_ = __builtin_readcyclecounter();                                    \
        if (threadIdx.x == 0) atomicAdd(&g_phase[K][P], now_ - ph_t_);                                   \
        ph_t_ = now_;                                                                                    \
    } while (0)
#else
#define D3D_PHASE_DECL do { } while (0)
#define D3D_PHASE(K, P) do { } while (0)
#endif

constexpr int kBinTile = 4096;                // points per pass of a workgroup of k_bin_count / k_bin_scatter
// passes per workgroup: a workgroup's tile is kBinTile * bin_passes(n) points.  Frames above 2 M points take 4 -- the tile x
// bucket matrix (written by k_bin_count, scanned by k_bin_scan, read by k_bin_scatter) shrinks by that factor: 64 -> 16 MB
// at 8 M points, where scanning it cost 89 us
static inline int bin_passes(int64_t n) { return n > (2 << 20) ? 4 : 1; }
constexpr int kBinThreads = 1024;             // ... 4 per lane: 16 wavefronts per CU keep the loads in flight
constexpr int kBinBits = 14;
constexpr int kBinMax = 1 << kBinBits;        // buckets (14 bits of the per-point word, 14 more for the rank in the tile): frames
                                              // of up to 16 M points (round 3; 13 bits = 8 M before)
constexpr int kBucketTarget = 1024;           // mean points per bucket the partition aims at (round 4: 512 before -- two points per lane
                                              // in flight hide the phases' latencies: config 2 134 -> 123 us per call, profiles/r04_bucket_target.txt)
constexpr int kBucketCap = 2048;              // points one k_bucket_index workgroup holds in registers
constexpr int kBucketThreads = 512;           // ... 4 per lane (256: 42 us, 512: 38 us, 1024: 44 us at config 2)
constexpr int kBucketSlots = 2048;            // LDS table slots (>= distinct cells of a bucket, always)
constexpr uint32_t kNoBin = 0xffffffffu;
thread_local int64_t g_last_plan[4] = {0, 0, 0, 0};       // d3d_voxelize_dense_last_plan
constexpr int64_t kFillSortMax16 = 76ll << 16;   // ZeroFill: 16-byte zeros under k_tile_sort (76 MB)
constexpr int kNumCUs = 256;                  // MI355X (the library is built for gfx950 only): the fillers of ZeroFill take the CUs a launch leaves idle
constexpr uint32_t kXcdBucketsMin = 2048;     // k_bucket_index: XCD-aware bucket numbering from this many buckets (a power of two) on
// Round 5 (k_bucket_index<.., V2> -> k_emit): the first-point entry of a voxel carries the voxel itself --
//   firstmap[first] = {count : 8 | base : 24}   count 1 .. 254 points, base = the voxel's segment in the ranked index lists (its
//                                                cell is rebuilt from the first point's row, which k_emit reads anyway: no record),
//                     {255 | record position}    255 points or more (or a big bucket): the 16-byte record as before,
//                     kInf                        not a first point.
// One dependent random gather less per multi-point voxel in k_emit, no record stores in the index, and the points kept
// (min(count, max_points)) are known to k_emit as soon as its first load returns.  Positions stay below 2^24 - 1 (host-checked).
// experiment knobs of the diagnostic build (make TUNE=1 -> libd3d_hip_tune.so; tools/tune_ab.py): the product build compiles the defaults in
#ifdef D3D_TUNE
int g_d3d_tune[16] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
#define D3D_TUNE_VAL(K, DEF) (g_d3d_tune[K] >= 0 ? g_d3d_tune[K] : (DEF))
#else
#define D3D_TUNE_VAL(K, DEF) (DEF)
#endif
constexpr uint32_t kFmShift = 24, kFmMask = 0xffffffu, kFmRecord = 255u;
constexpr int64_t kFmMaxPoints = (1 << 24) - 2;
constexpr uint32_t kDenseMin = 32;            // V2: cells with more points are ranked by ONE wavefront (select + all-pairs on the kept ones);
                                              // 16 .. 48: the same within noise, 64: +2.5 us, 128: +4 us at config 2 (profiles/r05_b_tune.txt)

// Round 6: part of the zero padding of voxels[V,P,4] leaves UNDER the index launches.  Workgroups `first` ..
// `first + nblk - 1` of a launch's grid take no tile, they stream 16-byte zeros over [0, n16) of `dst`, one 16 KiB piece per
// workgroup per step, the pieces of one step side by side (the grid-stride pattern of the store probes).  Only where CUs would
// otherwise stand idle for the whole launch: k_tile_sort on tiles of 8192 points (124 workgroups at config 2, one per CU: 132
// fillers write 76 MB for +2 us) and k_first_count (62 workgroups).  Everything else that was tried costs the index what it saves
// k_emit_split, or more -- zeros stored by the tiles' / buckets' own workgroups behind their first loads, filler workgroups
// behind the buckets (the CUs are full: they only run in the launch's tail): profiles/r06_ab_prefill.txt.  The range is fixed by
// the host before the first launch, from the frame's size alone (no state from earlier calls): rows past the final V are never
// returned (the tensor has min(n, max_voxels) rows), so an overshoot only wastes stores.
struct ZeroFill {
    float4 *dst = nullptr;
    int64_t n16 = 0;
    uint32_t first = 0xffffffffu, nblk = 0;
};
__device__ __forceinline__ bool zero_fill_role(const ZeroFill &z)
{
    typedef float vec4 __attribute__((ext_vector_type(4)));
    if (blockIdx.x < z.first) return false;
    vec4 *out = reinterpret_cast<vec4 *>(z.dst);
    const vec4 zero = {0.f, 0.f, 0.f, 0.f};
    const int64_t step = (int64_t)z.nblk * blockDim.x;
    for (int64_t i = (int64_t)(blockIdx.x - z.first) * blockDim.x + threadIdx.x; i < z.n16; i += step)
        __builtin_nontemporal_store(zero, &out[i]);
    return true;
}

// reduce contract (d3d_voxelize_3d_reduce) on the binned path
struct BinnedExtras {
    int64_t *first_out;       // [V] index_offset + first point index
    int64_t index_offset;
    int64_t *keys_out;        // [V] cell key; row status_row = -1 - status bits
    int64_t status_row;
    uint32_t *vidof;          // [npad] voxel id of the voxel whose first point is i, kNoVoxel elsewhere (for the point -> voxel map)
    int64_t *host_counts;     // optional host-mapped copy of counts[] + ready flag (d3d_voxelize_3d_dense_notify)
    uint32_t npoints_clamp = 0xffffffffu;
    uint32_t *voff = nullptr; // [V] segment base (dense contract, C != 4: k_aggregate reads the index lists through it)
    int32_t *count_out = nullptr;    // fused sparse + filter, DESCENDING: [V] the voxel's point count, unclamped (the sort key)
    bool has_coord_sub = false;      // fused sparse + filter: coords - offset (VoxelGenerator.__call__, voxel/__init__.py:103)
    long long coord_sub[3] = {0, 0, 0};
    int64_t aux_value = 0;           // what k_emit leaves in counts[D3D_COUNT_AUX] (d3d_voxelize_3d_reduce: 1 = ranked index lists)
    uint16_t *row_state = nullptr;   // d3d_voxelize_3d_dense_resident: [capacity] rows of voxels[v] that may be non-zero (k_emit<.., true>)
    bool fm_packed = false;          // firstmap entries are {count : 8 | segment : 24} words (k_bucket_index<.., V2>), see kFmShift
    bool early_zero = false;         // k_emit: the all-zero lines of the stretch are stored as soon as the entries are known
};

// sparse contract fused with the voxel filter (d3d_voxelize_3d_sparse_filter): only voxels that pass get a first-point
// entry, so the first-seen numbering directly yields the filtered voxel ids (voxelize.cpp:374-403 in id order)
struct VoxelPass {
    bool on;
    int32_t min_points;
    long long lo[3], hi[3];
};

// counts[] are final: publish them to host-mapped pinned memory, flag last (one lane)
__device__ __forceinline__ void notify_host(const int64_t *counts, int64_t *host)
{
    for (int k = 0; k < D3D_NUM_COUNTS; k++) host[k] = counts[k];
    __threadfence_system();
    __hip_atomic_store(&host[D3D_NUM_COUNTS], (int64_t)1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_notify_host(const int64_t *counts, int64_t *host)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) notify_host(counts, host);
}

// per tile: bucket histogram in LDS; every point remembers its cell key and {bucket, arrival number inside the tile}.
// ROWS = dense contract on C == 4 rows, otherwise sparse contract (any C).
// What travels to the buckets: dense contract (C == 4 rows) {cell : 32 | point index : 32} in one 8-byte store -- the row
// itself is gathered by k_bucket_index, late, behind its LDS phases; sparse contract {63-bit cell, point index} in one
// 16-byte store.  One scattered request per point either way.
template <bool ROWS> struct BinEntry;
template <> struct BinEntry<true> {
    typedef u64 type;
    typedef uint32_t key_store_t;                   // per-point key between k_bin_count and k_bin_scatter
    static __device__ __forceinline__ type pack(u64 key, uint32_t idx) { return key | ((u64)idx << 32); }
    static __device__ __forceinline__ u64 key(type e) { return e & 0xffffffffull; }
    static __device__ __forceinline__ uint32_t idx(type e) { return (uint32_t)(e >> 32); }
    static constexpr int kIdxStride = 2, kIdxOff = 1;   // where the index sits, in 32-bit words
};
template <> struct BinEntry<false> {
    typedef uint4 type;
    typedef u64 key_store_t;
    static __device__ __forceinline__ type pack(u64 key, uint32_t idx) { return make_uint4((uint32_t)key, (uint32_t)(key >> 32), idx, 0u); }
    static __device__ __forceinline__ u64 key(type e) { return (u64)e.x | ((u64)e.y << 32); }
    static __device__ __forceinline__ uint32_t idx(type e) { return e.z; }
    static constexpr int kIdxStride = 4, kIdxOff = 2;
};

constexpr uint32_t kBadBin = 0xfffffffeu;     // the point's coordinates overflowed: status raised by k_bin_scatter (this
                                              // kernel resets the counters while it runs)
template <class Key, bool VEC4, bool ROWS>
__global__ __launch_bounds__(kBinThreads) void k_bin_count(Key kf, const float *__restrict__ points, int64_t n, int c, uint32_t nbins,
                                                           uint32_t passes, uint32_t *__restrict__ pbin,
                                                           typename BinEntry<ROWS>::key_store_t *__restrict__ pkey,
                                                           uint32_t *__restrict__ tilecnt, uint32_t *__restrict__ firstmap,
                                                           int64_t *counts, int64_t *mapping, unsigned char *trimmed,
                                                           int32_t *keepid, u64 *zero_words /* look-back words of a later launch */,
                                                           uint32_t nzero, unsigned int *zero_ticket)
{
    extern __shared__ uint32_t h[];                       // [nbins]: 2 KB (1 M points) .. 64 KB (16 M points), sized by the launch
    for (uint32_t b = threadIdx.x; b < nbins; b += kBinThreads) h[b] = 0;
    if (blockIdx.x == 0 && threadIdx.x < D3D_NUM_COUNTS) counts[threadIdx.x] = 0;
    if (zero_words && blockIdx.x == 0) {
        for (uint32_t t = threadIdx.x; t < nzero; t += kBinThreads) zero_words[t] = 0ull;
        if (threadIdx.x == 0) *zero_ticket = 0u;
    }
    __syncthreads();
    for (uint32_t pass = 0; pass < passes; pass++) {
    const int64_t base = ((int64_t)blockIdx.x * passes + pass) * kBinTile + threadIdx.x;
    float v[kBinTile / kBinThreads][3];
#pragma unroll
    for (int r = 0; r < kBinTile / kBinThreads; r++) {
        const int64_t i = base + r * kBinThreads;
        if (i < n) {
            if (VEC4) {
                const float4 q = reinterpret_cast<const float4 *>(points)[i];
                v[r][0] = q.x; v[r][1] = q.y; v[r][2] = q.z;
            } else {
                const float *src = points + i * c;
                v[r][0] = src[0]; v[r][1] = src[1]; v[r][2] = src[2];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kBinTile / kBinThreads; r++) {
        const int64_t i = base + r * kBinThreads;
        uint32_t word = kNoBin, status = 0;
        if (i < n) {
            u64 key;
            if (kf.make(v[r], key, status)) {
                const uint32_t b = Key::bin_hash(key) & (nbins - 1);
                word = b | (atomicAdd(&h[b], 1u) << kBinBits);
                pkey[i] = (typename BinEntry<ROWS>::key_store_t)key;
            } else if (status) word = kBadBin;
        }
        pbin[i] = word;                 // arrays are padded to the tile
        firstmap[i] = kInf;
        if (mapping && i < n) mapping[i] = -1;      // points outside the grid keep it
        if (trimmed && i < n) trimmed[i] = 0;
        if (keepid && i < n) keepid[i] = -1;
    }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbins; b += kBinThreads) tilecnt[(size_t)blockIdx.x * nbins + b] = h[b];     // [tile][bucket]
}

// tilecnt[tile][bucket] -> exclusive prefix over the tiles, per bucket; bucket totals.  A workgroup owns 64 consecutive
// buckets (one per lane, coalesced rows), its 16 wavefronts split the tiles.
__global__ __launch_bounds__(1024) void k_bin_scan(uint32_t *tilecnt, uint32_t nbins, uint32_t ntiles, uint32_t *totals)
{
    __shared__ uint32_t wsum[16][kWave];
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * kWave + lane;
    const uint32_t per = (ntiles + 15) / 16, t0 = w * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
    uint32_t sum = 0;
    if (b < nbins)
        for (uint32_t t = t0; t < t1; t++) sum += tilecnt[(size_t)t * nbins + b];
    wsum[w][lane] = sum;
    __syncthreads();
    uint32_t run = 0, all = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t x = wsum[k][lane];
        if (k < w) run += x;
        all += x;
    }
    if (b >= nbins) return;
    if (w == 0) totals[b] = all;
    for (uint32_t t = t0; t < t1; t++) {
        const uint32_t x = tilecnt[(size_t)t * nbins + b];
        tilecnt[(size_t)t * nbins + b] = run;
        run += x;
    }
}

// exclusive scan of the bucket totals into LDS (every workgroup of k_bin_scatter repeats it: 4096 values at most)
__device__ __forceinline__ void bucket_bases(const uint32_t *__restrict__ totals, uint32_t nbins, uint32_t *base /* LDS [nbins] */,
                                             u64 *smem)
{
    constexpr int PER = kBinMax / kBinThreads;
    const uint32_t b0 = threadIdx.x * PER;
    uint32_t tot[PER];
    u64 sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        tot[k] = b0 + k < nbins ? totals[b0 + k] : 0u;
        sum += tot[k];
    }
    u64 all;
    u64 ex = block_excl_scan_u64<kBinThreads>(sum, &all, smem);
#pragma unroll
    for (int k = 0; k < PER; k++) {
        if (b0 + k < nbins) base[b0 + k] = (uint32_t)ex;
        ex += tot[k];
    }
    __syncthreads();
}

// {cell key, point index} to the bucket: position = base of the bucket + offset of the tile + arrival in the tile
template <bool ROWS>
__global__ __launch_bounds__(kBinThreads) void k_bin_scatter(const typename BinEntry<ROWS>::key_store_t *__restrict__ pkey, int64_t n,
                                                             uint32_t nbins, uint32_t *pbin,
                                                             const uint32_t *__restrict__ tileoff, const uint32_t *__restrict__ totals,
                                                             uint32_t *__restrict__ bucket_base,
                                                             typename BinEntry<ROWS>::type *__restrict__ bent, int64_t *counts,
                                                             bool keep_pos /* pbin[i] := kInf for the points in no bucket (it becomes pfirst) */,
                                                             uint32_t passes)
{
    extern __shared__ uint32_t off[];                     // [nbins], sized by the launch
    __shared__ u64 smem[kBinThreads / kWave];
    bucket_bases(totals, nbins, off, smem);
    if (blockIdx.x == 0)                                     // for k_bucket_index
        for (uint32_t b = threadIdx.x; b <= nbins; b += kBinThreads) bucket_base[b] = b < nbins ? off[b] : off[nbins - 1] + totals[nbins - 1];
    for (uint32_t b = threadIdx.x; b < nbins; b += kBinThreads) off[b] += tileoff[(size_t)blockIdx.x * nbins + b];
    __syncthreads();
    bool bad = false;
    for (uint32_t pass = 0; pass < passes; pass++) {
    const int64_t base = ((int64_t)blockIdx.x * passes + pass) * kBinTile + threadIdx.x;
#pragma unroll
    for (int r = 0; r < kBinTile / kBinThreads; r++) {
        const int64_t i = base + r * kBinThreads;
        if (i >= n) break;
        const uint32_t word = pbin[i];
        if (word == kBadBin) bad = true;
        if (word >= kBadBin) {
            if (keep_pos) pbin[i] = kInf;
            continue;
        }
        const uint32_t pos = off[word & (kBinMax - 1)] + (word >> kBinBits);
        bent[pos] = BinEntry<ROWS>::pack((u64)pkey[i], (uint32_t)i);       // (keep_pos: k_bucket_index overwrites pbin[i])
    }
    }
    if (bad) atomicOr(reinterpret_cast<u64 *>(&counts[D3D_COUNT_STATUS]), (u64)D3D_VOXEL_STATUS_COORD_OVERFLOW);
}

// ------------------------------------------------------------------ one-launch partition (round 4)
// k_bin_count + k_bin_scan + k_bin_scatter are three launch + drain floors (36-38 us at 1 M points for 24 MB of compulsory
// traffic) and 1 M scattered 8-byte stores.  Here a workgroup SORTS its tile of 8192 points by bucket in LDS (histogram with
// arrival numbers, scan over the buckets, entries placed, all on chip) and writes the tile back as ONE coalesced run in bucket
// order, plus one word per (bucket, tile) {offset of the bucket inside the tile : 16 | entries : 16} into a bucket-major
// table.  No tile waits for another one: the bucket workgroup of k_bucket_index reads its row of the table (coalesced),
// whose offsets add up to the number of entries in lower buckets -- its base in every per-bucket array, no scan over the
// buckets anywhere -- and gathers its entries as <= ntiles short runs (tile sort: mean run = 8192 / buckets = 4 entries at
// config 2, i.e. a quarter of the requests of one scattered store per point; frames above 4 M points keep the three-pass
// partition, where runs would shrink to single entries and the table would outgrow the entries).
constexpr int kSortThreads = 1024;
// the one-launch partition up to here; beyond, a bucket's run in a tile shrinks to a single entry and k_bucket_index gathers
// them one by one (profiles/r04_tile_sort_large.txt: dense 4 M / 8 M points 4 % faster than the three passes, sparse + trim
// 4 M 11 % faster, 8 M 7 % slower -- its 16-byte entries)
constexpr int64_t kTileSortMaxPoints = 8ll << 20, kTileSortMaxPointsSparse = 4ll << 20;
constexpr int64_t kBigTileMinPoints = 6 << 20;              // k_tile_sort: tiles of 16384 points from here on (dense contract; A/B: 8 M points -15 us, 4 M +3)
constexpr int kRunCap = 1024;                              // tiles: k_bucket_index keeps the run table in LDS

// kSortItems points per lane: tiles of 1024 * kSortItems points (offsets and run lengths fit 16 bits)
template <class Key, bool VEC4, bool ROWS, int kSortItems>
__global__ __launch_bounds__(kSortThreads) void k_tile_sort(Key kf, const float *__restrict__ points, int64_t n, int c, uint32_t nbins,
                                                            uint32_t ntiles, typename BinEntry<ROWS>::type *__restrict__ tsort,
                                                            uint32_t *__restrict__ table /* [nbins][ntiles] */,
                                                            uint32_t *__restrict__ tileinfo /* [ntiles] entries | bad << 31 */,
                                                            uint32_t *__restrict__ pfirst /* optional: [n] kInf for the points in no bucket
                                                                                             (k_bucket_index fills in the others) */,
                                                            uint32_t *__restrict__ firstmap, int64_t *counts, int64_t *mapping,
                                                            unsigned char *trimmed, int32_t *keepid,
                                                            u64 *zero_words /* look-back words of a later launch */, uint32_t nzero,
                                                            unsigned int *zero_ticket, bool pfirst_self = true /* false: pfirst <- kInf */,
                                                            ZeroFill zf = ZeroFill())
{
    typedef typename Key::bin_key_t KT;
    typedef BinEntry<ROWS> E;
    constexpr int kSortTile = kSortThreads * kSortItems;
    if (zero_fill_role(zf)) return;
    constexpr int kSortTileShift = kSortItems == 16 ? 14 : kSortItems == 8 ? 13 : kSortItems == 4 ? 12 : 11;
    static_assert(kSortTile == (1 << kSortTileShift) && kFlagTile % kSortTile == 0, "arrays are padded to kFlagTile");
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_lds[];
    KT *keys = reinterpret_cast<KT *>(tile_lds);                           // [kSortTile] in bucket order
    uint32_t *h = reinterpret_cast<uint32_t *>(keys + kSortTile);           // [nbins] histogram, then the buckets' offsets
    uint16_t *lidx = reinterpret_cast<uint16_t *>(h + nbins);               // [kSortTile] point index inside the tile
    __shared__ u64 smem[kSortThreads / kWave];
    __shared__ uint32_t sbad;
    // XCD-aware tile numbering: workgroup ids go round the eight XCDs, so the workgroups of ONE XCD take a contiguous range of
    // tiles -- a bucket's row of the table (one 4-byte word per tile) is then written as adjacent words into the same L2
    // instead of one word per line from eight different ones (the 8 M-point frame wrote 400 MB for a 33 MB table).  Tiles
    // are independent (no look-back between them), so any numbering is valid.
    // (A/B on one box: k_tile_sort 118 -> 96 us at 8 M points, 44.5 -> 42.5 at 4 M; at 1 M -- 123 tiles, a 0.5 MB table -- nothing
    // to gain, so small frames keep the identity.)
    const uint32_t per_xcd = (ntiles + 7u) >> 3;
    const uint32_t tile = ntiles >= 256u ? (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    if (tile >= ntiles) return;                             // (the grid is rounded up to a multiple of 8)
    const int64_t base = (int64_t)tile * kSortTile + threadIdx.x;
    float v[kSortItems][3];
#pragma unroll
    for (int r = 0; r < kSortItems; r++) {                  // (the loads fly while the histogram is cleared)
        const int64_t i = base + r * kSortThreads;
        if (i < n) {
            if (VEC4) {
                const float4 q = reinterpret_cast<const float4 *>(points)[i];
                v[r][0] = q.x; v[r][1] = q.y; v[r][2] = q.z;
            } else {
                const float *src = points + i * c;
                v[r][0] = src[0]; v[r][1] = src[1]; v[r][2] = src[2];
            }
        }
    }
    D3D_PHASE_DECL;
    for (uint32_t b = threadIdx.x; b < nbins; b += kSortThreads) h[b] = 0;
    if (threadIdx.x == 0) sbad = 0;
    if (tile == 0 && threadIdx.x < D3D_NUM_COUNTS) counts[threadIdx.x] = 0;
    if (zero_words && tile == 0) {
        for (uint32_t t = threadIdx.x; t < nzero; t += kSortThreads) zero_words[t] = 0ull;
        if (threadIdx.x == 0) *zero_ticket = 0u;
    }
    __syncthreads();
    D3D_PHASE(1, 0);                                        // points arrived, histogram cleared
    KT key[kSortItems];
    uint32_t word[kSortItems];
    bool bad = false;
#pragma unroll
    for (int r = 0; r < kSortItems; r++) {
        const int64_t i = base + r * kSortThreads;
        word[r] = kNoBin;
        key[r] = 0;
        if (i < n) {
            u64 k64;
            uint32_t status = 0;
            if (kf.make(v[r], k64, status)) {
                const uint32_t b = Key::bin_hash(k64) & (nbins - 1);
                word[r] = b | (atomicAdd(&h[b], 1u) << kBinBits);
                key[r] = (KT)k64;
            } else if (status) bad = true;
        }
        firstmap[i] = kInf;                         // arrays are padded to the tile
        if (mapping && i < n) mapping[i] = -1;      // points outside the grid keep it
        if (trimmed && i < n) trimmed[i] = 0;
        if (keepid && i < n) keepid[i] = -1;
    }
    if (bad) sbad = 1;
    __syncthreads();
    D3D_PHASE(1, 1);                                        // keys, histogram
    // buckets' offsets inside the tile: exclusive scan of the histogram (consecutive buckets per thread)
    constexpr int kPerMax = 8;                                  // nbins <= 8192 on this path (host-checked)
    const uint32_t per = nbins > (uint32_t)kSortThreads ? nbins / kSortThreads : 1u, b0 = threadIdx.x * per;
    uint32_t cnt[kPerMax];
    u64 mine = 0;
#pragma unroll
    for (int k = 0; k < kPerMax; k++) {
        cnt[k] = ((uint32_t)k < per && b0 + k < nbins) ? h[b0 + k] : 0u;
        mine += cnt[k];
    }
    u64 all;
    u64 ex = block_excl_scan_u64<kSortThreads>(mine, &all, smem);
#pragma unroll
    for (int k = 0; k < kPerMax; k++) {
        if ((uint32_t)k < per && b0 + k < nbins) {
            h[b0 + k] = (uint32_t)ex;
            table[(size_t)(b0 + k) * ntiles + tile] = (uint32_t)ex | (cnt[k] << 16);
            ex += cnt[k];
        }
    }
    if (threadIdx.x == 0) tileinfo[tile] = (uint32_t)all | (sbad << 31);
    __syncthreads();
    D3D_PHASE(1, 2);                                        // scan, table row stored
    const uint32_t tbase = tile << kSortTileShift;
#pragma unroll
    for (int r = 0; r < kSortItems; r++) {
        const int64_t i = base + r * kSortThreads;
        if (word[r] != kNoBin) {
            const uint32_t p = h[word[r] & (kBinMax - 1)] + (word[r] >> kBinBits);
            keys[p] = key[r];
            lidx[p] = (uint16_t)(r * kSortThreads + threadIdx.x);
            // (preset "the first point of my voxel is myself": true for the 80 % of a LiDAR frame's points that are alone in their
            // voxel -- k_bucket_index then only writes the others, one scattered store per point less for most of them)
            if (pfirst) pfirst[i] = pfirst_self ? (uint32_t)i : kInf;
        } else if (pfirst && i < n) pfirst[i] = kInf;
    }
    __syncthreads();
    D3D_PHASE(1, 3);                                        // entries placed in LDS
    const uint32_t total = (uint32_t)all;
    for (uint32_t p = threadIdx.x; p < total; p += kSortThreads) tsort[tbase + p] = E::pack((u64)keys[p], tbase + lidx[p]);
    D3D_PHASE(1, 4);                                        // run written (stores issued)
}

// One workgroup per bucket, everything per point in LDS: cell -> slot (open addressing), count, first index, segment of
// the indices, rank = number of smaller indices in the segment (early exit at max_points).  Outputs: the rows next to
// their rank (staged), one record per voxel {cell, first, segment base, count} and firstmap[first] = record position.
// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>).  Per-item register arrays indexed
// through it are split into scalars up front; with `#pragma unroll` the compiler left k_bucket_index's rows in scratch.
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global load
// (s_waitcnt vmcnt(0)), which would expose the latency of k_bucket_index's row gather at the first barrier.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void wave_lds_fence()  // LDS traffic of THIS wavefront is ordered (the pipe is in-order per wave);
{                                                 // keep the compiler from moving accesses across, and let writes land
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int BLOCK>
__device__ __forceinline__ u64 block_excl_scan_u64_lds(u64 v, u64 *total, u64 *smem)
{
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const u64 incl = wave_incl_scan_u64(v);
    if (lane == kWave - 1) smem[w] = incl;
    lds_barrier();
    u64 woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / kWave; k++) {
        const u64 x = smem[k];
        if (k < w) woff += x;
        tot += x;
    }
    lds_barrier();
    *total = tot;
    return woff + incl - v;
}

// two scans behind ONE pair of barriers (smem: 2 x BLOCK / 64 words)
template <int BLOCK>
__device__ __forceinline__ void block_excl_scan_2u64_lds(u64 a, u64 b, u64 *ex_a, u64 *ex_b, u64 *tot_a, u64 *tot_b, u64 *smem)
{
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const u64 ia = wave_incl_scan_u64(a), ib = wave_incl_scan_u64(b);
    if (lane == kWave - 1) { smem[w] = ia; smem[BLOCK / kWave + w] = ib; }
    lds_barrier();
    u64 wa = 0, wb = 0, ta = 0, tb = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / kWave; k++) {
        const u64 x = smem[k], y = smem[BLOCK / kWave + k];
        if (k < w) { wa += x; wb += y; }
        ta += x; tb += y;
    }
    lds_barrier();
    *tot_a = ta; *tot_b = tb;
    *ex_a = wa + ia - a;
    *ex_b = wb + ib - b;
}

// STAGE (dense contract on C == 4 rows): the ranked rows themselves are gathered and staged next to their rank, for the
// two-launch output stage.  ROWS && LISTS && !STAGE is the index for k_emit: no row moves here at all -- a voxel's first
// row is points[first point], which k_emit reads coalesced, and ranks 1 .. P-1 leave their point INDEX in sorted_out.
template <class Key, bool ROWS, bool LISTS, bool STAGE = ROWS, bool V2 = false>
__global__ __launch_bounds__(kBucketThreads) void k_bucket_index(Key kf, VoxelPass vp,
                                                      const typename BinEntry<ROWS>::type *__restrict__ bent,
                                                      const float4 *__restrict__ points4 /* ROWS */,
                                                      const uint32_t *__restrict__ bucket_base,
                                                      int hshift, uint32_t P, int reduction /* NONE: no aggregates */,
                                                      float4 *__restrict__ staged, uint4 *__restrict__ vrec,
                                                      uint32_t *__restrict__ firstmap, int64_t *counts,
                                                      uint32_t *__restrict__ precpos /* optional: per bucket entry, the FIRST point of its voxel */,
                                                      uint32_t *__restrict__ pinfo, uint32_t *__restrict__ gseg /* big buckets */,
                                                      unsigned char *__restrict__ trimmed /* optional: [n] rank >= P */,
                                                      uint32_t *__restrict__ sorted_out /* optional: ranked indices (C != 4) */,
                                                      uint32_t *__restrict__ unsorted_out /* ... and all of an overflow voxel's */,
                                                      // tile-sorted input (k_tile_sort): row blockIdx.x of the bucket-major table
                                                      // holds this bucket's run in every tile; NULL = `bent` is partitioned
                                                      const uint32_t *__restrict__ table, uint32_t ntiles, int tshift /* log2 tile */,
                                                      const uint32_t *__restrict__ tileinfo, uint32_t *__restrict__ gpos /* big buckets */,
                                                      // fused sparse + filter: by POINT index, the first point of the point's voxel if
                                                      // the point is kept (its voxel passes the filter, its rank is below P when
                                                      // P > 0), else kInf -- all the compaction needs (one scattered store per point
                                                      // here instead of two dependent random reads per point there)
                                                      uint32_t *__restrict__ pfirst_out,
                                                      // ... and the output sizes ahead of the numbering: 64 pairs {voxels that pass
                                                      // the filter, points they keep (count clamped to early_clamp)}, added up
                                                      // by the wavefronts (k_meta_first_lb's first tile tells the host)
                                                      u64 *__restrict__ early_tot = nullptr, uint32_t early_clamp = 0,
                                                      uint32_t early_mask = 0 /* pairs - 1; one pair per 128-byte line */,
                                                      int idx_bits = 24 /* V2: point indices < 2^idx_bits */,
                                                      uint32_t dense_min = kDenseMin)
{
    static_assert(!V2 || (ROWS && LISTS && !STAGE), "V2: the index for k_emit (dense contract on C == 4 rows) / k_sparse_finish (8-byte entries)");
    constexpr int ITEMS = kBucketCap / kBucketThreads, T = kBucketSlots;
    constexpr bool SPV2 = V2 && std::is_same<Key, BoundKey>::value;      // the fused sparse + filter call on packed entries (see V2 below)
    typedef typename Key::bin_key_t KT;
    typedef BinEntry<ROWS> E;
    constexpr KT kFree = (KT)~(KT)0;
    __shared__ KT tkey[T];
    // LDS budget: 4 workgroups per CU need <= 40 KiB each (and <= 64 VGPRs).  A register bucket has at most 2048 points, so a
    // slot's count and segment base share one word after the records phase (tcnt = base << 16 | count); a big bucket keeps
    // the bases in seg[] (its index segments live in global memory).
    static_assert(kBucketCap <= 0xffff && kBucketCap == T, "packed count | base; tbase aliases seg");
    __shared__ uint32_t tcnt[T], tfirst[T];
    __shared__ uint32_t seg[kBucketCap];
    __shared__ u64 smem[2 * kBucketThreads / kWave];
    __shared__ uint16_t oslot[T];                   // overflow voxels of the bucket (one per slot at most: a big bucket can hold
                                                    // T distinct cells with more than P points each); before the records
                                                    // phase: the tile of every entry of a register bucket (tile-sorted input)
    uint16_t *const tileof = oslot;
    __shared__ uint32_t nover, fail;
    __shared__ uint32_t whist[V2 ? kBucketThreads / kWave : 1][kWave];      // V2: one 64-bin histogram per wavefront (dense cells)
    typedef float v4f __attribute__((ext_vector_type(4)));   // (an array of HIP float4 structs stayed in scratch)
    D3D_PHASE_DECL;
    uint32_t bb, m;
    if (table) {
        // this bucket's runs: {offset in the tile, entries} per tile.  The offsets add up to the entries of all lower
        // buckets = the bucket's base in the per-bucket arrays; the run starts (scan of the lengths) and the runs' places in
        // `bent` wait in seg[] (free until the segments phase) for the lanes to look their entries up
        static_assert(2 * kRunCap <= kBucketCap && kRunCap <= 2 * kBucketThreads, "run table lives in seg[]");
        // XCD-aware bucket numbering (large frames): inside a tile the runs of buckets b, b + 1, ... lie side by side -- at 8 M
        // points a run is one 8-byte entry, sixteen buckets to a 128-byte line -- and workgroup ids go round the eight XCDs: with
        // bucket = id every line was fetched by all eight L2s.  Buckets are independent (their bases come from the table), so
        // the workgroups of one XCD take a contiguous range of them.
        const uint32_t bucket = gridDim.x >= kXcdBucketsMin ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
        const uint32_t *row = table + (size_t)bucket * ntiles;
        const uint32_t per = ntiles > (uint32_t)kBucketThreads ? 2u : 1u, t0 = threadIdx.x * per;
        uint32_t wv[2];
#pragma unroll
        for (int k = 0; k < 2; k++) wv[k] = ((uint32_t)k < per && t0 + k < ntiles) ? row[t0 + k] : 0u;
        // (the tables are cleared while the row is on its way)
        for (int s = threadIdx.x; s < T; s += kBucketThreads) { tkey[s] = kFree; tcnt[s] = 0; tfirst[s] = kInf; }
        if (threadIdx.x == 0) { nover = 0; fail = 0; }
        u64 mine = 0;
#pragma unroll
        for (int k = 0; k < 2; k++) mine += (u64)(wv[k] >> 16) | ((u64)(wv[k] & 0xffffu) << 32);
        if (blockIdx.x == 0) {                      // a tile met a coordinate beyond the key range (k_tile_sort reset the word)
            bool bad = false;
            for (uint32_t t = threadIdx.x; t < ntiles; t += kBucketThreads) bad = bad || (tileinfo[t] >> 31);
            if (bad) atomicOr(reinterpret_cast<u64 *>(&counts[D3D_COUNT_STATUS]), (u64)D3D_VOXEL_STATUS_COORD_OVERFLOW);
        }
        u64 all;
        const u64 ex = block_excl_scan_u64_lds<kBucketThreads>(mine, &all, smem);
        m = (uint32_t)all;
        bb = (uint32_t)(all >> 32);
        uint32_t run = (uint32_t)ex;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if ((uint32_t)k < per && t0 + k < ntiles) {
                seg[t0 + k] = run;
                seg[kRunCap + t0 + k] = ((t0 + k) << tshift) + (wv[k] & 0xffffu);
                // a register bucket: the run's lane names the tile of each of its entries (a handful of stores; the lanes
                // then find their entry with three reads instead of a binary search over the run starts)
                if (m <= (uint32_t)kBucketCap)
                    for (uint32_t j = 0; j < (wv[k] >> 16); j++) tileof[run + j] = (uint16_t)(t0 + k);
                run += wv[k] >> 16;
            }
        }
    } else {
        bb = bucket_base[blockIdx.x];
        m = bucket_base[blockIdx.x + 1] - bb;
    }
    D3D_PHASE(0, 0);                                // row / bases loaded, run table
    if (m == 0) return;
    if (!table) {
        for (int s = threadIdx.x; s < T; s += kBucketThreads) { tkey[s] = kFree; tcnt[s] = 0; tfirst[s] = kInf; }
        if (threadIdx.x == 0) { nover = 0; fail = 0; }
    }
    // entry q of the bucket -> its place in `bent`
    // (the last tile whose run starts at or before q -- empty runs share a start; a fixed number of halvings, no branches)
    const int lsteps = table ? 32 - __builtin_clz(ntiles > 1 ? ntiles - 1 : 1u) : 0;
    auto locate = [&](uint32_t q) -> uint32_t {
        if (!table) return bb + q;
        uint32_t lo = 0, hi = ntiles;
        for (int it = 0; it < lsteps; it++) {
            const uint32_t mid = (lo + hi) >> 1;
            const bool le = hi - lo > 1 && seg[mid] <= q;
            lo = le ? mid : lo;
            hi = (le || hi - lo <= 1) ? hi : mid;
        }
        return seg[kRunCap + lo] + (q - seg[lo]);
    };

    uint32_t early_v = 0, early_p = 0;
    auto early_publish = [&]() {                            // (workgroup-uniform call sites: all lanes take part in the sum)
        if (!early_tot) return;
        const u64 both = wave_sum_u64(((u64)early_v << 32) | early_p);
        if ((threadIdx.x & (kWave - 1)) == 0 && both) {
            // (a pair per 128-byte line: atomics on one line serialise in its L2 channel -- 64 pairs packed into 16 lines made
            // this kernel 8 us slower at config 2)
            u64 *dst = early_tot + 16 * (((blockIdx.x << 3) + (threadIdx.x >> 6)) & early_mask);
            (void)__hip_atomic_fetch_add(&dst[0], both >> 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_add(&dst[1], both & 0xffffffffull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // phase B of both modes: segments in slot order, one record per voxel, firstmap
    auto records = [&](auto BIG) {
        constexpr int PER = T / kBucketThreads;
        // (the index for k_emit and the sparse contract's keep no record for a voxel of ONE point in a register bucket -- 80 % of a
        // LiDAR frame's voxels: k_emit / k_meta_first rebuild it from the point itself, firstmap says so)
        constexpr bool kSkipSingles = ((ROWS && LISTS && !STAGE) || std::is_same<Key, SparseKey>::value) && !decltype(BIG)::value;
        // one slot of the table -> its voxel's record, first-point entry, overflow listing
        auto slot_out = [&](uint32_t sl, uint32_t cnt, uint32_t base, uint32_t &j) {
            const uint32_t f = tfirst[sl];
            const u64 kk = (u64)tkey[sl];
            const bool single = kSkipSingles && cnt == 1;
            if (!single) vrec[bb + j] = make_uint4((uint32_t)kk, (uint32_t)(kk >> 32), bb + base, cnt);
            bool pass = true;
            if (vp.on) {                        // voxelize.cpp:376-384: coordinate bounds and min_points
                long long cc[3];
                kf.decode(kk, cc);
                pass = (int32_t)cnt >= vp.min_points && cc[0] >= vp.lo[0] && cc[0] < vp.hi[0] && cc[1] >= vp.lo[1] &&
                       cc[1] < vp.hi[1] && cc[2] >= vp.lo[2] && cc[2] < vp.hi[2];
            }
            // (an atomic store on purpose: with two plain conditional stores next to each other -- a second one at index
            // bb + j used to sit above this one -- hipcc 7.2 emitted, in one code shape of this kernel, a merged store that
            // wrote the value bb + j at firstmap[bb + j], i.e. the other store's index; found by the big-bucket tests, see
            // DESIGN.md 4a)
            if (pass) {
                __hip_atomic_store(&firstmap[f], V2 ? ((kFmRecord << kFmShift) | (bb + j)) : (single ? kSingleVoxel : bb + j),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                early_v++;
                early_p += cnt < early_clamp ? cnt : early_clamp;
            } else tfirst[sl] = kInf;             // (its points belong to no kept voxel: what precpos / pfirst_out hand on)
            if (!single) j++;
            if constexpr (ROWS)
                if (reduction != D3D_REDUCE_NONE && cnt > P) oslot[atomicAdd(&nover, 1u)] = (uint16_t)sl;
        };
        if constexpr (!decltype(BIG)::value) {
            // register bucket: lane t owns slots t, t + 512, .. -- every LDS access of this phase is a conflict-free row (four
            // CONSECUTIVE slots per lane put eight lanes on each bank) -- and the four rows' prefix sums travel as 16-bit
            // fields of one word (a row holds at most 2048 points): segments and records still follow slot order
            static_assert(PER == 4 && kBucketCap < 65536, "four 16-bit fields");
            uint32_t c[PER];
            u64 cs = 0, rs = 0;
#pragma unroll
            for (int k = 0; k < PER; k++) {
                c[k] = tcnt[threadIdx.x + k * kBucketThreads];
                cs |= (u64)c[k] << (16 * k);
                rs |= (u64)((kSkipSingles ? c[k] > 1 : c[k] > 0) ? 1u : 0u) << (16 * k);
            }
            u64 ex_c, ex_r, tot_c, tot_r;
            block_excl_scan_2u64_lds<kBucketThreads>(cs, rs, &ex_c, &ex_r, &tot_c, &tot_r, smem);
            D3D_PHASE(0, 7);                        // (records: the scan)
            uint32_t row_c = 0, row_r = 0;
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const uint32_t sl = threadIdx.x + k * kBucketThreads;
                const uint32_t base = row_c + (uint32_t)((ex_c >> (16 * k)) & 0xffffu);
                uint32_t j = row_r + (uint32_t)((ex_r >> (16 * k)) & 0xffffu);
                tcnt[sl] = c[k] | (base << 16);
                if (c[k]) slot_out(sl, c[k], base, j);
                row_c += (uint32_t)((tot_c >> (16 * k)) & 0xffffu);
                row_r += (uint32_t)((tot_r >> (16 * k)) & 0xffffu);
            }
        } else {
            const int s0 = threadIdx.x * PER;
            uint32_t c[PER];
            u64 mine = 0;
#pragma unroll
            for (int k = 0; k < PER; k++) { c[k] = tcnt[s0 + k]; mine += ((u64)c[k] << 32) | (c[k] > 0 ? 1u : 0u); }
            u64 all;
            u64 ex = block_excl_scan_u64_lds<kBucketThreads>(mine, &all, smem);
            uint32_t base = (uint32_t)(ex >> 32), j = (uint32_t)ex;
#pragma unroll
            for (int k = 0; k < PER; k++) {
                seg[s0 + k] = base;
                if (c[k]) slot_out((uint32_t)(s0 + k), c[k], base, j);
                base += c[k];
            }
        }
    };
    // Overflow voxels: every point counts (voxelize.cpp:137-157) but only P are ranked: one wavefront per voxel walks
    // its segment of point indices (sg: in LDS, or in global memory for a big bucket) 64 rows per step.  MEAN
    // accumulates in fp64 (insensitive to the order to ~1e-16 => the same float run to run; differs from the
    // reference's fp32 running sum by rounding only).  The result waits in row P of the voxel's segment, which no
    // ranked point uses.
    // (register buckets: the first 64 rows of the wavefront's first overflow voxel are requested BEFORE the rank phase --
    // overflow_first -- so that this gather's latency passes behind it)
    auto overflow_first = [&](const uint32_t *sg, v4f &pre) {
        const uint32_t no = ROWS && reduction != D3D_REDUCE_NONE ? nover : 0u, o = threadIdx.x >> 6;
        if (o >= no) return;
        const uint32_t w = tcnt[oslot[o]], lane = threadIdx.x & (kWave - 1);
        if (lane < (w & 0xffffu)) pre = *reinterpret_cast<const v4f *>(&points4[sg[(w >> 16) + lane]]);
    };
    auto reduce_overflow = [&](const uint32_t *sg, auto BIG, const v4f *pre) {
        const uint32_t no = ROWS && reduction != D3D_REDUCE_NONE ? nover : 0u;
        if (no == 0) return;
        const bool is_sum = reduction == D3D_REDUCE_MEAN || reduction == kReduceSum;
        const int lane = threadIdx.x & (kWave - 1);
        for (uint32_t o = threadIdx.x >> 6; o < no; o += kBucketThreads / kWave) {
            const uint32_t s = oslot[o], w = tcnt[s];
            const uint32_t base = decltype(BIG)::value ? seg[s] : w >> 16, cnt = decltype(BIG)::value ? w : w & 0xffffu;
            if (V2 && cnt <= P) continue;           // (V2 also lists the cells above kDenseMin points, for their ranking)
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            float e0, e1, e2, e3;
            e0 = e1 = e2 = e3 = reduction == D3D_REDUCE_MAX ? -INFINITY : INFINITY;
            // four steps of 64 rows in flight at a time (a voxel of 400 points was seven dependent gathers; the lane still adds
            // its rows in ascending order: the same sums)
            for (uint32_t k0 = lane; k0 < cnt; k0 += 4 * kWave) {
                v4f x[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t k = k0 + u * kWave;
                    if (k < cnt)
                        x[u] = (pre && o == (threadIdx.x >> 6) && k == (uint32_t)lane) ? *pre
                                                                                       : *reinterpret_cast<const v4f *>(&points4[sg[base + k]]);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (k0 + u * kWave >= cnt) break;
                    if (is_sum) { s0 += x[u].x; s1 += x[u].y; s2 += x[u].z; s3 += x[u].w; }
                    else if (reduction == D3D_REDUCE_MAX) {
                        e0 = e0 < x[u].x ? x[u].x : e0; e1 = e1 < x[u].y ? x[u].y : e1; e2 = e2 < x[u].z ? x[u].z : e2; e3 = e3 < x[u].w ? x[u].w : e3;
                    } else {
                        e0 = x[u].x < e0 ? x[u].x : e0; e1 = x[u].y < e1 ? x[u].y : e1; e2 = x[u].z < e2 ? x[u].z : e2; e3 = x[u].w < e3 ? x[u].w : e3;
                    }
                }
            }
            // over the wavefront on the DPP path (valid in lane 63)
            if (is_sum) {
                s0 = wave_sum_f64_lane63(s0); s1 = wave_sum_f64_lane63(s1);
                s2 = wave_sum_f64_lane63(s2); s3 = wave_sum_f64_lane63(s3);
            } else if (reduction == D3D_REDUCE_MAX) {
                e0 = wave_extreme_f32_lane63<true>(e0); e1 = wave_extreme_f32_lane63<true>(e1);
                e2 = wave_extreme_f32_lane63<true>(e2); e3 = wave_extreme_f32_lane63<true>(e3);
            } else {
                e0 = wave_extreme_f32_lane63<false>(e0); e1 = wave_extreme_f32_lane63<false>(e1);
                e2 = wave_extreme_f32_lane63<false>(e2); e3 = wave_extreme_f32_lane63<false>(e3);
            }
            if (lane == kWave - 1) {
                float4 out;
                if (is_sum) {
                    const float d = reduction == D3D_REDUCE_MEAN ? (float)(int32_t)cnt : 1.0f;
                    out = make_float4((float)s0 / d, (float)s1 / d, (float)s2 / d, (float)s3 / d);
                } else out = make_float4(e0, e1, e2, e3);
                staged[bb + base + P] = out;
            }
        }
    };

    if (m > (uint32_t)kBucketCap) {
        // BIG bucket (a few cells with thousands of points each -- returns piled up at the origin, a coarse grid): the same
        // phases as below, but looping over the bucket with the per-point state {slot, arrival} and the index segments
        // in global memory instead of registers / LDS.  Slower per point, no capacity limit on the points; only a bucket
        // with more distinct cells than table slots (or 2 M points) is handed back to the hash-table path.
        constexpr uint32_t kArrBits = 21, kArrMask = (1u << kArrBits) - 1;
        __syncthreads();
        if (m <= kArrMask) {
            for (uint32_t q = threadIdx.x; q < m; q += kBucketThreads) {
                const uint32_t pos = locate(q);
                gpos[bb + q] = pos;
                const typename E::type e = bent[pos];
                const u64 key64 = E::key(e);
                const KT key = (KT)key64;
                uint32_t s = (Key::bin_hash(key64) >> hshift) & (T - 1), probes = 0;
                for (;;) {
                    const KT old = atomicCAS(&tkey[s], kFree, key);
                    if (old == kFree || old == key) break;
                    s = (s + 1) & (T - 1);
                    if (++probes >= (uint32_t)T) break;      // every slot holds another cell
                }
                if (probes >= (uint32_t)T) { fail = 1; continue; }
                pinfo[bb + q] = (s << kArrBits) | atomicAdd(&tcnt[s], 1u);
                atomicMin(&tfirst[s], E::idx(e));
            }
        } else if (threadIdx.x == 0) fail = 1;
        __syncthreads();
        if (fail) {                                 // the caller repeats the call on the hash path; until then the outputs
            if (precpos)                            // stay consistent (these points map to no voxel)
                for (uint32_t q = threadIdx.x; q < m; q += kBucketThreads) precpos[m <= kArrMask ? gpos[bb + q] : locate(q)] = kInf;
            if (pfirst_out)
                for (uint32_t q = threadIdx.x; q < m; q += kBucketThreads)
                    pfirst_out[E::idx(bent[m <= kArrMask ? gpos[bb + q] : locate(q)])] = kInf;
            if (threadIdx.x == 0) atomicOr(reinterpret_cast<u64 *>(&counts[D3D_COUNT_STATUS]), (u64)D3D_VOXEL_STATUS_BIN_OVERFLOW);
            return;
        }
        const uint32_t *tbase = seg;
        records(std::true_type{});
        early_publish();
        __syncthreads();
        uint32_t *sg = gseg + bb;
        for (uint32_t q = threadIdx.x; q < m; q += kBucketThreads) {
            const uint32_t w = pinfo[bb + q];
            sg[tbase[w >> kArrBits] + (w & kArrMask)] = E::idx(bent[gpos[bb + q]]);
        }
        __threadfence_block();
        __syncthreads();                            // (waits for the stores: the segments are read back below)
        for (uint32_t q = threadIdx.x; q < m; q += kBucketThreads) {
            const uint32_t pos = gpos[bb + q];
            const uint32_t s = pinfo[bb + q] >> kArrBits, cnt = tcnt[s], base = tbase[s], me = E::idx(bent[pos]);
            uint32_t rank = 0, k = 0;
            const uint32_t *v = sg + base;
            // (the rank of a voxel's only point is 0; with nothing to stage or list, only voxels above P need ranks at all)
            const bool need = (STAGE || LISTS) ? cnt > 1 : cnt > P;
            for (; need && k + 4 <= cnt && rank < P; k += 4) rank += (v[k] < me) + (v[k + 1] < me) + (v[k + 2] < me) + (v[k + 3] < me);
            for (; need && k < cnt && rank < P; k++) rank += v[k] < me;
            if constexpr (STAGE) { if (rank < P) staged[bb + base + rank] = points4[me]; }
            if constexpr (LISTS) {
                if (rank < P && (rank > 0 || !ROWS)) sorted_out[bb + base + rank] = me;
                if (unsorted_out && cnt > P) unsorted_out[bb + base + (pinfo[bb + q] & kArrMask)] = me;
            }
            if (trimmed && rank >= P) trimmed[me] = 1;
            if (precpos) precpos[pos] = tfirst[s];
            if (pfirst_out) {
                const uint32_t pf = (P > 0 && rank >= P) ? kInf : tfirst[s];
                if (SPV2 || !table || pf != me) pfirst_out[me] = pf;        // (tile-sorted input: k_tile_sort preset pfirst[me] = me)
            }
        }
        reduce_overflow(sg, std::true_type{}, (const v4f *)nullptr);
        return;
    }

    if constexpr (V2) {
        // ---- round 5, register bucket of the index for k_emit.  Against the path below: no first-index table (the point of
        // rank 0 IS the first point and stores the voxel's entry itself: one LDS atomic per point less, no scattered stores
        // in the slot-ordered records phase, which shrinks to the scan of the counts), the four CAS of a lane in flight
        // together, no record for a voxel below 255 points (the entry carries count and segment), and the cells with more
        // than kDenseMin points -- whose per-point counting loops kept whole wavefronts waiting for one lane -- ranked by ONE
        // wavefront each: radix-64 select of the max_points smallest indices, all-pairs ranks among those.
        // SP: the sparse contract fused with its voxel filter on the SAME kernel (BoundKey): no rows, no reductions; every kept
        // point needs a handle of its voxel (its first point's index) for the compaction, a voxel below min_points gets no
        // entry, only cells above max_points need ranks at all (TRIM, voxelize.cpp:457-463) -- so the first index of a cell is
        // found by an atomicMin again instead of by the rank-0 point.
        constexpr bool SP = std::is_same<Key, BoundKey>::value;
        KT key[ITEMS];
        uint32_t idx[ITEMS], slot[ITEMS], arr[ITEMS];
        if (table) lds_barrier();                   // the run table is complete
        static_for<ITEMS>([&](auto R) {
            constexpr int r = decltype(R)::value;
            const uint32_t q = threadIdx.x + r * kBucketThreads;
            key[r] = 0; idx[r] = 0;
            if (q < m) {
                uint32_t p;
                if (table) {
                    const uint32_t t = tileof[q];
                    p = seg[kRunCap + t] + (q - seg[t]);
                } else p = bb + q;
                const typename E::type e = bent[p];
                key[r] = (KT)E::key(e);
                idx[r] = E::idx(e);
            }
        });
        lds_barrier();
        D3D_PHASE(0, 1);
#ifdef D3D_PHASE_CLOCKS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        D3D_PHASE(0, 8);
#endif
        KT old[ITEMS];
        static_for<ITEMS>([&](auto R) {             // first probe of all items: the atomics' round trips overlap
            constexpr int r = decltype(R)::value;
            const uint32_t q = threadIdx.x + r * kBucketThreads;
            slot[r] = (Key::bin_hash((u64)key[r]) >> hshift) & (T - 1);
            old[r] = kFree;
            if (q < m) old[r] = atomicCAS(&tkey[slot[r]], kFree, key[r]);
        });
        static_for<ITEMS>([&](auto R) {             // (distinct cells <= m <= T: a free slot always exists)
            constexpr int r = decltype(R)::value;
            const uint32_t q = threadIdx.x + r * kBucketThreads;
            if (q < m) {
                uint32_t s = slot[r];
                KT o = old[r];
                while (o != kFree && o != key[r]) {
                    s = (s + 1) & (T - 1);
                    o = atomicCAS(&tkey[s], kFree, key[r]);
                }
                slot[r] = s;
            }
        });
        static_for<ITEMS>([&](auto R) {
            constexpr int r = decltype(R)::value;
            const uint32_t q = threadIdx.x + r * kBucketThreads;
            arr[r] = 0;
            if (q < m) {
                arr[r] = atomicAdd(&tcnt[slot[r]], 1u);
                if constexpr (SP) atomicMin(&tfirst[slot[r]], idx[r]);
            }
        });
        lds_barrier();
        D3D_PHASE(0, 2);
        {
            // segments in slot order: lane t owns slots t, t + 512, .. (conflict-free rows); the four rows' prefix sums travel
            // as 16-bit fields of one word.  Cells for a wavefront of their own -- more than kDenseMin points (ranking) or more
            // than P with a reduction (every point counts, voxelize.cpp:137-157) -- are listed.
            constexpr int PER = T / kBucketThreads;
            static_assert(PER == 4 && kBucketCap < 65536, "four 16-bit fields");
            uint32_t c[PER];
            u64 cs = 0;
#pragma unroll
            for (int k = 0; k < PER; k++) {
                c[k] = tcnt[threadIdx.x + k * kBucketThreads];
                cs |= (u64)c[k] << (16 * k);
            }
            u64 tot_c;
            const u64 ex_c = block_excl_scan_u64_lds<kBucketThreads>(cs, &tot_c, smem);
            D3D_PHASE(0, 7);
            uint32_t row_c = 0;
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const uint32_t sl = threadIdx.x + k * kBucketThreads;
                const uint32_t base = row_c + (uint32_t)((ex_c >> (16 * k)) & 0xffffu);
                tcnt[sl] = c[k] | (base << 16);
                const bool listed = SP ? (P > 0 && c[k] > P && c[k] > dense_min) : (c[k] > dense_min || (reduction != D3D_REDUCE_NONE && c[k] > P));
                if (listed) oslot[atomicAdd(&nover, 1u)] = (uint16_t)sl;
                row_c += (uint32_t)((tot_c >> (16 * k)) & 0xffffu);
            }
        }
        lds_barrier();
        D3D_PHASE(0, 3);
        static_for<ITEMS>([&](auto R) {
            constexpr int r = decltype(R)::value;
            const uint32_t q = threadIdx.x + r * kBucketThreads;
            if (q < m) {
                const uint32_t cb = tcnt[slot[r]];
                if (!SP || (P > 0 && (cb & 0xffffu) > P)) seg[(cb >> 16) + arr[r]] = idx[r];
            }
        });
        lds_barrier();
        D3D_PHASE(0, 4);
        // the voxel's entry, by its first point (a relaxed atomic store: see the note at the other firstmap store)
        auto first_entry = [&](uint32_t f, uint32_t cnt, uint32_t base, uint32_t s) {
            uint32_t w = (cnt << kFmShift) | (bb + base);
            if (cnt >= kFmRecord) {
                const u64 kk = (u64)tkey[s];
                vrec[bb + base] = make_uint4((uint32_t)kk, (uint32_t)(kk >> 32), bb + base, cnt);
                w = (kFmRecord << kFmShift) | (bb + base);
            }
            // (a plain store -- inline assembly, so nothing can be merged -- left the kernel unchanged at 1 M points and made
            // it 5 % faster at 8 M, but k_emit behind it 5 % SLOWER at 4 M points: profiles/r05_b_tune.txt.  Not taken.)
            __hip_atomic_store(&firstmap[f], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // the `want` smallest point indices of a crowded cell's segment: radix-64 select, six bits a round -- bins
        // [lo + (b << shift), lo + ((b + 1) << shift)); returns tau: the wanted indices are exactly those below it
        auto select_tau = [&](const uint32_t *sg, uint32_t cnt, uint32_t want, uint32_t *hist) -> uint32_t {
            const uint32_t lane = threadIdx.x & (kWave - 1), K = (cnt + kWave - 1) / kWave;
            uint32_t lo = 0, below = 0;
            int shift = idx_bits > 6 ? idx_bits - 6 : 0;
            if (cnt <= want) return 0xffffffffu;
            for (;;) {
                hist[lane] = 0;
                wave_lds_fence();
                for (uint32_t k = 0; k < K; k++) {
                    const uint32_t t = k * kWave + lane;
                    if (t < cnt) {
                        const uint32_t e = sg[t], d = (e - lo) >> shift;
                        if (e >= lo && d < (uint32_t)kWave) atomicAdd(&hist[d], 1u);
                    }
                }
                wave_lds_fence();
                const uint32_t c = hist[lane], incl = wave_incl_scan_u32(c);
                const unsigned long long hit = __ballot(below + incl >= want);     // never empty: the range holds >= want - below
                const int b = __ffsll((long long)hit) - 1;
                const uint32_t ex_b = (uint32_t)__builtin_amdgcn_readlane((int)(incl - c), b);
                const uint32_t c_b = (uint32_t)__builtin_amdgcn_readlane((int)c, b);
                if (below + ex_b + c_b == want || shift == 0) return lo + (((uint32_t)b + 1u) << shift);
                below += ex_b;
                lo += (uint32_t)b << shift;
                shift = shift > 6 ? shift - 6 : 0;
            }
        };
        if constexpr (SP) {
            // every kept point leaves the handle of its voxel -- the index of the voxel's first point, under which
            // k_sparse_finish publishes the voxel's id -- for the compaction; the first point leaves the voxel's entry instead,
            // if the voxel passes min_points (its coordinates are inside the bounds by construction of the key)
            static_for<ITEMS>([&](auto R) {
                constexpr int r = decltype(R)::value;
                const uint32_t q = threadIdx.x + r * kBucketThreads;
                if (q < m) {
                    const uint32_t s = slot[r], cb = tcnt[s], cnt = cb & 0xffffu, base = cb >> 16, me = idx[r];
                    const bool pass = !vp.on || (int32_t)cnt >= vp.min_points;
                    bool kept = pass, later = false;
                    if (pass && P > 0 && cnt > P) {                      // TRIM: the first P points by index stay (voxelize.cpp:457-463)
                        if (cnt > dense_min) later = true;               // (the cell's wavefront decides, below)
                        else {
                            uint32_t rank = 0, k = 0;
                            const uint32_t *sg = seg + base;
                            for (; k + 8 <= cnt && rank < P; k += 8)
                                rank += (sg[k] < me) + (sg[k + 1] < me) + (sg[k + 2] < me) + (sg[k + 3] < me) + (sg[k + 4] < me) +
                                        (sg[k + 5] < me) + (sg[k + 6] < me) + (sg[k + 7] < me);
                            for (; k < cnt && rank < P; k++) rank += sg[k] < me;
                            kept = rank < P;
                        }
                    }
                    if (!later && kept) {
                        const uint32_t f = tfirst[s];
                        if (f == me) {
                            first_entry(me, cnt, base, s);
                            early_v++;
                            early_p += cnt < early_clamp ? cnt : early_clamp;
                        } else pfirst_out[me] = f;
                    }
                }
            });
            D3D_PHASE(0, 5);
            const uint32_t no = nover, lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
            uint32_t *hist = whist[V2 ? wv : 0];
            for (uint32_t o = wv; o < no; o += kBucketThreads / kWave) {
                const uint32_t s = oslot[o], cb = tcnt[s], cnt = cb & 0xffffu, base = cb >> 16;
                if (vp.on && (int32_t)cnt < vp.min_points) continue;
                const uint32_t *sg = seg + base;
                const uint32_t tau = select_tau(sg, cnt, P, hist), f = tfirst[s];
                for (uint32_t t = lane; t < cnt; t += kWave) {
                    const uint32_t e = sg[t];
                    if (e >= tau) continue;
                    if (e == f) {
                        first_entry(e, cnt, base, s);
                        early_v++;
                        early_p += cnt < early_clamp ? cnt : early_clamp;
                    } else pfirst_out[e] = f;
                }
            }
            early_publish();
            D3D_PHASE(0, 6);
            return;
        }
        v4f over_pre = {0.f, 0.f, 0.f, 0.f};
        overflow_first(seg, over_pre);
        static_for<ITEMS>([&](auto R) {
            constexpr int r = decltype(R)::value;
            const uint32_t q = threadIdx.x + r * kBucketThreads;
            if (q < m) {
                const uint32_t s = slot[r], cb = tcnt[s], cnt = cb & 0xffffu, base = cb >> 16, me = idx[r];
                if (cnt <= dense_min) {
                    uint32_t rank = 0, k = 0;
                    const uint32_t *sg = seg + base;
                    for (; k + 8 <= cnt && rank < P; k += 8)          // 8 independent LDS reads per exit test
                        rank += (sg[k] < me) + (sg[k + 1] < me) + (sg[k + 2] < me) + (sg[k + 3] < me) + (sg[k + 4] < me) + (sg[k + 5] < me) +
                                (sg[k + 6] < me) + (sg[k + 7] < me);
                    for (; k < cnt && rank < P; k++) rank += sg[k] < me;
                    if (rank == 0) first_entry(me, cnt, base, s);
                    else if (rank < P) sorted_out[bb + base + rank] = me;
                }
            }
        });
        D3D_PHASE(0, 5);
        // listed cells, one wavefront each: the reduction over ALL points first (it reads the whole segment), then the ranking,
        // which compacts the segment in place
        {
            const uint32_t no = nover, lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
            const bool reduce = reduction != D3D_REDUCE_NONE;
            if (reduce) reduce_overflow(seg, std::false_type{}, &over_pre);
            uint32_t *hist = whist[V2 ? wv : 0];
            for (uint32_t o = wv; o < no; o += kBucketThreads / kWave) {
                const uint32_t s = oslot[o], cb = tcnt[s], cnt = cb & 0xffffu, base = cb >> 16;
                if (cnt <= dense_min) continue;                     // (listed for its reduction only)
                uint32_t *sg = seg + base;
                const uint32_t want = cnt < P ? cnt : P, K = (cnt + kWave - 1) / kWave;
                const uint32_t tau = select_tau(sg, cnt, want, hist);
                // kept = below tau, compacted to the front of the segment (forward, in place: position <= index read)
                uint32_t run = 0;
                for (uint32_t k = 0; k < K; k++) {
                    const uint32_t t = k * kWave + lane;
                    const uint32_t e = t < cnt ? sg[t] : 0xffffffffu;
                    const bool keep = t < cnt && e < tau;
                    const unsigned long long kb = __ballot(keep);
                    wave_lds_fence();
                    if (keep) sg[run + (uint32_t)__popcll(kb & ((1ull << lane) - 1ull))] = e;
                    run += (uint32_t)__popcll(kb);
                    wave_lds_fence();
                }
                // ranks among the kept (run == want of them): every kept index against all of them, broadcast reads
                for (uint32_t j0 = 0; j0 < want; j0 += kWave) {
                    const uint32_t t = j0 + lane;
                    const uint32_t me = t < want ? sg[t] : 0u;
                    uint32_t rank = 0;
                    for (uint32_t u = 0; u < want; u++) rank += sg[u] < me;
                    if (t < want) {
                        if (rank == 0) first_entry(me, cnt, base, s);
                        else sorted_out[bb + base + rank] = me;
                    }
                }
            }
        }
        D3D_PHASE(0, 6);
        return;
    }

    v4f row[STAGE ? ITEMS : 1];
    u64 key_in[ITEMS];
    uint32_t idx[ITEMS], slot[ITEMS], arr[ITEMS], pos[ITEMS];
    if (table) lds_barrier();                       // the run table is complete
    // branch-free loads (lanes past the end repeat the last entry): all ITEMS entry loads, then all row gathers, in flight
    // together.  The rows are only needed when the ranks are known: their gather (one scattered 16-byte load per point
    // from the cache-resident point tensor) runs behind the LDS phases (lds_barrier does not wait for it).
    static_for<ITEMS>([&](auto R) {
        constexpr int r = decltype(R)::value;
        const uint32_t q = threadIdx.x + r * kBucketThreads;
        pos[r] = 0; key_in[r] = 0; idx[r] = 0;
        if (q < m) {                                // (items past the bucket's end are never looked at again)
            if (table) {
                const uint32_t t = tileof[q];
                pos[r] = seg[kRunCap + t] + (q - seg[t]);
            } else pos[r] = bb + q;
            const typename E::type e = bent[pos[r]];
            key_in[r] = E::key(e);
            idx[r] = E::idx(e);
        }
    });
    if constexpr (STAGE) {
        static_for<ITEMS>([&](auto R) {
            constexpr int r = decltype(R)::value;
            row[r] = *reinterpret_cast<const v4f *>(&points4[idx[r]]);
        });
    }
    lds_barrier();
    D3D_PHASE(0, 1);                                // tables cleared, entries located, loads issued
#ifdef D3D_PHASE_CLOCKS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    D3D_PHASE(0, 8);                                // (insert: the entries arrived)
#endif
    static_for<ITEMS>([&](auto R) {
        constexpr int r = decltype(R)::value;
        const uint32_t q = threadIdx.x + r * kBucketThreads;
        if (q < m) {
            const KT key = (KT)key_in[r];
            uint32_t s = (Key::bin_hash(key_in[r]) >> hshift) & (T - 1);
            for (;;) {                              // distinct cells <= m <= T: a free slot always exists
                const KT old = atomicCAS(&tkey[s], kFree, key);
                if (old == kFree || old == key) break;
                s = (s + 1) & (T - 1);
            }
            slot[r] = s;
            arr[r] = atomicAdd(&tcnt[s], 1u);
            atomicMin(&tfirst[s], idx[r]);
        }
    });
    lds_barrier();
    D3D_PHASE(0, 2);                                // entries arrived, inserted
    records(std::false_type{});
    early_publish();
    lds_barrier();
    D3D_PHASE(0, 3);                                // records
    static_for<ITEMS>([&](auto R) {
        constexpr int r = decltype(R)::value;
        const uint32_t q = threadIdx.x + r * kBucketThreads;
        if (q < m) seg[(tcnt[slot[r]] >> 16) + arr[r]] = idx[r];
    });
    lds_barrier();
    D3D_PHASE(0, 4);                                // segments
    v4f over_pre = {0.f, 0.f, 0.f, 0.f};
    if constexpr (ROWS) overflow_first(seg, over_pre);
    static_for<ITEMS>([&](auto R) {
        constexpr int r = decltype(R)::value;
        const uint32_t q = threadIdx.x + r * kBucketThreads;
        if (q < m) {
            const uint32_t s = slot[r], cb = tcnt[s], cnt = cb & 0xffffu, base = cb >> 16, me = idx[r];
            uint32_t rank = 0, k = 0;
            const uint32_t *sg = seg + base;
            // (the rank of a voxel's only point is 0; with nothing to stage or list -- the sparse contract's TRIM filter -- only
            // the points of voxels above P need ranks at all)
            const bool need = (STAGE || LISTS) ? cnt > 1 : cnt > P;
            for (; need && k + 8 <= cnt && rank < P; k += 8)      // 8 independent LDS reads per exit test
                rank += (sg[k] < me) + (sg[k + 1] < me) + (sg[k + 2] < me) + (sg[k + 3] < me) + (sg[k + 4] < me) + (sg[k + 5] < me) +
                        (sg[k + 6] < me) + (sg[k + 7] < me);
            for (; need && k < cnt && rank < P; k++) rank += sg[k] < me;
            if constexpr (STAGE) { if (rank < P) *reinterpret_cast<v4f *>(&staged[bb + base + rank]) = row[r]; }
            if constexpr (LISTS) {                                 // dense contract, C != 4: index lists for k_fill_generic /
                if (rank < P && (rank > 0 || !ROWS)) sorted_out[bb + base + rank] = me;   // k_aggregate; C == 4: for k_emit
                if (unsorted_out && cnt > P) unsorted_out[bb + base + arr[r]] = me;
            }
            if (trimmed && rank >= P) trimmed[me] = 1;     // sparse contract + TRIM filter (voxelize.cpp:457-463)
            if (precpos) precpos[pos[r]] = tfirst[s];
            if (pfirst_out) {
                const uint32_t pf = (P > 0 && rank >= P) ? kInf : tfirst[s];
                if (!table || pf != me) pfirst_out[me] = pf;        // (tile-sorted input: k_tile_sort preset pfirst[me] = me)
            }
        }
    });
    D3D_PHASE(0, 5);                                // ranks, list stores issued
    reduce_overflow(seg, std::false_type{}, ROWS ? &over_pre : (const v4f *)nullptr);
    D3D_PHASE(0, 6);                                // overflow voxels reduced
}

// 64 firstmap entries -> one count; counts scanned inside the block (fwpre), block totals -> bsumF (<= 1024 of them:
// k_meta_first adds up the ones before its tile itself, which is cheaper than a scan launch or a last-block pass)
__global__ __launch_bounds__(1024) void k_first_count(const uint32_t *__restrict__ firstmap, uint32_t *fwpre, uint32_t *bsumF,
                                                      uint32_t *clear_word = nullptr /* k_emit_c's overflow-voxel counter */,
                                                      ZeroFill zf = ZeroFill())
{
    if (zero_fill_role(zf)) return;
    if (clear_word && blockIdx.x == 0 && threadIdx.x == 0) *clear_word = 0;
    __shared__ u64 smem[1024 / kWave];
    __shared__ uint32_t wcnt[256];
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    // wavefront w: words [16 w, 16 w + 16) of the tile's 256, all 16 loads in flight
    const uint32_t *src = firstmap + ((size_t)blockIdx.x * 256 + w * 16) * 64 + lane;
    uint32_t e[16];
#pragma unroll
    for (int j = 0; j < 16; j++) e[j] = src[(size_t)j * 64];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const unsigned long long bal = __ballot(e[j] != kInf);
        if (lane == j) wcnt[w * 16 + j] = (uint32_t)__popcll(bal);
    }
    __syncthreads();
    u64 tot;
    const u64 ex = block_excl_scan_u64<1024>(threadIdx.x < 256 ? (u64)wcnt[threadIdx.x] : 0ull, &tot, smem);
    if (threadIdx.x < 256) fwpre[(size_t)blockIdx.x * 256 + threadIdx.x] = (uint32_t)ex;
    if (threadIdx.x == 0) bsumF[blockIdx.x] = (uint32_t)tot;
}

// one lane per point index: the lanes that are a voxel's first point number it (prefix count = the reference's
// first-occurrence order, voxelize.cpp:119), fetch its record and write all per-voxel outputs -- coalesced, because
// consecutive first points are consecutive voxel ids
// the lane whose point index i is a voxel's first point (e = its firstmap entry) and that voxel's id: record, per-voxel outputs.
// Returns the points the voxel keeps (fused sparse + filter); every lane leaves the id of the voxel that starts at i.
template <class Key, bool AGG4>
__device__ __forceinline__ uint32_t meta_first_lane(const Key &kf, int64_t i, uint32_t e, uint32_t vid, const uint4 *__restrict__ vrec,
                                                    uint32_t max_voxels, uint4 *__restrict__ vinfo, const float4 *__restrict__ staged,
                                                    uint32_t P, int reduction, int64_t *coords, int32_t *npoints, unsigned char *pmask,
                                                    float4 *agg, const BinnedExtras &x, const float *__restrict__ points, int c)
{
    uint32_t kept = 0, myvid = kNoVoxel;
    if (e != kInf && vid < max_voxels) {                    // voxelize.cpp:116-117: later voxels are never created
        myvid = vid;
        if (x.first_out) x.first_out[vid] = x.index_offset + i;
        uint4 rec;
        if (e != kSingleVoxel) rec = vrec[e];
        else {              // (sparse contract) a voxel of one point has no record: its cell from the point -- this lane's own index
            const float *src = points + i * c;
            const float v3[3] = {src[0], src[1], src[2]};
            u64 key = 0;
            uint32_t st = 0;
            (void)kf.make(v3, key, st);                     // the same arithmetic on the same floats as k_bin_count
            rec = make_uint4((uint32_t)key, (uint32_t)(key >> 32), 0u, 1u);
        }
        const uint4 vi = rec;                               // {key lo, key hi, segment base, count}
        if (vinfo) vinfo[vid] = vi;
        if (x.count_out) x.count_out[vid] = (int32_t)vi.w;
        kept = vi.w < x.npoints_clamp ? vi.w : x.npoints_clamp;
        meta_voxel<Key, AGG4>(kf, (int64_t)vid, vi, staged, P, reduction, coords, npoints, x.voff, pmask, agg, nullptr, nullptr,
                              x.keys_out, x.npoints_clamp, x.has_coord_sub ? x.coord_sub : (const long long *)nullptr);
    }
    if (x.vidof) x.vidof[i] = myvid;                        // every lane: one coalesced store (the map looks it up by first point)
    return kept;
}

template <class Key, bool AGG4>
__global__ __launch_bounds__(256) void k_meta_first(Key kf, int64_t npad, const uint32_t *__restrict__ firstmap,
                                                    const uint32_t *__restrict__ fwpre, const uint32_t *__restrict__ bsumF,
                                                    const uint4 *__restrict__ vrec, uint32_t max_voxels, uint4 *__restrict__ vinfo,
                                                    const float4 *__restrict__ staged, uint32_t P, int reduction, int64_t *coords,
                                                    int32_t *npoints, unsigned char *pmask, float4 *agg, int64_t *counts,
                                                    BinnedExtras x, const float *__restrict__ points, int c)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    // voxels before this workgroup's tile of 16384 point indices (wave-uniform; <= 1024 tile totals)
    const uint32_t tile = (uint32_t)(i / kFlagTile), ntile = (uint32_t)(npad / kFlagTile);
    uint32_t before = 0, all = 0;
    for (uint32_t t = lane; t < ntile; t += kWave) {
        const uint32_t x = bsumF[t];
        all += x;
        if (t < tile) before += x;
    }
    {
        const u64 both = wave_sum_u64(((u64)all << 32) | before);       // (each below 2^32: the halves do not carry)
        before = (uint32_t)both;
        all = (uint32_t)(both >> 32);
    }
    if (i == 0) {
        counts[D3D_COUNT_VOXELS] = (int64_t)(all < max_voxels ? all : max_voxels);
        counts[D3D_COUNT_AUX] = 0;
        // sharded voxelizer: the status bits travel with the key list (row `status_row`, negative = not a cell)
        if (x.keys_out && x.status_row >= 0) x.keys_out[x.status_row] = -1 - counts[D3D_COUNT_STATUS];
        if (x.host_counts) notify_host(counts, x.host_counts);
    }
    const uint32_t e = firstmap[i];
    const unsigned long long bal = __ballot(e != kInf);
    const uint32_t vid = before + fwpre[i >> 6] + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    (void)meta_first_lane<Key, AGG4>(kf, i, e, vid, vrec, max_voxels, vinfo, staged, P, reduction, coords, npoints, pmask, agg, x, points, c);
}

// Fused sparse + filter (round 4): numbering, per-voxel outputs AND both output sizes in ONE launch -- no k_first_count before
// it, no k_publish_kept behind it.  Tiles of 4096 point indices (a few hundred workgroups: the ticket word of the look-back
// serves them in a microsecond or two; k_emit's 4000 waited 14 us for it, see binned_index): the voxels before a tile by
// decoupled look-back (common.hpp), the points its voxels keep published next to it; the LAST tile -- it has waited for
// every predecessor's voxel count anyway -- adds the kept points of all tiles up and tells the host.  Also clears the
// look-back words of the launch behind it (k_compact_kept).
constexpr int kMetaLbThreads = 1024, kMetaLbItems = 4, kMetaLbTile = kMetaLbThreads * kMetaLbItems;
struct MetaLb {
    u64 *stat;              // [tiles] voxel look-back words, [tiles .. 2 tiles) kept points | ready bit; cleared by k_tile_sort /
    unsigned int *ticket;   //   k_bin_count with the ticket
    u64 *next_stat;         // k_compact_kept's words + ticket, cleared here
    uint32_t next_n;
    unsigned int *next_ticket;
    int64_t *host;          // 2 * D3D_NUM_COUNTS + 1 words (d3d_voxelize_3d_sparse_filter)
    const u64 *early;       // k_bucket_index's pairs {passing voxels, kept points}, one per 16 words: complete when this launch starts
    uint32_t early_pairs;   // 0 = off
};
template <class Key>
__global__ __launch_bounds__(kMetaLbThreads) void k_meta_first_lb(Key kf, int64_t npad, const uint32_t *__restrict__ firstmap,
                                                                  const uint4 *__restrict__ vrec, uint32_t max_voxels, int64_t *coords,
                                                                  int32_t *npoints, int64_t *counts, BinnedExtras x, MetaLb lb,
                                                                  const float *__restrict__ points, int c)
{
    __shared__ unsigned int sid;
    __shared__ uint32_t wtot[kMetaLbThreads / kWave];
    __shared__ u64 sprefix;
    const unsigned int tile = lookback_ticket(lb.ticket, &sid);
    const unsigned int ntiles = gridDim.x;
    if (tile == 0) {
        // the output sizes are known before any voxel has its number -- unless max_voxels cuts the frame short (which voxels
        // then exist is a matter of the numbering: the last tile reports, as it did for every frame before) -- and the host,
        // waiting for them to size the outputs, gets them a launch earlier: it returns and enqueues the NEXT frame while this
        // launch and k_compact_kept (38 us at config 2) run, instead of k_compact_kept (18 us) alone
        if (lb.host && lb.early_pairs && threadIdx.x < kWave) {
            const bool have = threadIdx.x < lb.early_pairs;
            const u64 tv = wave_sum_u64(have ? lb.early[16 * threadIdx.x] : 0ull), tp = wave_sum_u64(have ? lb.early[16 * threadIdx.x + 1] : 0ull);
            if (threadIdx.x == 0 && tv <= (u64)max_voxels) {
                for (int k = 0; k < D3D_NUM_COUNTS; k++) {
                    lb.host[k] = counts[k];
                    lb.host[D3D_NUM_COUNTS + 1 + k] = 0;
                }
                lb.host[D3D_COUNT_VOXELS] = (int64_t)tv;
                lb.host[D3D_COUNT_AUX] = 0;
                lb.host[D3D_NUM_COUNTS + 1 + D3D_COUNT_VOXELS] = (int64_t)tv;
                lb.host[D3D_NUM_COUNTS + 1 + D3D_COUNT_POINTS] = (int64_t)tp;
                __threadfence_system();
                __hip_atomic_store(&lb.host[D3D_NUM_COUNTS], (int64_t)1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        for (uint32_t t = threadIdx.x; t < lb.next_n; t += kMetaLbThreads) lb.next_stat[t] = 0ull;
        if (threadIdx.x == 0) *lb.next_ticket = 0u;
    }
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int64_t base = (int64_t)tile * kMetaLbTile + (int64_t)w * (kWave * kMetaLbItems) + lane;     // (w, row, lane) = index order
    uint32_t e[kMetaLbItems], ex[kMetaLbItems], carry = 0;
#pragma unroll
    for (int k = 0; k < kMetaLbItems; k++) e[k] = firstmap[base + (int64_t)k * kWave];                // (padded to the tile)
#pragma unroll
    for (int k = 0; k < kMetaLbItems; k++) {
        const unsigned long long bal = __ballot(e[k] != kInf);
        ex[k] = carry + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        carry += (uint32_t)__popcll(bal);
    }
    if (lane == 0) wtot[w] = carry;
    __syncthreads();
    uint32_t total = 0, woff = 0;
#pragma unroll
    for (int k = 0; k < kMetaLbThreads / kWave; k++) {
        const uint32_t t = wtot[k];
        total += t;
        if (k < w) woff += t;
    }
    if (w == 0) {
        const u64 before = lookback_exclusive(lb.stat, tile, (u64)total);
        if (lane == 0) sprefix = before;
    }
    __syncthreads();
    const uint32_t pre = (uint32_t)sprefix + woff;
    uint32_t kept = 0;
#pragma unroll
    for (int k = 0; k < kMetaLbItems; k++)
        kept += meta_first_lane<Key, false>(kf, base + (int64_t)k * kWave, e[k], pre + ex[k], vrec, max_voxels, (uint4 *)nullptr,
                                            (const float4 *)nullptr, 0u, (int)D3D_REDUCE_NONE, coords, npoints, (unsigned char *)nullptr,
                                            (float4 *)nullptr, x, points, c);
    // the tile's kept points next to its voxel word; the last tile adds all of them up
    kept = (uint32_t)wave_sum_u64((u64)kept);
    __syncthreads();                                        // (wtot is read above by every thread)
    if (lane == 0) wtot[w] = kept;
    __syncthreads();
    if (w != 0) return;
    u64 mine = 0;
    for (int k = lane; k < kMetaLbThreads / kWave; k += kWave) mine += wtot[k];
    mine = wave_sum_u64(mine);
    if (tile + 1 != ntiles) {
        if (lane == 0) __hip_atomic_store(&lb.stat[ntiles + tile], (mine << 1) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    u64 acc = 0;
    for (unsigned int j0 = 0; j0 + 1 < ntiles; j0 += kWave) {
        const unsigned int j = j0 + lane;
        if (j + 1 < ntiles) {
            u64 t;
            do { t = __hip_atomic_load(&lb.stat[ntiles + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (!(t & 1ull));
            acc += t >> 1;
        }
    }
    acc = wave_sum_u64(acc) + mine;
    if (lane == 0) {
        const u64 all = sprefix + total;                    // voxels of the frame (this is the last tile)
        const int64_t nvox = (int64_t)(all < max_voxels ? all : max_voxels);
        counts[D3D_COUNT_VOXELS] = nvox;
        counts[D3D_COUNT_AUX] = 0;
        if (lb.host && !(lb.early_pairs && all <= (u64)max_voxels)) {          // (else the first tile has told the host)
            for (int k = 0; k < D3D_NUM_COUNTS; k++) {
                lb.host[k] = counts[k];
                lb.host[D3D_NUM_COUNTS + 1 + k] = 0;
            }
            lb.host[D3D_NUM_COUNTS + 1 + D3D_COUNT_VOXELS] = nvox;
            lb.host[D3D_NUM_COUNTS + 1 + D3D_COUNT_POINTS] = (int64_t)acc;
            __threadfence_system();
            __hip_atomic_store(&lb.host[D3D_NUM_COUNTS], (int64_t)1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Dense contract, C == 4: numbering, per-voxel outputs AND the voxel's rows of voxels[V,P,4] in one launch.  (The two-launch
// stage -- k_meta_first, then k_fill_c4 -- reads every voxel's record and first row twice, through two random 16-byte gathers
// each, passes a 16-byte vinfo record per voxel between them, and needs every row staged by k_bucket_index first.)
// A wavefront owns 64 consecutive point indices; the nv of them that are a voxel's first point are nv CONSECUTIVE voxel ids
// vid0 .. vid0 + nv - 1 (first-seen numbering, voxelize.cpp:119), i.e. one contiguous stretch of every output:
//   1. the record positions and the point indices of those first points are compacted across the lanes (ds_permute); lane
//      l < nv gathers voxel vid0 + l's record -- the one random access per voxel;
//   2. ALL kept rows of the wavefront's voxels (sum of min(count, P): ~63 for a LiDAR frame) are fetched in one flat,
//      lane-parallel pass into a row buffer in LDS: row t belongs to voxel j (binary search over the voxels' row offsets) at
//      rank k; its point index is the voxel's first point (k = 0: ascending indices inside one 1 KiB window of the point
//      tensor, nothing was staged for it) or entry k of the voxel's ranked list.  Two dependent loads for the whole
//      wavefront, however many multi-point voxels it holds;
//   3. reductions: lane j walks ITS voxel's rows in the buffer in point order (sequential fp32, bit-identical to
//      voxelize.cpp:137-164); overflow voxels take the fp64 result k_bucket_index left in row P of their segment;
//   4. the stretch of nv * P rows is written 64 rows = 1 KiB per store instruction, rows from the buffer, zeros elsewhere: no
//      global load in this loop, the stores of a wavefront stream back to back;
//   5. coords / npoints / pmask / aggregates: one lane per voxel, coalesced over the stretch.
// A wavefront whose rows exceed the buffer (kEmitCap) takes its voxels in batches of consecutive ids.
constexpr int kEmitCap = 256;                     // rows per wavefront in LDS (4 KiB); max_points <= kEmitCap on this path


// RESIDENT (d3d_voxelize_3d_dense_resident): voxels[capacity, P, 4] is a buffer the caller keeps from frame to frame, with
// row_state[v] = the number of leading rows of voxels[v] that may be non-zero (both zero-filled once, by the caller).  95 % of
// the dense tensor is padding (1.6 points per voxel at config 2, P = 32) and is ALREADY zero there: the stretch then stores
// only the rows below max(kept now, row_state[v]) -- the new rows, and zeros over what the previous occupant of id v left --
// and row_state[v] <- kept.  Ids past this frame's voxel count keep their state until a later frame reaches them.
template <class Key, bool AGG4, bool RESIDENT = false>
__global__ __launch_bounds__(256) void k_emit(Key kf, int64_t npad, const uint32_t *__restrict__ firstmap,
                                              const uint32_t *__restrict__ fwpre, const uint32_t *__restrict__ bsumF,
                                              const uint4 *__restrict__ vrec, uint32_t max_voxels,
                                              const float4 *__restrict__ points4, const uint32_t *__restrict__ ranked,
                                              const float4 *__restrict__ staged, uint32_t P, int pshift /* log2 P or -1 */,
                                              int reduction, int64_t *coords, int32_t *npoints, unsigned char *pmask, float4 *agg,
                                              float4 *voxels /* NULL: no rows (reduce contract) */, int64_t *counts,
                                              int64_t *host_counts, BinnedExtras x)
{
    typedef float vec4 __attribute__((ext_vector_type(4)));
    __shared__ vec4 rowbuf_all[256 / kWave][kEmitCap];
    __shared__ uint32_t off_all[256 / kWave][kWave], base_all[256 / kWave][kWave], first_all[256 / kWave][kWave];
    __shared__ uint16_t kept_all[256 / kWave][kWave];
    __shared__ uint16_t lim_all[256 / kWave][kWave];
    __shared__ uint32_t loff_all[RESIDENT ? 256 / kWave : 1][kWave + 1];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    vec4 *rowbuf = rowbuf_all[w];
    uint32_t *sh_off = off_all[w], *sh_base = base_all[w], *sh_first = first_all[w];
    uint16_t *sh_kept = kept_all[w];
    uint16_t *sh_lim = lim_all[w];
    uint32_t *sh_loff = loff_all[RESIDENT ? w : 0];
    D3D_PHASE_DECL;
    const uint32_t tile = (uint32_t)(i / kFlagTile), ntile = (uint32_t)(npad / kFlagTile);
    uint32_t before = 0, all = 0;
    for (uint32_t t = lane; t < ntile; t += kWave) {
        const uint32_t x = bsumF[t];
        all += x;
        if (t < tile) before += x;
    }
    {
        const u64 both = wave_sum_u64(((u64)all << 32) | before);       // (each below 2^32: the halves do not carry)
        before = (uint32_t)both;
        all = (uint32_t)(both >> 32);
    }
    if (i == 0) {
        counts[D3D_COUNT_VOXELS] = (int64_t)(all < max_voxels ? all : max_voxels);
        counts[D3D_COUNT_AUX] = x.aux_value;
        // sharded voxelizer: the status bits travel with the key list (row `status_row`, negative = not a cell)
        if (x.keys_out && x.status_row >= 0) x.keys_out[x.status_row] = -1 - counts[D3D_COUNT_STATUS];
        if (host_counts) notify_host(counts, host_counts);
    }
    const uint32_t e = firstmap[i];
    const unsigned long long bal = __ballot(e != kInf);
    D3D_PHASE(2, 0);                                        // tile prefix, firstmap
    const uint32_t nfirst = (uint32_t)__popcll(bal);
    uint32_t nv = nfirst;
    const uint32_t vid0 = before + fwpre[i >> 6];
    const uint32_t r = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    if (x.vidof)                                            // every lane: the voxel that STARTS at this point index, for the map
        x.vidof[i] = (e != kInf && vid0 < max_voxels && r < max_voxels - vid0) ? vid0 + r : kNoVoxel;
    if (nv == 0 || vid0 >= max_voxels) return;             // wave-uniform
    if (nv > max_voxels - vid0) nv = max_voxels - vid0;     // voxelize.cpp:116-117: later voxels are never created
    // 1. compaction as a full permutation: the r-th first point sends {record position, own index} to lane r, the rest fill up
    const uint32_t dst = (e != kInf ? r : nfirst + ((uint32_t)lane - r)) << 2;
    const uint32_t el = (uint32_t)__builtin_amdgcn_ds_permute((int)dst, (int)e);
    const uint32_t il = (uint32_t)__builtin_amdgcn_ds_permute((int)dst, (int)(uint32_t)i);
    const bool mine = (uint32_t)lane < nv;
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    if (mine) {
        // packed entries (round 5): count and segment travel in the entry, the record only for 255 points and more
        const bool has_rec = x.fm_packed ? (el >> kFmShift) == kFmRecord : el != kSingleVoxel;
        if (has_rec) rec = vrec[x.fm_packed ? (el & kFmMask) : el];   // the one random access per such voxel
        else {                                              // no record: the voxel's cell from its first point itself
            const float4 p0 = points4[il];                  // (ascending indices inside the wavefront's 1 KiB window)
            const float v3[3] = {p0.x, p0.y, p0.z};
            u64 key = 0;
            uint32_t st = 0;
            (void)kf.make(v3, key, st);                     // the same arithmetic on the same floats as k_bin_count: the same cell
            rec = make_uint4((uint32_t)key, (uint32_t)(key >> 32), x.fm_packed ? (el & kFmMask) : 0u, x.fm_packed ? (el >> kFmShift) : 1u);
        }
    }
    // (round 5) With packed entries the rows a voxel keeps are known NOW, before its record / first row / ranked rows have
    // arrived: the 128-byte lines of the stretch that hold no row at all -- three of four at config 2 -- are stored while
    // those loads are in flight; the lines with rows follow below.  Whole lines only (P * 16 and the tensor 128-byte
    // aligned), every line stored exactly once.
    bool early = false;
    vec4 *out = reinterpret_cast<vec4 *>(voxels) + (int64_t)vid0 * P;
    const vec4 zero = {0.f, 0.f, 0.f, 0.f};
    if constexpr (!RESIDENT) {
        if (x.early_zero && voxels && (P & 7u) == 0 && (reinterpret_cast<uintptr_t>(voxels) & 127) == 0) {
            const uint32_t c8 = el >> kFmShift;
            const bool unsure = mine && c8 == kFmRecord && P >= kFmRecord;      // (the count itself is in the record)
            if (!__ballot(unsure)) {
                early = true;
                const uint32_t kept_e = mine ? (c8 < P ? c8 : P) : 0u;
                sh_lim[lane] = (uint16_t)((kept_e + 7u) & ~7u);
                wave_lds_fence();
                const uint32_t qa = nv * P;
                for (uint32_t q0 = 0; q0 < qa; q0 += 4 * kWave) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t q = q0 + u * kWave + lane;
                        if (q < qa) {
                            const uint32_t j = pshift >= 0 ? (q >> pshift) : q / P;
                            if (q - j * P >= sh_lim[j]) __builtin_nontemporal_store(zero, &out[q]);
                        }
                    }
                }
            }
        }
    }
#ifdef D3D_PHASE_CLOCKS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    D3D_PHASE(2, 1);                                        // records / single points arrived
    const uint32_t base = rec.z, cnt = rec.w;
    const uint32_t kept = cnt < P ? cnt : P;                // 0 for the lanes past nv
    const uint32_t incl = wave_incl_scan_u32(kept);         // rows before this voxel in the wavefront's flat row list
    const uint32_t off = incl - kept;
    sh_off[lane] = off; sh_base[lane] = base; sh_first[lane] = il; sh_kept[lane] = (uint16_t)kept;
    uint32_t loff = 0;
    if (RESIDENT) {                                         // rows to store: the new ones, and zeros over the previous occupant's
        uint32_t lim = 0;
        if (mine) {
            const uint32_t prev = x.row_state[(int64_t)vid0 + lane];
            lim = prev > kept ? prev : kept;
            if (lim > P) lim = P;
            if (prev != kept) x.row_state[(int64_t)vid0 + lane] = (uint16_t)kept;
        }
        sh_lim[lane] = (uint16_t)lim;
        loff = wave_incl_scan_u32(lim) - lim;               // rows to store before this voxel, over the wavefront
        sh_loff[lane] = loff;
    }
    wave_lds_fence();
    const bool is_sum = reduction == D3D_REDUCE_MEAN || reduction == kReduceSum;
    float a0, a1, a2, a3;
    a0 = a1 = a2 = a3 = is_sum ? 0.0f : (reduction == D3D_REDUCE_MAX ? -INFINITY : INFINITY);
    uint32_t ja = 0;
    while (ja < nv) {                                       // wave-uniform: one batch unless the rows exceed the buffer
        const uint32_t oa = (uint32_t)__shfl((int)off, (int)ja, kWave);
        const bool fits = (uint32_t)lane >= ja && mine && incl - oa <= (uint32_t)kEmitCap;
        const unsigned long long nf = ~(__ballot(fits) >> ja);
        const uint32_t jb = ja + (nf ? (uint32_t)__ffsll((long long)nf) - 1u : (uint32_t)kWave - ja);     // >= ja + 1: kept <= P <= kEmitCap
        const uint32_t rows = (uint32_t)__shfl((int)incl, (int)jb - 1, kWave) - oa;
        // 2. the batch's rows, flat and lane-parallel
        for (uint32_t t0 = 0; t0 < rows; t0 += kWave) {
            const uint32_t t = t0 + lane;
            if (t < rows) {
                uint32_t lo = ja, hi = jb;                  // largest j in [ja, jb) with off[j] - oa <= t
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (sh_off[mid] - oa <= t) lo = mid; else hi = mid;
                }
                const uint32_t k = t - (sh_off[lo] - oa);
                const uint32_t idx = k == 0 ? sh_first[lo] : ranked[sh_base[lo] + k];
                rowbuf[t] = *reinterpret_cast<const vec4 *>(&points4[idx]);
            }
        }
#ifdef D3D_PHASE_CLOCKS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        D3D_PHASE(2, 2);                                    // rows gathered
        wave_lds_fence();
        // 3. reductions in point order, one lane per voxel
        if (AGG4 && (uint32_t)lane >= ja && (uint32_t)lane < jb && cnt <= P) {
            const vec4 *rw = rowbuf + (off - oa);
            for (uint32_t k = 0; k < kept; k++) {
                const vec4 x = rw[k];
                if (is_sum) { a0 += x.x; a1 += x.y; a2 += x.z; a3 += x.w; }
                else if (reduction == D3D_REDUCE_MAX) {      // std::max(acc, x) = acc < x ? x : acc
                    a0 = a0 < x.x ? x.x : a0; a1 = a1 < x.y ? x.y : a1; a2 = a2 < x.z ? x.z : a2; a3 = a3 < x.w ? x.w : a3;
                } else {
                    a0 = x.x < a0 ? x.x : a0; a1 = x.y < a1 ? x.y : a1; a2 = x.z < a2 ? x.z : a2; a3 = x.w < a3 ? x.w : a3;
                }
            }
        }
        D3D_PHASE(2, 3);                                    // reductions
        // 4. the stretch of the batch's voxels
        if (RESIDENT) {
            // the rows to store, flat over the batch's voxels (a few dozen per wavefront, not nv * P slots)
            const uint32_t l0 = (uint32_t)__shfl((int)loff, (int)ja, kWave);
            const uint32_t l1 = jb < (uint32_t)kWave ? (uint32_t)__shfl((int)loff, (int)jb, kWave)
                                                    : (uint32_t)__shfl((int)(loff + sh_lim[lane]), kWave - 1, kWave);
            for (uint32_t t0 = l0; t0 < l1; t0 += kWave) {
                const uint32_t t = t0 + lane;
                if (t < l1) {
                    uint32_t lo = ja, hi = jb;              // largest j in [ja, jb) with loff[j] <= t
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (sh_loff[mid] <= t) lo = mid; else hi = mid;
                    }
                    const uint32_t slot = t - sh_loff[lo];
                    vec4 val = zero;
                    if (slot < sh_kept[lo]) val = rowbuf[sh_off[lo] - oa + slot];
                    out[lo * P + slot] = val;
                }
            }
        }
        const uint32_t q1 = voxels && !RESIDENT ? jb * P : 0u;
        for (uint32_t q0 = ja * P; q0 < q1; q0 += 4 * kWave) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t q = q0 + u * kWave + lane;
                if (q < q1) {
                    const uint32_t j = pshift >= 0 ? (q >> pshift) : q / P;
                    const uint32_t slot = q - j * P;
                    vec4 val = zero;
                    if (slot < sh_kept[j]) val = rowbuf[sh_off[j] - oa + slot];
                    if (!early || slot < sh_lim[j]) __builtin_nontemporal_store(val, &out[q]);
                }
            }
        }
        D3D_PHASE(2, 4);                                    // stretch stores issued
        wave_lds_fence();                                   // the next batch overwrites the buffer
        ja = jb;
    }
    // 5. per-voxel outputs, each as contiguous full-width stores over the wavefront's stretch (streamed past the caches like
    //    the rows: partially written lines left in the L2 are written back during the next launches)
    const int64_t v = (int64_t)vid0 + lane;
    {
        // coords[nv][3] i64: through LDS (the row buffer is free now), then 8-byte words 64 at a time
        long long *cbuf = reinterpret_cast<long long *>(rowbuf);       // 64 * 3 * 8 B = 1.5 KiB
        if (mine) {
            long long cc[3];
            kf.decode(((u64)rec.y << 32) | rec.x, cc);
            cbuf[lane * 3 + 0] = cc[0]; cbuf[lane * 3 + 1] = cc[1]; cbuf[lane * 3 + 2] = cc[2];
            __builtin_nontemporal_store((int32_t)cnt, &npoints[v]);
            // reduce contract (sharded voxelizer): linear cell key, global index of the first point, segment base
            if (x.keys_out) x.keys_out[v] = (int64_t)(((u64)rec.y << 32) | rec.x);
            if (x.first_out) x.first_out[v] = x.index_offset + (int64_t)il;
            if (x.voff) x.voff[v] = base;
        }
        wave_lds_fence();
        if (coords) {
            long long *cdst = reinterpret_cast<long long *>(coords) + (int64_t)vid0 * 3;
            for (uint32_t t = lane; t < nv * 3; t += kWave) __builtin_nontemporal_store(cbuf[t], &cdst[t]);
        }
        if (pmask) {                                         // P % 16 == 0, 16-byte aligned (host-checked): 16-byte pieces
            const uint32_t per = P >> 4, total = nv * per;
            typedef uint32_t uvec4 __attribute__((ext_vector_type(4)));
            uvec4 *pdst = reinterpret_cast<uvec4 *>(pmask + (int64_t)vid0 * P);
            for (uint32_t t = lane; t < total; t += kWave) {
                const uint32_t j = t / per, k0 = (t - j * per) << 4, kj = sh_kept[j];
                uvec4 w4;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t b = 0;
#pragma unroll
                    for (int x = 0; x < 4; x++) b |= ((k0 + q * 4 + x) < kj ? 1u : 0u) << (8 * x);
                    w4[q] = b;
                }
                __builtin_nontemporal_store(w4, &pdst[t]);
            }
        }
    }
    if (AGG4 && mine) {
        vec4 res;
        if (cnt > P) res = *reinterpret_cast<const vec4 *>(&staged[base + P]);   // fp64 reduction of k_bucket_index (voxelize.cpp:137-157)
        else {
            if (reduction == D3D_REDUCE_MEAN) {              // voxelize.cpp:164 (float / int)
                const float d = (float)(int32_t)cnt;
                a0 = a0 / d; a1 = a1 / d; a2 = a2 / d; a3 = a3 / d;
            }
            res.x = a0; res.y = a1; res.z = a2; res.w = a3;
        }
        __builtin_nontemporal_store(res, reinterpret_cast<vec4 *>(&agg[v]));
    }
    D3D_PHASE(2, 5);                                        // per-voxel outputs issued
}

// Round 6 -- the dense contract's output launch in TWO ROLES (packed entries, rows of 4 floats).  k_emit is a
// latency chain (entries -> first rows -> ranked indices -> rows) AND a 345 MB store stream in every wavefront, and the two did not
// overlap: with the stretch stores switched off it takes 32 us, the stores by themselves 49-54 us, together 81-86 us
// (profiles/r06_emit_parts.txt) -- the wavefronts of a launch walk their chains at the same time, then store at the same time.
// Here two workgroups take each 256 points.  The EVEN one is the chain: rows gathered into LDS, reductions, the voxels' leading
// 128-byte lines (rows kept rounded up to 8) and the aggregates -- no cell arithmetic, its first dependent load is already a row or
// a ranked index.  The ODD one needs nothing but the entries to know which lines of the stretch hold no row -- three of four at
// config 2 -- and streams them, after it has written coords / voxel_npoints / voxel_pmask (one gather: the voxels' first points,
// for the cells).  Voxels of 255 points and more (count and cell in a record) belong to the even role entirely.  Both roles derive a voxel's split point `lim` from its entry alone; every byte has exactly one writer.
// voxels[0 .. prefilled) x P rows were zero-filled under the index launches (ZeroFill): the odd role skips those.
template <class Key, bool AGG4, int WG = 256>
__global__ __launch_bounds__(WG) void k_emit_split(Key kf, int64_t npad, const uint32_t *__restrict__ firstmap,
                                                    const uint32_t *__restrict__ fwpre, const uint32_t *__restrict__ bsumF,
                                                    const uint4 *__restrict__ vrec, uint32_t max_voxels,
                                                    const float4 *__restrict__ points4, const uint32_t *__restrict__ ranked,
                                                    const float4 *__restrict__ staged, uint32_t P, int pshift /* log2 P or -1 */,
                                                    int reduction, int64_t *coords, int32_t *npoints, unsigned char *pmask /* or NULL */,
                                                    float4 *agg, float4 *voxels, int64_t *counts, int64_t *host_counts,
                                                    uint32_t prefilled, int64_t aux_value,
                                                    uint32_t dbg /* TUNE build: timing experiments with WRONG outputs; else 0 */)
{
    typedef float vec4 __attribute__((ext_vector_type(4)));
    typedef uint32_t uvec4 __attribute__((ext_vector_type(4)));
    __shared__ vec4 rowbuf_all[256 / kWave][kEmitCap];
    __shared__ uint32_t off_all[256 / kWave][kWave], base_all[256 / kWave][kWave], first_all[256 / kWave][kWave];
    __shared__ uint32_t loff_all[256 / kWave][kWave];
    __shared__ uint16_t kept_all[WG / kWave][kWave], lim_all[WG / kWave][kWave];
    __shared__ long long zbuf_all[WG == 512 ? 4 : 1][WG == 512 ? 3 * kWave : 1];
    // WG == 256: two workgroups per 256 points, the role by the workgroup's parity; 512: one workgroup, wavefronts 4 .. 7 the second role
    const bool zrole = WG == 512 ? threadIdx.x >= 256u : (blockIdx.x & 1u) != 0;
    const int64_t i = (int64_t)(WG == 512 ? blockIdx.x : blockIdx.x >> 1) * 256 + (threadIdx.x & 255u);
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    vec4 *rowbuf = rowbuf_all[w & 3];
    uint32_t *sh_off = off_all[w & 3], *sh_base = base_all[w & 3], *sh_first = first_all[w & 3], *sh_loff = loff_all[w & 3];
    uint16_t *sh_kept = kept_all[w], *sh_lim = lim_all[w];
    const uint32_t tile = (uint32_t)(i / kFlagTile), ntile = (uint32_t)(npad / kFlagTile);
    uint32_t before = 0, all = 0;
    for (uint32_t t = lane; t < ntile; t += kWave) {
        const uint32_t x = bsumF[t];
        all += x;
        if (t < tile) before += x;
    }
    {
        const u64 both = wave_sum_u64(((u64)all << 32) | before);
        before = (uint32_t)both;
        all = (uint32_t)(both >> 32);
    }
    if (i == 0 && !zrole) {
        counts[D3D_COUNT_VOXELS] = (int64_t)(all < max_voxels ? all : max_voxels);
        counts[D3D_COUNT_AUX] = aux_value;
        if (host_counts) notify_host(counts, host_counts);
    }
    const uint32_t e = firstmap[i];
    const unsigned long long bal = __ballot(e != kInf);
    const uint32_t nfirst = (uint32_t)__popcll(bal);
    uint32_t nv = nfirst;
    const uint32_t vid0 = before + fwpre[i >> 6];
    const uint32_t r = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    if (nv == 0 || vid0 >= max_voxels) return;             // wave-uniform
    if (nv > max_voxels - vid0) nv = max_voxels - vid0;     // voxelize.cpp:116-117: later voxels are never created
    const uint32_t dst = (e != kInf ? r : nfirst + ((uint32_t)lane - r)) << 2;
    const uint32_t el = (uint32_t)__builtin_amdgcn_ds_permute((int)dst, (int)e);
    const uint32_t il = (uint32_t)__builtin_amdgcn_ds_permute((int)dst, (int)(uint32_t)i);
    const bool mine = (uint32_t)lane < nv;
    const uint32_t c8 = el >> kFmShift, seg = el & kFmMask;
    const bool isrec = c8 == kFmRecord;
    const unsigned long long recmask = __ballot(mine && isrec);
    const int64_t v = (int64_t)vid0 + lane;
    vec4 *out = reinterpret_cast<vec4 *>(voxels) + (int64_t)vid0 * P;
    const vec4 zero = {0.f, 0.f, 0.f, 0.f};
    if (zrole) {
        if (dbg & 8u) return;                               // no second role at all
        float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mine && !isrec && !(dbg & 2u)) p0 = points4[il];               // (ascending indices inside the wavefront's 1 KiB window)
        {
            const uint32_t kept_e = c8 < P ? c8 : P, l8 = (kept_e + 7u) & ~7u;
            sh_lim[lane] = (uint16_t)(!mine ? 0u : (isrec || l8 > P) ? P : l8);
            sh_kept[lane] = (uint16_t)(mine ? kept_e : 0u);
        }
        // the per-voxel outputs that need no row, FIRST: the wavefront then ends on its zero lines without ever waiting for a
        // store (behind the store loop the compiler can only wait for everything in flight; zeros first: 62 -> 56 us with the
        // filled range below, profiles/r06_ab_split.txt).  Cell: the same arithmetic on the same floats as k_tile_sort.
        long long *cbuf = WG == 512 ? zbuf_all[w & 3] : reinterpret_cast<long long *>(rowbuf);       // 64 * 3 * 8 B = 1.5 KiB
        if (mine && !isrec && !(dbg & 2u)) {
            const float v3[3] = {p0.x, p0.y, p0.z};
            u64 key = 0;
            uint32_t st = 0;
            (void)kf.make(v3, key, st);
            long long cc[3];
            kf.decode(key, cc);
            cbuf[lane * 3 + 0] = cc[0]; cbuf[lane * 3 + 1] = cc[1]; cbuf[lane * 3 + 2] = cc[2];
            __builtin_nontemporal_store((int32_t)c8, &npoints[v]);
        }
        wave_lds_fence();
        long long *cdst = reinterpret_cast<long long *>(coords) + (int64_t)vid0 * 3;
        for (uint32_t t = lane; t < ((dbg & 2u) ? 0u : nv * 3); t += kWave)
            if (!((recmask >> (t / 3u)) & 1ull)) __builtin_nontemporal_store(cbuf[t], &cdst[t]);
        if (pmask && !(dbg & 2u)) {                                         // P % 16 == 0, 16-byte aligned (host-checked): 16-byte pieces
            const uint32_t per = P >> 4, total = nv * per;
            uvec4 *pdst = reinterpret_cast<uvec4 *>(pmask + (int64_t)vid0 * P);
            for (uint32_t t = lane; t < total; t += kWave) {
                const uint32_t j = t / per, k0 = (t - j * per) << 4, kj = sh_kept[j];
                if ((recmask >> j) & 1ull) continue;
                uvec4 w4;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t b = 0;
#pragma unroll
                    for (int x = 0; x < 4; x++) b |= ((k0 + q * 4 + x) < kj ? 1u : 0u) << (8 * x);
                    w4[q] = b;
                }
                __builtin_nontemporal_store(w4, &pdst[t]);
            }
        }
        // rows [lim, P) of the voxels past the range filled under the index launches
        const uint32_t npre = prefilled <= vid0 ? 0u : (prefilled - vid0 < nv ? prefilled - vid0 : nv);
        const uint32_t qa = (dbg & 1u) ? 0u : nv * P;
        for (uint32_t q0 = npre * P; q0 < qa; q0 += 4 * kWave) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t q = q0 + u * kWave + lane;
                if (q < qa) {
                    const uint32_t j = pshift >= 0 ? (q >> pshift) : q / P;
                    if (q - j * P >= sh_lim[j]) __builtin_nontemporal_store(zero, &out[q]);
                }
            }
        }
        return;
    }
    if (dbg & 4u) return;                                   // no chain role at all
    // the chain
    uint32_t cnt = mine ? c8 : 0u, base = seg;
    u64 key = 0;
    if (mine && isrec) {                                    // 255 points and more: count, segment and cell in the record
        const uint4 rec = vrec[seg];
        key = ((u64)rec.y << 32) | rec.x;
        base = rec.z;
        cnt = rec.w;
    }
    const uint32_t kept = cnt < P ? cnt : P;                // 0 for the lanes past nv
    const uint32_t incl = wave_incl_scan_u32(kept);
    const uint32_t off = incl - kept;
    uint32_t lim = (kept + 7u) & ~7u;
    if (lim > P || (mine && isrec)) lim = P;
    const uint32_t lincl = wave_incl_scan_u32(lim);
    const uint32_t loff = lincl - lim;                      // rows to store before this voxel, over the wavefront
    sh_off[lane] = off; sh_base[lane] = base; sh_first[lane] = il; sh_kept[lane] = (uint16_t)kept;
    sh_lim[lane] = (uint16_t)lim; sh_loff[lane] = loff;
    wave_lds_fence();
    const bool is_sum = reduction == D3D_REDUCE_MEAN || reduction == kReduceSum;
    float a0, a1, a2, a3;
    a0 = a1 = a2 = a3 = is_sum ? 0.0f : (reduction == D3D_REDUCE_MAX ? -INFINITY : INFINITY);
    uint32_t ja = 0;
    while (ja < nv) {                                       // wave-uniform: one batch unless the rows exceed the buffer
        const uint32_t oa = (uint32_t)__shfl((int)off, (int)ja, kWave);
        const bool fits = (uint32_t)lane >= ja && mine && incl - oa <= (uint32_t)kEmitCap;
        const unsigned long long nf = ~(__ballot(fits) >> ja);
        const uint32_t jb = ja + (nf ? (uint32_t)__ffsll((long long)nf) - 1u : (uint32_t)kWave - ja);     // >= ja + 1: kept <= P <= kEmitCap
        const uint32_t rows = (uint32_t)__shfl((int)incl, (int)jb - 1, kWave) - oa;
        for (uint32_t t0 = 0; t0 < rows; t0 += kWave) {     // the batch's rows, flat and lane-parallel
            const uint32_t t = t0 + lane;
            if (t < rows) {
                uint32_t lo = ja, hi = jb;                  // largest j in [ja, jb) with off[j] - oa <= t
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (sh_off[mid] - oa <= t) lo = mid; else hi = mid;
                }
                const uint32_t k = t - (sh_off[lo] - oa);
                if (dbg & 16u) continue;                    // no row gather
                const uint32_t idx = k == 0 ? sh_first[lo] : ranked[sh_base[lo] + k];
                rowbuf[t] = *reinterpret_cast<const vec4 *>(&points4[idx]);
            }
        }
        wave_lds_fence();
        if (AGG4 && !(dbg & 32u) && (uint32_t)lane >= ja && (uint32_t)lane < jb && cnt <= P) {     // reductions in point order, one lane per voxel
            const vec4 *rw = rowbuf + (off - oa);
            for (uint32_t k = 0; k < kept; k++) {
                const vec4 x = rw[k];
                if (is_sum) { a0 += x.x; a1 += x.y; a2 += x.z; a3 += x.w; }
                else if (reduction == D3D_REDUCE_MAX) {      // std::max(acc, x) = acc < x ? x : acc
                    a0 = a0 < x.x ? x.x : a0; a1 = a1 < x.y ? x.y : a1; a2 = a2 < x.z ? x.z : a2; a3 = a3 < x.w ? x.w : a3;
                } else {
                    a0 = x.x < a0 ? x.x : a0; a1 = x.y < a1 ? x.y : a1; a2 = x.z < a2 ? x.z : a2; a3 = x.w < a3 ? x.w : a3;
                }
            }
        }
        // rows [0, lim) of the batch's voxels, flat: eight lanes per 128-byte line
        const uint32_t l0 = (uint32_t)__shfl((int)loff, (int)ja, kWave);
        const uint32_t l1 = (uint32_t)__shfl((int)lincl, (int)jb - 1, kWave);
        for (uint32_t t0 = l0; t0 < ((dbg & 64u) ? l0 : l1); t0 += kWave) {     // (64: no row lines)
            const uint32_t t = t0 + lane;
            if (t < l1) {
                uint32_t lo = ja, hi = jb;                  // largest j in [ja, jb) with loff[j] <= t
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (sh_loff[mid] <= t) lo = mid; else hi = mid;
                }
                const uint32_t slot = t - sh_loff[lo];
                vec4 val = zero;
                if (slot < sh_kept[lo]) val = rowbuf[sh_off[lo] - oa + slot];
                __builtin_nontemporal_store(val, &out[lo * P + slot]);
            }
        }
        wave_lds_fence();                                   // the next batch overwrites the buffer
        ja = jb;
    }
    if (AGG4 && mine && !(dbg & 128u)) {
        vec4 res;
        if (cnt > P) res = *reinterpret_cast<const vec4 *>(&staged[base + P]);   // fp64 reduction of k_bucket_index (voxelize.cpp:137-157)
        else {
            if (reduction == D3D_REDUCE_MEAN) {              // voxelize.cpp:164 (float / int)
                const float d = (float)(int32_t)cnt;
                a0 = a0 / d; a1 = a1 / d; a2 = a2 / d; a3 = a3 / d;
            }
            res.x = a0; res.y = a1; res.z = a2; res.w = a3;
        }
        __builtin_nontemporal_store(res, reinterpret_cast<vec4 *>(&agg[v]));
    }
    if (mine && isrec) {                                    // the record voxels' own per-voxel outputs (a handful per frame)
        long long cc[3];
        kf.decode(key, cc);
        coords[v * 3 + 0] = cc[0]; coords[v * 3 + 1] = cc[1]; coords[v * 3 + 2] = cc[2];
        npoints[v] = (int32_t)cnt;
        if (pmask)
            for (uint32_t k = 0; k < P; k++) pmask[v * (int64_t)P + k] = k < kept ? 1 : 0;
    }
}

// aggregates of the voxels with MORE than max_points points, all channels of a voxel at once (k_emit_c did the others and
// listed these): one wavefront per listed voxel walks its arrival-ordered index list, 64 rows per step, every row read once.
// Sums in fp64 (see k_aggregate).
__global__ __launch_bounds__(256) void k_aggregate_overflow(const float *__restrict__ points, int c,
                                                            const uint32_t *__restrict__ ov_list, const uint32_t *__restrict__ ov_count,
                                                            const int32_t *__restrict__ npoints, const uint32_t *__restrict__ voff,
                                                            const uint32_t *__restrict__ unsorted, int reduction, float *agg)
{
    constexpr int kMaxC = 8;
    const uint32_t total = *ov_count;
    const int lane = threadIdx.x & (kWave - 1);
    const bool is_sum = reduction == D3D_REDUCE_MEAN || reduction == kReduceSum;
    const uint32_t nwaves = gridDim.x * (blockDim.x / kWave);
    for (uint32_t e = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6); e < total; e += nwaves) {
        const int64_t vv = ov_list[e];
        const uint32_t cc = (uint32_t)npoints[vv];
        const uint32_t *seg = unsorted + voff[vv];
        double sum[kMaxC];
        float ext[kMaxC];
#pragma unroll
        for (int d = 0; d < kMaxC; d++) { sum[d] = 0.0; ext[d] = reduction == D3D_REDUCE_MAX ? -INFINITY : INFINITY; }
        // four steps of 64 rows at a time: their four indices, then their rows, are requested together -- a voxel of 400 points
        // was a chain of 14 dependent round trips (index, row, index, row ...: 19 us for config 2's 3 k overflow voxels, the
        // longest one deciding), now 4.  A lane still adds ITS rows in ascending order: the same sums.
        for (uint32_t k0 = lane; k0 < cc; k0 += 4 * kWave) {
            uint32_t idx[4];
#pragma unroll
            for (int u = 0; u < 4; u++) idx[u] = k0 + u * kWave < cc ? seg[k0 + u * kWave] : 0u;
            float x[4][kMaxC];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float *row = points + (int64_t)idx[u] * c;
#pragma unroll
                for (int d = 0; d < kMaxC; d++)
                    if (d < c) x[u][d] = row[d];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (k0 + u * kWave >= cc) break;
#pragma unroll
                for (int d = 0; d < kMaxC; d++)
                    if (d < c) {
                        if (is_sum) sum[d] += (double)x[u][d];
                        else if (reduction == D3D_REDUCE_MAX) ext[d] = ext[d] < x[u][d] ? x[u][d] : ext[d];
                        else ext[d] = x[u][d] < ext[d] ? x[u][d] : ext[d];
                    }
            }
        }
#pragma unroll
        for (int d = 0; d < kMaxC; d++) {
            if (d >= c) break;
#pragma unroll
            for (int o = kWave / 2; o > 0; o >>= 1) {
                const double s2 = __shfl_xor(sum[d], o, kWave);
                const float e2 = __shfl_xor(ext[d], o, kWave);
                sum[d] += s2;
                if (reduction == D3D_REDUCE_MAX) ext[d] = ext[d] < e2 ? e2 : ext[d];
                else ext[d] = e2 < ext[d] ? e2 : ext[d];
            }
            if (lane == 0) {
                float r;
                if (reduction == D3D_REDUCE_MEAN) r = (float)sum[d] / (float)(int32_t)cc;      // (as k_aggregate)
                else if (is_sum) r = (float)sum[d];
                else r = ext[d];
                agg[vv * c + d] = r;
            }
        }
    }
}

// The same for rows of C != 4 floats (C = 3, 5 .. 8: x, y, z + up to five features): the row buffer holds C floats per row,
// the stretch of a wavefront's voxels is still ONE contiguous run of nv * P * C floats, written 16 bytes per lane -- a piece
// may straddle two rows, so its four floats are looked up one by one (P * C is a multiple of 4: every voxel, hence every
// batch, starts on a 16-byte boundary).  The index came from k_bucket_index<LISTS>: ranked point indices per voxel, all ranks.
// Overflow voxels (more than P points) get their reduction from k_aggregate afterwards (one wavefront per (voxel, channel)
// over the arrival-ordered list, fp64), which needs the segment base of every voxel: voff.
// Replaces k_meta_first + k_fill_generic_lds + k_aggregate (config 2 with a fifth column: 34 + 109 + 60 us).
template <class Key, int C>
__global__ __launch_bounds__(256) void k_emit_c(Key kf, int64_t npad, const uint32_t *__restrict__ firstmap,
                                                const uint32_t *__restrict__ fwpre, const uint32_t *__restrict__ bsumF,
                                                const uint4 *__restrict__ vrec, uint32_t max_voxels,
                                                const float *__restrict__ points, const uint32_t *__restrict__ ranked,
                                                uint32_t P, int pshift /* log2 P or -1 */, int reduction, int64_t *coords,
                                                int32_t *npoints, unsigned char *pmask, float *agg, uint32_t *voff, float *voxels,
                                                int64_t *counts, int64_t *host_counts, uint32_t *ov_list, uint32_t *ov_count,
                                                uint16_t *row_state /* resident output (see k_emit<.., RESIDENT>) or NULL */)
{
    typedef float vec4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float rowbuf_all[256 / kWave][kEmitCap * C];
    __shared__ uint32_t off_all[256 / kWave][kWave], base_all[256 / kWave][kWave], first_all[256 / kWave][kWave];
    __shared__ uint16_t kept_all[256 / kWave][kWave], lim_all[256 / kWave][kWave];
    __shared__ uint32_t loff_all[256 / kWave][kWave];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    float *rowbuf = rowbuf_all[w];
    uint32_t *sh_off = off_all[w], *sh_base = base_all[w], *sh_first = first_all[w];
    uint16_t *sh_kept = kept_all[w];
    const uint32_t tile = (uint32_t)(i / kFlagTile), ntile = (uint32_t)(npad / kFlagTile);
    uint32_t before = 0, all = 0;
    for (uint32_t t = lane; t < ntile; t += kWave) {
        const uint32_t x = bsumF[t];
        all += x;
        if (t < tile) before += x;
    }
    {
        const u64 both = wave_sum_u64(((u64)all << 32) | before);       // (each below 2^32: the halves do not carry)
        before = (uint32_t)both;
        all = (uint32_t)(both >> 32);
    }
    if (i == 0) {
        counts[D3D_COUNT_VOXELS] = (int64_t)(all < max_voxels ? all : max_voxels);
        counts[D3D_COUNT_AUX] = 0;
        if (host_counts) notify_host(counts, host_counts);
    }
    const uint32_t e = firstmap[i];
    const unsigned long long bal = __ballot(e != kInf);
    const uint32_t nfirst = (uint32_t)__popcll(bal);
    uint32_t nv = nfirst;
    const uint32_t vid0 = before + fwpre[i >> 6];
    if (nv == 0 || vid0 >= max_voxels) return;             // wave-uniform
    if (nv > max_voxels - vid0) nv = max_voxels - vid0;     // voxelize.cpp:116-117: later voxels are never created
    const uint32_t r = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    const uint32_t dst = (e != kInf ? r : nfirst + ((uint32_t)lane - r)) << 2;
    const uint32_t el = (uint32_t)__builtin_amdgcn_ds_permute((int)dst, (int)e);
    const uint32_t il = (uint32_t)__builtin_amdgcn_ds_permute((int)dst, (int)(uint32_t)i);
    const bool mine = (uint32_t)lane < nv;
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    if (mine) rec = vrec[el];
    const uint32_t base = rec.z, cnt = rec.w;
    const uint32_t kept = cnt < P ? cnt : P;                // 0 for the lanes past nv
    const uint32_t incl = wave_incl_scan_u32(kept);
    const uint32_t off = incl - kept;
    sh_off[lane] = off; sh_base[lane] = base; sh_first[lane] = il; sh_kept[lane] = (uint16_t)kept;
    uint16_t *sh_lim = lim_all[w];
    uint32_t loff = 0, lpieces = 0;
    if (row_state) {                                        // rows to store: the new ones, and zeros over the previous occupant's
        uint32_t lim = 0;
        if (mine) {
            const uint32_t prev = row_state[(int64_t)vid0 + lane];
            lim = prev > kept ? prev : kept;
            if (lim > P) lim = P;
            if (prev != kept) row_state[(int64_t)vid0 + lane] = (uint16_t)kept;
        }
        sh_lim[lane] = (uint16_t)lim;
        const uint32_t pieces = (lim * (uint32_t)C + 3u) >> 2;         // 16-byte pieces of the voxel that reach into rows in use
        loff = wave_incl_scan_u32(pieces) - pieces;
        loff_all[w][lane] = loff;
        lpieces = pieces;
    }
    wave_lds_fence();
    const bool is_sum = reduction == D3D_REDUCE_MEAN || reduction == kReduceSum;
    float acc[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) acc[ch] = is_sum ? 0.0f : (reduction == D3D_REDUCE_MAX ? -INFINITY : INFINITY);
    vec4 *out = reinterpret_cast<vec4 *>(voxels + (int64_t)vid0 * P * C);      // 16-byte aligned: P * C % 4 == 0 (host-checked)
    const uint32_t PC = P * (uint32_t)C;
    uint32_t ja = 0;
    while (ja < nv) {                                       // wave-uniform: one batch unless the rows exceed the buffer
        const uint32_t oa = (uint32_t)__shfl((int)off, (int)ja, kWave);
        const bool fits = (uint32_t)lane >= ja && mine && incl - oa <= (uint32_t)kEmitCap;
        const unsigned long long nf = ~(__ballot(fits) >> ja);
        const uint32_t jb = ja + (nf ? (uint32_t)__ffsll((long long)nf) - 1u : (uint32_t)kWave - ja);
        const uint32_t rows = (uint32_t)__shfl((int)incl, (int)jb - 1, kWave) - oa;
        for (uint32_t t0 = 0; t0 < rows; t0 += kWave) {
            const uint32_t t = t0 + lane;
            if (t < rows) {
                uint32_t lo = ja, hi = jb;                  // largest j in [ja, jb) with off[j] - oa <= t
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (sh_off[mid] - oa <= t) lo = mid; else hi = mid;
                }
                const uint32_t k = t - (sh_off[lo] - oa);
                const uint32_t idx = k == 0 ? sh_first[lo] : ranked[sh_base[lo] + k];
                const float *src = points + (size_t)idx * C;
                float x[C];
#pragma unroll
                for (int ch = 0; ch < C; ch++) x[ch] = src[ch];
#pragma unroll
                for (int ch = 0; ch < C; ch++) rowbuf[t * C + ch] = x[ch];
            }
        }
        wave_lds_fence();
        if (reduction != D3D_REDUCE_NONE && (uint32_t)lane >= ja && (uint32_t)lane < jb && cnt <= P) {
            const float *rw = rowbuf + (size_t)(off - oa) * C;
            for (uint32_t k = 0; k < kept; k++) {
#pragma unroll
                for (int ch = 0; ch < C; ch++) {
                    const float x = rw[k * C + ch];
                    if (is_sum) acc[ch] += x;
                    else if (reduction == D3D_REDUCE_MAX) acc[ch] = acc[ch] < x ? x : acc[ch];     // std::max(acc, x)
                    else acc[ch] = x < acc[ch] ? x : acc[ch];
                }
            }
        }
        if (row_state) {
            // resident: only the pieces that reach into rows in use, flat over the batch's voxels
            const uint32_t *sh_loff = loff_all[w];
            const uint32_t l0 = (uint32_t)__shfl((int)loff, (int)ja, kWave);
            const uint32_t l1 = jb < (uint32_t)kWave ? (uint32_t)__shfl((int)loff, (int)jb, kWave)
                                                    : (uint32_t)__shfl((int)(loff + lpieces), kWave - 1, kWave);
            for (uint32_t t0 = l0; t0 < l1; t0 += kWave) {
                const uint32_t t = t0 + lane;
                if (t < l1) {
                    uint32_t lo = ja, hi = jb;              // largest j in [ja, jb) with loff[j] <= t
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (sh_loff[mid] <= t) lo = mid; else hi = mid;
                    }
                    const uint32_t j = lo, f0 = (t - sh_loff[j]) * 4u;          // first float of the piece inside the voxel
                    uint32_t slot = f0 / (uint32_t)C, ch = f0 - slot * (uint32_t)C;
                    const uint32_t kj = sh_kept[j], rb0 = sh_off[j] - oa;
                    vec4 val = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        val[x] = slot < kj ? rowbuf[(size_t)(rb0 + slot) * C + ch] : 0.f;
                        if (++ch == (uint32_t)C) { ch = 0; slot++; }
                    }
                    out[j * (PC / 4) + (t - sh_loff[j])] = val;
                }
            }
        }
        // the stretch of the batch's voxels, in 16-byte pieces
        const uint32_t q1 = row_state ? 0u : jb * PC / 4;
        for (uint32_t q0 = ja * PC / 4; q0 < q1; q0 += 4 * kWave) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t q = q0 + u * kWave + lane;
                if (q < q1) {
                    uint32_t row = (q * 4u) / (uint32_t)C, ch = q * 4u - row * (uint32_t)C;     // row = voxel * P + slot
                    // (a piece never straddles two voxels: every voxel starts on a multiple of 4 floats)
                    const uint32_t j = pshift >= 0 ? (row >> pshift) : row / P;
                    const uint32_t kj = sh_kept[j], rb0 = sh_off[j] - oa;
                    uint32_t slot = row - j * P;
                    vec4 val = {0.f, 0.f, 0.f, 0.f};
                    if (slot < kj) {                         // (97 % of a LiDAR frame's pieces lie behind the kept rows: zeros)
#pragma unroll
                        for (int x = 0; x < 4; x++) {
                            val[x] = slot < kj ? rowbuf[(size_t)(rb0 + slot) * C + ch] : 0.f;
                            if (++ch == (uint32_t)C) { ch = 0; slot++; }
                        }
                    }
                    __builtin_nontemporal_store(val, &out[q]);
                }
            }
        }
        wave_lds_fence();                                   // the next batch overwrites the buffer
        ja = jb;
    }
    const int64_t v = (int64_t)vid0 + lane;
    {
        long long *cbuf = reinterpret_cast<long long *>(rowbuf);       // 64 * 3 * 8 B = 1.5 KiB (kEmitCap * C * 4 >= 3 KiB)
        if (mine) {
            long long cc[3];
            kf.decode(((u64)rec.y << 32) | rec.x, cc);
            cbuf[lane * 3 + 0] = cc[0]; cbuf[lane * 3 + 1] = cc[1]; cbuf[lane * 3 + 2] = cc[2];
            __builtin_nontemporal_store((int32_t)cnt, &npoints[v]);
            if (voff) voff[v] = base;
            // voxels with more than P points: listed for k_aggregate_overflow (they sit next to the sensor = among the first
            // voxel ids: found by a scan over the voxels, a handful of wavefronts would get them all)
            if (reduction != D3D_REDUCE_NONE && cnt > P) ov_list[atomicAdd(ov_count, 1u)] = (uint32_t)v;
        }
        wave_lds_fence();
        long long *cdst = reinterpret_cast<long long *>(coords) + (int64_t)vid0 * 3;
        for (uint32_t t = lane; t < nv * 3; t += kWave) __builtin_nontemporal_store(cbuf[t], &cdst[t]);
        if (pmask) {                                         // P % 16 == 0, 16-byte aligned (host-checked): 16-byte pieces
            const uint32_t per = P >> 4, total = nv * per;
            typedef uint32_t uvec4 __attribute__((ext_vector_type(4)));
            uvec4 *pdst = reinterpret_cast<uvec4 *>(pmask + (int64_t)vid0 * P);
            for (uint32_t t = lane; t < total; t += kWave) {
                const uint32_t j = t / per, k0 = (t - j * per) << 4, kj = sh_kept[j];
                uvec4 w4;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t b = 0;
#pragma unroll
                    for (int x = 0; x < 4; x++) b |= ((k0 + q * 4 + x) < kj ? 1u : 0u) << (8 * x);
                    w4[q] = b;
                }
                __builtin_nontemporal_store(w4, &pdst[t]);
            }
        }
    }
    if (reduction != D3D_REDUCE_NONE && mine && cnt <= P) {
        const float d = reduction == D3D_REDUCE_MEAN ? (float)(int32_t)cnt : 1.0f;      // voxelize.cpp:164 (float / int)
#pragma unroll
        for (int ch = 0; ch < C; ch++) agg[v * C + ch] = reduction == D3D_REDUCE_MEAN ? acc[ch] / d : acc[ch];
    }
}

// point -> voxel id, from bucket order: k_bucket_index left the first point of every entry's voxel, k_meta_first the id of the
// voxel that starts at a point index
__global__ __launch_bounds__(256) void k_map_binned(const uint32_t *__restrict__ bucket_base, uint32_t nbins,
                                                    const uint32_t *__restrict__ precpos, const uint32_t *__restrict__ ent32,
                                                    int idx_stride, int idx_off, const uint32_t *__restrict__ vidof,
                                                    int64_t *mapping, int32_t *keepid, const unsigned char *__restrict__ trimmed,
                                                    const uint32_t *__restrict__ tileinfo /* tile-sorted entries: */, uint32_t ntiles,
                                                    int tshift)
{
    // entries partitioned by bucket: positions [0, total); tile-sorted: the first tileinfo[t] positions of every tile
    const uint32_t total = tileinfo ? ntiles << tshift : bucket_base[nbins];
    for (uint32_t p = blockIdx.x * 256 + threadIdx.x; p < total; p += gridDim.x * 256) {
        if (tileinfo && (p & ((1u << tshift) - 1)) >= (tileinfo[p >> tshift] & 0x7fffffffu)) continue;
        const uint32_t e = precpos[p];
        const uint32_t vid = e == kInf ? kNoVoxel : vidof[e];
        const uint32_t i = ent32[(size_t)p * idx_stride + idx_off];
        if (keepid) keepid[i] = (vid == kNoVoxel || (trimmed && trimmed[i])) ? -1 : (int32_t)vid;   // fused filter: kept points
        else mapping[i] = vid == kNoVoxel ? -1ll : (long long)vid;
    }
}

// ------------------------------------------------------------------ filter (direct-addressed by voxel id)
struct FilterVoxels {
    static constexpr const char *kName = "k_scan_count<FilterVoxels>", *kName2 = "k_scan_apply<FilterVoxels>";
    const int64_t *coords;
    const int32_t *npoints;
    const int32_t *order;     // null = id order; else voxel id per rank (descending count)
    const int64_t *nvox_device;   // null, or the number of valid rows (the scan then runs over an upper bound)
    long long lo[3], hi[3];
    int32_t min_points;
    uint32_t max_points;      // P for TRIM, 0xffffffff for NONE
    unsigned long long max_voxels;
    int32_t *newid;           // [nvox]
    uint32_t *coff;           // [nvox] offset of the index list of overflow voxels
    int64_t *out_coords;
    int32_t *out_npoints;

    __device__ __forceinline__ unsigned long long value(int64_t k) const
    {
        if (nvox_device && k >= *nvox_device) return 0;
        const int64_t v = order ? order[k] : k;
        const int32_t cnt = npoints[v];
        bool ok = cnt >= min_points;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            long long x = coords[v * 3 + d];
            ok = ok && x >= lo[d] && x < hi[d];
        }
        if (!ok) return 0;
        return (1ull << 32) | ((uint32_t)cnt > max_points ? (uint32_t)cnt : 0u);   // lo: list cells of overflow voxels
    }
    __device__ __forceinline__ unsigned long long value2(int64_t k) const { return value(k); }
    __device__ __forceinline__ void apply(int64_t k, unsigned long long val, unsigned long long excl) const
    {
        const int64_t v = order ? order[k] : k;
        const unsigned long long id = excl >> 32;
        if (!val || id >= max_voxels) { newid[v] = -1; return; }
        newid[v] = (int32_t)id;
        coff[v] = (uint32_t)excl;
#pragma unroll
        for (int d = 0; d < 3; d++) out_coords[id * 3 + d] = coords[v * 3 + d];
        const uint32_t cnt = (uint32_t)npoints[v];
        out_npoints[id] = (int32_t)(cnt > max_points ? max_points : cnt);
    }
};

// TRIM: the points of every kept overflow voxel (count > max_points) are listed (arrival order); a point is
// kept iff fewer than max_points indices of its voxel's list are smaller (voxelize.cpp:457-463) -- the same
// rank-by-counting as the dense contract, early exit at max_points.
__global__ __launch_bounds__(256) void k_filter_scatter(const int64_t *__restrict__ mapping, int64_t n, int64_t nvox,
                                                        const int32_t *__restrict__ npoints,
                                                        const int32_t *__restrict__ newid, const uint32_t *__restrict__ coff,
                                                        uint32_t max_points, uint32_t *fcur, uint32_t *cells, uint32_t *cellvox,
                                                        int64_t ncells)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t v = mapping[i];
    if (v < 0 || v >= nvox || newid[v] < 0) return;
    const uint32_t cnt = (uint32_t)npoints[v];
    if (cnt <= max_points) return;
    const uint32_t a = atomicAdd(&fcur[v], 1u);
    const int64_t pos = (int64_t)coff[v] + a;
    if (a < cnt && pos < ncells) { cells[pos] = (uint32_t)i; cellvox[pos] = (uint32_t)v; }
}

// one lane per list cell (dense wavefronts; the lanes of a wavefront mostly share a segment, so its reads are
// broadcasts): trimmed[point] = at least max_points indices of the voxel's list are smaller.  Doing this per POINT
// inside the compaction scan left 1-6 lanes of every wavefront walking 400-entry segments (80 us at config 2).
__global__ __launch_bounds__(256) void k_filter_rank(const int64_t *__restrict__ counts, const uint32_t *__restrict__ cells,
                                                     const uint32_t *__restrict__ cellvox, int64_t ncells,
                                                     const int32_t *__restrict__ npoints, const uint32_t *__restrict__ coff,
                                                     uint32_t max_points, unsigned char *trimmed)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncells || t >= counts[D3D_COUNT_AUX]) return;
    const uint32_t v = cellvox[t];
    if (v == 0xffffffffu) return;                      // cell of a voxel dropped by max_voxels, or never filled
    const uint32_t cnt = (uint32_t)npoints[v], me = cells[t];
    const int64_t base = coff[v];
    if (base + cnt > ncells) return;                   // inconsistent voxel_npoints: cannot trim
    const uint32_t *seg = cells + base;
    uint32_t rank = 0, k = 0;
    for (; k + 4 <= cnt && rank < max_points; k += 4)
        rank += (seg[k] < me) + (seg[k + 1] < me) + (seg[k + 2] < me) + (seg[k + 3] < me);
    for (; k < cnt && rank < max_points; k++) rank += seg[k] < me;
    if (rank >= max_points) trimmed[me] = 1;
}

struct FilterPoints {
    static constexpr const char *kName = "k_scan_count<FilterPoints>", *kName2 = "k_scan_apply<FilterPoints>";
    const float *feats;
    int c;
    const int64_t *mapping;
    int64_t nvox;
    const int32_t *npoints;
    const int32_t *newid;
    const unsigned char *trimmed;   // [n] set by k_filter_rank for the points beyond max_points of their voxel
    uint32_t max_points;      // 0xffffffff for NONE
    int32_t *keepid;          // [n] new voxel id of a kept point, -1 otherwise (written by the count pass)
    float *out_feats;
    int64_t *out_mask, *out_mapping;
    bool precomputed = false; // keepid was filled by the fused sparse index (k_map_binned)
    bool vec4 = false;        // c == 4 and 16-byte aligned rows: one float4 copy per kept point
    __device__ __forceinline__ int32_t keep(int64_t i) const
    {
        const int64_t v = mapping[i];
        if (v < 0 || v >= nvox) return -1;
        const int32_t id = newid[v];
        if (id < 0) return -1;
        const uint32_t cnt = (uint32_t)npoints[v];
        if (cnt > max_points && (max_points == 0 || trimmed[i])) return -1;
        return id;
    }
    __device__ __forceinline__ unsigned long long value(int64_t i) const
    {
        if (precomputed) return keepid[i] >= 0 ? 1ull : 0ull;
        const int32_t id = keep(i);
        keepid[i] = id;
        return id >= 0 ? 1ull : 0ull;
    }
    __device__ __forceinline__ unsigned long long value2(int64_t i) const { return keepid[i] >= 0 ? 1ull : 0ull; }
    __device__ __forceinline__ void apply(int64_t i, unsigned long long val, unsigned long long excl) const
    {
        if (!val) return;
        out_mask[excl] = i;
        out_mapping[excl] = keepid[i];
        if (vec4) reinterpret_cast<float4 *>(out_feats)[excl] = reinterpret_cast<const float4 *>(feats)[i];   // c == 4, aligned
        else
            for (int d = 0; d < c; d++) out_feats[excl * c + d] = feats[i * c + d];
    }
};

__global__ void k_fill_u32(uint32_t *p, int64_t n, uint32_t val, int64_t *counts)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < n; i += stride) p[i] = val;
    if (counts && t0 < D3D_NUM_COUNTS) counts[t0] = 0;
}



// out_coords[v, :] -= offset for the counts[VOXELS] valid rows (the two-operator form of d3d_voxelize_3d_sparse_filter)
// wide keys: a voxel's coordinates from its first point (the table compared them, nothing decodes them)
__global__ __launch_bounds__(256) void k_wide_coords(WideKey kf, const float *__restrict__ points, int c, const int64_t *__restrict__ counts,
                                                     const int64_t *__restrict__ first, const uint4 *__restrict__ vinfo,
                                                     int64_t *__restrict__ coords, int32_t *__restrict__ npoints)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= counts[D3D_COUNT_VOXELS]) return;
    const float *p = points + (size_t)first[v] * c;
    for (int d = 0; d < 3; d++) coords[v * 3 + d] = (int64_t)WideKey::coord(p[d], kf.size[d]);
    npoints[v] = (int32_t)vinfo[v].w;
}

__global__ __launch_bounds__(256) void k_sub_offset(int64_t *coords, const int64_t *__restrict__ counts, int64_t rows, long long o0,
                                                    long long o1, long long o2)
{
    const int64_t valid = counts[D3D_COUNT_VOXELS] < rows ? counts[D3D_COUNT_VOXELS] : rows;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < valid * 3; t += (int64_t)gridDim.x * 256) {
        const int d = (int)(t % 3);
        coords[t] -= d == 0 ? o0 : (d == 1 ? o1 : o2);
    }
}

// Fused sparse + filter, the compaction of the kept points in ONE launch (round 4; before: count pass, block-sum pass, apply
// pass = three launches that each re-derived the point's voxel through two dependent random reads).  pfirst[i] (left by
// k_bucket_index) names the first point of point i's voxel when the point is kept, vidof[] (k_meta_first) that voxel's
// filtered id: one coalesced and one random read per point.  Tiles of 4096 points; the tile's place in the output comes from
// decoupled look-back over the ~n / 4096 tiles of this launch (common.hpp; few enough for the ticket word), the last tile
// leaves the sizes in counts[].  Output order = point order (voxelize.cpp:441-468).
constexpr int kCompactThreads = 1024, kCompactItems = 4, kCompactTile = kCompactThreads * kCompactItems;
template <bool VEC4>
__global__ __launch_bounds__(kCompactThreads) void k_compact_kept(const float *__restrict__ feats, int c, int64_t n, int64_t npad,
                                                                  const uint32_t *__restrict__ pfirst, const uint32_t *__restrict__ vidof,
                                                                  float *__restrict__ out_feats, int64_t *__restrict__ out_mask,
                                                                  int64_t *__restrict__ out_mapping, u64 *lbstat, unsigned int *ticket,
                                                                  int64_t *counts, const int64_t *__restrict__ first_counts,
                                                                  int64_t *host /* optional: both count rows + flag, by the last tile */,
                                                                  int voxels_index /* of first_counts: where the voxel count is */)
{
    __shared__ unsigned int sid;
    __shared__ uint32_t wtot[kCompactThreads / kWave];
    __shared__ u64 sprefix;
    const unsigned int tile = lookback_ticket(ticket, &sid);
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    // wavefront w owns 256 consecutive points, row k of it = 64 consecutive points: coalesced, and (w, k, lane) is point order
    const int64_t base = (int64_t)tile * kCompactTile + (int64_t)w * (kWave * kCompactItems) + lane;
    uint32_t f[kCompactItems];
#pragma unroll
    for (int k = 0; k < kCompactItems; k++) {
        const int64_t i = base + (int64_t)k * kWave;
        f[k] = i < n ? pfirst[i] : kInf;
    }
    uint32_t id[kCompactItems];
#pragma unroll
    for (int k = 0; k < kCompactItems; k++) id[k] = (int64_t)f[k] < npad ? vidof[f[k]] : kNoVoxel;     // (kInf, or a stale word after a
                                                                                                       //  BIN_OVERFLOW: in bounds)
    uint32_t ex[kCompactItems], carry = 0;
#pragma unroll
    for (int k = 0; k < kCompactItems; k++) {
        const unsigned long long bal = __ballot(id[k] != kNoVoxel);
        ex[k] = carry + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        carry += (uint32_t)__popcll(bal);
    }
    if (lane == 0) wtot[w] = carry;
    __syncthreads();
    uint32_t total = 0, woff = 0;
#pragma unroll
    for (int k = 0; k < kCompactThreads / kWave; k++) {
        const uint32_t x = wtot[k];
        total += x;
        if (k < w) woff += x;
    }
    if (w == 0) {
        const u64 before = lookback_exclusive(lbstat, tile, (u64)total);
        if (lane == 0) sprefix = before;
    }
    __syncthreads();
    const u64 pre = sprefix + woff;
#pragma unroll
    for (int k = 0; k < kCompactItems; k++) {
        if (id[k] == kNoVoxel) continue;
        const int64_t i = base + (int64_t)k * kWave;
        const u64 e = pre + ex[k];
        out_mask[e] = i;
        out_mapping[e] = (int64_t)id[k];
        if (VEC4) reinterpret_cast<float4 *>(out_feats)[e] = reinterpret_cast<const float4 *>(feats)[i];
        else
            for (int d = 0; d < c; d++) out_feats[e * c + d] = feats[i * c + d];
    }
    if (tile == gridDim.x - 1 && threadIdx.x == 0) {        // the sizes, as the filter operator leaves them (voxels: the index's)
        counts[D3D_COUNT_POINTS] = (int64_t)(sprefix + total);
        counts[D3D_COUNT_VOXELS] = first_counts[voxels_index];
        counts[D3D_COUNT_STATUS] = 0;
        counts[D3D_COUNT_AUX] = 0;
        if (host) {
            for (int k = 0; k < D3D_NUM_COUNTS; k++) {
                host[k] = first_counts[k];
                host[D3D_NUM_COUNTS + 1 + k] = counts[k];
            }
            __threadfence_system();
            __hip_atomic_store(&host[D3D_NUM_COUNTS], (int64_t)1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Round 5: numbering, per-voxel outputs AND the compaction of the kept points of the fused sparse + filter call in ONE launch
// (k_meta_first_lb + k_compact_kept before), behind k_bucket_index<BoundKey, .., V2>.  Tiles of 4096 point indices in ticket
// order, decoupled look-back:
//   A  first points -> filtered voxel ids (the first-seen numbering of the voxels that pass, voxelize.cpp:380-403); a first point
//      writes its voxel's coords / voxel_npoints -- the cell from its own row, the count from its entry -- and then REPLACES its
//      entry by the id: firstmap[first point] = {0 : 8 | id : 24} (an entry's count field is never 0): a coalesced store;
//   B  kept points -> places in the compacted outputs.  A first point knows its id; any other kept point carries the index of
//      its voxel's first point (k_bucket_index) and reads the id there.  That point has a lower index, i.e. sits in this tile or
//      in one with an earlier ticket: its workgroup is running and publishes without waiting for anything but tiles before
//      ITS OWN -- so a reader that still finds the entry polls, and cannot wait for ever.
//   When max_voxels cannot cut the frame (the index has added up the passing voxels) every point with an entry or a handle is
//   kept whatever its voxel's id: ONE look-back carries both prefixes; otherwise a second one follows the ids.
// Against the two launches: one read of the entries and the handles instead of firstmap / vidof / pfirst twice, no vidof array.
constexpr int kFinThreads = 1024, kFinItems = 4, kFinTile = kFinThreads * kFinItems;   // (tiles of 2048 points on 512 lanes, three
                                                  // workgroups per CU: 30 -> 39 us at config 2 -- twice the tiles in the look-back)
constexpr uint32_t kVoxelCut = 0x00ffffffu;       // published id: the voxel exists but lies behind the max_voxels cut
struct SparseFin {
    u64 *stat_a, *stat_b;       // [tiles] each, cleared by k_tile_sort together with the ticket
    unsigned int *ticket;
    int64_t *host;              // 2 * D3D_NUM_COUNTS + 1 words (d3d_voxelize_3d_sparse_filter), or NULL
    const u64 *early;           // k_bucket_index's pairs {passing voxels, kept points}: complete when this launch starts
    uint32_t early_pairs;
    long long coord_sub[3];     // VoxelGenerator's coords - offset (voxel/__init__.py:103), applied in the store
};
template <bool VEC4>
__global__ __launch_bounds__(kFinThreads) void k_sparse_finish(BoundKey kf, int64_t n, uint32_t *firstmap,
                                                               const uint32_t *__restrict__ phandle, const uint4 *__restrict__ vrec,
                                                               const float *__restrict__ points, int c,
                                                               uint32_t max_voxels, uint32_t npoints_clamp, int64_t *__restrict__ out_coords,
                                                               int32_t *__restrict__ out_npoints, float *__restrict__ out_feats,
                                                               int64_t *__restrict__ out_mask, int64_t *__restrict__ out_mapping,
                                                               int64_t *sparse_counts, int64_t *counts, SparseFin lb)
{
    typedef float vec4 __attribute__((ext_vector_type(4)));
    __shared__ unsigned int sid;
    __shared__ uint32_t wtot[kFinThreads / kWave], wtot2[kFinThreads / kWave];
    __shared__ u64 sprefix;
    const unsigned int tile = lookback_ticket(lb.ticket, &sid);
    const unsigned int ntiles = gridDim.x;
    if (tile == 0 && lb.host && lb.early_pairs && threadIdx.x < kWave) {
        // both output sizes are known before any voxel has its number -- unless max_voxels cuts the frame short -- and the host,
        // which waits for them to size what it returns, gets them now (as k_meta_first_lb's first tile did)
        const bool have = threadIdx.x < lb.early_pairs;
        const u64 tv = wave_sum_u64(have ? lb.early[16 * threadIdx.x] : 0ull), tp = wave_sum_u64(have ? lb.early[16 * threadIdx.x + 1] : 0ull);
        if (threadIdx.x == 0 && tv <= (u64)max_voxels) {
            for (int k = 0; k < D3D_NUM_COUNTS; k++) {
                lb.host[k] = sparse_counts[k];
                lb.host[D3D_NUM_COUNTS + 1 + k] = 0;
            }
            lb.host[D3D_COUNT_VOXELS] = (int64_t)tv;
            lb.host[D3D_COUNT_AUX] = 0;
            lb.host[D3D_NUM_COUNTS + 1 + D3D_COUNT_VOXELS] = (int64_t)tv;
            lb.host[D3D_NUM_COUNTS + 1 + D3D_COUNT_POINTS] = (int64_t)tp;
            __threadfence_system();
            __hip_atomic_store(&lb.host[D3D_NUM_COUNTS], (int64_t)1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int64_t base = (int64_t)tile * kFinTile + (int64_t)w * (kWave * kFinItems) + lane;          // (w, row, lane) = index order
    uint32_t e[kFinItems], h[kFinItems], ex[kFinItems], carry = 0;
#pragma unroll
    for (int k = 0; k < kFinItems; k++) {
        const int64_t i = base + (int64_t)k * kWave;
        e[k] = firstmap[i];                                 // (padded to the tile, preset by k_tile_sort)
        h[k] = i < n ? phandle[i] : kInf;
    }
    // the rows of every point that may be kept -- a first point (its cell comes from its row too) or the holder of a handle --
    // are on their way while the numbering is resolved: nothing but stores is left behind the second look-back
    vec4 row[VEC4 ? kFinItems : 1];
    float px[kFinItems][3];
#pragma unroll
    for (int k = 0; k < kFinItems; k++) {
        const int64_t i = base + (int64_t)k * kWave;
        px[k][0] = px[k][1] = px[k][2] = 0.f;
        if (VEC4) {
            row[VEC4 ? k : 0] = vec4{0.f, 0.f, 0.f, 0.f};
            if (e[k] != kInf || h[k] != kInf) {
                row[VEC4 ? k : 0] = reinterpret_cast<const vec4 *>(points)[i];
                px[k][0] = row[VEC4 ? k : 0].x; px[k][1] = row[VEC4 ? k : 0].y; px[k][2] = row[VEC4 ? k : 0].z;
            }
        } else if (e[k] != kInf) {
            const float *src = points + i * c;
            px[k][0] = src[0]; px[k][1] = src[1]; px[k][2] = src[2];
        }
    }
#pragma unroll
    for (int k = 0; k < kFinItems; k++) {
        const unsigned long long bal = __ballot(e[k] != kInf);
        ex[k] = carry + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        carry += (uint32_t)__popcll(bal);
    }
    if (lane == 0) wtot[w] = carry;
    __syncthreads();
    uint32_t total = 0, woff = 0;
#pragma unroll
    for (int k = 0; k < kFinThreads / kWave; k++) {
        const uint32_t t = wtot[k];
        total += t;
        if (k < w) woff += t;
    }
    // When max_voxels cannot cut the frame -- the index added up the passing voxels: lb.early -- every point with an entry or a
    // handle is kept whatever its voxel's id turns out to be, so ONE look-back carries both prefixes {voxels : 32 | kept : 31}
    // and the handles are only read for the ids' values, not for the positions.
    bool nocut = false;
    if (lb.early_pairs) {
        u64 tv = 0;
        for (uint32_t t = lane; t < lb.early_pairs; t += kWave) tv += lb.early[16 * t];
        nocut = wave_sum_u64(tv) <= (u64)max_voxels;                    // (the same in every wavefront)
    }
    uint32_t kex[kFinItems], ktot_w = 0;
    if (nocut) {
#pragma unroll
        for (int k = 0; k < kFinItems; k++) {
            const unsigned long long bal = __ballot(e[k] != kInf || h[k] != kInf);
            kex[k] = ktot_w + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            ktot_w += (uint32_t)__popcll(bal);
        }
        if (lane == 0) wtot2[w] = ktot_w;
        __syncthreads();
    }
    uint32_t ktotal1 = 0, kwoff1 = 0;
    if (nocut) {
#pragma unroll
        for (int k = 0; k < kFinThreads / kWave; k++) {
            const uint32_t t = wtot2[k];
            ktotal1 += t;
            if (k < w) kwoff1 += t;
        }
    }
    if (w == 0) {
        const u64 before = lookback_exclusive(lb.stat_a, tile, nocut ? (((u64)total << 31) | (u64)ktotal1) : (u64)total);
        if (lane == 0) sprefix = before;
    }
    __syncthreads();
    const u64 voxels_before = nocut ? sprefix >> 31 : sprefix;
    const u64 kept_before1 = sprefix & 0x7fffffffull;
    // A: the first points' ids, published under the voxel's handle for its other points
    uint32_t id[kFinItems];
    bool cutk[kFinItems];
#pragma unroll
    for (int k = 0; k < kFinItems; k++) {
        id[k] = kNoVoxel;
        cutk[k] = false;
        if (e[k] == kInf) continue;
        const u64 vid64 = voxels_before + woff + ex[k];
        cutk[k] = vid64 >= (u64)max_voxels;                         // voxelize.cpp:396-397: later voxels are not taken
        __hip_atomic_store(&firstmap[base + (int64_t)k * kWave], cutk[k] ? kVoxelCut : (uint32_t)vid64, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);               // (ids < 2^24: the count field reads 0 = "an id")
        if (!cutk[k]) id[k] = (uint32_t)vid64;
    }
    __syncthreads();                                        // (this workgroup's own ids are out: s_waitcnt vmcnt(0) + barrier)
    // B: the other kept points read their voxel's id under its handle -- first attempt now, the per-voxel outputs meanwhile
    uint32_t got[kFinItems];
#pragma unroll
    for (int k = 0; k < kFinItems; k++) {
        got[k] = kInf;
        if (e[k] == kInf && h[k] != kInf) got[k] = __hip_atomic_load(&firstmap[h[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int k = 0; k < kFinItems; k++) {
        if (id[k] == kNoVoxel) continue;                    // (not a first point, or behind the cut)
        const uint32_t vid = id[k], c8 = e[k] >> kFmShift;
        const uint32_t cnt = c8 == kFmRecord ? vrec[e[k] & kFmMask].w : c8;
        u64 key = 0;
        uint32_t st = 0;
        (void)kf.make(px[k], key, st);                      // the same arithmetic on the same floats as k_tile_sort: the same cell
        long long cc[3];
        kf.decode(key, cc);
        out_coords[(int64_t)vid * 3 + 0] = cc[0] - lb.coord_sub[0];
        out_coords[(int64_t)vid * 3 + 1] = cc[1] - lb.coord_sub[1];
        out_coords[(int64_t)vid * 3 + 2] = cc[2] - lb.coord_sub[2];
        out_npoints[vid] = (int32_t)(cnt < npoints_clamp ? cnt : npoints_clamp);
    }
#pragma unroll
    for (int k = 0; k < kFinItems; k++) {
        if (e[k] != kInf || h[k] == kInf) continue;
        uint32_t v = got[k];
        unsigned long long waited = 0;
        // the voxel's first point sits in this tile or in one with an earlier ticket, whose workgroup is running and publishes
        // without waiting for this one (bounded by time like the look-back's poll: common.hpp, poll_or_trap)
        while ((v >> kFmShift) != 0u) {                     // (still the voxel's entry)
            poll_or_trap(waited);
            v = __hip_atomic_load(&firstmap[h[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        id[k] = v == kVoxelCut ? kNoVoxel : v;
    }
    uint32_t px2[kFinItems];
    uint32_t ktotal = 0, kwoff = 0;
    u64 kept_before = kept_before1;
    if (nocut) {                                            // (positions from the one look-back above)
#pragma unroll
        for (int k = 0; k < kFinItems; k++) px2[k] = kex[k];
        ktotal = ktotal1;
        kwoff = kwoff1;
    } else {
        carry = 0;
#pragma unroll
        for (int k = 0; k < kFinItems; k++) {
            const unsigned long long bal = __ballot(id[k] != kNoVoxel);
            px2[k] = carry + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            carry += (uint32_t)__popcll(bal);
        }
        if (lane == 0) wtot[w] = carry;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kFinThreads / kWave; k++) {
            const uint32_t t = wtot[k];
            ktotal += t;
            if (k < w) kwoff += t;
        }
        if (w == 0) {
            const u64 before = lookback_exclusive(lb.stat_b, tile, (u64)ktotal);
            if (lane == 0) sprefix = before;
        }
        __syncthreads();
        kept_before = sprefix;
    }
    const u64 pre = kept_before + kwoff;
#pragma unroll
    for (int k = 0; k < kFinItems; k++) {
        if (id[k] == kNoVoxel) continue;
        const int64_t i = base + (int64_t)k * kWave;
        const u64 o = pre + px2[k];
        out_mask[o] = i;
        out_mapping[o] = (int64_t)id[k];
        if (VEC4) reinterpret_cast<vec4 *>(out_feats)[o] = row[VEC4 ? k : 0];
        else
            for (int d = 0; d < c; d++) out_feats[o * c + d] = points[i * c + d];
    }
    if (tile == ntiles - 1 && threadIdx.x == 0) {           // the sizes, as the two operators leave them
        const u64 all = voxels_before + total;
        const int64_t nvox = (int64_t)(all < (u64)max_voxels ? all : (u64)max_voxels), kept = (int64_t)(kept_before + ktotal);
        sparse_counts[D3D_COUNT_VOXELS] = nvox;
        sparse_counts[D3D_COUNT_AUX] = 0;
        counts[D3D_COUNT_POINTS] = kept;
        counts[D3D_COUNT_VOXELS] = nvox;
        counts[D3D_COUNT_STATUS] = 0;
        counts[D3D_COUNT_AUX] = 0;
        if (lb.host && !(lb.early_pairs && all <= (u64)max_voxels)) {          // (else the first tile has told the host)
            for (int k = 0; k < D3D_NUM_COUNTS; k++) {
                lb.host[k] = sparse_counts[k];
                lb.host[D3D_NUM_COUNTS + 1 + k] = counts[k];
            }
            __threadfence_system();
            __hip_atomic_store(&lb.host[D3D_NUM_COUNTS], (int64_t)1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// DESCENDING voxel filter, fused (voxelize.cpp:404-420): the voxels that pass the filter, numbered v in first-seen order by
// k_meta_first_lb, are ranked r by a stable descending sort of their counts; the first max_voxels ranks are the result.
// Thread = rank: the voxel's per-voxel outputs move to row r, and the entry of its FIRST POINT in vidof[] -- what every
// point of the voxel looks up in k_compact_kept -- becomes r (or "no voxel" behind the cut).
__global__ __launch_bounds__(256) void k_desc_finish(const int32_t *__restrict__ order, int64_t *counts /* [VOXELS]: in V', out the cut */,
                                                     uint32_t max_voxels, const int64_t *__restrict__ tmp_coords,
                                                     const int32_t *__restrict__ tmp_npoints, const int64_t *__restrict__ first_of,
                                                     int64_t *__restrict__ out_coords, int32_t *__restrict__ out_npoints,
                                                     uint32_t *__restrict__ vidof, int64_t *cut_out)
{
    const int64_t nv = counts[D3D_COUNT_VOXELS], r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t cut = nv < (int64_t)max_voxels ? nv : (int64_t)max_voxels;
    if (r == 0) *cut_out = cut;                             // (counts[VOXELS] itself is rewritten by the launch behind this one)
    if (r >= nv) return;
    const int32_t v = order[r];
    const int64_t f = first_of[v];
    if (r < cut) {
        out_coords[r * 3] = tmp_coords[(int64_t)v * 3]; out_coords[r * 3 + 1] = tmp_coords[(int64_t)v * 3 + 1];
        out_coords[r * 3 + 2] = tmp_coords[(int64_t)v * 3 + 2];
        out_npoints[r] = tmp_npoints[v];
        vidof[f] = (uint32_t)r;
    } else vidof[f] = kNoVoxel;
}

// ------------------------------------------------------------------ D3D_VOXEL_EXACT_MEAN: the reference's fp32 running sum
// voxelize.cpp:142 adds every in-range point of a voxel to its aggregate in point order, in fp32; :164 divides by the count.
// Voxels within max_points are reduced that way by the output kernels already.  For the others this post-pass repeats it
// literally, from the operator's own outputs (so it serves every index path): a table {cell -> voxel id} of the overflow
// voxels, every point looked up and the hits compacted IN POINT ORDER (look-back over 4096-point tiles), a stable sort of the
// hits by voxel id (sort.hip), then one wavefront per voxel: 64 rows gathered at a time, added one after the other.
constexpr int kExactThreads = 1024, kExactItems = 4, kExactTile = kExactThreads * kExactItems;

__global__ __launch_bounds__(256) void k_exact_clear(u64 *tkeys, int64_t cap, int32_t *sortkey, int64_t n, u64 *lbstat, uint32_t nlb,
                                                     unsigned int *ticket)
{
    const int64_t stride = (int64_t)gridDim.x * 256, t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (int64_t t = t0; t < cap; t += stride) tkeys[t] = kEmpty;
    for (int64_t t = t0; t < n; t += stride) sortkey[t] = INT_MIN;             // sorts behind every hit (descending sort of -id)
    for (int64_t t = t0; t < nlb; t += stride) lbstat[t] = 0ull;
    if (t0 == 0) *ticket = 0u;
}

__global__ __launch_bounds__(256) void k_exact_table(DenseKey kf, const int64_t *__restrict__ coords, const int32_t *__restrict__ npoints,
                                                     const int64_t *__restrict__ counts, uint32_t P, u64 *tkeys, uint32_t *tvals, u64 mask)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= counts[D3D_COUNT_VOXELS] || (uint32_t)npoints[v] <= P) return;
    const u64 key = ((u64)coords[v * 3] * (unsigned)kf.shape[1] + (u64)coords[v * 3 + 1]) * (unsigned)kf.shape[2] + (u64)coords[v * 3 + 2];
    u64 h = mix64(key) & mask;
    for (;;) {                                              // (capacity >= 2 x the overflow voxels: a free slot exists)
        const u64 old = atomicCAS(&tkeys[h], kEmpty, key);
        if (old == kEmpty || old == key) break;
        h = (h + 1) & mask;
    }
    tvals[h] = (uint32_t)v;
}

template <bool VEC4>
__global__ __launch_bounds__(kExactThreads) void k_exact_collect(DenseKey kf, const float *__restrict__ points, int64_t n, int c,
                                                                 const u64 *__restrict__ tkeys, const uint32_t *__restrict__ tvals, u64 mask,
                                                                 int32_t *__restrict__ sortkey, uint32_t *__restrict__ oidx, u64 *lbstat,
                                                                 unsigned int *ticket)
{
    __shared__ unsigned int sid;
    __shared__ uint32_t wtot[kExactThreads / kWave];
    __shared__ u64 sprefix;
    const unsigned int tile = lookback_ticket(ticket, &sid);
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int64_t base = (int64_t)tile * kExactTile + (int64_t)w * (kWave * kExactItems) + lane;       // (w, row, lane) = point order
    uint32_t vid[kExactItems], ex[kExactItems], carry = 0;
#pragma unroll
    for (int k = 0; k < kExactItems; k++) {
        const int64_t i = base + (int64_t)k * kWave;
        vid[k] = kNoVoxel;
        if (i < n) {
            float v3[3];
            if (VEC4) { const float4 q = reinterpret_cast<const float4 *>(points)[i]; v3[0] = q.x; v3[1] = q.y; v3[2] = q.z; }
            else { const float *src = points + i * c; v3[0] = src[0]; v3[1] = src[1]; v3[2] = src[2]; }
            u64 key;
            uint32_t st = 0;
            if (kf.make(v3, key, st)) {
                u64 h = mix64(key) & mask;
                for (;;) {
                    const u64 cur = tkeys[h];
                    if (cur == key) { vid[k] = tvals[h]; break; }
                    if (cur == kEmpty) break;
                    h = (h + 1) & mask;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kExactItems; k++) {
        const unsigned long long bal = __ballot(vid[k] != kNoVoxel);
        ex[k] = carry + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        carry += (uint32_t)__popcll(bal);
    }
    if (lane == 0) wtot[w] = carry;
    __syncthreads();
    uint32_t total = 0, woff = 0;
#pragma unroll
    for (int k = 0; k < kExactThreads / kWave; k++) {
        const uint32_t t = wtot[k];
        total += t;
        if (k < w) woff += t;
    }
    if (w == 0) {
        const u64 before = lookback_exclusive(lbstat, tile, (u64)total);
        if (lane == 0) sprefix = before;
    }
    __syncthreads();
    const u64 pre = sprefix + woff;
#pragma unroll
    for (int k = 0; k < kExactItems; k++)
        if (vid[k] != kNoVoxel) {
            const u64 e = pre + ex[k];
            sortkey[e] = -(int32_t)vid[k];                  // descending stable sort of these = ascending voxel id, point order kept
            oidx[e] = (uint32_t)(base + (int64_t)k * kWave);
        }
}

// one wavefront per 64 sorted hits: it owns the voxels whose first hit lies among them
__global__ __launch_bounds__(256) void k_exact_sum(const float *__restrict__ points, int c, const int32_t *__restrict__ sortkey,
                                                   const int32_t *__restrict__ order, const uint32_t *__restrict__ oidx, int64_t n,
                                                   const int32_t *__restrict__ npoints, float *__restrict__ agg)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t p0 = ((int64_t)blockIdx.x * (256 / kWave) + (threadIdx.x >> 6)) * kWave, p = p0 + lane;
    const int32_t key = p < n ? sortkey[order[p]] : INT_MIN;
    const int32_t prev = p > 0 && p - 1 < n ? sortkey[order[p - 1]] : INT_MIN;
    unsigned long long heads = __ballot(key != INT_MIN && (p == 0 || prev != key));
    while (heads) {
        const int hl = __builtin_ctzll(heads);
        heads &= heads - 1;
        const int64_t h = p0 + hl;
        const uint32_t v = (uint32_t)(-__shfl(key, hl, kWave));
        const uint32_t cnt = (uint32_t)npoints[v];
        for (int d0 = 0; d0 < c; d0 += 4) {                 // four channels at a time, each a running fp32 sum in point order
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            for (uint32_t k0 = 0; k0 < cnt; k0 += kWave) {
                const uint32_t k = k0 + lane;
                float x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f;
                if (k < cnt) {
                    const float *row = points + (size_t)oidx[order[h + k]] * c + d0;
                    x0 = row[0];
                    if (d0 + 1 < c) x1 = row[1];
                    if (d0 + 2 < c) x2 = row[2];
                    if (d0 + 3 < c) x3 = row[3];
                }
                const int m = (int)(cnt - k0 < (uint32_t)kWave ? cnt - k0 : (uint32_t)kWave);
                for (int j = 0; j < m; j++) {               // (wave-uniform: every lane keeps the same sums)
                    s0 += __shfl(x0, j, kWave); s1 += __shfl(x1, j, kWave); s2 += __shfl(x2, j, kWave); s3 += __shfl(x3, j, kWave);
                }
            }
            if (lane == 0) {
                const float dv = (float)(int32_t)cnt;       // voxelize.cpp:164 (float / int)
                float *out = agg + (size_t)v * c + d0;
                out[0] = s0 / dv;
                if (d0 + 1 < c) out[1] = s1 / dv;
                if (d0 + 2 < c) out[2] = s2 / dv;
                if (d0 + 3 < c) out[3] = s3 / dv;
            }
        }
    }
}

// ------------------------------------------------------------------ workspace layout
struct VoxelWs {
    u64 *tabA;            // packed words, or plain keys
    uint2 *tabB;          // plain {first, count}
    u64 *aux;
    uint32_t *vidarr;
    u64 cap;
    size_t tab_bytes;     // bytes of tabA..vidarr (reused as sort scratch by the filter)
    uint32_t *pslot, *parr;
    unsigned char *flags;
    u64 *fwords;
    uint32_t *fwpre, *bsumF, *bsumA;
    uint32_t *list;       // per-voxel segments: point indices sorted ascending (first max_points valid)
    uint32_t *unsorted;   // per-voxel segments in arrival order
    uint32_t *voff;
    uint4 *vinfo;         // [V] {key lo, key hi, segment base, count}
    float4 *staged;       // C == 4: point rows in per-voxel segments, point order (first max_points valid)
    uint32_t *big_list;   // overflow voxels (count > max_points) awaiting k_overflow_reduce
    uint32_t *big_count;
    u64 *bsum;            // generic scans (filter)
    int32_t *newid;
    uint32_t *coff;
    uint32_t *fcur;       // filter: arrival cursor per voxel
    int64_t npad;
    size_t bytes;
};

static u64 table_capacity(int64_t n)
{
    u64 cap = 1024;
    while (cap < (u64)n * 2ull) cap <<= 1;
    return cap;
}

static VoxelWs carve(void *ws, size_t ws_bytes, int64_t n, int64_t nvox)
{
    WsCarver w(ws, ws_bytes);
    VoxelWs r;
    r.npad = d3d_divup(n > 0 ? n : 1, kFlagTile) * kFlagTile;      // multiple of 16384 (and of kScanTile)
    const int64_t m = n > nvox ? n : nvox;
    r.cap = table_capacity(n);
    r.tabA = w.take<u64>(r.cap);
    r.tabB = w.take<uint2>(r.cap);
    r.aux = w.take<u64>(r.cap);
    r.vidarr = w.take<uint32_t>(r.cap);
    r.tab_bytes = w.off;
    r.pslot = w.take<uint32_t>(r.npad);
    r.parr = w.take<uint32_t>(r.npad);
    r.flags = w.take<unsigned char>(r.npad);
    r.fwords = w.take<u64>(r.npad / 64);
    r.fwpre = w.take<uint32_t>(r.npad / 64);
    r.bsumF = w.take<uint32_t>(r.npad / kFlagTile + 1);
    r.bsumA = w.take<uint32_t>(r.cap / kSweepTile + 1);
    r.list = w.take<uint32_t>(r.npad + 4);
    r.unsorted = w.take<uint32_t>(r.npad + 4);
    r.voff = w.take<uint32_t>(r.npad + 4);
    r.vinfo = w.take<uint4>(r.npad);
    r.staged = w.take<float4>(r.npad + 4);
    r.big_list = w.take<uint32_t>(r.npad);
    r.big_count = w.take<uint32_t>(64);
    r.bsum = w.take<u64>(d3d_divup(m > 0 ? m : 1, kScanTile) + 1);
    r.newid = w.take<int32_t>(nvox > 0 ? nvox : 1);
    r.coff = w.take<uint32_t>(nvox > 0 ? nvox : 1);
    r.fcur = w.take<uint32_t>(nvox > 0 ? nvox : 1);
    r.bytes = w.off;
    return r;
}

static inline unsigned grid_for(int64_t work, int block, int64_t maxblocks = 256 * 16)
{
    int64_t g = d3d_divup(work > 0 ? work : 1, block);
    return (unsigned)(g < maxblocks ? g : maxblocks);
}

static inline int bits_for(u64 maxval)   // smallest b with maxval <= 2^b - 1
{
    int b = 1;
    while (b < 64 && ((1ull << b) - 1) < maxval) b++;
    return b;
}

struct IndexOpts {
    uint32_t max_points;      // 0: no ranking lists
    uint32_t max_voxels;
    int64_t *first_out;
    int64_t index_offset;
    int64_t *mapping;         // optional point -> voxel id
    bool stage4;              // C == 4 and 16-byte aligned points: stage rows for fill / reduction
};

// table + first-seen numbering (+ per-voxel sorted index lists when max_points > 0)
template <class Key, class Tab>
static int build_index(const Key &kf, const Tab &tab, const float *points, int64_t n, int c, const VoxelWs &w,
                       int64_t *counts, const IndexOpts &o, hipStream_t st)
{
    const int64_t cap = (int64_t)w.cap;
    BoxParams *box = nullptr;
    if constexpr (Key::kBox) box = kf.prm;
    // (the sparse contract's k_bbox reads the points anyway)
    const bool warm = !Key::kBox && n > 0 && ((reinterpret_cast<uintptr_t>(points) & 15) == 0) &&
                      (size_t)n * c * 4 <= ((size_t)64 << 20);          // must fit the cache next to the table
    D3D_LAUNCH("k_init", k_init<Tab>, dim3(grid_for(cap, 256)), dim3(256), 0, st, tab, cap, w.flags, w.npad / 16, counts,
               w.big_count, box, warm ? reinterpret_cast<const float4 *>(points) : nullptr, warm ? n * c / 4 : (int64_t)0);
    if (n > 0) {
        const bool vec4 = (c == 4) && ((reinterpret_cast<uintptr_t>(points) & 15) == 0);
        if constexpr (Key::kBox) {
            const unsigned nb = (unsigned)std::min<int64_t>(d3d_divup(n, 256 * 4), 512);
            int *partial = reinterpret_cast<int *>(w.big_list);          // nb x 6 ints (the sparse contract has no work list)
            unsigned int *ticket = w.big_count + 32;                      // zeroed by k_init
            if (vec4) D3D_LAUNCH("k_bbox", k_bbox<true>, dim3(nb), dim3(256), 0, st, kf, points, n, c, counts, partial, ticket);
            else D3D_LAUNCH("k_bbox", k_bbox<false>, dim3(nb), dim3(256), 0, st, kf, points, n, c, counts, partial, ticket);
        }
        dim3 grid((unsigned)d3d_divup(n, 256));
        uint32_t *parr = o.max_points ? w.parr : nullptr;
        if (vec4)
            D3D_LAUNCH("k_insert", (k_insert<Key, Tab, true>), grid, dim3(256), 0, st, kf, tab, points, n, c, w.cap - 1,
                       w.pslot, parr, counts);
        else
            D3D_LAUNCH("k_insert", (k_insert<Key, Tab, false>), grid, dim3(256), 0, st, kf, tab, points, n, c, w.cap - 1,
                       w.pslot, parr, counts);
    }
    const unsigned nbA = (unsigned)(cap / kSweepTile), nbF = (unsigned)(w.npad / kFlagTile);
    D3D_LAUNCH("k_sweep1", k_sweep1<Tab>, dim3(nbA), dim3(256), 0, st, tab, cap, w.flags, w.bsumA);
    D3D_LAUNCH("k_flagpack", k_flagpack, dim3(nbF), dim3(256), 0, st, w.flags, w.npad / 64, w.fwords, w.fwpre, w.bsumF);
    D3D_LAUNCH("k_scan2", k_scan2, dim3(1), dim3(1024), 0, st, w.bsumF, (int64_t)nbF, w.bsumA, (int64_t)nbA, counts,
               (u64)o.max_voxels);
    NumberOut no{w.aux, o.mapping ? w.vidarr : nullptr, w.vinfo, o.first_out, o.index_offset, o.max_voxels};
    D3D_LAUNCH("k_sweep2", k_sweep2<Tab>, dim3(nbA), dim3(256), 0, st, tab, cap, w.fwords, w.fwpre, w.bsumF, w.bsumA, no);
    if (n == 0) return D3D_OK;
    if (o.mapping)
        D3D_LAUNCH("k_map", k_map, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, w.vidarr, w.pslot, n, o.mapping);
    if (o.max_voxels == 0 || o.max_points == 0) return D3D_OK;
    const float4 *p4 = reinterpret_cast<const float4 *>(points);
    float4 *staged = o.stage4 ? w.staged : nullptr;
    D3D_LAUNCH("k_scatter", k_scatter, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, n, w.aux, w.pslot, w.parr,
               w.unsorted, w.list, p4, staged);
    D3D_LAUNCH("k_select", k_select, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, n, w.pslot, w.parr, w.unsorted,
               w.list, o.max_points, p4, staged);
    return D3D_OK;
}

static int dense_index(const DenseKey &kf, const float *points, int64_t n, int c, const VoxelWs &w, int64_t *counts,
                       const IndexOpts &o, uint32_t flags, hipStream_t st)
{
    // packed one-word slots whenever [count | key | first] fits 64 bits with >= 8 count bits
    const double cells = (double)kf.shape[0] * (double)kf.shape[1] * (double)kf.shape[2];
    bool packed = !(flags & D3D_VOXEL_PLAIN_SLOTS) && cells < 9.0e18;
    int ib = 0, kb = 0;
    if (packed) {
        ib = bits_for((u64)(n > 1 ? n - 1 : 1));
        kb = bits_for((u64)cells);            // keys are < cells <= 2^kb - 1: never all ones
        packed = ib + kb <= 56;
    }
    if (packed) {
        TabPacked tab{w.tabA, ib, kb, reinterpret_cast<uint32_t *>(w.tabB)};     // the plain table's second array is free
        return build_index(kf, tab, points, n, c, w, counts, o, st);
    }
    TabPlain tab{w.tabA, w.tabB};
    return build_index(kf, tab, points, n, c, w, counts, o, st);
}

struct DenseOut {
    uint32_t P, max_voxels;
    int reduction;
    bool agg4, fuse_pmask;
    int64_t *coords;
    int32_t *npoints;
    unsigned char *pmask;
    float *aggregates;
    BinnedExtras x;           // all null for the dense contract
    int64_t *mapping;
    unsigned char *trimmed = nullptr;   // sparse contract: flag the points beyond P of their voxel (for the TRIM filter)
    VoxelPass pass = {false, 0, {0, 0, 0}, {0, 0, 0}};   // sparse contract fused with the voxel filter
    int32_t *keepid = nullptr;          // ... then: filtered voxel id of every point (-1: dropped) instead of `mapping`
    bool map_later = false;             // ... computed by the caller's point scan (FilterPoints), not by k_map_binned here
    int64_t *early_counts = nullptr;    // ... whose output sizes k_meta_first then publishes: counts of the filter call
    int64_t *early_host = nullptr;      //     + the host-mapped notify buffer (d3d_voxelize_3d_sparse_filter)
    const int64_t *coord_offset = nullptr;   // ... subtracted from the output coords (host, 3 values)
    int32_t *count_out = nullptr;       // ... DESCENDING voxel filter: unclamped counts per first-seen voxel
    int64_t *first_out = nullptr;       // ...                          and every voxel's first point
    u64 *compact_stat = nullptr;        // ... look-back words of k_compact_kept (cleared by k_meta_first)
    uint32_t compact_tiles = 0;
    uint32_t npoints_clamp = 0xffffffffu;   // ... and voxel_npoints = min(count, max_points) (voxelize.cpp:403)
    bool lists = false;                 // dense contract with C != 4: ranked index lists + voff instead of staged rows
    uint32_t *seg_out = nullptr;        // reduce contract: segment base of every voxel's staged rows (for the caller)
    float4 *emit_voxels = nullptr;      // dense contract on C == 4 rows: k_emit writes voxels[V,P,4] too (no staging, no fill)
    float *emit_generic = nullptr;      // dense contract, C = 3, 5 .. 8: k_emit_c writes voxels[V,P,C] and the per-voxel outputs
    bool emit_reduce = false;           // reduce contract without rows: k_emit without the stretch (nothing staged by the index)
    int stage = 0;                      // d3d_voxelize_3d_dense_staged: 1 = index launches only, 2 = the output launch only
};

// n points -> which index path: bucket count / hash shift of the binned index, or false for the hash table
static bool binned_eligible(int64_t n, const VoxelWs &w, uint32_t flags, uint32_t *nbins_out, int *hshift_out)
{
    if ((flags & D3D_VOXEL_PATH_HASH) || n <= 0) return false;
    uint32_t nbins = 1;
    int hshift = 0;
    // buckets of ~kBucketTarget points up to 8192 buckets (8 M points with 16384 buckets of 512: scatter and bucket kernel
    // 10 % slower than with 8192 of 1024), the last doubling only for frames that need it
    while (nbins < 8192u && (int64_t)nbins * kBucketTarget < n) { nbins <<= 1; hshift++; }
    while (nbins < (uint32_t)kBinMax && (int64_t)nbins * 1024 < n) { nbins <<= 1; hshift++; }
    if ((int64_t)nbins * 1024 < n) return false;        // more than 16 M points: buckets would outgrow a workgroup
    const uint64_t ntiles = d3d_divup((int64_t)(w.npad / kBinTile), (int64_t)bin_passes(n));
    if ((uint64_t)nbins * ntiles * 4 > w.cap * 8 || 2 * (uint64_t)nbins + 2 > w.cap) return false;
    *nbins_out = nbins;
    *hshift_out = hshift;
    return true;
}

static bool dense_cells_fit_u32(const DenseKey &kf)     // 32-bit cell keys in LDS
{
    return (double)kf.shape[0] * (double)kf.shape[1] * (double)kf.shape[2] < 4294967295.0;
}

// ROWS: dense contract on C == 4 rows (ranked rows staged, reductions); !ROWS: keys only (sparse contract, any C)
template <class Key, bool ROWS>
static int binned_index(const Key &kf, const float *points, int64_t n, int c, const VoxelWs &w, uint32_t nbins, int hshift,
                        int64_t *counts, const DenseOut &o, hipStream_t st, bool tile_sort)
{
    const uint32_t passes = (uint32_t)bin_passes(n), ntiles = (uint32_t)d3d_divup((int64_t)(w.npad / kBinTile), (int64_t)passes);
    const float4 *p4 = reinterpret_cast<const float4 *>(points);
    typedef BinEntry<ROWS> E;
    typename E::type *bent = reinterpret_cast<typename E::type *>(w.tabA);      // cap * 8 bytes >= 16 n
    // per-point keys until the scatter: dense u32 (an index array of the hash path), sparse u64 (`staged` is not used there)
    typename E::key_store_t *pkey = ROWS ? reinterpret_cast<typename E::key_store_t *>(w.big_list)
                                         : reinterpret_cast<typename E::key_store_t *>(w.staged);
    uint32_t *tilecnt = reinterpret_cast<uint32_t *>(w.tabB);
    uint4 *vrec = reinterpret_cast<uint4 *>(w.aux);
    uint32_t *bucket_base = w.vidarr, *totals = w.vidarr + nbins + 2;
    uint32_t *pbin = w.pslot, *firstmap = w.list;
    const bool want_map = o.mapping || o.keepid || o.map_later;
    uint32_t *precpos = want_map && !o.map_later ? w.unsorted : nullptr;        // the hash path's lists are not used here
    BinnedExtras x = o.x;
    x.vidof = want_map ? w.voff : nullptr;
    x.npoints_clamp = o.npoints_clamp;
    if (o.count_out) { x.count_out = o.count_out; x.first_out = o.first_out; x.index_offset = 0; }
    if (o.coord_offset) {
        x.has_coord_sub = true;
        for (int k = 0; k < 3; k++) x.coord_sub[k] = (long long)o.coord_offset[k];
    }
    // fused sparse + filter: numbering + per-voxel outputs + output sizes in one launch (k_meta_first_lb)
    const bool meta_lb = o.map_later;
    const uint32_t mtiles = (uint32_t)(w.npad / kMetaLbTile);
    // (w.fwords: npad / 64 words of the hash path; here 2 * npad / 4096 look-back words, then, from the next multiple of 16
    // words on, up to 64 pairs of early totals at 16 words each -- as many as fit, a power of two)
    const uint32_t early_at = (2 * mtiles + 15u) & ~15u;
    uint32_t early_pairs = 0;
    if (meta_lb && o.early_host)
        for (early_pairs = 64; early_pairs > 1 && (uint64_t)early_at + 16ull * early_pairs > (uint64_t)(w.npad / 64); early_pairs >>= 1) { }
    MetaLb mlb{w.fwords, w.big_count + 40, o.compact_stat, o.compact_tiles, w.big_count + 41, o.early_host, w.fwords + early_at, early_pairs};
    u64 *zero_words = meta_lb ? mlb.stat : nullptr;
    const uint32_t nzero = early_at + 16u * early_pairs;
    u64 *early_tot = early_pairs ? w.fwords + early_at : nullptr;
    unsigned int *zero_ticket = meta_lb ? mlb.ticket : nullptr;
    x.voff = o.seg_out ? o.seg_out : (o.lists ? w.voff : nullptr);
    const bool vec4 = ROWS || (c == 4 && (reinterpret_cast<uintptr_t>(points) & 15) == 0);
    const size_t bin_lds = (size_t)nbins * 4;                 // (at 16384 buckets the scatter's 64 KB + 128 B exceed the default limit)
    // one-launch partition (k_tile_sort) whenever the frame and the table fit; else, or on request, the three-pass one
    // tiles of 8192 points (4096: profiles/r04_bucket_target.txt); large frames of the dense contract on C == 4 rows: 16384 -- half
    // the tiles, so half the table, and k_bucket_index finds a bucket's entries in half as many runs of twice the length
    const bool big_tiles = ROWS && vec4 && sizeof(typename Key::bin_key_t) == 4 && n >= kBigTileMinPoints;
    // round 5: the lean bucket kernel + packed first-point entries, whenever k_emit is the consumer and nothing needs the
    // cells' first indices per point (the point -> voxel map, the sparse contract's filters)
    const bool fm_packed = ROWS && (o.emit_voxels || o.emit_reduce) && !precpos && !o.map_later && !o.trimmed &&
                           !o.pass.on && n <= kFmMaxPoints;
    x.fm_packed = fm_packed;
    // round 6: the dense contract itself (nothing for the sharded operator, no resident rows, no staged call) leaves through
    // k_emit_split, and part of its zero padding under the index launches (ZeroFill)
    const bool split = fm_packed && o.emit_voxels && !x.row_state && o.stage == 0 && !x.keys_out && !x.first_out && !x.voff && !want_map &&
                       D3D_TUNE_VAL(3, 1) != 0;
    const int fill_sort_k = split ? D3D_TUNE_VAL(6, 36) : 0, fill_count_k = split ? D3D_TUNE_VAL(11, 12) : 0;    // 1024 x 16 B per filler
    // (frames of 0.72 .. 1.3 M points: below, k_emit_split is not store-bound and the fillers only lengthen the partition -- 0.5 M
    // points 71 -> 78 us, 0.1 M 57 -> 64 --, above, every CU has a tile: profiles/r06_ab_sizes.txt)
    const bool fill_tiles = fill_sort_k > 0 && !big_tiles && vec4 && tile_sort && (w.npad >> 13) >= 88 && (w.npad >> 13) <= 160;
    // tiles of 4096 points while tiles of 8192 would leave a third of the CUs without a workgroup (one workgroup per tile):
    // 1 M points 16.2 -> 13.7 us with the bucket kernel unchanged; at 2 M points (245 tiles of 8192) +4 us, at 4 M +6; tiles of
    // 2048 points: +2 us in the bucket kernel (runs of two entries) -- profiles/r05_b_tune.txt.  Round 6: where fillers take the
    // CUs without a tile, tiles of 8192 it is (those CUs then write 76 MB of zeros in the 16 us).
    const int tune_tile = !big_tiles && vec4 && ROWS ? D3D_TUNE_VAL(2, (w.npad >> 13) <= 160 && !fill_tiles ? 12 : 0) : 0;
    const bool small_tiles = tune_tile == 12, tiny_tiles = tune_tile == 11;
    const int tshift = big_tiles ? 14 : small_tiles ? 12 : tiny_tiles ? 11 : 13;
    const uint32_t stiles = (uint32_t)(w.npad >> tshift);
    uint32_t *table = nullptr, *tileinfo = nullptr, *gpos = reinterpret_cast<uint32_t *>(w.vinfo) + w.npad;
    if (tile_sort && n <= (ROWS ? kTileSortMaxPoints : kTileSortMaxPointsSparse) && nbins <= 8192u && stiles <= (uint32_t)kRunCap &&
        ((uint64_t)nbins + 1) * stiles * 4 <= w.cap * 8) {
        table = tilecnt;
        tileinfo = tilecnt + (size_t)nbins * stiles;
    }
    // (early zero lines: -6 us at 1 M points, -7 at 2 M, -3 at 4 M, nothing at 8 M, where the launch is in its steady state)
    x.early_zero = fm_packed && D3D_TUNE_VAL(4, n < kBigTileMinPoints ? 1 : 0) != 0;
    const bool do_index = o.stage != 2;                        // (stage 2: this frame's index was launched by an earlier call)
    // voxels[0 .. prefilled): as much as the idle CUs of the two launches write in passing, at most the voxels a LiDAR frame of
    // this size has (9 n / 16; config 2: 0.585 n) and never more than the tensor's rows
    ZeroFill zf_sort, zf_count;
    uint32_t prefilled = 0;
    if (split && table && do_index) {
        const int64_t capv = n < (int64_t)o.max_voxels ? n : (int64_t)o.max_voxels;
        const int64_t all16 = (n * 9 / 16 < capv ? n * 9 / 16 : capv) * (int64_t)o.P;
        const uint32_t sort_wgs = ((stiles + 7u) >> 3) << 3, count_wgs = (uint32_t)(w.npad / kFlagTile);
        int64_t at = 0;
        if (fill_tiles && tshift == 13 && sort_wgs + 64u <= (uint32_t)kNumCUs) {
            zf_sort.nblk = (uint32_t)kNumCUs - sort_wgs;
            zf_sort.first = sort_wgs;
            zf_sort.dst = o.emit_voxels;
            zf_sort.n16 = (int64_t)zf_sort.nblk * fill_sort_k * 1024;
            if (zf_sort.n16 > kFillSortMax16) zf_sort.n16 = kFillSortMax16;      // (what leaves in the launch's 16 us, however many fillers)
            if (zf_sort.n16 > all16) zf_sort.n16 = all16;
            at = zf_sort.n16;
        }
        if (fill_count_k > 0 && fill_tiles && count_wgs + 64u <= (uint32_t)kNumCUs && at < all16) {
            zf_count.nblk = (uint32_t)kNumCUs - count_wgs;
            zf_count.first = count_wgs;
            zf_count.dst = o.emit_voxels + at;
            zf_count.n16 = (int64_t)zf_count.nblk * fill_count_k * 1024;
            if (zf_count.n16 > all16 - at) zf_count.n16 = all16 - at;
            at += zf_count.n16;
        }
        prefilled = (uint32_t)(at / (int64_t)o.P);
    }
    if (o.emit_voxels && o.stage == 0) {
        g_last_plan[0] = split ? 1 : 0;
        g_last_plan[1] = prefilled;
        g_last_plan[2] = zf_sort.n16 * 16;
        g_last_plan[3] = zf_count.n16 * 16;
    }
    if (do_index && table) {
        const size_t lds = ((size_t)1 << tshift) * (sizeof(typename Key::bin_key_t) + 2) + bin_lds;
        uint32_t *ppos = o.map_later ? pbin : nullptr;      // (pfirst: by point, the first point of its voxel when it is kept)
#define D3D_TILE_SORT(V4, IT)                                                                                                   \
    do {                                                                                                                        \
        if (lds + 1024 > 65536)                                                                                                 \
            D3D_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_sort<Key, V4, ROWS, IT>),                  \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
        D3D_LAUNCH("k_tile_sort", (k_tile_sort<Key, V4, ROWS, IT>), dim3((((stiles + 7u) >> 3) << 3) + zf_sort.nblk), dim3(kSortThreads), lds, st, kf, points, n, c, \
                   nbins, stiles, bent, table, tileinfo, ppos, firstmap, counts, o.mapping, o.trimmed, o.keepid, zero_words,    \
                   nzero, zero_ticket, true, zf_sort);                                                                          \
    } while (0)
        if (big_tiles) D3D_TILE_SORT(true, 16);
        else if (small_tiles) D3D_TILE_SORT(true, 4);
        else if (tiny_tiles) D3D_TILE_SORT(true, 2);
        else if (vec4) D3D_TILE_SORT(true, 8);
        else D3D_TILE_SORT(false, 8);
#undef D3D_TILE_SORT
    } else if (do_index) {
    if (bin_lds + 256 > 65536) {
        D3D_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bin_scatter<ROWS>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)bin_lds));
    }
    if (vec4)
        D3D_LAUNCH("k_bin_count", (k_bin_count<Key, true, ROWS>), dim3(ntiles), dim3(kBinThreads), bin_lds, st, kf, points, n, c, nbins,
                   passes, pbin, pkey, tilecnt, firstmap, counts, o.mapping, o.trimmed, o.keepid, zero_words, nzero, zero_ticket);
    else
        D3D_LAUNCH("k_bin_count", (k_bin_count<Key, false, ROWS>), dim3(ntiles), dim3(kBinThreads), bin_lds, st, kf, points, n, c, nbins,
                   passes, pbin, pkey, tilecnt, firstmap, counts, o.mapping, o.trimmed, o.keepid, zero_words, nzero, zero_ticket);
    D3D_LAUNCH("k_bin_scan", k_bin_scan, dim3((nbins + kWave - 1) / kWave), dim3(1024), 0, st, tilecnt, nbins, ntiles, totals);
    D3D_LAUNCH("k_bin_scatter", k_bin_scatter<ROWS>, dim3(ntiles), dim3(kBinThreads), bin_lds, st, pkey, n, nbins, pbin, tilecnt, totals,
               bucket_base, bent, counts, o.map_later, passes);
    }
    if (!do_index) { }
    else if (!ROWS && o.lists)
        D3D_LAUNCH("k_bucket_index", (k_bucket_index<Key, false, true>), dim3(nbins), dim3(kBucketThreads), 0, st, kf, o.pass,
                   reinterpret_cast<const typename BinEntry<false>::type *>(bent), p4, bucket_base, hshift, o.P, (int)D3D_REDUCE_NONE,
                   w.staged, vrec, firstmap, counts, precpos, w.parr, reinterpret_cast<uint32_t *>(w.vinfo), o.trimmed, w.big_list,
                   o.reduction != D3D_REDUCE_NONE ? w.unsorted : (uint32_t *)nullptr, table, stiles, tshift, tileinfo, gpos, o.map_later ? pbin : (uint32_t *)nullptr,
                   early_tot, x.npoints_clamp, early_pairs ? early_pairs - 1u : 0u);
    else if (fm_packed) {
        if constexpr (ROWS)
            D3D_LAUNCH("k_bucket_index", (k_bucket_index<Key, true, true, false, true>), dim3(nbins), dim3(kBucketThreads), 0, st, kf, o.pass,
                       bent, p4, bucket_base, hshift, o.P, o.agg4 ? o.reduction : (int)D3D_REDUCE_NONE, w.staged, vrec, firstmap,
                       counts, (uint32_t *)nullptr, w.parr, reinterpret_cast<uint32_t *>(w.vinfo), (unsigned char *)nullptr, w.big_list,
                       (uint32_t *)nullptr, table, stiles, tshift, tileinfo, gpos, (uint32_t *)nullptr, (u64 *)nullptr, 0u, 0u,
                       bits_for((u64)(n > 1 ? n - 1 : 1)), (uint32_t)D3D_TUNE_VAL(0, (int)kDenseMin));
    } else if (ROWS && (o.emit_voxels || o.emit_reduce))
        D3D_LAUNCH("k_bucket_index", (k_bucket_index<Key, ROWS, true, false>), dim3(nbins), dim3(kBucketThreads), 0, st, kf, o.pass,
                   bent, p4, bucket_base, hshift, o.P, o.agg4 ? o.reduction : (int)D3D_REDUCE_NONE, w.staged, vrec, firstmap,
                   counts, precpos, w.parr, reinterpret_cast<uint32_t *>(w.vinfo), o.trimmed, w.big_list /* the per-point keys are
                   done with it; w.unsorted may be precpos */, (uint32_t *)nullptr, table, stiles, tshift, tileinfo, gpos, o.map_later ? pbin : (uint32_t *)nullptr,
                   early_tot, x.npoints_clamp, early_pairs ? early_pairs - 1u : 0u);
    else
        D3D_LAUNCH("k_bucket_index", (k_bucket_index<Key, ROWS, false>), dim3(nbins), dim3(kBucketThreads), 0, st, kf, o.pass,
                   bent, p4, bucket_base, hshift, o.P, o.agg4 ? o.reduction : (int)D3D_REDUCE_NONE, w.staged, vrec, firstmap, counts,
                   precpos, w.parr, reinterpret_cast<uint32_t *>(w.vinfo), o.trimmed, (uint32_t *)nullptr, (uint32_t *)nullptr,
                   table, stiles, tshift, tileinfo, gpos, o.map_later ? pbin : (uint32_t *)nullptr, early_tot, x.npoints_clamp, early_pairs ? early_pairs - 1u : 0u);
    if constexpr (!ROWS) {
        if (meta_lb) {
            D3D_LAUNCH("k_meta_first_lb", k_meta_first_lb<Key>, dim3(mtiles), dim3(kMetaLbThreads), 0, st, kf, w.npad, firstmap, vrec,
                       o.max_voxels, o.coords, o.npoints, counts, x, mlb, points, c);
            return D3D_OK;
        }
    }
    const unsigned nbF = (unsigned)(w.npad / kFlagTile);            // <= 1024 (n <= 16 M)
    if (do_index) D3D_LAUNCH("k_first_count", k_first_count, dim3(nbF + zf_count.nblk), dim3(1024), 0, st, firstmap, w.fwpre, w.bsumF, w.big_count, zf_count);
    if (o.stage == 1) return D3D_OK;
    const dim3 grid((unsigned)(w.npad / 256));
    if constexpr (!ROWS && std::is_same<Key, DenseKey>::value) {
        if (o.emit_generic) {
            const int pshift = (o.P & (o.P - 1)) == 0 ? __builtin_ctz(o.P) : -1;
#define D3D_EMIT_C(CC)                                                                                                          \
    D3D_LAUNCH("k_emit_c", (k_emit_c<Key, CC>), grid, dim3(256), 0, st, kf, w.npad, firstmap, w.fwpre, w.bsumF, vrec, o.max_voxels,  \
               points, w.big_list, o.P, pshift, o.reduction, o.coords, o.npoints, o.fuse_pmask ? o.pmask : nullptr, o.aggregates,   \
               w.voff, o.emit_generic, counts, x.host_counts, reinterpret_cast<uint32_t *>(w.vinfo), w.big_count, x.row_state)
            switch (c) {
            case 3: D3D_EMIT_C(3); break;
            case 5: D3D_EMIT_C(5); break;
            case 6: D3D_EMIT_C(6); break;
            case 7: D3D_EMIT_C(7); break;
            default: D3D_EMIT_C(8); break;
            }
#undef D3D_EMIT_C
            return D3D_OK;
        }
    }
    if constexpr (ROWS) {
        if (o.emit_voxels || o.emit_reduce) {
            const int pshift = (o.P & (o.P - 1)) == 0 ? __builtin_ctz(o.P) : -1;
            // both roles in ONE workgroup of 512 lanes from 3 M points on (less LDS per wavefront, 32 instead of 28 per CU): 4 M
            // points 384 -> 358 us, 8 M 704 -> 664, a uniform cloud of 1 M 99 -> 93; 2 M and below: the same or worse (config 2
            // with its fillers 61 -> 65) -- profiles/r06_ab_sizes.txt
            if (split && D3D_TUNE_VAL(1, n >= (3ll << 20) ? 1 : 0) == 1) {
                if (o.agg4)
                    D3D_LAUNCH("k_emit_split", (k_emit_split<Key, true, 512>), dim3(grid.x), dim3(512), 0, st, kf, w.npad, firstmap, w.fwpre,
                               w.bsumF, vrec, o.max_voxels, p4, w.big_list, w.staged, o.P, pshift, o.reduction, o.coords, o.npoints,
                               o.fuse_pmask ? o.pmask : nullptr, reinterpret_cast<float4 *>(o.aggregates), o.emit_voxels, counts,
                               x.host_counts, prefilled, x.aux_value, (uint32_t)D3D_TUNE_VAL(15, 0));
                else
                    D3D_LAUNCH("k_emit_split", (k_emit_split<Key, false, 512>), dim3(grid.x), dim3(512), 0, st, kf, w.npad, firstmap, w.fwpre,
                               w.bsumF, vrec, o.max_voxels, p4, w.big_list, w.staged, o.P, pshift, o.reduction, o.coords, o.npoints,
                               o.fuse_pmask ? o.pmask : nullptr, (float4 *)nullptr, o.emit_voxels, counts, x.host_counts, prefilled,
                               x.aux_value, (uint32_t)D3D_TUNE_VAL(15, 0));
            } else if (split) {
                if (o.agg4)
                    D3D_LAUNCH("k_emit_split", (k_emit_split<Key, true>), dim3(2u * grid.x), dim3(256), 0, st, kf, w.npad, firstmap, w.fwpre,
                               w.bsumF, vrec, o.max_voxels, p4, w.big_list, w.staged, o.P, pshift, o.reduction, o.coords, o.npoints,
                               o.fuse_pmask ? o.pmask : nullptr, reinterpret_cast<float4 *>(o.aggregates), o.emit_voxels, counts,
                               x.host_counts, prefilled, x.aux_value, (uint32_t)D3D_TUNE_VAL(15, 0));
                else
                    D3D_LAUNCH("k_emit_split", (k_emit_split<Key, false>), dim3(2u * grid.x), dim3(256), 0, st, kf, w.npad, firstmap, w.fwpre,
                               w.bsumF, vrec, o.max_voxels, p4, w.big_list, w.staged, o.P, pshift, o.reduction, o.coords, o.npoints,
                               o.fuse_pmask ? o.pmask : nullptr, (float4 *)nullptr, o.emit_voxels, counts, x.host_counts, prefilled,
                               x.aux_value, (uint32_t)D3D_TUNE_VAL(15, 0));
            } else
            if (x.row_state && o.emit_voxels) {
                if (o.agg4)
                    D3D_LAUNCH("k_emit_resident", (k_emit<Key, true, true>), grid, dim3(256), 0, st, kf, w.npad, firstmap, w.fwpre, w.bsumF, vrec,
                               o.max_voxels, p4, w.big_list, w.staged, o.P, pshift, o.reduction, o.coords, o.npoints,
                               o.fuse_pmask ? o.pmask : nullptr, reinterpret_cast<float4 *>(o.aggregates), o.emit_voxels, counts,
                               x.host_counts, x);
                else
                    D3D_LAUNCH("k_emit_resident", (k_emit<Key, false, true>), grid, dim3(256), 0, st, kf, w.npad, firstmap, w.fwpre, w.bsumF, vrec,
                               o.max_voxels, p4, w.big_list, w.staged, o.P, pshift, o.reduction, o.coords, o.npoints,
                               o.fuse_pmask ? o.pmask : nullptr, (float4 *)nullptr, o.emit_voxels, counts, x.host_counts, x);
            } else if (o.agg4)
                D3D_LAUNCH("k_emit", (k_emit<Key, true>), grid, dim3(256), 0, st, kf, w.npad, firstmap, w.fwpre, w.bsumF, vrec,
                           o.max_voxels, p4, w.big_list, w.staged, o.P, pshift, o.reduction, o.coords, o.npoints,
                           o.fuse_pmask ? o.pmask : nullptr, reinterpret_cast<float4 *>(o.aggregates), o.emit_voxels, counts,
                           x.host_counts, x);
            else
                D3D_LAUNCH("k_emit", (k_emit<Key, false>), grid, dim3(256), 0, st, kf, w.npad, firstmap, w.fwpre, w.bsumF, vrec,
                           o.max_voxels, p4, w.big_list, w.staged, o.P, pshift, o.reduction, o.coords, o.npoints,
                           o.fuse_pmask ? o.pmask : nullptr, (float4 *)nullptr, o.emit_voxels, counts, x.host_counts, x);
            if (want_map && !o.map_later)
                D3D_LAUNCH("k_map_binned", k_map_binned, dim3(grid_for(n, 256)), dim3(256), 0, st, bucket_base, nbins, precpos,
                           reinterpret_cast<const uint32_t *>(bent), E::kIdxStride, E::kIdxOff, x.vidof, o.mapping, o.keepid,
                           (const unsigned char *)o.trimmed, (const uint32_t *)tileinfo, stiles, tshift);
            return D3D_OK;
        }
    }
    if (o.agg4)
        D3D_LAUNCH("k_meta_first", (k_meta_first<Key, true>), grid, dim3(256), 0, st, kf, w.npad, firstmap, w.fwpre, w.bsumF, vrec,
                   o.max_voxels, w.vinfo, w.staged, o.P, o.reduction, o.coords, o.npoints, o.fuse_pmask ? o.pmask : nullptr,
                   reinterpret_cast<float4 *>(o.aggregates), counts, x, points, c);
    else
        D3D_LAUNCH("k_meta_first", (k_meta_first<Key, false>), grid, dim3(256), 0, st, kf, w.npad, firstmap, w.fwpre, w.bsumF, vrec,
                   o.max_voxels, ROWS || o.lists ? w.vinfo : (uint4 *)nullptr, w.staged, o.P, o.reduction, o.coords, o.npoints,
                   o.fuse_pmask ? o.pmask : nullptr, (float4 *)nullptr, counts, x, points, c);
    if (want_map && !o.map_later)
        D3D_LAUNCH("k_map_binned", k_map_binned, dim3(grid_for(n, 256)), dim3(256), 0, st, bucket_base, nbins, precpos,
                   reinterpret_cast<const uint32_t *>(bent), E::kIdxStride, E::kIdxOff, x.vidof, o.mapping, o.keepid,
                   (const unsigned char *)o.trimmed, (const uint32_t *)tileinfo, stiles, tshift);
    return D3D_OK;
}

// Round 5: the fused sparse + filter call in THREE launches on the dense contract's index machinery (BoundKey: the cell inside
// the filter's coordinate bounds is a 32-bit key): k_tile_sort -> k_bucket_index<.., V2> (the voxel filter and the TRIM point
// filter inside; a handle of its voxel per kept point) -> k_sparse_finish (numbering, per-voxel outputs, compaction).
// Returns D3D_ERR_UNSUPPORTED when the frame does not take this path (the caller then runs round 4's four launches).
static int sparse_fused_index(const BoundKey &kf, const float *points, int64_t n, int c, const VoxelWs &w, uint32_t nbins, int hshift,
                              const VoxelPass &pass, uint32_t P /* 0: no point filter */, uint32_t max_voxels, uint32_t npoints_clamp,
                              const int64_t *coord_offset, float *out_feats, int64_t *out_mask, int64_t *out_mapping,
                              int32_t *out_npoints, int64_t *out_coords, int64_t *sparse_counts, int64_t *counts, int64_t *host,
                              hipStream_t st)
{
    typedef BinEntry<true> E;
    if (n > kFmMaxPoints || nbins > 8192u) return D3D_ERR_UNSUPPORTED;
    const bool vec4 = c == 4 && ((reinterpret_cast<uintptr_t>(points) | reinterpret_cast<uintptr_t>(out_feats)) & 15) == 0;
    const bool big_tiles = vec4 && n >= kBigTileMinPoints;
    const int tune_tile = !big_tiles && vec4 ? D3D_TUNE_VAL(2, (w.npad >> 13) <= 160 ? 12 : 0) : 0;
    const int tshift = big_tiles ? 14 : tune_tile == 12 ? 12 : 13;
    const uint32_t stiles = (uint32_t)(w.npad >> tshift);
    if (n > kTileSortMaxPoints || stiles > (uint32_t)kRunCap || ((uint64_t)nbins + 1) * stiles * 4 > w.cap * 8) return D3D_ERR_UNSUPPORTED;
    E::type *bent = reinterpret_cast<E::type *>(w.tabA);
    uint32_t *table = reinterpret_cast<uint32_t *>(w.tabB), *tileinfo = table + (size_t)nbins * stiles;
    uint4 *vrec = reinterpret_cast<uint4 *>(w.aux);
    uint32_t *firstmap = w.list, *phandle = w.pslot;
    uint32_t *gpos = reinterpret_cast<uint32_t *>(w.vinfo) + w.npad;
    // look-back words of k_sparse_finish (two per tile of 4096 points) and, behind them, up to 64 pairs of early totals
    const uint32_t ftiles = (uint32_t)(w.npad / kFinTile);
    const uint32_t early_at = (2 * ftiles + 15u) & ~15u;
    uint32_t early_pairs = 64;       // (also without a host buffer: k_sparse_finish learns from them that max_voxels cannot cut)
    for (; early_pairs > 1 && (uint64_t)early_at + 16ull * early_pairs > (uint64_t)(w.npad / 64); early_pairs >>= 1) { }
    if ((uint64_t)early_at + 16ull * early_pairs > (uint64_t)(w.npad / 64)) early_pairs = 0;
    u64 *early_tot = early_pairs ? w.fwords + early_at : nullptr;
    unsigned int *ticket = w.big_count + 40;
    const size_t lds = ((size_t)1 << tshift) * (sizeof(uint32_t) + 2) + (size_t)nbins * 4;
#define D3D_TILE_SORT_B(V4, IT)                                                                                                  \
    do {                                                                                                                        \
        if (lds + 1024 > 65536)                                                                                                 \
            D3D_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_sort<BoundKey, V4, true, IT>),             \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
        D3D_LAUNCH("k_tile_sort", (k_tile_sort<BoundKey, V4, true, IT>), dim3(((stiles + 7u) >> 3) << 3), dim3(kSortThreads), lds, st, kf, points, \
                   n, c, nbins, stiles, bent, table, tileinfo, phandle, firstmap, sparse_counts, (int64_t *)nullptr,           \
                   (unsigned char *)nullptr, (int32_t *)nullptr, w.fwords, early_at + 16u * early_pairs, ticket, false); \
    } while (0)
    if (big_tiles) D3D_TILE_SORT_B(true, 16);
    else if (tshift == 12) D3D_TILE_SORT_B(true, 4);
    else if (vec4) D3D_TILE_SORT_B(true, 8);
    else D3D_TILE_SORT_B(false, 8);
#undef D3D_TILE_SORT_B
    D3D_LAUNCH("k_bucket_index", (k_bucket_index<BoundKey, true, true, false, true>), dim3(nbins), dim3(kBucketThreads), 0, st, kf, pass, bent,
               (const float4 *)nullptr, (const uint32_t *)nullptr, hshift, P, (int)D3D_REDUCE_NONE, (float4 *)nullptr, vrec, firstmap,
               sparse_counts, (uint32_t *)nullptr, w.parr, reinterpret_cast<uint32_t *>(w.vinfo), (unsigned char *)nullptr, w.big_list,
               (uint32_t *)nullptr, table, stiles, tshift, tileinfo, gpos, phandle, early_tot, npoints_clamp,
               early_pairs ? early_pairs - 1u : 0u, bits_for((u64)(n > 1 ? n - 1 : 1)), (uint32_t)D3D_TUNE_VAL(0, (int)kDenseMin));
    SparseFin lb{w.fwords, w.fwords + ftiles, ticket, host, early_tot, early_pairs, {0, 0, 0}};
    if (coord_offset)
        for (int k = 0; k < 3; k++) lb.coord_sub[k] = (long long)coord_offset[k];
    if (vec4)
        D3D_LAUNCH("k_sparse_finish", k_sparse_finish<true>, dim3(ftiles), dim3(kFinThreads), 0, st, kf, n, firstmap, phandle, vrec,
                   points, c, max_voxels, npoints_clamp, out_coords, out_npoints, out_feats, out_mask, out_mapping, sparse_counts, counts, lb);
    else
        D3D_LAUNCH("k_sparse_finish", k_sparse_finish<false>, dim3(ftiles), dim3(kFinThreads), 0, st, kf, n, firstmap, phandle, vrec,
                   points, c, max_voxels, npoints_clamp, out_coords, out_npoints, out_feats, out_mask, out_mapping, sparse_counts, counts, lb);
    return D3D_OK;
}

static int make_dense_key(const int32_t *shape, const float *bound, DenseKey &kf)
{
    if (shape[0] <= 0 || shape[1] <= 0 || shape[2] <= 0) return D3D_ERR_BAD_ARG;
    if ((double)shape[0] * (double)shape[1] * (double)shape[2] >= 9.0e18) return D3D_ERR_BAD_ARG;
    for (int d = 0; d < 3; d++) {
        kf.lo[d] = bound[d << 1];
        kf.shape[d] = shape[d];
        // voxelize.cpp:84-86: float(hi - lo) / int, evaluated in fp32 on the host
        volatile float diff = bound[(d << 1) | 1] - bound[d << 1];
        volatile float sz = diff / (float)shape[d];
        kf.size[d] = sz;
    }
    return D3D_OK;
}

}  // namespace

// ====================================================================== C ABI
extern "C" size_t d3d_voxelize_workspace_bytes(int64_t n_points, int64_t n_voxels)
{
    if (n_points < 0) n_points = 0;
    if (n_voxels < 0) n_voxels = 0;
    return carve(nullptr, 0, n_points, n_voxels).bytes + 256;
}

extern "C" int d3d_internal_argsort_desc_i32(const int32_t *keys, int64_t n, int32_t *order, void *ws, size_t ws_bytes,
                                             hipStream_t st);
extern "C" size_t d3d_internal_argsort_i32_bytes(int64_t n);
extern "C" size_t d3d_internal_argsort_counts_bytes(int64_t n);
extern "C" int d3d_internal_argsort_desc_counts_dev(const int32_t *keys, int64_t n, const int64_t *n_dev, int64_t max_key_sum, int32_t *order,
                                                    void *ws, size_t ws_bytes, hipStream_t st);

// D3D_VOXEL_EXACT_MEAN (kernels above): everything it needs is in the operator's outputs; scratch = the index's, which is done
static int exact_mean_pass(const DenseKey &kf, const float *points, int64_t n, int32_t c, uint32_t P, const int64_t *coords,
                           const int32_t *npoints, float *aggregates, const int64_t *counts, const VoxelWs &w, hipStream_t st)
{
    u64 cap2 = 1024;                                        // >= 2 x the voxels that can hold more than P points
    while (cap2 < 2ull * ((u64)n / (P + 1) + 1)) cap2 <<= 1;
    if (cap2 > w.cap) return D3D_ERR_WORKSPACE;             // (w.cap >= 2 n)
    u64 *tkeys = w.aux;
    uint32_t *tvals = w.vidarr;
    int32_t *sortkey = reinterpret_cast<int32_t *>(w.unsorted), *order = reinterpret_cast<int32_t *>(w.list);
    uint32_t *oidx = w.parr;
    unsigned int *ticket = w.big_count + 42;
    const uint32_t tiles = (uint32_t)d3d_divup(n, kExactTile);
    D3D_LAUNCH("k_exact_clear", k_exact_clear, dim3(grid_for(n, 256, 1024)), dim3(256), 0, st, tkeys, (int64_t)cap2, sortkey, n, w.bsum,
               tiles, ticket);
    D3D_LAUNCH("k_exact_table", k_exact_table, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, kf, coords, npoints, counts, P, tkeys,
               tvals, cap2 - 1);
    const bool v4 = c == 4 && (reinterpret_cast<uintptr_t>(points) & 15) == 0;
    if (v4)
        D3D_LAUNCH("k_exact_collect", k_exact_collect<true>, dim3(tiles), dim3(kExactThreads), 0, st, kf, points, n, (int)c, (const u64 *)tkeys,
                   (const uint32_t *)tvals, cap2 - 1, sortkey, oidx, w.bsum, ticket);
    else
        D3D_LAUNCH("k_exact_collect", k_exact_collect<false>, dim3(tiles), dim3(kExactThreads), 0, st, kf, points, n, (int)c, (const u64 *)tkeys,
                   (const uint32_t *)tvals, cap2 - 1, sortkey, oidx, w.bsum, ticket);
    // stable sort of the hits by voxel (the table is done with: its region is the sort's scratch)
    const size_t sort_bytes = d3d_internal_argsort_i32_bytes(n);
    if (sort_bytes > w.tab_bytes) return D3D_ERR_WORKSPACE;
    if (int rc = d3d_internal_argsort_desc_i32(sortkey, n, order, w.tabA, w.tab_bytes, st)) return rc;
    D3D_LAUNCH("k_exact_sum", k_exact_sum, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, points, (int)c, (const int32_t *)sortkey,
               (const int32_t *)order, (const uint32_t *)oidx, n, npoints, aggregates);
    return D3D_OK;
}

static int voxelize_dense_core(const float *points, int64_t n, int32_t c, const int32_t *shape, const float *bound,
                               int32_t max_points, int32_t max_voxels, int32_t reduction, float *voxels, int64_t *coords,
                               uint8_t *pmask, int32_t *npoints, float *aggregates, int64_t *counts, void *workspace,
                               size_t workspace_bytes, void *stream, int64_t *host_counts, uint32_t flags, int stage, uint16_t *row_state);

static int voxelize_dense_impl(const float *points, int64_t n, int32_t c, const int32_t *shape, const float *bound,
                               int32_t max_points, int32_t max_voxels, int32_t reduction, float *voxels, int64_t *coords,
                               uint8_t *pmask, int32_t *npoints, float *aggregates, int64_t *counts, void *workspace,
                               size_t workspace_bytes, void *stream, int64_t *host_counts, uint32_t flags, int stage = 0,
                               uint16_t *row_state = nullptr)
{
    int rc = voxelize_dense_core(points, n, c, shape, bound, max_points, max_voxels, reduction, voxels, coords, pmask, npoints,
                                 aggregates, counts, workspace, workspace_bytes, stream, host_counts, flags, stage, row_state);
    if (rc == D3D_OK && (flags & D3D_VOXEL_EXACT_MEAN) && reduction == D3D_REDUCE_MEAN && max_points > 0 && max_voxels > 0 && n > 0 &&
        stage != 1) {
        DenseKey kf;
        rc = make_dense_key(shape, bound, kf);
        if (rc) return rc;
        const VoxelWs w = carve(workspace, workspace_bytes, n, 0);
        rc = exact_mean_pass(kf, points, n, c, (uint32_t)max_points, coords, npoints, aggregates, counts, w, (hipStream_t)stream);
    }
    return rc;
}

static int voxelize_dense_core(const float *points, int64_t n, int32_t c, const int32_t *shape, const float *bound,
                               int32_t max_points, int32_t max_voxels, int32_t reduction, float *voxels, int64_t *coords,
                               uint8_t *pmask, int32_t *npoints, float *aggregates, int64_t *counts, void *workspace,
                               size_t workspace_bytes, void *stream, int64_t *host_counts, uint32_t flags, int stage, uint16_t *row_state)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || c < 3 || !shape || !bound || !counts || max_points < 0 || max_voxels < 0) return D3D_ERR_BAD_ARG;
    if (flags & ~(uint32_t)D3D_VOXEL_FLAGS_ALL) return D3D_ERR_BAD_ARG;
    if (n > 0 && !points) return D3D_ERR_BAD_ARG;
    if (n >= (1ll << 31) - kFlagTile) return D3D_ERR_BAD_ARG;
    if (reduction < D3D_REDUCE_NONE || reduction > D3D_REDUCE_MIN) return D3D_ERR_UNSUPPORTED;  // voxelize.cpp:196
    if (reduction != D3D_REDUCE_NONE && !aggregates && n > 0 && max_voxels > 0) return D3D_ERR_BAD_ARG;
    if (reduction != D3D_REDUCE_NONE && max_points == 0 && n > 0 && max_voxels > 0) return D3D_ERR_UNSUPPORTED;   // aggregates need the lists
    DenseKey kf;
    int rc = make_dense_key(shape, bound, kf);
    if (rc) return rc;
    VoxelWs w = carve(workspace, workspace_bytes, n, 0);
    if (!workspace || w.bytes > workspace_bytes) return D3D_ERR_WORKSPACE;
    if ((max_voxels > 0 && n > 0) && (!voxels || !coords || !pmask || !npoints)) return D3D_ERR_BAD_ARG;

    const bool al16 = ((reinterpret_cast<uintptr_t>(points) & 15) == 0) && ((reinterpret_cast<uintptr_t>(voxels) & 15) == 0);
    const bool vec4 = (c == 4) && al16 && (reduction == D3D_REDUCE_NONE || (reinterpret_cast<uintptr_t>(aggregates) & 15) == 0);
    const int64_t cap = n < max_voxels ? n : (int64_t)max_voxels;
    const uint32_t P = (uint32_t)max_points;
    const bool fuse_pmask = P > 0 && (P % 16 == 0) && ((reinterpret_cast<uintptr_t>(pmask) & 15) == 0);
    const bool agg4 = vec4 && reduction != D3D_REDUCE_NONE && P > 0;
    const float4 *p4 = reinterpret_cast<const float4 *>(points);
    uint32_t nbins = 0;
    int hshift = 0;
    bool lists_ready = false;           // C != 4 on the binned index: w.big_list / w.unsorted / w.voff hold the lists
    bool emitted = false;               // k_emit wrote voxels[V,P,4] as well
    bool emitted_generic = false;       // k_emit_c wrote voxels[V,P,C], the per-voxel outputs and the aggregates of voxels within P
    if (max_voxels > 0 && dense_cells_fit_u32(kf) && binned_eligible(n, w, flags, &nbins, &hshift)) {
        DenseOut d{P, (uint32_t)max_voxels, reduction, agg4, fuse_pmask, coords, npoints, pmask, aggregates,
                   BinnedExtras{nullptr, 0, nullptr, -1, nullptr, host_counts}, nullptr};
        if (vec4 && P > 0 && P <= (uint32_t)kEmitCap && !(flags & D3D_VOXEL_SPLIT_FILL)) {
            d.emit_voxels = reinterpret_cast<float4 *>(voxels);
            emitted = true;
        }
        // staged calls (d3d_voxelize_3d_dense_staged): only the path whose output is ONE launch
        if (stage != 0 && !(emitted && fuse_pmask)) return D3D_ERR_UNSUPPORTED;
        d.x.row_state = row_state;
        d.stage = stage;
        if (vec4) {
            if (row_state && !emitted) return D3D_ERR_UNSUPPORTED;       // (resident rows: the one-launch output kernels only)
            rc = binned_index<DenseKey, true>(kf, points, n, c, w, nbins, hshift, counts, d, st, !(flags & D3D_VOXEL_PARTITION_3PASS));
        } else {
            // any C: the {cell, index} entries travel alone, the bucket kernel leaves per-voxel index lists in point order
            // and the generic output kernels below gather through them
            d.lists = true;
            lists_ready = true;
            if ((c == 3 || (c >= 5 && c <= 8)) && P > 0 && P <= (uint32_t)kEmitCap && (P * (uint32_t)c) % 4 == 0 &&
                (reinterpret_cast<uintptr_t>(voxels) & 15) == 0 && !(flags & D3D_VOXEL_SPLIT_FILL)) {
                d.emit_generic = voxels;
                emitted_generic = true;
            }
            if (row_state && !emitted_generic) return D3D_ERR_UNSUPPORTED;
            rc = binned_index<DenseKey, false>(kf, points, n, c, w, nbins, hshift, counts, d, st, !(flags & D3D_VOXEL_PARTITION_3PASS));
        }
        if (rc) return rc;
    } else {
        if (stage != 0 || row_state) return D3D_ERR_UNSUPPORTED;
        IndexOpts o{(uint32_t)max_points, (uint32_t)max_voxels, nullptr, 0, nullptr, vec4};
        rc = dense_index(kf, points, n, c, w, counts, o, flags, st);
        if (rc) return rc;
        if (host_counts) D3D_LAUNCH("k_notify_host", k_notify_host, dim3(1), dim3(64), 0, st, counts, host_counts);
        if (n == 0 || max_voxels == 0) return D3D_OK;
        const dim3 mgrid(grid_for(cap, 256));
        if (agg4)
            D3D_LAUNCH("k_meta", (k_meta<DenseKey, true>), mgrid, dim3(256), 0, st, kf, p4, counts, w.vinfo, w.staged, w.unsorted,
                       P, reduction, coords, npoints, w.voff, fuse_pmask ? pmask : nullptr,
                       reinterpret_cast<float4 *>(aggregates), w.big_list, w.big_count);
        else
            D3D_LAUNCH("k_meta", (k_meta<DenseKey, false>), mgrid, dim3(256), 0, st, kf, p4, counts, w.vinfo, w.staged, w.unsorted,
                       P, reduction, coords, npoints, w.voff, fuse_pmask ? pmask : nullptr, (float4 *)nullptr, w.big_list,
                       w.big_count);
        if (agg4)
            D3D_LAUNCH("k_overflow_reduce", k_overflow_reduce, dim3(512), dim3(256), 0, st, p4, w.vinfo, w.unsorted, w.big_list,
                       w.big_count, reduction, reinterpret_cast<float4 *>(aggregates));
    }
    if (P == 0) return D3D_OK;
    if (emitted || emitted_generic) {
        if (!fuse_pmask)
            D3D_LAUNCH("k_pmask", k_pmask, dim3(grid_for(d3d_divup(cap * P, 16), 256)), dim3(256), 0, st, counts, npoints, P, pmask);
        if (emitted_generic && reduction != D3D_REDUCE_NONE)          // voxels with more than P points: all their points count
            D3D_LAUNCH("k_aggregate_overflow", k_aggregate_overflow, dim3(1024), dim3(256), 0, st, points, c,
                       (const uint32_t *)reinterpret_cast<uint32_t *>(w.vinfo), (const uint32_t *)w.big_count, (const int32_t *)npoints,
                       (const uint32_t *)w.voff, (const uint32_t *)w.unsorted, reduction, aggregates);
        return D3D_OK;
    }
    if (vec4 && n <= kFillRowsMaxPoints) {
        D3D_LAUNCH("k_fill_c4", k_fill_c4_rows, dim3(grid_for(cap * P, 256, 256 * 32)), dim3(256), 0, st, w.staged, counts, w.vinfo,
                   P, reinterpret_cast<float4 *>(voxels));
    } else if (vec4) {
        // large frames: 64 voxels per wavefront (the wavefronts past the actual voxel count exit at once)
        const int pshift = (P & (P - 1)) == 0 ? __builtin_ctz(P) : -1;
        const dim3 fgrid((unsigned)d3d_divup(d3d_divup(cap > 0 ? cap : 1, 64), 256 / kWave));
        D3D_LAUNCH("k_fill_c4", k_fill_c4<64>, fgrid, dim3(256), 0, st, w.staged, counts, w.vinfo, P, pshift,
                   reinterpret_cast<float4 *>(voxels));
    }
    else if (c <= kFillLdsMaxC && (reinterpret_cast<uintptr_t>(voxels) & 15) == 0)
        D3D_LAUNCH("k_fill_generic_lds", k_fill_generic_lds, dim3(grid_for(cap * P, 256, 256 * 32)), dim3(256), 0, st, points, c,
                   counts, w.vinfo, lists_ready ? w.big_list : w.list, P, voxels);
    else
        D3D_LAUNCH("k_fill_generic", k_fill_generic, dim3(grid_for(cap * P, 256, 256 * 32)), dim3(256), 0, st, points, c,
                   counts, w.vinfo, lists_ready ? w.big_list : w.list, P, voxels);
    if (!fuse_pmask)
        D3D_LAUNCH("k_pmask", k_pmask, dim3(grid_for(d3d_divup(cap * P, 16), 256)), dim3(256), 0, st, counts, npoints, P, pmask);
    if (reduction != D3D_REDUCE_NONE && !agg4)
        D3D_LAUNCH("k_aggregate", k_aggregate, dim3(grid_for(cap * c, 256)), dim3(256), 0, st, points, c, counts, npoints,
                   w.voff, lists_ready ? w.big_list : w.list, w.unsorted, P, reduction, aggregates);
    return D3D_OK;
}

// What the calling thread's last d3d_voxelize_3d_dense[_notify] launched (bench.py prices k_emit_split on the bytes IT moves):
// out[0] = 1 when the output left through k_emit_split, out[1] = voxels whose zero padding was written under the index launches,
// out[2] / out[3] = the bytes of zeros k_tile_sort's / k_first_count's filler workgroups stored.
extern "C" int d3d_voxelize_dense_last_plan(int64_t *out4)
{
    if (!out4) return D3D_ERR_BAD_ARG;
    for (int k = 0; k < 4; k++) out4[k] = g_last_plan[k];
    return D3D_OK;
}

extern "C" int d3d_voxelize_3d_dense(const float *points, int64_t n, int32_t c, const int32_t *shape, const float *bound,
                                     int32_t max_points, int32_t max_voxels, int32_t reduction, float *voxels,
                                     int64_t *coords, uint8_t *pmask, int32_t *npoints, float *aggregates,
                                     int64_t *counts, void *workspace, size_t workspace_bytes, void *stream, uint32_t flags)
{
    return voxelize_dense_impl(points, n, c, shape, bound, max_points, max_voxels, reduction, voxels, coords, pmask, npoints,
                               aggregates, counts, workspace, workspace_bytes, stream, nullptr, flags);
}

// Same, and as soon as counts[] are final -- before the HBM-bound fill of voxels[V,P,C] is launched -- they are also
// written to host_counts[0 .. D3D_NUM_COUNTS) followed by host_counts[D3D_NUM_COUNTS] = 1.  host_counts must be
// host-mapped, coherent pinned memory (hipHostMalloc) whose flag word the caller cleared: the host learns the output
// sizes by polling it while the GPU is still writing the outputs, instead of draining the stream first.
extern "C" int d3d_voxelize_3d_dense_notify(const float *points, int64_t n, int32_t c, const int32_t *shape, const float *bound,
                                            int32_t max_points, int32_t max_voxels, int32_t reduction, float *voxels,
                                            int64_t *coords, uint8_t *pmask, int32_t *npoints, float *aggregates,
                                            int64_t *counts, void *workspace, size_t workspace_bytes, void *stream,
                                            int64_t *host_counts, uint32_t flags)
{
    if (!host_counts) return D3D_ERR_BAD_ARG;
    return voxelize_dense_impl(points, n, c, shape, bound, max_points, max_voxels, reduction, voxels, coords, pmask, npoints,
                               aggregates, counts, workspace, workspace_bytes, stream, host_counts, flags);
}

// The dense contract into a RESIDENT output (round 4): `voxels` is a buffer [capacity >= min(n, max_voxels) of any call, max_points,
// 4] that the caller keeps from frame to frame and never writes, `row_state` [capacity] uint16 beside it; both zero-filled by the
// caller ONCE (hipMemset).  A frame's voxels[0 .. V) then equal what d3d_voxelize_3d_dense writes, bit for bit, but only the rows
// that hold points -- and zeros over the rows the previous frame's voxel of the same id held -- are stored: the padding, 95 % of
// the tensor on a LiDAR frame, is already there.  The result aliases the buffer: valid until the next call on it.  Everything
// else (coords, masks, counts, aggregates, host_counts -- optional here) as in d3d_voxelize_3d_dense_notify.  Rows of 3 .. 8
// floats on 16-byte aligned buffers, max_points <= 256 (c != 4: max_points * c a multiple of 4), the binned index (frames up to
// 8 M points): else D3D_ERR_UNSUPPORTED and nothing is touched.  max_points and c are the buffer's layout: the same in every call.
extern "C" int d3d_voxelize_3d_dense_resident(const float *points, int64_t n, int32_t c, const int32_t *shape, const float *bound,
                                              int32_t max_points, int32_t max_voxels, int32_t reduction, float *voxels,
                                              uint16_t *row_state, int64_t *coords, uint8_t *pmask, int32_t *npoints, float *aggregates,
                                              int64_t *counts, void *workspace, size_t workspace_bytes, void *stream,
                                              int64_t *host_counts, uint32_t flags)
{
    if (!row_state) return D3D_ERR_BAD_ARG;
    if (max_voxels <= 0 || max_points <= 0) return D3D_ERR_UNSUPPORTED;
    if (n == 0) row_state = nullptr;               // an empty frame: no voxel, no row, the state stays
    if (flags & (D3D_VOXEL_PATH_HASH | D3D_VOXEL_SPLIT_FILL)) return D3D_ERR_UNSUPPORTED;
    return voxelize_dense_impl(points, n, c, shape, bound, max_points, max_voxels, reduction, voxels, coords, pmask, npoints,
                               aggregates, counts, workspace, workspace_bytes, stream, host_counts, flags, 0, row_state);
}

// d3d_voxelize_3d_dense_notify in two calls, for callers that pipeline a stream of frames (round 4): stage 1 enqueues the
// index launches (partition, bucket index, first-point counts: latency-bound, little traffic), stage 2 the output launch
// (k_emit: bandwidth-bound) -- on a DIFFERENT stream if the caller likes, behind an event recorded after stage 1, so that frame
// k + 1's index runs under frame k's output.  Same arguments in both calls (the workspace carries the index from one to the
// other: one workspace per frame in flight).  Available where the output is one launch (C == 4 rows, 16-byte aligned buffers,
// max_points a multiple of 16 up to 256, a frame the binned index takes); D3D_ERR_UNSUPPORTED otherwise -- run stage 0.
extern "C" int d3d_voxelize_3d_dense_staged(const float *points, int64_t n, int32_t c, const int32_t *shape, const float *bound,
                                            int32_t max_points, int32_t max_voxels, int32_t reduction, float *voxels,
                                            int64_t *coords, uint8_t *pmask, int32_t *npoints, float *aggregates,
                                            int64_t *counts, void *workspace, size_t workspace_bytes, void *stream,
                                            int64_t *host_counts, uint32_t flags, int32_t stage)
{
    if (stage < 0 || stage > 2) return D3D_ERR_BAD_ARG;
    if (n <= 0 && stage != 0) return D3D_ERR_UNSUPPORTED;
    return voxelize_dense_impl(points, n, c, shape, bound, max_points, max_voxels, reduction, voxels, coords, pmask, npoints,
                               aggregates, counts, workspace, workspace_bytes, stream, host_counts, flags, stage);
}

// The "voxel feature grid" without the dense [V,P,C] copy: first-seen voxel ids, counts, per-voxel
// reduction of all in-range points, point -> voxel map and each voxel's first point index.  Same grid
// semantics as d3d_voxelize_3d_dense (voxelize.cpp:100-101).  Used stand-alone ("dynamic voxelization")
// and as the per-rank stage of the point-sharded voxelizer (d3d_amd/voxel/sharded.py).
extern "C" size_t d3d_voxelize_reduce_rows(int64_t n) { return (size_t)carve(nullptr, 0, n > 0 ? n : 0, 0).npad + 4; }

extern "C" int d3d_voxelize_3d_reduce(const float *points, int64_t n, int32_t c, const int32_t *shape, const float *bound,
                                      int32_t reduction, int64_t index_offset, int64_t *coords, int32_t *npoints,
                                      float *aggregates, int64_t *first, int64_t *mapping, int64_t *keys,
                                      int32_t max_points, uint32_t *seg_base, float *rows,
                                      int64_t *counts, void *workspace, size_t workspace_bytes, void *stream, uint32_t flags)
{
    hipStream_t st = (hipStream_t)stream;
    if (flags & ~(uint32_t)D3D_VOXEL_FLAGS_ALL) return D3D_ERR_BAD_ARG;
    if (keys && n >= 0) D3D_HIP_CHECK(hipMemsetAsync(keys, 0xff, (size_t)(n + 1) * 8, st));   // -1 = no voxel in this row
    if (n < 0 || c < 3 || !shape || !bound || !counts) return D3D_ERR_BAD_ARG;
    if (n >= (1ll << 31) - kFlagTile) return D3D_ERR_BAD_ARG;
    if (reduction < D3D_REDUCE_MEAN || reduction > kReduceSum) return D3D_ERR_UNSUPPORTED;
    if (n > 0 && (!points || !npoints || !aggregates || (!coords && !keys))) return D3D_ERR_BAD_ARG;
    DenseKey kf;
    int rc = make_dense_key(shape, bound, kf);
    if (rc) return rc;
    VoxelWs w = carve(workspace, workspace_bytes, n, 0);
    if (!workspace || w.bytes > workspace_bytes) return D3D_ERR_WORKSPACE;
    if (max_points < 0 || ((seg_base != nullptr) != (rows != nullptr))) return D3D_ERR_BAD_ARG;
    // voxels up to P points are reduced sequentially in point order, larger ones cooperatively; with `rows` the first
    // min(count, P) rows of every voxel, in point order, are left in the caller's buffer at seg_base[voxel]
    const uint32_t P = max_points > 0 ? (uint32_t)max_points : 32u;
    const bool agg4 = (c == 4) && ((reinterpret_cast<uintptr_t>(points) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(aggregates) & 15) == 0);
    if (rows && (!agg4 || (reinterpret_cast<uintptr_t>(rows) & 15))) return D3D_ERR_UNSUPPORTED;    // staged rows are float4
    const float4 *p4 = reinterpret_cast<const float4 *>(points);
    uint32_t nbins = 0;
    int hshift = 0;
    if (agg4 && dense_cells_fit_u32(kf) && binned_eligible(n, w, flags, &nbins, &hshift)) {
        DenseOut d{P, 0xffffffffu, reduction, true, false, coords, npoints, nullptr, aggregates,
                   BinnedExtras{first, index_offset, keys, keys ? n : (int64_t)-1, nullptr, nullptr}, mapping};
        d.seg_out = seg_base;
        d.emit_reduce = P <= (uint32_t)kEmitCap && !(flags & D3D_VOXEL_SPLIT_FILL);
        // With `rows` on this path NO row is moved: the caller's buffer receives the voxels' ranked point INDICES (uint32; entry
        // seg_base[v] + k = the voxel's point of rank k >= 1, rank 0 = its first point) and counts[D3D_COUNT_AUX] = 1 says so --
        // d3d_owner_pack gathers the rows it sends from the points through them.  (Round 3 staged every row here: the bucket
        // kernel 45 instead of 31 us and k_meta_first 43 instead of k_emit's 22 us at config 5's shards.)
        if (rows && d.emit_reduce) {
            w.big_list = reinterpret_cast<uint32_t *>(rows);
            d.x.aux_value = 1;
        } else if (rows) w.staged = reinterpret_cast<float4 *>(rows);
        return binned_index<DenseKey, true>(kf, points, n, c, w, nbins, hshift, counts, d, st, !(flags & D3D_VOXEL_PARTITION_3PASS));
    }
    if (rows) w.staged = reinterpret_cast<float4 *>(rows);
    IndexOpts o{P, 0xffffffffu, first, index_offset, mapping, agg4};
    rc = dense_index(kf, points, n, c, w, counts, o, flags, st);
    if (rc) return rc;
    if (n == 0) return D3D_OK;
    if (agg4) {
        D3D_LAUNCH("k_meta", (k_meta<DenseKey, true>), dim3(grid_for(n, 256)), dim3(256), 0, st, kf, p4, counts, w.vinfo,
                   w.staged, w.unsorted, P, reduction, coords, npoints, seg_base ? seg_base : w.voff, (unsigned char *)nullptr,
                   reinterpret_cast<float4 *>(aggregates), w.big_list, w.big_count, keys, n);
        D3D_LAUNCH("k_overflow_reduce", k_overflow_reduce, dim3(512), dim3(256), 0, st, p4, w.vinfo, w.unsorted, w.big_list,
                   w.big_count, reduction, reinterpret_cast<float4 *>(aggregates));
    } else {
        D3D_LAUNCH("k_meta", (k_meta<DenseKey, false>), dim3(grid_for(n, 256)), dim3(256), 0, st, kf, p4, counts, w.vinfo,
                   w.staged, w.unsorted, P, reduction, coords, npoints, w.voff, (unsigned char *)nullptr, (float4 *)nullptr,
                   w.big_list, w.big_count, keys, n);
        D3D_LAUNCH("k_aggregate", k_aggregate, dim3(grid_for(n * c, 256)), dim3(256), 0, st, points, c, counts, npoints,
                   w.voff, w.list, w.unsorted, P, reduction, aggregates);
    }
    // (rows were STAGED on this path: counts[D3D_COUNT_AUX] -- the hash path's list-cell count until here -- must not read 1)
    if (rows) D3D_HIP_CHECK(hipMemsetAsync(counts + D3D_COUNT_AUX, 0, sizeof(int64_t), st));
    return D3D_OK;
}

static int voxelize_sparse_impl(const float *points, int64_t n, int32_t c, const float *voxel_size, int64_t *points_mapping,
                                int64_t *coords, int32_t *npoints, int64_t *counts, void *workspace, size_t workspace_bytes,
                                int64_t ws_nvox, void *stream, uint32_t flags, bool tolerant)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || c < 3 || !voxel_size || !counts) return D3D_ERR_BAD_ARG;
    if (flags & ~(uint32_t)D3D_VOXEL_FLAGS_ALL) return D3D_ERR_BAD_ARG;
    if (n > 0 && (!points || !points_mapping || !coords || !npoints)) return D3D_ERR_BAD_ARG;
    if (n >= (1ll << 31) - kFlagTile) return D3D_ERR_BAD_ARG;
    VoxelWs w = carve(workspace, workspace_bytes, n, ws_nvox);   // (the arrays used here do not move with ws_nvox)
    if (!workspace || w.bytes > workspace_bytes) return D3D_ERR_WORKSPACE;
    uint32_t nbins = 0;
    int hshift = 0;
    if (flags & D3D_VOXEL_WIDE_KEYS) {
        // any int32 coordinates (the caller repeats a call that came back with COORD_OVERFLOW): 96-bit compare in the table
        WideKey kf;
        for (int d = 0; d < 3; d++) kf.size[d] = voxel_size[d];
        TabWide tab{w.tabA, w.aux, w.tabB, points, (int)c, {voxel_size[0], voxel_size[1], voxel_size[2]}};
        int64_t *first = reinterpret_cast<int64_t *>(w.staged);          // [n] (free on the sparse contract)
        IndexOpts ow{0u, 0xffffffffu, first, 0, points_mapping, false};
        int rc = build_index(kf, tab, points, n, c, w, counts, ow, st);
        if (rc || n == 0) return rc;
        D3D_LAUNCH("k_wide_coords", k_wide_coords, dim3(grid_for(n, 256, (int64_t)1 << 30)), dim3(256), 0, st, kf, points, (int)c,
                   (const int64_t *)counts, (const int64_t *)first, (const uint4 *)w.vinfo, coords, npoints);
        return D3D_OK;
    }
    if (binned_eligible(n, w, flags, &nbins, &hshift)) {
        // up to 16 M points: partition by hash(cell) and index every bucket in LDS -- the 63-bit cell key itself is the
        // table key there, so no bounding box pass and no packed-slot limits
        SparseKey kf;
        for (int d = 0; d < 3; d++) kf.size[d] = voxel_size[d];
        kf.tolerant = tolerant;
        DenseOut d{0u, 0xffffffffu, D3D_REDUCE_NONE, false, false, coords, npoints, nullptr, nullptr,
                   BinnedExtras{nullptr, 0, nullptr, -1, nullptr, nullptr}, points_mapping};
        return binned_index<SparseKey, false>(kf, points, n, c, w, nbins, hshift, counts, d, st, !(flags & D3D_VOXEL_PARTITION_3PASS));
    }
    IndexOpts o{0u, 0xffffffffu, nullptr, 0, points_mapping, false};
    const int ib = bits_for((u64)(n > 1 ? n - 1 : 1));
    if (!(flags & D3D_VOXEL_PLAIN_SLOTS) && ib <= 40) {
        // one-word slots keyed inside the frame's bounding box; PACK_OVERFLOW (box too large for the word, or a
        // voxel with more points than the count field holds) -> the caller repeats the call with plain slots
        BoxKey kf;
        for (int d = 0; d < 3; d++) kf.size[d] = voxel_size[d];
        kf.tolerant = tolerant;
        kf.prm = reinterpret_cast<BoxParams *>(w.big_count + 16);
        kf.kb_max = 56 - ib;
        TabPacked tab{w.tabA, ib, 0, reinterpret_cast<uint32_t *>(w.tabB), kf.prm};
        int rc = build_index(kf, tab, points, n, c, w, counts, o, st);
        if (rc || n == 0) return rc;
        D3D_LAUNCH("k_meta", (k_meta<BoxKey, false>), dim3(grid_for(n, 256)), dim3(256), 0, st, kf, (const float4 *)nullptr,
                   counts, w.vinfo, w.staged, w.unsorted, 0u, 0, coords, npoints, (uint32_t *)nullptr,
                   (unsigned char *)nullptr, (float4 *)nullptr, w.big_list, w.big_count);
        return D3D_OK;
    }
    SparseKey kf;
    for (int d = 0; d < 3; d++) kf.size[d] = voxel_size[d];
    kf.tolerant = tolerant;
    TabPlain tab{w.tabA, w.tabB};
    int rc = build_index(kf, tab, points, n, c, w, counts, o, st);
    if (rc || n == 0) return rc;
    D3D_LAUNCH("k_meta", (k_meta<SparseKey, false>), dim3(grid_for(n, 256)), dim3(256), 0, st, kf, (const float4 *)nullptr,
               counts, w.vinfo, w.staged, w.unsorted, 0u, 0, coords, npoints, (uint32_t *)nullptr, (unsigned char *)nullptr,
               (float4 *)nullptr, w.big_list, w.big_count);
    return D3D_OK;
}

// order (descending stable argsort of voxel_npoints) for MAXVOX_DESCENDING, implemented in sort.hip
extern "C" int d3d_voxelize_3d_sparse(const float *points, int64_t n, int32_t c, const float *voxel_size,
                                      int64_t *points_mapping, int64_t *coords, int32_t *npoints, int64_t *counts,
                                      void *workspace, size_t workspace_bytes, void *stream, uint32_t flags)
{
    return voxelize_sparse_impl(points, n, c, voxel_size, points_mapping, coords, npoints, counts, workspace, workspace_bytes, 0,
                                stream, flags, false);
}

extern "C" int d3d_internal_argsort_desc_i32(const int32_t *keys, int64_t n, int32_t *order, void *ws, size_t ws_bytes,
                                             hipStream_t st);
extern "C" size_t d3d_internal_argsort_i32_bytes(int64_t n);

// nvox_device != NULL: `nvox` is only an upper bound (buffer rows), the number of voxels is read on the device
static int filter_impl(const float *feats, int64_t n, int32_t c, const int64_t *points_mapping, const int64_t *coords,
                       const int32_t *voxel_npoints, int64_t nvox, const int64_t *nvox_device, const int64_t *coords_bound,
                       int32_t min_points, int32_t max_points, int32_t max_voxels, int32_t max_points_filter,
                       int32_t max_voxels_filter, float *out_feats, int64_t *out_mask, int64_t *out_mapping,
                       int32_t *out_npoints, int64_t *out_coords, int64_t *counts, void *workspace, size_t workspace_bytes,
                       void *stream, int64_t *host_counts = nullptr, const int64_t *first_counts = nullptr)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || nvox < 0 || c < 1 || !coords_bound || !counts) return D3D_ERR_BAD_ARG;
    if (nvox_device && max_voxels_filter == D3D_MAXVOX_DESCENDING) return D3D_ERR_UNSUPPORTED;   // the sort needs the size
    if (max_points_filter == D3D_MAXPTS_FARTHEST_SAMPLING) return D3D_ERR_UNSUPPORTED;   // voxelize.cpp:469-471
    if (max_points_filter < 0 || max_points_filter > 2 || max_voxels_filter < 0 || max_voxels_filter > 2)
        return D3D_ERR_BAD_ARG;
    if (max_points < 0 || max_voxels < 0) return D3D_ERR_BAD_ARG;
    if (n >= (1ll << 31) - kScanTile || nvox >= (1ll << 31) - kFlagTile) return D3D_ERR_BAD_ARG;
    if (n > 0 && (!feats || !points_mapping || !out_feats || !out_mask || !out_mapping)) return D3D_ERR_BAD_ARG;
    if (nvox > 0 && (!coords || !voxel_npoints || !out_npoints || !out_coords)) return D3D_ERR_BAD_ARG;
    VoxelWs w = carve(workspace, workspace_bytes, n, nvox);
    if (!workspace || w.bytes > workspace_bytes) return D3D_ERR_WORKSPACE;

    const bool trim_pts = max_points_filter == D3D_MAXPTS_TRIM;
    const uint32_t P = trim_pts ? (uint32_t)max_points : 0xffffffffu;
    // index lists of the overflow voxels live in w.list (their counts sum to <= n); cursors start at zero
    D3D_LAUNCH("k_fill_u32", k_fill_u32, dim3(grid_for(trim_pts ? nvox : 0, 256)), dim3(256), 0, st, w.fcur,
               trim_pts ? nvox : (int64_t)0, 0u, counts);
    const bool rank_pts = trim_pts && n > 0 && nvox > 0 && max_points > 0;
    uint32_t *cellvox = w.parr;                                 // [n] voxel of each list cell
    unsigned char *trimmed = w.flags;                           // [n]
    if (rank_pts) {
        D3D_HIP_CHECK(hipMemsetAsync(cellvox, 0xff, (size_t)n * 4, st));
        D3D_HIP_CHECK(hipMemsetAsync(trimmed, 0, (size_t)n, st));
    }

    const int32_t *order = nullptr;
    if (max_voxels_filter == D3D_MAXVOX_DESCENDING && nvox > 0) {
        // stable descending argsort of the counts (reference: unstable torch::argsort, voxelize.cpp:406);
        // scratch: the hash-table region of the workspace (unused by the filter)
        int32_t *ord = reinterpret_cast<int32_t *>(w.pslot);   // npad >= ... only n entries guaranteed
        // pslot holds ceil(n/1024)*1024 entries; nvox may exceed n for hand-made inputs -> use table region
        size_t need = (size_t)nvox * sizeof(int32_t);
        char *tb = reinterpret_cast<char *>(w.tabA);
        size_t tbytes = w.tab_bytes;
        size_t sort_bytes = d3d_internal_argsort_i32_bytes(nvox);
        if (d3d_align_up(need) + sort_bytes > tbytes) return D3D_ERR_WORKSPACE;
        ord = reinterpret_cast<int32_t *>(tb);
        int rc = d3d_internal_argsort_desc_i32(voxel_npoints, nvox, ord, tb + d3d_align_up(need), sort_bytes, st);
        if (rc) return rc;
        order = ord;
    }

    FilterVoxels fv;
    fv.coords = coords; fv.npoints = voxel_npoints; fv.order = order; fv.nvox_device = nvox_device;
    for (int d = 0; d < 3; d++) { fv.lo[d] = coords_bound[2 * d]; fv.hi[d] = coords_bound[2 * d + 1]; }
    fv.min_points = min_points;
    fv.max_points = P;
    fv.max_voxels = max_voxels_filter == D3D_MAXVOX_NONE ? ~0ull : (unsigned long long)max_voxels;
    fv.newid = w.newid; fv.coff = w.coff; fv.out_coords = out_coords; fv.out_npoints = out_npoints;
    int rc = d3d_run_scan(fv, nvox, w.bsum, counts, D3D_COUNT_VOXELS, D3D_COUNT_AUX, fv.max_voxels, st);   // AUX = list cells
    if (rc) return rc;

    if (rank_pts) {
        D3D_LAUNCH("k_filter_scatter", k_filter_scatter, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, points_mapping,
                   n, nvox, voxel_npoints, w.newid, w.coff, P, w.fcur, w.list, cellvox, n);
        D3D_LAUNCH("k_filter_rank", k_filter_rank, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, (const int64_t *)counts,
                   (const uint32_t *)w.list, (const uint32_t *)cellvox, n, voxel_npoints, (const uint32_t *)w.coff, P, trimmed);
    }
    FilterPoints fp{feats, c, points_mapping, nvox, voxel_npoints, w.newid, trimmed, P,
                    reinterpret_cast<int32_t *>(w.pslot), out_feats, out_mask, out_mapping, false,
                    c == 4 && ((reinterpret_cast<uintptr_t>(feats) | reinterpret_cast<uintptr_t>(out_feats)) & 15) == 0};
    rc = d3d_run_scan(fp, n, w.bsum, counts, -1, D3D_COUNT_POINTS, ~0ull, st, host_counts, first_counts);
    return rc;
}

extern "C" int d3d_voxelize_3d_filter(const float *feats, int64_t n, int32_t c, const int64_t *points_mapping,
                                      const int64_t *coords, const int32_t *voxel_npoints, int64_t nvox,
                                      const int64_t *coords_bound, int32_t min_points, int32_t max_points,
                                      int32_t max_voxels, int32_t max_points_filter, int32_t max_voxels_filter,
                                      float *out_feats, int64_t *out_mask, int64_t *out_mapping, int32_t *out_npoints,
                                      int64_t *out_coords, int64_t *counts, void *workspace, size_t workspace_bytes,
                                      void *stream)
{
    return filter_impl(feats, n, c, points_mapping, coords, voxel_npoints, nvox, nullptr, coords_bound, min_points, max_points,
                       max_voxels, max_points_filter, max_voxels_filter, out_feats, out_mask, out_mapping, out_npoints,
                       out_coords, counts, workspace, workspace_bytes, stream);
}

extern "C" int d3d_voxelize_3d_filter_chained(const float *feats, int64_t n, int32_t c, const int64_t *points_mapping,
                                              const int64_t *coords, const int32_t *voxel_npoints, int64_t nvox_rows,
                                              const int64_t *sparse_counts, const int64_t *coords_bound,
                                              int32_t min_points, int32_t max_points, int32_t max_voxels,
                                              int32_t max_points_filter, int32_t max_voxels_filter, float *out_feats,
                                              int64_t *out_mask, int64_t *out_mapping, int32_t *out_npoints,
                                              int64_t *out_coords, int64_t *counts, void *workspace,
                                              size_t workspace_bytes, void *stream)
{
    if (!sparse_counts) return D3D_ERR_BAD_ARG;
    return filter_impl(feats, n, c, points_mapping, coords, voxel_npoints, nvox_rows, sparse_counts + D3D_COUNT_VOXELS,
                       coords_bound, min_points, max_points, max_voxels, max_points_filter, max_voxels_filter, out_feats,
                       out_mask, out_mapping, out_npoints, out_coords, counts, workspace, workspace_bytes, stream);
}

// VoxelGenerator.__call__'s sparse branch in one call (voxel/__init__.py:93-102): voxelize_sparse followed by
// voxelize_filter on its outputs, the voxel count staying on the device.  With the TRIM point filter the ranking "is this
// point among the first max_points of its voxel" (voxelize.cpp:457-463) is taken from the binned index, which has every
// voxel's indices in LDS anyway, instead of building and ranking index lists afterwards.  host_counts (optional, 2 *
// D3D_NUM_COUNTS + 1 int64 of host-mapped pinned memory, word [D3D_NUM_COUNTS] cleared by the caller): sparse_counts -> [0..4),
// counts -> [5..9), then flag [4] = 1, written BEFORE the compaction of the kept points is launched.
extern "C" int d3d_voxelize_3d_sparse_filter(const float *points, int64_t n, int32_t c, const float *voxel_size,
                                             const int64_t *coords_bound, int32_t min_points, int32_t max_points,
                                             int32_t max_voxels, int32_t max_points_filter, int32_t max_voxels_filter,
                                             int64_t *points_mapping, int64_t *coords, int32_t *npoints, int64_t *sparse_counts,
                                             float *out_feats, int64_t *out_mask, int64_t *out_mapping, int32_t *out_npoints,
                                             int64_t *out_coords, int64_t *counts, void *workspace, size_t workspace_bytes,
                                             void *stream, int64_t *host_counts, uint32_t flags, const int64_t *coord_offset)
{
    if (!sparse_counts || max_points < 0) return D3D_ERR_BAD_ARG;
    if (flags & ~(uint32_t)D3D_VOXEL_FLAGS_ALL) return D3D_ERR_BAD_ARG;
    const bool desc = max_voxels_filter == D3D_MAXVOX_DESCENDING;     // fused below; the two-operator form needs the count on the host
    // points outside the 3 x 21-bit key range (NaN / inf, |floor(p/size)| >= 2^20): the reference gives them a far-away
    // voxel (voxelize.cpp:309) that its coordinate-bound filter drops (:376-384).  With the bounds inside the key range the
    // same points are simply dropped here; only bounds reaching beyond it keep the COORD_OVERFLOW error.
    bool tolerant = coords_bound != nullptr;
    for (int k = 0; tolerant && k < 3; k++)
        tolerant = coords_bound[2 * k] >= -1048576 && coords_bound[2 * k + 1] <= 1048576;
    {
        // Binned index + voxel filter in one numbering: a voxel that fails the filter (coordinate bounds, min_points) gets no
        // first-point entry, so the first-seen numbering IS the filtered numbering (and its max_voxels cut the TRIM voxel
        // filter); the per-voxel outputs are written once, filtered; k_map_binned leaves every point's filtered voxel id
        // (or -1: no voxel / filtered voxel / trimmed point), which is all the point compaction needs.  The intermediate
        // sparse outputs (points_mapping, coords, npoints) are not materialised on this path; sparse_counts holds the
        // status bits (and the filtered voxel count).
        hipStream_t st = (hipStream_t)stream;
        const bool pf_ok = max_points_filter == D3D_MAXPTS_NONE || (max_points_filter == D3D_MAXPTS_TRIM && max_points > 0);
        const bool vf_ok = max_voxels_filter == D3D_MAXVOX_NONE || max_voxels_filter == D3D_MAXVOX_TRIM ||
                           (desc && points_mapping && coords && npoints);
        if (!(flags & D3D_VOXEL_WIDE_KEYS) && pf_ok && vf_ok && n > 0 && c >= 3 && n < (1ll << 31) - kFlagTile && points && voxel_size && coords_bound && counts &&
            out_feats && out_mask && out_mapping && out_npoints && out_coords && workspace && max_voxels >= 0) {
            VoxelWs w = carve(workspace, workspace_bytes, n, n);
            if (w.bytes > workspace_bytes) return D3D_ERR_WORKSPACE;
            uint32_t nbins = 0;
            int hshift = 0;
            if (binned_eligible(n, w, flags, &nbins, &hshift)) {
                const bool trim = max_points_filter == D3D_MAXPTS_TRIM;
                // round 5: three launches on 32-bit cells inside the coordinate bounds (not DESCENDING, whose sort sits between
                // the numbering and the compaction; not on request of round 4's kernels; boxes of 2^32 - 1 cells and more, or
                // reaching INT_MIN -- where the reference files its NaN points -- keep the 63-bit keys)
                if (!desc && !(flags & D3D_VOXEL_PARTITION_3PASS)) {
                    BoundKey bk;
                    double cells = 1.0;
                    bool ok = true;
                    for (int d = 0; d < 3; d++) {
                        const int64_t lo = coords_bound[2 * d], hi = coords_bound[2 * d + 1];
                        ok = ok && lo > (int64_t)INT_MIN && hi <= (int64_t)INT_MAX + 1 && hi > lo && hi - lo < (1ll << 32);
                        bk.size[d] = voxel_size[d];
                        bk.lo[d] = (long long)lo;
                        bk.ext[d] = ok ? (unsigned)(hi - lo) : 1u;
                        cells *= (double)(hi - lo);
                    }
                    if (ok && cells < 4294967295.0) {
                        VoxelPass vp{true, min_points, {0, 0, 0}, {0, 0, 0}};
                        for (int k = 0; k < 3; k++) { vp.lo[k] = coords_bound[2 * k]; vp.hi[k] = coords_bound[2 * k + 1]; }
                        const uint32_t vcap2 = max_voxels_filter == D3D_MAXVOX_NONE ? 0xffffffffu : (uint32_t)max_voxels;
                        int rc = sparse_fused_index(bk, points, n, c, w, nbins, hshift, vp, trim ? (uint32_t)max_points : 0u, vcap2,
                                                    trim ? (uint32_t)max_points : 0xffffffffu, coord_offset, out_feats, out_mask, out_mapping,
                                                    out_npoints, out_coords, sparse_counts, counts, host_counts, st);
                        if (rc != D3D_ERR_UNSUPPORTED) return rc;
                    }
                }
                SparseKey kf;
                for (int d = 0; d < 3; d++) kf.size[d] = voxel_size[d];
                kf.tolerant = tolerant;
                // DESCENDING (voxelize.cpp:404-420): first-seen numbering of ALL passing voxels into scratch rows, then a stable
                // sort of their counts decides the ranks, the first max_voxels of which are the result (k_desc_finish)
                const uint32_t vcap = (max_voxels_filter == D3D_MAXVOX_NONE || desc) ? 0xffffffffu : (uint32_t)max_voxels;
                DenseOut d{trim ? (uint32_t)max_points : 0u, vcap, D3D_REDUCE_NONE, false, false, desc ? coords : out_coords,
                           desc ? npoints : out_npoints, nullptr, nullptr, BinnedExtras{nullptr, 0, nullptr, -1, nullptr, nullptr}, nullptr};
                int32_t *desc_keys = reinterpret_cast<int32_t *>(w.big_list), *desc_order = reinterpret_cast<int32_t *>(w.parr);
                if (desc) { d.count_out = desc_keys; d.first_out = points_mapping; }
                d.pass.on = true;
                d.pass.min_points = min_points;
                for (int k = 0; k < 3; k++) { d.pass.lo[k] = coords_bound[2 * k]; d.pass.hi[k] = coords_bound[2 * k + 1]; }
                d.npoints_clamp = trim ? (uint32_t)max_points : 0xffffffffu;
                // the compaction maps the points itself, in point order: k_bucket_index leaves every kept point's voxel (its
                // first point) by point index, trimmed points and filtered voxels already taken out (k_map_binned as a launch of
                // its own: 21 us of kernel, 173 vs 158 us per call)
                d.map_later = true;
                d.early_host = desc ? nullptr : host_counts;     // output sizes to the host right after the numbering (DESCENDING:
                                                                 // they are known after the sort; k_compact_kept's last tile tells)
                d.coord_offset = coord_offset;
                d.compact_stat = w.bsum;                // look-back words of k_compact_kept: k_meta_first clears them
                d.compact_tiles = (uint32_t)d3d_divup(n, kCompactTile);
                int rc = binned_index<SparseKey, false>(kf, points, n, c, w, nbins, hshift, sparse_counts, d, st, !(flags & D3D_VOXEL_PARTITION_3PASS));
                if (rc) return rc;
                if (desc) {
                    // (one counting pass on min(count, 255) + the few larger ones ranked among themselves: sort.hip)
                    const size_t sort_bytes = d3d_internal_argsort_counts_bytes(n);
                    if (sort_bytes > w.tab_bytes) return D3D_ERR_WORKSPACE;
                    rc = d3d_internal_argsort_desc_counts_dev(desc_keys, n, sparse_counts + D3D_COUNT_VOXELS, n, desc_order, w.tabA,
                                                              w.tab_bytes, st);
                    if (rc) return rc;
                    D3D_LAUNCH("k_desc_finish", k_desc_finish, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, (const int32_t *)desc_order,
                               sparse_counts, (uint32_t)max_voxels, (const int64_t *)coords, (const int32_t *)npoints,
                               (const int64_t *)points_mapping, out_coords, out_npoints, w.voff, sparse_counts + D3D_COUNT_AUX);
                }
                // (both output sizes went to the host from k_meta_first_lb's last tile, before this launch; DESCENDING: from this one's)
                const int vix = desc ? D3D_COUNT_AUX : D3D_COUNT_VOXELS;
                int64_t *late_host = desc ? host_counts : nullptr;
                const bool v4 = c == 4 && ((reinterpret_cast<uintptr_t>(points) | reinterpret_cast<uintptr_t>(out_feats)) & 15) == 0;
                unsigned int *cticket = w.big_count + 41;
                if (v4)
                    D3D_LAUNCH("k_compact_kept", k_compact_kept<true>, dim3(d.compact_tiles), dim3(kCompactThreads), 0, st, points, (int)c, n,
                               w.npad, (const uint32_t *)w.pslot, (const uint32_t *)w.voff, out_feats, out_mask, out_mapping, w.bsum, cticket,
                               counts, (const int64_t *)sparse_counts, late_host, vix);
                else
                    D3D_LAUNCH("k_compact_kept", k_compact_kept<false>, dim3(d.compact_tiles), dim3(kCompactThreads), 0, st, points, (int)c, n,
                               w.npad, (const uint32_t *)w.pslot, (const uint32_t *)w.voff, out_feats, out_mask, out_mapping, w.bsum, cticket,
                               counts, (const int64_t *)sparse_counts, late_host, vix);
                return D3D_OK;
            }
        }
    }
    if (desc) return D3D_ERR_UNSUPPORTED;      // (the two-operator form sorts on a host-side count: the caller runs the two calls)
    // other filter combinations / sizes: the two operators one after the other, the voxel count staying on the device
    int rc = voxelize_sparse_impl(points, n, c, voxel_size, points_mapping, coords, npoints, sparse_counts, workspace,
                                  workspace_bytes, n, stream, flags, tolerant);
    if (rc) return rc;
    rc = filter_impl(points, n, c, points_mapping, coords, npoints, n, sparse_counts + D3D_COUNT_VOXELS, coords_bound,
                     min_points, max_points, max_voxels, max_points_filter, max_voxels_filter, out_feats, out_mask,
                     out_mapping, out_npoints, out_coords, counts, workspace, workspace_bytes, stream, host_counts,
                     sparse_counts);
    if (rc == D3D_OK && coord_offset && n > 0)
        D3D_LAUNCH("k_sub_offset", k_sub_offset, dim3(grid_for(n * 3, 256)), dim3(256), 0, (hipStream_t)stream, out_coords,
                   (const int64_t *)counts, n, (long long)coord_offset[0], (long long)coord_offset[1], (long long)coord_offset[2]);
    return rc;
}

// d3d_voxelize_3d_sparse_filter through one prepared argument block (include/d3d_hip.h, D3DSparseFilterCall)
static size_t call_front_bytes(int64_t n)
{
    const size_t rows = (size_t)(n > 0 ? n : 1);
    return d3d_align_up(2 * D3D_NUM_COUNTS * 8) + d3d_align_up(rows * 8) + d3d_align_up(rows * 24) + d3d_align_up(rows * 4);
}

extern "C" size_t d3d_voxelize_3d_sparse_filter_call_layout(int64_t n, int32_t c, size_t *offsets5)
{
    const size_t rows = (size_t)(n > 0 ? n : 1), cc = (size_t)(c > 0 ? c : 1);
    const size_t sizes[5] = {rows * cc * 4, rows * 8, rows * 8, rows * 4, rows * 24};
    size_t at = 0;
    for (int k = 0; k < 5; k++) {
        if (offsets5) offsets5[k] = at;
        at += d3d_align_up(sizes[k]);
    }
    return at;
}

extern "C" size_t d3d_voxelize_3d_sparse_filter_call_workspace_bytes(int64_t n)
{
    return call_front_bytes(n) + d3d_voxelize_workspace_bytes(n, n);
}

extern "C" int d3d_voxelize_3d_sparse_filter_call(const D3DSparseFilterCall *a)
{
    if (!a || a->n < 0) return D3D_ERR_BAD_ARG;
    size_t off[5];
    const size_t need = d3d_voxelize_3d_sparse_filter_call_layout(a->n, a->c, off), front = call_front_bytes(a->n);
    if (!a->outputs || a->outputs_bytes < need) return D3D_ERR_BAD_ARG;
    if (!a->workspace || a->workspace_bytes < front + d3d_voxelize_workspace_bytes(a->n, a->n)) return D3D_ERR_WORKSPACE;
    char *w = static_cast<char *>(a->workspace), *o = static_cast<char *>(a->outputs);
    const size_t rows = (size_t)(a->n > 0 ? a->n : 1);
    int64_t *counts2 = reinterpret_cast<int64_t *>(w);
    int64_t *mapping = reinterpret_cast<int64_t *>(w + d3d_align_up(2 * D3D_NUM_COUNTS * 8));
    int64_t *coords = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(mapping) + d3d_align_up(rows * 8));
    int32_t *npoints = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(coords) + d3d_align_up(rows * 24));
    return d3d_voxelize_3d_sparse_filter(a->points, a->n, a->c, a->voxel_size, a->coords_bound, a->min_points, a->max_points, a->max_voxels,
                                         a->max_points_filter, a->max_voxels_filter, mapping, coords, npoints, counts2,
                                         reinterpret_cast<float *>(o + off[0]), reinterpret_cast<int64_t *>(o + off[1]),
                                         reinterpret_cast<int64_t *>(o + off[2]), reinterpret_cast<int32_t *>(o + off[3]),
                                         reinterpret_cast<int64_t *>(o + off[4]), counts2 + D3D_NUM_COUNTS, w + front,
                                         a->workspace_bytes - front, a->stream, a->host_counts, a->flags,
                                         a->has_coord_offset ? a->coord_offset : nullptr);
}

#ifdef D3D_TUNE
extern "C" void d3d_debug_set_tune(int k, int v) { if (k >= 0 && k < 16) g_d3d_tune[k] = v; }
#endif
#ifdef D3D_PHASE_CLOCKS
// diagnostic build: read (and clear) the phase clocks -- out[4][16] u64 host array
extern "C" int d3d_debug_phase_clocks(unsigned long long *out)
{
    D3D_HIP_CHECK(hipDeviceSynchronize());
    D3D_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 64));
    unsigned long long zero[64] = {0};
    D3D_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zero, sizeof(zero)));
    return D3D_OK;
}
#endif
