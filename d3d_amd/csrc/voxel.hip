// voxel.hip -- point cloud -> voxels on MI355X (gfx950).  Replaces the sequential
// std::unordered_map loops of the reference (d3d/voxel/voxelize.cpp) with:
//
//   insert   one point per lane, coalesced float4 loads, open-addressing hash table in
//            HBM (16-byte slots {key, first_point, count}); count by atomicAdd, the
//            voxel's first point index by atomicMin
//   number   a point is "first" iff slot.first == its index; an exclusive scan of the
//            first-flags over the point array reproduces the reference's first-seen
//            voxel numbering (voxelize.cpp:119,317) with no sort
//   rank     per-voxel list of the max_points smallest point indices, built by an
//            atomicMin insertion chain (order independent -> exact, no sort)
//   fill     streaming write of voxels[V,P,C] (the HBM-roofline kernel), pmask,
//            aggregates (sequential in point order for voxels that fit -> bit-exact MEAN)
//
// Build: hipcc --offload-arch=gfx950 -ffp-contract=off (IEEE div, no FMA contraction:
// voxel coordinates must round exactly like the reference's CPU code).
#include "common.hpp"
#include <stdlib.h>

namespace {

constexpr unsigned long long kEmptyKey = ~0ull;
constexpr uint32_t kInf = 0xffffffffu;        // empty list cell / "no first point yet"
constexpr uint32_t kNoVoxel = 0xffffffffu;    // slot.first after numbering: voxel dropped by max_voxels
constexpr uint32_t kNoSlot = 0x7fffffffu;     // pslot: point not in any voxel
constexpr uint32_t kFirstBit = 0x80000000u;   // pslot: this point is the first of its voxel

// 16-byte hash slot.  Two layouts (template parameter PK):
//   plain  : key | first | cnt            voxel id overwrites `first` in the numbering pass
//   packed : (key << 24 | cnt) | first | vid    -- claim + count in ONE 64-bit atomic (CAS for the
//            first arrival, atomicAdd afterwards); usable when key < 2^40-1 and n < 2^24
struct __attribute__((aligned(16))) Slot {
    unsigned long long key;
    uint32_t first;   // min point index
    uint32_t cnt;     // plain: count; packed: voxel id
};
constexpr int kReduceSum = 4;   // internal: MEAN without the division (per-rank partial of the sharded voxelizer)
constexpr int kCntBits = 24;
constexpr uint32_t kCntMask = (1u << kCntBits) - 1u;
template <bool PK> __device__ __forceinline__ uint32_t slot_cnt(const uint4 &s) { return PK ? (s.x & kCntMask) : s.w; }
template <bool PK> __device__ __forceinline__ uint32_t slot_vid(const uint4 &s) { return PK ? s.w : s.z; }
template <bool PK> __device__ __forceinline__ unsigned long long slot_key(const uint4 &s)
{
    unsigned long long k = ((unsigned long long)s.y << 32) | s.x;
    return PK ? (k >> kCntBits) : k;
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long h)
{
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return h;
}

// ------------------------------------------------------------------ coordinate keys
// dense contract: idx = int((p - lo) / size), 0 <= idx < shape   (voxelize.cpp:100-101)
struct DenseKey {
    float lo[3], size[3];
    int shape[3];
    __device__ __forceinline__ bool make(const float *p, unsigned long long &key, uint32_t &status) const
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
        key = ((unsigned long long)c[0] * (unsigned)shape[1] + (unsigned)c[1]) * (unsigned)shape[2] + (unsigned)c[2];
        return true;
    }
    __device__ __forceinline__ void decode(unsigned long long key, long long *c) const
    {
        c[2] = (long long)(key % (unsigned)shape[2]); key /= (unsigned)shape[2];
        c[1] = (long long)(key % (unsigned)shape[1]);
        c[0] = (long long)(key / (unsigned)shape[1]);
    }
};

// sparse contract: coord = floor(p / size), unbounded (voxelize.cpp:309); 3 x 21-bit packing
struct SparseKey {
    float size[3];
    __device__ __forceinline__ bool make(const float *p, unsigned long long &key, uint32_t &status) const
    {
        unsigned long long k = 0;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            float q = floorf(p[d] / size[d]);
            if (!(q >= -1048576.0f && q < 1048576.0f)) { status |= D3D_VOXEL_STATUS_COORD_OVERFLOW; return false; }
            k = (k << 21) | (unsigned long long)(unsigned)((int)q + 1048576);
        }
        key = k;
        return true;
    }
    __device__ __forceinline__ void decode(unsigned long long key, long long *c) const
    {
        c[2] = (long long)(key & 0x1fffff) - 1048576;
        c[1] = (long long)((key >> 21) & 0x1fffff) - 1048576;
        c[0] = (long long)((key >> 42) & 0x1fffff) - 1048576;
    }
};

// ------------------------------------------------------------------ kernels
__global__ void k_init(Slot *table, int64_t cap, uint32_t *list, int64_t nlist, int64_t *counts, uint32_t aux_init)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 e = make_uint4(0xffffffffu, 0xffffffffu, kInf, aux_init);
    uint4 *tb = reinterpret_cast<uint4 *>(table);
    for (int64_t i = t0; i < cap; i += stride) tb[i] = e;
    uint4 f = make_uint4(kInf, kInf, kInf, kInf);
    uint4 *l4 = reinterpret_cast<uint4 *>(list);
    for (int64_t i = t0; i < nlist / 4; i += stride) l4[i] = f;   // nlist is padded to a multiple of 4
    if (t0 < D3D_NUM_COUNTS) counts[t0] = 0;
}

template <class Key, bool VEC4, bool PK>
__global__ __launch_bounds__(256) void k_insert(Key kf, const float *__restrict__ points, int64_t n, int c,
                                                Slot *table, unsigned long long mask, uint32_t *pslot,
                                                uint32_t *parr, int64_t npad, int64_t *counts)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    uint32_t slot = kNoSlot, status = 0, arrival = 0;
    if (i < n) {
        float p[3];
        if (VEC4) {
            float4 v = reinterpret_cast<const float4 *>(points)[i];
            p[0] = v.x; p[1] = v.y; p[2] = v.z;
        } else {
            const float *src = points + i * c;
            p[0] = src[0]; p[1] = src[1]; p[2] = src[2];
        }
        unsigned long long key;
        if (kf.make(p, key, status)) {
            unsigned long long h = mix64(key) & mask;
            bool found = false;
            for (unsigned long long probe = 0; probe <= mask; probe++) {
                Slot *s = &table[h];
                unsigned long long k = __hip_atomic_load(&s->key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (PK) {
                    if (k == kEmptyKey) {
                        unsigned long long old = atomicCAS(&s->key, kEmptyKey, (key << kCntBits) | 1ull);
                        if (old == kEmptyKey) { found = true; arrival = 0; break; }   // claimed and counted at once
                        k = old;
                    }
                    if ((k >> kCntBits) == key) {
                        arrival = (uint32_t)(atomicAdd(&s->key, 1ull) & kCntMask);
                        found = true;
                        break;
                    }
                } else {
                    if (k == kEmptyKey) {
                        unsigned long long old = atomicCAS(&s->key, kEmptyKey, key);
                        k = (old == kEmptyKey) ? key : old;
                    }
                    if (k == key) {
                        arrival = atomicAdd(&s->cnt, 1u);   // arrival position inside the voxel (any order)
                        found = true;
                        break;
                    }
                }
                h = (h + 1) & mask;
            }
            if (found) {
                slot = (uint32_t)h;
                // `first` only decreases, so a stale read can only cause a redundant atomicMin
                if (__hip_atomic_load(&table[h].first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (uint32_t)i)
                    atomicMin(&table[h].first, (uint32_t)i);
            } else status |= D3D_VOXEL_STATUS_TABLE_FULL;
        }
    }
    pslot[i] = slot;
    if (parr) parr[i] = arrival;
    if (status) atomicOr(reinterpret_cast<unsigned long long *>(&counts[D3D_COUNT_STATUS]), (unsigned long long)status);
}

// scan functor: voxel numbering in first-seen order (+ list offsets)
template <class Key, bool PK>
struct NumberVoxels {
    static constexpr const char *kName = "k_scan_count<NumberVoxels>", *kName2 = "k_scan_apply<NumberVoxels>";
    Key kf;
    Slot *table;
    uint32_t *pslot;
    uint32_t *voff;       // [cap_voxels] list offset per voxel
    int64_t *coords;      // [cap_voxels,3]
    int32_t *npoints;     // [cap_voxels]
    uint32_t max_points;  // 0 -> no lists (sparse contract)
    uint32_t max_voxels;
    int64_t *first_out;   // optional [cap_voxels]: global index of each voxel's first point
    int64_t index_offset; // added to point indices in first_out (rank shard offset)

    __device__ __forceinline__ unsigned long long value(int64_t i) const
    {
        uint32_t ps = pslot[i];
        if (ps == kNoSlot) return 0;
        uint4 s = reinterpret_cast<const uint4 *>(table)[ps];
        if (s.z != (uint32_t)i) return 0;
        pslot[i] = ps | kFirstBit;
        uint32_t w = max_points ? slot_cnt<PK>(s) : 0u;     // list space: every point of the voxel gets a cell
        return (1ull << 32) | w;
    }
    __device__ __forceinline__ unsigned long long value2(int64_t i) const
    {
        uint32_t ps = pslot[i];
        if (!(ps & kFirstBit)) return 0;
        uint32_t w = max_points ? slot_cnt<PK>(reinterpret_cast<const uint4 *>(table)[ps & ~kFirstBit]) : 0u;
        return (1ull << 32) | w;
    }
    __device__ __forceinline__ void apply(int64_t i, unsigned long long v, unsigned long long excl) const
    {
        if (!v) return;
        Slot *sp = &table[pslot[i] & ~kFirstBit];
        const uint4 s = *reinterpret_cast<const uint4 *>(sp);
        uint32_t vid = (uint32_t)(excl >> 32);
        uint32_t *vid_field = PK ? &sp->cnt : &sp->first;
        if (vid >= max_voxels) { *vid_field = kNoVoxel; return; }   // voxelize.cpp:116-117
        *vid_field = vid;
        if (max_points) {
            voff[vid] = (uint32_t)excl;
            if (PK) sp->first = (uint32_t)excl;   // `first` has done its job: reuse it for the list offset
        }
        long long cc[3];
        kf.decode(slot_key<PK>(s), cc);
        coords[(int64_t)vid * 3 + 0] = cc[0];
        coords[(int64_t)vid * 3 + 1] = cc[1];
        coords[(int64_t)vid * 3 + 2] = cc[2];
        npoints[vid] = (int32_t)slot_cnt<PK>(s);
        if (first_out) first_out[vid] = index_offset + i;
    }
};

// monotone float <-> uint map for atomicMax/atomicMin on floats
__device__ __forceinline__ uint32_t enc_f32(float f)
{
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(uint32_t e)
{
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

// Ranking without a sort and without dependent atomic chains:
//   k_scatter  every point drops its index at unsorted[voff[v] + arrival]  (arrival = value returned by
//              the count atomicAdd in k_insert; any order)
//   k_select   every point counts the indices smaller than its own in its voxel's (contiguous, L2-hot)
//              segment; that count IS its rank in point order.  It stops as soon as max_points smaller
//              ones were seen (the point is then not among the first max_points, voxelize.cpp:128-134),
//              so a voxel of c points costs O(c * max_points) loads when arrival order is roughly
//              index order.  Ranks < max_points land in sorted[voff[v] + rank].
template <bool PK>
__global__ __launch_bounds__(256) void k_scatter(int64_t n, const Slot *table, uint32_t *__restrict__ pslot,
                                                 uint32_t *__restrict__ parr, const uint32_t *__restrict__ voff,
                                                 uint32_t *unsorted, uint32_t *sorted)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t ps = pslot[i] & ~kFirstBit;
    uint32_t todo = 0, base = 0;
    if (ps != kNoSlot) {
        uint4 s = reinterpret_cast<const uint4 *>(table)[ps];     // the one random access of this pass
        const uint32_t vid = slot_vid<PK>(s), cnt = slot_cnt<PK>(s);
        if (vid != kNoVoxel) {
            base = PK ? s.z : voff[vid];     // packed slots carry the list offset in `first` after numbering
            if (cnt == 1) sorted[base] = (uint32_t)i;
            else { unsorted[base + parr[i]] = (uint32_t)i; todo = cnt; }
        }
    }
    // hand (segment length, segment base) to k_select through the per-point arrays: coalesced there
    pslot[i] = todo;
    parr[i] = base;
}

__global__ __launch_bounds__(256) void k_select(int64_t n, const uint32_t *__restrict__ pcnt,
                                                const uint32_t *__restrict__ pbase,
                                                const uint32_t *__restrict__ unsorted, uint32_t *sorted,
                                                uint32_t max_points)
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
    if (rank < max_points) sorted[base + rank] = me;
}

// voxels[V, P, 4]: one 16-byte row per lane, grid-stride, non-temporal stores
__global__ __launch_bounds__(256) void k_fill_c4(const float4 *__restrict__ points, const int64_t *__restrict__ counts,
                                                 const int32_t *__restrict__ npoints, const uint32_t *__restrict__ voff,
                                                 const uint32_t *__restrict__ list, uint32_t max_points, float4 *voxels)
{
    const int64_t rows = counts[D3D_COUNT_VOXELS] * (int64_t)max_points;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
        const int64_t v = r / max_points;
        const uint32_t k = (uint32_t)(r - v * max_points);
        const uint32_t cnt = (uint32_t)npoints[v];
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < cnt) val = points[list[voff[v] + k]];
        __builtin_nontemporal_store(val.x, &voxels[r].x);
        __builtin_nontemporal_store(val.y, &voxels[r].y);
        __builtin_nontemporal_store(val.z, &voxels[r].z);
        __builtin_nontemporal_store(val.w, &voxels[r].w);
    }
}

// generic C: one float per lane
__global__ __launch_bounds__(256) void k_fill_generic(const float *__restrict__ points, int c,
                                                      const int64_t *__restrict__ counts,
                                                      const int32_t *__restrict__ npoints,
                                                      const uint32_t *__restrict__ voff,
                                                      const uint32_t *__restrict__ list, uint32_t max_points,
                                                      float *voxels)
{
    const int64_t pc = (int64_t)max_points * c;
    const int64_t total = counts[D3D_COUNT_VOXELS] * pc;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t v = e / pc;
        const uint32_t rem = (uint32_t)(e - v * pc);
        const uint32_t k = rem / (uint32_t)c, d = rem - k * (uint32_t)c;
        float val = 0.f;
        if (k < (uint32_t)npoints[v]) val = points[(int64_t)list[voff[v] + k] * c + d];
        voxels[e] = val;
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

template <bool PK>
__global__ __launch_bounds__(256) void k_map(const Slot *table, const uint32_t *__restrict__ pslot, int64_t n,
                                             int64_t *mapping)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t ps = pslot[i] & ~kFirstBit;
    long long m = -1;
    if (ps != kNoSlot) {
        uint32_t vid = slot_vid<PK>(reinterpret_cast<const uint4 *>(table)[ps]);
        if (vid != kNoVoxel) m = (long long)vid;
    }
    mapping[i] = m;
}

// ------------------------------------------------------------------ filter (direct-addressed by voxel id)
struct FilterVoxels {
    static constexpr const char *kName = "k_scan_count<FilterVoxels>", *kName2 = "k_scan_apply<FilterVoxels>";
    const int64_t *coords;
    const int32_t *npoints;
    const int32_t *order;     // null = id order; else voxel id per rank (descending count)
    long long lo[3], hi[3];
    int32_t min_points;
    uint32_t max_points;      // P for TRIM, 0xffffffff for NONE
    unsigned long long max_voxels;
    int32_t *newid;           // [nvox]
    uint32_t *coff;           // [nvox] chain cell offset of overflow voxels
    int64_t *out_coords;
    int32_t *out_npoints;

    __device__ __forceinline__ unsigned long long value(int64_t k) const
    {
        const int64_t v = order ? order[k] : k;
        const int32_t cnt = npoints[v];
        bool ok = cnt >= min_points;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            long long x = coords[v * 3 + d];
            ok = ok && x >= lo[d] && x < hi[d];
        }
        if (!ok) return 0;
        return (1ull << 32) | ((uint32_t)cnt > max_points ? max_points : 0u);
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

// TRIM: per kept overflow voxel keep the max_points smallest point indices (cells chain)
__global__ __launch_bounds__(256) void k_filter_rank(const int64_t *__restrict__ mapping, int64_t n, int64_t nvox,
                                                     const int32_t *__restrict__ npoints,
                                                     const int32_t *__restrict__ newid, const uint32_t *__restrict__ coff,
                                                     uint32_t max_points, uint32_t *cells, int64_t ncells)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t v = mapping[i];
    if (v < 0 || v >= nvox || newid[v] < 0) return;
    if ((uint32_t)npoints[v] <= max_points) return;
    const uint32_t base = coff[v];
    if ((int64_t)base + max_points > ncells) return;   // inconsistent voxel_npoints: cannot trim (see keep())
    uint32_t last = __hip_atomic_load(&cells[base + max_points - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (last < (uint32_t)i) return;
    uint32_t x = (uint32_t)i;
    for (uint32_t k = 0; k < max_points; k++) {
        uint32_t old = atomicMin(&cells[base + k], x);
        if (old == kInf) break;
        if (old > x) x = old;
    }
}

struct FilterPoints {
    static constexpr const char *kName = "k_scan_count<FilterPoints>", *kName2 = "k_scan_apply<FilterPoints>";
    const float *feats;
    int c;
    const int64_t *mapping;
    int64_t nvox;
    const int32_t *npoints;
    const int32_t *newid;
    const uint32_t *coff;
    const uint32_t *cells;
    int64_t ncells;
    uint32_t max_points;      // 0xffffffff for NONE
    float *out_feats;
    int64_t *out_mask, *out_mapping;

    __device__ __forceinline__ int32_t keep(int64_t i) const
    {
        const int64_t v = mapping[i];
        if (v < 0 || v >= nvox) return -1;
        const int32_t id = newid[v];
        if (id < 0) return -1;
        if ((uint32_t)npoints[v] > max_points) {
            // voxelize.cpp:457-463: first max_points points of the voxel in point order
            if (max_points == 0) return -1;
            const uint32_t base = coff[v];
            if ((int64_t)base + max_points <= ncells && (uint32_t)i > cells[base + max_points - 1]) return -1;
        }
        return id;
    }
    __device__ __forceinline__ unsigned long long value(int64_t i) const { return keep(i) >= 0 ? 1ull : 0ull; }
    __device__ __forceinline__ unsigned long long value2(int64_t i) const { return value(i); }
    __device__ __forceinline__ void apply(int64_t i, unsigned long long val, unsigned long long excl) const
    {
        if (!val) return;
        out_mask[excl] = i;
        out_mapping[excl] = keep(i);
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

// ------------------------------------------------------------------ workspace layout
struct VoxelWs {
    Slot *table;
    unsigned long long cap;
    uint32_t *pslot;
    uint32_t *parr;      // arrival position of each point inside its voxel
    uint32_t *list;      // per-voxel segments: point indices sorted ascending (first max_points valid)
    uint32_t *unsorted;  // per-voxel segments in arrival order
    uint32_t *voff;
    unsigned long long *bsum;
    int32_t *newid;
    uint32_t *coff;
    size_t bytes;
};

static unsigned long long table_capacity(int64_t n)
{
    unsigned long long cap = 1024;
    while (cap < (unsigned long long)n * 2ull) cap <<= 1;
    return cap;
}

static VoxelWs carve(void *ws, size_t ws_bytes, int64_t n, int64_t nvox)
{
    WsCarver w(ws, ws_bytes);
    VoxelWs r;
    const int64_t npad = d3d_divup(n > 0 ? n : 1, kScanTile) * kScanTile;
    const int64_t m = n > nvox ? n : nvox;
    r.cap = table_capacity(n);
    r.table = w.take<Slot>(r.cap);
    r.pslot = w.take<uint32_t>(npad);
    r.parr = w.take<uint32_t>(npad);
    r.list = w.take<uint32_t>(npad + 4);
    r.unsorted = w.take<uint32_t>(npad + 4);
    r.voff = w.take<uint32_t>(npad + 4);
    r.bsum = w.take<unsigned long long>(d3d_divup(m > 0 ? m : 1, kScanTile) + 1);
    r.newid = w.take<int32_t>(nvox > 0 ? nvox : 1);
    r.coff = w.take<uint32_t>(nvox > 0 ? nvox : 1);
    r.bytes = w.off;
    return r;
}

static inline unsigned grid_for(int64_t work, int block, int64_t maxblocks = 256 * 16)
{
    int64_t g = d3d_divup(work > 0 ? work : 1, block);
    return (unsigned)(g < maxblocks ? g : maxblocks);
}

template <class Key, bool PK>
static int build_table(const Key &kf, const float *points, int64_t n, int c, const VoxelWs &w, int64_t *counts,
                       uint32_t max_points, hipStream_t st)
{
    const int64_t npad = d3d_divup(n > 0 ? n : 1, kScanTile) * kScanTile;
    D3D_LAUNCH("k_init", k_init, dim3(grid_for((int64_t)w.cap, 256)), dim3(256), 0, st, w.table, (int64_t)w.cap, w.list,
               (int64_t)0, counts, PK ? kNoVoxel : 0u);
    const bool vec4 = (c == 4) && ((reinterpret_cast<uintptr_t>(points) & 15) == 0);
    dim3 grid((unsigned)d3d_divup(npad, 256));
    uint32_t *parr = max_points ? w.parr : nullptr;
    if (vec4)
        D3D_LAUNCH("k_insert", (k_insert<Key, true, PK>), grid, dim3(256), 0, st, kf, points, n, c, w.table, w.cap - 1,
                   w.pslot, parr, npad, counts);
    else
        D3D_LAUNCH("k_insert", (k_insert<Key, false, PK>), grid, dim3(256), 0, st, kf, points, n, c, w.table, w.cap - 1,
                   w.pslot, parr, npad, counts);
    return D3D_OK;
}

// table + first-seen numbering + per-voxel sorted index lists for the dense contract
template <bool PK>
static int dense_index(const DenseKey &kf, const float *points, int64_t n, int c, const VoxelWs &w, int64_t *counts,
                       int64_t *coords, int32_t *npoints, uint32_t max_points, uint32_t max_voxels, hipStream_t st,
                       int64_t *first_out = nullptr, int64_t index_offset = 0, int64_t *mapping = nullptr)
{
    int rc = build_table<DenseKey, PK>(kf, points, n, c, w, counts, max_points, st);
    if (rc) return rc;
    NumberVoxels<DenseKey, PK> nv{kf, w.table, w.pslot, w.voff, coords, npoints, max_points, max_voxels,
                                  first_out, index_offset};
    rc = d3d_run_scan(nv, n, w.bsum, counts, D3D_COUNT_VOXELS, D3D_COUNT_AUX, (unsigned long long)max_voxels, st);
    if (rc) return rc;
    if (mapping && n > 0)   // before k_scatter recycles pslot
        D3D_LAUNCH("k_map", k_map<PK>, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, w.table, w.pslot, n, mapping);
    if (n == 0 || max_voxels == 0 || max_points == 0) return D3D_OK;
    D3D_LAUNCH("k_scatter", k_scatter<PK>, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, n, w.table, w.pslot,
               w.parr, w.voff, w.unsorted, w.list);
    D3D_LAUNCH("k_select", k_select, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, n, w.pslot, w.parr,
               w.unsorted, w.list, max_points);
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

extern "C" int d3d_voxelize_3d_dense(const float *points, int64_t n, int32_t c, const int32_t *shape, const float *bound,
                                     int32_t max_points, int32_t max_voxels, int32_t reduction, float *voxels,
                                     int64_t *coords, uint8_t *pmask, int32_t *npoints, float *aggregates,
                                     int64_t *counts, void *workspace, size_t workspace_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || c < 3 || !shape || !bound || !counts || max_points < 0 || max_voxels < 0) return D3D_ERR_BAD_ARG;
    if (n > 0 && !points) return D3D_ERR_BAD_ARG;
    if (n >= (1ll << 31) - kScanTile) return D3D_ERR_BAD_ARG;
    if (reduction < D3D_REDUCE_NONE || reduction > D3D_REDUCE_MIN) return D3D_ERR_UNSUPPORTED;  // voxelize.cpp:196
    if (reduction != D3D_REDUCE_NONE && !aggregates && n > 0 && max_voxels > 0) return D3D_ERR_BAD_ARG;
    if (shape[0] <= 0 || shape[1] <= 0 || shape[2] <= 0) return D3D_ERR_BAD_ARG;
    VoxelWs w = carve(workspace, workspace_bytes, n, 0);
    if (!workspace || w.bytes > workspace_bytes) return D3D_ERR_WORKSPACE;
    if ((max_voxels > 0 && n > 0) && (!voxels || !coords || !pmask || !npoints)) return D3D_ERR_BAD_ARG;

    DenseKey kf;
    for (int d = 0; d < 3; d++) {
        kf.lo[d] = bound[d << 1];
        kf.shape[d] = shape[d];
        // voxelize.cpp:84-86: float(hi - lo) / int, evaluated in fp32 on the host
        volatile float diff = bound[(d << 1) | 1] - bound[d << 1];
        volatile float sz = diff / (float)shape[d];
        kf.size[d] = sz;
    }
    // packed slots (claim + count in one 64-bit atomic) whenever key and count fit 40 + 24 bits
    const double cells = (double)shape[0] * (double)shape[1] * (double)shape[2];
    if (cells >= 9.0e18) return D3D_ERR_BAD_ARG;
    // D3D_FORCE_PLAIN_SLOTS=1 selects the general layout (test hook for the n >= 2^24 path)
    const char *force_plain = getenv("D3D_FORCE_PLAIN_SLOTS");
    const bool packed = cells < 1.0e12 && n < (1ll << kCntBits) && !(force_plain && force_plain[0] == '1');
    int rc = packed ? dense_index<true>(kf, points, n, c, w, counts, coords, npoints, (uint32_t)max_points,
                                        (uint32_t)max_voxels, st)
                    : dense_index<false>(kf, points, n, c, w, counts, coords, npoints, (uint32_t)max_points,
                                         (uint32_t)max_voxels, st);
    if (rc) return rc;
    if (n == 0 || max_voxels == 0) return D3D_OK;

    if (max_points > 0) {
        const int64_t cap = n < max_voxels ? n : (int64_t)max_voxels;
        const bool vec4 = (c == 4) && ((reinterpret_cast<uintptr_t>(points) & 15) == 0) &&
                          ((reinterpret_cast<uintptr_t>(voxels) & 15) == 0);
        if (vec4)
            D3D_LAUNCH("k_fill_c4", k_fill_c4, dim3(grid_for(cap * max_points, 256, 256 * 32)), dim3(256), 0, st,
                               reinterpret_cast<const float4 *>(points), counts, npoints, w.voff, w.list,
                               (uint32_t)max_points, reinterpret_cast<float4 *>(voxels));
        else
            D3D_LAUNCH("k_fill_generic", k_fill_generic, dim3(grid_for(cap * max_points * c, 256, 256 * 32)), dim3(256), 0, st,
                               points, c, counts, npoints, w.voff, w.list, (uint32_t)max_points, voxels);
        D3D_LAUNCH("k_pmask", k_pmask, dim3(grid_for(d3d_divup(cap * max_points, 16), 256)), dim3(256), 0, st, counts,
                           npoints, (uint32_t)max_points, pmask);
    }
    if (reduction != D3D_REDUCE_NONE) {
        const int64_t cap = n < max_voxels ? n : (int64_t)max_voxels;
        if (max_points == 0) return D3D_ERR_UNSUPPORTED;   // aggregates need the lists (documented)
        D3D_LAUNCH("k_aggregate", k_aggregate, dim3(grid_for(cap * c, 256)), dim3(256), 0, st, points, c, counts, npoints,
                           w.voff, w.list, w.unsorted, (uint32_t)max_points, reduction, aggregates);
    }
    return D3D_OK;
}

// The "voxel feature grid" without the dense [V,P,C] copy: first-seen voxel ids, counts, per-voxel
// reduction of all in-range points, point -> voxel map and each voxel's first point index.  Same grid
// semantics as d3d_voxelize_3d_dense (voxelize.cpp:100-101).  Used stand-alone ("dynamic voxelization")
// and as the per-rank stage of the point-sharded voxelizer (d3d_amd/voxel/sharded.py).
extern "C" int d3d_voxelize_3d_reduce(const float *points, int64_t n, int32_t c, const int32_t *shape, const float *bound,
                                      int32_t reduction, int64_t index_offset, int64_t *coords, int32_t *npoints,
                                      float *aggregates, int64_t *first, int64_t *mapping, int64_t *counts,
                                      void *workspace, size_t workspace_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || c < 3 || !shape || !bound || !counts) return D3D_ERR_BAD_ARG;
    if (n >= (1ll << 31) - kScanTile) return D3D_ERR_BAD_ARG;
    if (reduction < D3D_REDUCE_MEAN || reduction > kReduceSum) return D3D_ERR_UNSUPPORTED;
    if (shape[0] <= 0 || shape[1] <= 0 || shape[2] <= 0) return D3D_ERR_BAD_ARG;
    if (n > 0 && (!points || !coords || !npoints || !aggregates)) return D3D_ERR_BAD_ARG;
    VoxelWs w = carve(workspace, workspace_bytes, n, 0);
    if (!workspace || w.bytes > workspace_bytes) return D3D_ERR_WORKSPACE;
    DenseKey kf;
    for (int d = 0; d < 3; d++) {
        kf.lo[d] = bound[d << 1];
        kf.shape[d] = shape[d];
        volatile float diff = bound[(d << 1) | 1] - bound[d << 1];
        volatile float sz = diff / (float)shape[d];
        kf.size[d] = sz;
    }
    const double cells = (double)shape[0] * (double)shape[1] * (double)shape[2];
    if (cells >= 9.0e18) return D3D_ERR_BAD_ARG;
    const char *force_plain = getenv("D3D_FORCE_PLAIN_SLOTS");
    const bool packed = cells < 1.0e12 && n < (1ll << kCntBits) && !(force_plain && force_plain[0] == '1');
    const uint32_t P = 32;   // voxels up to 32 points are reduced sequentially in point order, larger ones cooperatively
    int rc = packed ? dense_index<true>(kf, points, n, c, w, counts, coords, npoints, P, 0xffffffffu, st, first,
                                        index_offset, mapping)
                    : dense_index<false>(kf, points, n, c, w, counts, coords, npoints, P, 0xffffffffu, st, first,
                                         index_offset, mapping);
    if (rc) return rc;
    if (n == 0) return D3D_OK;
    D3D_LAUNCH("k_aggregate", k_aggregate, dim3(grid_for(n * c, 256)), dim3(256), 0, st, points, c, counts, npoints,
               w.voff, w.list, w.unsorted, P, reduction, aggregates);
    return D3D_OK;
}

extern "C" int d3d_voxelize_3d_sparse(const float *points, int64_t n, int32_t c, const float *voxel_size,
                                      int64_t *points_mapping, int64_t *coords, int32_t *npoints, int64_t *counts,
                                      void *workspace, size_t workspace_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || c < 3 || !voxel_size || !counts) return D3D_ERR_BAD_ARG;
    if (n > 0 && (!points || !points_mapping || !coords || !npoints)) return D3D_ERR_BAD_ARG;
    if (n >= (1ll << 31) - kScanTile) return D3D_ERR_BAD_ARG;
    VoxelWs w = carve(workspace, workspace_bytes, n, 0);
    if (!workspace || w.bytes > workspace_bytes) return D3D_ERR_WORKSPACE;
    SparseKey kf;
    for (int d = 0; d < 3; d++) kf.size[d] = voxel_size[d];
    int rc = build_table<SparseKey, false>(kf, points, n, c, w, counts, 0u, st);
    if (rc) return rc;
    NumberVoxels<SparseKey, false> nv{kf, w.table, w.pslot, w.voff, coords, npoints, 0u, 0xffffffffu, nullptr, 0};
    rc = d3d_run_scan(nv, n, w.bsum, counts, D3D_COUNT_VOXELS, -1, ~0ull, st);
    if (rc) return rc;
    if (n > 0) {
        D3D_LAUNCH("k_map", k_map<false>, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, w.table, w.pslot, n,
                           points_mapping);
    }
    return D3D_OK;
}

// order (descending stable argsort of voxel_npoints) for MAXVOX_DESCENDING, implemented in sort.hip
extern "C" int d3d_internal_argsort_desc_i32(const int32_t *keys, int64_t n, int32_t *order, void *ws, size_t ws_bytes,
                                             hipStream_t st);
extern "C" size_t d3d_internal_argsort_i32_bytes(int64_t n);

extern "C" int d3d_voxelize_3d_filter(const float *feats, int64_t n, int32_t c, const int64_t *points_mapping,
                                      const int64_t *coords, const int32_t *voxel_npoints, int64_t nvox,
                                      const int64_t *coords_bound, int32_t min_points, int32_t max_points,
                                      int32_t max_voxels, int32_t max_points_filter, int32_t max_voxels_filter,
                                      float *out_feats, int64_t *out_mask, int64_t *out_mapping, int32_t *out_npoints,
                                      int64_t *out_coords, int64_t *counts, void *workspace, size_t workspace_bytes,
                                      void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || nvox < 0 || c < 1 || !coords_bound || !counts) return D3D_ERR_BAD_ARG;
    if (max_points_filter == D3D_MAXPTS_FARTHEST_SAMPLING) return D3D_ERR_UNSUPPORTED;   // voxelize.cpp:469-471
    if (max_points_filter < 0 || max_points_filter > 2 || max_voxels_filter < 0 || max_voxels_filter > 2)
        return D3D_ERR_BAD_ARG;
    if (max_points < 0 || max_voxels < 0) return D3D_ERR_BAD_ARG;
    if (n >= (1ll << 31) - kScanTile || nvox >= (1ll << 31) - kScanTile) return D3D_ERR_BAD_ARG;
    if (n > 0 && (!feats || !points_mapping || !out_feats || !out_mask || !out_mapping)) return D3D_ERR_BAD_ARG;
    if (nvox > 0 && (!coords || !voxel_npoints || !out_npoints || !out_coords)) return D3D_ERR_BAD_ARG;
    VoxelWs w = carve(workspace, workspace_bytes, n, nvox);
    if (!workspace || w.bytes > workspace_bytes) return D3D_ERR_WORKSPACE;

    const bool trim_pts = max_points_filter == D3D_MAXPTS_TRIM;
    const uint32_t P = trim_pts ? (uint32_t)max_points : 0xffffffffu;
    // chain cells live in w.list: at most one group of P cells per overflow voxel, and an
    // overflow voxel owns > P points, so sum <= n.
    D3D_LAUNCH("k_fill_u32", k_fill_u32, dim3(grid_for(trim_pts ? n : 0, 256)), dim3(256), 0, st, w.list,
                       trim_pts ? n : (int64_t)0, kInf, counts);

    const int32_t *order = nullptr;
    if (max_voxels_filter == D3D_MAXVOX_DESCENDING && nvox > 0) {
        // stable descending argsort of the counts (reference: unstable torch::argsort, voxelize.cpp:406);
        // scratch: the hash-table region of the workspace (unused by the filter)
        int32_t *ord = reinterpret_cast<int32_t *>(w.pslot);   // npad >= ... only n entries guaranteed
        // pslot holds ceil(n/1024)*1024 entries; nvox may exceed n for hand-made inputs -> use table region
        size_t need = (size_t)nvox * sizeof(int32_t);
        char *tb = reinterpret_cast<char *>(w.table);
        size_t tbytes = (size_t)w.cap * sizeof(Slot);
        size_t sort_bytes = d3d_internal_argsort_i32_bytes(nvox);
        if (d3d_align_up(need) + sort_bytes > tbytes) return D3D_ERR_WORKSPACE;
        ord = reinterpret_cast<int32_t *>(tb);
        int rc = d3d_internal_argsort_desc_i32(voxel_npoints, nvox, ord, tb + d3d_align_up(need), sort_bytes, st);
        if (rc) return rc;
        order = ord;
    }

    FilterVoxels fv;
    fv.coords = coords; fv.npoints = voxel_npoints; fv.order = order;
    for (int d = 0; d < 3; d++) { fv.lo[d] = coords_bound[2 * d]; fv.hi[d] = coords_bound[2 * d + 1]; }
    fv.min_points = min_points;
    fv.max_points = P;
    fv.max_voxels = max_voxels_filter == D3D_MAXVOX_NONE ? ~0ull : (unsigned long long)max_voxels;
    fv.newid = w.newid; fv.coff = w.coff; fv.out_coords = out_coords; fv.out_npoints = out_npoints;
    int rc = d3d_run_scan(fv, nvox, w.bsum, counts, D3D_COUNT_VOXELS, -1, fv.max_voxels, st);
    if (rc) return rc;

    if (trim_pts && n > 0 && nvox > 0 && max_points > 0) {
        D3D_LAUNCH("k_filter_rank", k_filter_rank, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, points_mapping, n, nvox,
                           voxel_npoints, w.newid, w.coff, P, w.list, n);
    }
    FilterPoints fp{feats, c, points_mapping, nvox, voxel_npoints, w.newid, w.coff, w.list, n, P,
                    out_feats, out_mask, out_mapping};
    rc = d3d_run_scan(fp, n, w.bsum, counts, -1, D3D_COUNT_POINTS, ~0ull, st);
    return rc;
}
