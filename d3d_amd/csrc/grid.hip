// grid.hip -- compact, rank-independent numbering of occupied grid cells (used by the point-sharded
// voxelizer to give every rank the same slot for the same voxel without exchanging a dictionary):
// all ranks mark the cells named by the all-gathered key lists in a bitmap over the grid, a popcount
// prefix scan of the bitmap words turns "cell is occupied" into "index among the occupied cells in
// linear-key order".  RCCL has no bitwise-OR reduction, hence the keys are all-gathered and the
// bitmap is built locally.
#include "common.hpp"
#include <cstdint>
#include <algorithm>

namespace {

__global__ __launch_bounds__(256) void k_grid_mark(const int64_t *__restrict__ keys, int64_t m, int64_t ncells,
                                                   unsigned long long *bitmap)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int64_t k = keys[j];
    if (k < 0 || k >= ncells) return;
    // (testing the word first -- the gathered lists name a cell 1.4 times on average at world 8 -- costs a coherent load
    // per key and was slower: 733 vs 520 us for 8 M keys on config 5's 136 MB bitmap)
    atomicOr(&bitmap[k >> 6], 1ull << (k & 63));
}

// ---- bitmap exchange (grids whose bitmap is smaller than the key lists): every rank marks its own cells, the bitmaps
// are all-gathered and OR-ed here -- a streaming pass instead of one atomic per gathered key
constexpr int kMaxOwnerWorld = 16;   // ranks for which the ownership bookkeeping below is kept in registers
constexpr int kOwnerStride = 16;     // u64 words between the per-rank counters (own cache line each)

// OR of the ranks' bitmaps.  A cell is OWNED by the lowest rank that has it -- that rank holds the voxel's first point,
// because shards are contiguous point ranges in rank order.  With `lower` (this rank's view: cells some lower rank has)
// and `newc` (cells owned per rank) the global first-seen numbering needs no exchange of first indices (see
// OwnedRows below).  Grid-stride; one atomic per workgroup and rank, on separate cache lines.
__global__ __launch_bounds__(256) void k_grid_or(const unsigned long long *__restrict__ parts, int64_t stride_words, int world,
                                                 int64_t nw, unsigned long long *__restrict__ bitmap, int rank,
                                                 unsigned long long *__restrict__ lower, unsigned long long *newc)
{
    __shared__ unsigned int part[256 / kWave][kMaxOwnerWorld];
    unsigned int mine[kMaxOwnerWorld];
#pragma unroll
    for (int q = 0; q < kMaxOwnerWorld; q++) mine[q] = 0;
    const bool own = newc != nullptr;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long acc = 0;
        if (own) {
#pragma unroll
            for (int q = 0; q < kMaxOwnerWorld; q++) {
                if (q < world) {
                    const unsigned long long x = parts[(int64_t)q * stride_words + w];
                    if (q == rank) lower[w] = acc;
                    mine[q] += (unsigned int)__popcll(x & ~acc);
                    acc |= x;
                }
            }
        } else {
            for (int q = 0; q < world; q++) acc |= parts[(int64_t)q * stride_words + w];
        }
        bitmap[w] = acc;
    }
    if (!own) return;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < kMaxOwnerWorld; q++) {
        unsigned int v = mine[q];
        for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
        if (lane == 0) part[wv][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)world && threadIdx.x < (unsigned)kMaxOwnerWorld) {
        unsigned int t = 0;
        for (int k = 0; k < 256 / kWave; k++) t += part[k][threadIdx.x];
        if (t) atomicAdd(&newc[threadIdx.x * kOwnerStride], (unsigned long long)t);
    }
}

// key_of_slot by enumerating the set bits: slot = prefix[word] + index of the bit within the word
__global__ __launch_bounds__(256) void k_grid_keys(const unsigned long long *__restrict__ bitmap, const uint32_t *__restrict__ prefix,
                                                   int64_t nw, int64_t *__restrict__ key_of_slot)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    unsigned long long x = bitmap[w];
    int64_t s = prefix[w];
    while (x) {
        key_of_slot[s++] = w * 64 + __builtin_ctzll(x);
        x &= x - 1;
    }
}

struct PopcountWords {
    static constexpr const char *kName = "k_scan_count<PopcountWords>", *kName2 = "k_scan_apply<PopcountWords>";
    const unsigned long long *bitmap;
    uint32_t *prefix;
    __device__ __forceinline__ unsigned long long value(int64_t w) const { return (unsigned long long)__popcll(bitmap[w]); }
    __device__ __forceinline__ unsigned long long value2(int64_t w) const { return value(w); }
    __device__ __forceinline__ void apply(int64_t w, unsigned long long, unsigned long long excl) const
    {
        prefix[w] = (uint32_t)excl;
    }
};

__global__ __launch_bounds__(256) void k_grid_lookup(const int64_t *__restrict__ keys, int64_t m, int64_t ncells,
                                                     const unsigned long long *__restrict__ bitmap,
                                                     const uint32_t *__restrict__ prefix, int64_t *slot,
                                                     long long missing)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int64_t k = keys[j];
    long long s = missing;
    if (k >= 0 && k < ncells) {
        const unsigned long long word = bitmap[k >> 6], bit = 1ull << (k & 63);
        if (word & bit) s = (long long)prefix[k >> 6] + __popcll(word & (bit - 1));
    }
    slot[j] = s;
}


__device__ __forceinline__ long long compact_rank(long long k, int64_t ncells, const unsigned long long *__restrict__ bitmap,
                                                  const uint32_t *__restrict__ prefix)
{
    if (k < 0 || k >= ncells) return -1;
    const unsigned long long word = bitmap[k >> 6], bit = 1ull << (k & 63);
    return (word & bit) ? (long long)prefix[k >> 6] + __popcll(word & (bit - 1)) : -1;
}

// sharded voxelizer, step 4a: identity rows for the all-reduce (sum: 0, max: -inf, min: +inf; first index: max)
__global__ __launch_bounds__(256) void k_sharded_fill(int64_t nvox, int tstride, float identity, float *table,
                                                      int32_t *cnt_table, int64_t *first)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nvox * tstride) table[i] = identity;
    if (i < nvox) {
        if (cnt_table) cnt_table[i] = 0;
        first[i] = INT64_MAX;
    }
}

// step 4b, one lane per gathered key: slot of the key; every rank writes key_of_slot (duplicates write the same value),
// the lanes of this rank's own segment also place its partial rows into the table
__global__ __launch_bounds__(256) void k_sharded_scatter(const int64_t *__restrict__ keys_all, int64_t m, int64_t begin,
                                                         int64_t n_local, int64_t ncells,
                                                         const unsigned long long *__restrict__ bitmap,
                                                         const uint32_t *__restrict__ prefix, int c, int with_count,
                                                         const float *__restrict__ agg, const int32_t *__restrict__ cnt,
                                                         const int64_t *__restrict__ first_local, float *table, int tstride,
                                                         int32_t *cnt_table, int64_t *first, int64_t *key_of_slot,
                                                         int64_t *slot_of_local)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int64_t k = keys_all[j];
    const long long s = compact_rank(k, ncells, bitmap, prefix);
    const int64_t v = j - begin;                       // row of this rank's local result
    const bool mine = v >= 0 && v < n_local;
    if (mine) slot_of_local[v] = s;
    if (s < 0) return;
    if (key_of_slot) key_of_slot[s] = k;
    if (!mine) return;
    for (int d = 0; d < c; d++) table[s * tstride + d] = agg[v * c + d];
    if (with_count) table[s * tstride + c] = (float)cnt[v];      // counts < 2^24 are exact in fp32
    else cnt_table[s] = cnt[v];
    first[s] = first_local[v];
}

// sharded voxelizer, last step: slot-ordered reduced table -> voxel-id-ordered outputs.  The voxel id of a slot is
// the rank of its first point index among all first indices (bitmap over the frame's points, popcount prefix).
template <bool PACKED>
__global__ __launch_bounds__(256) void k_sharded_finalize(int64_t nvox, int c, const int64_t *__restrict__ first,
                                                          int64_t n_total, const unsigned long long *__restrict__ bitmap,
                                                          const uint32_t *__restrict__ prefix,
                                                          const int64_t *__restrict__ key_of_slot,
                                                          const float *__restrict__ table, int tstride, int mean,
                                                          const int32_t *__restrict__ cnt_in, int64_t sy, int64_t sz,
                                                          int64_t *vid_of_slot, int64_t *coords, int32_t *cnt_out, float *feats,
                                                          bool vec4)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nvox) return;
    const long long v = compact_rank(first[s], n_total, bitmap, prefix);
    vid_of_slot[s] = v;
    if (v < 0) return;                                 // cannot happen: every slot has a first point
    const int64_t k = key_of_slot[s];
    long long *cp = reinterpret_cast<long long *>(coords) + v * 3;
    if (PACKED) {
        // two scattered 16-byte stores per voxel instead of four requests: {key, count} parked in the 16-byte-aligned half
        // of the voxel's coordinate row (k_sharded_unpack expands it in place, streaming) + the features
        const float *row = table + s * tstride;
        const float n = mean ? row[c] : 1.f;
        const int32_t cnt = mean ? (int32_t)(n + 0.5f) : cnt_in[s];
        *reinterpret_cast<longlong2 *>(cp + (v & 1)) = make_longlong2(k, (long long)cnt);
        float4 f = make_float4(row[0], row[1], row[2], row[3]);
        if (mean) f = make_float4(f.x / n, f.y / n, f.z / n, f.w / n);
        reinterpret_cast<float4 *>(feats)[v] = f;
        return;
    }
    const long long c0 = k / (sy * sz), c1 = (k / sz) % sy, c2 = k % sz;
    if (vec4) {                                          // 24 bytes as 16 + 8 (rows alternate their 16-byte alignment)
        if ((v & 1) == 0) { *reinterpret_cast<longlong2 *>(cp) = make_longlong2(c0, c1); cp[2] = c2; }
        else { cp[0] = c0; *reinterpret_cast<longlong2 *>(cp + 1) = make_longlong2(c1, c2); }
    } else { cp[0] = c0; cp[1] = c1; cp[2] = c2; }
    const float *row = table + s * tstride;
    const float n = mean ? row[c] : 1.f;              // counts < 2^24 are exact in fp32
    cnt_out[v] = mean ? (int32_t)(n + 0.5f) : cnt_in[s];
    if (c == 4 && vec4) {                             // one 16-byte scattered store instead of four 4-byte ones
        float4 f = make_float4(row[0], row[1], row[2], row[3]);
        if (mean) f = make_float4(f.x / n, f.y / n, f.z / n, f.w / n);
        reinterpret_cast<float4 *>(feats)[v] = f;
    } else {
        for (int d = 0; d < c; d++) feats[v * c + d] = mean ? row[d] / n : row[d];
    }
}

// ---- numbering by ownership (bitmap exchange): the voxels a rank owns, in the rank's local first-seen order, are a
// contiguous run of the global first-seen order; runs follow each other in rank order.  So
//   voxel id = (cells owned by lower ranks) + (owned voxels before it in local order)
// -- a scan over the rank's own voxels.  The owner writes the id into an extra column of the table, whose all-reduce
// (sum: the others add 0; max / min: the others hold the identity) hands it to everybody: no exchange of first
// indices, no bitmap over the frame's points.  Ids are exact in fp32 below 2^24 voxels (host-checked).
struct OwnedRows {
    static constexpr const char *kName = "k_scan_count<OwnedRows>", *kName2 = "k_scan_apply<OwnedRows>";
    const int64_t *keys;              // [n_local] linear cell per local voxel, -1 beyond the rank's voxels
    int64_t ncells;
    const unsigned long long *bitmap; // merged occupancy + prefix: slot of a cell
    const uint32_t *prefix;
    const unsigned long long *lower;  // cells some lower rank has
    const unsigned long long *newc;   // cells owned per rank (stride kOwnerStride)
    int rank, c, with_count, tstride;
    const float *agg;
    const int32_t *cnt;
    float *table;
    int32_t *cnt_table;
    int64_t *slot_of_local;
    float identity;                   // what k_sharded_fill_owned wrote
    bool row_pairs;                   // table and agg aligned for float2 / float4 access
    bool row_quads;                   // ... and the table for float4 access
    __device__ __forceinline__ unsigned long long value(int64_t v) const
    {
        const int64_t k = keys[v];
        if (k < 0 || k >= ncells) return 0;
        return (lower[k >> 6] >> (k & 63)) & 1ull ? 0ull : 1ull;
    }
    __device__ __forceinline__ unsigned long long value2(int64_t v) const { return value(v); }
    __device__ __forceinline__ void apply(int64_t v, unsigned long long owned, unsigned long long excl) const
    {
        const int64_t k = keys[v];
        const long long s = compact_rank(k, ncells, bitmap, prefix);
        slot_of_local[v] = s;
        if (s < 0) return;
        float *row = table + s * tstride;
        float id = 0.f;
        if (owned) {
            unsigned long long base = 0;
            for (int q = 0; q < rank; q++) base += newc[q * kOwnerStride];
            id = (float)(base + excl);
        }
        if (c == 4 && with_count && tstride == 6 && row_pairs) {
            // the common shape (4 features + count + id = 24-byte rows): three 8-byte stores instead of six 4-byte ones --
            // scattered requests are what this pass costs.  The id column of a row another rank owns keeps the identity.
            const float4 a = reinterpret_cast<const float4 *>(agg)[v];
            const float cf = (float)cnt[v], idf = owned ? id : identity;
            if (row_quads) {                          // 24 bytes as 16 + 8: rows alternate their 16-byte alignment
                if ((s & 1) == 0) {
                    *reinterpret_cast<float4 *>(row) = a;
                    *reinterpret_cast<float2 *>(row + 4) = make_float2(cf, idf);
                } else {
                    *reinterpret_cast<float2 *>(row) = make_float2(a.x, a.y);
                    *reinterpret_cast<float4 *>(row + 2) = make_float4(a.z, a.w, cf, idf);
                }
                return;
            }
            float2 *r2 = reinterpret_cast<float2 *>(row);
            r2[0] = make_float2(a.x, a.y);
            r2[1] = make_float2(a.z, a.w);
            r2[2] = make_float2(cf, idf);
            return;
        }
        for (int d = 0; d < c; d++) row[d] = agg[v * c + d];
        if (with_count) row[c] = (float)cnt[v];       // counts < 2^24 are exact in fp32
        else cnt_table[s] = cnt[v];
        if (owned) row[tstride - 1] = id;
    }
};

// table[nvox, tstride] <- identity (the id column too), cnt_table <- 0
__global__ __launch_bounds__(256) void k_sharded_fill_owned(int64_t nvox, int tstride, float identity, float *table,
                                                            int32_t *cnt_table)
{
    // 16 bytes per lane where the table allows it (the table is all-reduced next: every rank writes all of it)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, total = nvox * tstride;
    if ((reinterpret_cast<uintptr_t>(table) & 15) == 0) {
        if (i * 4 + 4 <= total) reinterpret_cast<float4 *>(table)[i] = make_float4(identity, identity, identity, identity);
        else
            for (int64_t j = i * 4; j < total; j++) table[j] = identity;
    } else {
        for (int64_t j = i * 4; j < total && j < i * 4 + 4; j++) table[j] = identity;
    }
    if (cnt_table)
        for (int64_t j = i * 4; j < nvox && j < i * 4 + 4; j++) cnt_table[j] = 0;
}

// PACKED (C == 4, 16-byte aligned outputs): two scattered 16-byte stores per voxel instead of four -- the features (final)
// and {cell key, count} parked in the 16-byte aligned half of the voxel's own 24-byte coordinate slot; k_sharded_unpack
// then turns the slots into coordinates and counts in place, streaming.  Scattered store requests are what this pass costs
// (4 per global voxel on every rank: 179 us at 2.55 M voxels).
template <bool PACKED>
__global__ __launch_bounds__(256) void k_sharded_finalize_owned(int64_t nvox, int c, const int64_t *__restrict__ key_of_slot,
                                                                const float *__restrict__ table, int tstride, int mean,
                                                                const int32_t *__restrict__ cnt_in, int64_t sy, int64_t sz,
                                                                int64_t *vid_of_slot, int64_t *coords, int32_t *cnt_out,
                                                                float *feats, bool vec4)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nvox) return;
    const float *row = table + s * tstride;
    const int64_t v = (int64_t)(row[tstride - 1] + 0.5f);
    vid_of_slot[s] = v;
    if (v < 0 || v >= nvox) return;                     // cannot happen: every slot has exactly one owner
    const int64_t k = key_of_slot[s];
    const float n = mean ? row[c] : 1.f;
    const int32_t cnt = mean ? (int32_t)(n + 0.5f) : cnt_in[s];
    long long *cp = reinterpret_cast<long long *>(coords) + v * 3;
    if (PACKED) {
        *reinterpret_cast<longlong2 *>(cp + (v & 1)) = make_longlong2(k, (long long)cnt);   // (3 v + (v & 1)) * 8 is a multiple of 16
        float4 f = make_float4(row[0], row[1], row[2], row[3]);
        if (mean) f = make_float4(f.x / n, f.y / n, f.z / n, f.w / n);
        reinterpret_cast<float4 *>(feats)[v] = f;
        return;
    }
    const long long c0 = k / (sy * sz), c1 = (k / sz) % sy, c2 = k % sz;
    if (vec4) {                                          // 24 bytes as 16 + 8 (rows alternate their 16-byte alignment)
        if ((v & 1) == 0) { *reinterpret_cast<longlong2 *>(cp) = make_longlong2(c0, c1); cp[2] = c2; }
        else { cp[0] = c0; *reinterpret_cast<longlong2 *>(cp + 1) = make_longlong2(c1, c2); }
    } else { cp[0] = c0; cp[1] = c1; cp[2] = c2; }
    cnt_out[v] = cnt;
    if (c == 4 && vec4) {
        float4 f = make_float4(row[0], row[1], row[2], row[3]);
        if (mean) f = make_float4(f.x / n, f.y / n, f.z / n, f.w / n);
        reinterpret_cast<float4 *>(feats)[v] = f;
    } else {
        for (int d = 0; d < c; d++) feats[v * c + d] = mean ? row[d] / n : row[d];
    }
}

// in place, one lane per voxel id: {key, count} parked by the packed finalize -> coords[v, 0..3), cnt_out[v]
__global__ __launch_bounds__(256) void k_sharded_unpack(int64_t nvox, int64_t sy, int64_t sz, int64_t *coords, int32_t *cnt_out)
{
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvox) return;
    long long *cp = reinterpret_cast<long long *>(coords) + v * 3;
    const longlong2 kc = *reinterpret_cast<const longlong2 *>(cp + (v & 1));
    const long long k = kc.x;
    const long long c0 = k / (sy * sz), c1 = (k / sz) % sy, c2 = k % sz;
    if ((v & 1) == 0) { *reinterpret_cast<longlong2 *>(cp) = make_longlong2(c0, c1); cp[2] = c2; }
    else { cp[0] = c0; *reinterpret_cast<longlong2 *>(cp + 1) = make_longlong2(c1, c2); }
    cnt_out[v] = (int32_t)kc.y;
}

// global voxel id of each local point: local voxel -> slot -> voxel id
__global__ __launch_bounds__(256) void k_sharded_map(int64_t n, const int64_t *__restrict__ local_map,
                                                     const int64_t *__restrict__ slot_of_local, int64_t nvox,
                                                     const int64_t *__restrict__ vid_of_slot, int64_t *gmap)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t m = local_map[i];
    long long g = -1;
    if (m >= 0) {
        const int64_t s = slot_of_local[m];
        if (s >= 0 && s < nvox) g = vid_of_slot[s];
    }
    gmap[i] = g;
}

}  // namespace

extern "C" size_t d3d_grid_compact_workspace_bytes(int64_t ncells)
{
    if (ncells < 1) ncells = 1;
    const size_t nw = (size_t)d3d_divup(ncells, 64);
    return d3d_align_up(nw * 8) + d3d_align_up(nw * 4) + d3d_align_up(((size_t)d3d_divup((int64_t)nw, kScanTile) + 1) * 8) + 256;
}

// keys[m] (linear cell indices in [0, ncells)) -> occupancy index in `workspace`; counts[0] = number of
// distinct occupied cells.  The workspace is then read by d3d_grid_compact_lookup.
extern "C" int d3d_grid_compact_index(const int64_t *keys, int64_t m, int64_t ncells, int64_t *counts, void *workspace,
                                      size_t workspace_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (m < 0 || ncells <= 0 || !counts || (m > 0 && !keys)) return D3D_ERR_BAD_ARG;
    if (ncells >= (1ll << 37)) return D3D_ERR_BAD_ARG;
    const int64_t nw = d3d_divup(ncells, 64);
    WsCarver w(workspace, workspace_bytes);
    unsigned long long *bitmap = w.take<unsigned long long>(nw);
    uint32_t *prefix = w.take<uint32_t>(nw);
    unsigned long long *bsum = w.take<unsigned long long>(d3d_divup(nw, kScanTile) + 1);
    if (!workspace || !w.ok()) return D3D_ERR_WORKSPACE;
    D3D_HIP_CHECK(hipMemsetAsync(bitmap, 0, (size_t)nw * 8, st));
    if (m > 0) D3D_LAUNCH("k_grid_mark", k_grid_mark, dim3((unsigned)d3d_divup(m, 256)), dim3(256), 0, st, keys, m, ncells, bitmap);
    PopcountWords f{bitmap, prefix};
    return d3d_run_scan(f, nw, bsum, counts, -1, 0, ~0ull, st);
}

// bitmap exchange, step 1: this rank's occupancy bitmap (ceil(ncells/64) words; negative keys are ignored)
extern "C" int d3d_grid_bitmap_mark(const int64_t *keys, int64_t m, int64_t ncells, unsigned long long *bitmap, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (m < 0 || ncells <= 0 || !bitmap || (m > 0 && !keys)) return D3D_ERR_BAD_ARG;
    const int64_t nw = d3d_divup(ncells, 64);
    D3D_HIP_CHECK(hipMemsetAsync(bitmap, 0, (size_t)nw * 8, st));
    if (m > 0) D3D_LAUNCH("k_grid_mark", k_grid_mark, dim3((unsigned)d3d_divup(m, 256)), dim3(256), 0, st, keys, m, ncells, bitmap);
    return D3D_OK;
}

// bitmap exchange, step 2: OR of the `world` all-gathered bitmaps (rank r at parts + r * stride_words) into the compact
// index of `workspace` (as d3d_grid_compact_index does from key lists); counts[0] = distinct occupied cells; optionally
// key_of_slot[counts[0]] (cells in ascending order).
extern "C" size_t d3d_grid_owner_workspace_bytes(int64_t ncells)
{
    if (ncells < 1) ncells = 1;
    return d3d_align_up((size_t)d3d_divup(ncells, 64) * 8) + d3d_align_up((size_t)kMaxOwnerWorld * kOwnerStride * 8) + 256;
}

extern "C" int d3d_grid_compact_from_bitmaps(const unsigned long long *parts, int64_t stride_words, int32_t world, int64_t ncells,
                                             int64_t *counts, void *workspace, size_t workspace_bytes, int32_t rank,
                                             void *owner_ws, size_t owner_ws_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (world < 1 || ncells <= 0 || !counts || !parts) return D3D_ERR_BAD_ARG;
    if (ncells >= (1ll << 37)) return D3D_ERR_BAD_ARG;
    const int64_t nw = d3d_divup(ncells, 64);
    if (stride_words < nw) return D3D_ERR_BAD_ARG;
    WsCarver w(workspace, workspace_bytes);
    unsigned long long *bitmap = w.take<unsigned long long>(nw);
    uint32_t *prefix = w.take<uint32_t>(nw);
    unsigned long long *bsum = w.take<unsigned long long>(d3d_divup(nw, kScanTile) + 1);
    if (!workspace || !w.ok()) return D3D_ERR_WORKSPACE;
    unsigned long long *lower = nullptr, *newc = nullptr;
    if (owner_ws) {                                  // ownership bookkeeping for d3d_sharded_scatter_owned
        if (world > kMaxOwnerWorld || rank < 0 || rank >= world) return D3D_ERR_UNSUPPORTED;
        WsCarver ow(owner_ws, owner_ws_bytes);
        lower = ow.take<unsigned long long>(nw);
        newc = ow.take<unsigned long long>(kMaxOwnerWorld * kOwnerStride);
        if (!ow.ok()) return D3D_ERR_WORKSPACE;
        D3D_HIP_CHECK(hipMemsetAsync(newc, 0, (size_t)kMaxOwnerWorld * kOwnerStride * 8, st));
    }
    const unsigned nblk = (unsigned)std::min<int64_t>(d3d_divup(nw, 256), 1024);
    D3D_LAUNCH("k_grid_or", k_grid_or, dim3(nblk), dim3(256), 0, st, parts, stride_words, (int)world, nw, bitmap, (int)rank, lower,
               newc);
    PopcountWords f{bitmap, prefix};
    return d3d_run_scan(f, nw, bsum, counts, -1, 0, ~0ull, st);
}

// key of every slot of a compact index (ascending cells): key_of_slot must hold counts[0] entries
extern "C" int d3d_grid_compact_keys(int64_t ncells, const void *workspace, size_t workspace_bytes, int64_t *key_of_slot,
                                     void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (ncells <= 0 || !key_of_slot) return D3D_ERR_BAD_ARG;
    const int64_t nw = d3d_divup(ncells, 64);
    WsCarver w((void *)workspace, workspace_bytes);
    unsigned long long *bitmap = w.take<unsigned long long>(nw);
    uint32_t *prefix = w.take<uint32_t>(nw);
    if (!workspace || !w.ok()) return D3D_ERR_WORKSPACE;
    D3D_LAUNCH("k_grid_keys", k_grid_keys, dim3((unsigned)d3d_divup(nw, 256)), dim3(256), 0, st, bitmap, prefix, nw, key_of_slot);
    return D3D_OK;
}

// slot[j] = index of keys[j] among the occupied cells (ascending linear key), or `missing` if the cell is not marked
extern "C" int d3d_grid_compact_lookup(const int64_t *keys, int64_t m, int64_t ncells, const void *workspace,
                                       size_t workspace_bytes, int64_t missing, int64_t *slot, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (m < 0 || ncells <= 0 || (m > 0 && (!keys || !slot))) return D3D_ERR_BAD_ARG;
    const int64_t nw = d3d_divup(ncells, 64);
    WsCarver w((void *)workspace, workspace_bytes);
    unsigned long long *bitmap = w.take<unsigned long long>(nw);
    uint32_t *prefix = w.take<uint32_t>(nw);
    if (!workspace || !w.ok()) return D3D_ERR_WORKSPACE;
    if (m > 0)
        D3D_LAUNCH("k_grid_lookup", k_grid_lookup, dim3((unsigned)d3d_divup(m, 256)), dim3(256), 0, st, keys, m, ncells,
                   bitmap, prefix, slot, (long long)missing);
    return D3D_OK;
}

extern "C" int d3d_sharded_scatter(const int64_t *keys_all, int64_t m, int64_t begin, int64_t n_local, int64_t ncells,
                                   const void *compact_ws, size_t compact_ws_bytes, int64_t nvox, int32_t c,
                                   int32_t reduction, const float *agg, const int32_t *cnt, const int64_t *first_local,
                                   float *table, int32_t table_stride, int32_t *cnt_table, int64_t *first,
                                   int64_t *key_of_slot, int64_t *slot_of_local, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (m < 0 || n_local < 0 || nvox < 0 || ncells <= 0 || c < 1 || begin < 0) return D3D_ERR_BAD_ARG;
    if (reduction != D3D_REDUCE_MEAN && reduction != D3D_REDUCE_MAX && reduction != D3D_REDUCE_MIN) return D3D_ERR_UNSUPPORTED;
    const bool mean = reduction == D3D_REDUCE_MEAN;
    if (table_stride < c + (mean ? 1 : 0)) return D3D_ERR_BAD_ARG;
    if (nvox > 0 && (!table || !first || (!mean && !cnt_table))) return D3D_ERR_BAD_ARG;
    if (m > 0 && !keys_all) return D3D_ERR_BAD_ARG;
    if (n_local > 0 && (!agg || !cnt || !first_local || !slot_of_local)) return D3D_ERR_BAD_ARG;
    const int64_t nw = d3d_divup(ncells, 64);
    WsCarver w((void *)compact_ws, compact_ws_bytes);
    unsigned long long *bitmap = w.take<unsigned long long>(nw);
    uint32_t *prefix = w.take<uint32_t>(nw);
    if (!compact_ws || !w.ok()) return D3D_ERR_WORKSPACE;
    const float identity = mean ? 0.f : (reduction == D3D_REDUCE_MAX ? -INFINITY : INFINITY);
    if (nvox > 0)
        D3D_LAUNCH("k_sharded_fill", k_sharded_fill, dim3((unsigned)d3d_divup(nvox * table_stride, 256)), dim3(256), 0, st, nvox,
                   table_stride, identity, table, mean ? nullptr : cnt_table, first);
    if (m > 0)
        D3D_LAUNCH("k_sharded_scatter", k_sharded_scatter, dim3((unsigned)d3d_divup(m, 256)), dim3(256), 0, st, keys_all, m,
                   begin, n_local, ncells, bitmap, prefix, c, mean ? 1 : 0, agg, cnt, first_local, table, table_stride,
                   cnt_table, first, key_of_slot, slot_of_local);
    return D3D_OK;
}

extern "C" int d3d_sharded_finalize(int64_t nvox, int32_t c, const int64_t *first, int64_t n_total, int64_t *counts,
                                    void *compact_ws, size_t compact_ws_bytes, const int64_t *key_of_slot,
                                    const float *table, int32_t table_stride, int32_t mean, const int32_t *cnt_in,
                                    const int32_t *shape, int64_t *vid_of_slot, int64_t *coords, int32_t *cnt_out,
                                    float *feats, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (nvox < 0 || c < 1 || !shape || n_total < 0) return D3D_ERR_BAD_ARG;
    if (nvox == 0) return D3D_OK;
    if (!first || !key_of_slot || !table || !vid_of_slot || !coords || !cnt_out || !feats || (!mean && !cnt_in) || !counts)
        return D3D_ERR_BAD_ARG;
    int rc = d3d_grid_compact_index(first, nvox, n_total > 0 ? n_total : 1, counts, compact_ws, compact_ws_bytes, stream);
    if (rc) return rc;
    const int64_t nw = d3d_divup(n_total > 0 ? n_total : 1, 64);
    WsCarver w(compact_ws, compact_ws_bytes);
    unsigned long long *bitmap = w.take<unsigned long long>(nw);
    uint32_t *prefix = w.take<uint32_t>(nw);
    const bool al16 = ((reinterpret_cast<uintptr_t>(feats) | reinterpret_cast<uintptr_t>(coords)) & 15) == 0;
    if (c == 4 && al16) {
        D3D_LAUNCH("k_sharded_finalize", k_sharded_finalize<true>, dim3((unsigned)d3d_divup(nvox, 256)), dim3(256), 0, st, nvox, c, first,
                   n_total > 0 ? n_total : (int64_t)1, bitmap, prefix, key_of_slot, table, table_stride, mean, cnt_in,
                   (int64_t)shape[1], (int64_t)shape[2], vid_of_slot, coords, cnt_out, feats, true);
        D3D_LAUNCH("k_sharded_unpack", k_sharded_unpack, dim3((unsigned)d3d_divup(nvox, 256)), dim3(256), 0, st, nvox,
                   (int64_t)shape[1], (int64_t)shape[2], coords, cnt_out);
        return D3D_OK;
    }
    D3D_LAUNCH("k_sharded_finalize", k_sharded_finalize<false>, dim3((unsigned)d3d_divup(nvox, 256)), dim3(256), 0, st, nvox, c, first,
               n_total > 0 ? n_total : (int64_t)1, bitmap, prefix, key_of_slot, table, table_stride, mean, cnt_in,
               (int64_t)shape[1], (int64_t)shape[2], vid_of_slot, coords, cnt_out, feats,
               (reinterpret_cast<uintptr_t>(feats) & 15) == 0);
    return D3D_OK;
}

extern "C" int d3d_sharded_scatter_owned(const int64_t *keys_local, int64_t n_local, int64_t ncells, const void *compact_ws,
                                         size_t compact_ws_bytes, const void *owner_ws, size_t owner_ws_bytes, int32_t rank,
                                         int64_t nvox, int32_t c, int32_t reduction, const float *agg, const int32_t *cnt,
                                         float *table, int32_t table_stride, int32_t *cnt_table, int64_t *slot_of_local,
                                         void *scan_ws, size_t scan_ws_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n_local < 0 || nvox < 0 || ncells <= 0 || c < 1 || rank < 0 || rank >= kMaxOwnerWorld) return D3D_ERR_BAD_ARG;
    if (reduction != D3D_REDUCE_MEAN && reduction != D3D_REDUCE_MAX && reduction != D3D_REDUCE_MIN) return D3D_ERR_UNSUPPORTED;
    if (nvox >= (1ll << 24)) return D3D_ERR_UNSUPPORTED;              // ids travel in an fp32 column
    const bool mean = reduction == D3D_REDUCE_MEAN;
    if (table_stride < c + (mean ? 2 : 1)) return D3D_ERR_BAD_ARG;
    if (nvox > 0 && (!table || (!mean && !cnt_table))) return D3D_ERR_BAD_ARG;
    if (n_local > 0 && (!keys_local || !agg || !cnt || !slot_of_local)) return D3D_ERR_BAD_ARG;
    const int64_t nw = d3d_divup(ncells, 64);
    WsCarver w((void *)compact_ws, compact_ws_bytes);
    unsigned long long *bitmap = w.take<unsigned long long>(nw);
    uint32_t *prefix = w.take<uint32_t>(nw);
    WsCarver ow((void *)owner_ws, owner_ws_bytes);
    unsigned long long *lower = ow.take<unsigned long long>(nw);
    unsigned long long *newc = ow.take<unsigned long long>(kMaxOwnerWorld * kOwnerStride);
    WsCarver sw(scan_ws, scan_ws_bytes);
    unsigned long long *bsum = sw.take<unsigned long long>(d3d_divup(n_local > 0 ? n_local : 1, kScanTile) + 1);
    int64_t *scratch_counts = sw.take<int64_t>(D3D_NUM_COUNTS);
    if (!compact_ws || !owner_ws || !scan_ws || !w.ok() || !ow.ok() || !sw.ok()) return D3D_ERR_WORKSPACE;
    const float identity = mean ? 0.f : (reduction == D3D_REDUCE_MAX ? -INFINITY : INFINITY);
    if (nvox > 0)
        D3D_LAUNCH("k_sharded_fill_owned", k_sharded_fill_owned, dim3((unsigned)d3d_divup(d3d_divup(nvox * table_stride, 4), 256)), dim3(256), 0,
                   st, nvox, table_stride, identity, table, mean ? nullptr : cnt_table);
    if (n_local == 0) return D3D_OK;
    OwnedRows f{keys_local, ncells, bitmap, prefix, lower, newc, (int)rank, (int)c, mean ? 1 : 0, (int)table_stride,
                agg, cnt, table, cnt_table, slot_of_local, identity,
                ((reinterpret_cast<uintptr_t>(table) & 7) == 0) && ((reinterpret_cast<uintptr_t>(agg) & 15) == 0),
                (reinterpret_cast<uintptr_t>(table) & 15) == 0};
    return d3d_run_scan(f, n_local, bsum, scratch_counts, -1, 0, ~0ull, st);
}

extern "C" int d3d_sharded_finalize_owned(int64_t nvox, int32_t c, const int64_t *key_of_slot, const float *table,
                                          int32_t table_stride, int32_t mean, const int32_t *cnt_in, const int32_t *shape,
                                          int64_t *vid_of_slot, int64_t *coords, int32_t *cnt_out, float *feats, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (nvox < 0 || c < 1 || !shape) return D3D_ERR_BAD_ARG;
    if (nvox == 0) return D3D_OK;
    if (!key_of_slot || !table || !vid_of_slot || !coords || !cnt_out || !feats || (!mean && !cnt_in)) return D3D_ERR_BAD_ARG;
    const bool al16 = ((reinterpret_cast<uintptr_t>(feats) | reinterpret_cast<uintptr_t>(coords)) & 15) == 0;
    if (c == 4 && al16) {
        D3D_LAUNCH("k_sharded_finalize_owned", k_sharded_finalize_owned<true>, dim3((unsigned)d3d_divup(nvox, 256)), dim3(256), 0, st,
                   nvox, c, key_of_slot, table, table_stride, mean, cnt_in, (int64_t)shape[1], (int64_t)shape[2], vid_of_slot,
                   coords, cnt_out, feats, true);
        D3D_LAUNCH("k_sharded_unpack", k_sharded_unpack, dim3((unsigned)d3d_divup(nvox, 256)), dim3(256), 0, st, nvox,
                   (int64_t)shape[1], (int64_t)shape[2], coords, cnt_out);
        return D3D_OK;
    }
    D3D_LAUNCH("k_sharded_finalize_owned", k_sharded_finalize_owned<false>, dim3((unsigned)d3d_divup(nvox, 256)), dim3(256), 0, st,
               nvox, c, key_of_slot, table, table_stride, mean, cnt_in, (int64_t)shape[1], (int64_t)shape[2], vid_of_slot, coords,
               cnt_out, feats, al16);
    return D3D_OK;
}

extern "C" int d3d_sharded_map(int64_t n, const int64_t *local_map, const int64_t *slot_of_local, int64_t nvox,
                               const int64_t *vid_of_slot, int64_t *gmap, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || nvox < 0) return D3D_ERR_BAD_ARG;
    if (n == 0) return D3D_OK;
    if (!local_map || !slot_of_local || !gmap || (nvox > 0 && !vid_of_slot)) return D3D_ERR_BAD_ARG;
    D3D_LAUNCH("k_sharded_map", k_sharded_map, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, n, local_map,
               slot_of_local, nvox, vid_of_slot, gmap);
    return D3D_OK;
}
