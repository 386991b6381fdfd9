// owner.hip -- point-sharded voxelizer, owner-computes exchange (the reference has no distributed code).
//
// Every cell of the grid has ONE owner rank, owner(cell) = hash(cell) * W >> 64.  A rank voxelizes its own shard
// (d3d_voxelize_3d_reduce), sends each local voxel's partial record {cell, first global point index, count, partial
// features} to the cell's owner (one all-to-all), and the owner merges the <= W records of each of its cells: the
// "all-reduce of the voxel feature grid" of north_star, done sparsely -- a reduce-scatter whose scatter is the hash.  All
// per-voxel work after the local pass (merge, numbering, division, output order) touches 1/W of the frame's voxels per
// rank; nothing is sized by the grid (no bitmap over cells) and nothing by the frame except one bit per POINT:
//   numbering   the owners' voxels are disjoint and every voxel has one first point, so the owners' "first point" bit
//               sets over the frame's point indices are disjoint -> their integer SUM (an all-reduce RCCL has) is their
//               OR; its popcount prefix is the reference's first-seen voxel id (voxelize.cpp:119).  1 bit per point.
//   merge       deterministic: a slot keeps one record index per source rank and the sums run in rank order -- the
//               result does not depend on the arrival order of the records or on the atomics that place them.
#include "common.hpp"
#include <cstdint>
#include <algorithm>

namespace {

typedef unsigned long long u64;
constexpr int kMaxWorld = 64;
constexpr u64 kFree = ~0ull;
constexpr int kPackTile = 4096;      // records per workgroup in the partition by owner
constexpr int kFusedEntries = 8192;  // tile x destination counts up to which k_owner_scatter derives its offsets itself

__device__ __forceinline__ u64 mix64(u64 h)
{
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return h;
}
__device__ __forceinline__ uint32_t owner_of(int64_t key, uint32_t world)
{
    return (uint32_t)(((mix64((u64)key) >> 32) * (u64)world) >> 32);
}

// record = RS 32-bit words: key (2), first global point index (2), count (1), c partial features, padding to even
__host__ __device__ inline int rec_stride(int c) { return (5 + c + 1) & ~1; }

// ---------------------------------------------------------------- partition of the local voxels by owner (3 launches)
// DENSE: the voxels' ranked rows (the first min(count, P) points of every local voxel, d3d_voxelize_3d_reduce's `rows`) travel
// too, in a second buffer with the same grouping; a record then carries, in its last word, the offset of its rows inside
// its (source, destination) batch.
struct PackDense {
    uint32_t P;
    const uint32_t *seg_base;     // [n] first row of every local voxel in rows_local
    const float4 *rows_local;     // the staged rows -- or, when counts[D3D_COUNT_AUX] == 1 (d3d_voxelize_3d_reduce on the binned
                                  // index), uint32 ranked point indices: row k >= 1 of voxel i = points4[ranked[seg_base[i] + k]],
                                  // row 0 = its first point points4[first[i] - index_offset]
    const float4 *points4;        // the shard's points (ranked mode)
    int64_t index_offset;
    float4 *send_rows;
    uint32_t *tilerows;           // [ntiles][world] rows per tile and destination -> exclusive prefix over the tiles
    uint32_t *dest_rowbase;       // [world + 1]
};

// per tile of kPackTile voxels: how many go to each rank (DENSE: and how many rows)
template <bool DENSE>
__global__ __launch_bounds__(1024) void k_owner_count(const int64_t *__restrict__ keys, const int32_t *__restrict__ cnt,
                                                      const int64_t *__restrict__ counts, uint32_t world,
                                                      uint32_t *__restrict__ tilecnt, PackDense pd)
{
    __shared__ uint32_t h[kMaxWorld], hr[kMaxWorld];
    const int64_t V = counts[D3D_COUNT_VOXELS];
    if (threadIdx.x < kMaxWorld) { h[threadIdx.x] = 0; hr[threadIdx.x] = 0; }
    __syncthreads();
    for (int k = 0; k < kPackTile / 1024; k++) {
        const int64_t i = (int64_t)blockIdx.x * kPackTile + k * 1024 + threadIdx.x;
        if (i < V) {
            const uint32_t d = owner_of(keys[i], world);
            atomicAdd(&h[d], 1u);
            if (DENSE) { const uint32_t c = (uint32_t)cnt[i]; atomicAdd(&hr[d], c < pd.P ? c : pd.P); }
        }
    }
    __syncthreads();
    if (threadIdx.x < world) {
        tilecnt[(size_t)blockIdx.x * world + threadIdx.x] = h[threadIdx.x];
        if (DENSE) pd.tilerows[(size_t)blockIdx.x * world + threadIdx.x] = hr[threadIdx.x];
    }
}

// one wavefront per destination (and matrix): exclusive prefix of its column over the tiles; then the destinations' bases.
// send_counts[0 .. world) = records per destination, [world] = the shard's status bits (keys[n] = -1 - status),
// [world + 1 .. 2 world + 1) = rows per destination (DENSE, else 0)
__global__ __launch_bounds__(1024) void k_owner_offsets(uint32_t *tilecnt, uint32_t *tilerows, uint32_t ntiles, uint32_t world,
                                                        int64_t *send_counts, uint32_t *dest_base, uint32_t *dest_rowbase,
                                                        const int64_t *status_key)
{
    __shared__ uint32_t tot[2][kMaxWorld];
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t jobs = tilerows ? 2 * world : world;
    for (uint32_t job = threadIdx.x >> 6; job < jobs; job += 1024 / kWave) {
        const uint32_t d = job % world, which = job / world;
        uint32_t *mat = which ? tilerows : tilecnt;
        uint32_t carry = 0;
        for (uint32_t t0 = 0; t0 < ntiles; t0 += kWave) {
            const uint32_t t = t0 + lane;
            const uint32_t x = t < ntiles ? mat[(size_t)t * world + d] : 0u;
            uint32_t incl = x;
#pragma unroll
            for (int s = 1; s < kWave; s <<= 1) {
                const uint32_t y = (uint32_t)__shfl_up((int)incl, s, kWave);
                if (lane >= s) incl += y;
            }
            if (t < ntiles) mat[(size_t)t * world + d] = carry + incl - x;
            carry += (uint32_t)__shfl((int)incl, kWave - 1, kWave);
        }
        if (lane == 0) tot[which][d] = carry;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0, rrun = 0;
        for (uint32_t d = 0; d < world; d++) {
            dest_base[d] = run;
            send_counts[d] = tot[0][d];
            run += tot[0][d];
            const uint32_t r = tilerows ? tot[1][d] : 0u;
            if (tilerows) dest_rowbase[d] = rrun;
            send_counts[world + 1 + d] = r;
            rrun += r;
        }
        dest_base[world] = run;
        if (tilerows) dest_rowbase[world] = rrun;
        send_counts[world] = status_key ? -1 - *status_key : 0;
    }
}

// records to their place in the send buffer (grouped by destination, tile order inside a destination, the order inside a
// tile from the wavefronts' ballots: the same on every run)
template <bool DENSE>
__global__ __launch_bounds__(1024) void k_owner_scatter(const int64_t *__restrict__ keys, const int32_t *__restrict__ cnt,
                                                        const float *__restrict__ agg, const int64_t *__restrict__ first,
                                                        const int64_t *__restrict__ counts, int c, uint32_t world,
                                                        const uint32_t *__restrict__ tileoff, const uint32_t *__restrict__ dest_base,
                                                        int32_t *__restrict__ send, int32_t *__restrict__ perm,
                                                        int32_t *__restrict__ pos_of_local, PackDense pd,
                                                        // fused offsets (small matrices): tileoff / pd.tilerows hold k_owner_count's RAW
                                                        // counts and every workgroup adds up its own column prefixes -- no k_owner_offsets
                                                        uint32_t fused_tiles, int64_t *send_counts, const int64_t *status_key)
{
    __shared__ uint32_t run[kMaxWorld], runr[kMaxWorld];   // records / rows of this tile already placed, per destination
    __shared__ uint32_t wcnt[1024 / kWave][kMaxWorld], wrow[DENSE ? 1024 / kWave : 1][kMaxWorld];
    __shared__ uint32_t f_before[kMaxWorld], f_total[kMaxWorld], f_rbefore[kMaxWorld], f_rtotal[kMaxWorld], f_base[kMaxWorld + 1],
        f_rbase[kMaxWorld + 1];
    const int64_t V = counts[D3D_COUNT_VOXELS];
    const bool ranked = DENSE && pd.points4 && counts[D3D_COUNT_AUX] == 1;
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int RS = rec_stride(c);
    if (threadIdx.x < kMaxWorld) {
        run[threadIdx.x] = 0; runr[threadIdx.x] = 0;
        f_before[threadIdx.x] = 0; f_total[threadIdx.x] = 0; f_rbefore[threadIdx.x] = 0; f_rtotal[threadIdx.x] = 0;
    }
    __syncthreads();
    if (fused_tiles) {
        // the whole matrix is a few KB (<= kFusedEntries words): wavefront-level column sums, then LDS atomics per destination
        const uint32_t entries = fused_tiles * world;
        for (uint32_t e = threadIdx.x; e < entries; e += 1024) {
            const uint32_t t = e / world, d = e - t * world;
            const uint32_t x = tileoff[e];
            if (x) { atomicAdd(&f_total[d], x); if (t < blockIdx.x) atomicAdd(&f_before[d], x); }
            if (DENSE) {
                const uint32_t y = pd.tilerows[e];
                if (y) { atomicAdd(&f_rtotal[d], y); if (t < blockIdx.x) atomicAdd(&f_rbefore[d], y); }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t acc = 0, racc = 0;
            for (uint32_t d = 0; d < world; d++) {
                f_base[d] = acc; f_rbase[d] = racc;
                acc += f_total[d]; racc += f_rtotal[d];
            }
            f_base[world] = acc; f_rbase[world] = racc;
            if (blockIdx.x == 0) {                 // what k_owner_offsets leaves for the host
                for (uint32_t d = 0; d < world; d++) {
                    send_counts[d] = f_total[d];
                    send_counts[world + 1 + d] = DENSE ? f_rtotal[d] : 0u;
                }
                send_counts[world] = status_key ? -1 - *status_key : 0;
            }
        }
        __syncthreads();
    }
    for (int k = 0; k < kPackTile / 1024; k++) {
        const int64_t i = (int64_t)blockIdx.x * kPackTile + k * 1024 + threadIdx.x;
        const bool ok = i < V;
        const int64_t key = ok ? keys[i] : 0;
        const uint32_t d = ok ? owner_of(key, world) : 0xffffffffu;
        const uint32_t nc = ok ? (uint32_t)cnt[i] : 0u;
        const uint32_t kept = DENSE ? (nc < pd.P ? nc : pd.P) : 0u;
        uint32_t myrank = 0, myrows = 0, rsrc = 0, rdst = 0, rfirst = 0;
        for (uint32_t q = 0; q < world; q++) {               // wave-uniform
            const u64 b = __ballot(d == q);
            if (d == q) myrank = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
            if (lane == 0) wcnt[w][q] = (uint32_t)__popcll(b);
            if (DENSE) {                                      // rows of the same-destination records before mine in the wave
                uint32_t incl = d == q ? kept : 0u;
#pragma unroll
                for (int sft = 1; sft < kWave; sft <<= 1) {
                    const uint32_t y = (uint32_t)__shfl_up((int)incl, sft, kWave);
                    if (lane >= sft) incl += y;
                }
                if (d == q) myrows = incl - kept;
                if (lane == kWave - 1) wrow[w][q] = incl;
            }
        }
        __syncthreads();
        if (ok) {
            uint32_t before = run[d], rbefore = DENSE ? runr[d] : 0u;
            for (int ww = 0; ww < w; ww++) {
                before += wcnt[ww][d];
                if (DENSE) rbefore += wrow[ww][d];
            }
            const size_t tidx = (size_t)blockIdx.x * world + d;
            const uint32_t pos = (fused_tiles ? f_base[d] + f_before[d] : dest_base[d] + tileoff[tidx]) + before + myrank;
            int32_t *r = send + (size_t)pos * RS;
            *reinterpret_cast<int64_t *>(r) = key;
            *reinterpret_cast<int64_t *>(r + 2) = first[i];
            r[4] = (int32_t)nc;
            for (int f = 0; f < c; f++) r[5 + f] = __float_as_int(agg[i * c + f]);
            perm[pos] = (int32_t)i;
            pos_of_local[i] = (int32_t)pos;
            if (DENSE) {
                const uint32_t inbatch = (fused_tiles ? f_rbefore[d] : pd.tilerows[tidx]) + rbefore + myrows;   // offset inside the (me -> d) row batch
                r[RS - 1] = (int32_t)inbatch;
                rsrc = pd.seg_base[i];
                rfirst = (uint32_t)(first[i] - pd.index_offset);
                rdst = (fused_tiles ? f_rbase[d] : pd.dest_rowbase[d]) + inbatch;
            }
        }
        if (DENSE) {
            // the rows of the wavefront's 64 voxels, copied by all lanes together: row j of the wavefront belongs to the voxel
            // whose running row count first exceeds j (a lane copying its own voxel's rows one after the other made the
            // whole wavefront wait for its fullest voxel: up to max_points trips of one 16-byte copy per lane)
            uint32_t incl = kept;
#pragma unroll
            for (int sft = 1; sft < kWave; sft <<= 1) {
                const uint32_t y = (uint32_t)__shfl_up((int)incl, sft, kWave);
                if (lane >= sft) incl += y;
            }
            const uint32_t wtotal = (uint32_t)__shfl((int)incl, kWave - 1, kWave);
            for (uint32_t j0 = 0; j0 < wtotal; j0 += kWave) {
                const uint32_t j = j0 + lane;
                // owner of row j: the number of lanes whose inclusive count is <= j (ballot per probe: binary search over lanes)
                int lo = 0, hi = kWave - 1;
#pragma unroll
                for (int step = 0; step < 6; step++) {
                    const int mid = (lo + hi) >> 1;
                    const uint32_t v = (uint32_t)__shfl((int)incl, mid, kWave);
                    if (v <= j) lo = mid + 1; else hi = mid;
                }
                const uint32_t own_incl = (uint32_t)__shfl((int)incl, lo, kWave), own_kept = (uint32_t)__shfl((int)kept, lo, kWave);
                const uint32_t s0 = (uint32_t)__shfl((int)rsrc, lo, kWave), d0 = (uint32_t)__shfl((int)rdst, lo, kWave);
                const uint32_t f0 = (uint32_t)__shfl((int)rfirst, lo, kWave);
                if (j < wtotal) {
                    const uint32_t t = j - (own_incl - own_kept);
                    if (ranked) {
                        const uint32_t idx = t == 0 ? f0 : reinterpret_cast<const uint32_t *>(pd.rows_local)[s0 + t];
                        pd.send_rows[d0 + t] = pd.points4[idx];
                    } else pd.send_rows[d0 + t] = pd.rows_local[s0 + t];
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < world) {
            uint32_t add = 0, addr = 0;
            for (int ww = 0; ww < 1024 / kWave; ww++) {
                add += wcnt[ww][threadIdx.x];
                if (DENSE) addr += wrow[ww][threadIdx.x];
            }
            run[threadIdx.x] += add;
            runr[threadIdx.x] += addr;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- merge on the owner (3 launches)
// The shards are contiguous point ranges in rank order, so a cell's first point lies in the LOWEST rank that has the cell,
// and inside one source's batch the records follow the shard's first-seen order: the "leader" records (lowest record index
// = lowest source of their cell), taken in receive order, are this owner's voxels in ascending order of first point =
// ascending global voxel id.
//   k_merge_init    table cleared: 16-byte slots {cell, head of the cell's record chain} at load <= 1/2
//   k_merge_insert  a record finds / claims its cell's slot (one CAS when the slot is free) and pushes itself on the slot's
//                   chain (one exchange on the same line).  The chain's ORDER is the arrival order -- nobody reads it as
//                   an order: every consumer visits a chain by ascending record index (chain_in_order), i.e. in rank
//                   order, so sums and row order are the same on every run.
//   k_merge_finish  one launch, tiles of records numbered by ticket: a record is a leader iff it is the lowest index of its
//                   chain; a decoupled look-back over the tiles' leader counts numbers the leaders; each leader merges its
//                   cell's records and writes the voxel's finished row at its number (consecutive leaders -> consecutive
//                   rows); the other records then pick up their leader's number (the leader sits in the same or an earlier
//                   tile: it is running or done, a short poll).
// Round 3 kept one record index per (slot, source rank) -- 40 bytes per slot to clear and to gather from -- and needed
// seven launches (init, insert, a three-launch scan over a leader flag, the map of the records): 191 us per rank at config 5
// against ... now (profiles/r04_sharded_*).
struct MergeSlot { u64 key; uint32_t head, pad; };          // head: record index + 1 of the last record pushed, 0 = none

struct MergeWs {
    MergeSlot *slot;      // [cap]
    uint32_t *next;       // [R] record index + 1 of the record pushed before this one on the same chain, 0 = end
    uint32_t *rec_slot;   // [R]
    u64 *status;          // look-back words of k_merge_finish, [ntiles] + one word holding the ticket
    int64_t *src_off;     // [world + 1] copy of the caller's (d3d_owner_dense finds a record's source rank with it)
    u64 cap;
    uint32_t ntiles;
    size_t bytes;
};

constexpr int kFinishThreads = 256, kFinishItems = 4, kFinishTile = kFinishThreads * kFinishItems;

static u64 merge_cap(int64_t R)        // load factor <= 1/2; slots are found by multiply-shift, so no power of two is needed
{
    return (u64)d3d_divup((R > 0 ? R : 1) * 2, 1024) * 1024 + 1024;
}

static MergeWs carve_merge(void *ws, size_t bytes, int64_t R, int world)
{
    WsCarver w(ws, bytes);
    MergeWs m;
    m.cap = merge_cap(R);
    m.ntiles = (uint32_t)d3d_divup(R > 0 ? R : 1, kFinishTile);
    m.slot = w.take<MergeSlot>(m.cap);
    m.next = w.take<uint32_t>(R > 0 ? R : 1);
    m.rec_slot = w.take<uint32_t>(R > 0 ? R : 1);
    m.status = w.take<u64>((size_t)m.ntiles + 1);
    m.src_off = w.take<int64_t>((size_t)world + 1);
    m.bytes = w.off;
    return m;
}

__global__ __launch_bounds__(256) void k_merge_init(MergeSlot *slot, u64 cap, u64 *status, uint32_t nstatus, int64_t *counts,
                                                    const int64_t *__restrict__ src_off, int world, int64_t *src_off_copy)
{
    typedef uint32_t uvec4 __attribute__((ext_vector_type(4)));
    const u64 stride = (u64)gridDim.x * blockDim.x, t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const uvec4 empty = {0xffffffffu, 0xffffffffu, 0u, 0u};
    uvec4 *s4 = reinterpret_cast<uvec4 *>(slot);
    for (u64 i = t; i < cap; i += stride) s4[i] = empty;
    for (u64 i = t; i < nstatus; i += stride) status[i] = 0ull;
    if (t < D3D_NUM_COUNTS) counts[t] = 0;
    if (t <= (u64)world) src_off_copy[t] = src_off[t];
}

__global__ __launch_bounds__(256) void k_merge_insert(const int32_t *__restrict__ recv, int64_t R, int RS, MergeSlot *slot, u64 cap,
                                                      uint32_t *__restrict__ next, uint32_t *__restrict__ rec_slot,
                                                      int32_t *__restrict__ rec_owned)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const u64 key = (u64)*reinterpret_cast<const int64_t *>(recv + (size_t)i * RS);
    // (a second mix: the owner hash already split on mix64's top bits)
    u64 h = __umul64hi(mix64(key * 0x9e3779b97f4a7c15ull), cap);
    for (;;) {
        const u64 old = atomicCAS(&slot[h].key, kFree, key);      // a free slot is the common case: one round trip
        if (old == kFree || old == key) break;
        h = h + 1 < cap ? h + 1 : 0;              // cap >= 2 R: a free slot always exists
    }
    next[i] = atomicExch(&slot[h].head, (uint32_t)i + 1u);
    rec_slot[i] = (uint32_t)h;
    rec_owned[i] = -1;                            // "my leader has no number yet" (k_merge_finish polls it)
}

__device__ __forceinline__ int record_source(const int64_t *__restrict__ src_off, int world, int64_t i)
{
    int lo = 0, hi = world;                       // the last s with src_off[s] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (src_off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// global first point of the voxel whose leader is record i.  Shards are contiguous point ranges in rank order and the leader
// is the cell's record from the LOWEST source rank, so the leader's first index is the voxel's (with global indices it equals
// the minimum over the cell's records, which is what rounds 3-4 took)
__device__ __forceinline__ int64_t leader_first(const struct MergeOut &f, int64_t i);

// the records of a chain by ascending record index (= source rank order); chains hold at most `world` records, nearly all one
// or two, so selection by repeated walks beats gathering them into a (dynamically indexed, i.e. scratch) list
template <class F>
__device__ __forceinline__ void chain_in_order(const uint32_t *__restrict__ next, uint32_t head, F &&visit)
{
    uint32_t prev = 0;
    for (;;) {
        uint32_t best = ~0u;
        for (uint32_t e = head; e; e = next[e - 1])
            if (e > prev && e < best) best = e;
        if (best == ~0u) return;
        visit(best - 1u);
        prev = best;
    }
}

struct MergeOut {
    int RS, c, reduction;                // reduction: MEAN (divide), 4 = sums, MAX, MIN
    int64_t sy, sz;
    int64_t *first_o, *coords;
    int32_t *npoints;
    float *feats;
    int32_t *lead_rec;                   // [owned voxel] its leader record
    // round 5: records may carry indices LOCAL to their source rank's shard (d3d_voxelize_3d_reduce with index_offset 0, so that
    // the local pass needs nobody's shard size); point_off[s] = global index of source s's first point then turns the leader's
    // index into the voxel's global first point.  NULL: the records carry global indices already.
    const int32_t *recv = nullptr;
    const int64_t *src_off = nullptr;    // [world + 1] records of source s = [src_off[s], src_off[s + 1])  (the workspace's copy)
    const int64_t *point_off = nullptr;  // [world]
    int world = 0;
};

__device__ __forceinline__ int64_t leader_first(const MergeOut &f, int64_t i)
{
    const int64_t local = *reinterpret_cast<const int64_t *>(f.recv + (size_t)i * f.RS + 2);
    return f.point_off ? local + f.point_off[record_source(f.src_off, f.world, i)] : local;
}

// leader record i -> owned voxel o: the cell's records merged in rank order, the voxel's finished row written
template <class Each>
__device__ __forceinline__ void merge_leader(const MergeOut &f, const int32_t *__restrict__ recv, int64_t i, int64_t o, bool sole,
                                             Each &&each_record /* each_record(visit): visit(record) in rank order */)
{
    const int c = f.c, RS = f.RS, reduction = f.reduction;
    const bool is_sum = reduction == D3D_REDUCE_MEAN || reduction == 4;
    const float ident = is_sum ? 0.0f : (reduction == D3D_REDUCE_MAX ? -INFINITY : INFINITY);
    f.lead_rec[o] = (int32_t)i;
    if (sole && c == 4) {                         // the cell's only record (most cells): everything is in record i
        const int32_t *r = recv + (size_t)i * RS;
        const int64_t key = *reinterpret_cast<const int64_t *>(r);
        const int32_t cnt = r[4];
        float a[4];
        for (int q = 0; q < 4; q++) {             // identity (op) x, exactly as the general loop below
            const float x = __int_as_float(r[5 + q]);
            a[q] = is_sum ? ident + x : (reduction == D3D_REDUCE_MAX ? (ident < x ? x : ident) : (x < ident ? x : ident));
        }
        f.first_o[o] = leader_first(f, i);
        f.npoints[o] = cnt;
        f.coords[o * 3 + 0] = key / (f.sy * f.sz);
        f.coords[o * 3 + 1] = (key / f.sz) % f.sy;
        f.coords[o * 3 + 2] = key % f.sz;
        const float d = reduction == D3D_REDUCE_MEAN ? (float)cnt : 1.0f;
        *reinterpret_cast<float4 *>(f.feats + o * 4) = make_float4(a[0] / d, a[1] / d, a[2] / d, a[3] / d);
        return;
    }
    float a0 = ident, a1 = ident, a2 = ident, a3 = ident;      // c == 4 in registers; other widths through feats[]
    if (c != 4)
        for (int q = 0; q < c; q++) f.feats[o * c + q] = ident;
    int32_t cnt = 0;
    int64_t key = 0;
    each_record([&](uint32_t ri) {                             // rank order: the same sums on every run
        const int32_t *r = recv + (size_t)ri * RS;
        key = *reinterpret_cast<const int64_t *>(r);
        cnt += r[4];
        if (c == 4) {
            const float x0 = __int_as_float(r[5]), x1 = __int_as_float(r[6]), x2 = __int_as_float(r[7]), x3 = __int_as_float(r[8]);
            if (is_sum) { a0 += x0; a1 += x1; a2 += x2; a3 += x3; }
            else if (reduction == D3D_REDUCE_MAX) { a0 = a0 < x0 ? x0 : a0; a1 = a1 < x1 ? x1 : a1; a2 = a2 < x2 ? x2 : a2; a3 = a3 < x3 ? x3 : a3; }
            else { a0 = x0 < a0 ? x0 : a0; a1 = x1 < a1 ? x1 : a1; a2 = x2 < a2 ? x2 : a2; a3 = x3 < a3 ? x3 : a3; }
        } else {
            for (int q = 0; q < c; q++) {
                const float x = __int_as_float(r[5 + q]), y = f.feats[o * c + q];
                f.feats[o * c + q] = is_sum ? y + x : (reduction == D3D_REDUCE_MAX ? (y < x ? x : y) : (x < y ? x : y));
            }
        }
    });
    f.first_o[o] = leader_first(f, i);
    f.npoints[o] = cnt;
    f.coords[o * 3 + 0] = key / (f.sy * f.sz);
    f.coords[o * 3 + 1] = (key / f.sz) % f.sy;
    f.coords[o * 3 + 2] = key % f.sz;
    const float d = reduction == D3D_REDUCE_MEAN ? (float)cnt : 1.0f;
    if (c == 4) *reinterpret_cast<float4 *>(f.feats + o * 4) = make_float4(a0 / d, a1 / d, a2 / d, a3 / d);
    else if (reduction == D3D_REDUCE_MEAN)
        for (int q = 0; q < c; q++) f.feats[o * c + q] = f.feats[o * c + q] / d;
}

__global__ __launch_bounds__(kFinishThreads) void k_merge_finish(const int32_t *__restrict__ recv, int64_t R,
                                                                 const MergeSlot *__restrict__ slot,
                                                                 const uint32_t *__restrict__ next,
                                                                 const uint32_t *__restrict__ rec_slot, u64 *status,
                                                                 uint32_t ntiles, MergeOut f, int32_t *rec_owned, int64_t *counts)
{
    __shared__ unsigned int sid;
    __shared__ u64 smem[kFinishThreads / kWave];
    __shared__ u64 sexcl;
    const unsigned int tile = lookback_ticket(reinterpret_cast<unsigned int *>(status + ntiles), &sid);
    // consecutive records per lane: the leaders' numbers then ascend with the record index across the tile
    const int64_t i0 = (int64_t)tile * kFinishTile + (int64_t)threadIdx.x * kFinishItems;
    uint32_t head[kFinishItems], lead[kFinishItems];          // lead: index + 1 of the chain's lowest record
    bool sole[kFinishItems];
    u64 mine = 0;
#pragma unroll
    for (int k = 0; k < kFinishItems; k++) {
        const int64_t i = i0 + k;
        head[k] = 0; lead[k] = 0; sole[k] = false;
        if (i < R) {
            head[k] = slot[rec_slot[i]].head;
            const uint32_t nx = next[i];
            sole[k] = head[k] == (uint32_t)i + 1u && nx == 0u;
            uint32_t m = (uint32_t)i + 1u;
            if (!sole[k])
                for (uint32_t e = head[k]; e; e = next[e - 1]) m = e < m ? e : m;
            lead[k] = m;
            if (m == (uint32_t)i + 1u) mine++;
        }
    }
    u64 total;
    u64 ex = block_excl_scan_u64<kFinishThreads>(mine, &total, smem);
    if (threadIdx.x < kWave) {
        const u64 e = lookback_exclusive(status, tile, total);
        if (threadIdx.x == 0) {
            sexcl = e;
            if (tile == ntiles - 1) counts[D3D_COUNT_VOXELS] = (int64_t)(e + total);
        }
    }
    __syncthreads();
    ex += sexcl;
#pragma unroll
    for (int k = 0; k < kFinishItems; k++) {
        const int64_t i = i0 + k;
        if (i < R && lead[k] == (uint32_t)i + 1u) {
            merge_leader(f, recv, i, (int64_t)ex, sole[k], [&](auto &&visit) { chain_in_order(next, head[k], visit); });
            __hip_atomic_store(&rec_owned[i], (int32_t)ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ex++;
        }
    }
    // the other records: their leader has a lower index -- this lane (done above), this workgroup (no barrier between its
    // store and here), or a tile with a lower ticket (running: it needs nothing from this one)
#pragma unroll
    for (int k = 0; k < kFinishItems; k++) {
        const int64_t i = i0 + k;
        if (i < R && lead[k] != (uint32_t)i + 1u) {
            int32_t o;
            while ((o = __hip_atomic_load(&rec_owned[lead[k] - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0)
                __builtin_amdgcn_s_sleep(1);
            rec_owned[i] = o;
        }
    }
}

// c == 4 records in registers: {cell, first point, count, four partial features}
struct Rec4 { int64_t key, first; int32_t cnt; float x0, x1, x2, x3; };
__device__ __forceinline__ Rec4 load_rec4(const int32_t *__restrict__ recv, int RS, uint32_t ri)
{
    const int32_t *r = recv + (size_t)ri * RS;
    Rec4 v;
    v.key = *reinterpret_cast<const int64_t *>(r);
    v.first = *reinterpret_cast<const int64_t *>(r + 2);
    v.cnt = r[4];
    v.x0 = __int_as_float(r[5]); v.x1 = __int_as_float(r[6]); v.x2 = __int_as_float(r[7]); v.x3 = __int_as_float(r[8]);
    return v;
}
struct Acc4 {
    float a0, a1, a2, a3;
    int64_t first;
    int32_t cnt;
    __device__ __forceinline__ void init(int reduction)
    {
        const bool is_sum = reduction == D3D_REDUCE_MEAN || reduction == 4;
        a0 = a1 = a2 = a3 = is_sum ? 0.0f : (reduction == D3D_REDUCE_MAX ? -INFINITY : INFINITY);
        first = INT64_MAX; cnt = 0;
    }
    __device__ __forceinline__ void add(const Rec4 &v, int reduction)      // identity (op) x for the first record: as merge_leader
    {
        const bool is_sum = reduction == D3D_REDUCE_MEAN || reduction == 4;
        if (is_sum) { a0 += v.x0; a1 += v.x1; a2 += v.x2; a3 += v.x3; }
        else if (reduction == D3D_REDUCE_MAX) { a0 = a0 < v.x0 ? v.x0 : a0; a1 = a1 < v.x1 ? v.x1 : a1; a2 = a2 < v.x2 ? v.x2 : a2; a3 = a3 < v.x3 ? v.x3 : a3; }
        else { a0 = v.x0 < a0 ? v.x0 : a0; a1 = v.x1 < a1 ? v.x1 : a1; a2 = v.x2 < a2 ? v.x2 : a2; a3 = v.x3 < a3 ? v.x3 : a3; }
        first = v.first < first ? v.first : first;
        cnt += v.cnt;
    }
    __device__ __forceinline__ void store(const MergeOut &f, int64_t key, int64_t o, int64_t i) const
    {
        f.lead_rec[o] = (int32_t)i;
        f.first_o[o] = leader_first(f, i);
        f.npoints[o] = cnt;
        f.coords[o * 3 + 0] = key / (f.sy * f.sz);
        f.coords[o * 3 + 1] = (key / f.sz) % f.sy;
        f.coords[o * 3 + 2] = key % f.sz;
        const float d = f.reduction == D3D_REDUCE_MEAN ? (float)cnt : 1.0f;
        *reinterpret_cast<float4 *>(f.feats + o * 4) = make_float4(a0 / d, a1 / d, a2 / d, a3 / d);
    }
};

// ---------------------------------------------------------------- merge on the owner, LDS buckets (3 launches, no global atomics)
// The chain path above pays two RETURNING global atomics per record -- 1.6 M of them at config 5, at the 17 G/s scattered
// returning atomics run at (k_nms_hits) that is its 92 us.  Up to 2 M records the merge is done the way the voxel
// index is built (voxel.hip k_tile_sort / k_bucket_index), with the hash tables in LDS:
//   k_rec_tile_sort  tile of 2048 records -> sorted by hash bucket in LDS, written as {cell, record} entries; a bucket-major
//                    table [bucket][tile] of the runs {offset : 16 | length : 16}
//   k_rec_bucket     one workgroup per bucket (<= 1024 records on average): its runs gathered, cells grouped by an LDS hash
//                    table; per cell the lowest record index (the leader) and the list of its records (fixed segment of
//                    kRecSeg entries per bucket in `cellrecs`); per record one word `rinfo`: a leader's list {base, length},
//                    or the index of its leader.  A bucket with more than kRecBucketCap entries raises BIN_OVERFLOW in
//                    counts[STATUS] (hashed cells: 8 sigma above the largest mean) -- the caller then uses the chain path.
//   k_rec_finish     as k_merge_finish: look-back over the leader flags, each leader merges its list in rank order and writes
//                    its voxel's row; the other records pick up their leader's number.
constexpr int kRecTileThreads = 512, kRecTileItems = 4, kRecTile = kRecTileThreads * kRecTileItems;
constexpr int kRecBucketThreads = 512, kRecSlots = 2048, kRecSeg = 2048, kRecBucketCap = 1280, kRecBucketMean = 1024;
constexpr int kRecMaxBuckets = 2048, kRecMaxTiles = 1024;
constexpr int kRecMulti = kRecBucketCap / 2;       // cells of several records a bucket can hold
constexpr int64_t kRecMaxRecords = (int64_t)kRecMaxTiles * kRecTile;
constexpr uint32_t kRecLeader = 1u << 31;

__device__ __forceinline__ u64 rec_hash(u64 key) { return mix64(key * 0x9e3779b97f4a7c15ull); }   // (owner_of took mix64(key))

// (a bucket's size varies by whole CELLS: with W ranks sharing most cells its spread is W times that of single records -- above
// eight ranks the buckets are made half as large, which keeps 1280 four and more sigma away even at W = 64; a bucket that
// outgrows it all the same hands the call over, tests/owner_merge_fuzz.py)
static uint32_t rec_buckets(int64_t R, int world)
{
    const int64_t mean = world > 8 ? kRecBucketMean / 2 : kRecBucketMean;
    uint32_t nb = 1;
    while ((int64_t)nb * mean < R && nb < (uint32_t)kRecMaxBuckets) nb <<= 1;
    return nb;
}

struct RecWs {
    u64 *ekey;            // [ntiles * kRecTile] tile-sorted entries: cell ...
    uint32_t *eidx;       //                       ... and record index
    uint32_t *table;      // [nb][ntiles]
    uint32_t *cellrecs;   // [nb * kRecSeg]
    uint32_t *rinfo;      // [R]
    uint32_t *minfo;      // [R] by leader record, cells of several records: where the merged record is (index into mrec)
    int32_t *mrec;        // [nb * kRecMulti][RS of c == 4] merged records, laid out like a received record
    u64 *status;          // look-back words of k_rec_finish, [ftiles] + ticket word
    int64_t *src_off;     // [world + 1]
    uint32_t *overflow;   // one word
    uint32_t nb, ntiles, ftiles;
    size_t bytes;
};

static RecWs carve_rec(void *ws, size_t bytes, int64_t R, int world)
{
    WsCarver w(ws, bytes);
    RecWs m;
    const int64_t r1 = R > 0 ? R : 1;
    m.nb = rec_buckets(r1, world);
    m.ntiles = (uint32_t)d3d_divup(r1, kRecTile);
    m.ftiles = (uint32_t)d3d_divup(r1, kFinishTile);
    m.ekey = w.take<u64>((size_t)m.ntiles * kRecTile);
    m.eidx = w.take<uint32_t>((size_t)m.ntiles * kRecTile);
    m.table = w.take<uint32_t>((size_t)m.nb * m.ntiles);
    m.cellrecs = w.take<uint32_t>((size_t)m.nb * kRecSeg);
    m.rinfo = w.take<uint32_t>(r1);
    m.minfo = w.take<uint32_t>(r1);
    m.mrec = w.take<int32_t>((size_t)m.nb * kRecMulti * rec_stride(4));
    m.status = w.take<u64>((size_t)m.ftiles + 1);
    m.src_off = w.take<int64_t>((size_t)world + 1);
    m.overflow = w.take<uint32_t>(1);
    m.bytes = w.off;
    return m;
}

__global__ __launch_bounds__(kRecTileThreads) void k_rec_tile_sort(const int32_t *__restrict__ recv, int64_t R, int RS, uint32_t nb,
                                                                   uint32_t ntiles, u64 *__restrict__ ekey, uint32_t *__restrict__ eidx,
                                                                   uint32_t *__restrict__ table, int32_t *__restrict__ rec_owned,
                                                                   u64 *status, uint32_t nstatus, uint32_t *overflow, int64_t *counts,
                                                                   const int64_t *__restrict__ src_off, int world, int64_t *src_off_copy)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char rec_lds[];
    u64 *keys = reinterpret_cast<u64 *>(rec_lds);                         // [kRecTile] in bucket order
    uint32_t *h = reinterpret_cast<uint32_t *>(keys + kRecTile);           // [nb] histogram, then the buckets' offsets
    uint16_t *lidx = reinterpret_cast<uint16_t *>(h + nb);                 // [kRecTile] record index inside the tile
    __shared__ u64 smem[kRecTileThreads / kWave];
    // (XCD-aware tile numbering as in k_tile_sort of voxel.hip: a bucket's table row is written as adjacent words into one L2)
    const uint32_t per_xcd = (ntiles + 7u) >> 3;
    const uint32_t tile = ntiles >= 64u ? (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    if (tile >= ntiles) return;
    const int64_t base = (int64_t)tile * kRecTile + threadIdx.x;
    u64 key[kRecTileItems];
#pragma unroll
    for (int r = 0; r < kRecTileItems; r++) {
        const int64_t i = base + r * kRecTileThreads;
        key[r] = i < R ? (u64)*reinterpret_cast<const int64_t *>(recv + (size_t)i * RS) : 0ull;
    }
    for (uint32_t b = threadIdx.x; b < nb; b += kRecTileThreads) h[b] = 0;
    if (tile == 0) {                                                 // housekeeping for the launches behind this one
        for (uint32_t t = threadIdx.x; t < nstatus; t += kRecTileThreads) status[t] = 0ull;
        if (threadIdx.x < D3D_NUM_COUNTS) counts[threadIdx.x] = 0;
        if (threadIdx.x <= (unsigned)world) src_off_copy[threadIdx.x] = src_off[threadIdx.x];
        if (threadIdx.x == 0) *overflow = 0;
    }
    __syncthreads();
    uint32_t word[kRecTileItems];                  // bucket : 12 | arrival inside the bucket << 12
#pragma unroll
    for (int r = 0; r < kRecTileItems; r++) {
        const int64_t i = base + r * kRecTileThreads;
        word[r] = ~0u;
        if (i < R) {
            const uint32_t b = (uint32_t)rec_hash(key[r]) & (nb - 1);
            word[r] = b | (atomicAdd(&h[b], 1u) << 12);
            rec_owned[i] = -1;                     // "my leader has no number yet" (k_rec_finish polls it)
        }
    }
    __syncthreads();
    constexpr int kPerMax = kRecMaxBuckets / kRecTileThreads;
    const uint32_t per = nb > (uint32_t)kRecTileThreads ? nb / kRecTileThreads : 1u, b0 = threadIdx.x * per;
    uint32_t cnt[kPerMax];
    u64 mine = 0;
#pragma unroll
    for (int k = 0; k < kPerMax; k++) {
        cnt[k] = ((uint32_t)k < per && b0 + k < nb) ? h[b0 + k] : 0u;
        mine += cnt[k];
    }
    u64 all;
    u64 ex = block_excl_scan_u64<kRecTileThreads>(mine, &all, smem);
#pragma unroll
    for (int k = 0; k < kPerMax; k++) {
        if ((uint32_t)k < per && b0 + k < nb) {
            h[b0 + k] = (uint32_t)ex;
            table[(size_t)(b0 + k) * ntiles + tile] = (uint32_t)ex | (cnt[k] << 16);
            ex += cnt[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRecTileItems; r++) {
        if (word[r] != ~0u) {
            const uint32_t p = h[word[r] & 0xfffu] + (word[r] >> 12);
            keys[p] = key[r];
            lidx[p] = (uint16_t)(r * kRecTileThreads + threadIdx.x);
        }
    }
    __syncthreads();
    const size_t tbase = (size_t)tile * kRecTile;
    for (uint32_t p = threadIdx.x; p < (uint32_t)all; p += kRecTileThreads) {
        ekey[tbase + p] = keys[p];
        eidx[tbase + p] = (uint32_t)tbase + lidx[p];
    }
}

__global__ __launch_bounds__(kRecBucketThreads) void k_rec_bucket(const u64 *__restrict__ ekey, const uint32_t *__restrict__ eidx,
                                                                  const uint32_t *__restrict__ table, uint32_t ntiles,
                                                                  uint32_t *__restrict__ cellrecs, uint32_t *__restrict__ rinfo,
                                                                  uint32_t *overflow, int64_t *counts, uint32_t cap,
                                                                  const int32_t *__restrict__ recv /* c == 4: merge here */,
                                                                  int reduction, uint32_t *__restrict__ minfo, int32_t *__restrict__ mrec)
{
    __shared__ uint16_t mq[kRecMulti];                                   // slots of the cells of several records
    __shared__ uint32_t nmulti;
    __shared__ u64 hkey[kRecSlots];
    __shared__ uint32_t hmin[kRecSlots], hcnt[kRecSlots];                 // hcnt: records of the cell, then {count : 8 | offset << 8}
    __shared__ uint32_t srcpos[kRecBucketCap];                           // where the bucket's entries are; then (llist) the
    uint32_t *llist = srcpos;                                            // cells' record lists as in cellrecs (160 KB / 4 workgroups)
    __shared__ u64 smem[kRecBucketThreads / kWave];
    // (XCD-aware numbering as in k_bucket_index of voxel.hip: the runs of neighbouring buckets share lines inside a tile)
    const uint32_t b = gridDim.x >= 64u ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    // this bucket's runs: one per tile
    constexpr int kRuns = kRecMaxTiles / kRecBucketThreads;
    uint32_t roff[kRuns], rlen[kRuns];
    u64 mine = 0;
#pragma unroll
    for (int k = 0; k < kRuns; k++) {
        const uint32_t t = threadIdx.x * kRuns + k;
        const uint32_t w = t < ntiles ? table[(size_t)b * ntiles + t] : 0u;
        roff[k] = w & 0xffffu; rlen[k] = w >> 16;
        mine += rlen[k];
    }
    for (uint32_t s = threadIdx.x; s < (uint32_t)kRecSlots; s += kRecBucketThreads) { hkey[s] = kFree; hmin[s] = ~0u; hcnt[s] = 0; }
    if (threadIdx.x == 0) nmulti = 0;
    u64 total;
    u64 ex = block_excl_scan_u64<kRecBucketThreads>(mine, &total, smem);
    const uint32_t m = (uint32_t)total;
    if (m > cap) {                                                          // (wave-uniform: every thread sees the same total)
        if (threadIdx.x == 0) {
            *overflow = 1;
            atomicOr(reinterpret_cast<u64 *>(&counts[D3D_COUNT_STATUS]), (u64)D3D_VOXEL_STATUS_BIN_OVERFLOW);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < kRuns; k++) {
        const uint32_t t = threadIdx.x * kRuns + k;
        for (uint32_t q = 0; q < rlen[k]; q++) srcpos[(uint32_t)ex + q] = t * kRecTile + roff[k] + q;
        ex += rlen[k];
    }
    __syncthreads();
    constexpr int kEnt = (kRecBucketCap + kRecBucketThreads - 1) / kRecBucketThreads;
    u64 key[kEnt];
    uint32_t idx[kEnt], slot[kEnt], pos[kEnt];
#pragma unroll
    for (int k = 0; k < kEnt; k++) {
        const uint32_t j = threadIdx.x + k * kRecBucketThreads;
        if (j < m) { const uint32_t sp = srcpos[j]; key[k] = ekey[sp]; idx[k] = eidx[sp]; }
    }
#pragma unroll
    for (int k = 0; k < kEnt; k++) {
        const uint32_t j = threadIdx.x + k * kRecBucketThreads;
        if (j < m) {
            uint32_t s = (uint32_t)(rec_hash(key[k]) >> 32) & (kRecSlots - 1);
            for (;;) {
                const u64 old = atomicCAS(&hkey[s], kFree, key[k]);
                if (old == kFree || old == key[k]) break;
                s = (s + 1) & (kRecSlots - 1);                              // m <= kRecBucketCap < kRecSlots: a free slot exists
            }
            slot[k] = s;
            atomicMin(&hmin[s], idx[k]);
            pos[k] = atomicAdd(&hcnt[s], 1u);
        }
    }
    __syncthreads();
    {   // the cells' list offsets inside the bucket's segment: exclusive scan of the counts over the slots
        constexpr int kPer = kRecSlots / kRecBucketThreads;
        uint32_t c[kPer];
        u64 sum = 0, tot;
#pragma unroll
        for (int k = 0; k < kPer; k++) { c[k] = hcnt[threadIdx.x * kPer + k]; sum += c[k]; }
        u64 e = block_excl_scan_u64<kRecBucketThreads>(sum, &tot, smem);
#pragma unroll
        for (int k = 0; k < kPer; k++) { hcnt[threadIdx.x * kPer + k] = c[k] | ((uint32_t)e << 8); e += c[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kEnt; k++) {
        const uint32_t j = threadIdx.x + k * kRecBucketThreads;
        if (j < m) {
            const uint32_t w = hcnt[slot[k]], L = w & 0xffu, off = w >> 8, lead = hmin[slot[k]];
            const uint32_t lbase = b * kRecSeg + off;
            cellrecs[lbase + pos[k]] = idx[k];
            llist[off + pos[k]] = idx[k];
            rinfo[idx[k]] = lead == idx[k] ? (kRecLeader | ((L - 1u) << 22) | lbase) : lead;
            if (recv && lead == idx[k] && L > 1u) mq[atomicAdd(&nmulti, 1u)] = (uint16_t)slot[k];
        }
    }
    if (!recv) return;
    __syncthreads();
    // cells of several records, 4 features: merged HERE, in rank order, into a record of the same layout -- k_rec_finish then
    // reads one record per leader whatever the cell (walking the lists there made every wavefront wait for its longest
    // cell, row after row: 60 of its 103 us at config 5, 30 with the lists sorted in registers).  One such cell per thread
    // (queued above: a few dozen per bucket), so the bucket pays one chain of loads, not one per entry of a thread.
    const int RS = rec_stride(4);
    for (uint32_t jm = threadIdx.x; jm < nmulti; jm += kRecBucketThreads) {
        const uint32_t sl = mq[jm], w = hcnt[sl], L = w & 0xffu, off = w >> 8;
        Acc4 acc;
        acc.init(reduction);
        uint32_t prev = 0;                                               // index + 1 of the record added last
        for (uint32_t n0 = 0; n0 < L; n0 += 4) {                         // the next four in index order, loaded together
            uint32_t e[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t best = ~0u;
                for (uint32_t t = 0; t < L; t++) {
                    const uint32_t x = llist[off + t] + 1u;
                    if (x > prev && x < best) best = x;
                }
                e[q] = best;
                if (best != ~0u) prev = best;
            }
            Rec4 v[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (e[q] != ~0u) v[q] = load_rec4(recv, RS, e[q] - 1u);
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (e[q] != ~0u) acc.add(v[q], reduction);
        }
        const uint32_t at = b * kRecMulti + jm;
        int32_t *r = mrec + (size_t)at * RS;
        *reinterpret_cast<int64_t *>(r) = (int64_t)hkey[sl];
        *reinterpret_cast<int64_t *>(r + 2) = acc.first;
        r[4] = acc.cnt;
        r[5] = __float_as_int(acc.a0); r[6] = __float_as_int(acc.a1); r[7] = __float_as_int(acc.a2); r[8] = __float_as_int(acc.a3);
        minfo[hmin[sl]] = at;
    }
}

// a leader's records by ascending record index (selection over its short list: the arrival order in the list is arbitrary)
template <class F>
__device__ __forceinline__ void list_in_order(const uint32_t *__restrict__ list, uint32_t L, F &&visit)
{
    uint32_t prev = 0;                            // index + 1 of the record visited last
    for (uint32_t n = 0; n < L; n++) {
        uint32_t best = ~0u;
        for (uint32_t q = 0; q < L; q++) {
            const uint32_t e = list[q] + 1u;
            if (e > prev && e < best) best = e;
        }
        visit(best - 1u);
        prev = best;
    }
}

// any width (c != 4: the lists are walked here)
__device__ __noinline__ void merge_listed(const MergeOut &f, const int32_t *__restrict__ recv, const uint32_t *__restrict__ list,
                                          uint32_t L, int64_t i, int64_t o)
{
    merge_leader(f, recv, i, o, L == 1u, [&](auto &&visit) { list_in_order(list, L, visit); });
}

__global__ __launch_bounds__(kFinishThreads) void k_rec_finish(const int32_t *__restrict__ recv, int64_t R,
                                                               const uint32_t *__restrict__ rinfo,
                                                               const uint32_t *__restrict__ cellrecs, const uint32_t *__restrict__ minfo,
                                                               const int32_t *__restrict__ mrec, u64 *status, uint32_t ntiles,
                                                               const uint32_t *__restrict__ overflow, MergeOut f, int32_t *rec_owned,
                                                               int64_t *counts)
{
    __shared__ unsigned int sid;
    __shared__ u64 smem[kFinishThreads / kWave];
    __shared__ u64 sexcl;
    if (*overflow) return;                        // a bucket did not fit: rinfo is incomplete, the caller repeats on the chain path
    const unsigned int tile = lookback_ticket(reinterpret_cast<unsigned int *>(status + ntiles), &sid);
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    // row k of a wavefront = 64 consecutive records, one per lane: coalesced reads, and consecutive leaders write consecutive rows
    const int64_t base = (int64_t)tile * kFinishTile + (int64_t)w * (kWave * kFinishItems) + lane;
    uint32_t info[kFinishItems];
    u64 ex[kFinishItems], carry = 0;
#pragma unroll
    for (int k = 0; k < kFinishItems; k++) {
        const int64_t i = base + (int64_t)k * kWave;
        info[k] = i < R ? rinfo[i] : 0u;
        const u64 v = (i < R && (info[k] & kRecLeader)) ? 1ull : 0ull;
        const u64 incl = wave_incl_scan_u64(v);
        ex[k] = carry + incl - v;
        carry += __shfl(incl, kWave - 1, kWave);
    }
    if (lane == 0) smem[w] = carry;
    __syncthreads();
    u64 woff = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kFinishThreads / kWave; k++) { if (k < w) woff += smem[k]; total += smem[k]; }
    if (threadIdx.x < kWave) {
        const u64 e = lookback_exclusive(status, tile, total);
        if (threadIdx.x == 0) {
            sexcl = e;
            if (tile == ntiles - 1) counts[D3D_COUNT_VOXELS] = (int64_t)(e + total);
        }
    }
    __syncthreads();
    woff += sexcl;
    // c == 4: one record per leader -- its own, or the merged record k_rec_bucket left for a cell of several; the records of
    // all rows are loaded before the first is used
    const bool fast4 = f.c == 4;
    uint32_t mi[kFinishItems];
#pragma unroll
    for (int k = 0; k < kFinishItems; k++) {
        const int64_t i = base + (int64_t)k * kWave;
        mi[k] = ~0u;
        if (fast4 && i < R && (info[k] & kRecLeader) && ((info[k] >> 22) & 0xffu) != 0u) mi[k] = minfo[i];
    }
    Rec4 one[kFinishItems];
#pragma unroll
    for (int k = 0; k < kFinishItems; k++) {
        const int64_t i = base + (int64_t)k * kWave;
        if (fast4 && i < R && (info[k] & kRecLeader)) one[k] = mi[k] == ~0u ? load_rec4(recv, f.RS, (uint32_t)i) : load_rec4(mrec, f.RS, mi[k]);
    }
#pragma unroll
    for (int k = 0; k < kFinishItems; k++) {
        const int64_t i = base + (int64_t)k * kWave;
        if (fast4 && i < R && (info[k] & kRecLeader)) {
            const int64_t o = (int64_t)(woff + ex[k]);
            Acc4 acc;
            acc.init(f.reduction);
            acc.add(one[k], f.reduction);
            acc.store(f, one[k].key, o, i);
            __hip_atomic_store(&rec_owned[i], (int32_t)o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!fast4) {
#pragma unroll
        for (int k = 0; k < kFinishItems; k++) {
            const int64_t i = base + (int64_t)k * kWave;
            if (i < R && (info[k] & kRecLeader)) {
                const int64_t o = (int64_t)(woff + ex[k]);
                merge_listed(f, recv, cellrecs + (info[k] & 0x3fffffu), ((info[k] >> 22) & 0xffu) + 1u, i, o);
                __hip_atomic_store(&rec_owned[i], (int32_t)o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // the other records: their leader has a lower index -- this wavefront (done above), this workgroup (no barrier between its
    // store and here), or a tile with a lower ticket (running: it needs nothing from this one)
#pragma unroll
    for (int k = 0; k < kFinishItems; k++) {
        const int64_t i = base + (int64_t)k * kWave;
        if (i < R && !(info[k] & kRecLeader)) {
            int32_t o;
            while ((o = __hip_atomic_load(&rec_owned[info[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0)
                __builtin_amdgcn_s_sleep(1);
            rec_owned[i] = o;
        }
    }
}

// ---------------------------------------------------------------- numbering
// first_o ascends with the owned voxel index, so a wavefront's 64 bits fall into a handful of words: the lanes that share a
// word OR their bits together (the lowest of them issues the atomic)
__global__ __launch_bounds__(256) void k_first_mark(const int64_t *__restrict__ first_o, const int64_t *__restrict__ counts,
                                                    int64_t n_total, u64 *bitmap, int64_t nw)
{
    const int64_t Vo = counts[D3D_COUNT_VOXELS];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && (counts[D3D_COUNT_STATUS] & D3D_VOXEL_STATUS_BIN_OVERFLOW)) bitmap[nw] = 1ull;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t f = i < Vo ? first_o[i] : -1;
    const bool ok = f >= 0 && f < n_total;
    const int64_t word = ok ? (f >> 6) : -1 - lane;          // lanes without a bit: a word of their own, never written
    u64 bits = ok ? 1ull << (f & 63) : 0ull;
    // segmented OR over runs of equal words (ascending input: equal words are adjacent)
    bool head = lane == 0 || __shfl_up(word, 1, kWave) != word;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const u64 other = __shfl_down(bits, d, kWave);
        const int64_t ow = __shfl_down(word, d, kWave);
        if (lane + d < kWave && ow == word) bits |= other;
    }
    if (ok && head) atomicOr(&bitmap[word], bits);
}

// popcount prefix of the all-reduced bitmap (every owner's first points): one launch, tiles of 4096 words by ticket, decoupled
// look-back over the tiles' popcounts (a frame of 8 M points: 31 tiles); the last tile writes the frame's voxel count.
// status[ntiles] + the ticket word are zeroed by the caller.  (Round 3: the three-launch generic scan, 23 us for 1 MB.)
constexpr int kPrefixThreads = 1024, kPrefixItems = 4, kPrefixTile = kPrefixThreads * kPrefixItems;
__global__ __launch_bounds__(kPrefixThreads) void k_first_prefix(const u64 *__restrict__ global, int64_t nw, uint32_t *__restrict__ pre,
                                                                 u64 *status, uint32_t ntiles, int64_t *counts_out)
{
    __shared__ unsigned int sid;
    __shared__ u64 smem[kPrefixThreads / kWave];
    __shared__ u64 sexcl;
    const unsigned int tile = lookback_ticket(reinterpret_cast<unsigned int *>(status + ntiles), &sid);
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    // row k of a wavefront = 64 consecutive words, one per lane (coalesced)
    const int64_t base = (int64_t)tile * kPrefixTile + (int64_t)w * (kWave * kPrefixItems) + lane;
    u64 v[kPrefixItems], ex[kPrefixItems], carry = 0;
#pragma unroll
    for (int k = 0; k < kPrefixItems; k++) {
        const int64_t i = base + (int64_t)k * kWave;
        v[k] = i < nw ? (u64)__popcll(global[i]) : 0ull;
        const u64 incl = wave_incl_scan_u64(v[k]);
        ex[k] = carry + incl - v[k];
        carry += __shfl(incl, kWave - 1, kWave);
    }
    if (lane == 0) smem[w] = carry;
    __syncthreads();
    u64 woff = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kPrefixThreads / kWave; k++) { if (k < w) woff += smem[k]; total += smem[k]; }
    if (threadIdx.x < kWave) {
        const u64 e = lookback_exclusive(status, tile, total);
        if (threadIdx.x == 0) {
            sexcl = e;
            if (tile == ntiles - 1) {
                counts_out[D3D_COUNT_VOXELS] = (int64_t)(e + total);
                counts_out[D3D_COUNT_STATUS] = global[nw] ? D3D_VOXEL_STATUS_BIN_OVERFLOW : 0;   // some owner's merge (see k_first_mark)
                counts_out[D3D_COUNT_POINTS] = 0; counts_out[D3D_COUNT_AUX] = 0;
            }
        }
    }
    __syncthreads();
    woff += sexcl;
#pragma unroll
    for (int k = 0; k < kPrefixItems; k++) {
        const int64_t i = base + (int64_t)k * kWave;
        if (i < nw) pre[i] = (uint32_t)(woff + ex[k]);
    }
}

// owned voxel (already in id order) -> its global voxel id = number of first points before its own in the whole frame
__global__ __launch_bounds__(256) void k_owner_number(const int64_t *__restrict__ counts_o, const u64 *__restrict__ gbits,
                                                      const uint32_t *__restrict__ pre, const int64_t *__restrict__ first_o,
                                                      int64_t *vids)
{
    const int64_t Vo = counts_o[D3D_COUNT_VOXELS];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Vo) return;
    const int64_t f = first_o[i];
    vids[i] = (int64_t)pre[f >> 6] + __popcll(gbits[f >> 6] & ((1ull << (f & 63)) - 1ull));
}

__global__ __launch_bounds__(256) void k_owner_reply(int64_t R, const int32_t *__restrict__ rec_owned,
                                                     const int64_t *__restrict__ vids, int64_t *reply)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) {
        const int32_t o = rec_owned[i];
        reply[i] = o >= 0 ? vids[o] : -1;         // (-1: the merge asked to be repeated, d3d_owner_merge)
    }
}

// the ids that came back (send order) -> id of every local point: point -> local voxel -> its place in the send buffer
__global__ __launch_bounds__(256) void k_owner_map(int64_t n, const int64_t *__restrict__ local_map,
                                                   const int32_t *__restrict__ pos_of_local, const int64_t *__restrict__ back,
                                                   int64_t *gmap)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t v = local_map[i];
    gmap[i] = v < 0 ? -1 : back[pos_of_local[v]];
}

// ---------------------------------------------------------------- dense contract on the owner: voxels[Vo, P, 4] + pmask
// The first P points of a voxel by GLOBAL index (voxelize.cpp:128-134): the ranks' candidate rows (each the first
// min(count, P) points of the voxel inside that shard, in point order) taken in rank order until P are filled -- shards are
// contiguous point ranges in rank order, so a lower rank's points all come first.  The shape of k_emit (voxel.hip): a
// wavefront owns 64 consecutive owned voxels = one contiguous stretch of the outputs; lane j walks voxel j's records in rank
// order and copies its rows into a row buffer in LDS; the stretch is then written 1 KiB per store instruction from LDS.
constexpr int kDenseCap = 256;                    // rows per wavefront in LDS; max_points <= kDenseCap

// BUCKETS: a voxel's records are the list its leader's rinfo word names; else the chain of its slot
// RESIDENT: voxels / row_state are kept by the caller from frame to frame (see k_emit<.., RESIDENT> in voxel.hip and
// d3d_voxelize_3d_dense_resident): only rows below max(kept, row_state[v]) are stored, row_state[v] <- kept.
template <bool BUCKETS, bool RESIDENT = false>
__global__ __launch_bounds__(256) void k_owner_dense(const int64_t *__restrict__ counts_o, const int32_t *__restrict__ lead_rec,
                                                     const int32_t *__restrict__ npoints, const uint32_t *__restrict__ rec_slot,
                                                     const MergeSlot *__restrict__ mslot, const uint32_t *__restrict__ next,
                                                     const uint32_t *__restrict__ rinfo, const uint32_t *__restrict__ cellrecs,
                                                     const int64_t *__restrict__ src_off, int world,
                                                     const int32_t *__restrict__ recv, int RS, const float4 *__restrict__ recv_rows,
                                                     const int64_t *__restrict__ rows_src_off, uint32_t P, int pshift,
                                                     float4 *voxels, unsigned char *pmask, uint16_t *row_state)
{
    typedef float vec4 __attribute__((ext_vector_type(4)));
    __shared__ vec4 rowbuf_all[256 / kWave][kDenseCap];
    __shared__ uint32_t off_all[256 / kWave][kWave];
    __shared__ uint16_t kept_all[256 / kWave][kWave];
    __shared__ uint16_t lim_all[RESIDENT ? 256 / kWave : 1][kWave];
    __shared__ uint32_t loff_all[RESIDENT ? 256 / kWave : 1][kWave];
    __shared__ int64_t so[kMaxWorld + 1], rso[kMaxWorld + 1];          // records / rows per source rank, exclusive prefixes
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    if (threadIdx.x <= (unsigned)world) { so[threadIdx.x] = src_off[threadIdx.x]; rso[threadIdx.x] = rows_src_off[threadIdx.x]; }
    __syncthreads();
    vec4 *rowbuf = rowbuf_all[w];
    uint32_t *sh_off = off_all[w];
    uint16_t *sh_kept = kept_all[w];
    const int64_t Vo = counts_o[D3D_COUNT_VOXELS];
    const int64_t v0 = ((int64_t)blockIdx.x * (256 / kWave) + w) * kWave;
    if (v0 >= Vo) return;                                   // wave-uniform
    const uint32_t nv = Vo - v0 < kWave ? (uint32_t)(Vo - v0) : (uint32_t)kWave;
    const bool mine = (uint32_t)lane < nv;
    uint32_t head = 0, kept = 0, lead = 0;
    if (mine) {
        lead = (uint32_t)lead_rec[v0 + lane];
        head = BUCKETS ? rinfo[lead] : mslot[rec_slot[lead]].head;
        const uint32_t n = (uint32_t)npoints[v0 + lane];
        kept = n < P ? n : P;
    }
    uint32_t incl = kept;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, d, kWave);
        if (lane >= d) incl += t;
    }
    const uint32_t off = incl - kept;
    sh_off[lane] = off; sh_kept[lane] = (uint16_t)kept;
    uint16_t *sh_lim = lim_all[RESIDENT ? w : 0];
    uint32_t *sh_loff = loff_all[RESIDENT ? w : 0];
    uint32_t loff = 0, lim = 0;
    if (RESIDENT) {
        if (mine) {
            const uint32_t prev = row_state[v0 + lane];
            lim = prev > kept ? prev : kept;
            if (lim > P) lim = P;
            if (prev != kept) row_state[v0 + lane] = (uint16_t)kept;
        }
        uint32_t li = lim;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)li, d, kWave);
            if (lane >= d) li += t;
        }
        loff = li - lim;
        sh_lim[lane] = (uint16_t)lim; sh_loff[lane] = loff;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    vec4 *out = reinterpret_cast<vec4 *>(voxels) + v0 * (int64_t)P;
    const vec4 zero = {0.f, 0.f, 0.f, 0.f};
    uint32_t ja = 0;
    while (ja < nv) {                                       // wave-uniform: one batch unless the rows exceed the buffer
        const uint32_t oa = (uint32_t)__shfl((int)off, (int)ja, kWave);
        const bool fits = (uint32_t)lane >= ja && mine && incl - oa <= (uint32_t)kDenseCap;
        const unsigned long long nf = ~(__ballot(fits) >> ja);
        const uint32_t jb = ja + (nf ? (uint32_t)__ffsll((long long)nf) - 1u : (uint32_t)kWave - ja);
        if ((uint32_t)lane >= ja && (uint32_t)lane < jb) {  // my voxel's rows, in rank order of the contributing records
            vec4 *dst = rowbuf + (off - oa);
            uint32_t have = 0;
            auto take_known = [&](uint32_t ri, uint32_t rcnt, uint32_t roff) {      // record ri holds rcnt points, its rows start at roff
                if (have >= kept) return;
                const uint32_t nr = rcnt < P ? rcnt : P;
                const uint32_t take = nr < kept - have ? nr : kept - have;
                const vec4 *src = reinterpret_cast<const vec4 *>(recv_rows) + rso[record_source(so, world, ri)] + roff;
                for (uint32_t t = 0; t < take; t++) dst[have + t] = src[t];
                have += take;
            };
            auto take_rows = [&](uint32_t ri) {
                const int32_t *r = recv + (size_t)ri * RS;
                take_known(ri, (uint32_t)r[4], (uint32_t)r[RS - 1]);
            };
            if (BUCKETS) {
                const uint32_t L = ((head >> 22) & 0xffu) + 1u;
                const uint32_t *list = cellrecs + (head & 0x3fffffu);
                if (L == 1u) take_rows(lead);                           // (the cell's only record is its leader: no list to read)
                else if (L <= 8u) {
                    // the list sorted in registers, the records' counts and row offsets loaded TOGETHER, then taken in rank order:
                    // two round trips whatever the length (walking the list by repeated selection, a load per step, was 52 of
                    // this kernel's 147 us at config 5: every wavefront waits for its longest cell)
                    uint32_t e[8], rc[4], ro[4];
#pragma unroll
                    for (int q = 0; q < 8; q++) e[q] = (uint32_t)q < L ? list[q] : ~0u;
#define D3D_CE(a, b) { const uint32_t lo_ = e[a] < e[b] ? e[a] : e[b], hi_ = e[a] < e[b] ? e[b] : e[a]; e[a] = lo_; e[b] = hi_; }
                    D3D_CE(0, 1) D3D_CE(2, 3) D3D_CE(4, 5) D3D_CE(6, 7)
                    D3D_CE(0, 2) D3D_CE(1, 3) D3D_CE(4, 6) D3D_CE(5, 7)
                    D3D_CE(1, 2) D3D_CE(5, 6)
                    D3D_CE(0, 4) D3D_CE(1, 5) D3D_CE(2, 6) D3D_CE(3, 7)
                    D3D_CE(2, 4) D3D_CE(3, 5)
                    D3D_CE(1, 2) D3D_CE(3, 4) D3D_CE(5, 6)
#undef D3D_CE
#pragma unroll
                    for (int h = 0; h < 8; h += 4) {        // four records' fields in flight at a time (eight: 102 VGPRs, half the occupancy)
                        if (h == 4 && e[4] == ~0u) break;
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (e[h + q] != ~0u) {
                                const int32_t *r = recv + (size_t)e[h + q] * RS;
                                rc[q] = (uint32_t)r[4]; ro[q] = (uint32_t)r[RS - 1];
                            }
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (e[h + q] != ~0u) take_known(e[h + q], rc[q], ro[q]);
                    }
                } else list_in_order(list, L, take_rows);
            } else chain_in_order(next, head, take_rows);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (RESIDENT) {                                     // the rows to store, flat over the batch's voxels
            const uint32_t l0 = (uint32_t)__shfl((int)loff, (int)ja, kWave);
            const uint32_t l1 = jb < (uint32_t)kWave ? (uint32_t)__shfl((int)loff, (int)jb, kWave)
                                                    : (uint32_t)__shfl((int)(loff + lim), kWave - 1, kWave);
            for (uint32_t t0 = l0; t0 < l1; t0 += kWave) {
                const uint32_t t = t0 + lane;
                if (t < l1) {
                    uint32_t lo = ja, hi = jb;              // largest j in [ja, jb) with loff[j] <= t
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (sh_loff[mid] <= t) lo = mid; else hi = mid;
                    }
                    const uint32_t sl = t - sh_loff[lo];
                    vec4 val = zero;
                    if (sl < sh_kept[lo]) val = rowbuf[sh_off[lo] - oa + sl];
                    out[lo * P + sl] = val;
                }
            }
        }
        const uint32_t q1 = RESIDENT ? 0u : jb * P;
        for (uint32_t q0 = ja * P; q0 < q1; q0 += 4 * kWave) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t q = q0 + u * kWave + lane;
                if (q < q1) {
                    const uint32_t j = pshift >= 0 ? (q >> pshift) : q / P;
                    const uint32_t sl = q - j * P;
                    vec4 val = zero;
                    if (sl < sh_kept[j]) val = rowbuf[sh_off[j] - oa + sl];
                    __builtin_nontemporal_store(val, &out[q]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ja = jb;
    }
    // pmask[nv][P] bytes, contiguous over the stretch: 16-byte pieces when P and the base allow it (one store per 16 flags
    // instead of sixteen)
    unsigned char *pdst = pmask + v0 * (int64_t)P;
    if ((P & 15u) == 0 && (reinterpret_cast<uintptr_t>(pmask) & 15) == 0) {
        typedef uint32_t uvec4 __attribute__((ext_vector_type(4)));
        const uint32_t per = P >> 4, total = nv * per;
        uvec4 *p16 = reinterpret_cast<uvec4 *>(pdst);
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
            __builtin_nontemporal_store(w4, &p16[t]);
        }
        return;
    }
    for (uint32_t t = lane; t < nv * P; t += kWave) {
        const uint32_t j = pshift >= 0 ? (t >> pshift) : t / P;
        pdst[t] = (t - j * P) < sh_kept[j] ? 1 : 0;
    }
}

// replicate: rows of all owners (any order) -> voxel-id order
__global__ __launch_bounds__(256) void k_owner_replicate(int64_t V, const int64_t *__restrict__ vids,
                                                         const int64_t *__restrict__ coords_in, const int32_t *__restrict__ cnt_in,
                                                         const float *__restrict__ feats_in, int c, int64_t *coords, int32_t *cnt,
                                                         float *feats)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const int64_t v = vids[i];
    coords[v * 3] = coords_in[i * 3]; coords[v * 3 + 1] = coords_in[i * 3 + 1]; coords[v * 3 + 2] = coords_in[i * 3 + 2];
    cnt[v] = cnt_in[i];
    if (c == 4 && ((reinterpret_cast<uintptr_t>(feats) | reinterpret_cast<uintptr_t>(feats_in)) & 15) == 0)
        reinterpret_cast<float4 *>(feats)[v] = reinterpret_cast<const float4 *>(feats_in)[i];      // one 16-byte piece, not four
    else
        for (int q = 0; q < c; q++) feats[v * c + q] = feats_in[i * c + q];
}

// The same when the rows arrive as the ranks' blocks one after the other (src_off) and every block is in ascending id order --
// what the all-gather of the owners' results is: an 8-way merge.  A workgroup owns 1024 consecutive OUTPUT rows: it finds, by
// binary search in every block, the stretch of that block whose ids fall into its window (every id exists exactly once), reads
// those stretches (contiguous), places the rows in LDS by id and writes its window out in one piece -- both sides coalesced,
// where the scatter above wrote 5.9 M rows of 44 bytes to scattered places (348 us at config 5).
constexpr int kRepTile = 1024, kRepMaxC = 8;
// bounds[t * world + s] = first row of block s whose id is >= t * kRepTile (t = 0 .. tiles): one lane per (window edge, block),
// 20 dependent loads each, all of them in flight at once (inside the merge kernel the same searches were 25 us of latency per
// workgroup, eleven rounds of workgroups: 274 us)
__global__ __launch_bounds__(256) void k_owner_replicate_bounds(int64_t V, const int64_t *__restrict__ vids,
                                                                const int64_t *__restrict__ src_off, int world, int64_t ntiles,
                                                                int64_t *__restrict__ bounds)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (ntiles + 1) * world) return;
    const int64_t t = g / world;
    const int s = (int)(g - t * world);
    const int64_t x = t * kRepTile < V ? t * kRepTile : V;
    int64_t l = src_off[s], h = src_off[s + 1];
    while (l < h) { const int64_t mid = (l + h) >> 1; if (vids[mid] < x) l = mid + 1; else h = mid; }
    bounds[g] = l;
}

template <int MAXC>
__global__ __launch_bounds__(256) void k_owner_replicate_merge(int64_t V, const int64_t *__restrict__ vids,
                                                               const int64_t *__restrict__ coords_in, const int32_t *__restrict__ cnt_in,
                                                               const float *__restrict__ feats_in, int c, int64_t *coords, int32_t *cnt,
                                                               float *feats, const int64_t *__restrict__ bounds, int world)
{
    __shared__ int64_t lo[kMaxWorld];
    __shared__ uint32_t pre[kMaxWorld + 1];
    __shared__ int64_t cbuf[kRepTile * 3];
    __shared__ int32_t nbuf[kRepTile];
    __shared__ __attribute__((aligned(16))) float fbuf[kRepTile * MAXC];   // (44 KB per workgroup at four features: three per CU)
    const int64_t v0 = (int64_t)blockIdx.x * kRepTile, v1 = v0 + kRepTile < V ? v0 + kRepTile : V;
    if ((int)threadIdx.x < world) {
        const int64_t l0 = bounds[(int64_t)blockIdx.x * world + threadIdx.x];
        lo[threadIdx.x] = l0;
        pre[threadIdx.x + 1] = (uint32_t)(bounds[((int64_t)blockIdx.x + 1) * world + threadIdx.x] - l0);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        pre[0] = 0;
        for (int s = 0; s < world; s++) pre[s + 1] += pre[s];
    }
    __syncthreads();
    const uint32_t total = pre[world];                      // = v1 - v0
    for (uint32_t r = threadIdx.x; r < total; r += 256) {
        int sl = 0, sh = world;                             // the source whose stretch holds row r
        while (sh - sl > 1) { const int mid = (sl + sh) >> 1; if (pre[mid] <= r) sl = mid; else sh = mid; }
        const int64_t i = lo[sl] + (r - pre[sl]);
        const uint32_t slot = (uint32_t)(vids[i] - v0);
        cbuf[slot * 3] = coords_in[i * 3]; cbuf[slot * 3 + 1] = coords_in[i * 3 + 1]; cbuf[slot * 3 + 2] = coords_in[i * 3 + 2];
        nbuf[slot] = cnt_in[i];
        if (c == 4 && (reinterpret_cast<uintptr_t>(feats_in) & 15) == 0)
            *reinterpret_cast<float4 *>(&fbuf[slot * 4]) = reinterpret_cast<const float4 *>(feats_in)[i];
        else
            for (int q = 0; q < c; q++) fbuf[slot * c + q] = feats_in[i * c + q];
    }
    __syncthreads();
    const uint32_t nrows = (uint32_t)(v1 - v0);
    for (uint32_t t = threadIdx.x; t < nrows * 3; t += 256) coords[v0 * 3 + t] = cbuf[t];
    for (uint32_t t = threadIdx.x; t < nrows; t += 256) cnt[v0 + t] = nbuf[t];
    for (uint32_t t = threadIdx.x; t < nrows * (uint32_t)c; t += 256) feats[v0 * c + t] = fbuf[t];
}

inline unsigned blocks_for(int64_t n, int per = 256) { return (unsigned)std::max<int64_t>(1, d3d_divup(n, per)); }

}  // namespace

// ====================================================================== C ABI
extern "C" int d3d_owner_record_words(int32_t c) { return rec_stride(c); }

extern "C" size_t d3d_owner_pack_workspace_bytes(int64_t n, int32_t world)
{
    const int64_t ntiles = d3d_divup(n > 0 ? n : 1, kPackTile);
    return 2 * (d3d_align_up((size_t)ntiles * world * 4) + d3d_align_up((size_t)(world + 1) * 4)) + 256;
}

// local voxels (outputs of d3d_voxelize_3d_reduce: keys[n + 1], cnt[n], agg[n, c], first[n], counts) -> records grouped by
// owner rank: send[n, d3d_owner_record_words(c)] (int32 words), perm[n] (send position -> local voxel), pos_of_local[n] (its inverse),
// send_counts[2 world + 1] (device): records per destination, the shard's status bits, rows per destination.
// Dense contract (max_points > 0; c == 4): seg_base / rows_local as left by d3d_voxelize_3d_reduce(max_points, ...) ->
// send_rows[kept rows, 4] with the same grouping; a record's last word = offset of its rows inside its batch.  `points` /
// `index_offset` = the shard d3d_voxelize_3d_reduce saw: when that call left ranked point indices instead of rows (its counts say
// so) the rows are gathered from the points here; NULL: rows_local must hold rows.
extern "C" int d3d_owner_pack(const int64_t *keys, const int32_t *cnt, const float *agg, const int64_t *first,
                              const int64_t *counts, int64_t n, int32_t c, int32_t world, int32_t max_points,
                              const uint32_t *seg_base, const float *rows_local, int32_t *send, int32_t *perm,
                              int32_t *pos_of_local, float *send_rows, int64_t *send_counts, void *workspace,
                              size_t workspace_bytes, void *stream, const float *points, int64_t index_offset)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || c < 1 || c > 16 || world < 1 || world > kMaxWorld || !counts || !send_counts || max_points < 0) return D3D_ERR_BAD_ARG;
    if (n > 0 && (!keys || !cnt || !agg || !first || !send || !perm || !pos_of_local)) return D3D_ERR_BAD_ARG;
    const bool dense = max_points > 0;
    if (dense && (c != 4 || (n > 0 && (!seg_base || !rows_local || !send_rows)))) return D3D_ERR_BAD_ARG;
    if (dense && ((reinterpret_cast<uintptr_t>(rows_local) | reinterpret_cast<uintptr_t>(send_rows)) & 15)) return D3D_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < d3d_owner_pack_workspace_bytes(n, world)) return D3D_ERR_WORKSPACE;
    const uint32_t ntiles = (uint32_t)d3d_divup(n > 0 ? n : 1, kPackTile);
    WsCarver w(workspace, workspace_bytes);
    uint32_t *tilecnt = w.take<uint32_t>((size_t)ntiles * world);
    uint32_t *dest_base = w.take<uint32_t>(world + 1);
    uint32_t *tilerows = w.take<uint32_t>((size_t)ntiles * world);
    uint32_t *dest_rowbase = w.take<uint32_t>(world + 1);
    if (points && (reinterpret_cast<uintptr_t>(points) & 15)) return D3D_ERR_BAD_ARG;
    PackDense pd{(uint32_t)max_points, seg_base, reinterpret_cast<const float4 *>(rows_local), reinterpret_cast<const float4 *>(points),
                 index_offset, reinterpret_cast<float4 *>(send_rows), tilerows, dest_rowbase};
    if (n > 0) {
        if (dense) D3D_LAUNCH("k_owner_count", k_owner_count<true>, dim3(ntiles), dim3(1024), 0, st, keys, cnt, counts, (uint32_t)world, tilecnt, pd);
        else D3D_LAUNCH("k_owner_count", k_owner_count<false>, dim3(ntiles), dim3(1024), 0, st, keys, cnt, counts, (uint32_t)world, tilecnt, pd);
    } else {
        D3D_HIP_CHECK(hipMemsetAsync(tilecnt, 0, (size_t)ntiles * world * 4, st));
        D3D_HIP_CHECK(hipMemsetAsync(tilerows, 0, (size_t)ntiles * world * 4, st));
    }
    // up to kFusedEntries matrix words every scatter workgroup adds up its own offsets (one launch less: -8 us at config 5)
    const bool fused = n > 0 && (uint64_t)ntiles * (uint64_t)world <= (uint64_t)kFusedEntries;
    const int64_t *status_key = keys ? keys + n : (const int64_t *)nullptr;
    if (!fused)
        D3D_LAUNCH("k_owner_offsets", k_owner_offsets, dim3(1), dim3(1024), 0, st, tilecnt, dense ? tilerows : (uint32_t *)nullptr, ntiles,
                   (uint32_t)world, send_counts, dest_base, dest_rowbase, status_key);
    if (n > 0) {
        if (dense)
            D3D_LAUNCH("k_owner_scatter", k_owner_scatter<true>, dim3(ntiles), dim3(1024), 0, st, keys, cnt, agg, first, counts, (int)c,
                       (uint32_t)world, tilecnt, dest_base, send, perm, pos_of_local, pd, fused ? ntiles : 0u, send_counts, status_key);
        else
            D3D_LAUNCH("k_owner_scatter", k_owner_scatter<false>, dim3(ntiles), dim3(1024), 0, st, keys, cnt, agg, first, counts, (int)c,
                       (uint32_t)world, tilecnt, dest_base, send, perm, pos_of_local, pd, fused ? ntiles : 0u, send_counts, status_key);
    }
    return D3D_OK;
}

static bool merge_on_chains(int64_t R, uint32_t flags) { return (flags & D3D_OWNER_MERGE_CHAINS) || R > kRecMaxRecords; }

extern "C" size_t d3d_owner_merge_workspace_bytes(int64_t R, int32_t world)
{
    const int64_t r = R > 0 ? R : 0;
    const size_t chains = carve_merge(nullptr, 0, r, world).bytes;
    const size_t buckets = r <= kRecMaxRecords ? carve_rec(nullptr, 0, r, world).bytes : 0;
    return std::max(chains, buckets) + 256;
}

// received records recv[R, words] (grouped by source rank: src_off[world + 1], device) -> this owner's voxels IN GLOBAL ID
// ORDER, finished: first_o / coords / npoints / feats [R rows, counts[D3D_COUNT_VOXELS] valid], rec_owned[R] = the owned voxel
// of every record, lead_rec[R] = the leader record of every owned voxel.  reduction: MEAN (sums in rank order, then the division
// of voxelize.cpp:164), MAX, MIN.  The workspace keeps the cells' record lists (for d3d_owner_dense, same flags) until it is
// reused.  flags: D3D_OWNER_MERGE_CHAINS = the general path (global hash table; taken by itself above 2 M records);
// otherwise counts[D3D_COUNT_STATUS] may come back with D3D_VOXEL_STATUS_BIN_OVERFLOW (nothing else is valid then): repeat
// with the flag.  D3D_OWNER_MERGE_TEST_TINY = test hook, buckets overflow at 4 records.
extern "C" int d3d_owner_merge(const int32_t *recv, int64_t R, const int64_t *src_off, int32_t world, int32_t c, int32_t reduction,
                               const int32_t *shape, int64_t *first_o, int64_t *coords, int32_t *npoints, float *feats,
                               int32_t *rec_owned, int32_t *lead_rec, int64_t *counts, void *workspace, size_t workspace_bytes,
                               void *stream, uint32_t flags, const int64_t *point_off)
{
    hipStream_t st = (hipStream_t)stream;
    if (R < 0 || c < 1 || c > 16 || world < 1 || world > kMaxWorld || !counts || !src_off || !shape) return D3D_ERR_BAD_ARG;
    if (flags & ~(uint32_t)(D3D_OWNER_MERGE_CHAINS | D3D_OWNER_MERGE_TEST_TINY)) return D3D_ERR_BAD_ARG;
    if (reduction < D3D_REDUCE_MEAN || reduction > D3D_REDUCE_MIN) return D3D_ERR_UNSUPPORTED;
    if (R >= (1ll << 31)) return D3D_ERR_BAD_ARG;
    if (R > 0 && (!recv || !first_o || !coords || !npoints || !feats || !rec_owned || !lead_rec)) return D3D_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < d3d_owner_merge_workspace_bytes(R, world)) return D3D_ERR_WORKSPACE;
    MergeOut f{rec_stride(c), (int)c, (int)reduction, (int64_t)shape[1], (int64_t)shape[2], first_o, coords, npoints, feats, lead_rec};
    f.recv = recv;
    f.point_off = point_off;
    f.world = (int)world;
    if (!merge_on_chains(R, flags)) {
        RecWs m = carve_rec(workspace, workspace_bytes, R, world);
        f.src_off = m.src_off;
        if (R == 0) {
            D3D_HIP_CHECK(hipMemsetAsync(counts, 0, D3D_NUM_COUNTS * sizeof(int64_t), st));
            return D3D_OK;
        }
        const size_t lds = (size_t)kRecTile * 10 + (size_t)m.nb * 4;
        D3D_LAUNCH("k_rec_tile_sort", k_rec_tile_sort, dim3((m.ntiles + 7u) & ~7u), dim3(kRecTileThreads), lds, st, recv, R, rec_stride(c), m.nb, m.ntiles,
                   m.ekey, m.eidx, m.table, rec_owned, m.status, m.ftiles + 1, m.overflow, counts, src_off, (int)world, m.src_off);
        D3D_LAUNCH("k_rec_bucket", k_rec_bucket, dim3(m.nb), dim3(kRecBucketThreads), 0, st, (const u64 *)m.ekey, (const uint32_t *)m.eidx,
                   (const uint32_t *)m.table, m.ntiles, m.cellrecs, m.rinfo, m.overflow, counts,
                   (flags & D3D_OWNER_MERGE_TEST_TINY) ? 4u : (uint32_t)kRecBucketCap, c == 4 ? recv : (const int32_t *)nullptr,
                   (int)reduction, m.minfo, m.mrec);
        D3D_LAUNCH("k_rec_finish", k_rec_finish, dim3(m.ftiles), dim3(kFinishThreads), 0, st, recv, R, (const uint32_t *)m.rinfo,
                   (const uint32_t *)m.cellrecs, (const uint32_t *)m.minfo, (const int32_t *)m.mrec, m.status, m.ftiles,
                   (const uint32_t *)m.overflow, f, rec_owned, counts);
        return D3D_OK;
    }
    MergeWs m = carve_merge(workspace, workspace_bytes, R, world);
    f.src_off = m.src_off;
    D3D_LAUNCH("k_merge_init", k_merge_init, dim3(blocks_for((int64_t)m.cap, 256 * 4)), dim3(256), 0, st, m.slot, m.cap, m.status,
               m.ntiles + 1, counts, src_off, (int)world, m.src_off);
    if (R > 0) {
        D3D_LAUNCH("k_merge_insert", k_merge_insert, dim3(blocks_for(R)), dim3(256), 0, st, recv, R, rec_stride(c), m.slot, m.cap, m.next,
                   m.rec_slot, rec_owned);
        D3D_LAUNCH("k_merge_finish", k_merge_finish, dim3(m.ntiles), dim3(kFinishThreads), 0, st, recv, R, (const MergeSlot *)m.slot,
                   (const uint32_t *)m.next, (const uint32_t *)m.rec_slot, m.status, m.ntiles, f, rec_owned, counts);
    }
    return D3D_OK;
}

// bit f of bitmap[(n_total + 63) / 64] set for every owned voxel's first point f (the words are cleared here); one word MORE
// behind them carries this owner's merge status (1 = a bucket overflowed): the SUM all-reduce of the bitmaps then tells every
// rank whether ANY owner has to repeat (d3d_owner_number raises BIN_OVERFLOW in counts_out[STATUS] on all of them alike)
extern "C" int d3d_owner_mark_first(const int64_t *first_o, const int64_t *counts_o, int64_t cap_o, int64_t n_total,
                                    uint64_t *bitmap, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (cap_o < 0 || n_total < 0 || !bitmap || !counts_o) return D3D_ERR_BAD_ARG;
    const int64_t nw = d3d_divup(n_total > 0 ? n_total : 1, 64);
    D3D_HIP_CHECK(hipMemsetAsync(bitmap, 0, (size_t)(nw + 1) * 8, st));
    D3D_LAUNCH("k_first_mark", k_first_mark, dim3(blocks_for(cap_o > 0 ? cap_o : 1)), dim3(256), 0, st, first_o, counts_o, n_total,
               (u64 *)bitmap, nw);
    return D3D_OK;
}

extern "C" size_t d3d_owner_number_workspace_bytes(int64_t n_total)
{
    const int64_t nw = d3d_divup(n_total > 0 ? n_total : 1, 64);
    return d3d_align_up((size_t)nw * 4) + d3d_align_up((size_t)(d3d_divup(nw, kPrefixTile) + 1) * 8) + 256;
}

// global_bits = the SUM all-reduce of every owner's d3d_owner_mark_first bitmap (disjoint bit sets: their OR; + the status word).  vids[i] =
// global voxel id (first-seen order over the whole frame, voxelize.cpp:119) of owned voxel i; counts_out[D3D_COUNT_VOXELS] =
// voxels of the whole frame.
extern "C" int d3d_owner_number(const uint64_t *global_bits, int64_t n_total, const int64_t *first_o, const int64_t *counts_o,
                                int64_t cap_o, int64_t *vids, int64_t *counts_out, void *workspace, size_t workspace_bytes,
                                void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n_total < 0 || cap_o < 0 || !counts_o || !counts_out || !global_bits) return D3D_ERR_BAD_ARG;
    if (n_total >= (1ll << 32)) return D3D_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < d3d_owner_number_workspace_bytes(n_total)) return D3D_ERR_WORKSPACE;
    const int64_t nw = d3d_divup(n_total > 0 ? n_total : 1, 64);
    WsCarver w(workspace, workspace_bytes);
    uint32_t *pre = w.take<uint32_t>(nw);
    const uint32_t ntiles = (uint32_t)d3d_divup(nw, kPrefixTile);
    u64 *status = w.take<u64>((size_t)ntiles + 1);
    D3D_HIP_CHECK(hipMemsetAsync(status, 0, ((size_t)ntiles + 1) * 8, st));
    D3D_LAUNCH("k_first_prefix", k_first_prefix, dim3(ntiles), dim3(kPrefixThreads), 0, st, (const u64 *)global_bits, nw, pre, status, ntiles,
               counts_out);
    if (cap_o > 0)
        D3D_LAUNCH("k_owner_number", k_owner_number, dim3(blocks_for(cap_o)), dim3(256), 0, st, counts_o, (const u64 *)global_bits, pre,
                   first_o, vids);
    return D3D_OK;
}

// dense contract on the owner, after d3d_owner_merge (whose workspace, untouched since, holds the cells' record lists; same flags):
// recv_rows[*, 4] = the candidate rows received (grouped by source rank; rows_src_off[world + 1], device), a record's last word =
// offset of its rows inside its batch.  -> voxels[cap_o, max_points, 4] and pmask[cap_o, max_points] of the owned voxels in id order.
// row_state (NULL: every row of voxels[0 .. Vo) is written): the resident form of d3d_voxelize_3d_dense_resident -- voxels and
// row_state[capacity] kept by the caller from frame to frame, zero-filled once; only rows with points and stale rows are stored.
extern "C" int d3d_owner_dense(const int32_t *recv, int64_t R, const float *recv_rows, const int64_t *rows_src_off, int32_t world,
                               int32_t max_points, const int32_t *lead_rec, const int32_t *npoints, const int64_t *counts_o,
                               int64_t cap_o, const void *merge_workspace, size_t merge_workspace_bytes, float *voxels,
                               uint8_t *pmask, void *stream, uint32_t flags, uint16_t *row_state)
{
    hipStream_t st = (hipStream_t)stream;
    if (R < 0 || cap_o < 0 || world < 1 || world > kMaxWorld || max_points < 1 || max_points > kDenseCap || !counts_o) return D3D_ERR_BAD_ARG;
    if (cap_o == 0) return D3D_OK;
    if (!recv || !recv_rows || !rows_src_off || !lead_rec || !npoints || !voxels || !pmask) return D3D_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(recv_rows) | reinterpret_cast<uintptr_t>(voxels)) & 15) return D3D_ERR_BAD_ARG;
    if (!merge_workspace || merge_workspace_bytes < d3d_owner_merge_workspace_bytes(R, world)) return D3D_ERR_WORKSPACE;
    const uint32_t P = (uint32_t)max_points;
    const int pshift = (P & (P - 1)) == 0 ? __builtin_ctz(P) : -1;
    if (!merge_on_chains(R, flags)) {
        RecWs m = carve_rec(const_cast<void *>(merge_workspace), merge_workspace_bytes, R, world);
        if (row_state)
            D3D_LAUNCH("k_owner_dense_resident", (k_owner_dense<true, true>), dim3(blocks_for(cap_o, 256)), dim3(256), 0, st, counts_o, lead_rec,
                       npoints, (const uint32_t *)nullptr, (const MergeSlot *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)m.rinfo,
                       (const uint32_t *)m.cellrecs, (const int64_t *)m.src_off, (int)world, recv, rec_stride(4),
                       reinterpret_cast<const float4 *>(recv_rows), rows_src_off, P, pshift, reinterpret_cast<float4 *>(voxels), pmask,
                       row_state);
        else
            D3D_LAUNCH("k_owner_dense", k_owner_dense<true>, dim3(blocks_for(cap_o, 256)), dim3(256), 0, st, counts_o, lead_rec, npoints,
                       (const uint32_t *)nullptr, (const MergeSlot *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)m.rinfo,
                       (const uint32_t *)m.cellrecs, (const int64_t *)m.src_off, (int)world, recv, rec_stride(4),
                       reinterpret_cast<const float4 *>(recv_rows), rows_src_off, P, pshift, reinterpret_cast<float4 *>(voxels), pmask,
                       (uint16_t *)nullptr);
        return D3D_OK;
    }
    MergeWs m = carve_merge(const_cast<void *>(merge_workspace), merge_workspace_bytes, R, world);
    if (row_state)
        D3D_LAUNCH("k_owner_dense_resident", (k_owner_dense<false, true>), dim3(blocks_for(cap_o, 256)), dim3(256), 0, st, counts_o, lead_rec,
                   npoints, (const uint32_t *)m.rec_slot, (const MergeSlot *)m.slot, (const uint32_t *)m.next, (const uint32_t *)nullptr,
                   (const uint32_t *)nullptr, (const int64_t *)m.src_off, (int)world, recv, rec_stride(4),
                   reinterpret_cast<const float4 *>(recv_rows), rows_src_off, P, pshift, reinterpret_cast<float4 *>(voxels), pmask, row_state);
    else
        D3D_LAUNCH("k_owner_dense", k_owner_dense<false>, dim3(blocks_for(cap_o, 256)), dim3(256), 0, st, counts_o, lead_rec, npoints,
                   (const uint32_t *)m.rec_slot, (const MergeSlot *)m.slot, (const uint32_t *)m.next, (const uint32_t *)nullptr,
                   (const uint32_t *)nullptr, (const int64_t *)m.src_off, (int)world, recv, rec_stride(4),
                   reinterpret_cast<const float4 *>(recv_rows), rows_src_off, P, pshift, reinterpret_cast<float4 *>(voxels), pmask,
                   (uint16_t *)nullptr);
    return D3D_OK;
}

// reply[i] = global voxel id of received record i (sent back to the record's source rank)
extern "C" int d3d_owner_reply(int64_t R, const int32_t *rec_owned, const int64_t *vids, int64_t *reply, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (R < 0) return D3D_ERR_BAD_ARG;
    if (R > 0) D3D_LAUNCH("k_owner_reply", k_owner_reply, dim3(blocks_for(R)), dim3(256), 0, st, R, rec_owned, vids, reply);
    return D3D_OK;
}

// back[] = the ids returned for this rank's records, in send order; pos_of_local as left by d3d_owner_pack; local_map[n] =
// point -> local voxel (d3d_voxelize_3d_reduce).  -> gmap[n] global voxel id per point.
extern "C" int d3d_owner_map(int64_t n, const int64_t *local_map, const int32_t *pos_of_local, const int64_t *back,
                             int64_t *gmap, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0) return D3D_ERR_BAD_ARG;
    if (n > 0) D3D_LAUNCH("k_owner_map", k_owner_map, dim3(blocks_for(n)), dim3(256), 0, st, n, local_map, pos_of_local, back, gmap);
    return D3D_OK;
}

// all owners' finished rows (concatenated in any order) -> the replicated feature grid in voxel-id order
extern "C" int d3d_owner_replicate(int64_t V, const int64_t *vids, const int64_t *coords_in, const int32_t *cnt_in,
                                   const float *feats_in, int32_t c, int64_t *coords, int32_t *cnt, float *feats, void *stream,
                                   const int64_t *src_off, int32_t world, void *workspace, size_t workspace_bytes)
{
    hipStream_t st = (hipStream_t)stream;
    if (V < 0 || c < 1 || (src_off && (world < 1 || world > kMaxWorld))) return D3D_ERR_BAD_ARG;
    if (V == 0) return D3D_OK;
    const int64_t ntiles = d3d_divup(V, kRepTile);
    if (src_off && c <= kRepMaxC && workspace && workspace_bytes >= (size_t)(ntiles + 1) * world * 8) {
        // the ranks' blocks in rank order, each ascending in id (what the all-gather delivers)
        int64_t *bounds = static_cast<int64_t *>(workspace);
        D3D_LAUNCH("k_owner_replicate_bounds", k_owner_replicate_bounds, dim3(blocks_for((ntiles + 1) * world)), dim3(256), 0, st, V, vids,
                   src_off, (int)world, ntiles, bounds);
        if (c <= 4)
            D3D_LAUNCH("k_owner_replicate_merge", k_owner_replicate_merge<4>, dim3((unsigned)ntiles), dim3(256), 0, st, V, vids, coords_in,
                       cnt_in, feats_in, (int)c, coords, cnt, feats, (const int64_t *)bounds, (int)world);
        else
            D3D_LAUNCH("k_owner_replicate_merge", k_owner_replicate_merge<kRepMaxC>, dim3((unsigned)ntiles), dim3(256), 0, st, V, vids,
                       coords_in, cnt_in, feats_in, (int)c, coords, cnt, feats, (const int64_t *)bounds, (int)world);
    } else
        D3D_LAUNCH("k_owner_replicate", k_owner_replicate, dim3(blocks_for(V)), dim3(256), 0, st, V, vids, coords_in, cnt_in, feats_in,
                   (int)c, coords, cnt, feats);
    return D3D_OK;
}
