// sort.hip -- stable descending argsort (the role torch::argsort plays inside the reference: nms.cpp:103,
// voxelize.cpp:406): one workgroup in LDS up to 2 k keys, a four-launch sample sort up to 128 k, an LSD radix sort above;
// voxel counts (the fused DESCENDING filter) have a counting sort of their own (k_cs_*).
// No library sort (round 4: rocprim::radix_sort_pairs is gone).
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include "common.hpp"
#include "lds_sort.hpp"

namespace {
// ---------------------------------------------------------------- radix path (any n; what runs above 128 k keys)
// Own LSD radix sort (round 4; rocprim::radix_sort_pairs before): 8-bit digits, stable, the (key, index) pair travels.
//   k_rs_hist     one sweep over the keys: the digit histograms of ALL passes (64 workgroups, LDS histograms, one global
//                 atomic per non-empty bin)
//   k_rs_plan     one workgroup: a pass whose digit is the same in every key is SKIPPED (promoted fp32 scores have three
//                 zero mantissa bytes, voxel counts two zero high bytes): which buffer every remaining pass reads and
//                 writes, which pass is the last (it writes `order`), the bins' bases
//   per pass      k_rs_tile_hist: digit counts of every 4096-key tile (tile-major rows of 1 KiB);  k_rs_scatter: ranks by
//                 wavefront ballots (eight per row of 64 keys: the lanes with my digit), the tile's offsets = the column sums
//                 of the rows before it (read by the workgroup itself: no scan launch, nobody waits for anybody), the tile
//                 reordered in LDS so that the global stores run along the digits' runs.  Both exit at once on a skipped pass.
// A one-launch pass with look-back was measured in round 2 (tools/radix_bench.hip): with every tile resident at once nobody
// has an inclusive prefix to offer and each tile walks all its predecessors -- 990 us for 1 M pairs.
constexpr int kRsThreads = 256, kRsItems = 16, kRsTile = kRsThreads * kRsItems, kRsBins = 256, kRsWaves = kRsThreads / kWave;
constexpr int kRsHistBlocks = 256, kRsHistThreads = 1024;    // (64 blocks: a quarter of the CUs, 49 us for 4 M keys)

struct RsPlan {             // per pass
    int32_t skip, src, dst;  // src: -1 = the caller's keys (index = position), 0 / 1 = ping-pong buffer; dst: 0 / 1, 2 = `order`
};

template <typename K>
__global__ __launch_bounds__(kRsHistThreads) void k_rs_hist(const K *__restrict__ keys, uint32_t n, uint32_t *ghist,
                                                            const int64_t *__restrict__ n_dev)
{
    if (n_dev && (uint64_t)*n_dev < n) n = (uint32_t)*n_dev;     // (the caller only knows an upper bound on the host)
    typedef typename KeyBits<K>::U U;
    constexpr int PASSES = sizeof(U);
    __shared__ uint32_t h[PASSES][kRsBins];
    for (int i = threadIdx.x; i < PASSES * kRsBins; i += kRsHistThreads) (&h[0][0])[i] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * kRsHistThreads + threadIdx.x; i < n; i += gridDim.x * kRsHistThreads) {
        const U k = KeyBits<K>::desc(keys[i]);
#pragma unroll
        for (int p = 0; p < PASSES; p++) {
            // a digit that is the same in the whole wavefront (the zero mantissa bytes of promoted fp32 scores, a sign / exponent
            // byte) is counted by one lane: 64 atomics on one LDS word serialise (k_rs_hist 71 us on such keys)
            const uint32_t d = (uint32_t)(k >> (8 * p)) & 255u;
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
            const unsigned long long act = __ballot(true);
            if (__ballot(d == d0) == act) {
                if ((threadIdx.x & (kWave - 1)) == (uint32_t)__builtin_ctzll(act)) atomicAdd(&h[p][d0], (uint32_t)__popcll(act));
            } else atomicAdd(&h[p][d], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PASSES * kRsBins; i += kRsHistThreads)
        if ((&h[0][0])[i]) atomicAdd(&ghist[i], (&h[0][0])[i]);
}

template <int PASSES>
__global__ __launch_bounds__(kRsBins) void k_rs_plan(uint32_t *ghist /* -> exclusive bases */, uint32_t n, RsPlan *plan,
                                                      const int64_t *__restrict__ n_dev, int first_pass)
{
    if (n_dev && (uint64_t)*n_dev < n) n = (uint32_t)*n_dev;
    __shared__ unsigned long long smem[kRsBins / kWave];
    __shared__ int skip[PASSES];
    for (int p = 0; p < PASSES; p++) {
        const uint32_t c = ghist[p * kRsBins + threadIdx.x];
        const int full = __syncthreads_or(c == n);                    // one bin holds every key: nothing to do in this pass
        unsigned long long tot;
        const unsigned long long ex = block_excl_scan_u64<kRsBins>(c, &tot, smem);
        ghist[p * kRsBins + threadIdx.x] = (uint32_t)ex;
        if (threadIdx.x == 0) skip[p] = full;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int p = 0; p < first_pass; p++) skip[p] = 1;          // (never launched: see radix_argsort_desc)
        int last = -1;
        for (int p = 0; p < PASSES; p++)
            if (!skip[p]) last = p;
        int cur = -1;                                                  // where the pairs are: -1 = the caller's keys
        for (int p = 0; p < PASSES; p++) {
            plan[p].skip = skip[p];
            plan[p].src = cur;
            plan[p].dst = p == last ? 2 : (cur == 0 ? 1 : 0);
            if (!skip[p]) cur = plan[p].dst;
        }
        plan[PASSES].skip = last < 0;                                  // every key equal: the order is the identity
    }
}

// the pass's key of item i and its index, from wherever the plan says the pairs are
template <typename K>
__device__ __forceinline__ typename KeyBits<K>::U rs_load(const K *__restrict__ keys, const typename KeyBits<K>::U *__restrict__ kin,
                                                          const uint32_t *__restrict__ vin, int src, uint32_t i, uint32_t &val)
{
    if (src < 0) { val = i; return KeyBits<K>::desc(keys[i]); }
    val = vin[i];
    return kin[i];
}

template <typename K>
__global__ __launch_bounds__(kRsThreads) void k_rs_tile_hist(const K *__restrict__ keys, typename KeyBits<K>::U *kb0,
                                                             typename KeyBits<K>::U *kb1, uint32_t n, int pass,
                                                             const RsPlan *__restrict__ plan, uint32_t *__restrict__ tilehist,
                                                             const int64_t *__restrict__ n_dev)
{
    typedef typename KeyBits<K>::U U;
    const RsPlan pl = plan[pass];
    if (pl.skip) return;
    if (n_dev && (uint64_t)*n_dev < n) n = (uint32_t)*n_dev;
    if ((uint64_t)blockIdx.x * kRsTile >= n) {                      // a tile beyond the device-side size: an empty row
        tilehist[(size_t)blockIdx.x * kRsBins + threadIdx.x] = 0;
        return;
    }
    __shared__ uint32_t h[kRsBins];
    h[threadIdx.x] = 0;
    __syncthreads();
    const U *kin = pl.src == 0 ? kb0 : kb1;
    const uint32_t base = blockIdx.x * kRsTile + threadIdx.x;
#pragma unroll
    for (int r = 0; r < kRsItems; r++) {
        const uint32_t i = base + r * kRsThreads;
        if (i < n) {
            const U k = pl.src < 0 ? KeyBits<K>::desc(keys[i]) : kin[i];
            atomicAdd(&h[(uint32_t)(k >> (8 * pass)) & 255u], 1u);
        }
    }
    __syncthreads();
    tilehist[(size_t)blockIdx.x * kRsBins + threadIdx.x] = h[threadIdx.x];
}

template <typename K, typename V>
__global__ __launch_bounds__(kRsThreads) void k_rs_scatter(const K *__restrict__ keys, typename KeyBits<K>::U *kb0,
                                                           typename KeyBits<K>::U *kb1, uint32_t *vb0, uint32_t *vb1, uint32_t n,
                                                           int pass, const RsPlan *__restrict__ plan,
                                                           const uint32_t *__restrict__ tilehist, const uint32_t *__restrict__ gbase,
                                                           V *__restrict__ order, const int64_t *__restrict__ n_dev, int first_pass)
{
    typedef typename KeyBits<K>::U U;
    if (n_dev && (uint64_t)*n_dev < n) n = (uint32_t)*n_dev;
    if ((uint64_t)blockIdx.x * kRsTile >= n) return;
    const RsPlan pl = plan[pass];
    if (pl.skip) {
        // (every key equal: pass 0 writes the identity)
        if (pass == first_pass && plan[sizeof(U)].skip)
            for (uint32_t i = blockIdx.x * kRsTile + threadIdx.x; i < n && i < (blockIdx.x + 1) * (uint32_t)kRsTile; i += kRsThreads)
                order[i] = (V)i;
        return;
    }
    __shared__ uint32_t cnt[kRsWaves][kRsBins];     // per-wavefront digit counts -> the wavefront's offset inside the digit's run
    __shared__ uint32_t dstart[kRsBins];            // the digit's run inside the tile (exclusive scan of the tile's counts)
    __shared__ uint32_t gpos[kRsBins];              // where that run goes: bin base + the same digit in all earlier tiles
    __shared__ U skey[kRsTile];
    __shared__ uint32_t sval[kRsTile];
    __shared__ unsigned long long smem[kRsThreads / kWave];
    for (int i = threadIdx.x; i < kRsWaves * kRsBins; i += kRsThreads) (&cnt[0][0])[i] = 0;
    // column sums of the earlier tiles' rows (thread = digit; coalesced rows, independent loads)
    uint32_t before = 0;
#pragma unroll 8
    for (uint32_t t = 0; t < blockIdx.x; t++) before += tilehist[(size_t)t * kRsBins + threadIdx.x];
    __syncthreads();
    const U *kin = pl.src == 0 ? kb0 : kb1;
    const uint32_t *vin = pl.src == 0 ? vb0 : vb1;
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * kRsTile + w * (kWave * kRsItems) + lane;     // wavefront-striped: row r at base + 64 r
    U key[kRsItems];
    uint32_t val[kRsItems], rank[kRsItems];
#pragma unroll
    for (int r = 0; r < kRsItems; r++) {
        const uint32_t i = base + r * kWave;
        key[r] = ~(U)0;
        val[r] = 0;
        if (i < n) key[r] = rs_load<K>(keys, kin, vin, pl.src, i, val[r]);
    }
    // rank of every pair among the pairs of its wavefront with the same digit, in input order
#pragma unroll
    for (int r = 0; r < kRsItems; r++) {
        const uint32_t i = base + r * kWave;
        const bool valid = i < n;
        const uint32_t d = (uint32_t)(key[r] >> (8 * pass)) & 255u;
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        const int leader = __builtin_ctzll(same | (valid ? 0ull : 1ull << lane));
        uint32_t old = 0;
        if (valid && lane == leader) { old = cnt[w][d]; cnt[w][d] = old + (uint32_t)__popcll(same); }
        __builtin_amdgcn_wave_barrier();
        old = __shfl(old, leader, kWave);
        rank[r] = old + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    {   // thread = digit: the wavefronts' offsets inside the digit's run, the runs inside the tile, their place in the output
        uint32_t run = 0;
#pragma unroll
        for (int ww = 0; ww < kRsWaves; ww++) { const uint32_t c = cnt[ww][threadIdx.x]; cnt[ww][threadIdx.x] = run; run += c; }
        unsigned long long tot;
        const uint32_t ex = (uint32_t)block_excl_scan_u64<kRsThreads>(run, &tot, smem);
        dstart[threadIdx.x] = ex;
        gpos[threadIdx.x] = gbase[pass * kRsBins + threadIdx.x] + before - ex;        // + position inside the tile = output place
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRsItems; r++) {
        const uint32_t i = base + r * kWave;
        if (i < n) {
            const uint32_t d = (uint32_t)(key[r] >> (8 * pass)) & 255u;
            const uint32_t p = dstart[d] + cnt[w][d] + rank[r];
            skey[p] = key[r];
            sval[p] = val[r];
        }
    }
    __syncthreads();
    const uint32_t m = n - blockIdx.x * kRsTile < (uint32_t)kRsTile ? n - blockIdx.x * kRsTile : (uint32_t)kRsTile;
    U *kout = pl.dst == 0 ? kb0 : kb1;
    uint32_t *vout = pl.dst == 0 ? vb0 : vb1;
    for (uint32_t p = threadIdx.x; p < m; p += kRsThreads) {            // consecutive lanes: consecutive places of a digit's run
        const U k = skey[p];
        const uint32_t pos = gpos[(uint32_t)(k >> (8 * pass)) & 255u] + p;
        if (pl.dst == 2) order[pos] = (V)sval[p];
        else { kout[pos] = k; vout[pos] = sval[p]; }
    }
}

template <typename K>
size_t radix_bytes(int64_t n)
{
    typedef typename KeyBits<K>::U U;
    const size_t ntiles = (size_t)d3d_divup(n, kRsTile);
    return 2 * d3d_align_up(sizeof(U) * (size_t)n) + 2 * d3d_align_up(4 * (size_t)n) + d3d_align_up(sizeof(U) * kRsBins * 4) +
           d3d_align_up(ntiles * kRsBins * 4) + d3d_align_up((sizeof(U) + 1) * sizeof(RsPlan)) + 256;
}

// order[n] <- stable descending argsort of keys.  n_dev (optional): the number of keys on the device, <= n (the host only has
// the bound).  key_bits (optional): the keys' desc() images agree above that many low bits (the caller knows the range of
// its keys): the higher passes are not even launched -- (skipped passes cost a launch each otherwise).
template <typename K, typename V>
int radix_argsort_desc(const K *keys, int64_t n, V *order, void *ws, size_t ws_bytes, hipStream_t st, const int64_t *n_dev = nullptr,
                       int key_bits = 0)
{
    typedef typename KeyBits<K>::U U;
    constexpr int PASSES = sizeof(U);
    const unsigned ntiles = (unsigned)d3d_divup(n, kRsTile);
    WsCarver w(ws, ws_bytes);
    U *kb0 = w.take<U>(n), *kb1 = w.take<U>(n);
    uint32_t *vb0 = w.take<uint32_t>(n), *vb1 = w.take<uint32_t>(n);
    uint32_t *ghist = w.take<uint32_t>((size_t)PASSES * kRsBins);
    uint32_t *tilehist = w.take<uint32_t>((size_t)ntiles * kRsBins);
    RsPlan *plan = w.take<RsPlan>(PASSES + 1);
    if (!ws || !w.ok()) return D3D_ERR_WORKSPACE;
    const int npass = key_bits > 0 && key_bits < PASSES * 8 ? (key_bits + 7) / 8 : PASSES;
    D3D_HIP_CHECK(hipMemsetAsync(ghist, 0, (size_t)PASSES * kRsBins * 4, st));
    D3D_LAUNCH("k_rs_hist", k_rs_hist<K>, dim3(kRsHistBlocks), dim3(kRsHistThreads), 0, st, keys, (uint32_t)n, ghist, n_dev);
    D3D_LAUNCH("k_rs_plan", k_rs_plan<PASSES>, dim3(1), dim3(kRsBins), 0, st, ghist, (uint32_t)n, plan, n_dev, 0);
    for (int p = 0; p < npass; p++) {
        D3D_LAUNCH("k_rs_tile_hist", k_rs_tile_hist<K>, dim3(ntiles), dim3(kRsThreads), 0, st, keys, kb0, kb1, (uint32_t)n, p,
                   (const RsPlan *)plan, tilehist, n_dev);
        D3D_LAUNCH("k_rs_scatter", (k_rs_scatter<K, V>), dim3(ntiles), dim3(kRsThreads), 0, st, keys, kb0, kb1, vb0, vb1, (uint32_t)n, p,
                   (const RsPlan *)plan, (const uint32_t *)tilehist, (const uint32_t *)ghist, order, n_dev, 0);
    }
    return D3D_OK;
}

// ---------------------------------------------------------------- voxel counts (the fused DESCENDING filter, voxelize.cpp:406)
// Keys are the point counts of voxels: nearly all below 255, a handful above (the cells next to the sensor).  The general path
// spends a histogram sweep, a plan, two live 8-bit passes and a skipped one on them (82 us for 585 k voxels).  Here ONE pass on
// the digit 255 - min(count, 255): descending, stable, written straight into `order` -- except the class "255 or more", which
// comes first and leaves in index order into a side list; those few (at most points / 255) are then ranked among themselves
// by (count descending, position ascending), all pairs, one thread each (quadratic in a number that the points bound: 4 M points
// cannot crowd more than 16 k voxels, 2.7 x 10^8 comparisons in that adversarial frame).  Three launches, no plan: every workgroup scans the
// tiles' digit rows itself (bases = exclusive scan of the column totals + the column sums of the tiles before it).
__device__ __forceinline__ uint32_t cs_digit(int32_t c) { return 255u - ((uint32_t)c < 255u ? (uint32_t)c : 255u); }      // counts >= 0

__global__ __launch_bounds__(kRsThreads) void k_cs_tile_hist(const int32_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ tilehist,
                                                             const int64_t *__restrict__ n_dev)
{
    if (n_dev && (uint64_t)*n_dev < n) n = (uint32_t)*n_dev;
    if ((uint64_t)blockIdx.x * kRsTile >= n) return;                 // (rows beyond the device-side size are never read)
    __shared__ uint32_t h[kRsBins];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kRsTile + threadIdx.x;
#pragma unroll
    for (int r = 0; r < kRsItems; r++) {
        const uint32_t i = base + r * kRsThreads;
        if (i < n) atomicAdd(&h[cs_digit(keys[i])], 1u);
    }
    __syncthreads();
    tilehist[(size_t)blockIdx.x * kRsBins + threadIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(kRsThreads) void k_cs_scatter(const int32_t *__restrict__ keys, uint32_t n, const uint32_t *__restrict__ tilehist,
                                                           int32_t *__restrict__ order, uint32_t *__restrict__ bigidx, uint32_t bigcap,
                                                           uint32_t *__restrict__ kbig, const int64_t *__restrict__ n_dev)
{
    if (n_dev && (uint64_t)*n_dev < n) n = (uint32_t)*n_dev;
    if (blockIdx.x == 0 && n == 0 && threadIdx.x == 0) *kbig = 0;
    if ((uint64_t)blockIdx.x * kRsTile >= n) return;
    __shared__ uint32_t cnt[kRsWaves][kRsBins];
    __shared__ uint32_t dstart[kRsBins];
    __shared__ uint32_t gpos[kRsBins];
    __shared__ uint32_t sdig[kRsTile];               // (one byte would do; a word keeps the LDS reads conflict-free)
    __shared__ uint32_t sval[kRsTile];
    __shared__ unsigned long long smem[kRsThreads / kWave];
    for (int i = threadIdx.x; i < kRsWaves * kRsBins; i += kRsThreads) (&cnt[0][0])[i] = 0;
    const uint32_t ntiles = (n + kRsTile - 1) / kRsTile;
    uint32_t before = 0, total = 0;                   // thread = digit: its count in the tiles before mine, and in all
#pragma unroll 8
    for (uint32_t t = 0; t < blockIdx.x; t++) before += tilehist[(size_t)t * kRsBins + threadIdx.x];
    total = before;
#pragma unroll 8
    for (uint32_t t = blockIdx.x; t < ntiles; t++) total += tilehist[(size_t)t * kRsBins + threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * kRsTile + w * (kWave * kRsItems) + lane;
    uint32_t dig[kRsItems], rank[kRsItems];
#pragma unroll
    for (int r = 0; r < kRsItems; r++) {
        const uint32_t i = base + r * kWave;
        dig[r] = i < n ? cs_digit(keys[i]) : 255u;
    }
#pragma unroll
    for (int r = 0; r < kRsItems; r++) {
        const uint32_t i = base + r * kWave;
        const bool valid = i < n;
        const uint32_t d = dig[r];
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        const int leader = __builtin_ctzll(same | (valid ? 0ull : 1ull << lane));
        uint32_t old = 0;
        if (valid && lane == leader) { old = cnt[w][d]; cnt[w][d] = old + (uint32_t)__popcll(same); }
        __builtin_amdgcn_wave_barrier();
        old = __shfl(old, leader, kWave);
        rank[r] = old + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    {
        uint32_t run = 0;
#pragma unroll
        for (int ww = 0; ww < kRsWaves; ww++) { const uint32_t c = cnt[ww][threadIdx.x]; cnt[ww][threadIdx.x] = run; run += c; }
        unsigned long long tot;
        const uint32_t ex = (uint32_t)block_excl_scan_u64<kRsThreads>(run, &tot, smem);
        dstart[threadIdx.x] = ex;
        __syncthreads();                               // (smem is reused by the second scan)
        const uint32_t gbase = (uint32_t)block_excl_scan_u64<kRsThreads>(total, &tot, smem);
        gpos[threadIdx.x] = gbase + before - ex;
        if (blockIdx.x == 0 && threadIdx.x == 0) *kbig = total < bigcap ? total : bigcap;      // digit 0: the class "255 or more"
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRsItems; r++) {
        const uint32_t i = base + r * kWave;
        if (i < n) {
            const uint32_t d = dig[r];
            const uint32_t p = dstart[d] + cnt[w][d] + rank[r];
            sdig[p] = d;
            sval[p] = i;
        }
    }
    __syncthreads();
    const uint32_t m = n - blockIdx.x * kRsTile < (uint32_t)kRsTile ? n - blockIdx.x * kRsTile : (uint32_t)kRsTile;
    for (uint32_t p = threadIdx.x; p < m; p += kRsThreads) {
        const uint32_t d = sdig[p];
        const uint32_t pos = gpos[d] + p;
        if (d == 0) { if (pos < bigcap) bigidx[pos] = sval[p]; }
        else order[pos] = (int32_t)sval[p];
    }
}

// the class "255 or more", in index order in bigidx[0 .. *kbig): rank by (count descending, position ascending)
__global__ __launch_bounds__(256) void k_cs_rank_big(const int32_t *__restrict__ keys, const uint32_t *__restrict__ bigidx,
                                                     const uint32_t *__restrict__ kbig, int32_t *__restrict__ order)
{
    const uint32_t kb = *kbig;
    if (blockIdx.x * 256u >= kb) return;
    __shared__ int32_t sk[256];
    const uint32_t me = blockIdx.x * 256u + threadIdx.x;
    const uint32_t myidx = me < kb ? bigidx[me] : 0u;
    const int32_t mykey = me < kb ? keys[myidx] : 0;
    uint32_t rank = 0;
    for (uint32_t c0 = 0; c0 < kb; c0 += 256) {
        const uint32_t j = c0 + threadIdx.x;
        __syncthreads();
        sk[threadIdx.x] = j < kb ? keys[bigidx[j]] : -1;            // (-1: behind every count)
        __syncthreads();
        const uint32_t lim = kb - c0 < 256u ? kb - c0 : 256u;
        for (uint32_t t = 0; t < lim; t++) {
            const int32_t k = sk[t];
            rank += (k > mykey || (k == mykey && c0 + t < me)) ? 1u : 0u;
        }
    }
    if (me < kb) order[rank] = (int32_t)myidx;
}

size_t counts_argsort_bytes(int64_t n)
{
    return d3d_align_up((size_t)d3d_divup(n, kRsTile) * kRsBins * 4) + d3d_align_up(((size_t)n / 255 + 1) * 4) + 256;
}

int counts_argsort_desc(const int32_t *keys, int64_t n, const int64_t *n_dev, int64_t max_key_sum, int32_t *order, void *ws, size_t ws_bytes,
                        hipStream_t st)
{
    const unsigned ntiles = (unsigned)d3d_divup(n, kRsTile);
    const uint32_t bigcap = (uint32_t)(max_key_sum / 255 + 1);
    WsCarver w(ws, ws_bytes);
    uint32_t *tilehist = w.take<uint32_t>((size_t)ntiles * kRsBins);
    uint32_t *bigidx = w.take<uint32_t>(bigcap);
    uint32_t *kbig = w.take<uint32_t>(1);
    if (!ws || !w.ok()) return D3D_ERR_WORKSPACE;
    D3D_LAUNCH("k_cs_tile_hist", k_cs_tile_hist, dim3(ntiles), dim3(kRsThreads), 0, st, keys, (uint32_t)n, tilehist, n_dev);
    D3D_LAUNCH("k_cs_scatter", k_cs_scatter, dim3(ntiles), dim3(kRsThreads), 0, st, keys, (uint32_t)n, (const uint32_t *)tilehist, order, bigidx,
               bigcap, kbig, n_dev);
    D3D_LAUNCH("k_cs_rank_big", k_cs_rank_big, dim3((unsigned)d3d_divup((int64_t)bigcap, 256)), dim3(256), 0, st, keys, (const uint32_t *)bigidx,
               (const uint32_t *)kbig, order);
    return D3D_OK;
}

// ---------------------------------------------------------------- bucket path (2 k .. 128 k keys: the NMS sizes)
// The library sorts this range with a block sort + one merge launch per doubling (9 launches, 70 us at 100 k keys: launch
// latency, not bandwidth).  Here: a sample sort in 4 launches --
//   k_ss_splitters  one workgroup sorts 1024 stratified samples in LDS and keeps every (1024 / B)-th as a splitter
//   k_ss_count      1024-key tiles: bucket of every key (binary search over the B - 1 splitters in LDS), LDS histogram whose
//                   atomicAdd return value is the key's arrival number in (tile, bucket); one global atomicAdd per (tile,
//                   bucket) reserves the tile's range inside the bucket (its return value = the tile's offset)
//   k_ss_scatter    bucket bases = scan of the B totals (repeated per workgroup); (key, index) -> its bucket
//   k_ss_bucket     one workgroup per bucket: sort in LDS, order[base + r] = index
// Everything compares the composite (key, index): it is unique, so the splitters cut runs of equal keys (all-equal input
// gives B equal buckets) and the result depends neither on the arrival order of the atomics nor on the sample.  Bucket
// sizes only depend on the sample: B = n / 192 buckets (<= 512, two samples per bucket), so a bucket's size is the mean
// (<= 256) times Gamma(2) / 2 -- the largest of 512 is ~5 x the mean; the workgroup that gets it sets the kernel's duration,
// which is why the buckets are small (256 buckets of mean 390: the largest, 1193, took 15 us).  One that outgrows the LDS
// (2048 entries, >= 8 x the mean: p ~ 1e-3 per sort at 128 k keys, 1e-7 at 100 k) is ranked from global memory by counting
// -- slow, still exact.
// The LDS sort (samples, buckets): every wavefront sorts runs of 64 in registers (bitonic network over __shfl_xor, no
// barrier), then log2(n / 64) merge passes in which every element finds its place in the merged run by a binary search
// of the sibling run (unique composites: position = own position + number of smaller siblings) -- ~5 barriers instead of
// the 55-66 of a workgroup-wide bitonic network, which took 22 us per bucket.
constexpr int kSsSamples = 1024, kSsTile = 1024, kSsCountThreads = 512, kSsMaxBuckets = 512, kSsBucketBits = 9;
constexpr int kSsSortThreads = 1024, kSsBucketCap = 2048;
constexpr int64_t kSsMinN = 2049, kSsMaxN = (int64_t)kSsMaxBuckets * 256;      // (below: k_sort_small, one launch)

__device__ __forceinline__ uint32_t ss_hash(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <typename K>
__global__ __launch_bounds__(kSsSortThreads) void k_ss_splitters(const K *__restrict__ keys, uint32_t n, int B,
                                                                 typename KeyBits<K>::U *spl_d, uint32_t *spl_i, uint32_t *bucket_cnt)
{
    typedef typename KeyBits<K>::U U;
    __shared__ U d0[kSsSamples], d1[kSsSamples];
    __shared__ uint32_t i0[kSsSamples], i1[kSsSamples];
    const uint32_t len = n / kSsSamples;                     // n >= kSsMinN: >= 2
    for (int j = threadIdx.x; j < kSsSamples; j += blockDim.x) {
        const uint32_t pos = (uint32_t)j * len + ss_hash((uint32_t)j) % len;     // one sample per stratum, jittered
        d0[j] = KeyBits<K>::desc(keys[pos]);
        i0[j] = pos;
    }
    for (int b = threadIdx.x; b <= kSsMaxBuckets; b += blockDim.x) bucket_cnt[b] = 0;
    __syncthreads();
    U *d;
    uint32_t *ii;
    sort_lds<1>(d0, i0, d1, i1, kSsSamples, &d, &ii);
    for (int b = threadIdx.x + 1; b < B; b += blockDim.x) {
        const int j = (int)((long long)b * kSsSamples / B);
        spl_d[b - 1] = d[j];
        spl_i[b - 1] = ii[j];
    }
}

template <typename K>
__global__ __launch_bounds__(kSsCountThreads) void k_ss_count(const K *__restrict__ keys, uint32_t n, int B,
                                                              const typename KeyBits<K>::U *__restrict__ spl_d,
                                                              const uint32_t *__restrict__ spl_i, uint32_t *__restrict__ pb,
                                                              uint32_t *__restrict__ tileoff, uint32_t *bucket_cnt)
{
    typedef typename KeyBits<K>::U U;
    __shared__ U sd[kSsMaxBuckets];
    __shared__ uint32_t si[kSsMaxBuckets], hist[kSsMaxBuckets];
    if ((int)threadIdx.x < B - 1) { sd[threadIdx.x] = spl_d[threadIdx.x]; si[threadIdx.x] = spl_i[threadIdx.x]; }
    hist[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSsTile / kSsCountThreads; k++) {
        const uint32_t i = blockIdx.x * kSsTile + k * kSsCountThreads + threadIdx.x;
        if (i < n) {
            const U dk = KeyBits<K>::desc(keys[i]);
            int lo = 0, hi = B - 1;                          // bucket = number of splitters <= (dk, i)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (comp_less(dk, i, sd[mid], si[mid])) hi = mid; else lo = mid + 1;
            }
            pb[i] = (uint32_t)lo | (atomicAdd(&hist[lo], 1u) << kSsBucketBits);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < B) {
        const uint32_t c = hist[threadIdx.x];
        tileoff[(size_t)blockIdx.x * kSsMaxBuckets + threadIdx.x] = c ? atomicAdd(&bucket_cnt[threadIdx.x], c) : 0u;
    }
}

template <typename K>
__global__ __launch_bounds__(kSsCountThreads) void k_ss_scatter(const K *__restrict__ keys, uint32_t n, int B,
                                                                const uint32_t *__restrict__ pb, const uint32_t *__restrict__ tileoff,
                                                                const uint32_t *__restrict__ bucket_cnt, uint32_t *bucket_base,
                                                                typename KeyBits<K>::U *__restrict__ dk, uint32_t *__restrict__ di)
{
    __shared__ uint32_t off[kSsMaxBuckets];
    __shared__ unsigned long long smem[kSsCountThreads / kWave];
    unsigned long long total;
    const uint32_t cnt = (int)threadIdx.x < B ? bucket_cnt[threadIdx.x] : 0u;
    const uint32_t base = (uint32_t)block_excl_scan_u64<kSsCountThreads>(cnt, &total, smem);
    off[threadIdx.x] = base + tileoff[(size_t)blockIdx.x * kSsMaxBuckets + threadIdx.x];
    if (blockIdx.x == 0) {                                   // for k_ss_bucket
        if ((int)threadIdx.x < B) bucket_base[threadIdx.x] = base;
        if (threadIdx.x == 0) bucket_base[B] = (uint32_t)total;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSsTile / kSsCountThreads; k++) {
        const uint32_t i = blockIdx.x * kSsTile + k * kSsCountThreads + threadIdx.x;
        if (i < n) {
            const uint32_t w = pb[i], pos = off[w & (kSsMaxBuckets - 1)] + (w >> kSsBucketBits);
            dk[pos] = KeyBits<K>::desc(keys[i]);
            di[pos] = i;
        }
    }
}

template <typename U, typename V>
__global__ __launch_bounds__(kSsSortThreads) void k_ss_bucket(const U *__restrict__ dk, const uint32_t *__restrict__ di,
                                                              const uint32_t *__restrict__ bucket_base, V *__restrict__ order)
{
    __shared__ U d0[kSsBucketCap], d1[kSsBucketCap];
    __shared__ uint32_t i0[kSsBucketCap], i1[kSsBucketCap];
    const uint32_t base = bucket_base[blockIdx.x], m = bucket_base[blockIdx.x + 1] - base;
    if (m == 0) return;
    if (m > (uint32_t)kSsBucketCap) {                        // rank by counting, from global memory
        for (uint32_t e = threadIdx.x; e < m; e += blockDim.x) {
            const U md = dk[base + e];
            const uint32_t mi = di[base + e];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < m; j++) rank += comp_less(dk[base + j], di[base + j], md, mi) ? 1u : 0u;
            order[base + rank] = (V)mi;
        }
        return;
    }
    int npad = kWave;
    while ((uint32_t)npad < m) npad <<= 1;
    for (int e = threadIdx.x; e < npad; e += blockDim.x) {
        d0[e] = (uint32_t)e < m ? dk[base + e] : ~(U)0;      // padding sorts behind every real entry (indices are < 2^31)
        i0[e] = (uint32_t)e < m ? di[base + e] : 0x80000000u + (uint32_t)e;      // ... and is unique too
    }
    __syncthreads();
    U *d;
    uint32_t *ii;
    if (npad <= kSsSortThreads) sort_lds<1>(d0, i0, d1, i1, npad, &d, &ii);
    else sort_lds<2>(d0, i0, d1, i1, npad, &d, &ii);
    for (uint32_t e = threadIdx.x; e < m; e += blockDim.x) order[base + e] = (V)ii[e];
}

// ---------------------------------------------------------------- up to kSsBucketCap keys: ONE launch, one workgroup, in LDS
// (the library: block sort + merge launches, 25-40 us of launch latency for a few hundred keys)
template <typename K, typename V>
__global__ __launch_bounds__(kSsSortThreads) void k_sort_small(const K *__restrict__ keys, uint32_t n, V *__restrict__ order)
{
    typedef typename KeyBits<K>::U U;
    __shared__ U d0[kSsBucketCap], d1[kSsBucketCap];
    __shared__ uint32_t i0[kSsBucketCap], i1[kSsBucketCap];
    int npad = kWave;
    while ((uint32_t)npad < n) npad <<= 1;
    for (int e = threadIdx.x; e < npad; e += blockDim.x) {
        d0[e] = (uint32_t)e < n ? KeyBits<K>::desc(keys[e]) : ~(U)0;     // padding sorts behind every real entry
        i0[e] = (uint32_t)e < n ? (uint32_t)e : 0x80000000u + (uint32_t)e;
    }
    __syncthreads();
    U *d;
    uint32_t *ii;
    if (npad <= kSsSortThreads) sort_lds<1>(d0, i0, d1, i1, npad, &d, &ii);
    else sort_lds<2>(d0, i0, d1, i1, npad, &d, &ii);
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) order[e] = (V)ii[e];
}

static inline int ss_buckets(int64_t n) { return (int)std::min<int64_t>(std::max<int64_t>(n / 192, 16), kSsMaxBuckets); }
static inline bool ss_eligible(int64_t n) { return n >= kSsMinN && n <= kSsMaxN; }

template <typename K>
size_t bucket_bytes(int64_t n)
{
    typedef typename KeyBits<K>::U U;
    const size_t ntiles = (size_t)d3d_divup(n, kSsTile);
    return d3d_align_up(kSsMaxBuckets * sizeof(U)) + d3d_align_up(kSsMaxBuckets * 4) + d3d_align_up((size_t)n * 4) +
           d3d_align_up(ntiles * kSsMaxBuckets * 4) + 2 * d3d_align_up((kSsMaxBuckets + 1) * 4) + d3d_align_up((size_t)n * sizeof(U)) +
           d3d_align_up((size_t)n * 4) + 256;
}

template <typename K, typename V>
int bucket_argsort_desc(const K *keys, int64_t n, V *order, void *ws, size_t ws_bytes, hipStream_t st)
{
    typedef typename KeyBits<K>::U U;
    const int B = ss_buckets(n);
    const unsigned ntiles = (unsigned)d3d_divup(n, kSsTile);
    WsCarver w(ws, ws_bytes);
    U *spl_d = w.take<U>(kSsMaxBuckets);
    uint32_t *spl_i = w.take<uint32_t>(kSsMaxBuckets);
    uint32_t *pb = w.take<uint32_t>(n);
    uint32_t *tileoff = w.take<uint32_t>((size_t)ntiles * kSsMaxBuckets);
    uint32_t *bucket_cnt = w.take<uint32_t>(kSsMaxBuckets + 1);
    uint32_t *bucket_base = w.take<uint32_t>(kSsMaxBuckets + 1);
    U *dk = w.take<U>(n);
    uint32_t *di = w.take<uint32_t>(n);
    if (!ws || !w.ok()) return D3D_ERR_WORKSPACE;
    D3D_LAUNCH("k_ss_splitters", k_ss_splitters<K>, dim3(1), dim3(kSsSortThreads), 0, st, keys, (uint32_t)n, B, spl_d, spl_i, bucket_cnt);
    D3D_LAUNCH("k_ss_count", k_ss_count<K>, dim3(ntiles), dim3(kSsCountThreads), 0, st, keys, (uint32_t)n, B, (const U *)spl_d,
               (const uint32_t *)spl_i, pb, tileoff, bucket_cnt);
    D3D_LAUNCH("k_ss_scatter", k_ss_scatter<K>, dim3(ntiles), dim3(kSsCountThreads), 0, st, keys, (uint32_t)n, B, (const uint32_t *)pb,
               (const uint32_t *)tileoff, (const uint32_t *)bucket_cnt, bucket_base, dk, di);
    D3D_LAUNCH("k_ss_bucket", (k_ss_bucket<U, V>), dim3((unsigned)B), dim3(kSsSortThreads), 0, st, (const U *)dk, (const uint32_t *)di,
               (const uint32_t *)bucket_base, order);
    return D3D_OK;
}

template <typename K, typename V>
size_t argsort_bytes(int64_t n)
{
    if (n < 1) n = 1;
    return std::max(radix_bytes<K>(n), ss_eligible(n) ? bucket_bytes<K>(n) : (size_t)0);
}

template <typename K, typename V>
int argsort_desc(const K *keys, int64_t n, V *order, void *ws, size_t ws_bytes, hipStream_t st, bool radix_only = false)
{
    if (n <= 0) return D3D_OK;
    if (n <= kSsBucketCap && !radix_only) {
        D3D_LAUNCH("k_sort_small", (k_sort_small<K, V>), dim3(1), dim3(kSsSortThreads), 0, st, keys, (uint32_t)n, order);
        return D3D_OK;
    }
    if (ss_eligible(n) && !radix_only) return bucket_argsort_desc<K, V>(keys, n, order, ws, ws_bytes, st);
    return radix_argsort_desc<K, V>(keys, n, order, ws, ws_bytes, st);
}
}  // namespace

extern "C" size_t d3d_internal_argsort_i32_bytes(int64_t n) { return argsort_bytes<int32_t, int32_t>(n); }
extern "C" int d3d_internal_argsort_desc_i32(const int32_t *keys, int64_t n, int32_t *order, void *ws, size_t ws_bytes,
                                             hipStream_t st)
{
    return argsort_desc<int32_t, int32_t>(keys, n, order, ws, ws_bytes, st);
}

// voxel counts on a device-side count: keys >= 0 whose SUM is at most max_key_sum (the points), n_dev <= n of them
extern "C" size_t d3d_internal_argsort_counts_bytes(int64_t n) { return counts_argsort_bytes(n); }
extern "C" int d3d_internal_argsort_desc_counts_dev(const int32_t *keys, int64_t n, const int64_t *n_dev, int64_t max_key_sum, int32_t *order,
                                                    void *ws, size_t ws_bytes, hipStream_t st)
{
    if (n <= 0) return D3D_OK;
    if (max_key_sum > n) max_key_sum = n;          // (the workspace was sized for n / 255 + 1 entries)
    return counts_argsort_desc(keys, n, n_dev, max_key_sum, order, ws, ws_bytes, st);
}

// (tests: the radix path at a size the bucket path would take)
extern "C" int d3d_internal_argsort_desc_radix(const void *keys, int64_t n, int32_t dtype, int64_t *order, void *ws, size_t ws_bytes,
                                                 void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    return dtype == D3D_F64 ? argsort_desc<double, int64_t>((const double *)keys, n, order, ws, ws_bytes, st, true)
                            : argsort_desc<float, int64_t>((const float *)keys, n, order, ws, ws_bytes, st, true);
}

extern "C" size_t d3d_argsort_desc_workspace_bytes(int64_t n, int32_t dtype)
{
    return dtype == D3D_F64 ? argsort_bytes<double, int64_t>(n) : argsort_bytes<float, int64_t>(n);
}

extern "C" int d3d_argsort_desc(const void *keys, int64_t n, int32_t dtype, int64_t *order, void *workspace,
                                size_t workspace_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || (dtype != D3D_F32 && dtype != D3D_F64)) return D3D_ERR_BAD_ARG;
    if (n == 0) return D3D_OK;
    if (!keys || !order) return D3D_ERR_BAD_ARG;
    if (n >= (1ll << 31)) return D3D_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < d3d_argsort_desc_workspace_bytes(n, dtype)) return D3D_ERR_WORKSPACE;
    return dtype == D3D_F64 ? argsort_desc<double, int64_t>((const double *)keys, n, order, workspace, workspace_bytes, st)
                            : argsort_desc<float, int64_t>((const float *)keys, n, order, workspace, workspace_bytes, st);
}
