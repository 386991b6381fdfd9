// sort.hip -- stable descending argsort (rocPRIM radix sort; a plain library sort, the same
// role torch::argsort plays inside the reference: nms.cpp:103, voxelize.cpp:406).
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include "common.hpp"
#include "lds_sort.hpp"
#include <rocprim/rocprim.hpp>

namespace {
// rocPRIM sorts inputs below 1 M items with a merge sort: block sort (1024 items per workgroup) + one launch per doubling,
// 7 at 100 k.  A larger block sort (merge_sort_config<512, 1024, 4>: 4096 items, 5 merge launches) was slower end to end
// (NMS of 100 k boxes 0.240 vs 0.231 ms), so the library default stays.
typedef rocprim::default_config SortConfig;

// ---------------------------------------------------------------- library path (rocPRIM; any n)
// V = type of the order entries (int32 inside the library, int64 at the C ABI: written directly, no widening pass)
template <typename K, typename V>
size_t sort_temp_bytes(int64_t n)
{
    typedef typename KeyBits<K>::U U;
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs<SortConfig, rocprim::transform_iterator<const K *, DescKey<K>, U>, U *, rocprim::counting_iterator<V>, V *>(
        nullptr, tmp, rocprim::transform_iterator<const K *, DescKey<K>, U>(nullptr, DescKey<K>()), nullptr,
        rocprim::counting_iterator<V>(0), nullptr, (size_t)n);
    return tmp;
}

template <typename K, typename V>
size_t library_bytes(int64_t n)
{
    return d3d_align_up(sizeof(typename KeyBits<K>::U) * n) + d3d_align_up(sort_temp_bytes<K, V>(n)) + 256;
}

// order[n] <- stable descending argsort of keys; the values 0 .. n-1 come from a counting iterator
template <typename K, typename V>
int library_argsort_desc(const K *keys, int64_t n, V *order, void *ws, size_t ws_bytes, hipStream_t st)
{
    typedef typename KeyBits<K>::U U;
    WsCarver w(ws, ws_bytes);
    U *keys_out = w.take<U>(n);
    size_t tmp = sort_temp_bytes<K, V>(n);
    char *temp = w.take<char>(tmp);
    if (!ws || !w.ok()) return D3D_ERR_WORKSPACE;
    D3D_HIP_CHECK((rocprim::radix_sort_pairs<SortConfig>(temp, tmp, rocprim::transform_iterator<const K *, DescKey<K>, U>(keys, DescKey<K>()),
                                                         keys_out, rocprim::counting_iterator<V>(0), order, (size_t)n, 0, sizeof(U) * 8, st)));
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
    return std::max(library_bytes<K, V>(n), ss_eligible(n) ? bucket_bytes<K>(n) : (size_t)0);
}

template <typename K, typename V>
int argsort_desc(const K *keys, int64_t n, V *order, void *ws, size_t ws_bytes, hipStream_t st, bool library_only = false)
{
    if (n <= 0) return D3D_OK;
    if (n <= kSsBucketCap && !library_only) {
        D3D_LAUNCH("k_sort_small", (k_sort_small<K, V>), dim3(1), dim3(kSsSortThreads), 0, st, keys, (uint32_t)n, order);
        return D3D_OK;
    }
    if (ss_eligible(n) && !library_only) return bucket_argsort_desc<K, V>(keys, n, order, ws, ws_bytes, st);
    return library_argsort_desc<K, V>(keys, n, order, ws, ws_bytes, st);
}
}  // namespace

extern "C" size_t d3d_internal_argsort_i32_bytes(int64_t n) { return argsort_bytes<int32_t, int32_t>(n); }
extern "C" int d3d_internal_argsort_desc_i32(const int32_t *keys, int64_t n, int32_t *order, void *ws, size_t ws_bytes,
                                             hipStream_t st)
{
    return argsort_desc<int32_t, int32_t>(keys, n, order, ws, ws_bytes, st);
}

// (tests: the library path at a size the bucket path would take)
extern "C" int d3d_internal_argsort_desc_library(const void *keys, int64_t n, int32_t dtype, int64_t *order, void *ws, size_t ws_bytes,
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
