// match.hip -- the association step of the detection evaluator on MI355X (gfx950).
// Replaces ScoreMatcher.match + BaseMatcher.match_by_order (reference d3d/tracking/matcher.pyx:83-162) as
// DetectionEvaluator.calc_stats drives them (d3d/benchmarks.pyx:218-238): boxes to match ("src", detections) are taken from
// the best score down; each takes the nearest still-unassigned fixed box ("dst", ground truth) of its own category whose
// distance is within that category's threshold.
//
// The reference re-sorts the whole n x m distance matrix for every score threshold (40 argsorts of 20 k x 5 k floats at
// config 4).  Two observations make one pass enough:
//   * a detection's choice depends only on the detections before it in score order, and every threshold's subset is a
//     PREFIX of that order -- so the matching of the full set, cut at a score, IS the matching of that threshold;
//   * only the pairs within the distance threshold can ever match -- a handful per row.
// k_match_candidates lists those per row (one wavefront per row: coalesced sweep, ballot compaction, 64-key bitonic sort by
// (distance, index) in registers); k_match_greedy walks the rows in score order in ONE wavefront: the 64 lanes test a row's
// candidates against the assignment bitmap at once and the first free one (ballot + ffs) is taken.
#include "common.hpp"
#include <math.h>

namespace {

constexpr int kMaxCand = 64;            // candidates listed per row (one per lane): its 64 NEAREST; a row with more is flagged
                                        // and, should all 64 be taken when its turn comes, sweeps its whole row (k_match_greedy)

__device__ __forceinline__ bool pair_less(float d1, int j1, float d2, int j2) { return d1 < d2 || (d1 == d2 && j1 < j2); }

// bitonic sort of 64 (distance, index) pairs across the lanes, ascending; empty lanes hold (inf, max)
__device__ __forceinline__ void wave_sort_pairs(float &bd, int &bj, int lane)
{
#pragma unroll
    for (int k = 2; k <= kWave; k <<= 1) {
#pragma unroll
        for (int s2 = k >> 1; s2 > 0; s2 >>= 1) {
            const float od = __shfl_xor(bd, s2, kWave);
            const int oj = __shfl_xor(bj, s2, kWave);
            const bool up = (lane & k) == 0, lower = (lane & s2) == 0;
            const bool other_less = pair_less(od, oj, bd, bj);
            // ascending block: the lower lane keeps the smaller; descending block: the larger
            const bool take = (up == lower) ? other_less : (!other_less && (od != bd || oj != bj));
            if (take) { bd = od; bj = oj; }
        }
    }
}

// (bd, bj): 64 pairs sorted ascending; (cd, cj): 64 more, sorted ascending -> the 64 smallest of the 128, sorted ascending.
// min(a[l], b[63 - l]) over the lanes is a bitonic sequence of exactly those; six compare-exchange steps sort it.
__device__ __forceinline__ void wave_merge_lowest(float &bd, int &bj, float cd, int cj, int lane)
{
    const float rd = __shfl(cd, kWave - 1 - lane, kWave);
    const int rj = __shfl(cj, kWave - 1 - lane, kWave);
    if (pair_less(rd, rj, bd, bj)) { bd = rd; bj = rj; }
#pragma unroll
    for (int s2 = kWave >> 1; s2 > 0; s2 >>= 1) {
        const float od = __shfl_xor(bd, s2, kWave);
        const int oj = __shfl_xor(bj, s2, kWave);
        const bool lower = (lane & s2) == 0;
        const bool other_less = pair_less(od, oj, bd, bj);
        const bool take = lower ? other_less : (!other_less && (od != bd || oj != bj));
        if (take) { bd = od; bj = oj; }
    }
}

// dist[n,m]; src_tag[n], dst_tag[m] categories (negative: not taking part); thr[m] = distance threshold of dst j's category
__global__ __launch_bounds__(256) void k_match_candidates(const float *__restrict__ dist, int64_t n, int64_t m,
                                                          const int32_t *__restrict__ src_tag, const int32_t *__restrict__ dst_tag,
                                                          const float *__restrict__ thr, int32_t *cand_dst, float *cand_dist,
                                                          int32_t *cand_cnt, int32_t *overflow,
                                                          const int64_t *__restrict__ row_src = nullptr /* row r reads dist row row_src[r] */,
                                                          const uint8_t *__restrict__ mask = nullptr /* [., m]: 0 = the pair takes no part */,
                                                          const int64_t *__restrict__ row_mask = nullptr /* row r reads mask row row_mask[r] */)
{
    __shared__ float sd[256 / kWave][kMaxCand];
    __shared__ int sj[256 / kWave][kMaxCand];
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * (256 / kWave) + w;
    if (row >= n) return;                       // (no workgroup barrier below: wavefronts are independent)
    const int32_t tag = src_tag[row];
    sd[w][lane] = INFINITY;
    sj[w][lane] = 0x7fffffff;
    int found = 0;
    bool streaming = false;                     // more than 64 candidates so far: (bd, bj) hold the 64 nearest, sorted
    float bd = INFINITY;
    int bj = 0x7fffffff;
    if (tag >= 0) {
        const float *drow = dist + (row_src ? row_src[row] : row) * m;
        const uint8_t *mrow = mask ? mask + (row_mask ? row_mask[row] : row) * m : nullptr;
        // four stretches of 64 columns in flight per step (the sweep is a chain of round trips otherwise: 78 for 5 k columns)
        for (int64_t jq = 0; jq < m; jq += 4 * kWave) {
          float dq[4];
          bool okq[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
              const int64_t j = jq + u * kWave + lane, jc = j < m ? j : m - 1;      // (no branch: the four loads leave together)
              dq[u] = drow[jc];
              okq[u] = (j < m) & (dst_tag[jc] == tag) & (dq[u] <= thr[jc]) & (!mrow || mrow[jc] != 0);
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int64_t j = jq + u * kWave + lane;
            const float d = dq[u];
            const bool ok = okq[u];
            const unsigned long long mask = __ballot(ok);
            if (!mask) continue;
            const int more = __popcll(mask);
            if (!streaming && found + more <= kMaxCand) {
                const int slot = found + __popcll(mask & ((1ull << lane) - 1));
                if (ok) { sd[w][slot] = d; sj[w][slot] = (int)j; }
                found += more;
                continue;
            }
            if (!streaming) {                   // the list outgrows the 64 slots: from here on, merge and keep the nearest
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                bd = sd[w][lane];
                bj = sj[w][lane];
                wave_sort_pairs(bd, bj, lane);
                streaming = true;
            }
            float cd = ok ? d : INFINITY;
            int cj = ok ? (int)j : 0x7fffffff;
            wave_sort_pairs(cd, cj, lane);
            wave_merge_lowest(bd, bj, cd, cj, lane);
            found += more;
          }
        }
    }
    if (found > kMaxCand && lane == 0) atomicOr(overflow, 1);
    if (!streaming) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();        // LDS ops of one wavefront complete in order
        bd = sd[w][lane];
        bj = sj[w][lane];
        wave_sort_pairs(bd, bj, lane);
    }
    cand_dst[row * kMaxCand + lane] = bj;
    cand_dist[row * kMaxCand + lane] = bd;
    if (lane == 0) cand_cnt[row] = found <= kMaxCand ? found : -kMaxCand;      // negative: more candidates exist than are listed
}

// order[n]: src rows from the best score down; src_match[n], dst_match[m] <- partner or -1.  ONE wavefront: 64 rows'
// candidate lists are staged in LDS at a time (64 independent loads in flight), then the rows are decided one after the other
// from LDS: the 64 lanes test a row's candidates against the assignment bitmap at once, the first free one (ballot + ffs)
// is taken.  The bitmap lives in LDS (dynamic, m <= 256 k) -- else in global memory, read and written with agent-scope atomics.
template <bool LDS_MAP>
__global__ __launch_bounds__(64) void k_match_greedy(const int64_t *__restrict__ order, int64_t n, int64_t m,
                                                     const int32_t *__restrict__ cand_dst, const int32_t *__restrict__ cand_cnt,
                                                     int32_t *src_match, int32_t *dst_match, unsigned int *taken_g /* zeroed */,
                                                     const float *__restrict__ dist, const int32_t *__restrict__ src_tag,
                                                     const int32_t *__restrict__ dst_tag, const float *__restrict__ thr,
                                                     const int32_t *only_if = nullptr, const int64_t *__restrict__ row_off = nullptr,
                                                     const int64_t *__restrict__ row_src = nullptr, const uint8_t *__restrict__ okmask = nullptr,
                                                     const int64_t *__restrict__ row_mask = nullptr)
{
    extern __shared__ unsigned int taken_l[];
    __shared__ int stage[kWave][kMaxCand];
    const int64_t nwords = (m + 31) / 32;
    if (row_off) {                             // batched (d3d_score_match_batched): workgroup b = problem b, rows [row_off[b], row_off[b + 1])
        const int64_t b = blockIdx.x, o = row_off[b];
        n = row_off[b + 1] - o;
        order += o; cand_dst += o * kMaxCand; cand_cnt += o; src_match += o; dst_match += b * m; taken_g += b * nwords;
        if (row_src) row_src += o; else dist += o * m;
        if (row_mask) row_mask += o; else if (okmask) okmask += o * m;
        src_tag += o;
        if (only_if) only_if += b;
    }
    if (only_if && !*only_if) return;          // k_match_stable decided everything (the usual case)
    const int lane = threadIdx.x;
    if (LDS_MAP)
        for (int64_t t = lane; t < nwords; t += kWave) taken_l[t] = 0;
    for (int64_t j = lane; j < m; j += kWave) dst_match[j] = -1;
    for (int64_t p0 = 0; p0 < n; p0 += kWave) {
        const int rows = (int)(n - p0 < kWave ? n - p0 : kWave);
        const int64_t myrow = lane < rows ? order[p0 + lane] : 0;
        const int mycnt = lane < rows ? cand_cnt[myrow] : 0;
        for (int r = 0; r < rows; r++) {
            const int64_t row = __shfl(myrow, r, kWave);
            stage[r][lane] = cand_dst[row * kMaxCand + lane];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int mypick = -1;
        for (int r = 0; r < rows; r++) {
            const int cntr = __shfl(mycnt, r, kWave);
            const int64_t rowr = __shfl(myrow, r, kWave);
            if (cntr == 0) continue;
            const int cnt = cntr < 0 ? -cntr : cntr;      // negative: the row has more candidates than the 64 nearest listed
            int dst = lane < cnt ? stage[r][lane] : -1;
            bool free_ = false;
            if (dst >= 0) {
                const unsigned int word = LDS_MAP ? taken_l[dst >> 5]
                                                  : __hip_atomic_load(&taken_g[dst >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                free_ = ((word >> (dst & 31)) & 1u) == 0;
            }
            unsigned long long mask = __ballot(free_);
            if (!mask && cntr < 0) {
                // all 64 nearest are taken and there are more: the nearest free candidate of the WHOLE row (the reference
                // considers every pair within the threshold, matcher.pyx:100-117), 64 columns per step
                const int32_t tag = src_tag[rowr];
                const float *drow = dist + (row_src ? row_src[rowr] : rowr) * m;
                const uint8_t *mrow = okmask ? okmask + (row_mask ? row_mask[rowr] : rowr) * m : nullptr;
                float bd = INFINITY;
                int bj = 0x7fffffff;
                for (int64_t j0 = 0; j0 < m; j0 += kWave) {
                    const int64_t j = j0 + lane;
                    if (j >= m) continue;
                    const float d = drow[j];
                    if (dst_tag[j] != tag || !(d <= thr[j]) || (mrow && mrow[j] == 0)) continue;
                    const unsigned int word = LDS_MAP ? taken_l[j >> 5]
                                                      : __hip_atomic_load(&taken_g[j >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((word >> (j & 31)) & 1u) continue;
                    if (pair_less(d, (int)j, bd, bj)) { bd = d; bj = (int)j; }
                }
#pragma unroll
                for (int o = kWave / 2; o > 0; o >>= 1) {
                    const float od = __shfl_xor(bd, o, kWave);
                    const int oj = __shfl_xor(bj, o, kWave);
                    if (pair_less(od, oj, bd, bj)) { bd = od; bj = oj; }
                }
                if (bj != 0x7fffffff) {                     // every lane holds the winner: lane 0 acts for it
                    dst = lane == 0 ? bj : -1;
                    mask = 1ull;
                }
            }
            if (mask) {
                const int l = __ffsll((long long)mask) - 1;                      // nearest free one
                const int pick = __shfl(dst, l, kWave);
                if (lane == l) {
                    if (LDS_MAP) taken_l[dst >> 5] |= 1u << (dst & 31);
                    else atomicOr(&taken_g[dst >> 5], 1u << (dst & 31));
                    dst_match[dst] = (int32_t)rowr;
                }
                if (lane == r) mypick = pick;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (lane < rows) src_match[myrow] = mypick;
    }
}

// The same matching without the walk (round 5).  "Sources in score order, each takes its nearest free destination" is the serial
// dictatorship of a market in which every destination prefers the better score -- ONE common ranking -- and that outcome is
// the market's unique stable matching, which deferred acceptance reaches in ANY order of proposals: a free source proposes
// to the next destination of its list (atomicMin of its rank on the destination's holder); the value that comes back tells it
// at once whether it now holds the destination (and whom it displaced: that source goes to the next round's queue and goes
// on from its next candidate) or was refused (next candidate, same round).  20 k sources against 5 k destinations: a few dozen
// rounds of a shrinking queue in one workgroup instead of 20 000 dependent steps of one wavefront (k_match_greedy: 8.7 of
// calc_stats' 13.5 ms).  A source whose list was cut at 64 and runs out of it raises `need_walk`, and so does a queue that is
// not empty after kStableRounds rounds: k_match_greedy then redoes the matching (it sweeps such a row); it exits at once otherwise.
constexpr int kStableThreads = 1024;
constexpr int kStableRounds = 512;      // a displacement chain longer than this (detections in a row, each preferring its left
                                        // neighbour's ground truth): one displacement per round -- the walk is the faster routine then
__global__ __launch_bounds__(kStableThreads) void k_match_stable(const int64_t *__restrict__ order, int64_t n, int64_t m,
                                                                 const int32_t *__restrict__ cand_dst, const int32_t *__restrict__ cand_cnt,
                                                                 int32_t *src_match, int32_t *dst_match, int32_t *rank, int32_t *ptr,
                                                                 int32_t *hold, int32_t *q0, int32_t *q1, int32_t *need_walk,
                                                                 const int64_t *__restrict__ row_off = nullptr)
{
    __shared__ unsigned int qn[2];
    if (row_off) {                             // batched: workgroup b = problem b (indices below are local to its rows)
        const int64_t b = blockIdx.x, o = row_off[b];
        n = row_off[b + 1] - o;
        order += o; cand_dst += o * kMaxCand; cand_cnt += o; src_match += o; dst_match += b * m;
        rank += o; ptr += o; q0 += o; q1 += o; hold += b * m; need_walk += b;
    }
    const int tid = threadIdx.x;
    constexpr int kFree = 0x7fffffff;
    for (int64_t p = tid; p < n; p += kStableThreads) {
        const int64_t sidx = order[p];
        rank[sidx] = (int32_t)p;
        src_match[sidx] = -1;
    }
    for (int64_t d = tid; d < m; d += kStableThreads) hold[d] = kFree;
    if (tid == 0) { qn[0] = 0; qn[1] = 0; *need_walk = 0; }
    __threadfence();
    __syncthreads();
    int cur = 0, rounds = 0;
    unsigned int count = (unsigned int)n;
    bool first = true;                              // round 0: every source, from the top of its list
    for (;;) {
        int32_t *qin = cur ? q1 : q0, *qout = cur ? q0 : q1;
        for (unsigned int k = tid; k < count; k += kStableThreads) {
            const int32_t sidx = first ? (int32_t)k : __hip_atomic_load(&qin[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int32_t c = cand_cnt[sidx], cabs = c < 0 ? -c : c;
            const int32_t r = __hip_atomic_load(&rank[sidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int32_t p = first ? 0 : __hip_atomic_load(&ptr[sidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
            const int32_t *list = cand_dst + (int64_t)sidx * kMaxCand;
            while (p < cabs) {
                const int32_t d = list[p];
                __hip_atomic_store(&ptr[sidx], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (before the proposal: whoever displaces this source
                const int32_t old = atomicMin(&hold[d], r);                                       //  sends it on from here)
                if (old > r) {
                    if (old != kFree) {
                        const unsigned int at = atomicAdd(&qn[cur ^ 1], 1u);
                        __hip_atomic_store(&qout[at], (int32_t)order[old], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    break;
                }
                p++;
            }
            if (p >= cabs && c < 0) *need_walk = 1;  // the 64 nearest were not enough: the walk sweeps the row
        }
        __threadfence();
        __syncthreads();
        count = qn[cur ^ 1];
        __syncthreads();
        if (tid == 0) qn[cur] = 0;
        cur ^= 1;
        first = false;
        if (count == 0) break;
        if (++rounds > kStableRounds) {             // (wave-uniform: every thread counts the same rounds)
            if (tid == 0) *need_walk = 1;
            break;
        }
        __syncthreads();
    }
    for (int64_t d = tid; d < m; d += kStableThreads) {
        const int32_t h = __hip_atomic_load(&hold[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int32_t sidx = h == kFree ? -1 : (int32_t)order[h];
        dst_match[d] = sidx;
        if (sidx >= 0) src_match[sidx] = (int32_t)d;
    }
}

}  // namespace

// B problems at once (the evaluator's score thresholds on a frame: 40 small associations, each a chain of short launches when
// issued one by one): the rows of all problems stacked in dist[N, m] (problem b = rows [row_off[b], row_off[b + 1])), the
// destinations -- dst_tag, dst_threshold, m -- common to all; order / src_match in the same stacked layout with indices LOCAL
// to the problem, dst_match[B, m] likewise local.  One candidate launch over all rows, one workgroup per problem after that.
// Optional indirection (the evaluator's literal association: row k of a problem takes its DISTANCES from one row of the cache and
// its ACCEPTABLE pairs from another, matcher.pyx:155-158): stacked row r reads dist row row_src[r] (NULL: r) and, with `mask`
// (u8 [., m], 0 = the pair takes no part), mask row row_mask[r] (NULL: r) -- no stacked matrix is materialised then.
extern "C" size_t d3d_score_match_batched_workspace_bytes(int64_t n_total, int64_t m, int64_t batches)
{
    if (n_total < 1) n_total = 1;
    if (m < 1) m = 1;
    if (batches < 1) batches = 1;
    return d3d_align_up((size_t)n_total * kMaxCand * 4) * 2 + d3d_align_up((size_t)n_total * 4) * 5 +
           d3d_align_up((size_t)batches * (((size_t)m + 31) / 32) * 4) + d3d_align_up((size_t)batches * (size_t)m * 4) +
           d3d_align_up((size_t)batches * 4) + 1024;
}

extern "C" int d3d_score_match_batched(const float *dist, const int64_t *row_src, const uint8_t *mask, const int64_t *row_mask,
                                       const int64_t *row_off, int64_t batches, int64_t n_total, int64_t m,
                                       const int32_t *src_tag, const int32_t *dst_tag, const float *dst_threshold,
                                       const int64_t *order, int32_t *src_match, int32_t *dst_match, int32_t *status, void *workspace,
                                       size_t workspace_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (batches < 0 || n_total < 0 || m < 0 || !status) return D3D_ERR_BAD_ARG;
    D3D_HIP_CHECK(hipMemsetAsync(status, 0, 4, st));
    if (batches == 0) return D3D_OK;
    if (!row_off || (m > 0 && !dst_match)) return D3D_ERR_BAD_ARG;
    if (m > 0) D3D_HIP_CHECK(hipMemsetAsync(dst_match, 0xff, (size_t)batches * (size_t)m * 4, st));
    if (n_total == 0) return D3D_OK;
    if (!src_match || !src_tag || !order) return D3D_ERR_BAD_ARG;
    if (m == 0) { D3D_HIP_CHECK(hipMemsetAsync(src_match, 0xff, (size_t)n_total * 4, st)); return D3D_OK; }
    if (!dist || !dst_tag || !dst_threshold || m >= (1ll << 31) || n_total >= (1ll << 31) || batches > 65535) return D3D_ERR_BAD_ARG;
    const size_t words = ((size_t)m + 31) / 32;
    WsCarver w(workspace, workspace_bytes);
    int32_t *cand_dst = w.take<int32_t>((size_t)n_total * kMaxCand);
    float *cand_dist = w.take<float>((size_t)n_total * kMaxCand);
    int32_t *cand_cnt = w.take<int32_t>((size_t)n_total);
    unsigned int *taken = w.take<unsigned int>((size_t)batches * words);
    int32_t *rank = w.take<int32_t>((size_t)n_total), *ptr = w.take<int32_t>((size_t)n_total), *q0 = w.take<int32_t>((size_t)n_total),
            *q1 = w.take<int32_t>((size_t)n_total);
    int32_t *hold = w.take<int32_t>((size_t)batches * (size_t)m), *need_walk = w.take<int32_t>((size_t)batches);
    if (!workspace || !w.ok()) return D3D_ERR_WORKSPACE;
    D3D_HIP_CHECK(hipMemsetAsync(taken, 0, (size_t)batches * words * 4, st));
    D3D_LAUNCH("k_match_candidates", k_match_candidates, dim3((unsigned)d3d_divup(n_total, 256 / kWave)), dim3(256), 0, st, dist, n_total, m,
               src_tag, dst_tag, dst_threshold, cand_dst, cand_dist, cand_cnt, status, row_src, mask, row_mask);
    D3D_LAUNCH("k_match_stable", k_match_stable, dim3((unsigned)batches), dim3(kStableThreads), 0, st, order, (int64_t)0, m,
               (const int32_t *)cand_dst, (const int32_t *)cand_cnt, src_match, dst_match, rank, ptr, hold, q0, q1, need_walk, row_off);
    const size_t map_bytes = words * 4;
    if (map_bytes <= 32 * 1024)
        D3D_LAUNCH("k_match_greedy", k_match_greedy<true>, dim3((unsigned)batches), dim3(64), map_bytes, st, order, (int64_t)0, m,
                   (const int32_t *)cand_dst, (const int32_t *)cand_cnt, src_match, dst_match, taken, dist, src_tag, dst_tag, dst_threshold,
                   (const int32_t *)need_walk, row_off, row_src, mask, row_mask);
    else
        D3D_LAUNCH("k_match_greedy", k_match_greedy<false>, dim3((unsigned)batches), dim3(64), 0, st, order, (int64_t)0, m,
                   (const int32_t *)cand_dst, (const int32_t *)cand_cnt, src_match, dst_match, taken, dist, src_tag, dst_tag, dst_threshold,
                   (const int32_t *)need_walk, row_off, row_src, mask, row_mask);
    return D3D_OK;
}

extern "C" size_t d3d_score_match_workspace_bytes(int64_t n, int64_t m)
{
    if (n < 1) n = 1;
    if (m < 1) m = 1;
    return d3d_align_up((size_t)n * kMaxCand * 4) * 2 + d3d_align_up((size_t)n * 4) + d3d_align_up(((size_t)m + 31) / 32 * 4) + 512 +
           d3d_align_up((size_t)n * 4) * 4 + d3d_align_up((size_t)m * 4) + 256;          // k_match_stable: rank, ptr, two queues, holders
}

// status word (device, int32): bit 0 = some row had more than 64 candidates within its threshold (informational: such a row
// lists its 64 nearest and sweeps its whole row if all of them are taken -- the result is the reference's in every case)
extern "C" int d3d_score_match(const float *dist, int64_t n, int64_t m, const int32_t *src_tag, const int32_t *dst_tag,
                               const float *dst_threshold, const int64_t *order, int32_t *src_match, int32_t *dst_match,
                               int32_t *status, void *workspace, size_t workspace_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || m < 0 || !status) return D3D_ERR_BAD_ARG;
    D3D_HIP_CHECK(hipMemsetAsync(status, 0, 4, st));
    if (m > 0 && !dst_match) return D3D_ERR_BAD_ARG;
    if (n == 0) { if (m > 0) D3D_HIP_CHECK(hipMemsetAsync(dst_match, 0xff, (size_t)m * 4, st)); return D3D_OK; }
    if (!src_match || !src_tag || !order) return D3D_ERR_BAD_ARG;
    if (m == 0) { D3D_HIP_CHECK(hipMemsetAsync(src_match, 0xff, (size_t)n * 4, st)); return D3D_OK; }
    if (!dist || !dst_tag || !dst_threshold || m >= (1ll << 31) || n >= (1ll << 31)) return D3D_ERR_BAD_ARG;
    WsCarver w(workspace, workspace_bytes);
    int32_t *cand_dst = w.take<int32_t>((size_t)n * kMaxCand);
    float *cand_dist = w.take<float>((size_t)n * kMaxCand);
    int32_t *cand_cnt = w.take<int32_t>((size_t)n);
    unsigned int *taken = w.take<unsigned int>(((size_t)m + 31) / 32);
    int32_t *rank = w.take<int32_t>((size_t)n), *ptr = w.take<int32_t>((size_t)n), *q0 = w.take<int32_t>((size_t)n), *q1 = w.take<int32_t>((size_t)n);
    int32_t *hold = w.take<int32_t>((size_t)m), *need_walk = w.take<int32_t>(1);
    if (!workspace || !w.ok()) return D3D_ERR_WORKSPACE;
    D3D_HIP_CHECK(hipMemsetAsync(taken, 0, ((size_t)m + 31) / 32 * 4, st));
    D3D_LAUNCH("k_match_candidates", k_match_candidates, dim3((unsigned)d3d_divup(n, 256 / kWave)), dim3(256), 0, st, dist, n, m,
               src_tag, dst_tag, dst_threshold, cand_dst, cand_dist, cand_cnt, status);
    D3D_LAUNCH("k_match_stable", k_match_stable, dim3(1), dim3(kStableThreads), 0, st, order, n, m, (const int32_t *)cand_dst,
               (const int32_t *)cand_cnt, src_match, dst_match, rank, ptr, hold, q0, q1, need_walk);
    const size_t map_bytes = ((size_t)m + 31) / 32 * 4;
    if (map_bytes <= 32 * 1024)          // (+ 16 KB of staging: inside the 64 KB a workgroup gets without opting in)
        D3D_LAUNCH("k_match_greedy", k_match_greedy<true>, dim3(1), dim3(64), map_bytes, st, order, n, m, (const int32_t *)cand_dst,
                   (const int32_t *)cand_cnt, src_match, dst_match, taken, dist, src_tag, dst_tag, dst_threshold, (const int32_t *)need_walk);
    else
        D3D_LAUNCH("k_match_greedy", k_match_greedy<false>, dim3(1), dim3(64), 0, st, order, n, m, (const int32_t *)cand_dst,
                   (const int32_t *)cand_cnt, src_match, dst_match, taken, dist, src_tag, dst_tag, dst_threshold, (const int32_t *)need_walk);
    return D3D_OK;
}
