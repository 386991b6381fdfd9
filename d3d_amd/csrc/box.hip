// box.hip -- pairwise IoU (axis-aligned "box", rotated "rbox", BEV x z "iou3d") and hard NMS
// for MI355X (gfx950).  Replaces reference d3d/box/iou.cpp + iou_cuda.cu (forward),
// d3d/box/nms.cpp + nms_cuda.cu, and the Cython pair loop over d3d/dgal_wrap.h.
//
//  * IoU matrix: every box is expanded once (BoxGeom: trig, corners, AABB, area; + a conservative fp32 AABB).
//    k_iou_pre streams the zeros of the matrix at HBM write speed and, interleaved with that, lists the pairs whose
//    AABBs overlap; k_iou_clip computes the exact IoU of one listed pair per lane.  Stores are row-major and
//    coalesced (the reference's kernel strides lanes along i and stores with stride M, iou_cuda.cu:22-27), pair
//    indices are 64-bit (reference overflows int at N*M >= 2^31, iou_cuda.cu:36,137).  k_iou2d is the single-kernel
//    form (no workspace, or list overflow).
//  * NMS: sweep-and-prune broad phase over the score-sorted boxes' AABBs -> candidate list -> exact IoU -> incoming
//    hit lists -> the greedy result as a fixed point, resolved in one launch.  The reference's structure -- all-pairs
//    "IoU > thr" bit matrix + a sweep of the sorted order (nms_cuda.cu:80-107 runs that on ONE thread) -- is kept as
//    the dense path behind a device-side flag.
#include "common.hpp"
#include <chrono>
#include "geom.hpp"
#include "lds_sort.hpp"
#include <stdlib.h>

namespace {

constexpr int kTileCols = 256;   // threads per block = columns per tile
constexpr int kTileRows = 64;    // rows per tile (LDS-staged)

// ---------------------------------------------------------------- box loaders
template <typename T> struct Box2D {     // rows of [.,5] = (x, y, w, h, r)
    static constexpr int kStride = 5;
    // (B: the element type in memory -- T, or float widened to double where it is loaded: D3D_F32_WIDE)
    template <typename B> __device__ static BoxGeom<T> load(const B *b) { return make_geom<T>((T)b[0], (T)b[1], (T)b[2], (T)b[3], (T)b[4]); }
};

struct Box3DGeom {
    BoxGeom<float> g;
    float zmin, zmax;
};

// ---------------------------------------------------------------- pairwise IoU, 2-D boxes
// K = columns per lane.  K = 16 / sizeof(T) makes every lane store 16 bytes per row (1 KiB per
// wave-instruction, the widest coalesced store) -- used whenever M % K == 0 keeps the rows 16-byte aligned.
template <typename T, int K> struct VecOf;
template <> struct VecOf<double, 2> { typedef double2 type; };
template <> struct VecOf<float, 4> { typedef float4 type; };
template <> struct VecOf<double, 1> { typedef double type; };
template <> struct VecOf<float, 1> { typedef float type; };

template <typename T, int K> __device__ __forceinline__ void store_row(T *out, const T (&v)[K])
{
    if constexpr (K * sizeof(T) > 16) {                 // several 16-byte stores per lane
        constexpr int V = 16 / (int)sizeof(T);
        typedef T vecv __attribute__((ext_vector_type(V)));
#pragma unroll
        for (int q = 0; q < K / V; q++) {
            vecv x;
#pragma unroll
            for (int e = 0; e < V; e++) x[e] = v[q * V + e];
            __builtin_nontemporal_store(x, reinterpret_cast<vecv *>(out) + q);
        }
    } else if constexpr (K == 1) __builtin_nontemporal_store(v[0], out);
    else if constexpr (K == 2) {
        typedef T vec2 __attribute__((ext_vector_type(2)));
        vec2 x = {v[0], v[1]};
        __builtin_nontemporal_store(x, reinterpret_cast<vec2 *>(out));
    } else {
        typedef T vec4 __attribute__((ext_vector_type(4)));
        vec4 x = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(x, reinterpret_cast<vec4 *>(out));
    }
}

template <typename T, bool ROTATED, int K>
__global__ __launch_bounds__(kTileCols) void k_iou2d(const T *__restrict__ b1, int64_t n, const T *__restrict__ b2,
                                                     int64_t m, T *__restrict__ ious, const unsigned int *only_if)
{
    if (only_if && !*only_if) return;     // fallback launch of the two-phase path: runs only after a list overflow
    __shared__ BoxGeom<T> rows[kTileRows];
    const int64_t i0 = (int64_t)blockIdx.y * kTileRows;
    const int64_t j0 = ((int64_t)blockIdx.x * kTileCols + threadIdx.x) * K;
    const int nrows = (int)((n - i0) < kTileRows ? (n - i0) : kTileRows);
    if (threadIdx.x < nrows) rows[threadIdx.x] = Box2D<T>::load(b1 + (i0 + threadIdx.x) * 5);
    BoxGeom<T> col[K];
    const bool active = j0 < m;      // M % K == 0 (host-checked): a lane's K columns are all valid or all not
    if (active) {
#pragma unroll
        for (int k = 0; k < K; k++) col[k] = Box2D<T>::load(b2 + (j0 + k) * 5);
    }
    __syncthreads();
    if (!active) return;
    T *out = ious + i0 * m + j0;
    for (int r = 0; r < nrows; r++) {
        const BoxGeom<T> a = rows[r];      // LDS broadcast read
        T v[K];
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = ROTATED ? iou_rbox(a, col[k]) : iou_aabb(a, col[k]);
        store_row<T, K>(out, v);
        out += m;
    }
}

// ---------------------------------------------------------------- rotated IoU, two-phase
// The polygon clip needs ~180 VGPRs in fp64; inside the streaming kernel that caps occupancy at 2 waves/SIMD and
// leaves most lanes idle whenever only a few pairs of a wavefront overlap.  So for `rbox` the matrix is produced
// in two phases (MI355X: 288 GB make a 1 GB candidate list a non-issue):
//   k_iou_pre   streams zeros at store bandwidth (few registers, full occupancy) and appends every pair whose
//               AABBs overlap to a global candidate list (wave-aggregated atomic append)
//   k_iou_clip  one candidate per lane -- dense wavefronts of clipping -- and scatters the non-zero IoUs
// If the list overflows (more candidates than its capacity) the monolithic kernel recomputes everything.
// The candidate list is split into nseg segments (1 or kListSegs) with a counter each, on separate cache lines:
// atomics on ONE address are serialised (~7 ns each), which dominated short launches where every workgroup reserves
// at about the same time.  A full segment counts as overflow (the single-kernel fallback recomputes everything).
constexpr int kListSegs = 8;
struct IouList { unsigned long long count[kListSegs * 16]; unsigned int overflow, nseg; };
__device__ __forceinline__ void list_reset(IouList *hdr, unsigned int nseg)
{
    for (int s = 0; s < kListSegs; s++) hdr->count[s * 16] = 0;
    hdr->overflow = 0;
    hdr->nseg = nseg;
}

// conservative fp32 AABB of a box for the candidate test (outward rounding; a degenerate box gets an empty AABB and is
// never a candidate: its IoU is 0 by the policy of geom.hpp).  Candidates are a superset of the exact AABB overlaps --
// k_iou_clip computes the exact value, which is 0 for the extra ones -- so the result does not depend on the rounding.
__device__ __forceinline__ float round_down(double x) { float f = (float)x; return (double)f > x ? nextafterf(f, -INFINITY) : f; }
__device__ __forceinline__ float round_up(double x) { float f = (float)x; return (double)f < x ? nextafterf(f, INFINITY) : f; }
__device__ __forceinline__ float round_down(float x) { return x; }
__device__ __forceinline__ float round_up(float x) { return x; }
template <typename T> __device__ __forceinline__ float4 cand_aabb(const BoxGeom<T> &g, bool rotated = true)
{
    // (method BOX measures the AABB itself, which has an area even when the rectangle has none)
    if (rotated && !(g.area > 0)) return make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
    return make_float4(round_down(g.xmin), round_down(g.ymin), round_up(g.xmax), round_up(g.ymax));
}
// strict overlap in x and y as ONE number: the smallest of the four gaps must be positive
__device__ __forceinline__ float aabb_gap(const float4 &a, const float4 &b)
{
    typedef float f2 __attribute__((ext_vector_type(2)));     // two packed subtractions (v_pk_add_f32) instead of four
    const f2 d1 = f2{b.z, b.w} - f2{a.x, a.y}, d2 = f2{a.z, a.w} - f2{b.x, b.y};
    return fminf(fminf(d1.x, d1.y), fminf(d2.x, d2.y));
}

// CORE: the six numbers of a box (centre, half-extent vectors) in one 64-byte (fp64) / 32-byte (fp32) aligned record instead of
// the 88 / 44-byte BoxGeom: what the rotated clip gathers per candidate -- one sector per box instead of two
// both operands of a pairwise call in ONE launch (a launch per operand costs ~6 us each at a few thousand boxes)
template <typename T, bool CORE, typename B = T>
__global__ __launch_bounds__(256) void k_geom2(const B *__restrict__ b1, int64_t n, BoxGeom<T> *g1, float4 *a1,
                                               const B *__restrict__ b2, int64_t m, BoxGeom<T> *g2, float4 *a2,
                                               IouList *hdr, unsigned int nseg, bool rotated)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (hdr && i == 0) list_reset(hdr, nseg);
    if (i >= n + m) return;
    const bool second = i >= n;
    if (second) i -= n;
    const BoxGeom<T> g = Box2D<T>::load((second ? b2 : b1) + i * 5);
    BoxGeom<T> *gd = second ? g2 : g1;
    if (CORE) reinterpret_cast<BoxCore<T> *>(gd)[i] = core_of(g);
    else gd[i] = g;
    (second ? a2 : a1)[i] = cand_aabb(g, rotated);
}

template <typename T, bool CORE = false>
__global__ __launch_bounds__(256) void k_geom(const T *__restrict__ boxes, int64_t n, BoxGeom<T> *geom, float4 *aabb,
                                              IouList *hdr, unsigned int nseg, bool rotated)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (hdr && i == 0) list_reset(hdr, nseg);
    if (i < n) {
        const BoxGeom<T> g = Box2D<T>::load(boxes + i * 5);
        if (CORE) reinterpret_cast<BoxCore<T> *>(geom)[i] = core_of(g);
        else geom[i] = g;
        aabb[i] = cand_aabb(g, rotated);  // 16 B per box: what k_iou_pre reads (coalesced) instead of the geometry
    }
}

// Tile = kTileRows rows x kPreCols columns.  Two independent jobs are interleaved row by row so that the stores of one
// overlap the ALU work of the other:
//   candidates  lane = 4 columns (AABBs in registers), rows broadcast from LDS; survivors go to a per-wavefront LDS
//               batch (fill count is wave-uniform: no workgroup barrier inside the row loop) that is flushed to the
//               global list with one atomic
//   zero fill   the matrix as one run of 4 KiB chunks dealt to the workgroups like a grid-stride loop (see below): every
//               wavefront store is 1 KiB and 1 KiB-aligned whatever m is (row-by-row stores are misaligned when
//               m * sizeof(T) is not a multiple of the 128-byte line: 2x slower), and the chip writes one moving window
constexpr int kPreK = 4;
constexpr int kPreCols = kTileCols * kPreK;
constexpr int kPreBatch = 1024;      // LDS batch entries per wavefront
// rows per workgroup: kTileRows, fewer for matrices that would otherwise launch too few workgroups to fill 256 CUs
// (5 k x 5 k at 64 rows: 395 workgroups = 1.5 wavefronts per SIMD, 81 us for 25 M tests + 200 MB)
// (round 6, profiles/r06_pre_rows_ab.txt: the rule below is the dense regime's -- the reference's benchmark boxes, 5 k x 5 k, want 8
// rows = 3125 workgroups: 57 us against 99 at 32 rows -- and costs the sparse regime on SMALL matrices: 8 k x 2 k iou3d 45 us at the
// 8 rows it picks, 31 at 32; config 4 and everything larger are at their best, or within 3 % of it.  The density is not known here.)
int g_pre_rows_override = 0;           // tools/pre_rows_ab.py
extern "C" void d3d_debug_set_pre_rows(int rows) { g_pre_rows_override = rows; }
static inline int pre_tile_rows(int64_t n, int64_t m)
{
    if (g_pre_rows_override) return g_pre_rows_override;
    int rows = kTileRows;
    int64_t wgs = d3d_divup(m, (int64_t)kPreCols) * d3d_divup(n, (int64_t)rows);
    while (rows > 8 && wgs < 2048) { rows >>= 1; wgs = d3d_divup(m, (int64_t)kPreCols) * d3d_divup(n, (int64_t)rows); }
    return rows;
}

template <typename T>
__global__ __launch_bounds__(kTileCols) void k_iou_pre(const float4 *__restrict__ ra, int64_t n,
                                                       const float4 *__restrict__ cb, int64_t m, T *__restrict__ ious,
                                                       IouList *hdr, unsigned long long *list, unsigned long long cap,
                                                       float fillv = 0.f /* T = float only: the matrix' background value */,
                                                       int tile_rows = kTileRows /* rows per workgroup, <= kTileRows */)
{
    constexpr int K = kPreK;
    typedef float vec16 __attribute__((ext_vector_type(4)));
    __shared__ float4 rbox[kTileRows];
    __shared__ unsigned int batch[kTileCols / 64][kPreBatch];   // (row << 16 | local column)
    __shared__ unsigned int wcnt[kTileCols / 64];
    __shared__ unsigned long long bbase;
    const int64_t i0 = (int64_t)blockIdx.y * tile_rows;
    const int64_t jb = (int64_t)blockIdx.x * kPreCols;          // first column of the block
    const int64_t j0 = jb + (int64_t)threadIdx.x * K;
    const int nrows = (int)((n - i0) < tile_rows ? (n - i0) : tile_rows);
    if (threadIdx.x < nrows) rbox[threadIdx.x] = ra[i0 + threadIdx.x];
    float4 cbox[K];
#pragma unroll
    for (int k = 0; k < K; k++)
        cbox[k] = j0 + k < m ? cb[j0 + k] : make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned int *q = batch[threadIdx.x >> 6];
    unsigned int wn = 0;                                          // wave-uniform fill of the batch
    const unsigned int sg = (blockIdx.x + blockIdx.y) & (hdr->nseg - 1);     // nseg is a power of two
    const unsigned long long segcap = cap / hdr->nseg;
    unsigned long long *seg = list + sg * segcap, *counter = &hdr->count[sg * 16];
    auto write_out = [&](unsigned long long base) {
        __builtin_amdgcn_wave_barrier();          // LDS ops of one wavefront complete in order: no s_barrier needed
        for (unsigned int t = lane; t < wn; t += 64) {
            const unsigned int e = q[t];
            if (base + t < segcap) seg[base + t] = ((unsigned long long)(i0 + (e >> 16)) << 32) | (unsigned long long)(jb + (e & 0xffffu));
            else hdr->overflow = 1;
        }
        wn = 0;
    };
    auto flush = [&]() {                                          // batch full (dense candidates): the wavefront reserves
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(counter, (unsigned long long)wn);
        write_out(__shfl(base, 0, 64));
    };
    // background fill of the WHOLE matrix as 16-byte vectors (the host checks the base pointer's alignment), in 4 KiB chunks
    // of 256 vectors dealt out like a grid-stride loop: workgroup L (in dispatch order) writes chunks L, L + G, L + 2 G, ...
    // (G = workgroups of the launch), two or so per row of its tile.  The workgroups resident at any moment have consecutive
    // L, so the chip writes ONE compact moving window of the matrix -- the DRAM-friendly order (bare nt stores: 6.5 TB/s
    // against 4.8-5.6 TB/s when every workgroup streams through a region of its own, tools/fill_bench.hip).  Which workgroup
    // zeroes which chunk is irrelevant to the result: the candidates are written by the next kernel.
    const size_t total_bytes = (size_t)n * (size_t)m * sizeof(T);
    const size_t nvec = total_bytes / 16, nchunk = (nvec + kTileCols - 1) / kTileCols;
    const size_t G = (size_t)gridDim.x * gridDim.y, L = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    vec16 *slab = ious ? reinterpret_cast<vec16 *>(ious) : nullptr;
    const vec16 z = {fillv, fillv, fillv, fillv};
    for (int r = 0; r < tile_rows; r++) {
        if (r < nrows) {
            const float4 fa = rbox[r];                        // LDS broadcast
            float g[K], best = -1.f;
#pragma unroll
            for (int k = 0; k < K; k++) { g[k] = aabb_gap(fa, cbox[k]); best = fmaxf(best, g[k]); }
            // (round 6: one test against the union of the lane's K boxes first, the K tests only behind it -- k_iou_pre<float> the
            // same within 3 % at 400 MB .. 6.4 GB, profiles/r06_iou3d_fused_ab.txt: the fill is not bound by these instructions)
            if (__ballot(best > 0.f)) {                       // some lane of the wavefront has a candidate in this row
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const bool cand = g[k] > 0.f;
                    const unsigned long long mask = __ballot(cand);
                    if (mask) {
                        const unsigned int cnt = (unsigned int)__popcll(mask);
                        if (wn + cnt > (unsigned int)kPreBatch) flush();
                        if (cand)
                            q[wn + __popcll(mask & ((1ull << lane) - 1))] = ((unsigned)r << 16) | (unsigned)(threadIdx.x * K + k);
                        wn += cnt;
                    }
                }
            }
        }
        if (slab)
            for (size_t c = L + (size_t)r * G; c < nchunk; c += (size_t)tile_rows * G) {
                const size_t v = c * kTileCols + threadIdx.x;
                if (v < nvec) __builtin_nontemporal_store(z, slab + v);
            }
    }
    if (slab && L == 0 && threadIdx.x == 0)                      // matrix size not a multiple of 16 bytes
        for (size_t e = nvec * (16 / sizeof(T)); e < (size_t)n * (size_t)m; e++) ious[e] = (T)fillv;
    // what is left in the four batches is reserved with ONE atomic per workgroup: atomics on the list counter are
    // serialised at ~7 ns each, and every workgroup of a short launch gets here at about the same time
    const int wave = threadIdx.x >> 6;
    if (lane == 0) wcnt[wave] = wn;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        bbase = total ? atomicAdd(counter, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    unsigned long long base = bbase;
    for (int w = 0; w < wave; w++) base += wcnt[w];
    write_out(base);
}

// One candidate per lane.  ROTATED: rectangles whose bounding boxes overlap are often disjoint all the same (a third of the
// candidates for boxes of random orientation), and the clip costs ~10x a separating-axis test -- so a workgroup takes 1024
// candidates at a time, tests them (4 per lane), compacts the survivors through LDS and clips THOSE on dense wavefronts.
constexpr int kClipChunk = 1024;
// S: the matrix' element type -- T, or float under double arithmetic (D3D_F64_M32: the value is rounded where it is stored, what
// box2d_iou(precise=True) does to fp32 boxes with a cast of the whole fp64 matrix, reference box/__init__.py:204-205, 224)
template <typename T, bool ROTATED, typename S = T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_iou_clip(const BoxGeom<T> *__restrict__ ga, const BoxGeom<T> *__restrict__ gb,
                                                  int64_t n, int64_t m, S *__restrict__ ious, const IouList *hdr,
                                                  const unsigned long long *__restrict__ list, unsigned long long cap)
{
    if (hdr->overflow) {
        // the candidate list overflowed (more overlapping pairs than the workspace's list holds): EVERY pair, one per lane, from the
        // same geometry records -- slow, correct, and (round 6) no launch of its own in the calls that never need it (k_iou2d's
        // workgroups used to be launched behind every call to look at this flag and leave: 6 us of a 100 us operator)
        const unsigned long long total = (unsigned long long)n * (unsigned long long)m, stride = (unsigned long long)gridDim.x * blockDim.x;
        for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
            const int64_t i = (int64_t)(t / (unsigned long long)m), j = (int64_t)(t % (unsigned long long)m);
            T v;
            if (ROTATED) {
                const BoxCore<T> *ca = reinterpret_cast<const BoxCore<T> *>(ga), *cb = reinterpret_cast<const BoxCore<T> *>(gb);
                v = sat_separated(ca[i], cb[j]) ? (T)0 : iou_rbox_core<T, true>(ca[i], cb[j]);
            } else v = iou_aabb(ga[i], gb[j]);
            if (v != 0) ious[t] = (S)v;
        }
        return;
    }
    const unsigned long long segcap = cap / hdr->nseg;
    for (unsigned int sg = 0; sg < hdr->nseg; sg++) {
    const unsigned long long cnt = hdr->count[sg * 16], total = cnt < segcap ? cnt : segcap;
    const unsigned long long *seg = list + sg * segcap;
    if (ROTATED) {
        const BoxCore<T> *ca = reinterpret_cast<const BoxCore<T> *>(ga), *cb = reinterpret_cast<const BoxCore<T> *>(gb);   // k_geom<T, true>
        __shared__ unsigned long long surv[kClipChunk];
        __shared__ unsigned int ns;
        for (unsigned long long c0 = (unsigned long long)blockIdx.x * kClipChunk; c0 < total; c0 += (unsigned long long)gridDim.x * kClipChunk) {
            if (threadIdx.x == 0) ns = 0;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < kClipChunk / 256; u++) {
                const unsigned long long t = c0 + (unsigned)u * 256u + threadIdx.x;
                if (t < total) {
                    const unsigned long long e = seg[t];
                    if (!sat_separated(ca[e >> 32], cb[e & 0xffffffffull])) surv[atomicAdd(&ns, 1u)] = e;
                }
            }
            __syncthreads();
            const unsigned int n_s = ns;
            for (unsigned int q = threadIdx.x; q < n_s; q += 256) {
                const unsigned long long e = surv[q];
                const int64_t i = (int64_t)(e >> 32), j = (int64_t)(e & 0xffffffffull);
                const T v = iou_rbox_core<T, true>(ca[i], cb[j]);
                if (v != 0) ious[i * m + j] = (S)v;
            }
            __syncthreads();
        }
    } else {
        const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
        for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
            const unsigned long long e = seg[t];
            const int64_t i = (int64_t)(e >> 32), j = (int64_t)(e & 0xffffffffull);
            const T v = iou_aabb(ga[i], gb[j]);
            if (v != 0) ious[i * m + j] = (S)v;
        }
    }
    }
}

// Small matrices (a frame's detections against its ground truth: up to kIouSmallPairs pairs): ONE launch, one pair per lane,
// the geometry of both boxes rebuilt per pair (two sincos: nothing at these sizes) instead of geometry x 2 + fill + clip +
// fallback = five launches of ~5 us each.  Same candidate test (conservative fp32 AABBs, empty for degenerate boxes) and the
// same per-pair function as the two-phase path: identical values.
constexpr unsigned long long kIouSmallPairs = 1ull << 16;
template <typename T, bool ROTATED, typename S = T, typename B = T>
__global__ __launch_bounds__(256) void k_iou_small(const B *__restrict__ b1, int64_t n, const B *__restrict__ b2, int64_t m,
                                                   S *__restrict__ ious)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * m) return;
    const int64_t i = idx / m, j = idx - i * m;
    const BoxGeom<T> a = Box2D<T>::load(b1 + i * 5), b = Box2D<T>::load(b2 + j * 5);
    T v = 0;
    if (aabb_gap(cand_aabb(a, ROTATED), cand_aabb(b, ROTATED)) > 0.f) v = ROTATED ? iou_rbox(a, b) : iou_aabb(a, b);
    ious[idx] = (S)v;
}

// ---------------------------------------------------------------- pairwise "3D IoU" (BEV x z), fp32
// box = (x, y, z, lx, ly, lz, rz); dgal_wrap.h:45-91
// clip_dims: the matcher's guard against "really weird boxes with unusual size" (matcher.pyx:49-51: np.clip(dims, -1e3, 1e3))
__device__ __forceinline__ Box3DGeom load3d(const float *b, bool clip_dims = false)
{
    float lx = b[3], ly = b[4], lz = b[5];
    if (clip_dims) {
        lx = fminf(fmaxf(lx, -1e3f), 1e3f); ly = fminf(fmaxf(ly, -1e3f), 1e3f); lz = fminf(fmaxf(lz, -1e3f), 1e3f);
    }
    Box3DGeom r;
    r.g = make_geom<float>(b[0], b[1], lx, ly, b[6]);
    r.zmax = b[2] + lz / 2;
    r.zmin = b[2] - lz / 2;
    return r;
}

// stride: floats per row (7, or 9 for the matcher's [n,9] arrays entered at column 2); complement: store 1 - iou
template <bool ROTATED, int K>
__global__ __launch_bounds__(kTileCols) void k_iou3d(const float *__restrict__ b1, int64_t n,
                                                     const float *__restrict__ b2, int64_t m, float *__restrict__ out_,
                                                     const unsigned int *only_if, int stride = 7, bool complement = false)
{
    if (only_if && !*only_if) return;
    __shared__ Box3DGeom rows[kTileRows];
    const int64_t i0 = (int64_t)blockIdx.y * kTileRows;
    const int64_t j0 = ((int64_t)blockIdx.x * kTileCols + threadIdx.x) * K;
    const int nrows = (int)((n - i0) < kTileRows ? (n - i0) : kTileRows);
    if (threadIdx.x < nrows) rows[threadIdx.x] = load3d(b1 + (i0 + threadIdx.x) * stride, complement);
    Box3DGeom col[K];
    const bool active = j0 < m;
    if (active) {
#pragma unroll
        for (int k = 0; k < K; k++) col[k] = load3d(b2 + (j0 + k) * stride, complement);
    }
    __syncthreads();
    if (!active) return;
    float *out = out_ + i0 * m + j0;
    for (int r = 0; r < nrows; r++) {
        const Box3DGeom a = rows[r];
        float v[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            float iou2d = ROTATED ? iou_rbox(a.g, col[k].g) : iou_aabb(a.g, col[k].g);
            v[k] = 0.f;
            if (iou2d != 0.f) {
                float imax = fminf(a.zmax, col[k].zmax), imin = fmaxf(a.zmin, col[k].zmin);
                float umax = fmaxf(a.zmax, col[k].zmax), umin = fminf(a.zmin, col[k].zmin);
                float i = fmaxf(imax - imin, 0.f);
                float u = fmaxf(umax - umin, (float)1e-6);
                v[k] = iou2d * (i / u);
            }
            if (complement) v[k] = 1 - v[k];
        }
        store_row<float, K>(out, v);
        out += m;
    }
}

// the same for the pairwise 3D IoU / the matcher's distance (see k_iou_small)
template <bool ROTATED>
__global__ __launch_bounds__(256) void k_iou3d_small(const float *__restrict__ b1, int64_t n, const float *__restrict__ b2, int64_t m,
                                                     float *__restrict__ out, int stride, bool complement)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * m) return;
    const int64_t i = idx / m, j = idx - i * m;
    const Box3DGeom a = load3d(b1 + i * stride, complement), b = load3d(b2 + j * stride, complement);
    float v = 0.f;
    if (aabb_gap(cand_aabb(a.g, ROTATED), cand_aabb(b.g, ROTATED)) > 0.f) {
        const float iou2d = ROTATED ? iou_rbox(a.g, b.g) : iou_aabb(a.g, b.g);
        if (iou2d != 0.f) {
            const float imax = fminf(a.zmax, b.zmax), imin = fmaxf(a.zmin, b.zmin);
            const float umax = fmaxf(a.zmax, b.zmax), umin = fminf(a.zmin, b.zmin);
            v = iou2d * (fmaxf(imax - imin, 0.f) / fmaxf(umax - umin, (float)1e-6));
        }
    }
    out[idx] = complement ? 1 - v : v;
}

// ---------------------------------------------------------------- IoU backward (loss path, "next" row 2)
// grad_boxes1[i] = sum_j grad[i,j] * dIoU(i,j)/d box1_i, grad_boxes2[j] likewise (reference iou.cpp:48-93, 143-211); analytic
// gradients (geom.hpp).  The reference's CUDA kernels accumulate with plain += from many threads (iou_cuda.cu:72-73,184-185: a
// data race).  Rounds 2-4 ran one listed pair per lane (k_iou_pre's candidate list), the row sums by a segmented scan across the
// wavefront and FIVE scattered fp64 atomics per pair for the columns; round 5 replaced that by tiles:
// IoU backward over TILES (round 5).  The list form paid five scattered fp64 atomics per overlapping pair for the
// column gradients: 720 of k_iou_grad's 1434 us on the reference's benchmark boxes at 5 k x 5 k (5.15 M overlapping pairs; without
// those atomics 714 us, without the segmented scan either 645 us) -- plus 109 us of k_iou_pre for the list.  Here a workgroup owns
// a tile of `tile_rows` x 256 pairs: every wavefront walks its 64 columns row by row, marks the pairs with a weight whose
// bounding boxes overlap (the candidate test of k_iou_pre, on the same conservative fp32 boxes) and queues them in LDS until 64
// are together; then every lane takes one pair through iou_rbox_grad and adds the ten products to LDS accumulators (ds_add_f64:
// per row of the tile and wavefront, per column).  The accumulators go to memory once per tile, side by side.  No list.
constexpr int kGradCols = 256;
constexpr int kMarkStripes = 64;             // counters of the marks

// The marking as a kernel of its own (a few dozen VGPRs, full occupancy): one bit per pair that has a weight, overlapping
// conservative bounding boxes and (rotated) no separating axis, one 64-bit word per row and wavefront of columns.  Inside the
// gradient kernel -- 246 VGPRs, two wavefronts per SIMD -- the same loop ran the 400 M pairs of 20 k x 20 k boxes at config 3's
// density in 0.62 ms; the list form of rounds 2-4 took 0.22 ms for that call.  Lane = 2 columns (two words per wavefront and row).
template <typename T, bool ROTATED, typename G = T /* element of grad[n,m]: T, or float under double arithmetic (D3D_F64_M32) */>
__global__ __launch_bounds__(kGradCols) void k_iou_grad_mark(const BoxGeom<T> *__restrict__ ga, const float4 *__restrict__ ra, int64_t n,
                                                             const BoxGeom<T> *__restrict__ gb, const float4 *__restrict__ cb, int64_t m,
                                                             const G *__restrict__ grad, unsigned long long *__restrict__ bitmap,
                                                             int64_t wpr, int tile_rows, unsigned long long *nmarks)
{
    __shared__ BoxCore<T> rcore[kTileRows];
    __shared__ float4 rbox[kTileRows];
    unsigned int mymarks = 0;                          // (lane 0 of every wavefront: the bits it wrote)
    const int64_t i0 = (int64_t)blockIdx.y * tile_rows, jb = (int64_t)blockIdx.x * (2 * kGradCols);
    const int nrows = (int)((n - i0) < tile_rows ? (n - i0) : tile_rows);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    if (threadIdx.x < nrows) { rcore[threadIdx.x] = core_of(ga[i0 + threadIdx.x]); rbox[threadIdx.x] = ra[i0 + threadIdx.x]; }
    __syncthreads();
    // the wavefront's 128 columns: lane l holds columns l and 64 + l of them (word 2 * wave and 2 * wave + 1 of the workgroup)
    const int64_t j0 = jb + wave * 128 + lane, j1 = j0 + 64;
    if (jb + wave * 128 >= m) return;
    const bool a0 = j0 < m, a1 = j1 < m;
    const float4 c0 = a0 ? cb[j0] : make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
    const float4 c1 = a1 ? cb[j1] : make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
    BoxCore<T> k0, k1;
    if (ROTATED) { k0 = core_of(gb[a0 ? j0 : m - 1]); k1 = core_of(gb[a1 ? j1 : m - 1]); }
    const int64_t w0 = (jb + wave * 128) >> 6;
    const bool two = jb + wave * 128 + 64 < m;
    for (int r0 = 0; r0 < nrows; r0 += 4) {
        bool x0[4], x1[4];
        bool some = false;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float4 rb = rbox[r0 + u < nrows ? r0 + u : r0];
            x0[u] = (r0 + u < nrows) & (aabb_gap(rb, c0) > 0.f);
            x1[u] = (r0 + u < nrows) & (aabb_gap(rb, c1) > 0.f);
            some |= x0[u] | x1[u];
        }
        if (!__any(some)) {                            // nothing near in these four rows (the usual case of a sparse scene): eight zero words
            const int u = lane >> 1;
            if (lane < 8 && r0 + u < nrows && ((lane & 1) == 0 || two)) bitmap[(i0 + r0 + u) * wpr + w0 + (lane & 1)] = 0ull;
            continue;
        }
        {
            T g0[4], g1[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {                          // the weights of the candidate pairs only, all loads in flight together
                g0[u] = x0[u] ? (T)grad[(i0 + r0 + u) * m + j0] : (T)0;
                g1[u] = x1[u] ? (T)grad[(i0 + r0 + u) * m + j1] : (T)0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                x0[u] = x0[u] & (g0[u] != 0);
                x1[u] = x1[u] & (g1[u] != 0);
                if (ROTATED) {
                    if (x0[u]) x0[u] = !sat_separated(rcore[r0 + u], k0);
                    if (x1[u]) x1[u] = !sat_separated(rcore[r0 + u], k1);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (r0 + u >= nrows) break;
            const unsigned long long b0 = __ballot(x0[u]), b1 = __ballot(x1[u]);
            if (lane == 0) {
                unsigned long long *dst = bitmap + (i0 + r0 + u) * wpr + w0;
                dst[0] = b0;
                if (two) dst[1] = b1;
                mymarks += (unsigned int)__popcll(b0) + (unsigned int)__popcll(b1);
            }
        }
    }
    // (64 counters: one address would take the whole grid's atomics one after the other -- the marking went from 208 to 395 us)
    if (lane == 0 && mymarks) atomicAdd(&nmarks[(blockIdx.x + 7u * blockIdx.y + 13u * (unsigned)wave) & (kMarkStripes - 1)], (unsigned long long)mymarks);
}

// dense or sparse?  The tiles below pay one pass through the gradient routine per wavefront and 64 marked pairs of its 64 x 64
// part -- right when the marks are many; when they are few (config 3's density: 5 per tile of 16 k pairs) almost every
// wavefront would run the routine for one or two lanes.  Below one mark per 128 pairs the marks are compacted GLOBALLY instead
// (k_iou_grad_sparse: 64 consecutive marks of the bitmap per wavefront, gradients by atomics -- few pairs, few atomics).  Both
// kernels are launched; the one whose case it is not exits at once.
__global__ __launch_bounds__(kMarkStripes) void k_iou_grad_decide(unsigned long long *nmarks, int64_t n, int64_t m)
{
    unsigned long long tot = nmarks[threadIdx.x];
#pragma unroll
    for (int o = kMarkStripes / 2; o > 0; o >>= 1) tot += __shfl_xor(tot, o, kWave);
    if (threadIdx.x == 0) nmarks[kMarkStripes] = tot * 128ull >= (unsigned long long)n * (unsigned long long)m ? 1ull : 0ull;
}
__device__ __forceinline__ bool grad_marks_dense(const unsigned long long *nmarks, int64_t, int64_t) { return nmarks[kMarkStripes] != 0; }

// a frame's worth of pairs (<= kIouSmallPairs): ONE launch, one pair per lane, geometry rebuilt per pair, atomics per pair --
// the marks / decision / tiles / compaction above are five launches of ~5 us each, more than the arithmetic at this size
template <typename T, bool ROTATED, typename G = T, typename B = T /* element of the box rows in memory (D3D_F32_WIDE: float) */>
__global__ __launch_bounds__(256) void k_iou_grad_small(const B *__restrict__ b1, int64_t n, const B *__restrict__ b2, int64_t m,
                                                        const G *__restrict__ grad, T *g1, T *g2)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * m) return;
    const T g = (T)grad[idx];
    if (g == 0) return;
    const int64_t i = idx / m, j = idx - i * m;
    const BoxGeom<T> a = Box2D<T>::load(b1 + i * 5), b = Box2D<T>::load(b2 + j * 5);
    if (!(aabb_gap(cand_aabb(a, ROTATED), cand_aabb(b, ROTATED)) > 0.f)) return;
    T da[5], db[5];
    if (ROTATED) iou_rbox_grad<T>(a, b, (T)b1[i * 5 + 2], (T)b1[i * 5 + 3], (T)b2[j * 5 + 2], (T)b2[j * 5 + 3], da, db);
    else iou_aabb_grad<T, B>(a, b, b1 + i * 5, b2 + j * 5, da, db);
#pragma unroll
    for (int k = 0; k < 5; k++) {
        if (da[k] != 0) atomicAdd(&g1[i * 5 + k], g * da[k]);
        if (db[k] != 0) atomicAdd(&g2[j * 5 + k], g * db[k]);
    }
}

constexpr int kSparseWords = 2048;                 // bitmap words per workgroup (131 k pairs)
template <typename T, bool ROTATED, typename G = T, typename B = T>
__global__ __launch_bounds__(256) void k_iou_grad_sparse(const BoxGeom<T> *__restrict__ ga, const B *__restrict__ b1, int64_t n,
                                                         const BoxGeom<T> *__restrict__ gb, const B *__restrict__ b2, int64_t m,
                                                         const G *__restrict__ grad, T *g1, T *g2,
                                                         const unsigned long long *__restrict__ bitmap, int64_t wpr,
                                                         const unsigned long long *nmarks)
{
    if (grad_marks_dense(nmarks, n, m)) return;
    __shared__ unsigned long long smem[256 / kWave];
    __shared__ unsigned int pre[kSparseWords];         // marks before word k of the chunk
    __shared__ unsigned long long wv[kSparseWords];
    const int64_t nwords = n * wpr, base = (int64_t)blockIdx.x * kSparseWords;
    constexpr int PER = kSparseWords / 256;
    unsigned int cnt[PER];
    unsigned long long mine = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {                    // thread t: words PER t .. PER t + PER - 1 of the chunk
        const int64_t w = base + (int64_t)threadIdx.x * PER + k;
        const unsigned long long x = w < nwords ? bitmap[w] : 0ull;
        wv[threadIdx.x * PER + k] = x;
        cnt[k] = (unsigned int)__popcll(x);
        mine += cnt[k];
    }
    unsigned long long total;
    unsigned long long ex = block_excl_scan_u64<256>(mine, &total, smem);
#pragma unroll
    for (int k = 0; k < PER; k++) { pre[threadIdx.x * PER + k] = (unsigned int)ex; ex += cnt[k]; }
    __syncthreads();
    for (unsigned int t = threadIdx.x; t < (unsigned int)total; t += 256) {
        int lo = 0, hi = kSparseWords;                 // the last word with pre[word] <= t
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= t) lo = mid; else hi = mid;
        }
        unsigned long long x = wv[lo];
        for (unsigned int skip = t - pre[lo]; skip; skip--) x &= x - 1;       // drop the marks before this one
        const int bit = __ffsll((long long)x) - 1;
        const int64_t w = base + lo, i = w / wpr, j = (w - i * wpr) * 64 + bit;
        const T g = (T)grad[i * m + j];
        T da[5], db[5];
        if (ROTATED) iou_rbox_grad<T>(ga[i], gb[j], (T)b1[i * 5 + 2], (T)b1[i * 5 + 3], (T)b2[j * 5 + 2], (T)b2[j * 5 + 3], da, db);
        else iou_aabb_grad<T, B>(ga[i], gb[j], b1 + i * 5, b2 + j * 5, da, db);
#pragma unroll
        for (int k = 0; k < 5; k++) {
            if (da[k] != 0) atomicAdd(&g1[i * 5 + k], g * da[k]);
            if (db[k] != 0) atomicAdd(&g2[j * 5 + k], g * db[k]);
        }
    }
}

// (round 6: three wavefronts per SIMD for the fp64 rotated form -- 226 -> 168 VGPRs with 80 bytes of scratch, possible since its LDS
// went from 57 to 44 KB: 603 -> 539 us on the reference's 5 k x 5 k benchmark boxes, profiles/r06_pre_rows_ab.txt; the other three
// forms need fewer registers than that anyway)
template <typename T, bool ROTATED, typename G = T, typename B = T>
__global__ __launch_bounds__(kGradCols) __attribute__((amdgpu_waves_per_eu(3))) void k_iou_grad_tiles(const BoxGeom<T> *__restrict__ ga, const float4 *__restrict__ ra,
                                                              const B *__restrict__ b1, int64_t n, const BoxGeom<T> *__restrict__ gb,
                                                              const float4 *__restrict__ cb, const B *__restrict__ b2, int64_t m,
                                                              const G *__restrict__ grad, T *g1, T *g2, int tile_rows,
                                                              const unsigned long long *__restrict__ bitmap, int64_t wpr,
                                                              const unsigned long long *nmarks)
{
    if (!grad_marks_dense(nmarks, n, m)) return;       // few marks: k_iou_grad_sparse's case
    // (round 6) rotated boxes: centre + half-extent vectors (48 of BoxGeom's 88 bytes in fp64) are all the gradient routine reads of a
    // MARKED pair -- 57 -> 44 KB of LDS, three workgroups per CU instead of two
    struct Core6 { T cx, cy, ux, uy, vx, vy; };
    typedef typename std::conditional<ROTATED, Core6, BoxGeom<T>>::type Geo;
    __shared__ Geo rgeo[kTileRows];
    __shared__ float4 rbox[kTileRows];
    __shared__ T rwh[kTileRows][2];
    __shared__ Geo cgeo[kGradCols];
    __shared__ T cwh[kGradCols][2];
    __shared__ T racc[kGradCols / 64][kTileRows][5];
    __shared__ T cacc[kGradCols][5];
    __shared__ unsigned short queue[kGradCols / 64][64];
    __shared__ T qg[kGradCols / 64][64];
    const int64_t i0 = (int64_t)blockIdx.y * tile_rows, jb = (int64_t)blockIdx.x * kGradCols, j = jb + threadIdx.x;
    const int nrows = (int)((n - i0) < tile_rows ? (n - i0) : tile_rows);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const bool active = j < m;
    if (threadIdx.x < nrows) {
        const int64_t i = i0 + threadIdx.x;
        if constexpr (ROTATED) { const BoxGeom<T> g = ga[i]; rgeo[threadIdx.x] = Core6{g.cx, g.cy, g.ux, g.uy, g.vx, g.vy}; }
        else rgeo[threadIdx.x] = ga[i];
        rbox[threadIdx.x] = ra[i];
        rwh[threadIdx.x][0] = (T)b1[i * 5 + 2]; rwh[threadIdx.x][1] = (T)b1[i * 5 + 3];
    }
    float4 cbox = make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
    if (active) {
        if constexpr (ROTATED) { const BoxGeom<T> g = gb[j]; cgeo[threadIdx.x] = Core6{g.cx, g.cy, g.ux, g.uy, g.vx, g.vy}; }
        else cgeo[threadIdx.x] = gb[j];
        cbox = cb[j];
        cwh[threadIdx.x][0] = (T)b2[j * 5 + 2]; cwh[threadIdx.x][1] = (T)b2[j * 5 + 3];
    }
#pragma unroll
    for (int k = 0; k < 5; k++) { racc[wave][lane][k] = 0; cacc[threadIdx.x][k] = 0; }
    __syncthreads();
    if ((j & ~(int64_t)63) >= m) return;               // a wavefront past the last column (no barrier below)
    unsigned short *q = queue[wave];
    T *qw = qg[wave];
    unsigned int wn = 0;
    auto process = [&]() {
        __builtin_amdgcn_wave_barrier();
        if (lane < (int)wn) {
            const unsigned int e = q[lane], r = e >> 8, c = e & 63u;
            const T g = qw[lane];
            T da[5], db[5];
            if constexpr (ROTATED) {
                const Core6 ca = rgeo[r], cc = cgeo[wave * 64 + c];
                BoxGeom<T> a, b;                            // (the fields the routine reads without its bounding-box test)
                a.cx = ca.cx; a.cy = ca.cy; a.ux = ca.ux; a.uy = ca.uy; a.vx = ca.vx; a.vy = ca.vy; a.area = 4 * (ca.ux * ca.vy - ca.uy * ca.vx);
                b.cx = cc.cx; b.cy = cc.cy; b.ux = cc.ux; b.uy = cc.uy; b.vx = cc.vx; b.vy = cc.vy; b.area = 4 * (cc.ux * cc.vy - cc.uy * cc.vx);
                iou_rbox_grad<T, false>(a, b, rwh[r][0], rwh[r][1], cwh[wave * 64 + c][0], cwh[wave * 64 + c][1], da, db);
            } else iou_aabb_grad<T, B>(rgeo[r], cgeo[wave * 64 + c], b1 + (i0 + r) * 5, b2 + (jb + wave * 64 + c) * 5, da, db);
#pragma unroll
            for (int k = 0; k < 5; k++) {
                if (da[k] != 0) atomicAdd(&racc[wave][r][k], g * da[k]);
                if (db[k] != 0) atomicAdd(&cacc[wave * 64 + c][k], g * db[k]);
            }
        }
        wn = 0;
        __builtin_amdgcn_wave_barrier();
    };
    // the tile's words of the bitmap (k_iou_grad_mark), one row per lane; a marked pair's weight is fetched when it is queued
    const unsigned long long words = lane < nrows ? bitmap[(i0 + lane) * wpr + (j >> 6)] : 0ull;
    if (__any(words != 0)) {
        const G *gp = grad + i0 * m + (active ? j : m - 1);
        for (int r = 0; r < nrows; r++) {
            const unsigned long long word = __shfl(words, r, kWave);
            if (word == 0) continue;
            const bool mark = (word >> lane) & 1ull;
            const T g = mark ? (T)gp[(int64_t)r * m] : (T)0;
            const unsigned int cnt = (unsigned int)__popcll(word);
            if (wn + cnt > 64u) process();
            if (mark) {
                const unsigned int at = wn + (unsigned int)__popcll(word & ((1ull << lane) - 1));
                q[at] = (unsigned short)((r << 8) | lane);
                qw[at] = g;
            }
            wn += cnt;
        }
    }
    if (wn) process();
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const T vr = lane < nrows ? racc[wave][lane][k] : (T)0, vc = cacc[threadIdx.x][k];
        if (vr != 0) atomicAdd(&g1[(i0 + lane) * 5 + k], vr);
        if (active && vc != 0) atomicAdd(&g2[j * 5 + k], vc);
    }
}

// ---------------------------------------------------------------- "3D IoU", two-phase (same scheme as rbox)
template <bool ROTATED>
__global__ __launch_bounds__(256) void k_iou3d_clip(const BoxGeom<float> *__restrict__ ga, const float2 *__restrict__ za,
                                                    const BoxGeom<float> *__restrict__ gb, const float2 *__restrict__ zb,
                                                    int64_t n, int64_t m, float *__restrict__ out, const IouList *hdr,
                                                    const unsigned long long *__restrict__ list, unsigned long long cap,
                                                    bool complement = false)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x, segcap = cap / hdr->nseg;
    auto pair = [&](int64_t i, int64_t j) -> float {
        const float iou2d = ROTATED ? iou_rbox(ga[i], gb[j]) : iou_aabb(ga[i], gb[j]);
        if (iou2d == 0.f) return 0.f;
        const float2 a = za[i], b = zb[j];
        const float imax = fminf(a.y, b.y), imin = fmaxf(a.x, b.x);
        const float umax = fmaxf(a.y, b.y), umin = fminf(a.x, b.x);
        return iou2d * (fmaxf(imax - imin, 0.f) / fmaxf(umax - umin, (float)1e-6));
    };
    if (hdr->overflow) {
        // the candidate list overflowed (more than the workspace's capacity of overlapping pairs): EVERY pair, one per lane --
        // slow, correct, and no launch of its own in the calls that never need it
        const unsigned long long total = (unsigned long long)n * (unsigned long long)m;
        for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
            const float v = pair((int64_t)(t / (unsigned long long)m), (int64_t)(t % (unsigned long long)m));
            out[t] = complement ? 1 - v : v;
        }
        return;
    }
    for (unsigned int sg = 0; sg < hdr->nseg; sg++) {
    const unsigned long long cnt = hdr->count[sg * 16], total = cnt < segcap ? cnt : segcap;
    const unsigned long long *seg = list + sg * segcap;
    for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const unsigned long long e = seg[t];
        const int64_t i = (int64_t)(e >> 32), j = (int64_t)(e & 0xffffffffull);
        const float v = pair(i, j);
        if (v != 0.f) out[i * m + j] = complement ? 1 - v : v;
    }
    }
}

// Round 6 (config 4 at 0.48 of HBM).  Five launches were three too many for a 100 us operator: both operands'
// geometry in ONE launch (k_geom3d2, which also resets the list), and the all-pairs fallback for an overflowed list inside
// k_iou3d_clip instead of a launch of its own that returns at once (6 us each on this stack): 105 -> 92 us per call.
// Built, measured and NOT kept (profiles/r06_iou3d_fused_ab.txt, bit-identical outputs): ONE launch that fills, tests and clips --
// every workgroup stores the background over the non-candidates of its own elements and clips its candidates itself, from its
// wavefronts' LDS batches, so that fill and values have disjoint writers and need no order -- (a) over tiles of its own (32 rows x
// 1024 columns): 141 us at config 4, 2.9 TB/s; (b) over dealt chunks whose columns repeat (G x 1024 a multiple of m): 143 us.  The
// clip is the reason, not the store pattern: at config 4's density a wavefront ends up with a handful of candidates, and clipping 5
// takes as long as clipping 64 -- every workgroup holds its wave slots for that latency with nothing to store.  (Small matrices,
// where the three launches' fixed costs dominate, did gain: 8 k x 2 k 50 -> 27 us.)
__global__ __launch_bounds__(256) void k_geom3d2(const float *__restrict__ b1, int64_t n, BoxGeom<float> *g1, float4 *a1, float2 *z1,
                                                 const float *__restrict__ b2, int64_t m, BoxGeom<float> *g2, float4 *a2, float2 *z2,
                                                 IouList *hdr, unsigned int nseg, bool rotated, int stride, bool clip_dims)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) list_reset(hdr, nseg);
    if (i < n) {
        const Box3DGeom g = load3d(b1 + i * stride, clip_dims);
        g1[i] = g.g; a1[i] = cand_aabb(g.g, rotated); z1[i] = make_float2(g.zmin, g.zmax);
    } else if (i - n < m) {
        const int64_t j = i - n;
        const Box3DGeom g = load3d(b2 + j * stride, clip_dims);
        g2[j] = g.g; a2[j] = cand_aabb(g.g, rotated); z2[j] = make_float2(g.zmin, g.zmax);
    }
}

// ---------------------------------------------------------------- NMS
// Greedy hard NMS in score order (nms.cpp:32-59) without the reference's serial collector:
//   prepare  geometry in score order (N trig evaluations, not N^2), conservative fp32 AABBs, initial states
//   cand     broad phase: sweep-and-prune along x over the conservative AABBs -> candidate pair list
//   hits     narrow phase: exact IoU per candidate; every hit (p suppresses q, p < q in score order) is appended
//            to q's incoming list
//   resolve  the greedy result is the unique fixed point of: q is suppressed iff some KEPT earlier p hits it,
//            kept iff all earlier hitters are suppressed; one launch in which every box polls its hitters
//   dense    only if a list overflowed or a dependency chain is very long: all-pairs bit matrix (the reference's
//            nms_cuda.cu layout) + one workgroup sweeping it (diagonal block resolved by a wavefront with lane
//            broadcasts, rows OR-ed into LDS)
extern "C" int d3d_internal_argsort_desc_i32(const int32_t *keys, int64_t n, int32_t *order, void *ws, size_t ws_bytes,
                                             hipStream_t st);      // sort.hip
extern "C" size_t d3d_internal_argsort_i32_bytes(int64_t n);
extern "C" int d3d_internal_crop2dr_grid_f32(const float *points, int64_t n, const float *boxes, int64_t m, uint8_t *out,
                                             hipStream_t st);      // crop.hip

enum { kUndecided = 0, kKept = 1, kSuppressed = 2 };
constexpr int kIncTile = 1024;      // boxes per workgroup of k_nms_incscan

struct NmsFlags { unsigned int need_sweep, undecided, scan_gave_up, pad; };     // FIRST thing in the workspace (d3d_nms2d_status)
constexpr int kNmsListSegs = 1;     // see list_segments(): segmenting the list did not pay
struct NmsCand { unsigned long long count[kNmsListSegs * 16]; };   // entries appended (may exceed the capacity)
constexpr int kHdrKill = 8;          // count[kHdrKill + level - 1]: entries of that level's kill list (k_nms_level_roots)
constexpr int kHdrDensity = 12;      // count[kHdrDensity]: sum over the grid's cells of (registrations in the cell)^2

// candidate-list capacity: ~8.5 upper-triangle AABB candidates per box for scattered boxes, half the
// cluster size for detector output (clusters of overlapping boxes around every object); 512 per box = 0.6 GB at 100 k
// boxes, small sets get the full triangle.  D3D_NMS_CAND_CAP overrides (tests).
static unsigned long long nms_cand_capacity(int64_t n)
{
    const unsigned long long tri = (unsigned long long)n * (unsigned long long)(n > 0 ? n - 1 : 0) / 2 + 1;
    return std::min<unsigned long long>(tri, std::max<unsigned long long>(512ull * (unsigned long long)n, 1ull << 24));
}


// Scan across the workgroups of ONE launch: every workgroup publishes its total with a ready bit (tot[] zeroed by an earlier
// kernel) and adds up the totals of the workgroups before it, polling the ones not there yet.  "Before" is the order of
// scan_ticket(), not blockIdx.x -- the dispatch order of workgroups is undefined: a workgroup holding ticket t only waits for
// tickets < t, whose holders are running (they took theirs first) and publish before they wait: always progress, wherever
// and in whatever order the workgroups are placed.  Value and ready bit travel in one agent-scope 64-bit word.
// Returns (in every thread) the sum of the totals of tickets 0 .. id - 1; both contain workgroup barriers.
// The poll is bounded all the same (kChainSpins reads of one word, seconds): a predecessor that never publishes -- only a
// defect elsewhere could cause that -- raises flags->need_sweep (the dense path then recomputes everything from the geometry)
// instead of hanging the GPU.
__device__ __forceinline__ unsigned int scan_ticket(unsigned int *ticket, unsigned int *sid)
{
    if (threadIdx.x == 0) *sid = atomicAdd(ticket, 1u);
    __syncthreads();
    return *sid;
}
// A predecessor holds a lower ticket, i.e. it is running and publishes within microseconds; kChainSpins polls (~0.1 s) without
// its word turning up means something is wrong (a lost store, a preempted queue): the scan GIVES UP -- flags->need_sweep is
// raised (the dense path then recomputes everything), flags->scan_gave_up counts it (d3d_nms2d_status reports both: a caller
// on a shared or profiled GPU can see that a call took seconds for THIS reason and retry) and the returned prefix is void:
// *sbase = ~0, callers must not store anything that depends on it.  After the first give-up the wavefront stops polling altogether (the remaining words
// would each cost another kChainSpins).  withhold (test hook, D3D_NMS_TEST_WITHHOLD): ticket 0 does not publish.
constexpr unsigned int kChainSpins = 1u << 18;
constexpr unsigned long long kChainVoid = ~0ull;
__device__ __forceinline__ unsigned long long chained_prefix(unsigned long long *tot, unsigned int id, unsigned long long total,
                                                             unsigned long long *sbase, NmsFlags *flags, bool withhold = false)
{
    const unsigned int lane = threadIdx.x & 63;
    if (threadIdx.x == 0 && !(withhold && id == 0))
        __hip_atomic_store(&tot[id], (total << 1) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 64) {
        unsigned long long acc = 0;
        bool dead = false;
        for (unsigned int j0 = 0; j0 < id && !dead; j0 += 64) {
            const unsigned int j = j0 + lane;
            bool mine_dead = false;
            if (j < id) {
                unsigned long long t;
                unsigned int spins = 0;
                do { t = __hip_atomic_load(&tot[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (!(t & 1ull) && ++spins < kChainSpins);
                if (!(t & 1ull)) { t = 0; mine_dead = true; }
                acc += t >> 1;
            }
            dead = __any(mine_dead);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) {
            *sbase = dead ? kChainVoid : acc;
            if (dead) { flags->need_sweep = 1; atomicAdd(&flags->scan_gave_up, 1u); }
        }
    }
    __syncthreads();
    return *sbase;
}

constexpr int kGridMax = 256, kGridCells = kGridMax * kGridMax;
constexpr int kGridReg = 8;            // registration capacity per box, on average
constexpr int kGridBoxCells = 1024;    // most cells one box may touch
constexpr float kGridCellScale = 2.f;  // cell size / mean AABB extent (4: scattered boxes 25 -> 37 us, clusters unchanged)
constexpr int kGridPad = 16;           // one cell counter per 64-byte line: atomics on neighbouring words serialise like
                                       // atomics on one word (gridreg 35 -> ? us with 5 k cells packed into 335 lines)
struct NmsGrid { float ox, oy, inv_h; int gx, gy; unsigned int ticket, entries; };
// the levels of the greedy result (k_nms_level_*, below) run when the grid is dense
constexpr int kNmsLevels = 2;             // number of levels (D3D_NMS_ONE_LEVEL: one)
constexpr unsigned long long kNmsLevelDensity = 128;
constexpr int kNmsLevelChunks = 2;      // k_nms_level_block: chunks of 64 partners a wavefront tests before it gives up
__device__ __forceinline__ bool nms_levels_on(const NmsGrid &g, const NmsCand *hdr)
{
    return hdr->count[kHdrDensity] > kNmsLevelDensity * (unsigned long long)g.entries;
}

__device__ __forceinline__ int grid_cell(float x, float o, float inv_h, int g)
{
    const float t = fminf((x - o) * inv_h, (float)(g - 1));
    return t > 0.f ? (int)t : 0;                     // NaN -> 0
}
__device__ __forceinline__ bool grid_valid(const float4 f)
{
    return f.x >= -3.0e38f && f.y >= -3.0e38f && f.z <= 3.0e38f && f.w <= 3.0e38f && f.z >= f.x && f.w >= f.y;   // finite, non-empty
}


// grid parameters from {min x, min y, max x, max y, sum of the mean extents, count} of the valid AABBs
__device__ __forceinline__ NmsGrid make_nms_grid(const float (&v)[6])
{
    NmsGrid g;
    g.ticket = 0; g.entries = 0;
    if (!(v[5] > 0.f)) { g.ox = 0.f; g.oy = 0.f; g.inv_h = 0.f; g.gx = 1; g.gy = 1; }
    else {
        const float rx = v[2] - v[0], ry = v[3] - v[1];
        float h = kGridCellScale * v[4] / v[5];
        h = fmaxf(h, fmaxf(rx, ry) / (float)(kGridMax - 1));
        if (!(h > 0.f)) h = 1.f;
        g.ox = v[0]; g.oy = v[1]; g.inv_h = 1.f / h;
        g.gx = (int)fminf(rx / h, (float)(kGridMax - 1)) + 1;
        g.gy = (int)fminf(ry / h, (float)(kGridMax - 1)) + 1;
    }
    return g;
}
// {min, min, max, max, sum, sum} over the workgroup (256 threads); the result is valid in every thread
__device__ __forceinline__ void block_reduce_extent(float (&v)[6], float (*sm)[6])
{
    for (int o = 32; o > 0; o >>= 1) {
        v[0] = fminf(v[0], __shfl_xor(v[0], o, 64)); v[1] = fminf(v[1], __shfl_xor(v[1], o, 64));
        v[2] = fmaxf(v[2], __shfl_xor(v[2], o, 64)); v[3] = fmaxf(v[3], __shfl_xor(v[3], o, 64));
        v[4] += __shfl_xor(v[4], o, 64); v[5] += __shfl_xor(v[5], o, 64);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) for (int k = 0; k < 6; k++) sm[w][k] = v[k];
    __syncthreads();
    for (int k = 0; k < 6; k++) v[k] = sm[0][k];
    for (int q = 1; q < 4; q++) {
        v[0] = fminf(v[0], sm[q][0]); v[1] = fminf(v[1], sm[q][1]); v[2] = fmaxf(v[2], sm[q][2]); v[3] = fmaxf(v[3], sm[q][3]);
        v[4] += sm[q][4]; v[5] += sm[q][5];
    }
    __syncthreads();
}
constexpr int kGridScanWgs = (kGridCells + 1 + 1023) / 1024;      // workgroups of k_nms_gridscan
constexpr int kGridFoldMax = 512;      // most k_nms_prepare partials that k_nms_gridreg folds itself (else: k_nms_extent)

template <typename T, typename B = T /* element type of boxes / scores in memory: T, or float widened on load (D3D_F32_WIDE) */>
__global__ __launch_bounds__(256) void k_nms_prepare(const B *__restrict__ boxes, const B *__restrict__ scores,
                              const int64_t *__restrict__ order, int64_t n, float score_threshold,
                              BoxCore<T> *geom, float4 *fbox, uint8_t *state, uint32_t *inc_cnt, float *farea,
                              unsigned long long *remv, int64_t nb, NmsFlags *flags, NmsCand *cand_hdr,
                              unsigned int force_dense, int32_t *xkey, unsigned int *grid_ticket, unsigned long long *tile_tot,
                              float *gpartial, uint32_t *cellcnt, unsigned long long *chunk_tot, uint8_t *blocked,
                              unsigned int force_levels)
{
    __shared__ float sm[4][6];
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // sorted position
    bool pre = false;
    // for the uniform-grid broad phase: cleared cell counters, and this workgroup's share of the AABBs' bounding range and
    // mean extent (folded by k_nms_gridreg<count>, or by k_nms_extent when there are more than kGridFoldMax workgroups)
    for (int64_t i = p; i <= kGridCells; i += (int64_t)gridDim.x * blockDim.x) cellcnt[i * kGridPad] = 0;
    float ext[6] = {INFINITY, INFINITY, -INFINITY, -INFINITY, 0.f, 0.f};
    if (p == 0) { flags->need_sweep = force_dense; flags->undecided = 0; flags->scan_gave_up = 0; *grid_ticket = 0; }
    // list counter(s), kill-list counters, density sum (kHdr*; D3D_NMS_FORCE_LEVELS: a density no grid reaches)
    if (p < kNmsListSegs * 16) cand_hdr->count[p] = (p == kHdrDensity && force_levels) ? (1ull << 62) : 0ull;
    if (p * kIncTile < n) tile_tot[p] = 0;                   // k_nms_incscan's ready words
    if (p < kGridScanWgs + 1) chunk_tot[p] = 0;              // ... and k_nms_gridscan's; behind them the two tickets
    if (p < n) {
        const int64_t i = order[p];
        const BoxGeom<T> g = Box2D<T>::load(boxes + i * 5);
        geom[p] = core_of(g);
        const float4 f = make_float4(round_down(g.xmin), round_down(g.ymin), round_up(g.xmax), round_up(g.ymax));
        fbox[p] = f;
        if (grid_valid(f)) { ext[0] = f.x; ext[1] = f.y; ext[2] = f.z; ext[3] = f.w; ext[4] = 0.5f * ((f.z - f.x) + (f.w - f.y)); ext[5] = 1.f; }
        // key for the sweep-and-prune order: the descending sort of ~ordered(xmin) is ascending in xmin
        const int32_t bits = __float_as_int(f.x);
        xkey[p] = ~(bits ^ ((bits >> 31) & 0x7fffffff));
        // nms.cpp:23-29: the tail with score <= threshold is suppressed up front, never position 0
        pre = p > 0 && !((T)scores[i] > (T)score_threshold);
        state[p] = pre ? kSuppressed : kUndecided;
        inc_cnt[p] = 0;
        blocked[p] = 0;                     // set by the broad phase for every box that has a better-ranked candidate partner
        farea[p] = round_down(g.area);      // (lower bound: for the IoU upper bound of the broad phase)
    }
    unsigned long long word = __ballot(pre);
    if ((threadIdx.x & 63) == 0 && (p >> 6) < nb) remv[p >> 6] = word;
    block_reduce_extent(ext, sm);
    if (threadIdx.x == 0)
        for (int k = 0; k < 6; k++) gpartial[(size_t)blockIdx.x * 6 + k] = ext[k];
}

// The pair phase is split so that the conservative AABB tests run in a kernel with a handful of registers and the
// fp64 clipping runs with every lane busy -- and the AABB tests are pruned by a sweep along x:
//   sort + k_nms_xgather   the boxes' AABBs in ascending-xmin order (radix sort of N keys), with their score ranks
//   k_nms_cand   lane = box i of that order, walking the boxes j > i until one starts right of i's xmax (everything
//                after it does too): consecutive lanes read consecutive AABBs, so the loads are coalesced and the
//                N^2/2 tests shrink to N x (boxes within one box length in x).  Survivors (min rank, max rank) go
//                to a per-wavefront LDS batch that is flushed to the global candidate list with one atomic
//   k_nms_hits   one candidate (p, q) per lane: exact IoU, hit -> append p to q's incoming list
// If the candidate list overflows or the fixed point needs too many rounds, need_sweep is set
// and the dense path runs instead (k_nms_pairs, k_nms_sweep -- gated on the flag, no host round trip).
constexpr int kColsPerBlock = 8;
constexpr int kCandLds = 512;        // LDS batch entries per wavefront (4 KiB): one list reservation per 512 candidates

constexpr int kCandUnroll = 8;       // independent loads in flight per lane
constexpr int kCandMaxSplit = 16;    // wavefronts sharing one block of 64 boxes
constexpr int kCandPad = 64;         // sentinel AABBs behind the last block of boxes (a chunk touches 128 entries)

__global__ __launch_bounds__(256) void k_nms_xgather(const float4 *__restrict__ fbox, const int32_t *__restrict__ perm,
                                                     int64_t n, int64_t nb, float4 *__restrict__ fbx,
                                                     uint32_t *__restrict__ rankx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb * 64 + kCandPad) return;
    if (i < n) {
        const int32_t r = perm[i];
        fbx[i] = fbox[r];
        rankx[i] = (uint32_t)r;
    } else {                                      // sentinels: never a candidate, and they end every walk
        fbx[i] = make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
        if (i < nb * 64) rankx[i] = 0xffffffffu;
    }
}

__global__ __launch_bounds__(256) void k_nms_cand(const float4 *__restrict__ fbx, uint32_t nb, uint32_t nsplit,
                                                  unsigned long long *__restrict__ list, unsigned long long cap,
                                                  NmsCand *hdr, NmsFlags *flags)
{
    constexpr int U = kCandUnroll;
    __shared__ unsigned int batch[4][kCandLds];             // (j - first box of the wavefront's block) << 6 | lane
    __shared__ float4 window[4][128];
    __shared__ unsigned int wcnt[4];
    __shared__ unsigned long long bbase;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t wid = blockIdx.x * 4 + wave;
    const uint32_t rb = wid / nsplit;
    const uint32_t split = wid - rb * nsplit;
    const uint32_t i = rb * 64 + lane;            // n < 2^28 (nb <= 65535)
    const float4 fa = rb < nb ? fbx[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned int *q = batch[wave];
    float4 *win = window[wave];
    unsigned int wn = 0;                                                  // wave-uniform fill of the batch
    bool overflow = false;
    const unsigned long long segcap = cap / kNmsListSegs;                 // list segment of this workgroup
    const unsigned int sgi = blockIdx.x % kNmsListSegs;
    unsigned long long *seg = list + sgi * segcap, *counter = &hdr->count[sgi * 16];
    auto write_out = [&](unsigned long long gb) {
        __builtin_amdgcn_wave_barrier();
        for (unsigned int t = lane; t < wn; t += 64)
            if (gb + t < segcap) {
                const unsigned int e = q[t];
                seg[gb + t] = ((unsigned long long)(rb * 64 + (e & 63u)) << 32) | (unsigned long long)(rb * 64 + (e >> 6));
            }
        if (gb + wn > segcap) { flags->need_sweep = 1; overflow = true; } // the dense path takes over: stop early
        wn = 0;
    };
    auto flush = [&]() {                                                  // batch full: the wavefront reserves
        unsigned long long gb = 0;
        if (lane == 0) gb = atomicAdd(counter, (unsigned long long)wn);
        write_out(__shfl(gb, 0, 64));
    };
    // lane i walks j = i + 1, i + 2, ... in chunks of 64 (the wavefronts sharing this block of boxes take every
    // nsplit-th chunk): the 128 AABBs a chunk can touch are staged in LDS with two coalesced loads and each lane reads
    // its own sliding window from there.  The walk ends with the first chunk whose last box starts right of every
    // lane's xmax -- everything after it does too, and so do the sentinels behind the last box.
    for (uint32_t c = split; rb + c < nb; c += nsplit) {
        const uint32_t base = (rb + c) * 64;                              // chunk: j = base + lane + d, d = 1..64
        win[lane] = fbx[base + lane];
        win[64 + lane] = fbx[base + 64 + lane];                           // < nb * 64 + 64: boxes or sentinels
        __builtin_amdgcn_wave_barrier();          // LDS ops of one wavefront complete in order: no s_barrier needed
#pragma unroll 1
        for (int d0 = 1; d0 <= 64; d0 += U) {
            float4 fb[U];
#pragma unroll
            for (int u = 0; u < U; u++) fb[u] = win[lane + d0 + u];
#pragma unroll
            for (int u = 0; u < U; u++) {
                // strict overlap in x and y as ONE number: the smallest of the four gaps must be positive
                const float g = fminf(fminf(fb[u].z - fa.x, fa.z - fb[u].x), fminf(fb[u].w - fa.y, fa.w - fb[u].y));
                const bool cand = g > 0.f;                                // sentinels have g = -inf
                const unsigned long long m = __ballot(cand);
                if (m) {
                    const unsigned int cnt = (unsigned int)__popcll(m);
                    if (wn + cnt > (unsigned int)kCandLds) flush();
                    if (cand)       // x-order indices; k_nms_hits turns them into score ranks
                        q[wn + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = ((base + lane + d0 + u - rb * 64) << 6) | lane;
                    wn += cnt;
                }
            }
        }
        if (__ballot(win[lane + 64].x < fa.z) == 0 || overflow) break;
    }
    // the rest is reserved with ONE atomic per workgroup (atomics on the list counter are serialised, ~7 ns each)
    if (lane == 0) wcnt[wave] = wn;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        bbase = total ? atomicAdd(counter, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    unsigned long long gb = bbase;
    for (uint32_t w = 0; w < wave; w++) gb += wcnt[w];
    if (wn) write_out(gb);
}

// ---- broad phase on a uniform grid (default; the x-sweep above stays selectable with D3D_NMS_BROAD=sweep)
// The sweep along x tests every box against all boxes within one box length in x -- 1 % of 100 k scattered boxes, 100 M
// AABB tests -- and needs the boxes sorted by xmin (a 9-launch merge sort, 48 us).  Here every box is REGISTERED in all
// cells of a uniform grid that its conservative AABB touches (cell = 2 x the mean AABB extent, at most 256 x 256 cells:
// ~2.25 cells per box), the registrations are counting-sorted by cell (atomic count, one-workgroup scan, atomic cursor)
// and a lane per registration walks the rest of its cell's list.  A pair that shares several cells is emitted only in the
// cell holding the lower-left corner of the two AABBs' intersection, so every overlapping pair appears exactly once --
// the same candidate set as the sweep (same strict test on the same rounded AABBs).  Boxes of any size are handled
// (a box touching more than 1024 cells, or more registrations than 8 per box on average, hands over to the dense path).
// folds k_nms_prepare's partial extents into the grid parameters: inside k_nms_gridreg<count> for up to kGridFoldMax
// partials (every workgroup repeats the fold -- 12 KB from L2 -- and workgroup 0 stores the result for the later kernels:
// no launch, no ticket), by this one-workgroup kernel above that
__device__ __forceinline__ NmsGrid fold_extents(const float *__restrict__ partial, unsigned int npart, float (*sm)[6])
{
    float v[6] = {INFINITY, INFINITY, -INFINITY, -INFINITY, 0.f, 0.f};
    for (unsigned int j = threadIdx.x; j < npart; j += 256) {
        const float *q = partial + (size_t)j * 6;
        v[0] = fminf(v[0], q[0]); v[1] = fminf(v[1], q[1]); v[2] = fmaxf(v[2], q[2]); v[3] = fmaxf(v[3], q[3]);
        v[4] += q[4]; v[5] += q[5];
    }
    block_reduce_extent(v, sm);
    return make_nms_grid(v);
}
__global__ __launch_bounds__(256) void k_nms_extent(const float *__restrict__ partial, unsigned int npart, NmsGrid *grid)
{
    __shared__ float sm[4][6];
    const NmsGrid g = fold_extents(partial, npart, sm);
    if (threadIdx.x == 0) *grid = g;
}

// pass 1 (SCATTER = false): count the registrations per cell; pass 2: place them (cursor = scanned counts)
constexpr int kEliteShift = 4;
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_nms_gridreg(const float4 *__restrict__ fbox, int64_t n, NmsGrid *grid, uint32_t *cellcur,
                                                     unsigned long long cap_e, uint32_t *__restrict__ cellbox,
                                                     uint32_t *__restrict__ cellof, float4 *__restrict__ fbc, NmsFlags *flags,
                                                     const float *__restrict__ farea, float *__restrict__ carea,
                                                     const float *__restrict__ partial, unsigned int npart /* 0: *grid is ready */,
                                                     const uint32_t *__restrict__ cellstart, const uint8_t *__restrict__ state,
                                                     uint8_t *__restrict__ regopen, const NmsCand *hdr = nullptr,
                                                     const uint8_t *__restrict__ blocked = nullptr, unsigned int levels = 0,
                                                     const NmsCand *density_hdr = nullptr, int *host_dense = nullptr)
{
    __shared__ float sm[4][6];
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // first placement: the grid's density is final (k_nms_gridscan summed it up) -- tell the host NOW, which is waiting to
    // decide whether the level kernels are worth launching (nms_typed): 2 = dense, 1 = sparse
    if (host_dense && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(host_dense, nms_levels_on(*grid, density_hdr) ? 2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (the word IS the message: nothing to release)
    // second registration (hdr != NULL), after the levels: only the boxes they left open, so that k_nms_cand_grid walks short
    // lists; nothing happens unless the levels ran
    if (hdr && !nms_levels_on(*grid, hdr)) return;
    if (hdr && p < n && !(state[p] == kUndecided && blocked[p] == levels)) return;
    NmsGrid g;
    if (!SCATTER && npart) {
        g = fold_extents(partial, npart, sm);
        if (blockIdx.x == 0 && threadIdx.x == 0) *grid = g;
    } else g = *grid;
    if (p >= n) return;
    const float4 f = fbox[p];
    if (!grid_valid(f)) return;
    const int cx0 = grid_cell(f.x, g.ox, g.inv_h, g.gx), cx1 = grid_cell(f.z, g.ox, g.inv_h, g.gx);
    const int cy0 = grid_cell(f.y, g.oy, g.inv_h, g.gy), cy1 = grid_cell(f.w, g.oy, g.inv_h, g.gy);
    if ((cx1 - cx0 + 1) * (cy1 - cy0 + 1) > kGridBoxCells) { flags->need_sweep = 1; return; }   // a frame-sized box: dense path
    for (int cy = cy0; cy <= cy1; cy++)
        for (int cx = cx0; cx <= cx1; cx++) {
            const uint32_t c = (uint32_t)(cy * g.gx + cx);
            // placement: the best-ranked sixteenth of the boxes (kEliteShift) fills a cell's list from the front, everybody
            // else from the back (a second cursor in the counter's padding).  A box then finds its better-ranked partners
            // early in the list, and one of the front section finds ALL of them there (k_nms_level_block)
            uint32_t pos;
            if (!SCATTER || p < (n >> kEliteShift)) pos = atomicAdd(&cellcur[(size_t)c * kGridPad], 1u);
            else pos = cellstart[c + 1] - 1u - atomicAdd(&cellcur[(size_t)c * kGridPad + 1], 1u);
            if (SCATTER && pos < cap_e) {
                cellbox[pos] = (uint32_t)p; cellof[pos] = c; fbc[pos] = f; carea[pos] = farea[p];
                regopen[pos] = state[p] == kUndecided;      // (the score threshold's verdict: for level 1 of k_nms_level_block)
            }
        }
}

// cellcnt[0 .. cells) -> exclusive offsets in cellstart[0 .. cells] and in the cursors: 1024 cells per workgroup (the padded
// counters are 64 KB of strided reads per workgroup; ONE workgroup reading all 4 MB took 11 us) + chained_prefix across
// the workgroups (chunk_tot[] zeroed by k_nms_prepare)
__global__ __launch_bounds__(1024) void k_nms_gridscan(uint32_t *cellcur, uint32_t *cellstart, NmsGrid *grid, unsigned long long cap_e,
                                                       NmsFlags *flags, unsigned long long *chunk_tot, unsigned int *ticket,
                                                       bool withhold, unsigned long long *density, const NmsCand *gate = nullptr)
{
    __shared__ unsigned long long smem[1024 / kWave], sbase;
    __shared__ unsigned int sid;
    // (second registration: runs only when the levels did.  Workgroups of this launch raise the density and lower the entries
    // while others still test them -- both move the test further towards "on")
    if (gate && !nms_levels_on(*grid, gate)) return;
    const int chunk = (int)scan_ticket(ticket, &sid);
    const int cells = grid->gx * grid->gy;                                      // entry `cells` = the total
    if (chunk * 1024 > cells) return;                                           // (nobody waits for a higher chunk)
    const int c = chunk * 1024 + threadIdx.x;
    const uint32_t x = c < cells ? cellcur[(size_t)c * kGridPad] : 0u;
    unsigned long long total;
    const unsigned long long ex = block_excl_scan_u64<1024>(x, &total, smem);
    {   // sum of squared list lengths: registrations x partners, what a walk over all pairs of the cells costs
        unsigned long long sq = (unsigned long long)x * x;
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, kWave);
        if ((threadIdx.x & (kWave - 1)) == 0 && sq) atomicAdd(density, sq);
    }
    const unsigned long long before = chained_prefix(chunk_tot, (unsigned int)chunk, total, &sbase, flags, withhold);
    if (before == kChainVoid) {                             // gave up: the dense path takes over; leave an EMPTY grid behind so that
        if (c <= cells) { cellstart[c] = 0; cellcur[(size_t)c * kGridPad] = 0; }      // the launches in between find nothing to do
        if (c == cells) grid->entries = 0;
        return;
    }
    if (c <= cells) {
        cellstart[c] = (uint32_t)(before + ex);
        cellcur[(size_t)c * kGridPad] = (uint32_t)(before + ex);
        cellcur[(size_t)c * kGridPad + 1] = 0;              // the cell's back cursor (k_nms_gridreg<place>)
    }
    if (c == cells) {
        const unsigned long long tot = before + ex;
        grid->entries = (unsigned int)(tot < cap_e ? tot : cap_e);
        if (tot > cap_e) flags->need_sweep = 1;
    }
}

// lane = one registration e (box a in cell c): walk the registrations behind it in the same cell
__global__ __launch_bounds__(256) void k_nms_cand_grid(const float4 *__restrict__ fbc, const uint32_t *__restrict__ cellof,
                                                       const uint32_t *__restrict__ cellstart, const NmsGrid *grid,
                                                       unsigned long long *__restrict__ list, unsigned long long cap, NmsCand *hdr,
                                                       NmsFlags *flags, const float *__restrict__ carea, float thr,
                                                       const uint32_t *__restrict__ cellbox, const uint8_t *__restrict__ blocked,
                                                       const uint8_t *__restrict__ state, unsigned int levels /* that were launched */)
{
    __shared__ unsigned long long batch[4][kCandLds];
    __shared__ float4 window[4][128 + 4];
    __shared__ float awindow[4][128 + 4];
    __shared__ unsigned int wcnt[4];
    __shared__ unsigned long long bbase;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const NmsGrid g = *grid;
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    // only the boxes the levels left open (undecided and blocked at the last level; level 0 = everybody when they did not
    // run) take part, on either side
    const unsigned int level = nms_levels_on(g, hdr) ? levels : 0u;
    auto open = [&](uint32_t reg) {
        const uint32_t r = cellbox[reg];
        return state[r] == kUndecided && blocked[r] == level;
    };
    const bool live = e < g.entries && open(e);
    if (__syncthreads_or(live) == 0) return;
    const float4 fa = live ? fbc[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t c = live ? cellof[e] : 0u;
    const uint32_t end = live ? cellstart[c + 1] : 0u;
    const float aa = live ? carea[e] : 0.f;
    // A pair whose IoU cannot exceed the threshold is dropped here already: the intersection is at most the overlap of the
    // (outward-rounded) AABBs and at most either area, the areas are rounded down, and the margin is far above the fp32
    // rounding of this test and of the exact IoU in either precision.  For scattered boxes most AABB overlaps are slivers
    // (config 3: 866 k overlapping pairs, 135 k left), and each one dropped saves the narrow phase two ~100-byte geometry
    // gathers.  thr < 0 or NaN: nothing is dropped.
    const bool bound_on = thr >= 0.f;
    const float thr_lhs = 1.f + thr, thr_rhs = thr * (1.f - 1e-4f);
    const int ccx = (int)(c % (uint32_t)g.gx), ccy = (int)(c / (uint32_t)g.gx);
    unsigned long long *q = batch[wave];
    unsigned int wn = 0;
    bool overflow = false;
    unsigned long long *counter = &hdr->count[0];
    auto write_out = [&](unsigned long long gb) {
        __builtin_amdgcn_wave_barrier();
        for (unsigned int t = lane; t < wn; t += 64)
            if (gb + t < cap) list[gb + t] = q[t];
        if (gb + wn > cap) { flags->need_sweep = 1; overflow = true; }
        wn = 0;
    };
    auto flush = [&]() {
        unsigned long long gb = 0;
        if (lane == 0) gb = atomicAdd(counter, (unsigned long long)wn);
        write_out(__shfl(gb, 0, 64));
    };
    // lane e walks e + 1, e + 2, ... up to the end of its cell's list, in chunks of 64: the 128 registrations a chunk can
    // touch are staged in LDS with two coalesced loads (consecutive lanes = consecutive registrations)
    float4 *win = window[wave];
    float *awin = awindow[wave];
    const uint32_t e0 = blockIdx.x * 256 + wave * 64, nent = g.entries;
    for (uint32_t ch = 0; !overflow; ch++) {
        const uint32_t base = e0 + ch * 64;                               // partner t = base + lane + d, d = 1..64
        if (__ballot(live && base + lane + 1 < end) == 0) break;
        const bool in0 = base + lane < nent && open(base + lane), in1 = base + 64 + lane < nent && open(base + 64 + lane);
        win[lane] = in0 ? fbc[base + lane] : make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
        win[64 + lane] = in1 ? fbc[base + 64 + lane] : make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
        awin[lane] = in0 ? carea[base + lane] : 0.f;
        awin[64 + lane] = in1 ? carea[base + 64 + lane] : 0.f;
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int d0 = 1; d0 <= 64; d0 += 4) {
            if (__ballot(live && base + lane + d0 < end) == 0) break;
            float4 fb[4];
#pragma unroll
            for (int u = 0; u < 4; u++) fb[u] = win[lane + d0 + u];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t t = base + lane + d0 + u;
                const float gap = fminf(fminf(fb[u].z - fa.x, fa.z - fb[u].x), fminf(fb[u].w - fa.y, fa.w - fb[u].y));
                bool cand = live && t < end && gap > 0.f;
                if (cand)       // report the pair only in the cell of the intersection's lower-left corner
                    cand = grid_cell(fmaxf(fa.x, fb[u].x), g.ox, g.inv_h, g.gx) == ccx &&
                           grid_cell(fmaxf(fa.y, fb[u].y), g.oy, g.inv_h, g.gy) == ccy;
                if (cand && bound_on) {
                    const float ab = awin[lane + d0 + u];
                    const float ix = fminf(fa.z, fb[u].z) - fmaxf(fa.x, fb[u].x), iy = fminf(fa.w, fb[u].w) - fmaxf(fa.y, fb[u].y);
                    const float iub = fminf(ix * iy, fminf(aa, ab));
                    cand = !(iub * thr_lhs < thr_rhs * (aa + ab));
                }
                const unsigned long long m = __ballot(cand);
                if (m) {
                    const unsigned int cnt = (unsigned int)__popcll(m);
                    if (wn + cnt > (unsigned int)kCandLds) flush();
                    if (cand) q[wn + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)e << 32) | (unsigned long long)t;
                    wn += cnt;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) wcnt[wave] = wn;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        bbase = total ? atomicAdd(counter, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    unsigned long long gb = bbase;
    for (uint32_t w = 0; w < wave; w++) gb += wcnt[w];
    if (wn) write_out(gb);
}

// narrow phase, pass A: exact IoU of one candidate per lane.  The entry is rewritten as (hit << 63 | p << 32 | q) in score
// ranks (p suppresses q, p < q) and the hits of every box are counted; the scan of the counts (k_nms_incscan) and pass B
// (k_nms_fill) then lay the hitters of every box out contiguously -- lists of any length, so clusters of hundreds of
// overlapping detections stay on this path.
constexpr unsigned long long kHitBit = 1ull << 63;

// exact test of one candidate pair (score ranks p < q)
template <typename T, bool ROTATED>
__device__ __forceinline__ bool nms_pair_hits(const BoxCore<T> *__restrict__ geom, uint32_t p, uint32_t q, T thr)
{
    const BoxGeom<T> a = expand(geom[p]), b = expand(geom[q]);           // one sector per box
    // the intersection is at most the overlap of the AABBs and at most either area: when even that bound gives
    // IoU <= thr (with a margin far above the rounding of either side) the clip is not needed
    const T ix = fmin(a.xmax, b.xmax) - fmax(a.xmin, b.xmin), iy = fmin(a.ymax, b.ymax) - fmax(a.ymin, b.ymin);
    const T iub = fmin(ix * iy, fmin(a.area, b.area));
    const T margin = sizeof(T) == 8 ? (T)1e-9 : (T)1e-4;
    if (ROTATED && thr >= 0 && iub * (1 + thr) < thr * (a.area + b.area) * (1 - margin)) return false;
    const T v = ROTATED ? iou_rbox(a, b) : iou_aabb(a, b);
    return v > thr;                                         // nms.cpp:53  iou > (scalar_t)(float)iou_threshold
}

// between the levels and the second registration: the cell counters and the cell scan's ready words / ticket start over
__global__ __launch_bounds__(256) void k_nms_regrid_reset(uint32_t *cellcnt, const NmsGrid *grid, const NmsCand *hdr,
                                                          unsigned long long *chunk_tot, unsigned int *ticket)
{
    if (!nms_levels_on(*grid, hdr)) return;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = p; i <= kGridCells; i += (int64_t)gridDim.x * blockDim.x) cellcnt[i * kGridPad] = 0;
    if (p < kGridScanWgs + 1) chunk_tot[p] = 0;
    if (p == 0) *ticket = 0;
}

// ---- levels of the greedy result on the grid, BEFORE any pair is listed
// A detector's raw output is clusters of hundreds of boxes around every object: listing all overlapping pairs of a cluster
// (200 objects x 500 boxes: 25 M pairs = 200 MB of list, one list reservation per 512 of them, one returning atomic per
// hit) costs milliseconds, although the greedy result is decided by a handful of boxes per cluster.  Level i, three launches
// over the cell-sorted registrations:
//   k_nms_level_block   a box that is still open (undecided, blocked at level i - 1) looks for ANY open better-ranked partner
//                       in its cells whose conservative test (AABB overlap + IoU upper bound) passes; if there is one -- or
//                       the search is cut short -- the
//                       box is "blocked" at level i (blocked[rank] = i).  An open box that is NOT blocked has only decided
//                       partners before it: the suppressed ones do not matter, the kept ones were roots of an earlier level
//                       and did not hit it (next kernels) -- it is KEPT: a root of level i.
//   k_nms_level_roots   every root of level i walks its cells' lists, 64 partners per step across the lanes, and lists the
//                       blocked boxes it may hit: (root, victim) pairs, about one per box of a cluster;
//   k_nms_level_kill    one listed pair per lane: exact IoU, a hit suppresses the victim for good (nms.cpp:36-43: a kept box
//                       suppresses what it overlaps, a suppressed box nobody).
// A root stays `undecided` with blocked != last level: it gets no list entries and k_nms_resolve keeps it.  After kNmsLevels
// levels only the boxes blocked at the last level are still open; k_nms_cand_grid lists the pairs among THOSE.
// The levels pay on clusters and cost on scattered boxes (where nearly every box is a root with a list of its own to walk):
// they run when the grid is dense -- sum of squared list lengths > kNmsLevelDensity x registrations, i.e. a registration has
// that many partners on average -- decided on the device (k_nms_gridscan adds the squares up), every kernel checks it first.
// k_nms_level_block's walk: a wavefront owns 64 consecutive registrations (sorted by cell) and walks the union of their
// cells' lists in chunks of 64 staged in LDS; the partner of a step is wave-uniform (LDS broadcast), only the open partners
// of a chunk (ballot) are visited, four per trip, and the wavefront stops as soon as none of its lanes needs more.  The
// lists hold the best-ranked sixteenth of all boxes first (k_nms_gridreg<place>): in a cluster nearly every box finds a
// blocker among the first entries, and a box OF that sixteenth only walks the front section (whoever ranks before it is there).

__global__ __launch_bounds__(256) void k_nms_level_block(const float4 *__restrict__ fbc, const uint32_t *__restrict__ cellof,
                                                         const uint32_t *__restrict__ cellstart, const uint32_t *__restrict__ cellcur,
                                                         const NmsGrid *grid, const NmsCand *hdr, const float *__restrict__ carea,
                                                         float thr, const uint32_t *__restrict__ cellbox, uint8_t *blocked,
                                                         const uint8_t *__restrict__ state, unsigned int level, uint32_t elite,
                                                         const uint8_t *__restrict__ regopen /* level 1: open flag per registration */)
{
    __shared__ float4 window[4][64];
    __shared__ uint32_t rwindow[4][64];
    __shared__ float awindow[4][64];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const NmsGrid g = *grid;
    if (!nms_levels_on(g, hdr)) return;
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    const bool live = e < g.entries;
    if (__ballot(live) == 0) return;
    const float4 fa = live ? fbc[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t c = live ? cellof[e] : 0u, ra = live ? cellbox[e] : 0xffffffffu;
    const float aa = live ? carea[e] : 0.f;
    const bool bound_on = thr >= 0.f;
    const float thr_lhs = 1.f + thr, thr_rhs = thr * (1.f - 1e-4f);
    // open boxes of this level: blocked at the level before.  Level 1: everybody the score threshold left (k_nms_gridreg<place>
    // noted it per registration: no gather by rank in this kernel's dependent chain); later levels look the rank up -- lanes of
    // this launch may already have raised an entry to `level`
    auto is_open = [&](uint32_t reg, uint32_t rank) {
        if (regopen) return regopen[reg] != 0;
        const uint8_t bb = __hip_atomic_load(&blocked[rank], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return state[rank] == kUndecided && bb + 1u >= level;
    };
    const bool mine = live && is_open(e, ra);
    if (__ballot(mine) == 0) return;
    const uint32_t mystart = mine ? cellstart[c] : 0xffffffffu;
    // a box of the front section: everything that ranks before it sits in the front section too
    const uint32_t myend = mine ? (ra < elite ? cellcur[(size_t)c * kGridPad] : cellstart[c + 1]) : 0u;
    // union of the lanes' cell lists (registrations are sorted by cell: the lists of consecutive lanes follow each other)
    uint32_t lo = mystart, hi = myend;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const uint32_t l2 = __shfl_xor(lo, o, kWave), h2 = __shfl_xor(hi, o, kWave);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    float4 *win = window[wave];
    uint32_t *rwin = rwindow[wave];
    float *awin = awindow[wave];
    bool settled = !mine;                       // blocked: nothing more to learn for this lane
    // one chunk ahead: the loads of chunk k + 1 are in flight while chunk k is tested
    float4 nfb = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t nrb = 0xffffffffu;
    float nab = 0.f;
    bool nopen = false;
    auto fetch = [&](uint32_t cb) {
        const uint32_t t = cb + lane;
        nopen = false;
        if (t < hi) {
            nrb = cellbox[t];
            nfb = fbc[t];
            nab = carea[t];
            nopen = is_open(t, nrb);
        }
    };
    fetch(lo);
    int walked = 0;
    for (uint32_t cb = lo; cb < hi;) {
        if (__ballot(!settled && cb < myend) == 0) break;                   // (lists are walked in ascending order)
        if (__ballot(!settled && cb + 64 > mystart && cb < myend) == 0) {   // nobody needs this chunk: on to the next list
            uint32_t nx = (!settled && mystart > cb) ? mystart : 0xffffffffu;   // that a lane still looking walks
#pragma unroll
            for (int o = kWave / 2; o > 0; o >>= 1) { const uint32_t x2 = __shfl_xor(nx, o, kWave); nx = x2 < nx ? x2 : nx; }
            cb = nx;
            if (cb >= hi) break;
            fetch(cb);
            continue;
        }
        const float4 cfb = nfb;
        const uint32_t crb = nrb;
        const float cab = nab;
        const bool partner = nopen;
        // A lane still looking after kNmsLevelChunks chunks OF ITS OWN LISTS gives up and counts as blocked -- always safe: it
        // stays open and its pairs are listed.  The kernel is bound by its vector ALUs (every lane of a wavefront pays for every
        // partner the slowest lane tests: six chunks cost 90 us per launch on 100 k boxes whatever the input, two cost 35), and
        // two chunks are what the roots that matter need: the best box of a cluster sits in the front section (a sixteenth of a
        // list of up to 2048 entries).  What gives up are roots of long lists outside the front section, and the many roots of
        // lists of boxes that merely touch (20 k boxes all over each other: every wavefront holds a few).
        if (!settled && cb + 64 > mystart && cb < myend && walked++ >= kNmsLevelChunks) settled = true;
        if (cb + 64 < hi) fetch(cb + 64);
        // only partners that rank before SOME lane still looking can block anybody: from the second chunk on most lanes are
        // settled and the ones left are the well-ranked ones
        uint32_t wmax = settled ? 0u : ra;
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) { const uint32_t x2 = __shfl_xor(wmax, o, kWave); wmax = x2 > wmax ? x2 : wmax; }
        const bool useful = partner && crb < wmax;
        if (useful) { win[lane] = cfb; rwin[lane] = crb; awin[lane] = cab; }
        unsigned long long pm = __ballot(useful);
        __builtin_amdgcn_wave_barrier();
        while (pm) {
            int j[4], nj = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                j[u] = pm ? __builtin_ctzll(pm) : j[0];
                if (pm) { pm &= pm - 1; nj = u + 1; }
            }
            float4 fb[4];
            uint32_t rbj[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { fb[u] = win[j[u]]; rbj[u] = rwin[j[u]]; }
            bool near[4];
            bool any = false;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t tj = cb + (uint32_t)j[u];
                const float gap = fminf(fminf(fb[u].z - fa.x, fa.z - fb[u].x), fminf(fb[u].w - fa.y, fa.w - fb[u].y));
                near[u] = !settled && u < nj && tj >= mystart && tj < myend && rbj[u] < ra && gap > 0.f;
                any = any || near[u];
            }
            if (!bound_on) settled = settled || any;
            else if (__ballot(any)) {            // the IoU bound only where some lane has an overlapping better-ranked partner
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const float ab = awin[j[u]];
                    const float ix = fminf(fa.z, fb[u].z) - fmaxf(fa.x, fb[u].x), iy = fminf(fa.w, fb[u].w) - fmaxf(fa.y, fb[u].y);
                    const float iub = fminf(ix * iy, fminf(aa, ab));
                    settled = settled || (near[u] && !(iub * thr_lhs < thr_rhs * (aa + ab)));
                }
            }
            if (__ballot(!settled) == 0) break;
        }
        __builtin_amdgcn_wave_barrier();
        cb += 64;
    }
    if (mine && settled) __hip_atomic_store(&blocked[ra], (uint8_t)level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// roots of the level -> (root rank, victim rank) pairs.  A wavefront takes 64 registrations; for each root among them (wave-
// uniform loop) all 64 lanes walk that root's cell list, one partner each per step.  A pair sharing several cells is listed
// in the cell of the lower-left corner of the AABBs' intersection only (as k_nms_cand_grid does).
__global__ __launch_bounds__(256) void k_nms_level_roots(const float4 *__restrict__ fbc, const uint32_t *__restrict__ cellof,
                                                         const uint32_t *__restrict__ cellstart, const NmsGrid *grid, NmsCand *hdr,
                                                         const float *__restrict__ carea, float thr,
                                                         const uint32_t *__restrict__ cellbox, const uint8_t *__restrict__ blocked,
                                                         const uint8_t *__restrict__ state, unsigned int level,
                                                         unsigned long long *__restrict__ list, unsigned long long cap, NmsFlags *flags)
{
    __shared__ unsigned long long batch[4][kCandLds];
    // the roots with LONG lists among the workgroup's 256 registrations (big clusters: a handful): ALL four wavefronts walk such a
    // list together, 256 partners per step -- a wavefront alone walked a 1500-entry list in 24 dependent steps.  Short lists
    // (many roots per workgroup: clusters of a few boxes) stay with the wavefront that holds the root
    __shared__ float4 rt_f[256];
    __shared__ float rt_a[256];
    __shared__ uint32_t rt_r[256], rt_c[256], rt_s0[256], rt_s1[256], nroots;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const NmsGrid g = *grid;
    if (!nms_levels_on(g, hdr)) return;
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    const bool live = e < g.entries;
    const uint32_t ra = live ? cellbox[e] : 0xffffffffu;
    const bool root = live && state[ra] == kUndecided && blocked[ra] + 1u == level;    // open at this level and not blocked in it
    if (threadIdx.x == 0) nroots = 0;
    __syncthreads();
    const float4 fa = root ? fbc[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t c = root ? cellof[e] : 0u;
    const uint32_t mystart = root ? cellstart[c] : 0u, myend = root ? cellstart[c + 1] : 0u;
    const float aa = root ? carea[e] : 0.f;
    const bool longlist = root && myend - mystart > 192u;
    if (longlist) {
        const uint32_t slot = atomicAdd(&nroots, 1u);
        rt_f[slot] = fa; rt_a[slot] = aa; rt_r[slot] = ra; rt_c[slot] = c; rt_s0[slot] = mystart; rt_s1[slot] = myend;
    }
    __syncthreads();
    const uint32_t nr = nroots;
    if (nr == 0 && __ballot(root) == 0) return;
    const bool bound_on = thr >= 0.f;
    const float thr_lhs = 1.f + thr, thr_rhs = thr * (1.f - 1e-4f);
    unsigned long long *q = batch[wave], *counter = &hdr->count[kHdrKill + level - 1];
    unsigned int wn = 0;
    bool overflow = false;
    auto flush = [&]() {
        unsigned long long gb = 0;
        if (lane == 0) gb = atomicAdd(counter, (unsigned long long)wn);
        gb = __shfl(gb, 0, kWave);
        __builtin_amdgcn_wave_barrier();
        for (unsigned int t = lane; t < wn; t += 64)
            if (gb + t < cap) list[gb + t] = q[t];
        if (gb + wn > cap) { flags->need_sweep = 1; overflow = true; }     // the dense path takes over
        wn = 0;
        __builtin_amdgcn_wave_barrier();
    };
    // one root's list: `first` = the caller's first partner index, `stride` = partners all its callers take per step
    auto walk = [&](const float4 fr, const float ar, const uint32_t rr, const uint32_t cr, const uint32_t s0, const uint32_t s1,
                    const uint32_t first, const uint32_t stride) {
        (void)s0;
        const int ccx = (int)(cr % (uint32_t)g.gx), ccy = (int)(cr / (uint32_t)g.gx);
        float4 nfb = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t nrb = 0;
        if (first + lane < s1) { nfb = fbc[first + lane]; nrb = cellbox[first + lane]; }
        for (uint32_t t0 = first; t0 < s1 && !overflow; t0 += stride) {
            const uint32_t t = t0 + lane;
            bool cand = false;
            const float4 fb = nfb;
            const uint32_t rb = nrb;
            if (t + stride < s1) { nfb = fbc[t + stride]; nrb = cellbox[t + stride]; }     // the next step's entries, in flight meanwhile
            if (t < s1) {
                const float gap = fminf(fminf(fb.z - fr.x, fr.z - fb.x), fminf(fb.w - fr.y, fr.w - fb.y));
                cand = rb > rr && gap > 0.f && grid_cell(fmaxf(fr.x, fb.x), g.ox, g.inv_h, g.gx) == ccx &&
                       grid_cell(fmaxf(fr.y, fb.y), g.oy, g.inv_h, g.gy) == ccy;
                if (cand && bound_on) {
                    const float ab = carea[t];
                    const float ix = fminf(fr.z, fb.z) - fmaxf(fr.x, fb.x), iy = fminf(fr.w, fb.w) - fmaxf(fr.y, fb.y);
                    const float iub = fminf(ix * iy, fminf(ar, ab));
                    cand = !(iub * thr_lhs < thr_rhs * (ar + ab));
                }
                if (cand) cand = state[rb] == kUndecided && blocked[rb] == level;      // an open box: blocked at this level
            }
            const unsigned long long m = __ballot(cand);
            if (m) {
                const unsigned int cnt = (unsigned int)__popcll(m);
                if (wn + cnt > (unsigned int)kCandLds) flush();
                if (cand) q[wn + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)rr << 32) | rb;
                wn += cnt;
            }
        }
    };
    // short lists: every wavefront takes the roots among its own 64 registrations (wave-uniform loop, data by shuffles)
    unsigned long long rm = __ballot(root && !longlist);
    while (rm && !overflow) {
        const int j = __builtin_ctzll(rm);
        rm &= rm - 1;
        const float4 fr = make_float4(__shfl(fa.x, j, kWave), __shfl(fa.y, j, kWave), __shfl(fa.z, j, kWave), __shfl(fa.w, j, kWave));
        const uint32_t s0 = __shfl(mystart, j, kWave);
        walk(fr, __shfl(aa, j, kWave), __shfl(ra, j, kWave), __shfl(c, j, kWave), s0, __shfl(myend, j, kWave), s0, 64u);
    }
    // long lists: all four wavefronts walk one together, 256 partners per step
    for (uint32_t ri = 0; ri < nr; ri++)
        walk(rt_f[ri], rt_a[ri], rt_r[ri], rt_c[ri], rt_s0[ri], rt_s1[ri], rt_s0[ri] + wave * 64, 256u);
    if (wn) flush();
}

template <typename T, bool ROTATED>
__global__ __launch_bounds__(256) void k_nms_level_kill(const BoxCore<T> *__restrict__ geom, const unsigned long long *__restrict__ list,
                                                        unsigned long long cap, const NmsGrid *grid, const NmsCand *hdr,
                                                        unsigned int level, T thr, uint8_t *state)
{
    if (!nms_levels_on(*grid, hdr)) return;
    const unsigned long long cnt = hdr->count[kHdrKill + level - 1], total = cnt < cap ? cnt : cap;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const unsigned long long e = list[t];
        const uint32_t p = (uint32_t)(e >> 32), q = (uint32_t)e;
        if (__hip_atomic_load(&state[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kUndecided) continue;   // another root got it
        if (nms_pair_hits<T, ROTATED>(geom, p, q, thr))
            __hip_atomic_store(&state[q], (uint8_t)kSuppressed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T, bool ROTATED>
__global__ __launch_bounds__(256) void k_nms_hits(const BoxCore<T> *__restrict__ geom,
                                                  const uint32_t *__restrict__ rankx,
                                                  unsigned long long *__restrict__ list, unsigned long long cap,
                                                  const NmsCand *hdr, T thr, uint32_t *inc_cnt, uint32_t *__restrict__ arrival)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x, segcap = cap / kNmsListSegs;
    for (int sg = 0; sg < kNmsListSegs; sg++) {
    const unsigned long long cnt = hdr->count[sg * 16], total = cnt < segcap ? cnt : segcap;
    unsigned long long *seg = list + sg * segcap;
    for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const unsigned long long e = seg[t];
        // list entry -> score ranks (through the broad phase's own numbering; the small-set path lists ranks directly)
        const uint32_t r1 = rankx ? rankx[e >> 32] : (uint32_t)(e >> 32), r2 = rankx ? rankx[e & 0xffffffffull] : (uint32_t)e;
        const uint32_t p = r1 < r2 ? r1 : r2, q = r1 < r2 ? r2 : r1;
        const bool hit = nms_pair_hits<T, ROTATED>(geom, p, q, thr);
        seg[t] = (hit ? kHitBit : 0ull) | ((unsigned long long)p << 32) | q;
        if (hit) arrival[sg * segcap + t] = atomicAdd(&inc_cnt[q], 1u);      // position inside q's segment
    }
    }
}

// inc_off = exclusive scan of inc_cnt in ONE launch (the generic count / block-sum / apply trio is three, ~4 us each in a
// stream): a local scan per 1024-box tile + chained_prefix over the tiles (tile_tot[] is zeroed by k_nms_prepare)
__global__ __launch_bounds__(256) void k_nms_incscan(const uint32_t *__restrict__ inc_cnt, int64_t n, uint32_t *__restrict__ inc_off,
                                                     unsigned long long *tile_tot, unsigned int *ticket, NmsFlags *flags, bool withhold)
{
    __shared__ unsigned long long smem[4], sbase;
    __shared__ unsigned int sid;
    const unsigned int tile = scan_ticket(ticket, &sid);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t base = (int64_t)tile * kIncTile + (int64_t)w * 256 + lane;             // wavefront w: 4 rows of 64 boxes
    unsigned long long ex[4], carry = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int64_t i = base + k * 64;
        const unsigned long long v = i < n ? inc_cnt[i] : 0u;
        const unsigned long long incl = wave_incl_scan_u64(v);
        ex[k] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) smem[w] = carry;
    __syncthreads();
    unsigned long long woff = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { if (k < w) woff += smem[k]; total += smem[k]; }
    const unsigned long long before = chained_prefix(tile_tot, tile, total, &sbase, flags, withhold);
    if (before == kChainVoid) return;                       // gave up: nothing that depends on the prefix is stored
    const unsigned long long off = before + woff;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int64_t i = base + k * 64;
        if (i < n) inc_off[i] = (uint32_t)(off + ex[k]);
    }
}

// pass B: the hits into the boxes' segments
__global__ __launch_bounds__(256) void k_nms_fill(const unsigned long long *__restrict__ list, unsigned long long cap,
                                                  const NmsCand *hdr, const uint32_t *__restrict__ inc_off,
                                                  const uint32_t *__restrict__ arrival, uint32_t *__restrict__ inc,
                                                  const NmsFlags *flags)
{
    // (the dense path has taken over -- possibly because k_nms_incscan GAVE UP, and then inc_off holds whatever an earlier call
    // left in the workspace: storing through it wrote out of bounds.  Found in round 4 when the give-up test first ran behind a
    // call of another size; rounds 2-3 were lucky with their workspace history)
    if (flags->need_sweep) return;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x, segcap = cap / kNmsListSegs;
    for (int sg = 0; sg < kNmsListSegs; sg++) {
    const unsigned long long cnt = hdr->count[sg * 16], total = cnt < segcap ? cnt : segcap;
    const unsigned long long *seg = list + sg * segcap;
    for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const unsigned long long e = seg[t];
        if (!(e & kHitBit)) continue;
        const uint32_t q = (uint32_t)e, p = (uint32_t)(e >> 32) & 0x7fffffffu;
        inc[inc_off[q] + arrival[sg * segcap + t]] = p;     // plain scattered store: no second atomic per hit
    }
    }
}

// ---- dense path (runs only when need_sweep is set; every workgroup checks the flag first)
// block = 4 wavefronts = 4 row blocks, swept against kColsPerBlock consecutive column blocks: the same conservative
// AABB sweep, with the exact test inlined on the survivors and the hits written to the dense bit matrix
template <typename T, bool ROTATED>
__global__ __launch_bounds__(256) void k_nms_pairs(const BoxCore<T> *__restrict__ geom, const float4 *__restrict__ fbox,
                                                   int64_t n, int64_t nb, T thr, unsigned long long *__restrict__ mask,
                                                   const NmsFlags *flags)
{
    if (!flags->need_sweep) return;
    const int64_t tx = (nb + kColsPerBlock - 1) / kColsPerBlock, ty = (nb + 3) / 4;
    for (int64_t tile = blockIdx.x; tile < tx * ty; tile += gridDim.x) {
    const int64_t by = tile / tx, bx = tile - by * tx;
    const int64_t cb0 = bx * kColsPerBlock;
    const int64_t rb = by * 4 + (threadIdx.x >> 6);
    if (cb0 + kColsPerBlock <= by * 4) continue;                         // whole tile below the diagonal
    const int lane = threadIdx.x & 63;
    const int64_t p = rb * 64 + lane;
    if (rb >= nb || p >= n) continue;
    const float4 fa = fbox[p];
    for (int cc = 0; cc < kColsPerBlock; cc++) {
        const int64_t cb = cb0 + cc;
        if (cb < rb || cb >= nb) continue;                                  // wave-uniform
        const int64_t q0 = cb * 64;
        const int ncols = (int)((n - q0) < 64 ? (n - q0) : 64);
        const float4 *__restrict__ fc = fbox + q0;
        uint32_t lo = 0, hi = 0;
#pragma unroll 1
        for (int c0 = 0; c0 < 64; c0 += 8) {          // 8 columns (32 SGPRs of AABBs) per trip: no SGPR spills
            uint32_t byte = 0;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const float4 fb = fc[c0 + c];
                const float g = fminf(fminf(fb.z - fa.x, fa.z - fb.x), fminf(fb.w - fa.y, fa.w - fb.y));
                byte |= (g > 0.f ? 1u : 0u) << c;
            }
            if (c0 < 32) lo |= byte << c0; else hi |= byte << (c0 - 32);
        }
        unsigned long long cand = ((unsigned long long)hi << 32) | lo;
        const int cstart = (rb == cb) ? lane + 1 : 0;
        cand &= cstart >= 64 ? 0ull : (~0ull << cstart);
        if (ncols < 64) cand &= (1ull << ncols) - 1ull;
        unsigned long long bits = 0;
        while (cand) {
            const int c = __builtin_ctzll(cand);
            cand &= cand - 1;
            const BoxGeom<T> a = expand(geom[p]), b = expand(geom[q0 + c]);
            T v = ROTATED ? iou_rbox(a, b) : iou_aabb(a, b);
            if (v > thr) bits |= 1ull << c;                // nms.cpp:53  iou > (scalar_t)(float)iou_threshold
        }
        mask[p * nb + cb] = bits;            // every word the sweep reads (upper triangle + diagonal) is written here
    }
    }
}

// The greedy result as a fixed point, in ONE launch: lane = box q, polling the states of the earlier boxes that hit
// it.  q is suppressed as soon as one of them is kept, kept once all of them are suppressed.  Hitters always rank
// before q, so the lowest undecided box can always be decided and every resident wavefront makes progress; the
// poll is bounded (kSpinPasses), and whatever is left -- dependency chains longer than that -- goes to the dense
// sweep.  States cross XCDs, hence the agent-scope atomic loads/stores.
constexpr int kSpinPasses = 4096;
__global__ __launch_bounds__(256) void k_nms_resolve(int64_t n, uint8_t *state, const uint32_t *__restrict__ inc_cnt,
                                                     const uint32_t *__restrict__ inc_off, uint32_t *inc, NmsFlags *flags,
                                                     const int64_t *__restrict__ order, uint8_t *__restrict__ suppressed,
                                                     uint8_t inv /* 1: the KEEP mask (D3D_NMS_KEEP_MASK) */)
{
    if (__hip_atomic_load(&flags->need_sweep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // the list overflowed
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool done = q >= n || state[q] != kUndecided;
    // every box reports its own result as soon as it is final (if the dense path takes over after all, its sweep rewrites
    // every entry): no separate pass over the states
    const int64_t mine_out = q < n ? order[q] : 0;
    if (q < n && done) suppressed[mine_out] = (uint8_t)(state[q] == kSuppressed) ^ inv;
    // hitters [pos, cnt) are still undecided as far as this lane knows: every pass looks at all of them (ONE kept hitter
    // decides, wherever it sits in the list) and moves the ones found suppressed in front of pos
    // Before the list, the box's lowest-ranked hitter alone (one load per pass): in a cluster of detections it is the cluster's
    // best box for almost every member, kept early; while it is undecided the box waits, the list is only walked once it
    // turns out suppressed (see k_nms_resolve_small).
    uint32_t cnt = 0, first = 0;
    uint32_t *mine = nullptr;
    if (!done) {
        cnt = inc_cnt[q]; mine = inc + inc_off[q];
        first = 0xffffffffu;            // (found here, by independent loads: a second atomic per hit on the box's counters
        for (uint32_t e = 0; e < cnt; e++) first = mine[e] < first ? mine[e] : first;   //  doubled k_nms_hits on clusters)
    }
    for (int pass = 0; pass < kSpinPasses; pass++) {
        if (!done) {
            bool hit = false, wait = false;
            if (cnt > 0) {
                const uint8_t s0 = __hip_atomic_load(&state[first], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hit = s0 == kKept;
                wait = s0 == kUndecided;
            }
            if (!hit && !wait) {
                // four entries per step (independent loads); the undecided ones are written back compacted -- the write cursor
                // never passes the read position -- so the list shrinks to what is still open
                uint32_t wr = 0;
                for (uint32_t e = 0; e < cnt && !hit; e += 4) {
                    uint32_t h[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) h[k] = e + k < cnt ? mine[e + k] : 0u;
                    uint8_t sp[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) sp[k] = __hip_atomic_load(&state[h[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (e + k < cnt) {
                            if (sp[k] == kKept) hit = true;
                            else if (sp[k] == kUndecided) mine[wr++] = h[k];
                        }
                }
                if (!hit) cnt = wr;
            }
            if (hit || (!wait && cnt == 0)) {
                __hip_atomic_store(&state[q], (uint8_t)(hit ? kSuppressed : kKept), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                suppressed[mine_out] = (uint8_t)(hit ? 1 : 0) ^ inv;
                done = true;
            }
        }
        if (__ballot(!done) == 0) return;
        __builtin_amdgcn_s_sleep(2);
    }
    if (!done) flags->need_sweep = 1;
}

// one workgroup; remv (nb words) lives in global scratch when it does not fit LDS
constexpr int kSweepThreads = 1024;
constexpr int kSweepLdsWords = 16384;   // 128 KiB of LDS -> up to 1,048,576 boxes on-chip

__global__ __launch_bounds__(kSweepThreads) void k_nms_sweep(const unsigned long long *__restrict__ mask, int64_t n,
                                                             int64_t nb, unsigned long long *remv_g,
                                                             const int64_t *__restrict__ order, const uint8_t *state,
                                                             const NmsFlags *flags, uint8_t *suppressed, uint8_t inv)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long lds[];
    if (!flags->need_sweep) return;      // the fixed point was reached: k_nms_resolve wrote the result
    const bool in_lds = nb <= kSweepLdsWords;
    unsigned long long *remv = in_lds ? lds : remv_g;
    __shared__ unsigned long long keep_word;
    if (in_lds)
        for (int64_t w = threadIdx.x; w < nb; w += kSweepThreads) lds[w] = remv_g[w];
    __syncthreads();
    for (int64_t cb = 0; cb < nb; cb++) {
        const int64_t p0 = cb * 64;
        const int rows = (int)((n - p0) < 64 ? (n - p0) : 64);
        if (threadIdx.x < 64) {
            // wave 0 resolves the diagonal block serially over its 64 rows (lane broadcast)
            const int lane = threadIdx.x;
            unsigned long long diag = lane < rows ? mask[(p0 + lane) * nb + cb] : 0ull;
            unsigned long long R = remv[cb];
            for (int r = 0; r < rows; r++) {
                unsigned long long d = __shfl(diag, r, 64);
                if (!((R >> r) & 1ull)) R |= d;
            }
            if (lane == 0) {
                remv[cb] = R;
                unsigned long long valid = rows == 64 ? ~0ull : ((1ull << rows) - 1ull);
                keep_word = ~R & valid;
            }
        }
        __syncthreads();
        const unsigned long long K = keep_word;
        if (K) {
            // all waves: OR the rows of the kept boxes of this chunk into the words to the right
            for (int64_t w = cb + 1 + threadIdx.x; w < nb; w += kSweepThreads) {
                unsigned long long acc = 0, k = K;
                while (k) {
                    const int r = __builtin_ctzll(k);
                    k &= k - 1;
                    acc |= mask[(p0 + r) * nb + w];
                }
                if (acc) remv[w] |= acc;
            }
        }
        __syncthreads();
    }
    for (int64_t p = threadIdx.x; p < n; p += kSweepThreads)
        suppressed[order[p]] = (uint8_t)((remv[p >> 6] >> (p & 63)) & 1ull) ^ inv;
}

// ---------------------------------------------------------------- small sets (n <= kNmsSmallMax): five launches
// A detector's top-k (1 k - 4 k boxes) is all launch latency on the general path: sort + 13 dependent launches, ~5 us each
// on the stream.  Here:
//   k_nms_small_front    ONE workgroup: the scores are sorted in LDS (lds_sort.hpp; skipped when the caller brings the
//                        order), then everything k_nms_prepare does, in rank order
//   k_nms_cand_all       all pairs p < q of the set against each other (64 rows from LDS x one column per lane): AABB
//                        overlap + the IoU upper bound -> candidate list, in ranks
//   k_nms_hits           (as above) exact IoU per candidate, hit counts
//   k_nms_fill_small     the offsets of the incoming lists = scan of <= 4096 counts, repeated in LDS by every workgroup;
//                        the hits into the lists
//   k_nms_resolve_small  ONE workgroup, the states in LDS: passes over the undecided boxes until none is left (the lowest
//                        undecided box can always be decided, so at most n passes; no spin limit, no dense fallback).
// The candidate list holds the full triangle at these sizes, so it cannot overflow.
constexpr int kNmsSmallMax = 4096;
constexpr int kNmsSmallResolveMax = 1024;     // one box per lane of k_nms_resolve_small; above: k_nms_resolve (state in global
                                              // memory, all CUs: 4 boxes per lane of one workgroup took 106 us at 4 k clustered boxes)

template <typename T, typename B = T>
__global__ __launch_bounds__(1024) void k_nms_small_front(const B *__restrict__ boxes, const B *__restrict__ scores,
                                                          const int64_t *__restrict__ order_in, uint32_t n, float score_threshold,
                                                          int64_t *order_out, BoxCore<T> *geom, float4 *fbox, float *farea,
                                                          uint8_t *state, uint32_t *inc_cnt, NmsFlags *flags, NmsCand *cand_hdr,
                                                          unsigned long long *remv)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char small_lds[];
    typedef typename KeyBits<T>::U U;
    uint32_t npad = kWave;
    while (npad < n) npad <<= 1;
    U *d0 = reinterpret_cast<U *>(small_lds), *d1 = d0 + npad;
    uint32_t *i0 = reinterpret_cast<uint32_t *>(d1 + npad), *i1 = i0 + npad;
    uint32_t *ii = i0;
    if (!order_in) {
        for (uint32_t e = threadIdx.x; e < npad; e += blockDim.x) {
            d0[e] = e < n ? KeyBits<T>::desc((T)scores[e]) : ~(U)0;         // padding sorts behind every real entry and is unique
            i0[e] = e < n ? e : 0x80000000u + e;
        }
        __syncthreads();
        U *d;
        if (npad <= 1024) sort_lds<1>(d0, i0, d1, i1, (int)npad, &d, &ii);          // entries per thread: no idle slots
        else if (npad <= 2048) sort_lds<2>(d0, i0, d1, i1, (int)npad, &d, &ii);
        else sort_lds<kNmsSmallMax / 1024>(d0, i0, d1, i1, (int)npad, &d, &ii);
    }
    if (threadIdx.x == 0) { flags->need_sweep = 0; flags->undecided = 0; flags->scan_gave_up = 0; cand_hdr->count[0] = 0; }
    for (uint32_t p0 = 0; p0 < n; p0 += blockDim.x) {                // (wave-uniform bound: the ballot below)
        const uint32_t p = p0 + threadIdx.x;
        bool pre = false;
        if (p < n) {
            const int64_t i = order_in ? order_in[p] : (int64_t)ii[p];
            if (!order_in) order_out[p] = i;
            const BoxGeom<T> g = Box2D<T>::load(boxes + i * 5);
            geom[p] = core_of(g);
            fbox[p] = make_float4(round_down(g.xmin), round_down(g.ymin), round_up(g.xmax), round_up(g.ymax));
            farea[p] = round_down(g.area);
            // nms.cpp:23-29: the tail with score <= threshold is suppressed up front, never position 0
            pre = p > 0 && !((T)scores[i] > (T)score_threshold);
            state[p] = pre ? kSuppressed : kUndecided;
            inc_cnt[p] = 0;
        }
        const unsigned long long word = __ballot(pre);               // (for the dense fallback's sweep)
        if ((threadIdx.x & 63) == 0 && p < n) remv[p >> 6] = word;
    }
}

__global__ __launch_bounds__(256) void k_nms_cand_all(const float4 *__restrict__ fbox, const float *__restrict__ farea, uint32_t n,
                                                      float thr, unsigned long long *__restrict__ list, unsigned long long cap,
                                                      NmsCand *hdr, NmsFlags *flags)
{
    __shared__ unsigned long long batch[4][kCandLds];
    __shared__ float4 rows[kWave];
    __shared__ float rarea[kWave];
    __shared__ unsigned int wcnt[4];
    __shared__ unsigned long long bbase;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t p0 = blockIdx.y * kWave, q = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 + 255 <= p0) return;                       // every column of the tile is at or before its first row
    if (threadIdx.x < kWave) {
        const uint32_t p = p0 + threadIdx.x;
        rows[threadIdx.x] = p < n ? fbox[p] : make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
        rarea[threadIdx.x] = p < n ? farea[p] : 0.f;
    }
    __syncthreads();
    const bool live = q < n;
    const float4 fb = live ? fbox[q] : make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
    const float ab = live ? farea[q] : 0.f;
    const bool bound_on = thr >= 0.f;                               // (see k_nms_cand_grid)
    const float thr_lhs = 1.f + thr, thr_rhs = thr * (1.f - 1e-4f);
    unsigned long long *qb = batch[wave];
    unsigned int wn = 0;
    unsigned long long *counter = &hdr->count[0];
    auto write_out = [&](unsigned long long gb) {
        __builtin_amdgcn_wave_barrier();
        for (unsigned int t = lane; t < wn; t += 64)
            if (gb + t < cap) list[gb + t] = qb[t];
        if (gb + wn > cap) flags->need_sweep = 1;                   // (cannot happen: the list holds the full triangle)
        wn = 0;
    };
    for (uint32_t r = 0; r < kWave; r++) {
        const uint32_t p = p0 + r;
        const float4 fa = rows[r];
        const float gap = fminf(fminf(fb.z - fa.x, fa.z - fb.x), fminf(fb.w - fa.y, fa.w - fb.y));
        bool cand = live && p < q && gap > 0.f;                     // (rows past n have empty AABBs)
        if (cand && bound_on) {
            const float aa = rarea[r];
            const float ix = fminf(fa.z, fb.z) - fmaxf(fa.x, fb.x), iy = fminf(fa.w, fb.w) - fmaxf(fa.y, fb.y);
            const float iub = fminf(ix * iy, fminf(aa, ab));
            cand = !(iub * thr_lhs < thr_rhs * (aa + ab));
        }
        const unsigned long long m = __ballot(cand);
        if (m) {
            const unsigned int cnt = (unsigned int)__popcll(m);
            if (wn + cnt > (unsigned int)kCandLds) {
                unsigned long long gb = 0;
                if (lane == 0) gb = atomicAdd(counter, (unsigned long long)wn);
                write_out(__shfl(gb, 0, 64));
            }
            if (cand) qb[wn + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)p << 32) | (unsigned long long)q;
            wn += cnt;
        }
    }
    if (lane == 0) wcnt[wave] = wn;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        bbase = total ? atomicAdd(counter, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    unsigned long long gb = bbase;
    for (uint32_t w = 0; w < wave; w++) gb += wcnt[w];
    if (wn) write_out(gb);
}

__global__ __launch_bounds__(1024) void k_nms_fill_small(const unsigned long long *__restrict__ list, unsigned long long cap,
                                                         const NmsCand *hdr, const uint32_t *__restrict__ inc_cnt, uint32_t n,
                                                         uint32_t *__restrict__ inc_off, const uint32_t *__restrict__ arrival,
                                                         uint32_t *__restrict__ inc)
{
    __shared__ uint32_t off[kNmsSmallMax];
    __shared__ unsigned long long smem[1024 / kWave];
    uint32_t c[4];
    unsigned long long mine = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t i = threadIdx.x * 4 + k; c[k] = i < n ? inc_cnt[i] : 0u; mine += c[k]; }
    unsigned long long total;
    unsigned long long ex = block_excl_scan_u64<1024>(mine, &total, smem);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t i = threadIdx.x * 4 + k;
        off[i] = (uint32_t)ex;
        if (blockIdx.x == 0 && i < n) inc_off[i] = (uint32_t)ex;     // for k_nms_resolve_small
        ex += c[k];
    }
    __syncthreads();
    const unsigned long long cnt = hdr->count[0], tot = cnt < cap ? cnt : cap;
    for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long e = list[t];
        if (!(e & kHitBit)) continue;
        const uint32_t q = (uint32_t)e, p = (uint32_t)(e >> 32) & 0x7fffffffu;
        inc[off[q] + arrival[t]] = p;
    }
}

__global__ __launch_bounds__(1024) void k_nms_resolve_small(uint32_t n, const uint8_t *__restrict__ state0,
                                                            const uint32_t *__restrict__ inc_cnt, const uint32_t *__restrict__ inc_off,
                                                            uint32_t *inc, const int64_t *__restrict__ order,
                                                            uint8_t *__restrict__ suppressed, uint8_t inv)
{
    __shared__ uint8_t st_lds[kNmsSmallResolveMax];
    volatile uint8_t *st = st_lds;
    constexpr int PER = kNmsSmallResolveMax / 1024;
    uint32_t cnt[PER];
    uint32_t *mine[PER], first[PER];
    bool done[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const uint32_t q = threadIdx.x + u * 1024;
        done[u] = true; cnt[u] = 0; mine[u] = nullptr; first[u] = 0;
        if (q < n) {
            const uint8_t s0 = state0[q];
            st[q] = s0;
            if (s0 == kUndecided) {
                done[u] = false; cnt[u] = inc_cnt[q]; mine[u] = inc + inc_off[q];
                first[u] = 0xffffffffu;
                for (uint32_t e = 0; e < cnt[u]; e++) first[u] = mine[u][e] < first[u] ? mine[u][e] : first[u];
            }
        }
    }
    __syncthreads();
    // Per pass and undecided box: FIRST its lowest-ranked hitter (one LDS read).  In a cluster of detections around an object
    // that is the cluster's best box for almost every member: kept in the first pass, so the members are suppressed in the
    // second without their lists being touched.  While it is undecided the box just waits (it is the hitter decided earliest);
    // only when it turns out suppressed is the list walked: ONE kept hitter decides, wherever it sits; the undecided entries
    // are written back compacted, so the list shrinks to what is still open.  A decision only uses final states, so reading a
    // neighbour's
    // state a pass late delays it and nothing else; the lowest undecided box can always be decided: at most n passes.
    for (uint32_t pass = 0; pass <= n; pass++) {               // (n passes always suffice; the bound only rules out a hang)
        int any = 0;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            if (done[u]) continue;
            const uint32_t q = threadIdx.x + u * 1024;
            bool hit = false, wait = false;
            if (cnt[u] > 0) {
                const uint8_t s0 = st[first[u]];
                hit = s0 == kKept;
                wait = s0 == kUndecided;
            }
            if (!hit && !wait) {
                uint32_t wr = 0;                                     // four entries per step, the undecided ones compacted
                for (uint32_t e = 0; e < cnt[u] && !hit; e += 4) {
                    uint32_t h[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) h[k] = e + k < cnt[u] ? mine[u][e + k] : 0u;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (e + k < cnt[u]) {
                            const uint8_t sp = st[h[k]];
                            if (sp == kKept) hit = true;
                            else if (sp == kUndecided) mine[u][wr++] = h[k];
                        }
                }
                if (!hit) cnt[u] = wr;
            }
            if (hit || (!wait && cnt[u] == 0)) { st[q] = hit ? kSuppressed : kKept; done[u] = true; }
            else any = 1;
        }
        if (!__syncthreads_or(any)) break;
    }
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const uint32_t q = threadIdx.x + u * 1024;
        if (q < n) suppressed[order[q]] = (uint8_t)(st[q] == kSuppressed) ^ inv;
    }
}


// Whether the level kernels are worth LAUNCHING (they decide on the device whether to run, but on scattered boxes their ten
// empty launches cost ~15 % of a 100 k-box call) is decided from THIS call's grid, not from a guess: the density is final when
// k_nms_gridscan ends; the first thread of the next launch (the placement, 18 us of work) stores the verdict in a host-mapped
// word of the CALLER's, and the host -- about six launches ahead of the GPU at that point -- waits for it before it enqueues
// the rest.  The queue never runs dry (the placement covers the host's reaction), sparse calls launch nothing for the levels,
// clustered calls always get them: no history, no first-call cliff, nothing shared between callers.  (Rounds 2-3 kept a
// process-wide streak counter + one mapped word for all devices, streams and threads; gone.)  Without the word -- plain
// d3d_nms2d, or a stream under capture -- the levels are always launched.
// The wait is bounded by WALL-CLOCK time (an iteration count depends on the host's speed, and with a backlog on the
// stream -- a model's forward pass in front of the NMS -- the host would burn a core for all of it): after kNmsWaitUs the verdict
// counts as "dense" and the ten level launches are enqueued unasked (~20 us of empty launches when the grid turns out sparse;
// every level kernel tests the density on the device, so the keep mask does not depend on what the host saw).  A verdict that
// arrives late lands in the caller's word, which the next call re-arms before it launches anything.
constexpr long kNmsWaitUs = 400;
static bool nms_wait_dense(volatile int *host_word, hipStream_t st)
{
    (void)st;
    const auto t0 = std::chrono::steady_clock::now();
    for (long spins = 1; *host_word == 0; spins++)
        if ((spins & 255) == 0 &&
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > kNmsWaitUs)
            return true;
    return *host_word != 1;                                 // 2 = dense; unknown counts as dense
}

// the small-set path takes hard NMS of up to kNmsSmallMax boxes unless a flag asks for a specific general path (tests)
static inline bool nms_small_eligible(int64_t n, uint32_t opts)
{
    return n <= kNmsSmallMax && !(opts & (D3D_NMS_BROAD_SWEEP | D3D_NMS_FORCE_DENSE | D3D_NMS_GENERAL | D3D_NMS_FORCE_LEVELS)) && (opts >> 8) == 0;
}

template <typename T, typename B = T>
int nms_typed(const B *boxes, const B *scores, const int64_t *order, int64_t n, int iou_type, float iou_thr,
              float score_thr, uint8_t *suppressed, void *ws, size_t ws_bytes, hipStream_t st, uint32_t opts, int64_t *order_ws,
              int32_t *host_word)
{
    const int64_t nb = d3d_divup(n, 64);
    const uint8_t inv = (opts & D3D_NMS_KEEP_MASK) ? 1 : 0;        // the mask comes out inverted: what box2d_nms returns
    WsCarver w(ws, ws_bytes);
    NmsFlags *flags = w.take<NmsFlags>(1);                 // at offset 0: d3d_nms2d_status reads it there
    BoxCore<T> *geom = w.take<BoxCore<T>>(nb * 64);
    float4 *fbox = w.take<float4>(nb * 64);
    uint8_t *state = w.take<uint8_t>(nb * 64);
    uint8_t *blocked = w.take<uint8_t>(nb * 64);
    uint32_t *inc_cnt = w.take<uint32_t>(nb * 64);
    uint32_t *inc_off = w.take<uint32_t>(nb * 64);
    float *farea = w.take<float>(nb * 64);
    unsigned long long *tile_tot = w.take<unsigned long long>(d3d_divup(nb * 64, kScanTile) + 1 + kGridScanWgs + 1);
    unsigned long long *chunk_tot = tile_tot + d3d_divup(nb * 64, kScanTile) + 1;
    unsigned int *tickets = reinterpret_cast<unsigned int *>(chunk_tot + kGridScanWgs);     // [0] k_nms_incscan, [1] k_nms_gridscan
    (void)w.take<int64_t>(D3D_NUM_COUNTS);
    static_assert(kIncTile == kScanTile, "workspace sized with kScanTile");
    unsigned long long *remv = w.take<unsigned long long>(nb);
    NmsCand *cand_hdr = w.take<NmsCand>(1);
    // the arrays are carved for the automatic capacity (what the workspace query counts); a per-call override
    // (D3D_NMS_CAND_CAP, tests of the overflow -> dense path hand-over) can only shrink what is used of them
    const unsigned long long cap_auto = nms_cand_capacity(n);
    unsigned long long cap = cap_auto;
    if ((opts >> 8) != 0 && (unsigned long long)(opts >> 8) < cap) cap = opts >> 8;
    unsigned long long *cand = w.take<unsigned long long>((size_t)cap_auto);
    uint32_t *inc = w.take<uint32_t>((size_t)cap_auto);                     // hits <= candidates
    uint32_t *arrival = w.take<uint32_t>((size_t)cap_auto);
    int32_t *xkey = w.take<int32_t>(nb * 64);
    int32_t *perm = w.take<int32_t>(nb * 64);
    float4 *fbx = w.take<float4>(nb * 64 + kCandPad);
    uint32_t *rankx = w.take<uint32_t>(nb * 64);
    const size_t sort_bytes = d3d_internal_argsort_i32_bytes(n);
    char *sort_ws = w.take<char>(sort_bytes);
    unsigned long long *mask = w.take<unsigned long long>((size_t)nb * 64 * nb);
    // uniform-grid broad phase
    const unsigned long long cap_e = (unsigned long long)kGridReg * (unsigned long long)(nb * 64);
    uint32_t *cellcur = w.take<uint32_t>((size_t)(kGridCells + 1) * kGridPad);
    uint32_t *cellstart = w.take<uint32_t>(kGridCells + 1);
    float *gpartial = w.take<float>((size_t)d3d_divup(nb * 64, 256) * 6);
    NmsGrid *grid = w.take<NmsGrid>(1);
    uint32_t *cellbox = w.take<uint32_t>((size_t)cap_e);
    uint32_t *cellof = w.take<uint32_t>((size_t)cap_e);
    float4 *fbc = w.take<float4>((size_t)cap_e);
    float *carea = w.take<float>((size_t)cap_e);
    uint8_t *regopen = w.take<uint8_t>((size_t)cap_e);
    if (!ws || !w.ok()) return D3D_ERR_WORKSPACE;
    const bool rot = iou_type == D3D_IOU_RBOX;
    if (nms_small_eligible(n, opts)) {
        typedef typename KeyBits<T>::U U;
        uint32_t npad = kWave;
        while ((int64_t)npad < n) npad <<= 1;
        const size_t lds = order ? 0 : (size_t)npad * (2 * sizeof(U) + 8);
        if (lds > 65536)
            D3D_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nms_small_front<T, B>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int64_t *ord = order ? order : order_ws;
        D3D_LAUNCH("k_nms_small_front", (k_nms_small_front<T, B>), dim3(1), dim3(1024), lds, st, boxes, scores, order, (uint32_t)n, score_thr,
                   order_ws, geom, fbox, farea, state, inc_cnt, flags, cand_hdr, remv);
        D3D_LAUNCH("k_nms_cand_all", k_nms_cand_all, dim3((unsigned)d3d_divup(n, 256), (unsigned)nb), dim3(256), 0, st,
                   (const float4 *)fbox, (const float *)farea, (uint32_t)n, rot ? iou_thr : -1.f, cand, cap, cand_hdr, flags);
        const unsigned hb = (unsigned)std::min<unsigned long long>(d3d_divup((int64_t)cap, 256), 4096);
        if (rot)
            D3D_LAUNCH("k_nms_hits", (k_nms_hits<T, true>), dim3(hb), dim3(256), 0, st, geom, (const uint32_t *)nullptr, cand, cap, cand_hdr,
                       (T)iou_thr, inc_cnt, arrival);
        else
            D3D_LAUNCH("k_nms_hits", (k_nms_hits<T, false>), dim3(hb), dim3(256), 0, st, geom, (const uint32_t *)nullptr, cand, cap, cand_hdr,
                       (T)iou_thr, inc_cnt, arrival);
        D3D_LAUNCH("k_nms_fill_small", k_nms_fill_small, dim3(64), dim3(1024), 0, st, (const unsigned long long *)cand, cap,
                   (const NmsCand *)cand_hdr, (const uint32_t *)inc_cnt, (uint32_t)n, inc_off, (const uint32_t *)arrival, inc);
        if (n <= kNmsSmallResolveMax) {
            D3D_LAUNCH("k_nms_resolve_small", k_nms_resolve_small, dim3(1), dim3(1024), 0, st, (uint32_t)n, (const uint8_t *)state,
                       (const uint32_t *)inc_cnt, (const uint32_t *)inc_off, inc, ord, suppressed, inv);
            return D3D_OK;
        }
        // 1 k - 4 k boxes: the general fixed point (+ its dense fallback for dependency chains beyond the poll limit)
        D3D_LAUNCH("k_nms_resolve", k_nms_resolve, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, n, state,
                   (const uint32_t *)inc_cnt, (const uint32_t *)inc_off, inc, flags, ord, suppressed, inv);
        const unsigned pb = (unsigned)std::min<int64_t>(d3d_divup(nb, kColsPerBlock) * d3d_divup(nb, 4), 8192);
        if (rot)
            D3D_LAUNCH("k_nms_pairs", (k_nms_pairs<T, true>), dim3(pb), dim3(256), 0, st, geom, fbox, n, nb, (T)iou_thr, mask, flags);
        else
            D3D_LAUNCH("k_nms_pairs", (k_nms_pairs<T, false>), dim3(pb), dim3(256), 0, st, geom, fbox, n, nb, (T)iou_thr, mask, flags);
        D3D_LAUNCH("k_nms_sweep", k_nms_sweep, dim3(1), dim3(kSweepThreads), (size_t)nb * 8, st, mask, n, nb, remv, ord, state, flags,
                   suppressed, inv);
        return D3D_OK;
    }
    const bool use_grid = !(opts & D3D_NMS_BROAD_SWEEP);
    const unsigned nbl = (unsigned)d3d_divup(n, 256);
    D3D_LAUNCH("k_nms_prepare", (k_nms_prepare<T, B>), dim3(nbl), dim3(256), 0, st, boxes, scores, order, n, score_thr,
               geom, fbox, state, inc_cnt, farea, remv, nb, flags, cand_hdr, (opts & D3D_NMS_FORCE_DENSE) ? 1u : 0u, xkey, &grid->ticket, tile_tot, gpartial, cellcur, chunk_tot, blocked,
               (opts & D3D_NMS_FORCE_LEVELS) ? 1u : 0u);
    if (use_grid) {
        const bool fold_inline = nbl <= (unsigned)kGridFoldMax;
        if (!fold_inline) D3D_LAUNCH("k_nms_extent", k_nms_extent, dim3(1), dim3(256), 0, st, (const float *)gpartial, nbl, grid);
        D3D_LAUNCH("k_nms_gridreg<count>", k_nms_gridreg<false>, dim3(nbl), dim3(256), 0, st, (const float4 *)fbox, n, grid,
                   cellcur, cap_e, cellbox, cellof, fbc, flags, (const float *)farea, carea, (const float *)gpartial, fold_inline ? nbl : 0u,
                   (const uint32_t *)cellstart, (const uint8_t *)state, regopen);
        D3D_LAUNCH("k_nms_gridscan", k_nms_gridscan, dim3(kGridScanWgs), dim3(1024), 0, st, cellcur, cellstart, grid, cap_e, flags,
                   chunk_tot, tickets + 1, (opts & D3D_NMS_TEST_WITHHOLD) != 0 && (opts & D3D_NMS_BROAD_SWEEP) == 0,
                   &cand_hdr->count[kHdrDensity]);
        int *host_dense = nullptr;
        if (host_word && !(opts & D3D_NMS_FORCE_LEVELS)) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
            else if (cs == hipStreamCaptureStatusNone) {
                void *dptr = nullptr;
                if (hipHostGetDevicePointer(&dptr, host_word, 0) == hipSuccess) host_dense = static_cast<int *>(dptr);
                else (void)hipGetLastError();
            }
        }
        if (host_dense) *reinterpret_cast<volatile int32_t *>(host_word) = 0;
        D3D_LAUNCH("k_nms_gridreg<place>", k_nms_gridreg<true>, dim3(nbl), dim3(256), 0, st, (const float4 *)fbox, n, grid,
                   cellcur, cap_e, cellbox, cellof, fbc, flags, (const float *)farea, carea, (const float *)gpartial, 0u,
                   (const uint32_t *)cellstart, (const uint8_t *)state, regopen, (const NmsCand *)nullptr, (const uint8_t *)nullptr, 0u,
                   (const NmsCand *)cand_hdr, host_dense);
        unsigned int levels = (opts & D3D_NMS_ONE_LEVEL) ? 1u : (unsigned int)kNmsLevels;
        if (host_dense && !nms_wait_dense(host_word, st)) levels = 0;
        for (unsigned int level = 1; level <= levels; level++) {
            const dim3 lg((unsigned)d3d_divup((int64_t)cap_e, 256));
            const float bthr = rot ? iou_thr : -1.f;
            D3D_LAUNCH("k_nms_level_block", k_nms_level_block, lg, dim3(256), 0, st, (const float4 *)fbc, (const uint32_t *)cellof,
                       (const uint32_t *)cellstart, (const uint32_t *)cellcur, (const NmsGrid *)grid, (const NmsCand *)cand_hdr,
                       (const float *)carea, bthr, (const uint32_t *)cellbox, blocked, (const uint8_t *)state, level,
                       (uint32_t)(n >> kEliteShift), level == 1 ? (const uint8_t *)regopen : (const uint8_t *)nullptr);
            D3D_LAUNCH("k_nms_level_roots", k_nms_level_roots, lg, dim3(256), 0, st, (const float4 *)fbc, (const uint32_t *)cellof,
                       (const uint32_t *)cellstart, (const NmsGrid *)grid, cand_hdr, (const float *)carea, bthr,
                       (const uint32_t *)cellbox, (const uint8_t *)blocked, (const uint8_t *)state, level, cand, cap, flags);
            const unsigned kb = (unsigned)std::min<int64_t>(d3d_divup(n, 256) * 2, 4096);
            if (rot)
                D3D_LAUNCH("k_nms_level_kill", (k_nms_level_kill<T, true>), dim3(kb), dim3(256), 0, st, (const BoxCore<T> *)geom,
                           (const unsigned long long *)cand, cap, (const NmsGrid *)grid, (const NmsCand *)cand_hdr, level, (T)iou_thr, state);
            else
                D3D_LAUNCH("k_nms_level_kill", (k_nms_level_kill<T, false>), dim3(kb), dim3(256), 0, st, (const BoxCore<T> *)geom,
                           (const unsigned long long *)cand, cap, (const NmsGrid *)grid, (const NmsCand *)cand_hdr, level, (T)iou_thr, state);
        }
        if (levels > 0) {
            // the boxes the levels left open are registered again, alone: the candidate walk is quadratic in the list lengths
            D3D_LAUNCH("k_nms_regrid_reset", k_nms_regrid_reset, dim3(64), dim3(256), 0, st, cellcur, (const NmsGrid *)grid,
                       (const NmsCand *)cand_hdr, chunk_tot, tickets + 1);
            D3D_LAUNCH("k_nms_gridreg<count>", k_nms_gridreg<false>, dim3(nbl), dim3(256), 0, st, (const float4 *)fbox, n, grid,
                       cellcur, cap_e, cellbox, cellof, fbc, flags, (const float *)farea, carea, (const float *)gpartial, 0u,
                       (const uint32_t *)cellstart, (const uint8_t *)state, regopen, (const NmsCand *)cand_hdr, (const uint8_t *)blocked,
                       levels);
            D3D_LAUNCH("k_nms_gridscan", k_nms_gridscan, dim3(kGridScanWgs), dim3(1024), 0, st, cellcur, cellstart, grid, cap_e, flags,
                       chunk_tot, tickets + 1, false, &cand_hdr->count[kHdrDensity], (const NmsCand *)cand_hdr);
            D3D_LAUNCH("k_nms_gridreg<place>", k_nms_gridreg<true>, dim3(nbl), dim3(256), 0, st, (const float4 *)fbox, n, grid,
                       cellcur, cap_e, cellbox, cellof, fbc, flags, (const float *)farea, carea, (const float *)gpartial, 0u,
                       (const uint32_t *)cellstart, (const uint8_t *)state, regopen, (const NmsCand *)cand_hdr, (const uint8_t *)blocked,
                       levels);
        }
        D3D_LAUNCH("k_nms_cand_grid", k_nms_cand_grid, dim3((unsigned)d3d_divup((int64_t)cap_e, 256)), dim3(256), 0, st,
                   (const float4 *)fbc, (const uint32_t *)cellof, (const uint32_t *)cellstart, (const NmsGrid *)grid, cand, cap,
                   cand_hdr, flags, (const float *)carea, rot ? iou_thr : -1.f, (const uint32_t *)cellbox, (const uint8_t *)blocked,
                   (const uint8_t *)state, levels);
        rankx = cellbox;                                   // registration -> score rank, for k_nms_hits
    } else {
        if (int rc = d3d_internal_argsort_desc_i32(xkey, n, perm, sort_ws, sort_bytes, st)) return rc;
        D3D_LAUNCH("k_nms_xgather", k_nms_xgather, dim3((unsigned)d3d_divup(nb * 64 + kCandPad, 256)), dim3(256), 0, st, fbox,
                   perm, n, nb, fbx, rankx);
        // enough wavefronts to fill the chip even when few row blocks exist: nsplit wavefronts share a row block
        const uint32_t nsplit = (uint32_t)std::min<int64_t>(std::max<int64_t>(8192 / nb, 1), kCandMaxSplit);
        D3D_LAUNCH("k_nms_cand", k_nms_cand, dim3((unsigned)d3d_divup(nb * nsplit, 4)), dim3(256), 0, st, (const float4 *)fbx,
                   (uint32_t)nb, nsplit, cand, cap, cand_hdr, flags);
    }
    const unsigned hits_blocks = (unsigned)std::min<unsigned long long>(d3d_divup((int64_t)cap, 256), 4096);
    if (rot)
        D3D_LAUNCH("k_nms_hits", (k_nms_hits<T, true>), dim3(hits_blocks), dim3(256), 0, st, geom, rankx, cand, cap, cand_hdr,
                   (T)iou_thr, inc_cnt, arrival);
    else
        D3D_LAUNCH("k_nms_hits", (k_nms_hits<T, false>), dim3(hits_blocks), dim3(256), 0, st, geom, rankx, cand, cap, cand_hdr,
                   (T)iou_thr, inc_cnt, arrival);
    D3D_LAUNCH("k_nms_incscan", k_nms_incscan, dim3((unsigned)d3d_divup(n, kIncTile)), dim3(256), 0, st, (const uint32_t *)inc_cnt, n,
               inc_off, tile_tot, tickets, flags, (opts & D3D_NMS_TEST_WITHHOLD) != 0 && (opts & D3D_NMS_BROAD_SWEEP) != 0);
    D3D_LAUNCH("k_nms_fill", k_nms_fill, dim3(hits_blocks), dim3(256), 0, st, (const unsigned long long *)cand, cap,
               (const NmsCand *)cand_hdr, (const uint32_t *)inc_off, (const uint32_t *)arrival, inc, (const NmsFlags *)flags);
    D3D_LAUNCH("k_nms_resolve", k_nms_resolve, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, n, state,
               (const uint32_t *)inc_cnt, (const uint32_t *)inc_off, inc, flags, order, suppressed, inv);
    // dense path, gated on need_sweep inside the kernels
    const unsigned pair_blocks = (unsigned)std::min<int64_t>(d3d_divup(nb, kColsPerBlock) * d3d_divup(nb, 4), 8192);
    if (rot)
        D3D_LAUNCH("k_nms_pairs", (k_nms_pairs<T, true>), dim3(pair_blocks), dim3(256), 0, st, geom, fbox, n, nb, (T)iou_thr,
                   mask, flags);
    else
        D3D_LAUNCH("k_nms_pairs", (k_nms_pairs<T, false>), dim3(pair_blocks), dim3(256), 0, st, geom, fbox, n, nb, (T)iou_thr,
                   mask, flags);
    size_t lds = nb <= kSweepLdsWords ? (size_t)nb * 8 : 0;
    D3D_LAUNCH("k_nms_sweep", k_nms_sweep, dim3(1), dim3(kSweepThreads), lds, st, mask, n, nb, remv, order, state, flags,
               suppressed, inv);
    return D3D_OK;
}

// ---------------------------------------------------------------- soft-NMS (linear / gaussian), nms.cpp:32-95
// Sequential by construction: every kept box rescales the scores of ALL later boxes it overlaps, and the order of the
// remaining boxes is re-established after every box (an insertion pass that sinks the suppressed ones).  One workgroup
// follows the reference's control flow literally: the inner loop over the later boxes (exact IoU, rescale, threshold)
// runs on all lanes, the insertion pass on lane 0 -- skipped when the pass is the identity (nothing rescaled in this
// round and the same suppressed suffix as in the round before: the block is a sub-block of one already in order).
// Position-indexed state (order, working scores, suppressed) lives in LDS up to kSoftLds boxes, in global scratch above.
constexpr int kSoftThreads = 1024;
constexpr int kSoftList = 1024;      // rescaled positions remembered per round
template <typename T> __device__ __forceinline__ T soft_decay(T iou, float param, int sup);
template <> __device__ __forceinline__ float soft_decay<float>(float iou, float param, int sup)
{
    // (the float ARGUMENTS as the host forms them, the functions in double and rounded once: glibc's powf / expf are the
    // correctly rounded floats in all but the rarest cases, the device's are 1 ulp off now and then -- see d3d_sincos)
    return sup == D3D_SUPPRESS_LINEAR ? 1 - (float)pow((double)iou, (double)param) : (float)exp((double)(-iou * iou / param));
}
template <> __device__ __forceinline__ double soft_decay<double>(double iou, float param, int sup)
{
    return sup == D3D_SUPPRESS_LINEAR ? 1 - pow(iou, (double)param) : exp(-iou * iou / (double)param);
}

template <typename T, bool ROTATED, typename B = T>
__global__ __launch_bounds__(kSoftThreads) void k_softnms(const B *__restrict__ boxes, const B *__restrict__ scores,
                                                          const int64_t *__restrict__ order_in, int n, int sup,
                                                          float iou_thr, float score_thr, float param,
                                                          BoxGeom<T> *geom, float4 *aabb, int *g_ord, T *g_sc, uint8_t *g_sp,
                                                          int in_lds, uint8_t *suppressed, uint8_t inv)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char soft_lds[];
    __shared__ int s_S, s_mod, s_prevS, s_cnt;
    __shared__ int s_list[kSoftList];
    T *sc = in_lds ? reinterpret_cast<T *>(soft_lds) : g_sc;                       // working score of the box at position p
    int *ord = in_lds ? reinterpret_cast<int *>(soft_lds + (size_t)n * sizeof(T)) : g_ord;
    uint8_t *sp = in_lds ? soft_lds + (size_t)n * (sizeof(T) + 4) : g_sp;          // suppressed, by position
    const int tid = threadIdx.x;
    if (tid == 0) { s_S = 0; s_prevS = -1; }
    for (int p = tid; p < n; p += kSoftThreads) {
        const int i = (int)order_in[p];
        ord[p] = p;                                          // initial rank of the box now at position p
        sc[p] = (T)scores[i];                                // nms.cpp:104: the scores are copied
        const BoxGeom<T> g = Box2D<T>::load(boxes + (size_t)i * 5);
        geom[p] = g;                                         // geometry by initial rank
        aabb[p] = cand_aabb(g, ROTATED);                     // 16 bytes: what the inner loop gathers first
    }
    __syncthreads();
    // nms.cpp:23-29: walking up from the last position, everything is suppressed until a score above the threshold
    // (position 0 is never touched) = the positions behind the LAST one whose score is above it
    for (int p = 1 + tid; p < n; p += kSoftThreads)
        if (sc[p] > (T)score_thr) atomicMax(&s_S, p);
    __syncthreads();
    const int last_above = s_S;
    for (int p = tid; p < n; p += kSoftThreads) sp[p] = p > last_above ? 1 : 0;
    __syncthreads();
    for (int pi = 0; pi < n; pi++) {
        if (sp[pi]) break;                                   // nms.cpp:38: the rest is suppressed (uniform)
        const BoxGeom<T> gi = geom[ord[pi]];
        const float4 fi = aabb[ord[pi]];
        if (tid == 0) { s_mod = 0; s_S = pi; s_cnt = 0; }
        __syncthreads();
        int smax = pi;
        for (int pj = pi + 1 + tid; pj < n; pj += kSoftThreads) {
            const int rj = ord[pj];
            T iou = 0;
            if (aabb_gap(fi, aabb[rj]) > 0.f) {              // conservative: the exact IoU is 0 otherwise
                const BoxGeom<T> gj = geom[rj];
                iou = ROTATED ? iou_rbox(gi, gj) : iou_aabb(gi, gj);
            }
            if (iou > (T)iou_thr) {                          // nms.cpp:53
                const T before = sc[pj];
                const uint8_t was = sp[pj];
                sc[pj] = before * soft_decay<T>(iou, param, sup);
                sp[pj] = sc[pj] < (T)score_thr ? 1 : 0;      // nms.cpp:58,62: assigned, so a box can come back
                // a score went UP (negative scores) or a suppressed box came back (a trailing score == threshold,
                // nms.cpp:23-29, that is not < threshold): the block is no longer ordered, no shortcut below
                if (sc[pj] > before || (was && !sp[pj])) s_mod = 2;
                const int slot = atomicAdd(&s_cnt, 1);
                if (slot < kSoftList) s_list[slot] = pj;
            }
            if (sp[pj]) smax = pj;                           // positions ascend: the last one seen is the largest
        }
        if (smax > pi) atomicMax(&s_S, smax);
        __syncthreads();
        // nms.cpp:74-94 on lane 0: S = last suppressed position; insertion pass over (pi, S), walking down from S - 1.
        // The pass moves only what is out of order.  When the block is a sub-block of the one ordered in the round before
        // (same S) and scores only went down, the unchanged boxes are already in place (each is followed by boxes that
        // did not score higher before and do not now), so visiting the rescaled positions, largest first, gives the
        // same result as visiting every position.
        if (tid < kWave) {                                   // wavefront 0; everything below is wave-uniform
            const int S = s_S, cnt = s_cnt;
            if (S > pi + 1) {
                // one insertion of nms.cpp:84-92, 64 positions per step: the box at pj moves right past every box
                // that outscores it (all the way to S - 1 when it is suppressed)
                auto insert = [&](int pj) {
                    const int j = ord[pj];
                    const T sj = sc[pj];
                    const uint8_t pjs = sp[pj];
                    int k = pj + 1;
                    while (k < S) {
                        const int idx = k + tid;
                        const bool valid = idx < S;
                        int o = 0; T v = 0; uint8_t u = 0;
                        if (valid) { o = ord[idx]; v = sc[idx]; u = sp[idx]; }
                        const unsigned long long pass = __ballot(valid && (pjs || v > sj));
                        const int c = pass == ~0ull ? kWave : __builtin_ctzll(~pass);     // leading boxes that stay ahead
                        if (!in_lds) __threadfence_block();
                        if (tid < c) { ord[idx - 1] = o; sc[idx - 1] = v; sp[idx - 1] = u; }
                        if (!in_lds) __threadfence_block();
                        k += c;
                        if (c < kWave) break;
                    }
                    if (tid == 0) { ord[k - 1] = j; sc[k - 1] = sj; sp[k - 1] = pjs; }
                    if (!in_lds) __threadfence_block();
                };
                if (S != s_prevS || cnt > kSoftList || s_mod == 2) {
                    for (int pj = S - 1; pj > pi; pj--) insert(pj);
                } else if (cnt > 0) {
                    if (tid == 0)
                        for (int a = 1; a < cnt; a++) {      // the few rescaled positions, descending
                            const int v = s_list[a];
                            int b = a - 1;
                            while (b >= 0 && s_list[b] < v) { s_list[b + 1] = s_list[b]; b--; }
                            s_list[b + 1] = v;
                        }
                    __builtin_amdgcn_wave_barrier();
                    for (int a = 0; a < cnt; a++) {
                        const int pj = s_list[a];
                        if (pj < S) insert(pj);
                    }
                }
            }
            if (tid == 0) s_prevS = S;
        }
        __syncthreads();
    }
    __syncthreads();
    for (int p = tid; p < n; p += kSoftThreads) suppressed[order_in[ord[p]]] = sp[p] ^ inv;
}

// ---------------------------------------------------------------- crop: points in rotated boxes
// indicators[i, j] = point j inside box i (closed AABB test, then the four closed half-plane tests); replaces
// crop_2dr (reference utils.cpp:9-47).  Lane = 4 consecutive points -> one 32-bit store per box row.
template <typename T>
__global__ __launch_bounds__(256) void k_crop2dr(const T *__restrict__ points, int64_t n, const T *__restrict__ boxes,
                                                 int64_t m, uint8_t *__restrict__ out)
{
    __shared__ BoxGeom<T> rows[kTileRows];
    const int64_t i0 = (int64_t)blockIdx.y * kTileRows;
    const int nrows = (int)((m - i0) < kTileRows ? (m - i0) : kTileRows);
    if (threadIdx.x < nrows) rows[threadIdx.x] = Box2D<T>::load(boxes + (i0 + threadIdx.x) * 5);
    const int64_t j0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    T px[4], py[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const bool ok = j0 + k < n;
        px[k] = ok ? points[(j0 + k) * 2] : (T)0;
        py[k] = ok ? points[(j0 + k) * 2 + 1] : (T)0;
    }
    __syncthreads();
    if (j0 >= n) return;
    const bool vec = (n % 4 == 0);
    for (int r = 0; r < nrows; r++) {
        const BoxGeom<T> g = rows[r];
        uint32_t word = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool in = quad_contains<T>(g, px[k], py[k]);
            word |= (in ? 1u : 0u) << (8 * k);
        }
        uint8_t *dst = out + (i0 + r) * n + j0;
        if (vec) *reinterpret_cast<uint32_t *>(dst) = word;
        else
            for (int k = 0; k < 4 && j0 + k < n; k++) dst[k] = (uint8_t)((word >> (8 * k)) & 1u);
    }
}

}  // namespace

// ====================================================================== C ABI
// Segmented lists were measured on MI355X (8 segments): with one reservation per workgroup the counter is no longer
// the bottleneck of the producer, while the consumers pay one dependent counter read per segment (k_iou_clip 12 -> 57 us),
// so every list is kept in one piece.
static unsigned int list_segments(unsigned long long) { return 1u; }

static unsigned long long iou_list_capacity(int64_t n, int64_t m)
{
    const unsigned long long pairs = (unsigned long long)n * (unsigned long long)m;
    return pairs < (1ull << 27) ? pairs : (1ull << 27);
}

// boxloss.hip
int d3d_internal_loss_iou_forward(const void *b1, int64_t n, const void *b2, int64_t m, int kind, int dtype, void *out, void *ws,
                                  size_t ws_bytes, unsigned long long list_cap, hipStream_t st);
int d3d_internal_loss_iou_backward(const void *b1, int64_t n, const void *b2, int64_t m, const void *grad, int kind, int dtype,
                                   void *g1, void *g2, void *ws, size_t ws_bytes, hipStream_t st);

extern "C" size_t d3d_iou2d_workspace_bytes(int64_t n, int64_t m, int32_t dtype)
{
    if (n < 1) n = 1;
    if (m < 1) m = 1;
    const size_t g = dtype != D3D_F32 ? sizeof(BoxGeom<double>) : sizeof(BoxGeom<float>);
    // 64 bytes per box instead of the 16 of the candidate boxes: GRBOX keeps its per-box hull terms there (boxloss.hip, HullPre)
    return d3d_align_up(g * n) + d3d_align_up(g * m) + d3d_align_up(64 * n) + d3d_align_up(64 * m) + d3d_align_up(sizeof(IouList)) + d3d_align_up(8 * iou_list_capacity(n, m)) +
           256;
}

template <typename T, bool ROTATED, typename S = T, typename B = T>
static int iou2d_two_phase(const B *b1, int64_t n, const B *b2, int64_t m, S *ious, void *ws, size_t ws_bytes, hipStream_t st,
                           uint32_t opts)
{
    WsCarver w(ws, ws_bytes);
    BoxGeom<T> *ga = w.take<BoxGeom<T>>(n);
    BoxGeom<T> *gb = w.take<BoxGeom<T>>(m);
    float4 *ra = w.take<float4>(n);
    float4 *cb = w.take<float4>(m);
    IouList *hdr = w.take<IouList>(1);
    unsigned long long cap = iou_list_capacity(n, m);
    unsigned long long *list = w.take<unsigned long long>(cap);
    if (!w.ok()) return D3D_ERR_WORKSPACE;
    if ((opts >> 8) != 0 && (unsigned long long)(opts >> 8) < cap) cap = opts >> 8;     // D3D_IOU_LIST_CAP: use less of it
    D3D_LAUNCH("k_geom", (k_geom2<T, ROTATED, B>), dim3((unsigned)d3d_divup(n + m, 256)), dim3(256), 0, st, b1, n, ga, ra, b2, m, gb, cb,
               hdr, list_segments(cap), ROTATED);
    S *fill = ious;
    if (reinterpret_cast<uintptr_t>(ious) & 15) {             // unaligned output: plain memset, candidates only
        D3D_HIP_CHECK(hipMemsetAsync(ious, 0, (size_t)n * (size_t)m * sizeof(S), st));
        fill = nullptr;
    }
    const int prows = pre_tile_rows(n, m);
    D3D_LAUNCH("k_iou_pre", k_iou_pre<S>, dim3((unsigned)d3d_divup(m, (int64_t)kPreCols), (unsigned)d3d_divup(n, (int64_t)prows)),
               dim3(kTileCols), 0, st, (const float4 *)ra, n, (const float4 *)cb, m, fill, hdr, list, cap, 0.f, prows);
    D3D_LAUNCH("k_iou_clip", (k_iou_clip<T, ROTATED, S>), dim3(256 * 16), dim3(256), 0, st, ga, gb, n, m, ious, hdr, list, cap);
    return D3D_OK;
}

extern "C" int d3d_iou2d_forward(const void *boxes1, int64_t n, const void *boxes2, int64_t m, int32_t iou_type,
                                 int32_t dtype, void *ious, void *workspace, size_t workspace_bytes, void *stream, uint32_t flags)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || m < 0 || (flags & 0xffu)) return D3D_ERR_BAD_ARG;
    if (dtype != D3D_F32 && dtype != D3D_F64 && dtype != D3D_F64_M32 && dtype != D3D_F32_WIDE) return D3D_ERR_BAD_ARG;
    const bool loss_kind = iou_type == D3D_IOU_GRBOX || iou_type == D3D_IOU_DRBOX;
    if (iou_type != D3D_IOU_BOX && iou_type != D3D_IOU_RBOX && !loss_kind) return D3D_ERR_UNSUPPORTED;
    if (loss_kind && (dtype == D3D_F64_M32 || dtype == D3D_F32_WIDE)) return D3D_ERR_UNSUPPORTED;
    if (n == 0 || m == 0) return D3D_OK;
    if (!boxes1 || !boxes2 || !ious) return D3D_ERR_BAD_ARG;
    if (loss_kind) {        // GIoU / DIoU: every pair has a value (boxloss.hip); GIoU lists the pairs that need the clip
        const bool use_ws = workspace && workspace_bytes >= d3d_iou2d_workspace_bytes(n, m, dtype) && n < (1ll << 32) && m < (1ll << 32);
        unsigned long long cap = iou_list_capacity(n, m);
        if ((flags >> 8) != 0 && (unsigned long long)(flags >> 8) < cap) cap = flags >> 8;        // D3D_IOU_LIST_CAP
        return d3d_internal_loss_iou_forward(boxes1, n, boxes2, m, iou_type == D3D_IOU_GRBOX ? 0 : 1, dtype, ious,
                                             use_ws ? workspace : nullptr, use_ws ? workspace_bytes : 0, cap, st);
    }
    const int64_t gy = d3d_divup(n, kTileRows);
    if (gy > 65535 || n >= (1ll << 32) || m >= (1ll << 32)) return D3D_ERR_BAD_ARG;   // callers tile above that
    const bool rot = iou_type == D3D_IOU_RBOX;
    if ((unsigned long long)n * (unsigned long long)m <= kIouSmallPairs && (flags >> 8) == 0) {      // (a list-cap flag asks for the list path)
        const dim3 sgrid((unsigned)d3d_divup(n * m, 256));
        if (dtype == D3D_F32_WIDE) {
            if (rot) D3D_LAUNCH("k_iou_small", (k_iou_small<double, true, float, float>), sgrid, dim3(256), 0, st, (const float *)boxes1, n, (const float *)boxes2, m, (float *)ious);
            else D3D_LAUNCH("k_iou_small", (k_iou_small<double, false, float, float>), sgrid, dim3(256), 0, st, (const float *)boxes1, n, (const float *)boxes2, m, (float *)ious);
        } else if (dtype == D3D_F64_M32) {
            if (rot) D3D_LAUNCH("k_iou_small", (k_iou_small<double, true, float>), sgrid, dim3(256), 0, st, (const double *)boxes1, n, (const double *)boxes2, m, (float *)ious);
            else D3D_LAUNCH("k_iou_small", (k_iou_small<double, false, float>), sgrid, dim3(256), 0, st, (const double *)boxes1, n, (const double *)boxes2, m, (float *)ious);
        } else if (dtype == D3D_F64) {
            if (rot) D3D_LAUNCH("k_iou_small", (k_iou_small<double, true>), sgrid, dim3(256), 0, st, (const double *)boxes1, n, (const double *)boxes2, m, (double *)ious);
            else D3D_LAUNCH("k_iou_small", (k_iou_small<double, false>), sgrid, dim3(256), 0, st, (const double *)boxes1, n, (const double *)boxes2, m, (double *)ious);
        } else {
            if (rot) D3D_LAUNCH("k_iou_small", (k_iou_small<float, true>), sgrid, dim3(256), 0, st, (const float *)boxes1, n, (const float *)boxes2, m, (float *)ious);
            else D3D_LAUNCH("k_iou_small", (k_iou_small<float, false>), sgrid, dim3(256), 0, st, (const float *)boxes1, n, (const float *)boxes2, m, (float *)ious);
        }
        return D3D_OK;
    }
    if (workspace && workspace_bytes >= d3d_iou2d_workspace_bytes(n, m, dtype)) {
        // zero fill + candidate list + one candidate per lane (BOX too: its IoU is non-zero only where the AABBs overlap)
#define D3D_TWO_PHASE(T, R) iou2d_two_phase<T, R>((const T *)boxes1, n, (const T *)boxes2, m, (T *)ious, workspace, workspace_bytes, st, flags)
        if (dtype == D3D_F32_WIDE) {
            if (rot) return iou2d_two_phase<double, true, float, float>((const float *)boxes1, n, (const float *)boxes2, m, (float *)ious, workspace, workspace_bytes, st, flags);
            return iou2d_two_phase<double, false, float, float>((const float *)boxes1, n, (const float *)boxes2, m, (float *)ious, workspace, workspace_bytes, st, flags);
        }
        if (dtype == D3D_F64_M32) {
            if (rot) return iou2d_two_phase<double, true, float>((const double *)boxes1, n, (const double *)boxes2, m, (float *)ious, workspace, workspace_bytes, st, flags);
            return iou2d_two_phase<double, false, float>((const double *)boxes1, n, (const double *)boxes2, m, (float *)ious, workspace, workspace_bytes, st, flags);
        }
        if (dtype == D3D_F64) return rot ? D3D_TWO_PHASE(double, true) : D3D_TWO_PHASE(double, false);
        return rot ? D3D_TWO_PHASE(float, true) : D3D_TWO_PHASE(float, false);
#undef D3D_TWO_PHASE
    }
    if (dtype == D3D_F64_M32 || dtype == D3D_F32_WIDE) return D3D_ERR_WORKSPACE;       // (the mixed forms have no workspace-free kernel)
    // single-kernel path: no workspace
    const bool al16 = (reinterpret_cast<uintptr_t>(ious) & 15) == 0;
#define D3D_IOU2D(T, R, K)                                                                                          \
    D3D_LAUNCH("k_iou2d", (k_iou2d<T, R, K>), dim3((unsigned)d3d_divup(m, (int64_t)kTileCols * K), (unsigned)gy),   \
               dim3(kTileCols), 0, st, (const T *)boxes1, n, (const T *)boxes2, m, (T *)ious, (const unsigned int *)nullptr)
    if (dtype == D3D_F64) {
        const bool vec = al16 && (m % 2 == 0);
        if (rot) { if (vec) D3D_IOU2D(double, true, 2); else D3D_IOU2D(double, true, 1); }
        else     { if (vec) D3D_IOU2D(double, false, 2); else D3D_IOU2D(double, false, 1); }
    } else {
        const bool vec = al16 && (m % 4 == 0);
        if (rot) { if (vec) D3D_IOU2D(float, true, 4); else D3D_IOU2D(float, true, 1); }
        else     { if (vec) D3D_IOU2D(float, false, 4); else D3D_IOU2D(float, false, 1); }
    }
#undef D3D_IOU2D
    return D3D_OK;
}

extern "C" size_t d3d_iou3d_workspace_bytes(int64_t n, int64_t m)
{
    if (n < 1) n = 1;
    if (m < 1) m = 1;
    return d3d_align_up(sizeof(BoxGeom<float>) * n) + d3d_align_up(sizeof(BoxGeom<float>) * m) + d3d_align_up(8 * n) +
           d3d_align_up(8 * m) + d3d_align_up(16 * n) + d3d_align_up(16 * m) + d3d_align_up(sizeof(IouList)) + d3d_align_up(8 * iou_list_capacity(n, m)) + 256;
}

static int iou3d_impl(const float *boxes1, int64_t n, const float *boxes2, int64_t m, int32_t rotated, float *out,
                      void *workspace, size_t workspace_bytes, hipStream_t st, int stride, bool complement)
{
    if (n < 0 || m < 0) return D3D_ERR_BAD_ARG;
    if (n == 0 || m == 0) return D3D_OK;
    if (!boxes1 || !boxes2 || !out) return D3D_ERR_BAD_ARG;
    if (d3d_divup(n, kTileRows) > 65535 || n >= (1ll << 32) || m >= (1ll << 32)) return D3D_ERR_BAD_ARG;
    if ((unsigned long long)n * (unsigned long long)m <= kIouSmallPairs) {
        const dim3 sgrid((unsigned)d3d_divup(n * m, 256));
        if (rotated) D3D_LAUNCH("k_iou3d_small", k_iou3d_small<true>, sgrid, dim3(256), 0, st, boxes1, n, boxes2, m, out, stride, complement);
        else D3D_LAUNCH("k_iou3d_small", k_iou3d_small<false>, sgrid, dim3(256), 0, st, boxes1, n, boxes2, m, out, stride, complement);
        return D3D_OK;
    }
    const unsigned gy = (unsigned)d3d_divup(n, kTileRows);
    const bool al16 = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    const bool vec = al16 && (m % 4 == 0);
#define D3D_IOU3D(R, K, FLAG)                                                                                      \
    D3D_LAUNCH("k_iou3d", (k_iou3d<R, K>), dim3((unsigned)d3d_divup(m, (int64_t)kTileCols * K), gy), dim3(kTileCols), \
               0, st, boxes1, n, boxes2, m, out, FLAG, stride, complement)
    if (workspace && workspace_bytes >= d3d_iou3d_workspace_bytes(n, m) && (al16 || !complement)) {
        // background fill + candidate list + dense clipping (see "rotated IoU, two-phase")
        WsCarver w(workspace, workspace_bytes);
        BoxGeom<float> *ga = w.take<BoxGeom<float>>(n);
        BoxGeom<float> *gb = w.take<BoxGeom<float>>(m);
        float2 *za = w.take<float2>(n);
        float2 *zb = w.take<float2>(m);
        float4 *ra = w.take<float4>(n);
        float4 *cb = w.take<float4>(m);
        IouList *hdr = w.take<IouList>(1);
        const unsigned long long cap = iou_list_capacity(n, m);
        unsigned long long *list = w.take<unsigned long long>(cap);
        if (!w.ok()) return D3D_ERR_WORKSPACE;
        D3D_LAUNCH("k_geom3d2", k_geom3d2, dim3((unsigned)d3d_divup(n + m, 256)), dim3(256), 0, st, boxes1, n, ga, ra, za, boxes2, m,
                   gb, cb, zb, hdr, list_segments(cap), rotated != 0, stride, complement);
        float *fill = out;
        if (!al16) {
            D3D_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)n * (size_t)m * sizeof(float), st));
            fill = nullptr;
        }
        const int prows = pre_tile_rows(n, m);
        D3D_LAUNCH("k_iou_pre", k_iou_pre<float>, dim3((unsigned)d3d_divup(m, (int64_t)kPreCols), (unsigned)d3d_divup(n, (int64_t)prows)),
                   dim3(kTileCols), 0, st, (const float4 *)ra, n, (const float4 *)cb, m, fill, hdr, list, cap, complement ? 1.f : 0.f,
                   prows);
        if (rotated)
            D3D_LAUNCH("k_iou3d_clip", k_iou3d_clip<true>, dim3(256 * 16), dim3(256), 0, st, ga, za, gb, zb, n, m, out, hdr, list, cap, complement);
        else
            D3D_LAUNCH("k_iou3d_clip", k_iou3d_clip<false>, dim3(256 * 16), dim3(256), 0, st, ga, za, gb, zb, n, m, out, hdr, list, cap, complement);
        return D3D_OK;
    }
    if (rotated) D3D_IOU3D(true, 1, (const unsigned int *)nullptr);
    else { if (vec) D3D_IOU3D(false, 4, (const unsigned int *)nullptr); else D3D_IOU3D(false, 1, (const unsigned int *)nullptr); }
#undef D3D_IOU3D
    return D3D_OK;
}

extern "C" int d3d_iou3d_forward(const float *boxes1, int64_t n, const float *boxes2, int64_t m, int32_t rotated,
                                 float *out, void *workspace, size_t workspace_bytes, void *stream)
{
    return iou3d_impl(boxes1, n, boxes2, m, rotated, out, workspace, workspace_bytes, (hipStream_t)stream, 7, false);
}

// BaseMatcher.prepare_boxes (reference d3d/tracking/matcher.pyx:46-80) for the IoU / RIoU metrics: src[n,9], dst[m,9] rows
// (label, score, x, y, z, lx, ly, lz, yaw) as Target3DArray.to_numpy lays them out; dimensions clipped to +-1e3
// (matcher.pyx:49-51); cache[i,j] = 1 - box3d_iou / box3dr_iou in fp32 (matcher.pyx:57-80).  One pass: the matrix is
// filled with 1 at store bandwidth and the overlapping pairs get 1 - iou.
extern "C" int d3d_match_distance(const float *src, int64_t n, const float *dst, int64_t m, int32_t rotated, float *cache,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    return iou3d_impl(src ? src + 2 : src, n, dst ? dst + 2 : dst, m, rotated, cache, workspace, workspace_bytes,
                      (hipStream_t)stream, 9, true);
}

constexpr size_t kSoftLdsBytes = 128 * 1024;      // position-indexed state of the soft-NMS kernel stays in LDS below this
template <typename T, bool ROTATED, typename B = T>
static int softnms_typed(const B *boxes, const B *scores, const int64_t *order, int64_t n, int sup, float iou_thr,
                         float score_thr, float param, uint8_t *suppressed, void *ws, size_t ws_bytes, hipStream_t st, uint32_t opts)
{
    WsCarver w(ws, ws_bytes);
    BoxGeom<T> *geom = w.take<BoxGeom<T>>(n);
    float4 *aabb = w.take<float4>(n);
    int *ord = w.take<int>(n);
    T *sc = w.take<T>(n);
    uint8_t *sp = w.take<uint8_t>(n);
    if (!ws || !w.ok()) return D3D_ERR_WORKSPACE;
    const size_t lds = (size_t)n * (sizeof(T) + 4 + 1);
    const bool in_lds = lds <= kSoftLdsBytes && !(opts & D3D_NMS_SOFT_NO_LDS);    // (flag: the global-scratch variant)
    if (in_lds)
        D3D_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_softnms<T, ROTATED, B>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSoftLdsBytes));
    D3D_LAUNCH("k_softnms", (k_softnms<T, ROTATED, B>), dim3(1), dim3(kSoftThreads), in_lds ? lds : 0, st, boxes, scores, order, (int)n,
               sup, iou_thr, score_thr, param, geom, aabb, ord, sc, sp, in_lds ? 1 : 0, suppressed, (uint8_t)((opts & D3D_NMS_KEEP_MASK) ? 1 : 0));
    return D3D_OK;
}

static size_t nms_core_workspace_bytes(int64_t n)
{
    if (n < 1) n = 1;
    const size_t nb = (size_t)d3d_divup(n, 64);
    return d3d_align_up(nb * 64 * sizeof(BoxGeom<double>)) + d3d_align_up(nb * 64 * 16) + 2 * d3d_align_up(nb * 64) +
           d3d_align_up(nb * 64 * 4) * 3 + d3d_align_up((nb * 64 / kScanTile + 2 + 80) * 8) + 256 + 2 * d3d_align_up((size_t)nms_cand_capacity(n) * 4) +
           256 + d3d_align_up(nb * 8) +
           d3d_align_up(sizeof(NmsCand)) +
           d3d_align_up((size_t)nms_cand_capacity(n) * 8) + 2 * d3d_align_up(nb * 64 * 4) + d3d_align_up((nb * 64 + kCandPad) * 16) +
           d3d_align_up(nb * 64 * 4) + d3d_align_up(d3d_internal_argsort_i32_bytes(n)) + d3d_align_up(nb * 64 * nb * 8) + 256 +
           d3d_align_up((size_t)(kGridCells + 1) * kGridPad * 4) + d3d_align_up((kGridCells + 1) * 4) + d3d_align_up((nb * 64 / 256 + 1) * 6 * 4) + d3d_align_up(sizeof(NmsGrid)) +
           d3d_align_up(kGridReg * nb * 64 * 4) * 3 + d3d_align_up(kGridReg * nb * 64 * 16) + d3d_align_up(kGridReg * nb * 64);
}

extern "C" size_t d3d_nms2d_workspace_bytes(int64_t n)
{
    if (n < 1) n = 1;
    // [NMS arrays][order, when the caller passes none][workspace of d3d_argsort_desc]
    return nms_core_workspace_bytes(n) + d3d_align_up((size_t)n * 8) + d3d_align_up(d3d_argsort_desc_workspace_bytes(n, D3D_F64)) + 256;
}

static int nms2d_impl(const void *boxes, const void *scores, const int64_t *order, int64_t n, int32_t iou_type,
                      int32_t suppression_type, int32_t dtype, float iou_threshold, float score_threshold,
                      float suppression_param, uint8_t *suppressed, void *workspace, size_t workspace_bytes,
                      void *stream, uint32_t flags, int32_t *host_word)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || (flags & 0xffu & ~(uint32_t)(D3D_NMS_BROAD_SWEEP | D3D_NMS_FORCE_DENSE | D3D_NMS_SOFT_NO_LDS | D3D_NMS_GENERAL | D3D_NMS_TEST_WITHHOLD | D3D_NMS_FORCE_LEVELS | D3D_NMS_ONE_LEVEL | D3D_NMS_KEEP_MASK))) return D3D_ERR_BAD_ARG;
    if (dtype != D3D_F32 && dtype != D3D_F64 && dtype != D3D_F32_WIDE) return D3D_ERR_BAD_ARG;
    if (iou_type != D3D_IOU_BOX && iou_type != D3D_IOU_RBOX) return D3D_ERR_UNSUPPORTED;   // common.h:25
    if (suppression_type != D3D_SUPPRESS_HARD && suppression_type != D3D_SUPPRESS_LINEAR &&
        suppression_type != D3D_SUPPRESS_GAUSSIAN)
        return D3D_ERR_UNSUPPORTED;                                                          // common.h:40
    if (n == 0) return D3D_OK;
    if (!boxes || !scores || !suppressed) return D3D_ERR_BAD_ARG;
    // order == NULL: the descending argsort of the scores happens here (nms.cpp:103 does it inside nms2d too) -- inside the
    // first kernel of the small-set path, by d3d_argsort_desc otherwise; both live behind the NMS part of the workspace
    const size_t nms_bytes = nms_core_workspace_bytes(n);
    if (!workspace || workspace_bytes < d3d_nms2d_workspace_bytes(n)) return D3D_ERR_WORKSPACE;
    int64_t *order_ws = reinterpret_cast<int64_t *>(static_cast<char *>(workspace) + nms_bytes);
    const bool small = suppression_type == D3D_SUPPRESS_HARD && nms_small_eligible(n, flags);
    if (!order && !small) {
        char *sort_ws = reinterpret_cast<char *>(order_ws) + d3d_align_up((size_t)n * 8);
        // (D3D_F32_WIDE: the order of the fp32 scores is the order of their widened values)
        const int sdt = dtype == D3D_F32_WIDE ? D3D_F32 : dtype;
        const int rc = d3d_argsort_desc(scores, n, sdt, order_ws, sort_ws, d3d_argsort_desc_workspace_bytes(n, sdt), stream);
        if (rc) return rc;
        order = order_ws;
    }
    if (suppression_type != D3D_SUPPRESS_HARD) {
        // one workgroup, one round per box that is still alive when its turn comes, each round a sweep over the boxes behind
        // it (16-byte AABB reads; position-indexed state in global scratch above kSoftLds boxes): ~n rounds of ~n / 50 us --
        // 100 k boxes with most of them alive take seconds (the reference's loop, nms.cpp:60-94: n^2 / 2 clips and an
        // insertion pass that is itself quadratic per round).  No size limit of its own (nms.cpp has none)
        if (n >= (1ll << 31) - 64) return D3D_ERR_BAD_ARG;
        const bool rot = iou_type == D3D_IOU_RBOX;
        if (dtype == D3D_F32_WIDE)
            return rot ? softnms_typed<double, true, float>((const float *)boxes, (const float *)scores, order, n, suppression_type,
                                                            iou_threshold, score_threshold, suppression_param, suppressed, workspace,
                                                            workspace_bytes, st, flags)
                       : softnms_typed<double, false, float>((const float *)boxes, (const float *)scores, order, n, suppression_type,
                                                             iou_threshold, score_threshold, suppression_param, suppressed, workspace,
                                                             workspace_bytes, st, flags);
        if (dtype == D3D_F64)
            return rot ? softnms_typed<double, true>((const double *)boxes, (const double *)scores, order, n, suppression_type,
                                                     iou_threshold, score_threshold, suppression_param, suppressed, workspace,
                                                     workspace_bytes, st, flags)
                       : softnms_typed<double, false>((const double *)boxes, (const double *)scores, order, n, suppression_type,
                                                      iou_threshold, score_threshold, suppression_param, suppressed, workspace,
                                                      workspace_bytes, st, flags);
        return rot ? softnms_typed<float, true>((const float *)boxes, (const float *)scores, order, n, suppression_type,
                                                iou_threshold, score_threshold, suppression_param, suppressed, workspace,
                                                workspace_bytes, st, flags)
                   : softnms_typed<float, false>((const float *)boxes, (const float *)scores, order, n, suppression_type,
                                                 iou_threshold, score_threshold, suppression_param, suppressed, workspace,
                                                 workspace_bytes, st, flags);
    }
    if (d3d_divup(n, 64) > 65535) return D3D_ERR_BAD_ARG;
    if (dtype == D3D_F32_WIDE)
        return nms_typed<double, float>((const float *)boxes, (const float *)scores, order, n, iou_type, iou_threshold,
                                        score_threshold, suppressed, workspace, nms_bytes, st, flags, order_ws, host_word);
    if (dtype == D3D_F64)
        return nms_typed<double>((const double *)boxes, (const double *)scores, order, n, iou_type, iou_threshold,
                                 score_threshold, suppressed, workspace, nms_bytes, st, flags, order_ws, host_word);
    return nms_typed<float>((const float *)boxes, (const float *)scores, order, n, iou_type, iou_threshold,
                            score_threshold, suppressed, workspace, nms_bytes, st, flags, order_ws, host_word);
}

extern "C" int d3d_nms2d(const void *boxes, const void *scores, const int64_t *order, int64_t n, int32_t iou_type,
                         int32_t suppression_type, int32_t dtype, float iou_threshold, float score_threshold,
                         float suppression_param, uint8_t *suppressed, void *workspace, size_t workspace_bytes,
                         void *stream, uint32_t flags)
{
    return nms2d_impl(boxes, scores, order, n, iou_type, suppression_type, dtype, iou_threshold, score_threshold, suppression_param,
                      suppressed, workspace, workspace_bytes, stream, flags, nullptr);
}

extern "C" int d3d_nms2d_notify(const void *boxes, const void *scores, const int64_t *order, int64_t n, int32_t iou_type,
                                int32_t suppression_type, int32_t dtype, float iou_threshold, float score_threshold,
                                float suppression_param, uint8_t *suppressed, void *workspace, size_t workspace_bytes,
                                void *stream, uint32_t flags, int32_t *host_word)
{
    return nms2d_impl(boxes, scores, order, n, iou_type, suppression_type, dtype, iou_threshold, score_threshold, suppression_param,
                      suppressed, workspace, workspace_bytes, stream, flags, host_word);
}

// which route the last hard-NMS call on this workspace took (a chained scan that gives up used to be silent)
extern "C" int d3d_nms2d_status(const void *workspace, int32_t suppression_type, void *stream, uint32_t *status)
{
    if (!status) return D3D_ERR_BAD_ARG;
    *status = 0;
    if (suppression_type != D3D_SUPPRESS_HARD) return D3D_OK;         // (soft-NMS has one route)
    if (!workspace) return D3D_ERR_BAD_ARG;
    NmsFlags h{};
    D3D_HIP_CHECK(hipMemcpyAsync(&h, workspace, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    D3D_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    *status = (h.need_sweep ? D3D_NMS_STATUS_DENSE_PATH : 0u) | (h.scan_gave_up ? D3D_NMS_STATUS_SCAN_GAVE_UP : 0u);
    return D3D_OK;
}

extern "C" int d3d_crop_2dr(const void *points, int64_t n, const void *boxes, int64_t m, int32_t dtype, uint8_t *out,
                            void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || m < 0) return D3D_ERR_BAD_ARG;
    if (dtype != D3D_F32 && dtype != D3D_F64) return D3D_ERR_BAD_ARG;
    if (n == 0 || m == 0) return D3D_OK;
    if (!points || !boxes || !out) return D3D_ERR_BAD_ARG;
    if (d3d_divup(m, kTileRows) > 65535) return D3D_ERR_BAD_ARG;
    if (dtype == D3D_F32) {          // fp32, up to 4096 boxes: a box grid per workgroup, only the hits are stored (crop.hip)
        const int rc = d3d_internal_crop2dr_grid_f32((const float *)points, n, (const float *)boxes, m, out, st);
        if (rc != D3D_ERR_UNSUPPORTED) return rc;
    }
    dim3 grid((unsigned)d3d_divup(n, 256 * 4), (unsigned)d3d_divup(m, kTileRows));
    if (dtype == D3D_F64)
        D3D_LAUNCH("k_crop2dr", k_crop2dr<double>, grid, dim3(256), 0, st, (const double *)points, n, (const double *)boxes, m, out);
    else
        D3D_LAUNCH("k_crop2dr", k_crop2dr<float>, grid, dim3(256), 0, st, (const float *)points, n, (const float *)boxes, m, out);
    return D3D_OK;
}

template <typename T, typename G = T, typename B = T>
static int iou2d_backward_typed(const B *b1, int64_t n, const B *b2, int64_t m, const G *grad, bool rot, T *g1, T *g2, void *ws,
                                size_t ws_bytes, hipStream_t st)
{
    WsCarver w(ws, ws_bytes);
    BoxGeom<T> *ga = w.take<BoxGeom<T>>(n);
    BoxGeom<T> *gb = w.take<BoxGeom<T>>(m);
    float4 *ra = w.take<float4>(n);
    float4 *cb = w.take<float4>(m);
    const int64_t wpr = d3d_divup(m, 64);
    // a row chunk's marks at a time, one bit per pair, inside what the forward's workspace holds for its candidate list (8 bytes x
    // iou_list_capacity): matrices beyond 8.6e9 pairs take several chunks (they came back with D3D_ERR_WORKSPACE)
    int64_t rows_bm = n < (int64_t)65535 * kTileRows ? n : (int64_t)65535 * kTileRows;
    {
        const int64_t rows_fit = (int64_t)(iou_list_capacity(n, m) / (unsigned long long)wpr);
        if (rows_bm > rows_fit) {                      // (never below 2^27 pairs: n x ceil(m / 64) words <= n x m)
            if (rows_fit < 8) return D3D_ERR_WORKSPACE;
            rows_bm = rows_fit & ~(int64_t)7;
        }
    }
    unsigned long long *bitmap = w.take<unsigned long long>((size_t)rows_bm * (size_t)wpr);
    unsigned long long *nmarks = w.take<unsigned long long>(kMarkStripes + 1);        // 64 counters + the decision
    if (!ws || !w.ok()) return D3D_ERR_WORKSPACE;      // (geometry and candidate boxes of both sets + one bit per pair: inside the forward's workspace)
    if (g2 == g1 + 5 * (size_t)n) D3D_HIP_CHECK(hipMemsetAsync(g1, 0, sizeof(T) * 5 * (size_t)(n + m), st));   // one buffer: one launch
    else {
        D3D_HIP_CHECK(hipMemsetAsync(g1, 0, sizeof(T) * 5 * (size_t)n, st));
        D3D_HIP_CHECK(hipMemsetAsync(g2, 0, sizeof(T) * 5 * (size_t)m, st));
    }
    if ((unsigned long long)n * (unsigned long long)m <= kIouSmallPairs) {
        if (rot) D3D_LAUNCH("k_iou_grad_small", (k_iou_grad_small<T, true, G, B>), dim3((unsigned)d3d_divup(n * m, 256)), dim3(256), 0, st, b1, n, b2, m, grad, g1, g2);
        else D3D_LAUNCH("k_iou_grad_small", (k_iou_grad_small<T, false, G, B>), dim3((unsigned)d3d_divup(n * m, 256)), dim3(256), 0, st, b1, n, b2, m, grad, g1, g2);
        return D3D_OK;
    }
    D3D_LAUNCH("k_geom", (k_geom2<T, false, B>), dim3((unsigned)d3d_divup(n + m, 256)), dim3(256), 0, st, b1, n, ga, ra, b2, m, gb, cb,
               (IouList *)nullptr, 1u, rot);           // (both operands in one launch, as the forward does)
    {                                                  // marks, then tiles with LDS accumulators (k_iou_grad_mark, k_iou_grad_tiles)
        int tr = kTileRows;                            // fewer rows per workgroup while the launch is short of 2048 workgroups
        while (tr > 8 && d3d_divup(m, kGradCols) * d3d_divup(n, tr) < 2048) tr >>= 1;
        while (tr > 8 && tr > rows_bm) tr >>= 1;
        // rows per pass: what the bitmap holds and one launch's grid.y covers (a pass may end on a partial tile -- rounding this down
        // to whole tiles gave 5000 rows a second pass of 8 rows with its four launches)
        const int64_t rows_max = (int64_t)65535 * tr < rows_bm ? (int64_t)65535 * tr : rows_bm;
        for (int64_t r0 = 0; r0 < n; r0 += rows_max) {
            const int64_t nr = (n - r0) < rows_max ? (n - r0) : rows_max;
#define D3D_GRAD_TILES(R)                                                                                                                   \
    D3D_HIP_CHECK(hipMemsetAsync(nmarks, 0, 8 * kMarkStripes, st));                                                                                         \
    D3D_LAUNCH("k_iou_grad_mark", (k_iou_grad_mark<T, R, G>), dim3((unsigned)d3d_divup(m, 2 * kGradCols), (unsigned)d3d_divup(nr, tr)),            \
               dim3(kGradCols), 0, st, (const BoxGeom<T> *)ga + r0, (const float4 *)ra + r0, nr, (const BoxGeom<T> *)gb, (const float4 *)cb, \
               m, grad + r0 * m, bitmap, wpr, tr, nmarks);                                                                                  \
    D3D_LAUNCH("k_iou_grad_decide", k_iou_grad_decide, dim3(1), dim3(kMarkStripes), 0, st, nmarks, nr, m);                                    \
    D3D_LAUNCH("k_iou_grad_tiles", (k_iou_grad_tiles<T, R, G, B>), dim3((unsigned)d3d_divup(m, kGradCols), (unsigned)d3d_divup(nr, tr)),               \
               dim3(kGradCols), 0, st, (const BoxGeom<T> *)ga + r0, (const float4 *)ra + r0, b1 + r0 * 5, nr, (const BoxGeom<T> *)gb,        \
               (const float4 *)cb, b2, m, grad + r0 * m, g1 + r0 * 5, g2, tr, (const unsigned long long *)bitmap, wpr,                      \
               (const unsigned long long *)nmarks);                                                                                         \
    D3D_LAUNCH("k_iou_grad_sparse", (k_iou_grad_sparse<T, R, G, B>), dim3((unsigned)d3d_divup(nr * wpr, (int64_t)kSparseWords)), dim3(256), 0, st,   \
               (const BoxGeom<T> *)ga + r0, b1 + r0 * 5, nr, (const BoxGeom<T> *)gb, b2, m, grad + r0 * m, g1 + r0 * 5, g2,                  \
               (const unsigned long long *)bitmap, wpr, (const unsigned long long *)nmarks)
            if (rot) { D3D_GRAD_TILES(true); } else { D3D_GRAD_TILES(false); }
#undef D3D_GRAD_TILES
        }
    }
    return D3D_OK;
}

extern "C" int d3d_iou2d_backward(const void *boxes1, int64_t n, const void *boxes2, int64_t m, const void *grad,
                                  int32_t iou_type, int32_t dtype, void *grad_boxes1, void *grad_boxes2, void *workspace,
                                  size_t workspace_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || m < 0) return D3D_ERR_BAD_ARG;
    if (dtype != D3D_F32 && dtype != D3D_F64 && dtype != D3D_F64_M32 && dtype != D3D_F32_WIDE) return D3D_ERR_BAD_ARG;
    const bool loss_kind = iou_type == D3D_IOU_GRBOX || iou_type == D3D_IOU_DRBOX;
    if (iou_type != D3D_IOU_BOX && iou_type != D3D_IOU_RBOX && !loss_kind) return D3D_ERR_UNSUPPORTED;
    if (loss_kind && (dtype == D3D_F64_M32 || dtype == D3D_F32_WIDE)) return D3D_ERR_UNSUPPORTED;
    if (n > 0 && (!boxes1 || !grad_boxes1)) return D3D_ERR_BAD_ARG;
    if (m > 0 && (!boxes2 || !grad_boxes2)) return D3D_ERR_BAD_ARG;
    const size_t esz = dtype != D3D_F32 ? 8 : 4;
    if (n == 0 || m == 0) {
        if (n > 0) D3D_HIP_CHECK(hipMemsetAsync(grad_boxes1, 0, esz * 5 * (size_t)n, st));
        if (m > 0) D3D_HIP_CHECK(hipMemsetAsync(grad_boxes2, 0, esz * 5 * (size_t)m, st));
        return D3D_OK;
    }
    if (!grad || n >= (1ll << 32) || m >= (1ll << 32)) return D3D_ERR_BAD_ARG;
    if (loss_kind)
        return d3d_internal_loss_iou_backward(boxes1, n, boxes2, m, grad, iou_type == D3D_IOU_GRBOX ? 0 : 1, dtype, grad_boxes1,
                                              grad_boxes2, workspace, workspace ? workspace_bytes : 0, st);
    if (workspace_bytes < d3d_iou2d_workspace_bytes(n, m, dtype)) return D3D_ERR_WORKSPACE;
    const bool rot = iou_type == D3D_IOU_RBOX;
    if (dtype == D3D_F32_WIDE)
        return iou2d_backward_typed<double, float, float>((const float *)boxes1, n, (const float *)boxes2, m, (const float *)grad, rot,
                                                          (double *)grad_boxes1, (double *)grad_boxes2, workspace, workspace_bytes, st);
    if (dtype == D3D_F64_M32)
        return iou2d_backward_typed<double, float>((const double *)boxes1, n, (const double *)boxes2, m, (const float *)grad, rot,
                                                   (double *)grad_boxes1, (double *)grad_boxes2, workspace, workspace_bytes, st);
    if (dtype == D3D_F64)
        return iou2d_backward_typed<double>((const double *)boxes1, n, (const double *)boxes2, m, (const double *)grad, rot,
                                            (double *)grad_boxes1, (double *)grad_boxes2, workspace, workspace_bytes, st);
    return iou2d_backward_typed<float>((const float *)boxes1, n, (const float *)boxes2, m, (const float *)grad, rot,
                                       (float *)grad_boxes1, (float *)grad_boxes2, workspace, workspace_bytes, st);
}
