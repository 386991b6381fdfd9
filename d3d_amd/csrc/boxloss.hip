// boxloss.hip -- the loss-path operators of d3d.box on MI355X (gfx950): GIoU / DIoU of rotated boxes forward + backward,
// the autograd bookkeeping outputs of the rotated IoU family, and the signed point-to-box distance forward + backward.
// Replaces reference d3d/box/iou.cpp:213-419 + iou_cuda.cu:216-440 (giou2dr_* / diou2dr_*), the nx / xflags outputs of
// iou.cpp:95-141, and d3d/box/dist.cpp + dist_cuda.cu (pdist2dr_*).
//
// Unlike IoU, GIoU / DIoU are non-zero for EVERY pair (disjoint boxes get a negative value), so there is no candidate list:
// one pair per lane over a 64-row x 256-column tile, row geometry broadcast from LDS, lanes along the columns (coalesced
// row-major stores).  Gradients are analytic (geom.hpp) and accumulated race-free: a row's five partials are reduced across
// the wavefront by shuffles (one atomic per wavefront and row), a column's stay in the lane's registers over the tile's rows
// (one atomic per lane and tile) -- the reference's kernels add from many threads with plain += (iou_cuda.cu:72-73,184-185).
#include "common.hpp"
#include "geom.hpp"

namespace {

constexpr int kCols = 256, kRows = 64;

template <typename T> struct RowBox { BoxGeom<T> g; T w, h, c, s; };

template <typename T> __device__ __forceinline__ RowBox<T> load_row(const T *b)
{
    RowBox<T> r;
    r.g = make_geom<T>(b[0], b[1], b[2], b[3], b[4]);
    r.w = b[2]; r.h = b[3];
    d3d_sincos(b[4], &r.s, &r.c);
    return r;
}

template <typename T> __device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// The five sums of a row's gradient over the wavefront in NINE exchanges instead of thirty (round 5: the thirty, two
// ds_bpermute each in fp64, and five atomics one after the other were most of k_diou_grad_main's 767 us on 6 k x 6 k):
// at every level a lane sends the value its partner keeps, so the number of values halves with the distance --
// (v0, v1), (v2, v3), v4 across xor 1; then two, one, and three plain levels.  Returns to lane L the total of value
// min(L & 7, 4): lanes 0 .. 4 hold the five sums (and add them to memory side by side).
template <typename T> __device__ __forceinline__ T wave_sum5(const T (&v)[5], int lane)
{
    const bool o1 = lane & 1, o2 = lane & 2, o4 = lane & 4;
    const T a0 = (o1 ? v[1] : v[0]) + __shfl_xor(o1 ? v[0] : v[1], 1, kWave);
    const T a1 = (o1 ? v[3] : v[2]) + __shfl_xor(o1 ? v[2] : v[3], 1, kWave);
    const T a2 = v[4] + __shfl_xor(v[4], 1, kWave);
    const T b0 = (o2 ? a1 : a0) + __shfl_xor(o2 ? a0 : a1, 2, kWave);
    const T b1 = a2 + __shfl_xor(a2, 2, kWave);
    T c = (o4 ? b1 : b0) + __shfl_xor(o4 ? b0 : b1, 4, kWave);
    c += __shfl_xor(c, 8, kWave);
    c += __shfl_xor(c, 16, kWave);
    c += __shfl_xor(c, 32, kWave);
    return c;
}

// GIoU / DIoU of one pair by the complete routine (clip + hull with the tie rules / diameter), out of line: what the forward-only forms defer
template <typename T, int KIND> __device__ __noinline__ T loss_complete(const BoxGeom<T> &a, const BoxGeom<T> &b)
{
    T da[5], db[5];
    return loss_iou_rbox<T, KIND, false>(a, b, (T)0, (T)0, (T)0, (T)0, da, db);       // forward: the sizes are not read
}

// ---------------------------------------------------------------- GIoU / DIoU forward
template <typename T, int KIND>
__global__ __launch_bounds__(kCols) void k_loss_iou(const T *__restrict__ b1, int64_t n, const T *__restrict__ b2, int64_t m,
                                                    T *__restrict__ out, const unsigned int *only_if = nullptr)
{
    if (only_if && !*only_if) return;                  // the redo after a list overflow of k_giou_main: nothing to do otherwise
    __shared__ RowBox<T> rows[kRows];
    __shared__ HullPre<T> hulls[kRows];
    const int64_t i0 = (int64_t)blockIdx.y * kRows, j = (int64_t)blockIdx.x * kCols + threadIdx.x;
    const int nrows = (int)((n - i0) < kRows ? (n - i0) : kRows);
    if (threadIdx.x < nrows) {
        const RowBox<T> r = load_row<T>(b1 + (i0 + threadIdx.x) * 5);
        rows[threadIdx.x] = r;
        hulls[threadIdx.x] = hull_pre<T>(r.g);
    }
    __syncthreads();
    if (j >= m) return;
    const RowBox<T> c = load_row<T>(b2 + j * 5);
    T *o = out + i0 * m + j;
    // the forward-only forms where they apply (geom.hpp, giou_rbox_apart / diou_rbox_apart), the complete routine for the rest
    const HullPre<T> hc = hull_pre<T>(c.g);
    for (int r = 0; r < nrows; r++) {
        bool defer;
        T v = loss_rbox_apart<T, KIND>(rows[r].g, hulls[r], c.g, hc, defer);
        if (__any(defer)) {
            const T full = loss_complete<T, KIND>(rows[r].g, c.g);
            v = defer ? full : v;
        }
        __builtin_nontemporal_store(v, o);
        o += m;
    }
}

// ---------------------------------------------------------------- GIoU forward in two kernels (round 5)
// Almost every pair of a large matrix is two boxes apart from each other: its value needs the hull only, and the hull in the
// forward-only form of geom.hpp (hull_area2_clear) is ~280 vector + ~190 scalar instructions on 97 VGPRs.  The clip for the
// pairs whose bounding boxes overlap and the tie rules for the pairs with a corner on an edge line cost 160+ VGPRs and a
// call -- kept in the same kernel they took the occupancy of ALL pairs from 4-5 wavefronts per SIMD to 1.1 (scratch for the call;
// MeanOccupancyPerCU 4.6 -> 13.6 without them).  So k_giou_main computes the pairs that need neither and LISTS the others,
// k_giou_fix computes the listed pairs, one per lane, with the complete routine (loss_iou_rbox: same code as the single-kernel
// path and as before this round).  A list that overflows raises a flag and the single-kernel path redoes the matrix.
struct FixList { unsigned long long count; unsigned int overflow, pad; };
constexpr int kFixBatch = 256;

// geometry of both box sets, once per box (the sine / cosine routines stay out of the pair kernel), and the list header reset
template <typename T>
__global__ __launch_bounds__(256) void k_giou_geom(const T *__restrict__ b1, int64_t n, const T *__restrict__ b2, int64_t m,
                                                   BoxGeom<T> *__restrict__ ga, HullPre<T> *__restrict__ ha, BoxGeom<T> *__restrict__ gb,
                                                   HullPre<T> *__restrict__ hb, FixList *hdr)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) { hdr->count = 0; hdr->overflow = 0; }
    if (t >= n + m) return;
    const T *b = t < n ? b1 + t * 5 : b2 + (t - n) * 5;
    const BoxGeom<T> g = make_geom<T>(b[0], b[1], b[2], b[3], b[4]);
    if (t < n) { ga[t] = g; ha[t] = hull_pre<T>(g); }
    else { gb[t - n] = g; hb[t - n] = hull_pre<T>(g); }
}

template <typename T, int KIND>
__global__ __launch_bounds__(kCols) __attribute__((amdgpu_waves_per_eu(4))) void k_giou_main(const BoxGeom<T> *__restrict__ ga, const HullPre<T> *__restrict__ ha, int64_t n,
                                                     const BoxGeom<T> *__restrict__ gb, const HullPre<T> *__restrict__ hb, int64_t m,
                                                     T *__restrict__ out, FixList *hdr, unsigned long long *__restrict__ list,
                                                     unsigned long long cap)
{
    __shared__ BoxGeom<T> rows[kRows];
    __shared__ HullPre<T> hulls[kRows];
    __shared__ unsigned int batch[kCols / 64][kFixBatch];       // (row << 16 | local column)
    __shared__ unsigned int wcnt[kCols / 64];
    __shared__ unsigned long long bbase;
    const int64_t i0 = (int64_t)blockIdx.y * kRows, jb = (int64_t)blockIdx.x * kCols, j = jb + threadIdx.x;
    const int nrows = (int)((n - i0) < kRows ? (n - i0) : kRows);
    if (threadIdx.x < nrows) { rows[threadIdx.x] = ga[i0 + threadIdx.x]; hulls[threadIdx.x] = ha[i0 + threadIdx.x]; }
    const bool valid = j < m;
    const BoxGeom<T> c = gb[valid ? j : m - 1];
    const HullPre<T> hc = hb[valid ? j : m - 1];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int *q = batch[wave];
    unsigned int wn = 0;                                          // wave-uniform fill of the batch
    auto write_out = [&](unsigned long long base) {
        __builtin_amdgcn_wave_barrier();          // LDS ops of one wavefront complete in order: no s_barrier needed
        for (unsigned int t = lane; t < wn; t += 64) {
            const unsigned int e = q[t];
            if (base + t < cap) list[base + t] = ((unsigned long long)(i0 + (e >> 16)) << 32) | (unsigned long long)(jb + (e & 0xffffu));
            else hdr->overflow = 1;
        }
        wn = 0;
    };
    // lanes past the last column work on the last column once more and store the same value there: a store under `valid` would
    // take the whole computation under that branch with it, behind ALL the side tests (whose results then all wait in SGPRs)
    T *o = out + i0 * m + (valid ? j : m - 1);
    for (int r = 0; r < nrows; r++) {
        bool defer;
        const T v = loss_rbox_apart<T, KIND>(rows[r], hulls[r], c, hc, defer);
        __builtin_nontemporal_store(v, o);
        o += m;
        const unsigned long long mask = __ballot(defer && valid);
        if (mask) {
            const unsigned int cnt = (unsigned int)__popcll(mask);
            if (wn + cnt > (unsigned int)kFixBatch) {             // batch full: the wavefront reserves
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(&hdr->count, (unsigned long long)wn);
                write_out(__shfl(base, 0, 64));
            }
            if (defer && valid) q[wn + __popcll(mask & ((1ull << lane) - 1))] = ((unsigned)r << 16) | (unsigned)threadIdx.x;
            wn += cnt;
        }
    }
    // what is left in the four batches: ONE atomic per workgroup (same-address atomics are serialised, box.hip k_iou_pre)
    if (lane == 0) wcnt[wave] = wn;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        bbase = total ? atomicAdd(&hdr->count, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    unsigned long long base = bbase;
    for (int w = 0; w < wave; w++) base += wcnt[w];
    write_out(base);
}

template <typename T, int KIND>
__global__ __launch_bounds__(256) void k_giou_fix(const BoxGeom<T> *__restrict__ ga, const BoxGeom<T> *__restrict__ gb, int64_t m,
                                                  T *__restrict__ out, const FixList *hdr, const unsigned long long *__restrict__ list,
                                                  unsigned long long cap)
{
    const unsigned long long cnt = hdr->count, total = cnt < cap ? cnt : cap;
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long e = list[k];
        const int64_t i = (int64_t)(e >> 32), j = (int64_t)(e & 0xffffffffull);
        const BoxGeom<T> a = ga[i], b = gb[j];
        out[i * m + j] = loss_complete<T, KIND>(a, b);
    }
}

// ---------------------------------------------------------------- GIoU / DIoU backward
// Round 5: as in forward, the pairs that are apart take a kernel of their own (k_giou_grad_main: geom.hpp giou_rbox_apart_grad)
// and leave one bit per pair for the others; this kernel, given that bitmap (`only`, `wpr` words per row), computes exactly
// those -- a wavefront whose word is zero skips the row.  Without a bitmap it computes every pair (small matrices, DIoU).
template <typename T, int KIND>
__global__ __launch_bounds__(kCols) void k_loss_iou_grad(const T *__restrict__ b1, int64_t n, const T *__restrict__ b2, int64_t m,
                                                         const T *__restrict__ grad, T *g1, T *g2,
                                                         const unsigned long long *__restrict__ only = nullptr, int64_t wpr = 0,
                                                         int tile_rows = kRows)
{
    __shared__ RowBox<T> rows[kRows];
    const int64_t i0 = (int64_t)blockIdx.y * tile_rows, j = (int64_t)blockIdx.x * kCols + threadIdx.x;
    const int nrows = (int)((n - i0) < tile_rows ? (n - i0) : tile_rows);
    const int lane = threadIdx.x & (kWave - 1);
    const bool active = j < m, wave_in = (j & ~(int64_t)63) < m;
    // the tile's words of the bitmap, one row per lane, before anything else waits (a load per row in the loop is a round trip per row)
    unsigned long long words = ~0ull;
    if (only) words = (lane < nrows && wave_in) ? only[(i0 + lane) * wpr + (j >> 6)] : 0ull;
    if (only && !__any(words != 0)) {                  // nothing left for this wavefront: it still has to load its rows for the others
        if (threadIdx.x < nrows) rows[threadIdx.x] = load_row<T>(b1 + (i0 + threadIdx.x) * 5);
        __syncthreads();
        return;
    }
    if (threadIdx.x < nrows) rows[threadIdx.x] = load_row<T>(b1 + (i0 + threadIdx.x) * 5);
    __syncthreads();
    if (!wave_in) return;                              // a wavefront past the last column
    RowBox<T> c = load_row<T>(b2 + (active ? j : 0) * 5);
    T col[5] = {0, 0, 0, 0, 0};
    for (int r = 0; r < nrows; r++) {
        bool mine = true;
        if (only) {
            const unsigned long long word = __shfl(words, r, kWave);
            if (word == 0) continue;
            mine = (word >> lane) & 1ull;
        }
        const RowBox<T> a = rows[r];
        T ga[5] = {0, 0, 0, 0, 0}, gb[5] = {0, 0, 0, 0, 0};
        const T g = (active && mine) ? grad[(i0 + r) * m + j] : (T)0;
        if (g != 0) {
            loss_iou_rbox<T, KIND, true>(a.g, c.g, a.w, a.h, c.w, c.h, ga, gb);
#pragma unroll
            for (int k = 0; k < 5; k++) { ga[k] *= g; col[k] += g * gb[k]; }
        }
        const T s = wave_sum5<T>(ga, lane);
        if (lane < 5 && s != 0) atomicAdd(&g1[(i0 + r) * 5 + lane], s);
    }
    if (active)
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (col[k] != 0) atomicAdd(&g2[j * 5 + k], col[k]);
}

// The pairs the bitmap leaves (those that need the clip / the tie rules), COMPACTED: a wavefront walks its 64 x 64 part of the
// tile row by row and queues the marked pairs until 64 are together (or the rows end), then every lane takes one -- row box from
// LDS, column box from LDS (the wavefront staged its 64), weight gathered -- through the complete routine.  The row-by-row form
// (k_loss_iou_grad with `only`) spent the routine's ~3000 instructions on every row that had ONE marked pair: 2 % of the rows of
// a sparse 6 k x 6 k matrix, 246 us next to the 760 of all other pairs.  Gradients go to LDS accumulators (ds_add, native in fp64
// too) -- per row of the tile and wavefront, per column -- and from there to memory once, side by side.
template <typename T, int KIND>
__global__ __launch_bounds__(kCols) void k_loss_grad_rest(const T *__restrict__ b1, int64_t n, const T *__restrict__ b2, int64_t m,
                                                          const T *__restrict__ grad, T *g1, T *g2,
                                                          const unsigned long long *__restrict__ only, int64_t wpr, int tile_rows)
{
    __shared__ RowBox<T> rows[kRows];
    __shared__ RowBox<T> cols[kCols];
    __shared__ T racc[kCols / 64][kRows][5];
    __shared__ T cacc[kCols][5];
    __shared__ unsigned short queue[kCols / 64][64];
    const int64_t i0 = (int64_t)blockIdx.y * tile_rows, jb = (int64_t)blockIdx.x * kCols, j = jb + threadIdx.x;
    const int nrows = (int)((n - i0) < tile_rows ? (n - i0) : tile_rows);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const bool active = j < m, wave_in = (j & ~(int64_t)63) < m;
    const unsigned long long words = (lane < nrows && wave_in) ? only[(i0 + lane) * wpr + (j >> 6)] : 0ull;
    const bool any = __any(words != 0);
    if (!__syncthreads_or(any)) return;                 // nothing marked in the whole tile (the usual case of a sparse scene)
    if (threadIdx.x < nrows) rows[threadIdx.x] = load_row<T>(b1 + (i0 + threadIdx.x) * 5);
    if (any) {
        if (active) cols[threadIdx.x] = load_row<T>(b2 + j * 5);
#pragma unroll
        for (int k = 0; k < 5; k++) { racc[wave][lane][k] = 0; cacc[threadIdx.x][k] = 0; }
    }
    __syncthreads();
    if (!any) return;
    unsigned short *q = queue[wave];
    unsigned int wn = 0;
    auto process = [&]() {
        __builtin_amdgcn_wave_barrier();
        if (lane < (int)wn) {
            const unsigned int e = q[lane], r = e >> 8, c = e & 63u;
            const T g = grad[(i0 + r) * m + (jb + wave * 64 + c)];
            if (g != 0) {
                const RowBox<T> a = rows[r], cb = cols[wave * 64 + c];
                T ga[5], gb[5];
                loss_iou_rbox<T, KIND, true>(a.g, cb.g, a.w, a.h, cb.w, cb.h, ga, gb);
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    if (ga[k] != 0) atomicAdd(&racc[wave][r][k], g * ga[k]);
                    if (gb[k] != 0) atomicAdd(&cacc[wave * 64 + c][k], g * gb[k]);
                }
            }
        }
        wn = 0;
        __builtin_amdgcn_wave_barrier();
    };
    for (int r = 0; r < nrows; r++) {
        const unsigned long long word = __shfl(words, r, kWave);
        if (word == 0) continue;
        const unsigned int cnt = (unsigned int)__popcll(word);
        if (wn + cnt > 64u) process();
        if ((word >> lane) & 1ull) q[wn + __popcll(word & ((1ull << lane) - 1))] = (unsigned short)((r << 8) | lane);
        wn += cnt;
    }
    if (wn) process();
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const T vr = lane < nrows ? racc[wave][lane][k] : (T)0, vc = cacc[threadIdx.x][k];
        if (vr != 0) atomicAdd(&g1[(i0 + lane) * 5 + k], vr);
        if (active && vc != 0) atomicAdd(&g2[j * 5 + k], vc);
    }
}

// the pairs that are apart (GIoU): gradient by giou_rbox_apart_grad, one bit per pair left for k_loss_iou_grad
template <typename T> struct GradRow { BoxGeom<T> g; HullPre<T> h; T w, hgt, iw, ih; };

template <typename T, int KIND>
__global__ __launch_bounds__(kCols) __attribute__((amdgpu_waves_per_eu(3))) void k_giou_grad_main(
    const BoxGeom<T> *__restrict__ ga, const HullPre<T> *__restrict__ ha, const T *__restrict__ b1, int64_t n,
    const BoxGeom<T> *__restrict__ gb, const HullPre<T> *__restrict__ hb, const T *__restrict__ b2, int64_t m, const T *__restrict__ grad, T *g1,
    T *g2, unsigned long long *__restrict__ bitmap, int64_t wpr, int tile_rows)
{
    __shared__ GradRow<T> rows[kRows];
    const int64_t i0 = (int64_t)blockIdx.y * tile_rows, j = (int64_t)blockIdx.x * kCols + threadIdx.x;
    const int nrows = (int)((n - i0) < tile_rows ? (n - i0) : tile_rows);
    if (threadIdx.x < nrows) {
        GradRow<T> &r = rows[threadIdx.x];             // (field by field: a local copy of the struct went through scratch)
        r.g = ga[i0 + threadIdx.x]; r.h = ha[i0 + threadIdx.x];
        const T w = b1[(i0 + threadIdx.x) * 5 + 2], hgt = b1[(i0 + threadIdx.x) * 5 + 3];
        r.w = w; r.hgt = hgt; r.iw = (T)1 / w; r.ih = (T)1 / hgt;
    }
    __syncthreads();
    if ((j & ~(int64_t)63) >= m) return;               // a wavefront past the last column
    const bool valid = j < m;
    const int64_t jc = valid ? j : m - 1;
    const BoxGeom<T> c = gb[jc];
    const HullPre<T> hc = hb[jc];
    const T cw = b2[jc * 5 + 2], chgt = b2[jc * 5 + 3], ciw = (T)1 / cw, cih = (T)1 / chgt;
    T col[5] = {0, 0, 0, 0, 0};
    const int lane = threadIdx.x & (kWave - 1);
    const T *gp = grad + i0 * m + jc;
    unsigned long long *bw = bitmap + i0 * wpr + (j >> 6);
    // the rows' weights arrive four rows ahead of their use (one row ahead, the round trip still showed: 2.2 TB/s on weights that
    // are zero almost everywhere)
    T ring[4];
#pragma unroll
    for (int q = 0; q < 4; q++) ring[q] = (valid && q < nrows) ? gp[(int64_t)q * m] : (T)0;
    gp += 4 * m;
    for (int r = 0; r < nrows; r++, bw += wpr, gp += m) {
        const T g = ring[0];
        ring[0] = ring[1]; ring[1] = ring[2]; ring[2] = ring[3];
        ring[3] = (valid && r + 4 < nrows) ? *gp : (T)0;
        if (!__any(g != 0)) {                          // nothing arrives for this stretch of the row (a loss on selected pairs)
            if (lane == 0) *bw = 0;
            continue;
        }
        T da[5], db[5];
        bool defer;
        loss_rbox_apart_grad<T, KIND>(rows[r].g, rows[r].h, rows[r].w, rows[r].hgt, rows[r].iw, rows[r].ih, c, hc, cw, chgt, ciw, cih, da, db, defer);
        const bool take = (g != 0) & !defer;
        const unsigned long long left = __ballot((g != 0) & defer);
        if (lane == 0) *bw = left;
        T wa[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            col[k] += take ? g * db[k] : (T)0;
            wa[k] = take ? g * da[k] : (T)0;
        }
        const T s = wave_sum5<T>(wa, lane);
        if (lane < 5 && s != 0) atomicAdd(&g1[(i0 + r) * 5 + lane], s);
    }
    if (valid)
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (col[k] != 0) atomicAdd(&g2[j * 5 + k], col[k]);
}

// ---------------------------------------------------------------- bookkeeping outputs (nx, xflags, nm, mflags, far)
// Sutherland-Hodgman with the origin of every vertex, Andrew's monotone chain for the hull's vertex list.  These ARE
// dynamically indexed vertex lists (scratch memory): they are what the reference's autograd saves between forward and
// backward, nothing on a hot path -- this library's backward recomputes the clip analytically and ignores them.
template <typename T> struct Pt { T x, y; };

template <typename T>
__device__ int clip_flags(const Pt<T> (&s)[4], const Pt<T> (&c)[4], uint8_t *flags)
{
    Pt<T> buf[2][16];
    uint8_t vf[2][16], ef[2][16];
    int n = 4, cur = 0;
    for (int k = 0; k < 4; k++) { buf[0][k] = s[k]; vf[0][k] = (uint8_t)k; ef[0][k] = (uint8_t)k; }
    for (int e = 0; e < 4 && n > 0; e++) {
        const Pt<T> a = c[e], b = c[(e + 1) & 3];
        const T ex = b.x - a.x, ey = b.y - a.y;
        int mm = 0;
        const int nxt = cur ^ 1;
        for (int k = 0; k < n; k++) {
            const Pt<T> p = buf[cur][k], q = buf[cur][(k + 1) % n];
            const uint8_t o = ef[cur][k];
            const T dp = ex * (p.y - a.y) - ey * (p.x - a.x), dq = ex * (q.y - a.y) - ey * (q.x - a.x);
            const bool pin = dp >= 0, qin = dq >= 0;
            if (pin) { buf[nxt][mm] = p; vf[nxt][mm] = vf[cur][k]; ef[nxt][mm] = o; mm++; }
            if (pin != qin) {
                const T t = dp / (dp - dq);
                Pt<T> x = {p.x + t * (q.x - p.x), p.y + t * (q.y - p.y)};
                uint8_t f;
                if (o < 4) f = (uint8_t)(0x20 | (o << 2) | e);
                else { const int e2 = o - 4; f = (uint8_t)(0x10 | (((e2 + 1) & 3) == e ? e : e2)); }
                buf[nxt][mm] = x; vf[nxt][mm] = f; ef[nxt][mm] = pin ? (uint8_t)(4 + e) : o; mm++;
            }
        }
        n = mm; cur = nxt;
    }
    if (n < 3) n = 0;                                  // no area: no polygon
    if (n > 8) n = 8;
    for (int k = 0; k < 8; k++) flags[k] = k < n ? vf[cur][k] : (uint8_t)0xff;
    return n;
}

template <typename T>
__device__ int hull_flags(const Pt<T> (&p)[8], uint8_t *hull)
{
    int idx[8];
    for (int k = 0; k < 8; k++) idx[k] = k;
    for (int a = 1; a < 8; a++) {
        const int v = idx[a];
        int b = a - 1;
        while (b >= 0 && (p[idx[b]].x > p[v].x || (p[idx[b]].x == p[v].x && p[idx[b]].y > p[v].y))) { idx[b + 1] = idx[b]; b--; }
        idx[b + 1] = v;
    }
    int st[17], mm = 0;
    for (int k = 0; k < 8; k++) {
        while (mm >= 2) {
            const Pt<T> a = p[st[mm - 2]], b = p[st[mm - 1]], c = p[idx[k]];
            if ((b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x) <= 0) mm--; else break;
        }
        st[mm++] = idx[k];
    }
    const int lower = mm + 1;
    for (int k = 6; k >= 0; k--) {
        while (mm >= lower) {
            const Pt<T> a = p[st[mm - 2]], b = p[st[mm - 1]], c = p[idx[k]];
            if ((b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x) <= 0) mm--; else break;
        }
        st[mm++] = idx[k];
    }
    mm--;
    if (mm < 0) mm = 0;
    if (mm > 8) mm = 8;
    for (int k = 0; k < 8; k++) hull[k] = k < mm ? (uint8_t)st[k] : (uint8_t)0xff;
    return mm;
}

// corners in the reference's formula order (utils.h:19 via dgal::poly2_from_xywhr; same expressions as the oracle)
template <typename T> __device__ void quad_pts(const T *b, Pt<T> (&q)[4])
{
    T s, c;
    d3d_sincos(b[4], &s, &c);
    const T dxs = b[2] * s / 2, dxc = b[2] * c / 2, dys = b[3] * s / 2, dyc = b[3] * c / 2;
    q[0] = {b[0] - dxc + dys, b[1] - dxs - dyc}; q[1] = {b[0] + dxc + dys, b[1] + dxs - dyc};
    q[2] = {b[0] + dxc - dys, b[1] + dxs + dyc}; q[3] = {b[0] - dxc - dys, b[1] - dxs + dyc};
}

template <typename T>
__global__ __launch_bounds__(256) void k_pair_flags(const T *__restrict__ b1, int64_t n, const T *__restrict__ b2, int64_t m,
                                                    uint8_t *nx, uint8_t *xflags, uint8_t *nm, uint8_t *mflags, uint8_t *far)
{
    const int64_t total = n * m;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / m, j = e - i * m;
        Pt<T> qa[4], qb[4], p8[8];
        quad_pts<T>(b1 + i * 5, qa);
        quad_pts<T>(b2 + j * 5, qb);
        for (int k = 0; k < 4; k++) { p8[k] = qa[k]; p8[4 + k] = qb[k]; }
        uint8_t fl[8];
        if (nx || xflags) {
            const int k = clip_flags<T>(qa, qb, fl);
            if (nx) nx[e] = (uint8_t)k;
            if (xflags) *reinterpret_cast<uint2 *>(xflags + e * 8) = *reinterpret_cast<const uint2 *>(fl);
        }
        if (nm || mflags) {
            const int k = hull_flags<T>(p8, fl);
            if (nm) nm[e] = (uint8_t)k;
            if (mflags) *reinterpret_cast<uint2 *>(mflags + e * 8) = *reinterpret_cast<const uint2 *>(fl);
        }
        if (far) {
            T best = -1;
            int f1 = 0, f2 = 1;
            for (int x = 0; x < 8; x++)
                for (int y = x + 1; y < 8; y++) {
                    const T dx = p8[x].x - p8[y].x, dy = p8[x].y - p8[y].y, d = dx * dx + dy * dy;
                    if (d > best) { best = d; f1 = x; f2 = y; }
                }
            far[e * 2] = (uint8_t)f1; far[e * 2 + 1] = (uint8_t)f2;
        }
    }
}

// ---------------------------------------------------------------- point-to-box distance
// dist[m, n] (box-major, like the reference: dist.cpp:39): tile = 64 boxes (LDS) x 256 * K points; a lane owns K consecutive
// points, so a row of the tile leaves as K * sizeof(T)-byte and K-byte stores per lane (K = 4 where n % 4 == 0 keeps the rows
// aligned: 16-byte / 4-byte stores for fp32 instead of 4-byte / 1-byte ones)
template <typename T, int K>
__global__ __launch_bounds__(kCols) void k_pdist(const T *__restrict__ points, int64_t n, const T *__restrict__ boxes, int64_t m,
                                                 T *__restrict__ dist, uint8_t *__restrict__ iedge)
{
    __shared__ RowBox<T> rows[kRows];
    const int64_t i0 = (int64_t)blockIdx.y * kRows, j0 = ((int64_t)blockIdx.x * kCols + threadIdx.x) * K;
    const int nrows = (int)((m - i0) < kRows ? (m - i0) : kRows);
    if (threadIdx.x < nrows) rows[threadIdx.x] = load_row<T>(boxes + (i0 + threadIdx.x) * 5);
    __syncthreads();
    if (j0 >= n) return;                   // n % K == 0 (host-checked): a lane's K points are all valid or all not
    T px[K], py[K];
#pragma unroll
    for (int k = 0; k < K; k++) { px[k] = points[(j0 + k) * 2]; py[k] = points[(j0 + k) * 2 + 1]; }
    T gp[2], gb[5];
    for (int r = 0; r < nrows; r++) {
        const RowBox<T> b = rows[r];
        T d[K];
        uint8_t f[K];
        const bool regular = b.w > 0 && b.h > 0;           // (wave-uniform: every lane works on box r)
#pragma unroll
        for (int k = 0; k < K; k++) {
            int feat;
            if (regular) d[k] = point_box_distance_local<T>(b.g.cx, b.g.cy, b.c, b.s, b.w / 2, b.h / 2, px[k], py[k], feat);
            else d[k] = point_box_distance<T, false>(b.g, b.w, b.h, px[k], py[k], feat, gp, gb);
            f[k] = (uint8_t)feat;
        }
        T *o = dist + (i0 + r) * n + j0;
        if constexpr (K == 4) {
            typedef T vecd __attribute__((ext_vector_type(4)));
            const vecd x = {d[0], d[1], d[2], d[3]};
            __builtin_nontemporal_store(x, reinterpret_cast<vecd *>(o));
            if (iedge)
                __builtin_nontemporal_store((unsigned int)f[0] | ((unsigned int)f[1] << 8) | ((unsigned int)f[2] << 16) | ((unsigned int)f[3] << 24),
                                            reinterpret_cast<unsigned int *>(iedge + (i0 + r) * n + j0));
        } else {
            o[0] = d[0];
            if (iedge) iedge[(i0 + r) * n + j0] = f[0];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kCols) void k_pdist_grad(const T *__restrict__ points, int64_t n, const T *__restrict__ boxes, int64_t m,
                                                      const T *__restrict__ grad, T *gboxes, T *gpoints)
{
    __shared__ RowBox<T> rows[kRows];
    const int64_t i0 = (int64_t)blockIdx.y * kRows, j = (int64_t)blockIdx.x * kCols + threadIdx.x;
    const int nrows = (int)((m - i0) < kRows ? (m - i0) : kRows);
    if (threadIdx.x < nrows) rows[threadIdx.x] = load_row<T>(boxes + (i0 + threadIdx.x) * 5);
    __syncthreads();
    const bool active = j < n;
    const T px = active ? points[j * 2] : (T)0, py = active ? points[j * 2 + 1] : (T)0;
    T accp[2] = {0, 0};
    const int lane = threadIdx.x & (kWave - 1);
    for (int r = 0; r < nrows; r++) {
        const RowBox<T> b = rows[r];
        T gp[2] = {0, 0}, gb[5] = {0, 0, 0, 0, 0};
        const T g = active ? grad[(i0 + r) * n + j] : (T)0;
        if (g != 0) {
            int feat;
            point_box_distance<T, true>(b.g, b.w, b.h, px, py, feat, gp, gb);
            accp[0] += g * gp[0]; accp[1] += g * gp[1];
#pragma unroll
            for (int k = 0; k < 5; k++) gb[k] *= g;
        }
        const T s = wave_sum5<T>(gb, lane);
        if (lane < 5 && s != 0) atomicAdd(&gboxes[(i0 + r) * 5 + lane], s);
    }
    if (active) {
        if (accp[0] != 0) atomicAdd(&gpoints[j * 2], accp[0]);
        if (accp[1] != 0) atomicAdd(&gpoints[j * 2 + 1], accp[1]);
    }
}

template <typename T>
int loss_forward(const T *b1, int64_t n, const T *b2, int64_t m, int kind, T *out, void *ws, size_t ws_bytes, unsigned long long list_cap,
                 hipStream_t st)
{
    const dim3 grid((unsigned)d3d_divup(m, kCols), (unsigned)d3d_divup(n, kRows));
    const unsigned int *no_gate = nullptr;
    if (ws && (int64_t)n * m > 65536) {                           // a matrix: two kernels (above)
        WsCarver w(ws, ws_bytes);
        BoxGeom<T> *ga = w.take<BoxGeom<T>>(n);
        BoxGeom<T> *gb = w.take<BoxGeom<T>>(m);
        HullPre<T> *ha = w.take<HullPre<T>>(n);
        HullPre<T> *hb = w.take<HullPre<T>>(m);
        FixList *hdr = w.take<FixList>(1);
        unsigned long long *list = w.take<unsigned long long>(list_cap);
        if (w.ok() && list_cap > 0) {
            const unsigned int *redo = &hdr->overflow;
            D3D_LAUNCH("k_giou_geom", k_giou_geom<T>, dim3((unsigned)d3d_divup(n + m, 256)), dim3(256), 0, st, b1, n, b2, m, ga, ha, gb, hb, hdr);
#define D3D_LOSS_TWO(K, NAME)                                                                                                            \
    D3D_LAUNCH(NAME "_main", (k_giou_main<T, K>), grid, dim3(kCols), 0, st, (const BoxGeom<T> *)ga, (const HullPre<T> *)ha, n,             \
               (const BoxGeom<T> *)gb, (const HullPre<T> *)hb, m, out, hdr, list, list_cap);                                              \
    D3D_LAUNCH(NAME "_fix", (k_giou_fix<T, K>), dim3(256 * 8), dim3(256), 0, st, ga, gb, m, out, hdr, list, list_cap);                      \
    D3D_LAUNCH("k_loss_iou<redo>", (k_loss_iou<T, K>), grid, dim3(kCols), 0, st, b1, n, b2, m, out, redo)
            if (kind == 0) { D3D_LOSS_TWO(0, "k_giou"); } else { D3D_LOSS_TWO(1, "k_diou"); }
#undef D3D_LOSS_TWO
            return D3D_OK;
        }
    }
    if (kind == 0) D3D_LAUNCH("k_loss_iou<giou>", (k_loss_iou<T, 0>), grid, dim3(kCols), 0, st, b1, n, b2, m, out, no_gate);
    else D3D_LAUNCH("k_loss_iou<diou>", (k_loss_iou<T, 1>), grid, dim3(kCols), 0, st, b1, n, b2, m, out, no_gate);
    return D3D_OK;
}

template <typename T>
int loss_backward(const T *b1, int64_t n, const T *b2, int64_t m, const T *grad, int kind, T *g1, T *g2, void *ws, size_t ws_bytes, hipStream_t st)
{
    D3D_HIP_CHECK(hipMemsetAsync(g1, 0, (size_t)n * 5 * sizeof(T), st));
    D3D_HIP_CHECK(hipMemsetAsync(g2, 0, (size_t)m * 5 * sizeof(T), st));
    const dim3 grid((unsigned)d3d_divup(m, kCols), (unsigned)d3d_divup(n, kRows));
    const unsigned long long *all = nullptr;
    if (ws && (int64_t)n * m > 65536) {                           // a matrix: the pairs apart first, the others by their bitmap
        const int64_t wpr = d3d_divup(m, 64);
        // rows per workgroup: 64, or fewer while the launch would not give every SIMD of the chip two wavefronts (2 k x 2 k boxes at
        // 64 rows: 256 workgroups = one wavefront per SIMD, each walking its rows' round trips alone)
        int tr = kRows;
        while (tr > 8 && d3d_divup(m, kCols) * d3d_divup(n, tr) < 2048) tr >>= 1;
        const dim3 tgrid((unsigned)d3d_divup(m, kCols), (unsigned)d3d_divup(n, tr));
        WsCarver w(ws, ws_bytes);
        BoxGeom<T> *ga = w.take<BoxGeom<T>>(n);
        BoxGeom<T> *gb = w.take<BoxGeom<T>>(m);
        HullPre<T> *ha = w.take<HullPre<T>>(n);
        HullPre<T> *hb = w.take<HullPre<T>>(m);
        FixList *hdr = w.take<FixList>(1);
        unsigned long long *bitmap = w.take<unsigned long long>((size_t)n * (size_t)wpr);
        if (w.ok()) {
            D3D_LAUNCH("k_giou_geom", k_giou_geom<T>, dim3((unsigned)d3d_divup(n + m, 256)), dim3(256), 0, st, b1, n, b2, m, ga, ha, gb, hb, hdr);
#define D3D_GRAD_TWO(K, NAME)                                                                                                            \
    D3D_LAUNCH(NAME "_grad_main", (k_giou_grad_main<T, K>), tgrid, dim3(kCols), 0, st, (const BoxGeom<T> *)ga, (const HullPre<T> *)ha, b1, n, \
               (const BoxGeom<T> *)gb, (const HullPre<T> *)hb, b2, m, grad, g1, g2, bitmap, wpr, tr);                                     \
    D3D_LAUNCH("k_loss_grad_rest", (k_loss_grad_rest<T, K>), tgrid, dim3(kCols), 0, st, b1, n, b2, m, grad, g1, g2,                       \
               (const unsigned long long *)bitmap, wpr, tr)
            if (kind == 0) { D3D_GRAD_TWO(0, "k_giou"); } else { D3D_GRAD_TWO(1, "k_diou"); }
#undef D3D_GRAD_TWO
            return D3D_OK;
        }
    }
    if (kind == 0) D3D_LAUNCH("k_loss_iou_grad<giou>", (k_loss_iou_grad<T, 0>), grid, dim3(kCols), 0, st, b1, n, b2, m, grad, g1, g2, all, (int64_t)0, (int)kRows);
    else D3D_LAUNCH("k_loss_iou_grad<diou>", (k_loss_iou_grad<T, 1>), grid, dim3(kCols), 0, st, b1, n, b2, m, grad, g1, g2, all, (int64_t)0, (int)kRows);
    return D3D_OK;
}

}  // namespace

// called by d3d_iou2d_forward / d3d_iou2d_backward (box.hip) for iou_type GRBOX / DRBOX
// workspace (optional; GIoU only): geometry of both box sets, the list header, `list_cap` 8-byte entries -- a prefix of what
// d3d_iou2d_workspace_bytes(n, m, dtype) sizes for the candidate list of RBOX (box.hip)
int d3d_internal_loss_iou_forward(const void *b1, int64_t n, const void *b2, int64_t m, int kind, int dtype, void *out, void *ws,
                                  size_t ws_bytes, unsigned long long list_cap, hipStream_t st)
{
    if (d3d_divup(n, kRows) > 65535) return D3D_ERR_BAD_ARG;
    if (dtype == D3D_F64)
        return loss_forward<double>((const double *)b1, n, (const double *)b2, m, kind, (double *)out, ws, ws_bytes, list_cap, st);
    return loss_forward<float>((const float *)b1, n, (const float *)b2, m, kind, (float *)out, ws, ws_bytes, list_cap, st);
}

int d3d_internal_loss_iou_backward(const void *b1, int64_t n, const void *b2, int64_t m, const void *grad, int kind, int dtype,
                                   void *g1, void *g2, void *ws, size_t ws_bytes, hipStream_t st)
{
    if (d3d_divup(n, kRows) > 65535) return D3D_ERR_BAD_ARG;
    if (dtype == D3D_F64)
        return loss_backward<double>((const double *)b1, n, (const double *)b2, m, (const double *)grad, kind, (double *)g1, (double *)g2, ws,
                                     ws_bytes, st);
    return loss_backward<float>((const float *)b1, n, (const float *)b2, m, (const float *)grad, kind, (float *)g1, (float *)g2, ws, ws_bytes, st);
}

extern "C" int d3d_iou2dr_flags(const void *boxes1, int64_t n, const void *boxes2, int64_t m, int32_t dtype, uint8_t *nx,
                                uint8_t *xflags, uint8_t *nm, uint8_t *mflags, uint8_t *far, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || m < 0 || (dtype != D3D_F32 && dtype != D3D_F64)) return D3D_ERR_BAD_ARG;
    if (n == 0 || m == 0) return D3D_OK;
    if (!boxes1 || !boxes2) return D3D_ERR_BAD_ARG;
    if (((reinterpret_cast<uintptr_t>(xflags) | reinterpret_cast<uintptr_t>(mflags)) & 7)) return D3D_ERR_BAD_ARG;
    const unsigned grid = (unsigned)(d3d_divup(n * m, 256) < 65536 ? d3d_divup(n * m, 256) : 65536);
    if (dtype == D3D_F64)
        D3D_LAUNCH("k_pair_flags", k_pair_flags<double>, dim3(grid), dim3(256), 0, st, (const double *)boxes1, n, (const double *)boxes2, m,
                   nx, xflags, nm, mflags, far);
    else
        D3D_LAUNCH("k_pair_flags", k_pair_flags<float>, dim3(grid), dim3(256), 0, st, (const float *)boxes1, n, (const float *)boxes2, m,
                   nx, xflags, nm, mflags, far);
    return D3D_OK;
}

extern "C" int d3d_pdist2dr_forward(const void *points, int64_t n, const void *boxes, int64_t m, int32_t dtype, void *dist,
                                    uint8_t *iedge, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || m < 0 || (dtype != D3D_F32 && dtype != D3D_F64)) return D3D_ERR_BAD_ARG;
    if (n == 0 || m == 0) return D3D_OK;
    if (!points || !boxes || !dist || d3d_divup(m, kRows) > 65535) return D3D_ERR_BAD_ARG;
    const bool k4 = n % 4 == 0 && (reinterpret_cast<uintptr_t>(dist) & 31) == 0 && (reinterpret_cast<uintptr_t>(iedge) & 3) == 0;
    const dim3 grid((unsigned)d3d_divup(n, (int64_t)kCols * (k4 ? 4 : 1)), (unsigned)d3d_divup(m, kRows));
#define D3D_PDIST(T, K) D3D_LAUNCH("k_pdist", (k_pdist<T, K>), grid, dim3(kCols), 0, st, (const T *)points, n, (const T *)boxes, m, (T *)dist, iedge)
    if (dtype == D3D_F64) { if (k4) D3D_PDIST(double, 4); else D3D_PDIST(double, 1); }
    else { if (k4) D3D_PDIST(float, 4); else D3D_PDIST(float, 1); }
#undef D3D_PDIST
    return D3D_OK;
}

extern "C" int d3d_pdist2dr_backward(const void *points, int64_t n, const void *boxes, int64_t m, const void *grad, int32_t dtype,
                                     void *grad_boxes, void *grad_points, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || m < 0 || (dtype != D3D_F32 && dtype != D3D_F64)) return D3D_ERR_BAD_ARG;
    const size_t es = dtype == D3D_F64 ? 8 : 4;
    if (m > 0) { if (!grad_boxes) return D3D_ERR_BAD_ARG; D3D_HIP_CHECK(hipMemsetAsync(grad_boxes, 0, (size_t)m * 5 * es, st)); }
    if (n > 0) { if (!grad_points) return D3D_ERR_BAD_ARG; D3D_HIP_CHECK(hipMemsetAsync(grad_points, 0, (size_t)n * 2 * es, st)); }
    if (n == 0 || m == 0) return D3D_OK;
    if (!points || !boxes || !grad || d3d_divup(m, kRows) > 65535) return D3D_ERR_BAD_ARG;
    const dim3 grid((unsigned)d3d_divup(n, kCols), (unsigned)d3d_divup(m, kRows));
    if (dtype == D3D_F64)
        D3D_LAUNCH("k_pdist_grad", k_pdist_grad<double>, grid, dim3(kCols), 0, st, (const double *)points, n, (const double *)boxes, m,
                   (const double *)grad, (double *)grad_boxes, (double *)grad_points);
    else
        D3D_LAUNCH("k_pdist_grad", k_pdist_grad<float>, grid, dim3(kCols), 0, st, (const float *)points, n, (const float *)boxes, m,
                   (const float *)grad, (float *)grad_boxes, (float *)grad_points);
    return D3D_OK;
}
