// crop.hip -- points inside 3D boxes: Target3DArray.crop_points and Target3DArray.paint_label of the
// reference (d3d/abstraction.pyx:308-324, 654-687), whose per-pair test is box3dr_contains (d3d/dgal_wrap.h:6-19): the
// CLOSED z interval [z - lz/2, z + lz/2] in fp32, then the bounding box of the rotated rectangle, then the rectangle.
// The reference loops boxes x points on one core and, for paint_label, materialises the bool[M,N] mask first.
#include "common.hpp"
#include "geom.hpp"

namespace {

constexpr int kBoxTile = 64;        // boxes staged in LDS at a time

struct Box3 {
    BoxGeom<float> g;
    float zlo, zhi;
};

// a box row is (x, y, z, lx, ly, lz, rz) at boxes + i * stride + offset  ([M,7] arrays: stride 7, offset 0; the [n,9] rows of
// Target3DArray.to_numpy -- label, score, x, y, z, lx, ly, lz, yaw: stride 9, offset 2)
__device__ __forceinline__ Box3 load_box3(const float *__restrict__ b)
{
    Box3 r;
    r.g = make_geom<float>(b[0], b[1], b[3], b[4], b[6]);
    r.zhi = b[2] + b[5] / 2;        // dgal_wrap.h:12, fp32
    r.zlo = b[2] - b[5] / 2;
    return r;
}

// the cheap part of the test: z interval and the rectangle's bounding box (a point that fails it is outside)
__device__ __forceinline__ bool near3(const Box3 &b, float x, float y, float z)
{
    if (z > b.zhi || z < b.zlo) return false;                                   // dgal_wrap.h:12-13 (NaN z: inside, as there)
    const BoxGeom<float> &g = b.g;
    return x >= g.xmin && x <= g.xmax && y >= g.ymin && y <= g.ymax;             // :14-15
}

__device__ __forceinline__ bool contains3(const Box3 &b, float x, float y, float z)
{
    if (z > b.zhi || z < b.zlo) return false;                                   // dgal_wrap.h:12-13 (NaN z: inside, as there)
    return quad_contains<float>(b.g, x, y);                                     // :14-17
}

// indicators[i, j] = box i contains point j.  Lane = 4 consecutive points -> one 32-bit store per box row.
__global__ __launch_bounds__(256) void k_crop3dr(const float *__restrict__ points, int64_t n, int pstride,
                                                 const float *__restrict__ boxes, int64_t m, int bstride, int boff,
                                                 uint8_t *__restrict__ out)
{
    __shared__ Box3 rows[kBoxTile];
    const int64_t i0 = (int64_t)blockIdx.y * kBoxTile;
    const int nrows = (int)((m - i0) < kBoxTile ? (m - i0) : kBoxTile);
    if (threadIdx.x < nrows) rows[threadIdx.x] = load_box3(boxes + (i0 + threadIdx.x) * bstride + boff);
    const int64_t j0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    float px[4], py[4], pz[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const bool ok = j0 + k < n;
        px[k] = ok ? points[(j0 + k) * pstride] : 0.f;
        py[k] = ok ? points[(j0 + k) * pstride + 1] : 0.f;
        pz[k] = ok ? points[(j0 + k) * pstride + 2] : 0.f;
    }
    __syncthreads();
    if (j0 >= n) return;
    const bool vec = (n % 4 == 0);
    for (int r = 0; r < nrows; r++) {
        const Box3 b = rows[r];
        uint32_t word = 0;           // (a bounding-box cull per wavefront ahead of this was slower: 704 -> 800 us, the kernel is
#pragma unroll                       //  bound by its 2 GB of stores, the tests run in their shadow)
        for (int k = 0; k < 4; k++) word |= (contains3(b, px[k], py[k], pz[k]) ? 1u : 0u) << (8 * k);
        uint8_t *dst = out + (i0 + r) * n + j0;
        if (vec) *reinterpret_cast<uint32_t *>(dst) = word;
        else
            for (int k = 0; k < 4 && j0 + k < n; k++) dst[k] = (uint8_t)((word >> (8 * k)) & 1u);
    }
}

// paint_label (abstraction.pyx:662-687): idarr[j] = ib + 1 for the LOWEST box index ib whose box contains point j and whose
// class equals the point's semantic label (the reference paints boxes M-1 .. 0, so the lowest index -- the best score of a
// sorted array -- is written last), 0 if none.  One lane per point walks the boxes in ascending order and stops at its first
// hit; the bool[M,N] mask of the reference never exists.  uint16 like there: ib + 1 wraps beyond 65535 boxes.
constexpr int kPaintPts = 4;        // points per lane: a box's record is read from LDS once for four tests
__global__ __launch_bounds__(256) void k_paint_label(const float *__restrict__ points, int64_t n, int pstride,
                                                     const uint8_t *__restrict__ semantics, const float *__restrict__ boxes,
                                                     int64_t m, int bstride, int boff, const uint8_t *__restrict__ labels,
                                                     uint16_t *__restrict__ idarr)
{
    __shared__ Box3 rows[kBoxTile];
    __shared__ uint8_t cls[kBoxTile];
    // lane = points j0 + k * 256 (k < kPaintPts): consecutive lanes read consecutive points
    const int64_t j0 = (int64_t)blockIdx.x * (256 * kPaintPts) + threadIdx.x;
    float x[kPaintPts], y[kPaintPts], z[kPaintPts];
    uint8_t sem[kPaintPts];
    uint16_t id[kPaintPts];
    bool found[kPaintPts];
#pragma unroll
    for (int k = 0; k < kPaintPts; k++) {
        const int64_t j = j0 + (int64_t)k * 256;
        const bool ok = j < n;
        x[k] = ok ? points[j * pstride] : 0.f;
        y[k] = ok ? points[j * pstride + 1] : 0.f;
        z[k] = ok ? points[j * pstride + 2] : 0.f;
        sem[k] = ok ? semantics[j] : 0;
        id[k] = 0;
        found[k] = !ok;
    }
    for (int64_t i0 = 0; i0 < m; i0 += kBoxTile) {
        const int nrows = (int)((m - i0) < kBoxTile ? (m - i0) : kBoxTile);
        __syncthreads();
        if (threadIdx.x < nrows) {
            rows[threadIdx.x] = load_box3(boxes + (i0 + threadIdx.x) * bstride + boff);
            cls[threadIdx.x] = labels[i0 + threadIdx.x];
        }
        bool all = true;
#pragma unroll
        for (int k = 0; k < kPaintPts; k++) all = all && found[k];
        if (__syncthreads_and(all)) break;                     // every point of the workgroup is painted (also the tile barrier)
        // the class, the z interval and the bounding box first; the four half-planes only for the boxes that SOME lane of the
        // wavefront comes near (a wavefront's points lie anywhere in the scene: a few % of the (wavefront, box) pairs)
        for (int r = 0; r < nrows; r++) {
            const uint8_t c = cls[r];
            bool near[kPaintPts], any = false;
#pragma unroll
            for (int k = 0; k < kPaintPts; k++) {
                near[k] = !found[k] && c == sem[k] && near3(rows[r], x[k], y[k], z[k]);
                any = any || near[k];
            }
            if (__ballot(any) == 0) continue;
#pragma unroll
            for (int k = 0; k < kPaintPts; k++)
                if (near[k] && contains3(rows[r], x[k], y[k], z[k])) {
                    id[k] = (uint16_t)(i0 + r + 1);
                    found[k] = true;
                }
        }
    }
#pragma unroll
    for (int k = 0; k < kPaintPts; k++) {
        const int64_t j = j0 + (int64_t)k * 256;
        if (j < n) idarr[j] = id[k];
    }
}

// ---------------------------------------------------------------- boxes on a grid (round 4)
// Both operators above test every point against every box: 2 k boxes x 1 M points = 2e9 tests, 1.8 / 2.5 ms, bound by the
// vector ALUs (paint_label at 0.16 of what its traffic would allow).  A point can only lie in a box whose bounding box
// covers it, and a scene's boxes are small against the scene: every workgroup first sorts the boxes into a 32 x 32 grid over
// their common bounding range (in LDS: counts by LDS atomics, a scan, the lists; sin / cos of every box kept, so that a test
// rebuilds the box from its row without another sincos), then each of its 4096 points tests the boxes of ITS cell only --
// about (boxes covering a cell) tests instead of M.  The grid costs M box expansions per workgroup, what loading the boxes
// cost before.  The same closed tests on the same expressions (contains3): identical results.  Up to 4096 boxes and 8192
// registrations (a box larger than 64 cells, more registrations, a non-finite box: the workgroup tests every box per point,
// still without the [M, N] sweep's redundant geometry); more boxes: the kernels above.
constexpr int kGridN = 32, kGridCellsN = kGridN * kGridN, kGridList = 8192, kGridMaxBoxes = 4096, kGridBoxCells = 64;
constexpr int kGridThreads = 256, kGridPts = 16;          // points per lane

struct BoxGrid {            // in LDS
    float2 cs[kGridMaxBoxes];                   // (cos, sin) per box
    uint32_t cur[kGridCellsN];                  // counts, then fill cursors
    uint16_t start[kGridCellsN + 1];
    uint16_t list[kGridList];
    float red[4][kGridThreads / kWave];
    float ox, oy, ix, iy;                       // cell = (x - ox) * ix
    int n;                                      // cells per axis: 32, or 16 / 8 when the registrations outgrow the list
    uint32_t all;                               // 1: no usable grid -- test every box
};

__device__ __forceinline__ Box3 box3_cs(const float *__restrict__ b, float2 cs)
{
    Box3 r;
    r.g = make_geom_cs<float>(b[0], b[1], b[3], b[4], cs.x, cs.y);
    r.zhi = b[2] + b[5] / 2;
    r.zlo = b[2] - b[5] / 2;
    return r;
}

__device__ __forceinline__ int grid_axis(float x, float o, float inv, int n)
{
    const float t = fminf((x - o) * inv, (float)(n - 1));
    return t > 0.f ? (int)t : 0;                // (NaN -> 0)
}

// where (x, y) = columns 0, 1 of a box row keep their extents and their angle: 3-D rows (x, y, z, lx, ly, lz, rz), 2-D rows
// (x, y, w, h, r) of crop_2dr
struct BoxCols { int w, h, r; };
constexpr BoxCols kCols3{3, 4, 6}, kCols2{2, 3, 4};

// whole workgroup; m <= kGridMaxBoxes
__device__ void build_box_grid(BoxGrid &G, const float *__restrict__ boxes, int64_t m, int bstride, int boff, BoxCols cols = kCols3)
{
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    float lo_x = INFINITY, lo_y = INFINITY, hi_x = -INFINITY, hi_y = -INFINITY;
    bool bad = false;
    for (int i = threadIdx.x; i < (int)m; i += kGridThreads) {
        const float *b = boxes + (size_t)i * bstride + boff;
        float sn, cn;
        d3d_sincos(b[cols.r], &sn, &cn);
        G.cs[i] = make_float2(cn, sn);
        const BoxGeom<float> g = make_geom_cs<float>(b[0], b[1], b[cols.w], b[cols.h], cn, sn);
        if (!(g.xmin >= -3.0e38f && g.ymin >= -3.0e38f && g.xmax <= 3.0e38f && g.ymax <= 3.0e38f)) { bad = true; continue; }
        lo_x = fminf(lo_x, g.xmin); lo_y = fminf(lo_y, g.ymin); hi_x = fmaxf(hi_x, g.xmax); hi_y = fmaxf(hi_y, g.ymax);
    }
    for (int c = threadIdx.x; c < kGridCellsN; c += kGridThreads) G.cur[c] = 0;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        lo_x = fminf(lo_x, __shfl_xor(lo_x, o, kWave)); lo_y = fminf(lo_y, __shfl_xor(lo_y, o, kWave));
        hi_x = fmaxf(hi_x, __shfl_xor(hi_x, o, kWave)); hi_y = fmaxf(hi_y, __shfl_xor(hi_y, o, kWave));
    }
    if (lane == 0) { G.red[0][w] = lo_x; G.red[1][w] = lo_y; G.red[2][w] = hi_x; G.red[3][w] = hi_y; }
    const int anybad = __syncthreads_or(bad);
    if (threadIdx.x == 0) {
        for (int k = 1; k < kGridThreads / kWave; k++) {
            lo_x = fminf(lo_x, G.red[0][k]); lo_y = fminf(lo_y, G.red[1][k]); hi_x = fmaxf(hi_x, G.red[2][k]); hi_y = fmaxf(hi_y, G.red[3][k]);
        }
        G.red[0][0] = hi_x - lo_x; G.red[1][0] = hi_y - lo_y;
        G.ox = lo_x; G.oy = lo_y;
        G.all = (anybad || !(G.red[0][0] >= 0 && G.red[1][0] >= 0)) ? 1u : 0u;     // (no box at all: -inf -> all, a loop over 0 boxes)
    }
    __syncthreads();
    // the cells a box's bounding box covers: the cell indices of its corners come from the same monotone map as a point's, so
    // a point inside [xmin, xmax] lands inside [cell(xmin), cell(xmax)].  The finest of 32 / 16 / 8 cells per axis whose
    // registrations fit the list (boxes a few cells wide register ~5 times each at 32)
    __shared__ unsigned long long gsm[kGridThreads / kWave];
    uint32_t c4[4], ex = 0;
    for (int gn = kGridN; gn >= 8 && !G.all; gn >>= 1) {
        if (threadIdx.x == 0) {
            G.n = gn;
            G.ix = G.red[0][0] > 0 ? (float)gn / G.red[0][0] : 0.f;
            G.iy = G.red[1][0] > 0 ? (float)gn / G.red[1][0] : 0.f;
        }
        for (int c = threadIdx.x; c < kGridCellsN; c += kGridThreads) G.cur[c] = 0;
        __syncthreads();
        bool over = false;
        for (int i = threadIdx.x; i < (int)m; i += kGridThreads) {
            const float *b = boxes + (size_t)i * bstride + boff;
            const BoxGeom<float> g = make_geom_cs<float>(b[0], b[1], b[cols.w], b[cols.h], G.cs[i].x, G.cs[i].y);
            const int x0 = grid_axis(g.xmin, G.ox, G.ix, gn), x1 = grid_axis(g.xmax, G.ox, G.ix, gn);
            const int y0 = grid_axis(g.ymin, G.oy, G.iy, gn), y1 = grid_axis(g.ymax, G.oy, G.iy, gn);
            if ((x1 - x0 + 1) * (y1 - y0 + 1) > kGridBoxCells) { over = true; continue; }
            for (int cy = y0; cy <= y1; cy++)
                for (int cx = x0; cx <= x1; cx++) atomicAdd(&G.cur[cy * gn + cx], 1u);
        }
        const int anyover = __syncthreads_or(over);
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { c4[k] = G.cur[threadIdx.x * 4 + k]; mine += c4[k]; }
        unsigned long long tot;
        ex = (uint32_t)block_excl_scan_u64<kGridThreads>(mine, &tot, gsm);
        if (!anyover && tot <= (unsigned long long)kGridList) break;          // (block-uniform)
        if (gn == 8 && threadIdx.x == 0) G.all = 1u;
        __syncthreads();
    }
    __syncthreads();
    if (G.all) return;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        G.start[threadIdx.x * 4 + k] = (uint16_t)ex;
        G.cur[threadIdx.x * 4 + k] = ex;
        ex += c4[k];
    }
    if (threadIdx.x == kGridThreads - 1) G.start[kGridCellsN] = (uint16_t)ex;
    __syncthreads();
    for (int i = threadIdx.x; i < (int)m; i += kGridThreads) {
        const float *b = boxes + (size_t)i * bstride + boff;
        const BoxGeom<float> g = make_geom_cs<float>(b[0], b[1], b[cols.w], b[cols.h], G.cs[i].x, G.cs[i].y);
        const int x0 = grid_axis(g.xmin, G.ox, G.ix, G.n), x1 = grid_axis(g.xmax, G.ox, G.ix, G.n);
        const int y0 = grid_axis(g.ymin, G.oy, G.iy, G.n), y1 = grid_axis(g.ymax, G.oy, G.iy, G.n);
        for (int cy = y0; cy <= y1; cy++)
            for (int cx = x0; cx <= x1; cx++) G.list[atomicAdd(&G.cur[cy * G.n + cx], 1u)] = (uint16_t)i;
    }
    __syncthreads();
}

// the boxes a point at (x, y) has to be tested against: f(box index) for each (any order)
template <class F>
__device__ __forceinline__ void for_candidates(const BoxGrid &G, int64_t m, float x, float y, F &&f)
{
    if (G.all) {
        for (int i = 0; i < (int)m; i++) f(i);
        return;
    }
    if (!(x >= G.ox && y >= G.oy)) return;                  // left of / below every box (or NaN): in none
    const int c = grid_axis(y, G.oy, G.iy, G.n) * G.n + grid_axis(x, G.ox, G.ix, G.n);
    for (int e = G.start[c]; e < (int)G.start[c + 1]; e++) f((int)G.list[e]);
}

__global__ __launch_bounds__(kGridThreads) void k_paint_label_grid(const float *__restrict__ points, int64_t n, int pstride,
                                                                   const uint8_t *__restrict__ semantics, const float *__restrict__ boxes,
                                                                   int64_t m, int bstride, int boff, const uint8_t *__restrict__ labels,
                                                                   uint16_t *__restrict__ idarr)
{
    __shared__ BoxGrid G;
    __shared__ uint8_t cls[kGridMaxBoxes];
    for (int i = threadIdx.x; i < (int)m; i += kGridThreads) cls[i] = labels[i];
    build_box_grid(G, boxes, m, bstride, boff);
    const int64_t j0 = (int64_t)blockIdx.x * (kGridThreads * kGridPts) + threadIdx.x;
    for (int k = 0; k < kGridPts; k++) {
        const int64_t j = j0 + (int64_t)k * kGridThreads;
        if (j >= n) break;
        const float x = points[j * pstride], y = points[j * pstride + 1], z = points[j * pstride + 2];
        const uint8_t sem = semantics[j];
        uint32_t best = 0xffffffffu;                            // the LOWEST index of a box that holds the point (abstraction.pyx:673-682)
        for_candidates(G, m, x, y, [&](int i) {
            if (cls[i] != sem || (uint32_t)i >= best) return;
            if (contains3(box3_cs(boxes + (size_t)i * bstride + boff, G.cs[i]), x, y, z)) best = (uint32_t)i;
        });
        idarr[j] = best == 0xffffffffu ? (uint16_t)0 : (uint16_t)(best + 1);
    }
}

// the mask was zeroed by the caller's memset: only the hits are stored
__global__ __launch_bounds__(kGridThreads) void k_crop3dr_grid(const float *__restrict__ points, int64_t n, int pstride,
                                                               const float *__restrict__ boxes, int64_t m, int bstride, int boff,
                                                               uint8_t *__restrict__ out)
{
    __shared__ BoxGrid G;
    build_box_grid(G, boxes, m, bstride, boff);
    const int64_t j0 = (int64_t)blockIdx.x * (kGridThreads * kGridPts) + threadIdx.x;
    for (int k = 0; k < kGridPts; k++) {
        const int64_t j = j0 + (int64_t)k * kGridThreads;
        if (j >= n) break;
        const float x = points[j * pstride], y = points[j * pstride + 1], z = points[j * pstride + 2];
        for_candidates(G, m, x, y, [&](int i) {
            if (contains3(box3_cs(boxes + (size_t)i * bstride + boff, G.cs[i]), x, y, z)) out[(size_t)i * n + j] = 1;
        });
    }
}

// crop_2dr on fp32 inputs through the same grid: points[n, 2], boxes[m, 5] = (x, y, w, h, r); the geometry of a candidate is
// make_geom_cs on the cached cos / sin = the numbers Box2D<float>::load gives k_crop2dr (same expressions), the test is the
// same quad_contains: same bits
__global__ __launch_bounds__(kGridThreads) void k_crop2dr_grid(const float *__restrict__ points, int64_t n,
                                                               const float *__restrict__ boxes, int64_t m, uint8_t *__restrict__ out)
{
    __shared__ BoxGrid G;
    build_box_grid(G, boxes, m, 5, 0, kCols2);
    const int64_t j0 = (int64_t)blockIdx.x * (kGridThreads * kGridPts) + threadIdx.x;
    for (int k = 0; k < kGridPts; k++) {
        const int64_t j = j0 + (int64_t)k * kGridThreads;
        if (j >= n) break;
        const float x = points[j * 2], y = points[j * 2 + 1];
        for_candidates(G, m, x, y, [&](int i) {
            const float *b = boxes + (size_t)i * 5;
            if (quad_contains<float>(make_geom_cs<float>(b[0], b[1], b[2], b[3], G.cs[i].x, G.cs[i].y), x, y)) out[(size_t)i * n + j] = 1;
        });
    }
}

// box3dp_crop along z in ONE launch (reference box/__init__.py:289-315 composes it from seven tensor operations around crop_2dr:
// two column gathers, [M,N] differences, comparisons and ANDs -- 200 us for a frame's 120 k points x 50 boxes, 6 M-element passes
// each): the rotated-rectangle test of k_crop2dr_grid on columns (x, y | w, h, r) = (0, 1 | 3, 4, 6) of the 7-float box rows, and
// the interval test in the reference's own float expressions: (p - d / 2 < b) & (b < p + d / 2), :311-313.
__global__ __launch_bounds__(kGridThreads) void k_crop3dp_grid(const float *__restrict__ points, int64_t n, int pstride,
                                                               const float *__restrict__ boxes, int64_t m, int bstride,
                                                               uint8_t *__restrict__ out)
{
    __shared__ BoxGrid G;
    build_box_grid(G, boxes, m, bstride, 0);
    const int64_t j0 = (int64_t)blockIdx.x * (kGridThreads * kGridPts) + threadIdx.x;
    for (int k = 0; k < kGridPts; k++) {
        const int64_t j = j0 + (int64_t)k * kGridThreads;
        if (j >= n) break;
        const float x = points[j * pstride], y = points[j * pstride + 1], z = points[j * pstride + 2];
        for_candidates(G, m, x, y, [&](int i) {
            const float *b = boxes + (size_t)i * bstride;
            if (!quad_contains<float>(make_geom_cs<float>(b[0], b[1], b[3], b[4], G.cs[i].x, G.cs[i].y), x, y)) return;
            const float hd = b[5] / 2;
            if ((z - hd < b[2]) & (b[2] < z + hd)) out[(size_t)i * n + j] = 1;
        });
    }
}

}  // namespace

// bool[M,N] of box3dp_crop (reference box/__init__.py:289-315) for project_axis = 2: points[n, point_stride >= 3] f32, boxes
// [m, box_stride >= 7] f32 rows (x, y, z, lx, ly, lz, rz).  D3D_ERR_UNSUPPORTED (nothing touched) for another axis, more than 4096
// boxes or fewer than 4096 points: the caller composes it from d3d_crop_2dr as the reference does.
extern "C" int d3d_crop_3dp(const float *points, int64_t n, int32_t point_stride, const float *boxes, int64_t m, int32_t box_stride,
                            int32_t project_axis, uint8_t *out, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || m < 0 || point_stride < 3 || box_stride < 7) return D3D_ERR_BAD_ARG;
    if (project_axis != 2 || m > kGridMaxBoxes || n < 4096) return D3D_ERR_UNSUPPORTED;
    if (m == 0) return D3D_OK;
    if (!points || !boxes || !out) return D3D_ERR_BAD_ARG;
    D3D_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)m * (size_t)n, st));
    D3D_LAUNCH("k_crop3dp_grid", k_crop3dp_grid, dim3((unsigned)d3d_divup(n, kGridThreads * kGridPts)), dim3(kGridThreads), 0, st, points, n,
               point_stride, boxes, m, box_stride, out);
    return D3D_OK;
}

// d3d_crop_2dr's fp32 path for up to kGridMaxBoxes boxes (box.hip): zeros at the memset's rate, then the hits
extern "C" int d3d_internal_crop2dr_grid_f32(const float *points, int64_t n, const float *boxes, int64_t m, uint8_t *out, hipStream_t st)
{
    if (m > kGridMaxBoxes || n < 4096) return D3D_ERR_UNSUPPORTED;
    D3D_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)m * (size_t)n, st));
    D3D_LAUNCH("k_crop2dr_grid", k_crop2dr_grid, dim3((unsigned)d3d_divup(n, kGridThreads * kGridPts)), dim3(kGridThreads), 0, st, points, n,
               boxes, m, out);
    return D3D_OK;
}

// bool[M,N] indicators of box3dr_contains (dgal_wrap.h:6-19) over boxes x points (Target3DArray.crop_points,
// abstraction.pyx:654-660, 684-687).  points[n, point_stride >= 3] f32 (x, y, z first), box row i = 7 floats
// (x, y, z, lx, ly, lz, rz) at boxes + i * box_stride + box_offset.
extern "C" int d3d_crop_3dr(const float *points, int64_t n, int32_t point_stride, const float *boxes, int64_t m,
                            int32_t box_stride, int32_t box_offset, uint8_t *out, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || m < 0 || point_stride < 3 || box_offset < 0 || box_stride < box_offset + 7) return D3D_ERR_BAD_ARG;
    if (n == 0 || m == 0) return D3D_OK;
    if (!points || !boxes || !out) return D3D_ERR_BAD_ARG;
    if (d3d_divup(m, kBoxTile) > 65535) return D3D_ERR_BAD_ARG;
    if (m <= kGridMaxBoxes && n >= 4096) {        // zeros at the memset's rate, then the hits from the boxes of each point's cell
        D3D_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)m * (size_t)n, st));
        D3D_LAUNCH("k_crop3dr_grid", k_crop3dr_grid, dim3((unsigned)d3d_divup(n, kGridThreads * kGridPts)), dim3(kGridThreads), 0, st, points, n,
                   (int)point_stride, boxes, m, (int)box_stride, (int)box_offset, out);
        return D3D_OK;
    }
    dim3 grid((unsigned)d3d_divup(n, 256 * 4), (unsigned)d3d_divup(m, kBoxTile));
    D3D_LAUNCH("k_crop3dr", k_crop3dr, grid, dim3(256), 0, st, points, n, (int)point_stride, boxes, m, (int)box_stride, (int)box_offset, out);
    return D3D_OK;
}

// Target3DArray.paint_label (abstraction.pyx:662-682): idarr[n] u16 = 1 + the lowest index of a box that contains the point
// and whose class labels[i] equals semantics[j]; 0 where there is none.  No [M,N] mask is materialised.
extern "C" int d3d_paint_label(const float *points, int64_t n, int32_t point_stride, const uint8_t *semantics, const float *boxes,
                               int64_t m, int32_t box_stride, int32_t box_offset, const uint8_t *labels, uint16_t *idarr, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || m < 0 || point_stride < 3 || box_offset < 0 || box_stride < box_offset + 7) return D3D_ERR_BAD_ARG;
    if (n == 0) return D3D_OK;
    if (!points || !semantics || !idarr || (m > 0 && (!boxes || !labels))) return D3D_ERR_BAD_ARG;
    if (m <= kGridMaxBoxes && n >= 4096) {
        D3D_LAUNCH("k_paint_label_grid", k_paint_label_grid, dim3((unsigned)d3d_divup(n, kGridThreads * kGridPts)), dim3(kGridThreads), 0, st,
                   points, n, (int)point_stride, semantics, boxes, m, (int)box_stride, (int)box_offset, labels, idarr);
        return D3D_OK;
    }
    D3D_LAUNCH("k_paint_label", k_paint_label, dim3((unsigned)d3d_divup(n, 256 * kPaintPts)), dim3(256), 0, st, points, n, (int)point_stride,
               semantics, boxes, m, (int)box_stride, (int)box_offset, labels, idarr);
    return D3D_OK;
}

// The host build of geom.hpp's HostSinCos, for tests/test_host_sincos.py: the restatement is checked against the libm it claims
// to reproduce on whatever machine runs the tests (no GPU involved).  Returns how many angles took the restated path.
extern "C" int64_t d3d_internal_host_sincosf(const float *angles, int64_t n, float *sines, float *cosines)
{
    int64_t covered = 0;
    for (int64_t i = 0; i < n; ++i) {
        float s = 0.f, c = 0.f;
        if (HostSinCos::eval(angles[i], &s, &c)) ++covered;
        else { s = sinf(angles[i]); c = cosf(angles[i]); }
        sines[i] = s; cosines[i] = c;
    }
    return covered;
}
