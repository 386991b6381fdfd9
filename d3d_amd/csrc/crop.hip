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

}  // namespace

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
    D3D_LAUNCH("k_paint_label", k_paint_label, dim3((unsigned)d3d_divup(n, 256 * kPaintPts)), dim3(256), 0, st, points, n, (int)point_stride,
               semantics, boxes, m, (int)box_stride, (int)box_offset, labels, idarr);
    return D3D_OK;
}
