// box.hip -- pairwise IoU (axis-aligned "box", rotated "rbox", BEV x z "iou3d") and hard NMS
// for MI355X (gfx950).  Replaces reference d3d/box/iou.cpp + iou_cuda.cu (forward),
// d3d/box/nms.cpp + nms_cuda.cu, and the Cython pair loop over d3d/dgal_wrap.h.
//
//  * IoU matrix: 2-D tiling, the row boxes of a tile are expanded once into LDS (BoxGeom:
//    trig, corners, AABB, area), every lane owns one column box in registers and walks the
//    tile's rows, so stores to ious[i, j..j+63] are coalesced row-major (the reference's
//    kernel strides lanes along i and stores with stride M, iou_cuda.cu:22-27) and pair
//    indices are 64-bit (reference overflows int at N*M >= 2^31, iou_cuda.cu:36,137).
//  * NMS: 64x64 tiles of the score-sorted "IoU > thr" bit matrix (one 64-bit word per lane =
//    one wavefront-wide row segment), then ONE workgroup sweeps the sorted order: the
//    64x64 diagonal block is resolved by a wavefront with lane broadcasts, and the rows of
//    the kept boxes are OR-ed into an LDS-resident removal bitmap by all 16 wavefronts
//    (the reference runs this on a single thread: nms_cuda.cu:80-107 <<<1,1>>>).
#include "common.hpp"
#include "geom.hpp"

namespace {

constexpr int kTileCols = 256;   // threads per block = columns per tile
constexpr int kTileRows = 64;    // rows per tile (LDS-staged)

// ---------------------------------------------------------------- box loaders
template <typename T> struct Box2D {     // rows of [.,5] = (x, y, w, h, r)
    static constexpr int kStride = 5;
    __device__ static BoxGeom<T> load(const T *b) { return make_geom<T>(b[0], b[1], b[2], b[3], b[4]); }
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
    if constexpr (K == 1) __builtin_nontemporal_store(v[0], out);
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
                                                     int64_t m, T *__restrict__ ious)
{
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

// ---------------------------------------------------------------- pairwise "3D IoU" (BEV x z), fp32
// box = (x, y, z, lx, ly, lz, rz); dgal_wrap.h:45-91
__device__ __forceinline__ Box3DGeom load3d(const float *b)
{
    Box3DGeom r;
    r.g = make_geom<float>(b[0], b[1], b[3], b[4], b[6]);
    r.zmax = b[2] + b[5] / 2;
    r.zmin = b[2] - b[5] / 2;
    return r;
}

template <bool ROTATED, int K>
__global__ __launch_bounds__(kTileCols) void k_iou3d(const float *__restrict__ b1, int64_t n,
                                                     const float *__restrict__ b2, int64_t m, float *__restrict__ out_)
{
    __shared__ Box3DGeom rows[kTileRows];
    const int64_t i0 = (int64_t)blockIdx.y * kTileRows;
    const int64_t j0 = ((int64_t)blockIdx.x * kTileCols + threadIdx.x) * K;
    const int nrows = (int)((n - i0) < kTileRows ? (n - i0) : kTileRows);
    if (threadIdx.x < nrows) rows[threadIdx.x] = load3d(b1 + (i0 + threadIdx.x) * 7);
    Box3DGeom col[K];
    const bool active = j0 < m;
    if (active) {
#pragma unroll
        for (int k = 0; k < K; k++) col[k] = load3d(b2 + (j0 + k) * 7);
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
        }
        store_row<float, K>(out, v);
        out += m;
    }
}

// ---------------------------------------------------------------- NMS
// geometry of the boxes in score order, computed once (N trig evaluations, not N^2)
template <typename T>
__global__ void k_nms_prepare(const T *__restrict__ boxes, const T *__restrict__ scores,
                              const int64_t *__restrict__ order, int64_t n, float score_threshold,
                              BoxGeom<T> *geom, unsigned long long *remv, int64_t nb)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // sorted position
    bool pre = false;
    if (p < n) {
        const int64_t i = order[p];
        geom[p] = Box2D<T>::load(boxes + i * 5);
        // nms.cpp:23-29: the tail with score <= threshold is suppressed up front, never position 0
        pre = p > 0 && !(scores[i] > (T)score_threshold);
    }
    unsigned long long word = __ballot(pre);
    if ((threadIdx.x & 63) == 0 && (p >> 6) < nb) remv[p >> 6] = word;
}

// mask[p, cb] bit c  <=>  q = 64*cb + c > p  and  IoU(sorted p, sorted q) > thr
template <typename T, bool ROTATED>
__global__ __launch_bounds__(64) void k_nms_mask(const BoxGeom<T> *__restrict__ geom, int64_t n, int64_t nb,
                                                 T thr, unsigned long long *__restrict__ mask)
{
    const int64_t rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb) return;                       // strictly-lower tiles are never read
    __shared__ BoxGeom<T> cols[64];
    const int64_t q0 = cb * 64, p = rb * 64 + threadIdx.x;
    const int ncols = (int)((n - q0) < 64 ? (n - q0) : 64);
    if ((int)threadIdx.x < ncols) cols[threadIdx.x] = geom[q0 + threadIdx.x];
    __syncthreads();
    if (p >= n) return;
    const BoxGeom<T> a = geom[p];
    unsigned long long bits = 0;
    const int cstart = (rb == cb) ? (int)threadIdx.x + 1 : 0;
    for (int c = cstart; c < ncols; c++) {
        const BoxGeom<T> &b = cols[c];
        T v = ROTATED ? iou_rbox(a, b) : iou_aabb(a, b);
        if (v > thr) bits |= 1ull << c;        // nms.cpp:53  iou > (scalar_t)(float)iou_threshold
    }
    mask[p * nb + cb] = bits;
}

// one workgroup; remv (nb words) lives in global scratch when it does not fit LDS
constexpr int kSweepThreads = 1024;
constexpr int kSweepLdsWords = 16384;   // 128 KiB of LDS -> up to 1,048,576 boxes on-chip

__global__ __launch_bounds__(kSweepThreads) void k_nms_sweep(const unsigned long long *__restrict__ mask, int64_t n,
                                                             int64_t nb, unsigned long long *remv_g,
                                                             const int64_t *__restrict__ order, uint8_t *suppressed)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long lds[];
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
        suppressed[order[p]] = (uint8_t)((remv[p >> 6] >> (p & 63)) & 1ull);
}

template <typename T>
int nms_typed(const T *boxes, const T *scores, const int64_t *order, int64_t n, int iou_type, float iou_thr,
              float score_thr, uint8_t *suppressed, void *ws, size_t ws_bytes, hipStream_t st)
{
    const int64_t nb = d3d_divup(n, 64);
    WsCarver w(ws, ws_bytes);
    BoxGeom<T> *geom = w.take<BoxGeom<T>>(nb * 64);
    unsigned long long *remv = w.take<unsigned long long>(nb);
    unsigned long long *mask = w.take<unsigned long long>((size_t)nb * 64 * nb);
    if (!ws || !w.ok()) return D3D_ERR_WORKSPACE;
    D3D_LAUNCH("k_nms_prepare", k_nms_prepare<T>, dim3((unsigned)nb), dim3(64), 0, st, boxes, scores, order, n, score_thr, geom,
                       remv, nb);
    dim3 grid((unsigned)nb, (unsigned)nb);
    if (iou_type == D3D_IOU_RBOX)
        D3D_LAUNCH("k_nms_mask", (k_nms_mask<T, true>), grid, dim3(64), 0, st, geom, n, nb, (T)iou_thr, mask);
    else
        D3D_LAUNCH("k_nms_mask", (k_nms_mask<T, false>), grid, dim3(64), 0, st, geom, n, nb, (T)iou_thr, mask);
    size_t lds = nb <= kSweepLdsWords ? (size_t)nb * 8 : 0;
    D3D_LAUNCH("k_nms_sweep", k_nms_sweep, dim3(1), dim3(kSweepThreads), lds, st, mask, n, nb, remv, order, suppressed);
    return D3D_OK;
}

}  // namespace

// ====================================================================== C ABI
extern "C" int d3d_iou2d_forward(const void *boxes1, int64_t n, const void *boxes2, int64_t m, int32_t iou_type,
                                 int32_t dtype, void *ious, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || m < 0) return D3D_ERR_BAD_ARG;
    if (dtype != D3D_F32 && dtype != D3D_F64) return D3D_ERR_BAD_ARG;
    if (iou_type != D3D_IOU_BOX && iou_type != D3D_IOU_RBOX) return D3D_ERR_UNSUPPORTED;
    if (n == 0 || m == 0) return D3D_OK;
    if (!boxes1 || !boxes2 || !ious) return D3D_ERR_BAD_ARG;
    const int64_t gy = d3d_divup(n, kTileRows);
    if (gy > 65535) return D3D_ERR_BAD_ARG;   // 4.19 M rows per call; callers tile above that
    const bool rot = iou_type == D3D_IOU_RBOX;
    const bool al16 = (reinterpret_cast<uintptr_t>(ious) & 15) == 0;
#define D3D_IOU2D(T, R, K)                                                                                          \
    D3D_LAUNCH("k_iou2d", (k_iou2d<T, R, K>), dim3((unsigned)d3d_divup(m, (int64_t)kTileCols * K), (unsigned)gy),   \
               dim3(kTileCols), 0, st, (const T *)boxes1, n, (const T *)boxes2, m, (T *)ious)
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

extern "C" int d3d_iou3d_forward(const float *boxes1, int64_t n, const float *boxes2, int64_t m, int32_t rotated,
                                 float *out, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || m < 0) return D3D_ERR_BAD_ARG;
    if (n == 0 || m == 0) return D3D_OK;
    if (!boxes1 || !boxes2 || !out) return D3D_ERR_BAD_ARG;
    if (d3d_divup(n, kTileRows) > 65535) return D3D_ERR_BAD_ARG;
    const unsigned gy = (unsigned)d3d_divup(n, kTileRows);
    const bool vec = (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (m % 4 == 0);
#define D3D_IOU3D(R, K)                                                                                            \
    D3D_LAUNCH("k_iou3d", (k_iou3d<R, K>), dim3((unsigned)d3d_divup(m, (int64_t)kTileCols * K), gy), dim3(kTileCols), \
               0, st, boxes1, n, boxes2, m, out)
    if (rotated) { if (vec) D3D_IOU3D(true, 4); else D3D_IOU3D(true, 1); }
    else         { if (vec) D3D_IOU3D(false, 4); else D3D_IOU3D(false, 1); }
#undef D3D_IOU3D
    return D3D_OK;
}

extern "C" size_t d3d_nms2d_workspace_bytes(int64_t n)
{
    if (n < 1) n = 1;
    const size_t nb = (size_t)d3d_divup(n, 64);
    return d3d_align_up(nb * 64 * sizeof(BoxGeom<double>)) + d3d_align_up(nb * 8) + d3d_align_up(nb * 64 * nb * 8) + 256;
}

extern "C" int d3d_nms2d(const void *boxes, const void *scores, const int64_t *order, int64_t n, int32_t iou_type,
                         int32_t suppression_type, int32_t dtype, float iou_threshold, float score_threshold,
                         float suppression_param, uint8_t *suppressed, void *workspace, size_t workspace_bytes,
                         void *stream)
{
    (void)suppression_param;
    hipStream_t st = (hipStream_t)stream;
    if (n < 0) return D3D_ERR_BAD_ARG;
    if (dtype != D3D_F32 && dtype != D3D_F64) return D3D_ERR_BAD_ARG;
    if (iou_type != D3D_IOU_BOX && iou_type != D3D_IOU_RBOX) return D3D_ERR_UNSUPPORTED;   // common.h:25
    if (suppression_type != D3D_SUPPRESS_HARD) return D3D_ERR_UNSUPPORTED;
    if (n == 0) return D3D_OK;
    if (!boxes || !scores || !order || !suppressed) return D3D_ERR_BAD_ARG;
    if (d3d_divup(n, 64) > 65535) return D3D_ERR_BAD_ARG;
    if (dtype == D3D_F64)
        return nms_typed<double>((const double *)boxes, (const double *)scores, order, n, iou_type, iou_threshold,
                                 score_threshold, suppressed, workspace, workspace_bytes, st);
    return nms_typed<float>((const float *)boxes, (const float *)scores, order, n, iou_type, iou_threshold,
                            score_threshold, suppressed, workspace, workspace_bytes, st);
}
