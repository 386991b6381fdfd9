// point.hip -- aligned_scatter ("next" row of SURVEY 8f): gather the 2^Dim neighbours of a fractional coordinate
// from a dense feature map (MEAN / LINEAR), and the scatter-add of its backward pass.  Replaces reference
// d3d/point/scatter.cpp + scatter_cuda.cu.  The reference launches one block per point with threads over
// channels (tiny blocks when C is small, scatter_cuda.cu:196-197); here one lane owns one (point, channel) pair,
// channel fastest, so the [N, C] output is written coalesced and wavefronts stay full for any C.
#include "common.hpp"

namespace {

struct MapDims { long long d[3]; int dim; long long C, vol; };

template <typename T> __device__ __forceinline__ int fl(T v) { int i = (int)v; return (i > v) ? i - 1 : i; }   // scatter.cpp:22
template <typename T> __device__ __forceinline__ int ce(T v) { int i = (int)v; return (i < v) ? i + 1 : i; }   // scatter.cpp:28

// neighbour j of a coordinate row: flat offset inside one [D1..Dm] map and its LINEAR weight (scatter.cpp:34-79)
template <typename T, bool LINEAR>
__device__ __forceinline__ long long neighbour(const MapDims &md, const T *cr, int j, T &w)
{
    long long off = 0;
    w = 1;
    for (int d = 0; d < md.dim; d++) {
        const int dmax = (int)md.d[d] - 1;
        const T dc = cr[d + 1];
        int lc;
        if (dc > dmax) { lc = dmax; if (LINEAR) w *= (T)0.5; }
        else if (dc < 0) { lc = 0; if (LINEAR) w *= (T)0.5; }
        else if (j & (1 << d)) { lc = ce(dc); if (LINEAR) w *= 1 + dc - lc; }
        else { lc = fl(dc); if (LINEAR) w *= 1 - dc + lc; }
        off = off * md.d[d] + lc;
    }
    return off;
}

// CL: `image` is the channels-last copy [B, D1..Dm, C] made by k_to_channels_last -- the C channels of a neighbour are
// then contiguous, so a wavefront (consecutive channels of one point) reads each neighbour with a few coalesced
// requests instead of one request per channel (planes of the [B, C, D1..Dm] map are vol * sizeof(T) apart).
template <typename T, bool LINEAR, bool CL>
__global__ __launch_bounds__(256) void k_scatter_fwd(const T *__restrict__ coord, long long n, const T *__restrict__ image,
                                                     MapDims md, T *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * md.C) return;
    const long long i = t / md.C, c = t - i * md.C;
    const T *cr = coord + i * (md.dim + 1);
    const long long b = (long long)(int)cr[0];
    const T *img = CL ? image + b * md.vol * md.C + c : image + (b * md.C + c) * md.vol;
    const long long step = CL ? md.C : 1;
    const int nb = 1 << md.dim;
    T sum = 0;
    for (int j = 0; j < nb; j++) {          // same accumulation order as the reference (scatter.cpp:108-127)
        T w;
        const long long off = neighbour<T, LINEAR>(md, cr, j, w);
        sum += LINEAR ? img[off * step] * w : img[off * step];
    }
    out[t] = LINEAR ? sum : sum / nb;
}

// [B, C, vol] -> [B, vol, C] (ADD = false) through 32 x 32 LDS tiles, both sides coalesced;
// ADD: dst[B, C, vol] += src[B, vol, C] (the channels-last gradient accumulator folded back into image_grad)
template <typename T, bool BACK_ADD>
__global__ __launch_bounds__(256) void k_channels_last(const T *__restrict__ src, T *__restrict__ dst, long long C, long long vol)
{
    __shared__ T tile[32][33];
    const long long b = blockIdx.z;
    const long long c0 = (long long)blockIdx.y * 32, v0 = (long long)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8 threads
    if (!BACK_ADD) {
        for (int r = ty; r < 32; r += 8) {                        // read rows of [C, vol]: vol fastest
            const long long c = c0 + r, v = v0 + tx;
            if (c < C && v < vol) tile[r][tx] = src[(b * C + c) * vol + v];
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {                        // write rows of [vol, C]: C fastest
            const long long v = v0 + r, c = c0 + tx;
            if (c < C && v < vol) dst[(b * vol + v) * C + c] = tile[tx][r];
        }
    } else {
        for (int r = ty; r < 32; r += 8) {                        // read rows of [vol, C]
            const long long v = v0 + r, c = c0 + tx;
            if (c < C && v < vol) tile[r][tx] = src[(b * vol + v) * C + c];
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {                        // accumulate into rows of [C, vol]
            const long long c = c0 + r, v = v0 + tx;
            if (c < C && v < vol) dst[(b * C + c) * vol + v] += tile[tx][r];
        }
    }
}

template <typename T, bool LINEAR, bool CL>
__global__ __launch_bounds__(256) void k_scatter_bwd(const T *__restrict__ coord, long long n, const T *__restrict__ grad,
                                                     MapDims md, T *image_grad)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * md.C) return;
    const long long i = t / md.C, c = t - i * md.C;
    const T *cr = coord + i * (md.dim + 1);
    const long long b = (long long)(int)cr[0];
    T *img = CL ? image_grad + b * md.vol * md.C + c : image_grad + (b * md.C + c) * md.vol;
    const long long step = CL ? md.C : 1;
    const int nb = 1 << md.dim;
    const T g = grad[t];
    for (int j = 0; j < nb; j++) {
        T w;
        const long long off = neighbour<T, LINEAR>(md, cr, j, w);
        atomicAdd(&img[off * step], LINEAR ? g * w : g / nb);     // the reference's CPU loop does a racy += (scatter.cpp:164,168)
    }
}

template <typename T>
int scatter_dispatch(bool backward, const T *coord, long long n, const T *a, MapDims md, int atype, T *b, long long batch,
                     void *ws, size_t ws_bytes, hipStream_t st)
{
    const dim3 grid((unsigned)d3d_divup(n * md.C, 256));
    const bool lin = atype == 2;
    // channels-last staging pays when a wavefront spans several channels of a point and the copy fits the workspace
    const size_t map_bytes = (size_t)batch * (size_t)md.C * (size_t)md.vol * sizeof(T);
    const bool cl = ws && batch > 0 && md.C >= 8 && ws_bytes >= map_bytes && md.vol < (1ll << 31) && batch < 65536;
    const dim3 tgrid((unsigned)d3d_divup(md.vol, 32), (unsigned)d3d_divup(md.C, 32), (unsigned)(cl ? batch : 1));
    T *stage = reinterpret_cast<T *>(ws);
    if (!backward) {
        if (cl) {
            D3D_LAUNCH("k_channels_last", (k_channels_last<T, false>), tgrid, dim3(256), 0, st, a, stage, md.C, md.vol);
            if (lin) D3D_LAUNCH("k_scatter_fwd", (k_scatter_fwd<T, true, true>), grid, dim3(256), 0, st, coord, n, (const T *)stage, md, b);
            else D3D_LAUNCH("k_scatter_fwd", (k_scatter_fwd<T, false, true>), grid, dim3(256), 0, st, coord, n, (const T *)stage, md, b);
        } else {
            if (lin) D3D_LAUNCH("k_scatter_fwd", (k_scatter_fwd<T, true, false>), grid, dim3(256), 0, st, coord, n, a, md, b);
            else D3D_LAUNCH("k_scatter_fwd", (k_scatter_fwd<T, false, false>), grid, dim3(256), 0, st, coord, n, a, md, b);
        }
    } else {
        if (cl) {
            D3D_HIP_CHECK(hipMemsetAsync(stage, 0, map_bytes, st));
            if (lin) D3D_LAUNCH("k_scatter_bwd", (k_scatter_bwd<T, true, true>), grid, dim3(256), 0, st, coord, n, a, md, stage);
            else D3D_LAUNCH("k_scatter_bwd", (k_scatter_bwd<T, false, true>), grid, dim3(256), 0, st, coord, n, a, md, stage);
            D3D_LAUNCH("k_channels_last", (k_channels_last<T, true>), tgrid, dim3(256), 0, st, (const T *)stage, b, md.C, md.vol);
        } else {
            if (lin) D3D_LAUNCH("k_scatter_bwd", (k_scatter_bwd<T, true, false>), grid, dim3(256), 0, st, coord, n, a, md, b);
            else D3D_LAUNCH("k_scatter_bwd", (k_scatter_bwd<T, false, false>), grid, dim3(256), 0, st, coord, n, a, md, b);
        }
    }
    return D3D_OK;
}

int scatter_common(bool backward, const void *coord, int64_t n, int32_t dim, const void *a, int64_t C, const int64_t *dims,
                   int32_t atype, int32_t dtype, void *b, int64_t batch, void *ws, size_t ws_bytes, void *stream)
{
    if (n < 0 || C < 0 || !dims) return D3D_ERR_BAD_ARG;
    if (dim < 1 || dim > 3) return D3D_ERR_UNSUPPORTED;             // "Unsupported dimension size" (scatter.h:33)
    if (atype != 1 && atype != 2) return D3D_ERR_UNSUPPORTED;       // "Unsupported align type!" (scatter.h:18)
    if (dtype != D3D_F32 && dtype != D3D_F64) return D3D_ERR_BAD_ARG;
    if (n == 0 || C == 0) return D3D_OK;
    if (!coord || !a || !b) return D3D_ERR_BAD_ARG;
    MapDims md;
    md.dim = dim; md.C = C; md.vol = 1;
    for (int d = 0; d < 3; d++) { md.d[d] = d < dim ? dims[d] : 1; if (md.d[d] <= 0) return D3D_ERR_BAD_ARG; md.vol *= md.d[d]; }
    if (d3d_divup(n * C, 256) > 0x7fffffffll) return D3D_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == D3D_F64)
        return scatter_dispatch<double>(backward, (const double *)coord, n, (const double *)a, md, atype, (double *)b, batch, ws,
                                        ws_bytes, st);
    return scatter_dispatch<float>(backward, (const float *)coord, n, (const float *)a, md, atype, (float *)b, batch, ws, ws_bytes, st);
}

}  // namespace

extern "C" size_t d3d_aligned_scatter_workspace_bytes(int64_t batch, int64_t channels, const int64_t *dims, int32_t dim,
                                                      int32_t dtype)
{
    if (batch < 1 || channels < 1 || !dims || dim < 1 || dim > 3) return 256;
    size_t vol = 1;
    for (int d = 0; d < dim; d++) vol *= (size_t)(dims[d] > 0 ? dims[d] : 1);
    return d3d_align_up((size_t)batch * (size_t)channels * vol * (dtype == D3D_F64 ? 8 : 4)) + 256;
}

extern "C" int d3d_aligned_scatter_forward(const void *coord, int64_t n, int32_t dim, const void *image, int64_t batch,
                                           int64_t channels, const int64_t *dims, int32_t align_type, int32_t dtype,
                                           void *out, void *workspace, size_t workspace_bytes, void *stream)
{
    return scatter_common(false, coord, n, dim, image, channels, dims, align_type, dtype, out, batch, workspace, workspace_bytes,
                          stream);
}

extern "C" int d3d_aligned_scatter_backward(const void *coord, int64_t n, int32_t dim, const void *grad, int64_t batch,
                                            int64_t channels, const int64_t *dims, int32_t align_type, int32_t dtype,
                                            void *image_grad, void *workspace, size_t workspace_bytes, void *stream)
{
    return scatter_common(true, coord, n, dim, grad, channels, dims, align_type, dtype, image_grad, batch, workspace,
                          workspace_bytes, stream);
}
