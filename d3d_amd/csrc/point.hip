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

template <typename T, bool LINEAR>
__global__ __launch_bounds__(256) void k_scatter_fwd(const T *__restrict__ coord, long long n, const T *__restrict__ image,
                                                     MapDims md, T *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * md.C) return;
    const long long i = t / md.C, c = t - i * md.C;
    const T *cr = coord + i * (md.dim + 1);
    const T *img = image + ((long long)(int)cr[0] * md.C + c) * md.vol;
    const int nb = 1 << md.dim;
    T sum = 0;
    for (int j = 0; j < nb; j++) {          // same accumulation order as the reference (scatter.cpp:108-127)
        T w;
        const long long off = neighbour<T, LINEAR>(md, cr, j, w);
        sum += LINEAR ? img[off] * w : img[off];
    }
    out[t] = LINEAR ? sum : sum / nb;
}

template <typename T, bool LINEAR>
__global__ __launch_bounds__(256) void k_scatter_bwd(const T *__restrict__ coord, long long n, const T *__restrict__ grad,
                                                     MapDims md, T *image_grad)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * md.C) return;
    const long long i = t / md.C, c = t - i * md.C;
    const T *cr = coord + i * (md.dim + 1);
    T *img = image_grad + ((long long)(int)cr[0] * md.C + c) * md.vol;
    const int nb = 1 << md.dim;
    const T g = grad[t];
    for (int j = 0; j < nb; j++) {
        T w;
        const long long off = neighbour<T, LINEAR>(md, cr, j, w);
        atomicAdd(&img[off], LINEAR ? g * w : g / nb);     // the reference's CPU loop does a racy += (scatter.cpp:164,168)
    }
}

template <typename T>
int scatter_dispatch(bool backward, const T *coord, long long n, const T *a, MapDims md, int atype, T *b, hipStream_t st)
{
    const dim3 grid((unsigned)d3d_divup(n * md.C, 256));
    const bool lin = atype == 2;
    if (!backward) {
        if (lin) D3D_LAUNCH("k_scatter_fwd", (k_scatter_fwd<T, true>), grid, dim3(256), 0, st, coord, n, a, md, b);
        else D3D_LAUNCH("k_scatter_fwd", (k_scatter_fwd<T, false>), grid, dim3(256), 0, st, coord, n, a, md, b);
    } else {
        if (lin) D3D_LAUNCH("k_scatter_bwd", (k_scatter_bwd<T, true>), grid, dim3(256), 0, st, coord, n, a, md, b);
        else D3D_LAUNCH("k_scatter_bwd", (k_scatter_bwd<T, false>), grid, dim3(256), 0, st, coord, n, a, md, b);
    }
    return D3D_OK;
}

int scatter_common(bool backward, const void *coord, int64_t n, int32_t dim, const void *a, int64_t C, const int64_t *dims,
                   int32_t atype, int32_t dtype, void *b, void *stream)
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
    if (dtype == D3D_F64) return scatter_dispatch<double>(backward, (const double *)coord, n, (const double *)a, md, atype, (double *)b, st);
    return scatter_dispatch<float>(backward, (const float *)coord, n, (const float *)a, md, atype, (float *)b, st);
}

}  // namespace

extern "C" int d3d_aligned_scatter_forward(const void *coord, int64_t n, int32_t dim, const void *image, int64_t channels,
                                           const int64_t *dims, int32_t align_type, int32_t dtype, void *out, void *stream)
{
    return scatter_common(false, coord, n, dim, image, channels, dims, align_type, dtype, out, stream);
}

extern "C" int d3d_aligned_scatter_backward(const void *coord, int64_t n, int32_t dim, const void *grad, int64_t channels,
                                            const int64_t *dims, int32_t align_type, int32_t dtype, void *image_grad,
                                            void *stream)
{
    return scatter_common(true, coord, n, dim, grad, channels, dims, align_type, dtype, image_grad, stream);
}
