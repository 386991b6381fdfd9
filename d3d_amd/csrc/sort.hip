// sort.hip -- stable descending argsort (rocPRIM radix sort; a plain library sort, the same
// role torch::argsort plays inside the reference: nms.cpp:103, voxelize.cpp:406).
#include <cstring>
#include <cstdlib>
#include "common.hpp"
#include <rocprim/rocprim.hpp>

namespace {
__global__ void k_widen(const int32_t *in, int64_t n, int64_t *out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

template <typename K>
size_t sort_temp_bytes(int64_t n)
{
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs_desc<rocprim::default_config, const K *, K *, rocprim::counting_iterator<int32_t>, int32_t *>(
        nullptr, tmp, nullptr, nullptr, rocprim::counting_iterator<int32_t>(0), nullptr, (size_t)n);
    return tmp;
}

template <typename K>
size_t argsort_bytes(int64_t n)
{
    if (n < 1) n = 1;
    return d3d_align_up(sizeof(K) * n) + 2 * d3d_align_up(sizeof(int32_t) * n) + d3d_align_up(sort_temp_bytes<K>(n)) + 256;
}

// order32[n] <- stable descending argsort of keys
template <typename K>
int argsort_desc(const K *keys, int64_t n, int32_t *order32, void *ws, size_t ws_bytes, hipStream_t st)
{
    if (n <= 0) return D3D_OK;
    WsCarver w(ws, ws_bytes);
    K *keys_out = w.take<K>(n);
    int32_t *iota = w.take<int32_t>(n);
    size_t tmp = sort_temp_bytes<K>(n);
    char *temp = w.take<char>(tmp);
    if (!ws || !w.ok()) return D3D_ERR_WORKSPACE;
    (void)iota;     // the values 0 .. n-1 come from a counting iterator: no kernel to materialise them
    D3D_HIP_CHECK((rocprim::radix_sort_pairs_desc(temp, tmp, keys, keys_out, rocprim::counting_iterator<int32_t>(0), order32,
                                                  (size_t)n, 0, sizeof(K) * 8, st)));
    return D3D_OK;
}
}  // namespace

extern "C" size_t d3d_internal_argsort_i32_bytes(int64_t n) { return argsort_bytes<int32_t>(n); }
extern "C" int d3d_internal_argsort_desc_i32(const int32_t *keys, int64_t n, int32_t *order, void *ws, size_t ws_bytes,
                                             hipStream_t st)
{
    return argsort_desc<int32_t>(keys, n, order, ws, ws_bytes, st);
}

extern "C" size_t d3d_argsort_desc_workspace_bytes(int64_t n, int32_t dtype)
{
    size_t b = dtype == D3D_F64 ? argsort_bytes<double>(n) : argsort_bytes<float>(n);
    return b + d3d_align_up(sizeof(int32_t) * (n > 0 ? n : 1));
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
    int32_t *o32 = (int32_t *)workspace;
    char *rest = (char *)workspace + d3d_align_up(sizeof(int32_t) * n);
    size_t rest_bytes = workspace_bytes - d3d_align_up(sizeof(int32_t) * n);
    int rc = dtype == D3D_F64 ? argsort_desc<double>((const double *)keys, n, o32, rest, rest_bytes, st)
                              : argsort_desc<float>((const float *)keys, n, o32, rest, rest_bytes, st);
    if (rc) return rc;
    D3D_LAUNCH("k_widen", k_widen, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, (const int32_t *)o32, n, order);
    return D3D_OK;
}
