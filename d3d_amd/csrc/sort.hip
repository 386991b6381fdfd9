// sort.hip -- stable descending argsort (rocPRIM radix sort; a plain library sort, the same
// role torch::argsort plays inside the reference: nms.cpp:103, voxelize.cpp:406).
#include <cstring>
#include <cstdlib>
#include "common.hpp"
#include <rocprim/rocprim.hpp>

namespace {
// rocPRIM sorts inputs below 1 M items with a merge sort: block sort (1024 items per workgroup) + one launch per doubling,
// 7 at 100 k.  A larger block sort (merge_sort_config<512, 1024, 4>: 4096 items, 5 merge launches) was slower end to end
// (NMS of 100 k boxes 0.240 vs 0.231 ms), so the library default stays.
typedef rocprim::default_config SortConfig;

// V = type of the order entries (int32 inside the library, int64 at the C ABI: written directly, no widening pass)
template <typename K, typename V>
size_t sort_temp_bytes(int64_t n)
{
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs_desc<SortConfig, const K *, K *, rocprim::counting_iterator<V>, V *>(
        nullptr, tmp, nullptr, nullptr, rocprim::counting_iterator<V>(0), nullptr, (size_t)n);
    return tmp;
}

template <typename K, typename V>
size_t argsort_bytes(int64_t n)
{
    if (n < 1) n = 1;
    return d3d_align_up(sizeof(K) * n) + d3d_align_up(sort_temp_bytes<K, V>(n)) + 256;
}

// order[n] <- stable descending argsort of keys; the values 0 .. n-1 come from a counting iterator
template <typename K, typename V>
int argsort_desc(const K *keys, int64_t n, V *order, void *ws, size_t ws_bytes, hipStream_t st)
{
    if (n <= 0) return D3D_OK;
    WsCarver w(ws, ws_bytes);
    K *keys_out = w.take<K>(n);
    size_t tmp = sort_temp_bytes<K, V>(n);
    char *temp = w.take<char>(tmp);
    if (!ws || !w.ok()) return D3D_ERR_WORKSPACE;
    D3D_HIP_CHECK((rocprim::radix_sort_pairs_desc<SortConfig>(temp, tmp, keys, keys_out, rocprim::counting_iterator<V>(0), order,
                                                              (size_t)n, 0, sizeof(K) * 8, st)));
    return D3D_OK;
}
}  // namespace

extern "C" size_t d3d_internal_argsort_i32_bytes(int64_t n) { return argsort_bytes<int32_t, int32_t>(n); }
extern "C" int d3d_internal_argsort_desc_i32(const int32_t *keys, int64_t n, int32_t *order, void *ws, size_t ws_bytes,
                                             hipStream_t st)
{
    return argsort_desc<int32_t, int32_t>(keys, n, order, ws, ws_bytes, st);
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
