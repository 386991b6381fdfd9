// grid.hip -- compact, rank-independent numbering of occupied grid cells (used by the point-sharded
// voxelizer to give every rank the same slot for the same voxel without exchanging a dictionary):
// all ranks mark the cells named by the all-gathered key lists in a bitmap over the grid, a popcount
// prefix scan of the bitmap words turns "cell is occupied" into "index among the occupied cells in
// linear-key order".  RCCL has no bitwise-OR reduction, hence the keys are all-gathered and the
// bitmap is built locally.
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void k_grid_mark(const int64_t *__restrict__ keys, int64_t m, int64_t ncells,
                                                   unsigned long long *bitmap)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int64_t k = keys[j];
    if (k < 0 || k >= ncells) return;
    atomicOr(&bitmap[k >> 6], 1ull << (k & 63));
}

struct PopcountWords {
    static constexpr const char *kName = "k_scan_count<PopcountWords>", *kName2 = "k_scan_apply<PopcountWords>";
    const unsigned long long *bitmap;
    uint32_t *prefix;
    __device__ __forceinline__ unsigned long long value(int64_t w) const { return (unsigned long long)__popcll(bitmap[w]); }
    __device__ __forceinline__ unsigned long long value2(int64_t w) const { return value(w); }
    __device__ __forceinline__ void apply(int64_t w, unsigned long long, unsigned long long excl) const
    {
        prefix[w] = (uint32_t)excl;
    }
};

__global__ __launch_bounds__(256) void k_grid_lookup(const int64_t *__restrict__ keys, int64_t m, int64_t ncells,
                                                     const unsigned long long *__restrict__ bitmap,
                                                     const uint32_t *__restrict__ prefix, int64_t *slot)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int64_t k = keys[j];
    long long s = -1;
    if (k >= 0 && k < ncells) {
        const unsigned long long word = bitmap[k >> 6], bit = 1ull << (k & 63);
        if (word & bit) s = (long long)prefix[k >> 6] + __popcll(word & (bit - 1));
    }
    slot[j] = s;
}

}  // namespace

extern "C" size_t d3d_grid_compact_workspace_bytes(int64_t ncells)
{
    if (ncells < 1) ncells = 1;
    const size_t nw = (size_t)d3d_divup(ncells, 64);
    return d3d_align_up(nw * 8) + d3d_align_up(nw * 4) + d3d_align_up(((size_t)d3d_divup((int64_t)nw, kScanTile) + 1) * 8) + 256;
}

// keys[m] (linear cell indices in [0, ncells)) -> occupancy index in `workspace`; counts[0] = number of
// distinct occupied cells.  The workspace is then read by d3d_grid_compact_lookup.
extern "C" int d3d_grid_compact_index(const int64_t *keys, int64_t m, int64_t ncells, int64_t *counts, void *workspace,
                                      size_t workspace_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (m < 0 || ncells <= 0 || !counts || (m > 0 && !keys)) return D3D_ERR_BAD_ARG;
    if (ncells >= (1ll << 37)) return D3D_ERR_BAD_ARG;
    const int64_t nw = d3d_divup(ncells, 64);
    WsCarver w(workspace, workspace_bytes);
    unsigned long long *bitmap = w.take<unsigned long long>(nw);
    uint32_t *prefix = w.take<uint32_t>(nw);
    unsigned long long *bsum = w.take<unsigned long long>(d3d_divup(nw, kScanTile) + 1);
    if (!workspace || !w.ok()) return D3D_ERR_WORKSPACE;
    D3D_HIP_CHECK(hipMemsetAsync(bitmap, 0, (size_t)nw * 8, st));
    if (m > 0) D3D_LAUNCH("k_grid_mark", k_grid_mark, dim3((unsigned)d3d_divup(m, 256)), dim3(256), 0, st, keys, m, ncells, bitmap);
    PopcountWords f{bitmap, prefix};
    return d3d_run_scan(f, nw, bsum, counts, -1, 0, ~0ull, st);
}

// slot[j] = index of keys[j] among the occupied cells (ascending linear key), or -1 if the cell is not marked
extern "C" int d3d_grid_compact_lookup(const int64_t *keys, int64_t m, int64_t ncells, const void *workspace,
                                       size_t workspace_bytes, int64_t *slot, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (m < 0 || ncells <= 0 || (m > 0 && (!keys || !slot))) return D3D_ERR_BAD_ARG;
    const int64_t nw = d3d_divup(ncells, 64);
    WsCarver w((void *)workspace, workspace_bytes);
    unsigned long long *bitmap = w.take<unsigned long long>(nw);
    uint32_t *prefix = w.take<uint32_t>(nw);
    if (!workspace || !w.ok()) return D3D_ERR_WORKSPACE;
    if (m > 0)
        D3D_LAUNCH("k_grid_lookup", k_grid_lookup, dim3((unsigned)d3d_divup(m, 256)), dim3(256), 0, st, keys, m, ncells,
                   bitmap, prefix, slot);
    return D3D_OK;
}
