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
                                                     const uint32_t *__restrict__ prefix, int64_t *slot,
                                                     long long missing)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int64_t k = keys[j];
    long long s = missing;
    if (k >= 0 && k < ncells) {
        const unsigned long long word = bitmap[k >> 6], bit = 1ull << (k & 63);
        if (word & bit) s = (long long)prefix[k >> 6] + __popcll(word & (bit - 1));
    }
    slot[j] = s;
}


// sharded voxelizer, last step: slot-ordered reduced table -> voxel-id-ordered outputs
__global__ __launch_bounds__(256) void k_sharded_finalize(int64_t nvox, int c, const int64_t *__restrict__ vid_of_slot,
                                                          const int64_t *__restrict__ key_of_slot,
                                                          const float *__restrict__ table, int tstride, int mean,
                                                          const int32_t *__restrict__ cnt_in, int64_t sy, int64_t sz,
                                                          int64_t *coords, int32_t *cnt_out, float *feats)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nvox) return;
    const int64_t v = vid_of_slot[s], k = key_of_slot[s];
    coords[v * 3 + 0] = k / (sy * sz);
    coords[v * 3 + 1] = (k / sz) % sy;
    coords[v * 3 + 2] = k % sz;
    const float *row = table + s * tstride;
    if (mean) {
        const float n = row[c];                       // counts < 2^24 are exact in fp32
        cnt_out[v] = (int32_t)(n + 0.5f);
        for (int d = 0; d < c; d++) feats[v * c + d] = row[d] / n;
    } else {
        cnt_out[v] = cnt_in[s];
        for (int d = 0; d < c; d++) feats[v * c + d] = row[d];
    }
}

// global voxel id of each local point: local voxel -> slot -> voxel id
__global__ __launch_bounds__(256) void k_sharded_map(int64_t n, const int64_t *__restrict__ local_map,
                                                     const int64_t *__restrict__ slot_of_local, int64_t nvox,
                                                     const int64_t *__restrict__ vid_of_slot, int64_t *gmap)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t m = local_map[i];
    long long g = -1;
    if (m >= 0) {
        const int64_t s = slot_of_local[m];
        if (s >= 0 && s < nvox) g = vid_of_slot[s];
    }
    gmap[i] = g;
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

// slot[j] = index of keys[j] among the occupied cells (ascending linear key), or `missing` if the cell is not marked
extern "C" int d3d_grid_compact_lookup(const int64_t *keys, int64_t m, int64_t ncells, const void *workspace,
                                       size_t workspace_bytes, int64_t missing, int64_t *slot, void *stream)
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
                   bitmap, prefix, slot, (long long)missing);
    return D3D_OK;
}

extern "C" int d3d_sharded_finalize(int64_t nvox, int32_t c, const int64_t *vid_of_slot, const int64_t *key_of_slot,
                                    const float *table, int32_t table_stride, int32_t mean, const int32_t *cnt_in,
                                    const int32_t *shape, int64_t *coords, int32_t *cnt_out, float *feats, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (nvox < 0 || c < 1 || !shape) return D3D_ERR_BAD_ARG;
    if (nvox == 0) return D3D_OK;
    if (!vid_of_slot || !key_of_slot || !table || !coords || !cnt_out || !feats || (!mean && !cnt_in)) return D3D_ERR_BAD_ARG;
    D3D_LAUNCH("k_sharded_finalize", k_sharded_finalize, dim3((unsigned)d3d_divup(nvox, 256)), dim3(256), 0, st, nvox, c,
               vid_of_slot, key_of_slot, table, table_stride, mean, cnt_in, (int64_t)shape[1], (int64_t)shape[2], coords,
               cnt_out, feats);
    return D3D_OK;
}

extern "C" int d3d_sharded_map(int64_t n, const int64_t *local_map, const int64_t *slot_of_local, int64_t nvox,
                               const int64_t *vid_of_slot, int64_t *gmap, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || nvox < 0) return D3D_ERR_BAD_ARG;
    if (n == 0) return D3D_OK;
    if (!local_map || !slot_of_local || !gmap || (nvox > 0 && !vid_of_slot)) return D3D_ERR_BAD_ARG;
    D3D_LAUNCH("k_sharded_map", k_sharded_map, dim3((unsigned)d3d_divup(n, 256)), dim3(256), 0, st, n, local_map,
               slot_of_local, nvox, vid_of_slot, gmap);
    return D3D_OK;
}
