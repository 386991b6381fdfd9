/*
 * d3d_hip.h -- C ABI of libd3d_hip.so: the MI355X (gfx950) implementation of the
 * data-parallel hot path of cmpute/d3d (d3d/voxel + d3d/box).
 *
 * This is the drop-in boundary.  Every entry point replaces one function that the
 * reference binds through pybind11 in d3d/voxel/impl.cpp:3-21 and d3d/box/impl.cpp:8-54
 * (or, for iou3d, the C shims in d3d/dgal_wrap.h reached from Cython).  Conventions:
 *
 *  - plain pointers + sizes, no torch / pybind types;
 *  - all tensor arguments are DEVICE pointers (HBM), contiguous row-major, unless
 *    the parameter comment says "host";
 *  - the caller owns every buffer, including the scratch `workspace` whose size is
 *    returned by the matching *_workspace_bytes() query (256-byte aligned base);
 *  - variable-size results are written into caller-provided upper-bound buffers; the
 *    actual sizes go to a small device array `counts` (int64) that the caller copies
 *    back when it needs them.  No entry point synchronises the stream, allocates,
 *    or touches the legacy default stream (cf. reference nms_cuda.cu:186-214);
 *  - every function returns a d3d_status (0 = ok, <0 = error) instead of throwing
 *    py::value_error or exit()-ing (reference common.h:33-46);
 *  - `stream` is a hipStream_t passed as void* (NULL = null stream).
 */
#ifndef D3D_HIP_H
#define D3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    D3D_OK = 0,
    D3D_ERR_BAD_ARG = -1,       /* null pointer / negative size / bad enum       -> ValueError   */
    D3D_ERR_UNSUPPORTED = -2,   /* option the reference rejects too              -> ValueError   */
    D3D_ERR_WORKSPACE = -3,     /* workspace smaller than *_workspace_bytes()    -> RuntimeError */
    D3D_ERR_HIP = -4            /* a HIP call failed; see d3d_last_hip_error()   -> RuntimeError */
} d3d_status;

/* enum values are the reference's (d3d/voxel/voxelize.h:5-7, d3d/box/common.h:5-10) */
enum { D3D_REDUCE_NONE = 0, D3D_REDUCE_MEAN = 1, D3D_REDUCE_MAX = 2, D3D_REDUCE_MIN = 3 };
enum { D3D_MAXPTS_NONE = 0, D3D_MAXPTS_TRIM = 1, D3D_MAXPTS_FARTHEST_SAMPLING = 2 };
enum { D3D_MAXVOX_NONE = 0, D3D_MAXVOX_TRIM = 1, D3D_MAXVOX_DESCENDING = 2 };
enum { D3D_IOU_NA = 0, D3D_IOU_BOX = 1, D3D_IOU_RBOX = 2, D3D_IOU_GBOX = 3, D3D_IOU_GRBOX = 4,
       D3D_IOU_DBOX = 5, D3D_IOU_DRBOX = 6 };
enum { D3D_SUPPRESS_HARD = 0, D3D_SUPPRESS_LINEAR = 1, D3D_SUPPRESS_GAUSSIAN = 2 };
enum { D3D_F32 = 0, D3D_F64 = 1,
       /* d3d_iou2d_forward / d3d_iou2d_backward, BOX and RBOX only: boxes and box gradients f64, arithmetic f64, the [n,m] MATRIX
        * (ious written, grad read) f32 -- every value rounded where it is stored / widened where it is read.  The results of
        * box2d_iou(precise=True) on fp32 boxes (reference box/__init__.py:204-205, 224: boxes.double(), ious.to(dtype)) without
        * the fp64 copy of the matrix: a third of the bytes. */
       D3D_F64_M32 = 2,
       /* d3d_iou2d_forward / _backward (BOX / RBOX), d3d_nms2d: everything in memory f32 (boxes, scores, the matrix), the arithmetic f64 --
        * every value widened where it is loaded.  box2d_iou / box2d_nms (precise=True) on fp32 tensors without the .double()
        * copies either (box/__init__.py:204-205, 254-255). */
       D3D_F32_WIDE = 3 };

/* status bits OR-ed into counts[D3D_COUNT_STATUS] by the voxel kernels */
enum { D3D_VOXEL_STATUS_COORD_OVERFLOW = 1,   /* sparse: 2^20 <= |floor(p/size)| < 2^31 (NaN, inf and
                                                 larger values become the reference's INT_MIN coordinate) */
       D3D_VOXEL_STATUS_TABLE_FULL = 2,       /* internal hash table overflow (cannot happen
                                                 with the documented workspace size)           */
       D3D_VOXEL_STATUS_PACK_OVERFLOW = 4,    /* a voxel holds more points than the packed one-word
                                                 hash slot can count: results are invalid; repeat the
                                                 call with D3D_VOXEL_PLAIN_SLOTS                 */
       D3D_VOXEL_STATUS_BIN_OVERFLOW = 8 };   /* binned index: a bucket of the partition holds more distinct
                                                 cells than its LDS table (or > 2 M points): results are
                                                 invalid; repeat the call with D3D_VOXEL_PATH_HASH */
enum { D3D_COUNT_VOXELS = 0, D3D_COUNT_POINTS = 1, D3D_COUNT_STATUS = 2, D3D_COUNT_AUX = 3, D3D_NUM_COUNTS = 4 };

int         d3d_abi_version(void);
int         d3d_last_hip_error(void);           /* hipError_t of the last D3D_ERR_HIP */
const char *d3d_status_string(int status);

/* ------------------------------------------------------------------ d3d/voxel */

/* Per-call options of the voxel entry points that build an index (`flags`, last argument; 0 = automatic).  They are
 * arguments, not library state: two threads / streams may use different ones at the same time, and the retry a caller
 * issues after a PACK_ / BIN_OVERFLOW status touches nothing shared.
 *   D3D_VOXEL_PATH_HASH    hash table in HBM (any input) instead of the default binned index (points partitioned into
 *                          buckets, per-bucket index in LDS; up to 16 M points, grids below 2^32 - 1 cells).  Both give
 *                          identical outputs; see DESIGN.md section 4.
 *   D3D_VOXEL_PLAIN_SLOTS  (hash table) general two-word slots, any count / key width, instead of the default one-word
 *                          slots {count | key | first index} that are used whenever they fit 64 bits.
 *   D3D_VOXEL_SPLIT_FILL   (dense contract, binned index, C == 4) the ranked rows staged by the index, then per-voxel outputs
 *                          and voxels[V,P,C] from two launches (k_meta_first + k_fill_c4) instead of the fused k_emit, which
 *                          reads first rows straight from `points`.  Identical outputs.
 *   D3D_VOXEL_PARTITION_3PASS  (binned index) the partition of rounds 1-3 -- tile histograms, a scan over the tile x bucket
 *                          matrix, one scattered store per point: three launches -- instead of the one-launch tile sort
 *                          (round 4: every tile of 8192 points is sorted by bucket in LDS and written as one coalesced
 *                          run; the bucket workgroups gather their runs).  Identical outputs; frames above 4 M points
 *                          always take the three-pass partition.
 *   D3D_VOXEL_EXACT_MEAN   (dense contract, reduction MEAN) the aggregates of voxels with MORE than max_points points are summed
 *                          sequentially in fp32 in point order, bit for bit like voxelize.cpp:142,164 -- a post-pass (table of
 *                          those voxels, their points compacted in point order, a stable sort by voxel, one wavefront per
 *                          voxel adding in order; ~2x the call).  Default: those voxels are summed in fp64 in arrival order,
 *                          which differs from the reference by ITS rounding error (count * 2^-24 relative) only; voxels within
 *                          max_points, MAX and MIN are bit-exact either way. */
/*   D3D_VOXEL_WIDE_KEYS   (sparse contract) ANY int32 voxel coordinate: the hash table compares all 96 bits of (x, y, z) instead of
 *                          the default 3 x 21-bit key (finite coordinates in (-2^20, 2^20) per axis; beyond it the call comes back
 *                          with D3D_VOXEL_STATUS_COORD_OVERFLOW and is to be repeated with this flag).  Every input
 *                          voxelize.cpp:309 accepts then gives the reference's voxels.  A general path, slower than the default. */
enum { D3D_VOXEL_PATH_HASH = 1, D3D_VOXEL_PARTITION_3PASS = 2, D3D_VOXEL_PLAIN_SLOTS = 4, D3D_VOXEL_SPLIT_FILL = 8,
       D3D_VOXEL_EXACT_MEAN = 16, /* 32: retired */ D3D_VOXEL_WIDE_KEYS = 64, D3D_VOXEL_FLAGS_ALL = 95 };

/* scratch for any of the three voxel entry points on n points / nvox voxels */
size_t d3d_voxelize_workspace_bytes(int64_t n_points, int64_t n_voxels);

/* replaces voxelize_3d_dense (reference d3d/voxel/voxelize.h:9-12, voxelize.cpp:45-199).
 *   points[n,c] f32; shape[3] i32 (host); bound[6] f32 (host: xmin,xmax,ymin,ymax,zmin,zmax)
 *   outputs sized for cap = min(n, max_voxels) voxels:
 *   voxels[cap,max_points,c] f32, coords[cap,3] i64, pmask[cap,max_points] u8 (0/1),
 *   npoints[cap] i32, aggregates[cap,c] f32 (NULL iff reduction == NONE).
 *   Rows >= counts[D3D_COUNT_VOXELS] are not part of the result (the caller slices them off, cf. voxelize.cpp:167-179):
 *   coords / pmask / npoints / aggregates leave them untouched; `voxels` may hold ZEROS there, anywhere inside its cap rows
 *   (round 6: part of the tensor's zero padding is stored under the index launches, over a range fixed before the voxel
 *   count exists -- the buffers MUST have the cap rows stated above, not just the rows a caller expects to come back).
 *   counts: device int64[D3D_NUM_COUNTS]. */
int d3d_voxelize_3d_dense(const float *points, int64_t n, int32_t c,
                          const int32_t *shape, const float *bound,
                          int32_t max_points, int32_t max_voxels, int32_t reduction,
                          float *voxels, int64_t *coords, uint8_t *pmask, int32_t *npoints,
                          float *aggregates, int64_t *counts,
                          void *workspace, size_t workspace_bytes, void *stream, uint32_t flags);

/* d3d_voxelize_3d_dense that also publishes counts[] to the host as soon as they are final, i.e. BEFORE the
 * HBM-bound fill of voxels[V,P,C] is launched: host_counts[0 .. D3D_NUM_COUNTS) = counts, then
 * host_counts[D3D_NUM_COUNTS] = 1 (system-scope release).  host_counts = D3D_NUM_COUNTS + 1 int64 of host-mapped,
 * coherent pinned memory (hipHostMalloc / torch pin_memory) with the flag word cleared by the caller, who polls it:
 * the output sizes (the reference returns exactly-sized tensors, voxelize.cpp:166-180) reach the host while the GPU
 * is still writing the outputs, and the next call can be queued behind them without draining the stream. */
int d3d_voxelize_3d_dense_notify(const float *points, int64_t n, int32_t c,
                          const int32_t *shape, const float *bound,
                          int32_t max_points, int32_t max_voxels, int32_t reduction,
                          float *voxels, int64_t *coords, uint8_t *pmask, int32_t *npoints,
                          float *aggregates, int64_t *counts,
                          void *workspace, size_t workspace_bytes, void *stream, int64_t *host_counts, uint32_t flags);

/* The dense contract into a RESIDENT output buffer (beyond the reference, which returns fresh tensors per frame,
 * voxelize.cpp:166-180; for callers that consume voxels[V,P,C] on the device before the next frame arrives).
 *   voxels[capacity, max_points, c] f32 and row_state[capacity] u16: caller-owned, kept from frame to frame, zero-filled by
 *   the caller ONCE and never written by it; capacity >= min(n, max_voxels) of every call; the same max_points and c in every call.
 * Invariant kept by the call: row r of voxels[v] is zero for r >= row_state[v].  After the call voxels[0 .. V) hold exactly
 * what d3d_voxelize_3d_dense writes (bit for bit, padding included) -- but only the rows that hold points, and zeros over the
 * rows the previous occupant of the same voxel id held, are stored: the zero padding (95 % of the tensor on a LiDAR frame:
 * 1.6 points per voxel at max_points 32) is already there.  The result aliases the buffer: it is valid until the next call
 * on it.  coords / pmask / npoints / aggregates / counts / host_counts (may be NULL) as in d3d_voxelize_3d_dense_notify.
 * D3D_ERR_UNSUPPORTED, with nothing touched: rows of more than 8 floats, buffers not 16-byte aligned, max_points > 256 (c != 4:
 * also max_points * c not a multiple of 4), a frame the binned index does not take (D3D_VOXEL_PATH_HASH, more than 8 M points),
 * D3D_VOXEL_SPLIT_FILL. */
int d3d_voxelize_3d_dense_resident(const float *points, int64_t n, int32_t c,
                          const int32_t *shape, const float *bound,
                          int32_t max_points, int32_t max_voxels, int32_t reduction,
                          float *voxels, uint16_t *row_state, int64_t *coords, uint8_t *pmask, int32_t *npoints,
                          float *aggregates, int64_t *counts,
                          void *workspace, size_t workspace_bytes, void *stream, int64_t *host_counts, uint32_t flags);

/* d3d_voxelize_3d_dense_notify split for a caller that pipelines a stream of frames (the reference is a per-frame loop,
 * voxelize.cpp:94; its callers feed it frame after frame): stage 1 = the index launches, stage 2 = the output launch, stage 0 =
 * both.  The two stages may go to two streams (event between them), so that frame k + 1's index -- latency-bound, a few tens of
 * MB -- runs under frame k's output -- bandwidth-bound.  Same arguments in both calls; the workspace carries the index (one
 * workspace per frame in flight).  host_counts may be NULL.  D3D_ERR_UNSUPPORTED for stage != 0 where the output is not one
 * launch (C != 4, unaligned buffers, max_points not a multiple of 16 or above 256, a frame the binned index does not take). */
int d3d_voxelize_3d_dense_staged(const float *points, int64_t n, int32_t c,
                          const int32_t *shape, const float *bound,
                          int32_t max_points, int32_t max_voxels, int32_t reduction,
                          float *voxels, int64_t *coords, uint8_t *pmask, int32_t *npoints,
                          float *aggregates, int64_t *counts,
                          void *workspace, size_t workspace_bytes, void *stream, int64_t *host_counts, uint32_t flags,
                          int32_t stage);

/* replaces voxelize_sparse, bound in Python as voxelize_3d_sparse
 * (reference voxelize.h:14-17, voxelize.cpp:288-335, impl.cpp:5).
 *   points[n,c] f32 (c >= 3); voxel_size[3] f32 (host)
 *   points_mapping[n] i64, coords[n,3] i64 (first counts[0] rows valid), npoints[n] i32. */
int d3d_voxelize_3d_sparse(const float *points, int64_t n, int32_t c, const float *voxel_size,
                           int64_t *points_mapping, int64_t *coords, int32_t *npoints,
                           int64_t *counts, void *workspace, size_t workspace_bytes, void *stream, uint32_t flags);

/* replaces voxelize_filter, bound as voxelize_3d_filter
 * (reference voxelize.h:19-25, voxelize.cpp:337-484).
 *   feats[n,c] f32, points_mapping[n] i64, coords[nvox,3] i64, voxel_npoints[nvox] i32,
 *   coords_bound[6] i64 (host: row-major [3][2] = lo,hi per axis)
 *   out_feats[n,c], out_mask[n] i64, out_mapping[n] i64, out_npoints[nvox] i32,
 *   out_coords[nvox,3] i64;  counts[D3D_COUNT_POINTS] kept points, counts[D3D_COUNT_VOXELS]
 *   kept voxels.  max_points_filter == FARTHEST_SAMPLING -> D3D_ERR_UNSUPPORTED
 *   (reference throws at voxelize.cpp:469-471). */
int d3d_voxelize_3d_filter(const float *feats, int64_t n, int32_t c,
                           const int64_t *points_mapping, const int64_t *coords,
                           const int32_t *voxel_npoints, int64_t nvox, const int64_t *coords_bound,
                           int32_t min_points, int32_t max_points, int32_t max_voxels,
                           int32_t max_points_filter, int32_t max_voxels_filter,
                           float *out_feats, int64_t *out_mask, int64_t *out_mapping,
                           int32_t *out_npoints, int64_t *out_coords,
                           int64_t *counts, void *workspace, size_t workspace_bytes, void *stream);

/* d3d_voxelize_3d_filter directly behind d3d_voxelize_3d_sparse on the same stream, without reading the voxel count
 * back in between (VoxelGenerator.__call__, voxel/__init__.py:93-102, does exactly this pair): coords / voxel_npoints
 * are the sparse call's buffers with nvox_rows rows, sparse_counts its device `counts`; rows >= counts[0] are ignored.
 * max_voxels_filter = DESCENDING is unsupported here (the sort needs the size on the host). */
int d3d_voxelize_3d_filter_chained(const float *feats, int64_t n, int32_t c, const int64_t *points_mapping,
                                   const int64_t *coords, const int32_t *voxel_npoints, int64_t nvox_rows,
                                   const int64_t *sparse_counts, const int64_t *coords_bound,
                                   int32_t min_points, int32_t max_points, int32_t max_voxels,
                                   int32_t max_points_filter, int32_t max_voxels_filter,
                                   float *out_feats, int64_t *out_mask, int64_t *out_mapping,
                                   int32_t *out_npoints, int64_t *out_coords,
                                   int64_t *counts, void *workspace, size_t workspace_bytes, void *stream);

/* VoxelGenerator.__call__'s sparse branch (reference voxel/__init__.py:93-102) in one call: d3d_voxelize_3d_sparse
 * followed by d3d_voxelize_3d_filter on its outputs (same arguments, same outputs as the two calls; the voxel count
 * stays on the device.  max_voxels_filter DESCENDING, round 4: fused as well wherever the binned index runs -- a stable sort
 * of the passing voxels' counts on the device decides the ranks, the output sizes reach host_counts from the LAST launch;
 * D3D_ERR_UNSUPPORTED for it elsewhere: issue the two calls).  Besides saving the round trip it
 * lets the TRIM point filter reuse the per-voxel index ranking the sparse index already holds (voxelize.cpp:457-463)
 * and the voxel filter run inside the index (up to 16 M points, filters NONE / TRIM): points_mapping, coords and npoints
 * are then scratch (not materialised), sparse_counts holds the status bits.
 * Non-finite points get the reference's INT_MIN coordinate on that axis (x86 (int)floor(NaN), voxelize.cpp:309) and are
 * then removed by the coordinate-bound filter exactly as there (:376-384).  Finite points outside the 3 x 21-bit key range
 * (2^20 <= |floor(p/size)| < 2^31) are DROPPED here as long as coords_bound lies inside [-2^20, 2^20] -- the same filter
 * would remove their far-away voxel; d3d_voxelize_3d_sparse alone raises D3D_VOXEL_STATUS_COORD_OVERFLOW for them.
 * Workspace: d3d_voxelize_workspace_bytes(n, n).  host_counts: NULL, or 2 * D3D_NUM_COUNTS + 1 int64 of host-mapped
 * pinned memory with word [D3D_NUM_COUNTS] cleared: receives sparse_counts in [0, 4), counts in [5, 9) and then the flag
 * [4] = 1 before the last kernel (the compaction of the kept points) is launched -- see d3d_voxelize_3d_dense_notify.
 * coord_offset: NULL, or 3 host values subtracted from every row of out_coords -- the `ret.coords - self._offset` that ends
 * VoxelGenerator.__call__ (voxel/__init__.py:103), done where the rows are written instead of by one more launch. */
int d3d_voxelize_3d_sparse_filter(const float *points, int64_t n, int32_t c, const float *voxel_size,
                                  const int64_t *coords_bound, int32_t min_points, int32_t max_points,
                                  int32_t max_voxels, int32_t max_points_filter, int32_t max_voxels_filter,
                                  int64_t *points_mapping, int64_t *coords, int32_t *npoints, int64_t *sparse_counts,
                                  float *out_feats, int64_t *out_mask, int64_t *out_mapping, int32_t *out_npoints,
                                  int64_t *out_coords, int64_t *counts, void *workspace, size_t workspace_bytes,
                                  void *stream, int64_t *host_counts, uint32_t flags, const int64_t *coord_offset);

/* The same call through ONE prepared argument block (round 6): a caller that voxelizes frame after frame fills the block once
 * per generator (sizes, bounds, filters -- what VoxelGenerator.__init__ derives, voxel/__init__.py:16-77) and per frame sets
 * `points`, `n` and `outputs`.  Binding the 26-argument form costs a ctypes / cgo caller more host time per call than the three
 * launches it makes; the intermediates (points_mapping, coords, npoints, both count rows) live at the front of the workspace
 * instead of in caller tensors, the five outputs are carved from ONE caller allocation:
 *   d3d_voxelize_3d_sparse_filter_call_layout(n, c, offsets) -> bytes of `outputs`; offsets[0..4] = byte offsets of
 *     points f32[n,c], points_mask i64[n], points_mapping i64[n], voxel_npoints i32[n], coords i64[n,3] (rows [0, kept) valid);
 *   d3d_voxelize_3d_sparse_filter_call_workspace_bytes(n) -> workspace bytes (the device count rows: its first
 *     2 * D3D_NUM_COUNTS int64 -- sparse_counts, then counts);
 *   d3d_voxelize_3d_sparse_filter_call(call) = d3d_voxelize_3d_sparse_filter on those buffers, same status codes. */
typedef struct D3DSparseFilterCall {
    const float *points;
    int64_t n;
    int32_t c, min_points, max_points, max_voxels, max_points_filter, max_voxels_filter;
    float voxel_size[3];
    uint32_t flags;
    int64_t coords_bound[6];
    int64_t coord_offset[3];
    int32_t has_coord_offset, reserved;
    void *outputs;
    size_t outputs_bytes;
    void *workspace;
    size_t workspace_bytes;
    void *stream;
    int64_t *host_counts;
} D3DSparseFilterCall;
size_t d3d_voxelize_3d_sparse_filter_call_layout(int64_t n, int32_t c, size_t *offsets5);
size_t d3d_voxelize_3d_sparse_filter_call_workspace_bytes(int64_t n);
int d3d_voxelize_3d_sparse_filter_call(const D3DSparseFilterCall *call);

/* ---- beyond the reference: the point-sharded voxelizer of north_star (d3d has no distributed code) ---- */

/* Voxel feature grid without the dense [V,P,C] copy ("dynamic voxelization"): grid semantics of
 * d3d_voxelize_3d_dense (voxelize.cpp:100-101), first-seen voxel ids (voxelize.cpp:119), reduction over
 * ALL in-range points (voxelize.cpp:137-164).  reduction: MEAN/MAX/MIN or 4 = SUM (MEAN without the division).
 *   coords[n,3] i64 (may be NULL when keys is given), npoints[n] i32, aggregates[n,c] f32, first[n] i64 (index_offset + index of the voxel's first
 *   point; may be NULL), mapping[n] i64 (voxel id per point, -1 = out of range; may be NULL),
 *   keys[n + 1] i64 (linear cell index (x*sy+y)*sz+z per voxel, -1 in the rows >= counts[0]; keys[n] = -1 - status
 *   bits of counts[2], so that the status travels with the key list; may be NULL).
 *   Optionally (C == 4, 16-byte aligned points / aggregates / rows) the ranked rows themselves: rows[d3d_voxelize_reduce_rows(n), 4]
 *   f32 receives, per voxel, its first min(count, max_points) points in point order at row seg_base[voxel] (seg_base[n] u32);
 *   max_points = 0 and seg_base = rows = NULL otherwise.  The sharded dense contract sends them to the voxel's owner. */
size_t d3d_voxelize_reduce_rows(int64_t n);
int d3d_voxelize_3d_reduce(const float *points, int64_t n, int32_t c, const int32_t *shape, const float *bound,
                           int32_t reduction, int64_t index_offset, int64_t *coords, int32_t *npoints,
                           float *aggregates, int64_t *first, int64_t *mapping, int64_t *keys,
                           int32_t max_points, uint32_t *seg_base, float *rows, int64_t *counts,
                           void *workspace, size_t workspace_bytes, void *stream, uint32_t flags);

/* Rank-independent compact numbering of occupied cells: mark keys[m] (linear cell index in [0,ncells)) in a
 * bitmap, popcount-prefix it; counts[0] = distinct occupied cells.  lookup: slot[j] = index of keys[j] among the
 * occupied cells in ascending key order (`missing` when unmarked).  Both use the same workspace. */
size_t d3d_grid_compact_workspace_bytes(int64_t ncells);
int d3d_grid_compact_index(const int64_t *keys, int64_t m, int64_t ncells, int64_t *counts,
                           void *workspace, size_t workspace_bytes, void *stream);
int d3d_grid_compact_lookup(const int64_t *keys, int64_t m, int64_t ncells, const void *workspace,
                            size_t workspace_bytes, int64_t missing, int64_t *slot, void *stream);

/* The same index from all-gathered occupancy BITMAPS instead of key lists (for grids whose bitmap, ncells / 8 bytes, is
 * smaller than the key lists: an OR pass replaces one atomic per gathered key): bitmap_mark = this rank's bitmap
 * (ceil(ncells/64) words); compact_from_bitmaps = OR of `world` bitmaps (rank r at parts + r * stride_words) + prefix,
 * counts[0] = distinct cells; compact_keys = the cell of every slot, ascending (key_of_slot[counts[0]]).
 * With owner_ws (d3d_grid_owner_workspace_bytes; world <= 16) compact_from_bitmaps also records, for `rank`, which
 * cells a lower rank has and how many cells every rank OWNS (a cell belongs to the lowest rank that has it -- the rank
 * holding the voxel's first point, shards being contiguous point ranges in rank order). */
int d3d_grid_bitmap_mark(const int64_t *keys, int64_t m, int64_t ncells, unsigned long long *bitmap, void *stream);
size_t d3d_grid_owner_workspace_bytes(int64_t ncells);
int d3d_grid_compact_from_bitmaps(const unsigned long long *parts, int64_t stride_words, int32_t world, int64_t ncells,
                                  int64_t *counts, void *workspace, size_t workspace_bytes, int32_t rank,
                                  void *owner_ws, size_t owner_ws_bytes, void *stream);
int d3d_grid_compact_keys(int64_t ncells, const void *workspace, size_t workspace_bytes, int64_t *key_of_slot,
                          void *stream);

/* Numbering by ownership (bitmap exchange, nvox < 2^24): the voxels a rank owns, in its local first-seen order, are a
 * contiguous run of the global first-seen order, and the runs follow each other in rank order.  scatter_owned writes the
 * reduction's identity into table[nvox, table_stride] (MEAN: c sums + count + id column; else c extrema + id column,
 * counts in cnt_table), then this rank's partial rows at their slots and, for the voxels it owns, the global voxel id
 * in the LAST column; slot_of_local[n_local].  scan_ws: (ceil(n_local/1024) + 2) * 8 + 1024 bytes.  After the all-reduce of
 * the table (same op as the features) finalize_owned reads the id from that column: no exchange of first indices. */
int d3d_sharded_scatter_owned(const int64_t *keys_local, int64_t n_local, int64_t ncells, const void *compact_ws,
                              size_t compact_ws_bytes, const void *owner_ws, size_t owner_ws_bytes, int32_t rank,
                              int64_t nvox, int32_t c, int32_t reduction, const float *agg, const int32_t *cnt,
                              float *table, int32_t table_stride, int32_t *cnt_table, int64_t *slot_of_local,
                              void *scan_ws, size_t scan_ws_bytes, void *stream);
int d3d_sharded_finalize_owned(int64_t nvox, int32_t c, const int64_t *key_of_slot, const float *table,
                               int32_t table_stride, int32_t mean, const int32_t *cnt_in, const int32_t *shape,
                               int64_t *vid_of_slot, int64_t *coords, int32_t *cnt_out, float *feats, void *stream);

/* Steps of the sharded voxelizer around the two all-reduces (no host synchronisation).
 * scatter: keys_all[m] = all-gathered key lists (negative = padding / status rows), already indexed by
 *   d3d_grid_compact_index into compact_ws; rows [begin, begin + n_local) are this rank's own.  Writes the identity
 *   of the reduction into table[nvox, table_stride] (MEAN: c sums + count, else c extrema with the counts in
 *   cnt_table[nvox]) and first[nvox] (INT64_MAX), then this rank's partial rows at their slots, key_of_slot[nvox]
 *   for every gathered key (skipped when NULL), and slot_of_local[n_local] (-1 beyond the rank's voxels).
 * finalize: voxel id of a slot = rank of first[slot] among all first indices (compact_ws sized for n_total cells is
 *   overwritten); slot-ordered all-reduced table -> voxel-id-ordered coords[nvox,3], cnt_out[nvox], feats[nvox,c],
 *   and vid_of_slot[nvox].
 * map: global voxel id of every local point: gmap[i] = vid_of_slot[slot_of_local[local_map[i]]] (-1 stays -1). */
int d3d_sharded_scatter(const int64_t *keys_all, int64_t m, int64_t begin, int64_t n_local, int64_t ncells,
                        const void *compact_ws, size_t compact_ws_bytes, int64_t nvox, int32_t c, int32_t reduction,
                        const float *agg, const int32_t *cnt, const int64_t *first_local, float *table,
                        int32_t table_stride, int32_t *cnt_table, int64_t *first, int64_t *key_of_slot,
                        int64_t *slot_of_local, void *stream);
int d3d_sharded_finalize(int64_t nvox, int32_t c, const int64_t *first, int64_t n_total, int64_t *counts,
                         void *compact_ws, size_t compact_ws_bytes, const int64_t *key_of_slot, const float *table,
                         int32_t table_stride, int32_t mean, const int32_t *cnt_in, const int32_t *shape,
                         int64_t *vid_of_slot, int64_t *coords, int32_t *cnt_out, float *feats, void *stream);
int d3d_sharded_map(int64_t n, const int64_t *local_map, const int64_t *slot_of_local, int64_t nvox,
                    const int64_t *vid_of_slot, int64_t *gmap, void *stream);

/* ---- point-sharded voxelizer, owner-computes exchange (owner.hip; d3d_amd/voxel/sharded.py drives it) ----
 * Every grid cell has one owner rank (a hash of the cell).  A rank sends the partial record of each of its local voxels to
 * the cell's owner (all-to-all), the owner merges the <= world records of a cell in rank order, numbers its voxels in the
 * frame's first-seen order (voxelize.cpp:119) from a one-bit-per-point bitmap whose SUM all-reduce is the OR of the owners'
 * disjoint bit sets, and finishes 1/world of the frame's voxels.  A record is d3d_owner_record_words(c) int32 words:
 * cell key (2), first global point index (2), count (1), c partial features (float bits), padding to an even count. */
int    d3d_owner_record_words(int32_t c);
size_t d3d_owner_pack_workspace_bytes(int64_t n, int32_t world);
/* outputs of d3d_voxelize_3d_reduce (keys[n + 1], cnt[n], agg[n, c], first[n], its counts) -> send[n, words] grouped by owner
 * rank, perm[n] (send position -> local voxel), pos_of_local[n] (its inverse), send_counts[2 world + 1] (device: records per
 * destination, the shard's status bits, rows per destination).  Dense contract (max_points > 0, c == 4): seg_base / rows_local
 * as left by d3d_voxelize_3d_reduce(max_points, ...) -> send_rows[kept rows, 4] with the same grouping, and a record's last word
 * = offset of its rows inside its (source, destination) batch; otherwise max_points = 0 and the three pointers NULL.
 * points / index_offset: the shard that call voxelized -- on the binned index it leaves, instead of rows, the voxels' ranked
 * point INDICES in `rows` (counts[D3D_COUNT_AUX] = 1) and the rows are gathered here; NULL = rows_local holds rows. */
int d3d_owner_pack(const int64_t *keys, const int32_t *cnt, const float *agg, const int64_t *first, const int64_t *counts,
                   int64_t n, int32_t c, int32_t world, int32_t max_points, const uint32_t *seg_base, const float *rows_local,
                   int32_t *send, int32_t *perm, int32_t *pos_of_local, float *send_rows, int64_t *send_counts,
                   void *workspace, size_t workspace_bytes, void *stream, const float *points, int64_t index_offset);
size_t d3d_owner_merge_workspace_bytes(int64_t n_records, int32_t world);
/* recv[R, words] grouped by source rank (src_off[world + 1], device) -> this owner's voxels in GLOBAL ID ORDER, finished
 * (the lowest source rank of a cell holds its first point; the records of one source follow the shard's first-seen order):
 * first_o / coords / npoints / feats (R rows allocated, counts[D3D_COUNT_VOXELS] valid) and rec_owned[R] = owned voxel of
 * every record, lead_rec[R] = leader record of every owned voxel.  reduction: MEAN (sums in rank order, then
 * voxelize.cpp:164's division), MAX, MIN.
 * flags: 0 = up to 2 M records the cells are grouped in LDS, bucket by bucket (three launches, no global atomics); should a
 * bucket not fit (hashed cells: it does) counts[D3D_COUNT_STATUS] carries D3D_VOXEL_STATUS_BIN_OVERFLOW, nothing else is valid
 * and the call is to be repeated with D3D_OWNER_MERGE_CHAINS = the general path (a global hash table with a record chain per
 * cell; taken by itself above 2 M records).  D3D_OWNER_MERGE_TEST_TINY: test hook, buckets overflow at 4 records. */
enum { D3D_OWNER_MERGE_CHAINS = 1, D3D_OWNER_MERGE_TEST_TINY = 2 };
int d3d_owner_merge(const int32_t *recv, int64_t n_records, const int64_t *src_off, int32_t world, int32_t c, int32_t reduction,
                    const int32_t *shape, int64_t *first_o, int64_t *coords, int32_t *npoints, float *feats,
                    int32_t *rec_owned, int32_t *lead_rec, int64_t *counts, void *workspace, size_t workspace_bytes, void *stream,
                    uint32_t flags, const int64_t *point_off);
/* point_off (device, [world]; may be NULL): when the ranks ran d3d_voxelize_3d_reduce with index_offset 0 -- no rank needs the
 * others' shard sizes before its local pass -- the records carry indices local to their source's shard, and point_off[s] =
 * global index of rank s's first point turns the leader's into the voxel's global first point (first_o).  NULL: global already. */
/* dense contract on the owner (voxelize.cpp:128-134: the first max_points points of a voxel by global index = the ranks'
 * candidate rows in rank order): after d3d_owner_merge, with ITS workspace untouched since and ITS flags; recv_rows[*, 4] grouped by source
 * rank (rows_src_off[world + 1], device); lead_rec / npoints / counts_o from d3d_owner_merge.
 * -> voxels[cap_o, max_points, 4], pmask[cap_o, max_points] of the owned voxels in id order (max_points <= 256, c == 4).
 * row_state: NULL, or the resident form of d3d_voxelize_3d_dense_resident -- voxels[capacity >= cap_o, max_points, 4] and
 * row_state[capacity] kept by the caller from frame to frame (zero-filled once): only the rows that hold points and the rows the
 * previous frame left under the same owned id are stored, the zero padding stays; same values in voxels[0 .. Vo). */
int d3d_owner_dense(const int32_t *recv, int64_t n_records, const float *recv_rows, const int64_t *rows_src_off, int32_t world,
                    int32_t max_points, const int32_t *lead_rec, const int32_t *npoints, const int64_t *counts_o, int64_t cap_o,
                    const void *merge_workspace, size_t merge_workspace_bytes, float *voxels, uint8_t *pmask, void *stream,
                    uint32_t flags, uint16_t *row_state);
/* bitmap[(n_total + 63) / 64 + 1] <- bit f for every owned voxel's first point f; the last word = 1 when counts_o carries
 * BIN_OVERFLOW (d3d_owner_merge), else 0 */
int d3d_owner_mark_first(const int64_t *first_o, const int64_t *counts_o, int64_t cap_o, int64_t n_total, uint64_t *bitmap,
                         void *stream);
size_t d3d_owner_number_workspace_bytes(int64_t n_total);
/* global_bits = SUM all-reduce of all owners' bitmaps (with their last word).  vids[i] = global voxel id of owned voxel i;
 * counts_out[D3D_COUNT_VOXELS] = voxels of the whole frame; counts_out[D3D_COUNT_STATUS] = BIN_OVERFLOW when ANY owner's merge
 * has to be repeated (every rank sees the same word: they repeat together). */
int d3d_owner_number(const uint64_t *global_bits, int64_t n_total, const int64_t *first_o, const int64_t *counts_o, int64_t cap_o,
                     int64_t *vids, int64_t *counts_out, void *workspace, size_t workspace_bytes, void *stream);
/* reply[i] = global voxel id of received record i (returned to the record's source rank by the reverse all-to-all) */
int d3d_owner_reply(int64_t n_records, const int32_t *rec_owned, const int64_t *vids, int64_t *reply, void *stream);
/* back[] (ids returned, in send order), pos_of_local (d3d_owner_pack), local_map[n] (point -> local voxel) -> gmap[n] */
int d3d_owner_map(int64_t n, const int64_t *local_map, const int32_t *pos_of_local, const int64_t *back, int64_t *gmap,
                  void *stream);
/* all owners' finished rows -> the replicated feature grid in voxel-id order.  src_off[world + 1] (device; may be NULL): the rows
 * are the ranks' blocks one after the other, block s = rows [src_off[s], src_off[s + 1]), each in ascending id order (what an
 * all-gather of the owners' results delivers) -- merged with coalesced reads and writes (workspace: (nvox / 1024 + 2) * world * 8
 * bytes); NULL: any order, one scattered row each. */
int d3d_owner_replicate(int64_t nvox, const int64_t *vids, const int64_t *coords_in, const int32_t *cnt_in,
                        const float *feats_in, int32_t c, int64_t *coords, int32_t *cnt, float *feats, void *stream,
                        const int64_t *src_off, int32_t world, void *workspace, size_t workspace_bytes);

/* ---- points in 3D boxes: Target3DArray.crop_points / paint_label (reference d3d/abstraction.pyx:308-324,
 * 654-687), per pair box3dr_contains (d3d/dgal_wrap.h:6-19): closed z interval in fp32, the rotated rectangle's bounding
 * box, the rectangle.  points[n, point_stride >= 3] f32 (x, y, z first); box row i = (x, y, z, lx, ly, lz, rz) at
 * boxes + i * box_stride + box_offset ([M,7]: 7, 0; the [n,9] rows of Target3DArray.to_numpy: 9, 2). */
int d3d_crop_3dr(const float *points, int64_t n, int32_t point_stride, const float *boxes, int64_t m, int32_t box_stride,
                 int32_t box_offset, uint8_t *out /* [m, n] 0/1 */, void *stream);
/* idarr[n] u16 = 1 + the lowest index of a box that contains the point and whose class labels[i] (u8) equals semantics[j]
 * (u8), 0 where there is none -- what the reference's descending paint loop leaves (abstraction.pyx:662-673) -- without the
 * bool[m, n] mask. */
int d3d_paint_label(const float *points, int64_t n, int32_t point_stride, const uint8_t *semantics, const float *boxes,
                    int64_t m, int32_t box_stride, int32_t box_offset, const uint8_t *labels, uint16_t *idarr, void *stream);

/* ------------------------------------------------------------------ d3d/point ("next" row) */

/* replaces aligned_scatter_forward[_cuda] / aligned_scatter_backward[_cuda] (reference d3d/point/scatter.h:39-56,
 * scatter.cpp:81-200, scatter_cuda.cu).  coord[n, dim+1] (batch index first), image[batch, channels, dims[0..dim-1]],
 * out / grad [n, channels], all in `dtype`; dims: host int64[dim]; align_type 1 = MEAN, 2 = LINEAR (others ->
 * D3D_ERR_UNSUPPORTED like the reference's py::value_error).  backward ACCUMULATES into image_grad (atomics).
 * workspace (d3d_aligned_scatter_workspace_bytes; may be NULL): a channels-last copy of the map / of the gradient
 * accumulator, which turns the per-channel requests of a point's neighbours into coalesced ones (channels >= 8). */
size_t d3d_aligned_scatter_workspace_bytes(int64_t batch, int64_t channels, const int64_t *dims, int32_t dim, int32_t dtype);
int d3d_aligned_scatter_forward(const void *coord, int64_t n, int32_t dim, const void *image, int64_t batch,
                                int64_t channels, const int64_t *dims, int32_t align_type, int32_t dtype, void *out,
                                void *workspace, size_t workspace_bytes, void *stream);
int d3d_aligned_scatter_backward(const void *coord, int64_t n, int32_t dim, const void *grad, int64_t batch,
                                 int64_t channels, const int64_t *dims, int32_t align_type, int32_t dtype,
                                 void *image_grad, void *workspace, size_t workspace_bytes, void *stream);

/* opt-in per-kernel timing with HIP events on the launch stream (bench.py's roofline leg);
 * report: "kernel,calls,total_ms" lines. */
int d3d_profile_enable(int on);
int d3d_profile_report(char *buf, size_t buf_bytes);

/* what the calling thread's last d3d_voxelize_3d_dense[_notify] launched (measurement only, bench.py): out4[0] = 1 when the
 * output left through the two-role launch (k_emit_split), out4[1] = voxel ids whose zero padding was stored under the index
 * launches, out4[2] / out4[3] = bytes of zeros the filler workgroups of k_tile_sort / k_first_count stored.  No reference
 * counterpart (voxelize.cpp:56-59 zero-fills with torch::zeros). */
int d3d_voxelize_dense_last_plan(int64_t *out4);

/* stream-bandwidth probe on the caller's buffer (bench.py: "fraction of the measured copy bandwidth of the same box",
 * SURVEY 8d).  mode 0 = nontemporal 16-byte stores over `bytes`, 1 = copy first half -> second half, 2 = read sweep,
 * 3 = hipMemsetAsync (the runtime's fill, for reference), 4 / 5 = nontemporal stores in the launch shapes of the output kernel /
 * of the IoU fill, 6 = an empty launch (the overhead of the event pair that d3d_profile_* puts around every launch). */
int d3d_stream_probe(int mode, void *buf, size_t bytes, void *stream);

/* -------------------------------------------------------------------- d3d/box */

/* replaces iou2d_forward[_cuda] (iou_type BOX), the `ious` output of iou2dr_forward[_cuda] (RBOX), of giou2dr_forward[_cuda]
 * (GRBOX) and of diou2dr_forward[_cuda] (DRBOX)  (reference d3d/box/iou.h:7-69, iou.cpp:12-46,95-141,213-258,322-367,
 * iou_cuda.cu:10-48,100-151,216-440).  boxes1[n,5], boxes2[m,5] = (x,y,w,h,r) in `dtype`;
 * ious[n,m] in `dtype`, row-major.  64-bit pair indexing (cf. iou_cuda.cu:36,137).  The workspace is optional
 * (NULL/0 selects the single-kernel path); with it RBOX runs as zero-fill + candidate list + dense clipping, and GRBOX /
 * DRBOX of more than 65536 pairs as a pair kernel for the boxes that are apart (hull / diameter only) + a list of the pairs
 * that need the clip or the tie rules + one listed pair per lane.  A pair's value is the same on every path.
 * Matrices of up to 65536 pairs (BOX / RBOX) take one launch with one pair per lane, with or without a workspace.
 * GIoU = IoU - (H - U) / H, H = area of the convex hull of the two rectangles, U = union area; DIoU = IoU - d^2 / D^2,
 * d = distance of the centres, D = diameter of that hull (dgal's source is not vendored: the published definitions);
 * a rectangle of non-positive area gives 0 for every type.  GBOX / DBOX are D3D_ERR_UNSUPPORTED (the reference's Python
 * layer raises "Unrecognized iou type!" for them, box/__init__.py:216-217).
 * dtype D3D_F64_M32 (BOX / RBOX; GRBOX / DRBOX: D3D_ERR_UNSUPPORTED): boxes f64, ious f32; above 65536 pairs the workspace is
 * required (D3D_ERR_WORKSPACE without it; size: the query with D3D_F64_M32). */
size_t d3d_iou2d_workspace_bytes(int64_t n, int64_t m, int32_t dtype);
int d3d_iou2d_forward(const void *boxes1, int64_t n, const void *boxes2, int64_t m,
                      int32_t iou_type, int32_t dtype, void *ious,
                      void *workspace, size_t workspace_bytes, void *stream, uint32_t flags);
/* flags of d3d_iou2d_forward: 0, or D3D_IOU_LIST_CAP(k) = use only k entries of the candidate list (tests of the
 * overflow -> single-kernel fallback) */
#define D3D_IOU_LIST_CAP(k) ((uint32_t)(k) << 8)

/* replaces iou2d_backward[_cuda] (BOX), iou2dr_backward[_cuda] (RBOX), giou2dr_backward[_cuda] (GRBOX) and
 * diou2dr_backward[_cuda] (DRBOX)  (reference iou.h:14-69, iou.cpp:48-93,143-211,260-320,369-419): grad[n,m] ->
 * grad_boxes1[n,5], grad_boxes2[m,5] (overwritten), all in `dtype`.  The flags the reference saves in forward (nx,
 * xflags, ...) are not needed: the geometry is recomputed analytically.
 * Workspace: d3d_iou2d_workspace_bytes(n, m, dtype); required for BOX / RBOX, optional for GRBOX / DRBOX (with it, matrices
 * of more than 65536 pairs take a gradient kernel for the pairs that are apart and the complete routine for the rest).
 * dtype D3D_F64_M32 (BOX / RBOX): boxes and grad_boxes f64, grad[n,m] f32 (widened as it is read); D3D_F32_WIDE: boxes f32 too,
 * grad_boxes still f64 (the sums are kept in f64: round them once).  grad_boxes2 == grad_boxes1 + 5 n (one buffer): cleared by
 * one launch. */
int d3d_iou2d_backward(const void *boxes1, int64_t n, const void *boxes2, int64_t m, const void *grad,
                       int32_t iou_type, int32_t dtype, void *grad_boxes1, void *grad_boxes2,
                       void *workspace, size_t workspace_bytes, void *stream);

/* the autograd bookkeeping outputs of iou2dr_forward (nx, xflags: iou.cpp:125-141), giou2dr_forward (nxm = {nx, nm},
 * xmflags = {xflags, mflags}: iou.cpp:243-258) and diou2dr_forward (nxd = {nx, far[0], far[1]}, xflags: iou.cpp:352-367).
 * Any output pointer may be NULL.  Per pair (i, j), row-major:
 *   nx[n,m]        vertex count of the intersection polygon (0 when there is none)
 *   xflags[n,m,8]  origin of its vertices, CCW: 0x00 | c = corner c of box 1 (inside box 2); 0x10 | c = corner c of box 2;
 *                  0x20 | e1 << 2 | e2 = crossing of edge e1 of box 1 (corner e1 -> e1 + 1) with edge e2 of box 2; 0xff unused
 *   nm[n,m], mflags[n,m,8]   vertex count of the convex hull of the two rectangles and the corner (0..3 box 1, 4..7 box 2)
 *                  at every hull vertex, CCW from the lowest (x, y); 0xff unused
 *   far[n,m,2]     the two corners (same numbering, far[0] < far[1]) farthest apart -- the hull's diameter
 * (dgal's own flag encoding is not published; this library's backward ignores the arrays.)  xflags / mflags 8-byte aligned. */
int d3d_iou2dr_flags(const void *boxes1, int64_t n, const void *boxes2, int64_t m, int32_t dtype, uint8_t *nx,
                     uint8_t *xflags, uint8_t *nm, uint8_t *mflags, uint8_t *far, void *stream);

/* replaces pdist2dr_forward[_cuda] / pdist2dr_backward[_cuda] (reference d3d/box/dist.h:7-20, dist.cpp:10-110,
 * dist_cuda.cu:10-80; Python box2dr_pdist / box3dr_pdist, box/__init__.py:333-381): SIGNED distance from points[n,2] to the
 * boundary of boxes[m,5], positive inside; dist[m,n] (box-major, as dist.cpp:39), iedge[m,n] (may be NULL) = nearest edge k
 * (corner k -> k + 1) or 4 + k when the nearest boundary point is corner k.  backward: grad[m,n] -> grad_boxes[m,5],
 * grad_points[n,2] (both overwritten; the reference's CUDA kernel accumulates them with a data race, dist_cuda.cu:78-79). */
int d3d_pdist2dr_forward(const void *points, int64_t n, const void *boxes, int64_t m, int32_t dtype, void *dist,
                         uint8_t *iedge, void *stream);
int d3d_pdist2dr_backward(const void *points, int64_t n, const void *boxes, int64_t m, const void *grad, int32_t dtype,
                          void *grad_boxes, void *grad_points, void *stream);

/* batched box3dr_iou (rotated=1) / box3d_iou (rotated=0)
 * (reference d3d/dgal_wrap.h:45-91; pair loop d3d/tracking/matcher.pyx:57-80).
 * boxes[.,7] f32 = (x,y,z,lx,ly,lz,rz); out[n,m] f32. */
size_t d3d_iou3d_workspace_bytes(int64_t n, int64_t m);
int d3d_iou3d_forward(const float *boxes1, int64_t n, const float *boxes2, int64_t m,
                      int32_t rotated, float *out, void *workspace, size_t workspace_bytes, void *stream);

/* replaces the pair loops of BaseMatcher.prepare_boxes (reference d3d/tracking/matcher.pyx:46-80) for the metrics IoU
 * (rotated = 0: box3d_iou) and RIoU (rotated = 1: box3dr_iou): src[n,9], dst[m,9] f32 rows = (label, score, x, y, z, lx, ly,
 * lz, yaw) as Target3DArray.to_numpy lays them out (abstraction.pyx:263-272); the dimensions are clipped to +-1e3
 * (matcher.pyx:49-51); cache[n,m] f32 = 1 - iou.  Workspace: d3d_iou3d_workspace_bytes(n, m) (optional). */
int d3d_match_distance(const float *src, int64_t n, const float *dst, int64_t m, int32_t rotated, float *cache,
                       void *workspace, size_t workspace_bytes, void *stream);

/* replaces ScoreMatcher.match + match_by_order (reference d3d/tracking/matcher.pyx:83-162) as the detection evaluator
 * drives them once per score threshold (d3d/benchmarks.pyx:218-238).  dist[n,m] f32 (d3d_match_distance); src_tag[n],
 * dst_tag[m] i32 categories (negative = takes no part); dst_threshold[m] f32 = distance threshold of dst j's category;
 * order[n] i64 = src rows from the best score down.  Every src row, in that order, takes the nearest unassigned dst of its
 * own category with dist <= threshold (ties: lower index).  src_match[n], dst_match[m] i32 = partner or -1.  Because a
 * row's choice depends only on the rows before it, the matching restricted to the first k rows of `order` IS the matching of
 * the score threshold that selects them: one call serves all thresholds.  Any threshold is exact: a row with more than 64
 * dst within its threshold lists its 64 nearest and, if those are all taken when its turn comes, sweeps its whole row for
 * the nearest free one.  status (device i32): bit 0 = some row had more than 64 (informational). */
size_t d3d_score_match_workspace_bytes(int64_t n, int64_t m);
int d3d_score_match(const float *dist, int64_t n, int64_t m, const int32_t *src_tag, const int32_t *dst_tag,
                    const float *dst_threshold, const int64_t *order, int32_t *src_match, int32_t *dst_match,
                    int32_t *status, void *workspace, size_t workspace_bytes, void *stream);

/* `batches` associations in one call (the evaluator's score thresholds on a frame, benchmarks.pyx:218-238: 40 short chains of
 * launches when issued one by one).  The rows of all problems are stacked: dist[n_total, m], src_tag[n_total], order[n_total],
 * src_match[n_total]; problem b owns rows row_off[b] .. row_off[b + 1] (row_off[batches + 1] i64, device) and its order /
 * src_match / dst_match entries are indices LOCAL to those rows; the destinations (m, dst_tag, dst_threshold) are common,
 * dst_match[batches, m].  Same result per problem as d3d_score_match.
 * Optional indirection (NULL = none): stacked row r takes its distances from dist row row_src[r] and, with mask (u8 [., m], 0 = the
 * pair takes no part), its acceptable pairs from mask row row_mask[r] -- the literal association of matcher.pyx:155-158 (the k-th
 * best source walks the k-th subset row's distances) without materialising a stacked matrix. */
size_t d3d_score_match_batched_workspace_bytes(int64_t n_total, int64_t m, int64_t batches);
int d3d_score_match_batched(const float *dist, const int64_t *row_src, const uint8_t *mask, const int64_t *row_mask,
                            const int64_t *row_off, int64_t batches, int64_t n_total, int64_t m,
                            const int32_t *src_tag, const int32_t *dst_tag, const float *dst_threshold, const int64_t *order,
                            int32_t *src_match, int32_t *dst_match, int32_t *status, void *workspace, size_t workspace_bytes,
                            void *stream);

/* replaces crop_2dr (reference d3d/box/utils.cpp:9-47, bound at box/impl.cpp as crop_2dr; Python box2dr_crop /
 * box3dp_crop, box/__init__.py:278-315): points[n,2], boxes[m,5] in `dtype`; out[m,n] u8 (0/1),
 * out[i,j] = point j lies in rotated box i (boundary inclusive). */
int d3d_crop_2dr(const void *points, int64_t n, const void *boxes, int64_t m, int32_t dtype, uint8_t *out, void *stream);

/* box3dp_crop for project_axis = 2 in one launch (reference d3d/box/__init__.py:289-315: crop_2dr on gathered columns, then
 * the interval test (p - d / 2 < b) & (b < p + d / 2) as [M,N] tensor operations, :311-313): points[n, point_stride >= 3] f32,
 * boxes[m, box_stride >= 7] f32 rows (x, y, z, lx, ly, lz, rz); out[m,n] u8 (0/1), the same bits as the composition.
 * D3D_ERR_UNSUPPORTED, nothing touched: another axis, m > 4096 or n < 4096 -- compose it from d3d_crop_2dr. */
int d3d_crop_3dp(const float *points, int64_t n, int32_t point_stride, const float *boxes, int64_t m, int32_t box_stride,
                 int32_t project_axis, uint8_t *out, void *stream);

/* stable descending argsort (the role torch::argsort plays inside the reference's nms2d,
 * nms.cpp:103): keys[n] in `dtype` -> order[n] i64; ties keep ascending index.  The order is torch's:
 * by value (-0 == +0), NaN before every number.  8 k .. 128 k keys: a 4-launch sample sort; other sizes: rocPRIM. */
size_t d3d_argsort_desc_workspace_bytes(int64_t n, int32_t dtype);
int d3d_argsort_desc(const void *keys, int64_t n, int32_t dtype, int64_t *order,
                     void *workspace, size_t workspace_bytes, void *stream);

size_t d3d_nms2d_workspace_bytes(int64_t n);

/* replaces nms2d / nms2d_cuda (reference d3d/box/nms.h:6-18, nms.cpp:10-119,
 * nms_cuda.cu:17-244).  Follows the CPU control flow (nms.cpp:23-59).
 *   boxes[n,5], scores[n] in `dtype`; order[n] i64 = descending argsort of scores (nms.cpp:103), or NULL: the library
 *   sorts the scores itself, as nms2d does (d3d_argsort_desc's order; inside the first kernel for up to 4096 boxes);
 *   suppressed[n] u8 (0/1) output.
 *   HARD: parallel (broad phase + exact IoU + fixed point, see box.hip); sets of up to 4096 boxes (a detector's top-k) take
 *   a five-launch path of their own unless a flag below names a general one.  LINEAR / GAUSSIAN (soft-NMS, nms.cpp:60-94)
 *   are sequential by construction -- every kept box rescales the later boxes it overlaps and the order is re-established
 *   after each -- and run in one workgroup that follows the reference's control flow.
 *   Other IoU types return D3D_ERR_UNSUPPORTED ("Unsupported iou type!", reference common.h:25).
 *   flags (per call, 0 = automatic): D3D_NMS_BROAD_SWEEP = sweep-and-prune broad phase instead of the uniform grid;
 *   D3D_NMS_FORCE_DENSE = the reference's all-pairs bit matrix (nms_cuda.cu layout) instead of candidate lists;
 *   D3D_NMS_SOFT_NO_LDS = soft-NMS state in global scratch; D3D_NMS_CAND_CAP(k) = use only k entries of the candidate
 *   list (tests of the overflow -> dense hand-over); D3D_NMS_GENERAL = the general path (uniform grid) also for sets of up to
 *   4096 boxes; D3D_NMS_TEST_WITHHOLD = test hook: the first workgroup of a one-launch scan (the grid's cell scan; with
 *   D3D_NMS_BROAD_SWEEP the scan of the incoming-list sizes) withholds its total, so the scan gives up after ~0.1 s and hands
 *   the call to the dense path; D3D_NMS_FORCE_LEVELS = the uniform grid's level kernels (roots of the greedy result decided
 *   before any pair is listed; automatic on dense grids: clusters of detections) on any input, implies the general path;
 *   D3D_NMS_ONE_LEVEL = one level of them instead of two (clusters of a few boxes).
 *   All give the same mask.
 *   soft-NMS has no size limit of its own (100 k boxes with most of them alive: seconds).
 *   dtype: D3D_F32 / D3D_F64 (boxes and scores), or D3D_F32_WIDE = f32 in memory, f64 arithmetic (what box2d_nms(precise=True)
 *   computes for fp32 tensors, box/__init__.py:254-255, without the copies). */
enum { D3D_NMS_BROAD_SWEEP = 1, D3D_NMS_FORCE_DENSE = 2, D3D_NMS_SOFT_NO_LDS = 4, D3D_NMS_GENERAL = 8, D3D_NMS_TEST_WITHHOLD = 16,
       D3D_NMS_FORCE_LEVELS = 32, D3D_NMS_ONE_LEVEL = 64,
       D3D_NMS_KEEP_MASK = 128   /* `suppressed` receives the KEEP mask -- what box2d_nms returns, ~suppressed (reference
                                    box/__init__.py:272) -- written by the kernels that decide it instead of a pass over the mask */ };
#define D3D_NMS_CAND_CAP(k) ((uint32_t)(k) << 8)
int d3d_nms2d(const void *boxes, const void *scores, const int64_t *order, int64_t n,
              int32_t iou_type, int32_t suppression_type, int32_t dtype,
              float iou_threshold, float score_threshold, float suppression_param,
              uint8_t *suppressed, void *workspace, size_t workspace_bytes, void *stream, uint32_t flags);

/* d3d_nms2d with a host-mapped word of the CALLER's (hipHostMalloc / pinned, int32, one per thread of callers): hard NMS on
 * the general path learns from it, half-way through ITS OWN launches, whether this call's boxes form clusters (a dense
 * broad-phase grid) -- only then are the ten launches of the level kernels enqueued; a scattered set skips them.  The call
 * waits for that word (the GPU keeps working: the launch that writes it has 18 us of work of its own), so it returns with
 * the second half of its kernels still in flight, like d3d_nms2d.  Same mask either way.  NULL, a stream under capture, or
 * D3D_NMS_FORCE_LEVELS: as d3d_nms2d, which always enqueues the level kernels.  Nothing is remembered between calls. */
int d3d_nms2d_notify(const void *boxes, const void *scores, const int64_t *order, int64_t n,
                     int32_t iou_type, int32_t suppression_type, int32_t dtype,
                     float iou_threshold, float score_threshold, float suppression_param,
                     uint8_t *suppressed, void *workspace, size_t workspace_bytes, void *stream, uint32_t flags,
                     int32_t *host_word);

/* Which route the LAST hard-NMS call that used `workspace` took (waits for `stream`).  Bits of *status:
 *   D3D_NMS_STATUS_DENSE_PATH    the n x n mask + sequential sweep decided the result (candidate list overflowed, a box covered
 *                                more than 1024 grid cells, the fixed point did not settle, D3D_NMS_FORCE_DENSE, or a scan gave up):
 *                                same mask, but seconds instead of microseconds at 100 k boxes;
 *   D3D_NMS_STATUS_SCAN_GAVE_UP  a one-launch scan stopped waiting for an earlier workgroup's total (~0.1 s of polling: a
 *                                preempted queue, a profiler holding workgroups back) and handed over to the dense path -- the
 *                                result is still exact; a caller that sees this on a shared GPU may simply call again.
 * The flags sit in the first bytes of the workspace: ask before anything else uses it.
 * suppression_type != HARD: *status = 0 (soft-NMS has a single route).  The reference has no counterpart (nms.cpp is one loop). */
enum { D3D_NMS_STATUS_DENSE_PATH = 1, D3D_NMS_STATUS_SCAN_GAVE_UP = 2 };
int d3d_nms2d_status(const void *workspace, int32_t suppression_type, void *stream, uint32_t *status);

#ifdef __cplusplus
}
#endif
#endif /* D3D_HIP_H */
