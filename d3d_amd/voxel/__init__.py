"""d3d_amd.voxel -- drop-in for d3d.voxel (reference d3d/voxel/__init__.py) on MI355X.

Public names mirror the reference: `VoxelGenerator` (voxel/__init__.py:12) and the three
functions + three enums of its compiled module `voxel_impl` (voxel/impl.cpp:3-21).  All
compute happens in hand-written HIP kernels behind the C ABI of include/d3d_hip.h; torch is
only used for device buffers and the current stream.  Tensors may live on the CPU (they are
staged through the current HIP device and results come back on the CPU, like the reference's
CPU-only operator) or on the GPU (results stay there).
"""
import ctypes
import enum
import threading

import numpy as np
import torch

from .. import _lib, options
from ..utils import Dict


class ReductionType(enum.IntEnum):          # voxelize.h:5
    NONE = 0
    MEAN = 1
    MAX = 2
    MIN = 3


class MaxPointsFilterType(enum.IntEnum):    # voxelize.h:6
    NONE = 0
    TRIM = 1
    FARTHEST_SAMPLING = 2


class MaxVoxelsFilterType(enum.IntEnum):    # voxelize.h:7
    NONE = 0
    TRIM = 1
    DESCENDING = 2


def _as_tensor(x):
    return torch.from_numpy(x) if isinstance(x, np.ndarray) else x


def _host_array(t, ctype, count):
    if isinstance(t, ctypes.Array):          # already marshalled (VoxelGenerator caches its grid description)
        return t
    vals = t.tolist() if isinstance(t, torch.Tensor) else list(t)
    return (ctype * count)(*vals[:count])


class _device_ctx:
    """torch.cuda.device(dev) only when dev is not already current (the context switch costs host time)"""

    def __init__(self, dev):
        self._ctx = None if dev.index == torch.cuda.current_device() else torch.cuda.device(dev)

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()

    def __exit__(self, *a):
        if self._ctx is not None:
            self._ctx.__exit__(*a)


def _stage(points, what="points"):
    points = _as_tensor(points)
    if not isinstance(points, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor or numpy array" % what)
    if points.dim() != 2:
        raise RuntimeError("%s must be a 2-D tensor" % what)       # reference: accessor<float,2> throws
    if points.dtype != torch.float32:
        raise RuntimeError("%s must be float32" % what)            # reference: accessor<float,2> throws
    dev = _lib.require_gpu() if not points.is_cuda else points.device
    return points.to(dev).contiguous(), points.device, dev


class _PackOverflow(Exception):
    """a voxel outgrew the packed hash slot's count field: the call is repeated with the general layout"""


class _NotFused(Exception):
    """the one-call sparse + filter entry does not take this combination here (DESCENDING off the binned index): two calls"""


class _CoordOverflow(Exception):
    """sparse contract: a finite voxel coordinate beyond the 3 x 21-bit key: the call is repeated with the 96-bit table"""


class _BinOverflow(Exception):
    """dense contract: a bucket of the binned index outgrew its workgroup: the call is repeated on the hash-table path"""


def _counts_to_host(counts, what):
    host = counts.cpu()     # the one host sync of a call: sizes of the variable-length outputs
    _check_status(int(host[_lib.COUNT_STATUS]), what)
    return host


def _check_status(status, what):
    if status & _lib.STATUS_BIN_OVERFLOW:
        raise _BinOverflow(what)
    if status & _lib.STATUS_PACK_OVERFLOW:
        raise _PackOverflow(what)
    if status & _lib.STATUS_TABLE_FULL:
        raise RuntimeError("%s: internal hash table overflow" % what)
    if status & _lib.STATUS_COORD_OVERFLOW:
        raise _CoordOverflow(what)


def _with_retry(run, flags=None):
    """run(flags) on the fast index first.  A bucket of the binned index that outgrew its workgroup's table -> once more on
    the hash table; a field of the packed one-word hash slot that overflowed (a voxel with > 2^cb points, or -- sparse
    contract -- a bounding box of voxel coordinates too large for the key field) -> once more with two-word slots."""
    flags = options.current().voxel_flags if flags is None else int(flags)     # per call, or the calling context's
    # every status names the option that lifts its limit; a call may meet several in a row (a frame with NaN points AND far
    # voxels on the hash path: packed slots -> plain slots -> wide keys), each at most once
    for _ in range(4):
        try:
            return run(flags)
        except _BinOverflow as e:
            add, err = _lib.VOXEL_PATH_HASH, e
        except _PackOverflow as e:
            add, err = _lib.VOXEL_PLAIN_SLOTS, e
        except _CoordOverflow as e:
            # a voxel coordinate beyond (-2^20, 2^20): the reference takes any int (voxelize.cpp:309) -- once more on the table
            # that compares all three 32-bit coordinates (D3D_VOXEL_WIDE_KEYS)
            add, err = _lib.VOXEL_WIDE_KEYS, e
        if flags & add:
            raise RuntimeError("%s: the index reported a limit that its own option should have lifted" % err)
        flags |= add
    raise RuntimeError("voxel index: no path took this frame")


_ws_bytes_cache = {}


def _workspace_bytes(lib, n):
    b = _ws_bytes_cache.get(n)
    if b is None:
        b = _ws_bytes_cache[n] = lib.d3d_voxelize_workspace_bytes(n, 0)
    return b


class DenseOutputBuffer:
    """A resident output for the dense contract (d3d_voxelize_3d_dense_resident; beyond the reference, which allocates per
    frame): voxels[capacity, max_points, C] kept on the device from frame to frame together with, per voxel id, the number of
    rows that may be non-zero.  A call then stores only the rows that hold points and zeros over what the previous frame's
    voxel of the same id held; the padding -- 95 % of the tensor on a LiDAR frame -- stays as it is.  The `voxels` a call
    returns is a view of this buffer: valid until the next call with it, and not to be written."""

    def __init__(self, capacity, max_points, device, columns=4):
        self.capacity, self.max_points, self.device, self.columns = int(capacity), int(max_points), torch.device(device), int(columns)
        self.voxels = torch.zeros((self.capacity, self.max_points, self.columns), dtype=torch.float32, device=self.device)
        self.row_state = torch.zeros((self.capacity,), dtype=torch.int16, device=self.device)       # (uint16 bits)


def voxelize_3d_dense(points, voxel_shape, voxel_bound, max_points, max_voxels, reduction_type, flags=None, poison=None,
                      resident=None):
    """voxelize_3d_dense of the reference (voxelize.h:9-12; voxelize.cpp:45-199).

    Returns dict(voxels[V,P,C] f32, coords[V,3] i64, voxel_pmask[V,P] bool,
    voxel_npoints[V] i32 [, aggregates[V,C] f32 when reduction != NONE]).

    flags: per-call index options (_lib.VOXEL_*; None = the calling context's, d3d_amd.options).  poison (test hook): fill
    every output buffer with NaN / 0xff patterns first, so that a row the kernels fail to write cannot hide behind fresh
    (zeroed) memory.  resident: a DenseOutputBuffer -- `voxels` comes back as a view of it (same values, see there; a frame the binned index does
    not take gets a fresh tensor instead); ValueError for a buffer that does not fit the call.
    """
    lib = _lib.load()
    pts, odev, dev = _stage(points)
    n, c = pts.shape
    if c < 3:
        raise RuntimeError("points need at least 3 columns (x, y, z)")
    red = int(reduction_type)
    max_points, max_voxels = int(max_points), int(max_voxels)
    shape_h = _host_array(voxel_shape, ctypes.c_int32, 3)
    bound_h = _host_array(voxel_bound, ctypes.c_float, 6)
    cap = max(min(n, max_voxels), 0)
    if resident is not None and (c != resident.columns or resident.max_points != max_points or resident.capacity < cap or
                                 resident.device != dev or max_points <= 0 or max_voxels <= 0):
        raise ValueError("resident output: needs the buffer's columns, max_points and device, capacity >= min(N, max_voxels)")
    with _device_ctx(dev):
        voxels = resident.voxels if resident is not None else torch.empty((cap, max_points, c), dtype=torch.float32, device=dev)
        coords = torch.empty((cap, 3), dtype=torch.int64, device=dev)
        pmask = torch.empty((cap, max_points), dtype=torch.uint8, device=dev)
        npts = torch.empty((cap,), dtype=torch.int32, device=dev)
        agg = torch.empty((cap, c), dtype=torch.float32, device=dev) if red != 0 else None
        counts = torch.empty((_lib.NUM_COUNTS,), dtype=torch.int64, device=dev)
        ws = _lib.workspace(_workspace_bytes(lib, n), dev)
        note = _lib.NotifyBuffer.get()
        if options.current().poison if poison is None else poison:
            if resident is None:
                voxels.fill_(float("nan"))
            coords.fill_(-(1 << 62)); pmask.fill_(0xff); npts.fill_(-7)
            if agg is not None:
                agg.fill_(float("nan"))

        def run_resident(fl):
            note.arm()
            rc = lib.d3d_voxelize_3d_dense_resident(
                _lib.ptr(pts), n, c, ctypes.cast(shape_h, ctypes.c_void_p), ctypes.cast(bound_h, ctypes.c_void_p),
                max_points, max_voxels, red, _lib.ptr(voxels), _lib.ptr(resident.row_state), _lib.ptr(coords), _lib.ptr(pmask),
                _lib.ptr(npts), _lib.ptr(agg), _lib.ptr(counts), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(), note.ptr, fl)
            if rc == _lib.ERR_UNSUPPORTED:
                return None                 # nothing was touched: the plain call decides (and reports a bad reduction type)
            _lib.check(rc, "voxelize_3d_dense_resident")
            host = note.wait(counts)
            _check_status(int(host[_lib.COUNT_STATUS]), "voxelize_3d_dense_resident")
            return int(host[_lib.COUNT_VOXELS])

        def run(fl):
            nonlocal voxels
            if resident is not None:
                if not (fl & _lib.VOXEL_PATH_HASH):
                    nv_res = run_resident(fl)
                    if nv_res is not None:
                        return nv_res
                # (a frame the binned index does not take -- above 8 M points, or a bucket overflowed and the call is repeated
                # on the hash path: those paths write whole tensors, a fresh one for this frame; the buffer's invariant is
                # untouched and the next frame uses it again)
                voxels = torch.empty((cap, max_points, c), dtype=torch.float32, device=dev)
            # the voxel count reaches the host through pinned memory as soon as it is final, while the GPU is still
            # writing voxels[V,P,C]: the call returns views of outputs in flight on the current stream, like any torch op
            note.arm()
            rc = lib.d3d_voxelize_3d_dense_notify(
                _lib.ptr(pts), n, c, ctypes.cast(shape_h, ctypes.c_void_p), ctypes.cast(bound_h, ctypes.c_void_p),
                max_points, max_voxels, red, _lib.ptr(voxels), _lib.ptr(coords), _lib.ptr(pmask), _lib.ptr(npts),
                _lib.ptr(agg), _lib.ptr(counts), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(), note.ptr, fl)
            if rc == _lib.ERR_UNSUPPORTED:
                raise ValueError("Unsupported reduction type in voxelization!")   # voxelize.cpp:196
            _lib.check(rc, "voxelize_3d_dense")
            host = note.wait(counts)
            _check_status(int(host[_lib.COUNT_STATUS]), "voxelize_3d_dense")
            return int(host[_lib.COUNT_VOXELS])
        nv = _with_retry(run, flags)
    ret = dict(voxels=voxels[:nv], coords=coords[:nv], voxel_pmask=pmask[:nv].view(torch.bool),
               voxel_npoints=npts[:nv])
    if red != 0:
        ret["aggregates"] = agg[:nv]
    ret = _lib.to_caller(ret, odev, dev)
    return ret


def voxelize_3d_sparse(points, voxel_size, ndim=3, flags=None):
    """voxelize_sparse of the reference, exported as voxelize_3d_sparse
    (voxelize.h:14-17; voxelize.cpp:288-335; impl.cpp:5)."""
    lib = _lib.load()
    if ndim not in (None, 3):
        raise ValueError("only ndim=3 is supported")   # the reference's map key is a 3-tuple (voxelize.cpp:16)
    pts, odev, dev = _stage(points)
    n, c = pts.shape
    if c < 3:
        raise RuntimeError("points need at least 3 columns (x, y, z)")
    size_h = _host_array(voxel_size, ctypes.c_float, 3)
    with torch.cuda.device(dev):
        mapping = torch.empty((n,), dtype=torch.int64, device=dev)
        coords = torch.empty((n, 3), dtype=torch.int64, device=dev)
        npts = torch.empty((n,), dtype=torch.int32, device=dev)
        counts = torch.empty((_lib.NUM_COUNTS,), dtype=torch.int64, device=dev)
        ws = _lib.workspace(lib.d3d_voxelize_workspace_bytes(n, 0), dev)

        def run(fl):
            rc = lib.d3d_voxelize_3d_sparse(_lib.ptr(pts), n, c, ctypes.cast(size_h, ctypes.c_void_p), _lib.ptr(mapping),
                                            _lib.ptr(coords), _lib.ptr(npts), _lib.ptr(counts), _lib.ptr(ws), ws.numel(),
                                            _lib.stream_ptr(), fl)
            _lib.check(rc, "voxelize_3d_sparse")
            return int(_counts_to_host(counts, "voxelize_3d_sparse")[_lib.COUNT_VOXELS])
        nv = _with_retry(run, flags)
    ret = dict(points_mapping=mapping, coords=coords[:nv], voxel_npoints=npts[:nv])
    ret = _lib.to_caller(ret, odev, dev)
    return ret


def voxelize_3d_filter(feats, points_mapping, coords, voxel_npoints, coords_bound,
                       min_points=None, max_points=None, max_voxels=None,
                       max_points_filter=None, max_voxels_filter=None):
    """voxelize_filter of the reference, exported as voxelize_3d_filter
    (voxelize.h:19-25; voxelize.cpp:337-484).  `voxel_npoints` must be the per-voxel counts of
    `points_mapping` (as produced by voxelize_3d_sparse).  `coords_bound` is required
    (None crashes the reference, voxelize.cpp:348-349)."""
    lib = _lib.load()
    pf = int(max_points_filter) if max_points_filter is not None else 0
    vf = int(max_voxels_filter) if max_voxels_filter is not None else 0
    if vf != 0 and max_voxels is None:
        raise ValueError("Must specify maximum voxel count to filter voxels!")            # voxelize.cpp:359
    if pf != 0 and max_points is None:
        raise ValueError("Must specify maximum points per voxel to filter points!")       # voxelize.cpp:362
    if pf == MaxPointsFilterType.FARTHEST_SAMPLING:
        raise ValueError("Farthest Sampling not implemented!")                            # voxelize.cpp:470
    if coords_bound is None:
        raise ValueError("coords_bound is required")
    fts, odev, dev = _stage(feats, "feats")
    n, c = fts.shape
    mapping = _as_tensor(points_mapping).to(dev, torch.int64).contiguous()
    crd = _as_tensor(coords).to(dev, torch.int64).contiguous()
    cnt = _as_tensor(voxel_npoints).to(dev, torch.int32).contiguous()
    if mapping.numel() != n:
        raise RuntimeError("points_mapping must have one entry per point")
    nvox = crd.shape[0]
    if crd.dim() != 2 or crd.shape[1] != 3 or cnt.numel() != nvox:
        raise RuntimeError("coords must be [V,3] and voxel_npoints [V]")
    cb = _as_tensor(coords_bound).reshape(-1).tolist()
    bound_h = (ctypes.c_int64 * 6)(*[int(x) for x in cb])
    with torch.cuda.device(dev):
        o_feats = torch.empty((n, c), dtype=torch.float32, device=dev)
        o_mask = torch.empty((n,), dtype=torch.int64, device=dev)
        o_map = torch.empty((n,), dtype=torch.int64, device=dev)
        o_cnt = torch.empty((nvox,), dtype=torch.int32, device=dev)
        o_crd = torch.empty((nvox, 3), dtype=torch.int64, device=dev)
        counts = torch.empty((_lib.NUM_COUNTS,), dtype=torch.int64, device=dev)
        ws = _lib.workspace(lib.d3d_voxelize_workspace_bytes(n, nvox), dev)
        rc = lib.d3d_voxelize_3d_filter(
            _lib.ptr(fts), n, c, _lib.ptr(mapping), _lib.ptr(crd), _lib.ptr(cnt), nvox,
            ctypes.cast(bound_h, ctypes.c_void_p), int(min_points or 0), int(max_points or 0), int(max_voxels or 0),
            pf, vf, _lib.ptr(o_feats), _lib.ptr(o_mask), _lib.ptr(o_map), _lib.ptr(o_cnt), _lib.ptr(o_crd),
            _lib.ptr(counts), _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "voxelize_3d_filter")
        host = _counts_to_host(counts, "voxelize_3d_filter")
        k, v = int(host[_lib.COUNT_POINTS]), int(host[_lib.COUNT_VOXELS])
    ret = dict(points=o_feats[:k], points_mask=o_mask[:k], points_mapping=o_map[:k],
               voxel_npoints=o_cnt[:v], coords=o_crd[:v])
    ret = _lib.to_caller(ret, odev, dev)
    return ret


# Output buffers of the NEXT sparse + filter call, allocated while the current one waits for its output sizes: the GPU-side
# work behind the size notification (the compaction, ~20 us) is shorter than the host's per-call work (six allocations, the
# launches), so the next frame's first kernel used to wait for the host.  One set per thread, handed over to the caller whole
# (the cache keeps no reference to buffers it has given out) and at most ONE call old: a set the next call does not take is
# dropped there.  Keyed by CAPACITY: real frames differ in size from call to call, so the set is sized for the
# frame's point count rounded up by at most 12.5 % and serves any frame of that size class (the outputs are views of its first n rows).
_spare = threading.local()
_SPARE_FLOOR = 1 << 12


def _spare_cap(n):
    """n rounded up in steps of 1/8 of its power of two (at least 4 k rows): the views handed out pin at most 12.5 % more than the
    frame needs -- a fixed 64 k step pinned 30 x the memory of a 2 k-point crop for as long as the caller kept one output"""
    n = max(int(n), 1)
    step = max(_SPARE_FLOOR, 1 << max(n.bit_length() - 4, 0))
    return -(-n // step) * step


def _spare_take(key, n, make):
    """-> (capacity, buffers) for n points"""
    got = getattr(_spare, "set", None)
    _spare.set = None
    cap = _spare_cap(n)
    if got is not None and got[0] == key and n <= got[1] <= 2 * cap:
        return got[1], got[2]
    return cap, make(cap)


def _spare_put(key, n, make):
    cap = _spare_cap(n)
    _spare.set = (key, cap, make(cap))


def release_cached_buffers():
    """drop the spare output set this thread holds for its next sparse VoxelGenerator call (~100 bytes per point)"""
    _spare.set = None


class _SparseFilterCall(ctypes.Structure):
    """include/d3d_hip.h: D3DSparseFilterCall"""
    _fields_ = [("points", ctypes.c_void_p), ("n", ctypes.c_int64), ("c", ctypes.c_int32), ("min_points", ctypes.c_int32),
                ("max_points", ctypes.c_int32), ("max_voxels", ctypes.c_int32), ("max_points_filter", ctypes.c_int32),
                ("max_voxels_filter", ctypes.c_int32), ("voxel_size", ctypes.c_float * 3), ("flags", ctypes.c_uint32),
                ("coords_bound", ctypes.c_int64 * 6), ("coord_offset", ctypes.c_int64 * 3), ("has_coord_offset", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("outputs", ctypes.c_void_p), ("outputs_bytes", ctypes.c_size_t),
                ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t), ("stream", ctypes.c_void_p),
                ("host_counts", ctypes.c_void_p)]


class _LazyCounts:
    """the two device count rows at the front of the workspace, materialised only if NotifyBuffer.wait falls back to reading them"""

    def __init__(self, ws):
        self.ws = ws

    def dim(self):
        return 2

    def cpu(self):
        return self.ws[:2 * _lib.NUM_COUNTS * 8].view(torch.int64).view(2, _lib.NUM_COUNTS).cpu()


class _SparsePlan:
    """what VoxelGenerator keeps for its sparse + filter calls (round 6): the argument block of d3d_voxelize_3d_sparse_filter_call,
    filled once, and the buffer layouts per frame size.  Per frame the host sets three pointers and a size, makes ONE allocation
    (the outputs, from the spare set) and one ctypes call with one argument -- 26 arguments, six allocations and twelve data_ptr()
    calls before (14 us of every 68 were host)."""

    def __init__(self, size_h, bound_h, min_points, max_points, max_voxels, pf, vf, offset_h):
        a = self.args = _SparseFilterCall()
        a.min_points, a.max_points, a.max_voxels = int(min_points or 0), int(max_points or 0), int(max_voxels or 0)
        a.max_points_filter, a.max_voxels_filter = int(pf), int(vf)
        for k in range(3):
            a.voxel_size[k] = size_h[k]
        for k in range(6):
            a.coords_bound[k] = bound_h[k]
        a.has_coord_offset = 0 if offset_h is None else 1
        if offset_h is not None:
            for k in range(3):
                a.coord_offset[k] = offset_h[k]
        self.ref = ctypes.byref(a)
        self.layouts = {}

    def layout(self, lib, n, c):
        got = self.layouts.get((n, c))
        if got is None:
            if len(self.layouts) > 64:
                self.layouts.clear()
            off = (ctypes.c_size_t * 5)()
            nbytes = int(lib.d3d_voxelize_3d_sparse_filter_call_layout(n, c, off))
            got = self.layouts[(n, c)] = (nbytes, [int(x) for x in off], int(lib.d3d_voxelize_3d_sparse_filter_call_workspace_bytes(n)))
        return got


def _sparse_filter_planned(plan, points, flags=None):
    """VoxelGenerator.__call__'s sparse branch through the prepared argument block -> dict, or raises _NotFused"""
    lib = _lib.load()
    pts, odev, dev = _stage(points)
    n, c = pts.shape
    if c < 3:
        raise RuntimeError("points need at least 3 columns (x, y, z)")
    a = plan.args
    with _device_ctx(dev):
        nbytes, off, ws_bytes = plan.layout(lib, n, c)
        stream = _lib.stream_raw()
        key = (dev, c, stream, "planned")

        def make(m):
            return torch.empty((plan.layout(lib, m, c)[0],), dtype=torch.uint8, device=dev)
        _, buf = _spare_take(key, n, make)
        ws = _lib.workspace(ws_bytes, dev)
        note = _lib.NotifyBuffer.get()
        a.points, a.n, a.c = pts.data_ptr(), n, c
        a.outputs, a.outputs_bytes = buf.data_ptr(), buf.numel()
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        a.stream, a.host_counts = stream, note.ptr.value
        put = []

        def run(fl):
            note.arm()
            a.flags = fl
            rc = lib.d3d_voxelize_3d_sparse_filter_call(plan.ref)
            if rc == _lib.ERR_UNSUPPORTED and a.max_voxels_filter == MaxVoxelsFilterType.DESCENDING:
                raise _NotFused()
            _lib.check(rc, "voxelize_3d_sparse + voxelize_3d_filter")
            if not put:                   # the next call's buffer, while this one's sizes are on their way
                _spare_put(key, n, make)
                put.append(True)
            host = note.wait(_LazyCounts(ws))
            _check_status(int(host[_lib.COUNT_STATUS]), "voxelize_3d_sparse")
            return int(host[_lib.NUM_COUNTS + _lib.COUNT_POINTS]), int(host[_lib.NUM_COUNTS + _lib.COUNT_VOXELS])
        k, v = _with_retry(run, flags)
        ret = dict(points=buf[off[0]:off[0] + k * c * 4].view(torch.float32).view(k, c),
                   points_mask=buf[off[1]:off[1] + k * 8].view(torch.int64),
                   points_mapping=buf[off[2]:off[2] + k * 8].view(torch.int64),
                   voxel_npoints=buf[off[3]:off[3] + v * 4].view(torch.int32),
                   coords=buf[off[4]:off[4] + v * 24].view(torch.int64).view(v, 3))
    ret = _lib.to_caller(ret, odev, dev)
    return ret


def _lookup(enum_cls, name, message):
    key = (name or "NONE").upper()
    if key not in enum_cls.__members__:
        raise ValueError(message)
    return enum_cls[key]


class VoxelGenerator:
    """Convert point cloud to voxels -- same constructor, call signature and results as the
    reference's d3d.voxel.VoxelGenerator (voxel/__init__.py:12-104)."""

    def __init__(self, bounds, shape, min_points=0, max_points=30, max_voxels=20000,
                 max_points_filter=None, max_voxels_filter=None, reduction=None, dense=False, resident=False):
        # resident (beyond the reference; dense only): `voxels` is returned as a view of a buffer the generator keeps on the
        # device (DenseOutputBuffer) -- same values, valid until the generator's next call.
        # The derived grid quantities are tiny fp32 host tensors computed with the same torch
        # ops as the reference so that `size` is bit-identical (it feeds the coordinate division).
        self._bounds = torch.tensor(bounds, dtype=torch.float)
        self._shape = torch.tensor(shape, dtype=torch.int32)
        self._min_points, self._max_points, self._max_voxels, self._dense = min_points, max_points, max_voxels, dense
        if resident and not dense:
            raise ValueError("resident output is for dense voxelization")
        self._resident, self._resident_buf = bool(resident), None
        self._sparse_plan = None             # the prepared argument block of the sparse + filter call (_SparsePlan), made on first use

        lohi = self._bounds.reshape(3, 2)
        self._size = (lohi[:, 1] - lohi[:, 0]) / self._shape                                  # :41
        origin_cells = lohi[:, 0] / self._size                                                # :42
        if torch.any(torch.abs(torch.round(origin_cells) - origin_cells) > 1e-3):            # :43-44
            raise ValueError("The voxelization grids is not aligned with the origin, "
                             "which could lead to unexpected behavior!")
        self._offset = torch.round(origin_cells).int()                                        # :45
        self._vbounds = torch.round(lohi / self._size.reshape(3, 1)).long()                   # :46
        self._shape_h = _host_array(self._shape, ctypes.c_int32, 3)      # marshalled once for the C ABI
        self._offset_dev = {}
        self._bounds_h = _host_array(self._bounds, ctypes.c_float, 6)
        self._size_h = _host_array(self._size, ctypes.c_float, 3)
        self._offset_h = (ctypes.c_int64 * 3)(*[int(v) for v in self._offset.tolist()])
        self._vbounds_h = (ctypes.c_int64 * 6)(*[int(v) for v in self._vbounds.reshape(-1).tolist()])

        red = (reduction or "NONE").upper()
        if red != "NONE" and not dense:
            raise ValueError("Reduction is only for dense voxelization!")                     # :50-51
        self._reduction = _lookup(ReductionType, red, "Unsupported reduction type in VoxelGenerator!")
        self._max_points_filter = _lookup(MaxPointsFilterType, max_points_filter,
                                          "Unsupported maximum points filter in VoxelGenerator!")
        self._max_voxels_filter = _lookup(MaxVoxelsFilterType, max_voxels_filter,
                                          "Unsupported maximum voxels filter in VoxelGenerator!")
        if dense:                                                                             # :71-77
            if min_points > 0:
                raise NotImplementedError("Minimum points filtering is not implemented for dense")
            if self._max_points_filter not in (MaxPointsFilterType.NONE, MaxPointsFilterType.TRIM):
                raise NotImplementedError("Only trim is implemented for max points filtering")
            if self._max_voxels_filter not in (MaxVoxelsFilterType.NONE, MaxVoxelsFilterType.TRIM):
                raise NotImplementedError("Only trim is implemented for max voxels filtering")

    def stream(self, frames, flags=None, poison=None, pipelined=False):
        """generator over `frames` (an iterable of point clouds): the results of self(frame), frame by frame, in order.

        pipelined=True (beyond the reference, whose callers loop over frames, voxel/__init__.py:79; dense=True and [N,4]
        float32 device tensors): frame k + 1's index launches (latency-bound) go to a side stream, to run under frame k's
        output launch (bandwidth-bound) on another -- two frames in flight, each with its own scratch
        (d3d_voxelize_3d_dense_staged); results are ordinary stream-ordered tensors on the caller's current stream.
        MEASURED on MI355X / ROCm 7.2 (profiles/r04_pipelined_*): no gain -- the runtime maps both streams to one
        hardware queue (144 vs 137 us per frame for the plain loop), and on two queues every cross-queue event wait costs
        50-60 us (247 us per frame).  Hence off by default: the plain loop."""
        if not (pipelined and self._dense) or self._resident:       # (a resident output is ONE buffer: one frame in flight)
            return (self(f, flags=flags, poison=poison) for f in frames)
        return _stream_frames(self, frames, flags, poison)

    def __call__(self, points, flags=None, poison=None):
        """Returns a Dict: dense -> voxels, coords, voxel_pmask, voxel_npoints[, aggregates];
        sparse -> points, points_mask, points_mapping, voxel_npoints, coords (voxel/__init__.py:79-104).
        flags / poison: per-call options beyond the reference's signature (see voxelize_3d_dense)."""
        points = _as_tensor(points)
        odev = points.device
        if not points.is_cuda:
            points = points.to(_lib.require_gpu())    # stage once; results go back to the caller's device
        if self._dense:
            buf = None
            if self._resident and points.shape[0] > 0:
                need = min(int(points.shape[0]), int(self._max_voxels))
                buf = self._resident_buf
                if buf is None or buf.capacity < need or buf.device != points.device or buf.columns != int(points.shape[1]):
                    grow = need if buf is None else max(need, buf.capacity + buf.capacity // 4)
                    buf = self._resident_buf = DenseOutputBuffer(min(grow, int(self._max_voxels)), self._max_points, points.device,
                                                                 columns=int(points.shape[1]))
            ret = Dict(voxelize_3d_dense(points, self._shape_h, self._bounds_h, self._max_points,
                                         self._max_voxels, self._reduction, flags=flags, poison=poison, resident=buf))
        else:
            pf, vf = int(self._max_points_filter), int(self._max_voxels_filter)
            ret = None
            if points.shape[0] > 0:
                try:    # (the offset of :103 is subtracted inside the call, where the coords are written)
                    if self._sparse_plan is None:
                        self._check_sparse_args(pf, vf)
                        self._sparse_plan = _SparsePlan(self._size_h, self._vbounds_h, self._min_points, self._max_points,
                                                        self._max_voxels, pf, vf, self._offset_h)
                    ret = Dict(_sparse_filter_planned(self._sparse_plan, points, flags=flags))
                except _NotFused:
                    pass
            if ret is None:
                # (an empty cloud; DESCENDING off the binned index: its sort needs the voxel count on the host) the two calls
                # of the reference, two read-backs
                sparse = voxelize_3d_sparse(points, self._size_h, 3, flags=flags)
                ret = Dict(voxelize_3d_filter(points, sparse["points_mapping"], sparse["coords"],
                                              sparse["voxel_npoints"], self._vbounds, self._min_points,
                                              self._max_points, self._max_voxels, self._max_points_filter,
                                              self._max_voxels_filter))
                off = self._offset_dev.get(ret.coords.device)
                if off is None:
                    off = self._offset_dev[ret.coords.device] = self._offset.to(ret.coords.device)
                ret.coords = ret.coords - off                                                 # :103
        if odev != points.device:
            ret = Dict(_lib.to_caller(dict(ret), odev, points.device))
        return ret


def _vg_check_sparse_args(self, pf, vf):
    if vf != 0 and self._max_voxels is None:
        raise ValueError("Must specify maximum voxel count to filter voxels!")            # voxelize.cpp:359
    if pf != 0 and self._max_points is None:
        raise ValueError("Must specify maximum points per voxel to filter points!")       # voxelize.cpp:362
    if pf == MaxPointsFilterType.FARTHEST_SAMPLING:
        raise ValueError("Farthest Sampling not implemented!")                            # voxelize.cpp:470


VoxelGenerator._check_sparse_args = _vg_check_sparse_args


class _FrameSlot:
    """what one frame in flight of VoxelGenerator.stream() owns: scratch, the size notification, its two events"""

    def __init__(self):
        self.ws = None
        self.note = _lib.NotifyBuffer()
        self.idx_done = torch.cuda.Event()
        self.out_done = torch.cuda.Event()
        self.busy = False
        self.used = False


def _stream_frames(gen, frames, flags, poison):
    lib = _lib.load()
    fl = options.current().voxel_flags if flags is None else int(flags)
    poison = options.current().poison if poison is None else poison
    red = int(gen._reduction)
    P, mv = int(gen._max_points), int(gen._max_voxels)
    slots = [_FrameSlot(), _FrameSlot()]
    s_idx = s_out = None
    pending = None          # (slot, outputs, frame) of the frame whose output launch is in flight
    staged = True

    def finish(item):
        slot, out, pts = item
        voxels, coords, pmask, npts, agg, counts = out
        # the outputs (and `counts`) are written on the side stream: the caller's stream waits for them BEFORE anything reads
        # them there -- note.wait() polls host memory, but its fallback after 50 ms is a device read of `counts` on this stream
        torch.cuda.current_stream().wait_event(slot.out_done)
        host = slot.note.wait(counts)
        slot.busy = False
        if int(host[_lib.COUNT_STATUS]) != 0:                       # a capacity limit of the fast index: this frame again, alone
            torch.cuda.current_stream().synchronize()
            return gen(pts, flags=flags, poison=poison)
        nv = int(host[_lib.COUNT_VOXELS])
        ret = Dict(voxels=voxels[:nv], coords=coords[:nv], voxel_pmask=pmask[:nv].view(torch.bool), voxel_npoints=npts[:nv])
        if red != 0:
            ret["aggregates"] = agg[:nv]
        return ret

    k = 0
    try:
        for frame in frames:
            pts = _as_tensor(frame)
            ok = (staged and isinstance(pts, torch.Tensor) and pts.is_cuda and pts.dtype == torch.float32 and pts.dim() == 2 and
                  pts.shape[1] == 4 and pts.shape[0] > 0 and pts.is_contiguous() and P > 0)
            if not ok:
                if pending is not None:
                    yield finish(pending)
                    pending = None
                yield gen(frame, flags=flags, poison=poison)
                continue
            dev = pts.device
            n = pts.shape[0]
            cap = max(min(n, mv), 0)
            with _device_ctx(dev):
                if s_idx is None:
                    s_idx, s_out = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
                slot = slots[k & 1]
                k += 1
                if slot.busy:                                            # (cannot happen: a slot's frame is finished two frames on)
                    raise RuntimeError("frame slot still in flight")
                cur = torch.cuda.current_stream()
                voxels = torch.empty((cap, P, 4), dtype=torch.float32, device=dev)
                coords = torch.empty((cap, 3), dtype=torch.int64, device=dev)
                pmask = torch.empty((cap, P), dtype=torch.uint8, device=dev)
                npts = torch.empty((cap,), dtype=torch.int32, device=dev)
                agg = torch.empty((cap, 4), dtype=torch.float32, device=dev) if red != 0 else None
                counts = torch.empty((_lib.NUM_COUNTS,), dtype=torch.int64, device=dev)
                if poison:
                    voxels.fill_(float("nan")); coords.fill_(-(1 << 62)); pmask.fill_(0xff); npts.fill_(-7)
                    if agg is not None:
                        agg.fill_(float("nan"))
                need = _workspace_bytes(lib, n)
                if slot.ws is None or slot.ws.numel() < need:
                    slot.ws = torch.empty(int(need * 1.25), dtype=torch.uint8, device=dev)
                args = (_lib.ptr(pts), n, 4, ctypes.cast(gen._shape_h, ctypes.c_void_p), ctypes.cast(gen._bounds_h, ctypes.c_void_p),
                        P, mv, red, _lib.ptr(voxels), _lib.ptr(coords), _lib.ptr(pmask), _lib.ptr(npts), _lib.ptr(agg),
                        _lib.ptr(counts), _lib.ptr(slot.ws), slot.ws.numel())
                s_idx.wait_stream(cur)                                   # the frame (and the poison fills) are ready
                if slot.used:
                    s_idx.wait_event(slot.out_done)                      # this slot's scratch: its last frame's output has read it
                slot.used = True
                rc = lib.d3d_voxelize_3d_dense_staged(*args, ctypes.c_void_p(s_idx.cuda_stream), None, fl, 1)
                if rc == _lib.OK:
                    slot.idx_done.record(s_idx)
                if rc == _lib.ERR_UNSUPPORTED:                           # this frame (and the rest) the ordinary way
                    staged = False
                    if pending is not None:
                        yield finish(pending)
                        pending = None
                    yield gen(frame, flags=flags, poison=poison)
                    continue
                _lib.check(rc, "voxelize_3d_dense (index stage)")
                slot.note.arm()
                s_out.wait_event(slot.idx_done)
                rc = lib.d3d_voxelize_3d_dense_staged(*args, ctypes.c_void_p(s_out.cuda_stream), slot.note.ptr, fl, 2)
                slot.out_done.record(s_out)
                _lib.check(rc, "voxelize_3d_dense (output stage)")
                slot.busy = True
                # (the buffers belong to the caller's stream for the allocator: by the time a result is handed out that stream has
                # waited for the side streams, so whatever it frees or reuses later is ordered behind their work -- no
                # record_stream, which would keep every freed block out of the cache until an event has been polled: a hipMalloc
                # per frame, 314 us per frame measured)
            if pending is not None:
                yield finish(pending)
            pending = (slot, (voxels, coords, pmask, npts, agg, counts), pts)
        if pending is not None:
            yield finish(pending)
    finally:
        # also when the consumer stops early or a frame raises: the side streams may still be writing the pending frame's
        # outputs and the slots' scratch, which go back to the CALLER's stream's allocator pool once they are dropped
        if s_out is not None:
            torch.cuda.current_stream().wait_stream(s_out)
            torch.cuda.current_stream().wait_stream(s_idx)


__all__ = ["VoxelGenerator", "voxelize_3d_dense", "voxelize_3d_sparse", "voxelize_3d_filter", "release_cached_buffers",
           "ReductionType", "MaxPointsFilterType", "MaxVoxelsFilterType"]
