"""oracle -- CPU checker for the d3d voxel / box hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; nothing under d3d_amd/ does.  It wraps oracle/liboracle.so (plain C restatement
of the reference algorithms, built by oracle/Makefile) with numpy in/out and mirrors the
reference's Python operator layer:

* reference d3d/voxel/__init__.py:12-104  -> VoxelGenerator
* reference d3d/box/__init__.py:180-276   -> box2d_iou, box2d_nms
* reference d3d/dgal_wrap.h:45-91 + d3d/tracking/matcher.pyx:57-80 -> iou3d
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

REDUCTION = {"NONE": 0, "MEAN": 1, "MAX": 2, "MIN": 3}            # voxelize.h:5
MAX_POINTS_FILTER = {"NONE": 0, "TRIM": 1, "FARTHEST_SAMPLING": 2}  # voxelize.h:6
MAX_VOXELS_FILTER = {"NONE": 0, "TRIM": 1, "DESCENDING": 2}        # voxelize.h:7
IOU_TYPE = {"NA": 0, "BOX": 1, "RBOX": 2, "GBOX": 3, "GRBOX": 4, "DBOX": 5, "DRBOX": 6}  # box/common.h:5-9
SUPRESSION = {"HARD": 0, "LINEAR": 1, "GAUSSIAN": 2}               # box/common.h:10


def build():
    """Compile oracle/liboracle.so with gcc (idempotent)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return os.path.join(_HERE, "liboracle.so")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if os.environ.get("D3D_ORACLE_SANITIZED") == "1":       # tools/run_sanitized.sh: the ASan + UBSan build of the checker
            path = os.path.join(_HERE, "liboracle_asan.so")
            subprocess.check_call(["make", "-s", "-C", _HERE, "asan"])
        elif not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.oracle_voxelize_3d_dense.restype = ctypes.c_int64
        L.oracle_voxelize_3d_sparse.restype = ctypes.c_int64
        L.oracle_voxelize_3d_filter.restype = ctypes.c_int32
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class Dict(dict):
    """attribute-access dict (stands in for addict.Dict, voxel/__init__.py:1)"""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


# --------------------------------------------------------------------------- voxel
def voxelize_3d_dense(points, shape, bound, max_points, max_voxels, reduction):
    """voxelize.cpp:182-199.  reduction: int or name.  Returns Dict of numpy arrays."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    n, c = points.shape
    shape = np.ascontiguousarray(shape, dtype=np.int32)
    bound = np.ascontiguousarray(bound, dtype=np.float32)
    red = REDUCTION[reduction.upper()] if isinstance(reduction, str) else int(reduction)
    cap = int(max_voxels)
    voxels = np.empty((cap, max_points, c), np.float32)
    coords = np.empty((cap, 3), np.int64)
    pmask = np.empty((cap, max_points), np.uint8)
    npoints = np.empty((cap,), np.int32)
    agg = np.empty((cap, c), np.float32) if red != 0 else None
    nv = lib().oracle_voxelize_3d_dense(
        _p(points), ctypes.c_int64(n), ctypes.c_int32(c), _p(shape), _p(bound),
        ctypes.c_int32(max_points), ctypes.c_int32(max_voxels), ctypes.c_int32(red),
        _p(voxels), _p(coords), _p(pmask), _p(npoints), _p(agg))
    if nv < 0:
        raise MemoryError("oracle_voxelize_3d_dense")
    ret = Dict(voxels=voxels[:nv], coords=coords[:nv], voxel_pmask=pmask[:nv].astype(bool),
               voxel_npoints=npoints[:nv])
    if red != 0:
        ret["aggregates"] = agg[:nv]
    return ret


def voxelize_3d_sparse(points, voxel_size, ndim=3):
    """voxelize.cpp:288-335"""
    assert ndim in (None, 3)
    points = np.ascontiguousarray(points, dtype=np.float32)
    n, c = points.shape
    voxel_size = np.ascontiguousarray(voxel_size, dtype=np.float32)
    mapping = np.empty((n,), np.int64)
    coords = np.empty((max(n, 1), 3), np.int64)
    npoints = np.empty((max(n, 1),), np.int32)
    nv = lib().oracle_voxelize_3d_sparse(_p(points), ctypes.c_int64(n), ctypes.c_int32(c), _p(voxel_size),
                                         _p(mapping), _p(coords), _p(npoints))
    if nv < 0:
        raise MemoryError("oracle_voxelize_3d_sparse")
    return Dict(points_mapping=mapping, coords=coords[:nv].copy(), voxel_npoints=npoints[:nv].copy())


def voxelize_3d_filter(feats, points_mapping, coords, voxel_npoints, coords_bound,
                       min_points, max_points, max_voxels, max_points_filter, max_voxels_filter):
    """voxelize.cpp:337-484"""
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    n, c = feats.shape
    points_mapping = np.ascontiguousarray(points_mapping, dtype=np.int64)
    coords = np.ascontiguousarray(coords, dtype=np.int64)
    voxel_npoints = np.ascontiguousarray(voxel_npoints, dtype=np.int32)
    coords_bound = np.ascontiguousarray(coords_bound, dtype=np.int64).reshape(3, 2)
    nvox = coords.shape[0]
    pf = MAX_POINTS_FILTER[max_points_filter.upper()] if isinstance(max_points_filter, str) else int(max_points_filter or 0)
    vf = MAX_VOXELS_FILTER[max_voxels_filter.upper()] if isinstance(max_voxels_filter, str) else int(max_voxels_filter or 0)
    o_feats = np.empty((max(n, 1), c), np.float32)
    o_mask = np.empty((max(n, 1),), np.int64)
    o_map = np.empty((max(n, 1),), np.int64)
    o_np = np.empty((max(nvox, 1),), np.int32)
    o_coords = np.empty((max(nvox, 1), 3), np.int64)
    counts = np.zeros((2,), np.int64)
    rc = lib().oracle_voxelize_3d_filter(
        _p(feats), ctypes.c_int64(n), ctypes.c_int32(c), _p(points_mapping), _p(coords), _p(voxel_npoints),
        ctypes.c_int64(nvox), _p(coords_bound), ctypes.c_int32(min_points or 0), ctypes.c_int32(max_points),
        ctypes.c_int32(max_voxels), ctypes.c_int32(pf), ctypes.c_int32(vf),
        _p(o_feats), _p(o_mask), _p(o_map), _p(o_np), _p(o_coords), _p(counts))
    if rc == -2:
        raise ValueError("Farthest Sampling not implemented!")  # voxelize.cpp:470
    if rc != 0:
        raise RuntimeError("oracle_voxelize_3d_filter rc=%d" % rc)
    k, v = int(counts[0]), int(counts[1])
    return Dict(points=o_feats[:k].copy(), points_mask=o_mask[:k].copy(), points_mapping=o_map[:k].copy(),
                voxel_npoints=o_np[:v].copy(), coords=o_coords[:v].copy())


class VoxelGenerator:
    """numpy restatement of reference d3d/voxel/__init__.py:12-104 (float32 arithmetic)."""

    def __init__(self, bounds, shape, min_points=0, max_points=30, max_voxels=20000,
                 max_points_filter=None, max_voxels_filter=None, reduction=None, dense=False):
        self._bounds = np.asarray(bounds, dtype=np.float32)
        self._shape = np.asarray(shape, dtype=np.int32)
        self._min_points, self._max_points, self._max_voxels, self._dense = min_points, max_points, max_voxels, dense
        b = self._bounds.reshape(3, 2)
        self._size = ((b[:, 1] - b[:, 0]) / self._shape.astype(np.float32)).astype(np.float32)   # :41
        dist = (b[:, 0] / self._size).astype(np.float32)                                           # :42
        if np.any(np.abs(np.round(dist) - dist) > 1e-3):                                           # :43-44
            raise ValueError("The voxelization grids is not aligned with the origin, which could lead to unexpected behavior!")
        self._offset = np.round(dist).astype(np.int32)                                             # :45
        self._vbounds = np.round(b / self._size.reshape(3, 1)).astype(np.int64)                    # :46
        reduction = (reduction or "NONE").upper()
        if reduction != "NONE" and not dense:
            raise ValueError("Reduction is only for dense voxelization!")
        if reduction not in REDUCTION:
            raise ValueError("Unsupported reduction type in VoxelGenerator!")
        self._reduction = REDUCTION[reduction]
        pf = (max_points_filter or "NONE").upper()
        if pf not in MAX_POINTS_FILTER:
            raise ValueError("Unsupported maximum points filter in VoxelGenerator!")
        self._pf = MAX_POINTS_FILTER[pf]
        vf = (max_voxels_filter or "NONE").upper()
        if vf not in MAX_VOXELS_FILTER:
            raise ValueError("Unsupported maximum voxels filter in VoxelGenerator!")
        self._vf = MAX_VOXELS_FILTER[vf]
        if dense:
            if min_points > 0:
                raise NotImplementedError("Minimum points filtering is not implemented for dense")
            if self._pf not in (0, 1):
                raise NotImplementedError("Only trim is implemented for max points filtering")
            if self._vf not in (0, 1):
                raise NotImplementedError("Only trim is implemented for max voxels filtering")

    def __call__(self, points):
        if self._dense:
            return voxelize_3d_dense(points, self._shape, self._bounds, self._max_points, self._max_voxels, self._reduction)
        sp = voxelize_3d_sparse(points, self._size, 3)
        ret = voxelize_3d_filter(points, sp.points_mapping, sp.coords, sp.voxel_npoints, self._vbounds,
                                 self._min_points, self._max_points, self._max_voxels, self._pf, self._vf)
        ret.coords = ret.coords - self._offset.astype(np.int64)   # :103
        return ret


# ----------------------------------------------------------------------------- box
def _split_rows(n, nthreads):
    nthreads = max(1, min(int(nthreads), int(n) if n > 0 else 1))
    edges = np.linspace(0, n, nthreads + 1).astype(np.int64)
    return [(int(edges[k]), int(edges[k + 1])) for k in range(nthreads)]


def _run_rows(fn, n, nthreads):
    parts = _split_rows(n, nthreads)
    if len(parts) == 1:
        fn(*parts[0])
        return
    import threading
    ts = [threading.Thread(target=fn, args=p) for p in parts]   # ctypes releases the GIL
    [t.start() for t in ts]
    [t.join() for t in ts]


def iou2d_forward(b1, b2, method, nthreads=1):
    """iou2d_forward / iou2dr_forward (iou.cpp:12-46, 95-141) in the dtype of b1 (f32 or f64)."""
    m = IOU_TYPE[method.upper()] if isinstance(method, str) else int(method)
    if m not in (1, 2):
        raise ValueError("Unrecognized iou type!")
    dt = np.float64 if b1.dtype == np.float64 else np.float32
    b1 = np.ascontiguousarray(b1, dtype=dt)
    b2 = np.ascontiguousarray(b2, dtype=dt)
    n, k = b1.shape[0], b2.shape[0]
    out = np.empty((n, k), dt)
    f = lib().oracle_iou2d_f64 if dt == np.float64 else lib().oracle_iou2d_f32
    f.restype = None

    def run(r0, r1):
        f(_p(b1), ctypes.c_int64(n), _p(b2), ctypes.c_int64(k), ctypes.c_int(m),
          ctypes.c_int64(r0), ctypes.c_int64(r1), _p(out))
    _run_rows(run, n, nthreads)
    return out


def iou2d_pairs(b1, b2, pi, pj, method):
    """IoU of the listed pairs (b1[pi[k]], b2[pj[k]]) with the arithmetic of iou2d_forward -> [k]"""
    m = IOU_TYPE[method.upper()] if isinstance(method, str) else int(method)
    dt = np.float64 if b1.dtype == np.float64 else np.float32
    b1 = np.ascontiguousarray(b1, dtype=dt)
    b2 = np.ascontiguousarray(b2, dtype=dt)
    pi = np.ascontiguousarray(pi, dtype=np.int64)
    pj = np.ascontiguousarray(pj, dtype=np.int64)
    out = np.empty((len(pi),), dt)
    f = lib().oracle_iou2d_pairs_f64 if dt == np.float64 else lib().oracle_iou2d_pairs_f32
    f.restype = None
    f(_p(b1), _p(b2), _p(pi), _p(pj), ctypes.c_int64(len(pi)), ctypes.c_int(m), _p(out))
    return out


def aabb_candidate_pairs(b1, b2=None):
    """all ordered pairs (i, j) whose axis-aligned bounding boxes (of the rotated rectangles) overlap or touch: sweep along x
    in numpy (the only pairs whose IoU can be non-zero).  b2 = None: b1 against itself (diagonal included)."""
    def aabb(b):
        b = np.asarray(b, np.float64)
        c, s = np.abs(np.cos(b[:, 4])), np.abs(np.sin(b[:, 4]))
        ex, ey = (np.abs(b[:, 2]) * c + np.abs(b[:, 3]) * s) / 2, (np.abs(b[:, 2]) * s + np.abs(b[:, 3]) * c) / 2
        pad = 1e-9 * (1 + np.abs(b[:, 0]) + np.abs(b[:, 1]) + ex + ey)        # superset: never lose a touching pair to rounding
        return b[:, 0] - ex - pad, b[:, 0] + ex + pad, b[:, 1] - ey - pad, b[:, 1] + ey + pad
    x0a, x1a, y0a, y1a = aabb(b1)
    x0b, x1b, y0b, y1b = aabb(b1 if b2 is None else b2)
    order = np.argsort(x0b, kind="stable")
    xs = x0b[order]
    # for every i: the j (in x order) with x0a[i] - (widest b) <= x0b[j] <= x1a[i]; of those keep x1b[j] >= x0a[i] and the
    # y overlap
    hi = np.searchsorted(xs, x1a, side="right")
    lo = np.searchsorted(xs, x0a - (np.max(x1b - x0b) if len(x0b) else 0.0), side="left")
    out_i, out_j = [], []
    step = max(1, int(4e7 // max(int((hi - lo).max()) if len(hi) else 1, 1)))
    for s0 in range(0, len(x0a), step):
        ii = np.arange(s0, min(s0 + step, len(x0a)))
        cnt = hi[ii] - lo[ii]
        rep = np.repeat(ii, cnt)
        pos = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt) + np.repeat(lo[ii], cnt)
        jj = order[pos]
        ok = (x1b[jj] >= x0a[rep]) & (y0b[jj] <= y1a[rep]) & (y1b[jj] >= y0a[rep])
        out_i.append(rep[ok])
        out_j.append(jj[ok])
    return np.concatenate(out_i).astype(np.int64), np.concatenate(out_j).astype(np.int64)


def loss_iou2dr(b1, b2, kind, nthreads=1):
    """giou2dr_forward (kind "grbox") / diou2dr_forward (kind "drbox") values (iou.cpp:213-258, 322-367), in the dtype of b1"""
    k = {"GRBOX": 0, "DRBOX": 1}[kind.upper()]
    dt = np.float64 if b1.dtype == np.float64 else np.float32
    b1 = np.ascontiguousarray(b1, dtype=dt)
    b2 = np.ascontiguousarray(b2, dtype=dt)
    n, m = b1.shape[0], b2.shape[0]
    out = np.empty((n, m), dt)
    f = lib().oracle_loss_iou2dr_f64 if dt == np.float64 else lib().oracle_loss_iou2dr_f32
    f.restype = None

    def run(r0, r1):
        f(_p(b1), ctypes.c_int64(n), _p(b2), ctypes.c_int64(m), ctypes.c_int(k), ctypes.c_int64(r0), ctypes.c_int64(r1), _p(out))
    _run_rows(run, n, nthreads)
    return out


def iou2dr_flags(b1, b2):
    """autograd bookkeeping of the rotated IoU family -> Dict(nx[n,m], xflags[n,m,8], nm[n,m], mflags[n,m,8], far[n,m,2])"""
    dt = np.float64 if b1.dtype == np.float64 else np.float32
    b1 = np.ascontiguousarray(b1, dtype=dt)
    b2 = np.ascontiguousarray(b2, dtype=dt)
    n, m = b1.shape[0], b2.shape[0]
    r = Dict(nx=np.empty((n, m), np.uint8), xflags=np.empty((n, m, 8), np.uint8), nm=np.empty((n, m), np.uint8),
             mflags=np.empty((n, m, 8), np.uint8), far=np.empty((n, m, 2), np.uint8))
    f = lib().oracle_iou2dr_flags_f64 if dt == np.float64 else lib().oracle_iou2dr_flags_f32
    f.restype = None
    f(_p(b1), ctypes.c_int64(n), _p(b2), ctypes.c_int64(m), _p(r.nx), _p(r.xflags), _p(r.nm), _p(r.mflags), _p(r.far))
    return r


def pdist2dr(points, boxes):
    """pdist2dr_forward (dist.cpp:36-52): (dist[m,n] signed, positive inside; iedge[m,n])"""
    dt = np.float64 if points.dtype == np.float64 else np.float32
    pts = np.ascontiguousarray(points, dtype=dt)
    bx = np.ascontiguousarray(boxes, dtype=dt)
    n, m = pts.shape[0], bx.shape[0]
    dist, iedge = np.empty((m, n), dt), np.empty((m, n), np.uint8)
    f = lib().oracle_pdist2dr_f64 if dt == np.float64 else lib().oracle_pdist2dr_f32
    f.restype = None
    f(_p(pts), ctypes.c_int64(n), _p(bx), ctypes.c_int64(m), _p(dist), _p(iedge))
    return dist, iedge


def box2d_iou(boxes1, boxes2, method="box", precise=True, nthreads=1):
    """box/__init__.py:180-224 (numpy in, numpy out)"""
    boxes1 = np.asarray(boxes1)
    boxes2 = np.asarray(boxes2)
    otype = boxes1.dtype
    if precise:
        boxes1 = boxes1.astype(np.float64)
        boxes2 = boxes2.astype(np.float64)
    if boxes1.ndim != 2 or boxes2.ndim != 2:
        raise ValueError("Input of rbox_2d_iou should be Nx2 tensors!")
    if boxes1.shape[1] != 5 or boxes2.shape[1] != 5:
        raise ValueError("Input boxes should have 5 fields: x, y, w, h, r")
    if method.upper() in ("GRBOX", "DRBOX"):
        res = loss_iou2dr(boxes1, boxes2, method, nthreads)
    else:
        res = iou2d_forward(boxes1, boxes2, method, nthreads)
    return res.astype(otype) if precise else res


def nms2d(boxes, scores, iou_type, supression_type, iou_threshold, score_threshold, supression_param):
    """nms2d (nms.cpp:98-119): returns the SUPPRESSED mask."""
    it = IOU_TYPE[iou_type.upper()] if isinstance(iou_type, str) else int(iou_type)
    st = SUPRESSION[supression_type.upper()] if isinstance(supression_type, str) else int(supression_type)
    if it not in (1, 2):
        raise ValueError("Unsupported iou type!")          # common.h:25
    dt = np.float64 if boxes.dtype == np.float64 else np.float32
    boxes = np.ascontiguousarray(boxes, dtype=dt)
    scores = np.ascontiguousarray(scores, dtype=dt)
    n = boxes.shape[0]
    order = np.argsort(-scores, kind="stable").astype(np.int64)   # nms.cpp:103 (stable tie-break = spec)
    sup = np.zeros((n,), np.uint8)
    f = lib().oracle_nms2d_f64 if dt == np.float64 else lib().oracle_nms2d_f32
    f.restype = None
    f(_p(boxes), _p(scores), ctypes.c_int64(n), _p(order), ctypes.c_int(it), ctypes.c_int(st),
      ctypes.c_float(iou_threshold), ctypes.c_float(score_threshold), ctypes.c_float(supression_param), _p(sup))
    return sup.astype(bool)


def box2d_nms_hard_candidates(boxes, scores, iou_method="rbox", iou_threshold=0.0, score_threshold=0.0):
    """box2d_nms(..., supression_method="hard") with the greedy loop of nms.cpp:32-59 visiting only the pairs whose bounding
    boxes touch (every other pair has IoU 0 <= threshold): the same KEEP mask, affordable at config 3's 100 k boxes"""
    assert iou_threshold >= 0
    boxes = np.ascontiguousarray(boxes, dtype=np.float64)
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    n = len(boxes)
    it = IOU_TYPE[iou_method.upper()]
    pi, pj = aabb_candidate_pairs(boxes)
    o = np.argsort(pi, kind="stable")
    pi, pj = pi[o], np.ascontiguousarray(pj[o])
    off = np.zeros((n + 1,), np.int64)
    np.cumsum(np.bincount(pi, minlength=n), out=off[1:])
    order = np.argsort(-scores, kind="stable").astype(np.int64)
    sup = np.zeros((n,), np.uint8)
    f = lib().oracle_nms2d_hard_candidates_f64
    f.restype = None
    f(_p(boxes), _p(scores), ctypes.c_int64(n), _p(order), _p(off), _p(pj), ctypes.c_int(it),
      ctypes.c_float(iou_threshold), ctypes.c_float(score_threshold), _p(sup))
    return ~sup.astype(bool)


def box2d_nms_soft_candidates(boxes, scores, iou_method="rbox", supression_method="linear", iou_threshold=0.0,
                              score_threshold=0.0, supression_param=1.0):
    """box2d_nms(..., supression_method="linear" | "gaussian") with the rescaling loop of nms.cpp:41-72 visiting only the pairs
    whose bounding boxes touch: the same KEEP mask (tested against the literal loop), affordable beyond 65 k boxes"""
    assert iou_threshold >= 0
    boxes = np.ascontiguousarray(boxes, dtype=np.float64)
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    n = len(boxes)
    it, st = IOU_TYPE[iou_method.upper()], SUPRESSION[supression_method.upper()]
    assert st in (1, 2)
    a, b = aabb_candidate_pairs(boxes)            # ordered pairs, both directions (positions change during the run)
    o = np.argsort(a, kind="stable")
    a, b = a[o], np.ascontiguousarray(b[o])
    off = np.zeros((n + 1,), np.int64)
    np.cumsum(np.bincount(a, minlength=n), out=off[1:])
    order = np.argsort(-scores, kind="stable").astype(np.int64)
    sup = np.zeros((n,), np.uint8)
    f = lib().oracle_nms2d_soft_candidates_f64
    f.restype = None
    f(_p(boxes), _p(scores), ctypes.c_int64(n), _p(order), _p(off), _p(b), ctypes.c_int(it), ctypes.c_int(st),
      ctypes.c_float(iou_threshold), ctypes.c_float(score_threshold), ctypes.c_float(supression_param), _p(sup))
    return ~sup.astype(bool)


def box2d_nms(boxes, scores, iou_method="box", supression_method="hard",
              iou_threshold=0, score_threshold=0, supression_param=0, precise=True):
    """box/__init__.py:226-276: returns the KEEP mask."""
    boxes = np.asarray(boxes)
    scores = np.asarray(scores)
    if precise:
        boxes = boxes.astype(np.float64)
        scores = scores.astype(np.float64)
    if len(boxes) != len(scores):
        raise ValueError("Numbers of boxes and scores are inconsistent!")
    if scores.ndim == 2:
        scores = scores.max(axis=1)
    if boxes.size == 0:
        return np.zeros((0,), bool)
    return ~nms2d(boxes, scores, iou_method, supression_method, iou_threshold, score_threshold, supression_param)


def iou3d(boxes1, boxes2, method="rbox", nthreads=1):
    """batched box3dr_iou ("rbox") / box3d_iou ("box") on [n,7] = (x,y,z,lx,ly,lz,rz), fp32
    (dgal_wrap.h:45-91; pair loop matcher.pyx:57-80)."""
    rot = {"RBOX": 1, "BOX": 0}[method.upper()]
    b1 = np.ascontiguousarray(boxes1, dtype=np.float32)
    b2 = np.ascontiguousarray(boxes2, dtype=np.float32)
    n, m = b1.shape[0], b2.shape[0]
    out = np.empty((n, m), np.float32)
    f = lib().oracle_iou3d
    f.restype = None

    def run(r0, r1):
        f(_p(b1), ctypes.c_int64(n), _p(b2), ctypes.c_int64(m), ctypes.c_int(rot),
          ctypes.c_int64(r0), ctypes.c_int64(r1), _p(out))
    _run_rows(run, n, nthreads)
    return out


def crop_2dr(points, boxes):
    """crop_2dr (utils.cpp:38-47): bool[M,N], entry [i,j] = point j lies in rotated box i"""
    dt = np.float64 if points.dtype == np.float64 else np.float32
    pts = np.ascontiguousarray(points, dtype=dt)
    bx = np.ascontiguousarray(boxes, dtype=dt)
    n, m = pts.shape[0], bx.shape[0]
    out = np.zeros((m, n), np.uint8)
    f = lib().oracle_crop_2dr_f64 if dt == np.float64 else lib().oracle_crop_2dr_f32
    f.restype = None
    f(_p(pts), ctypes.c_int64(n), _p(bx), ctypes.c_int64(m), _p(out))
    return out.astype(bool)


def box3dp_crop(points, boxes, project_axis=2):
    """box/__init__.py:289-315 on numpy arrays"""
    ax2 = {0: ([1, 2], [1, 2, 4, 5, 6]), 1: ([0, 2], [0, 2, 3, 5, 6]), 2: ([0, 1], [0, 1, 3, 4, 6])}
    if project_axis not in ax2:
        raise ValueError("The projection axis can only be 0-x, 1-y and 2-z!")
    pi, bi = ax2[project_axis]
    mask_2d = crop_2dr(np.ascontiguousarray(points[:, pi]), np.ascontiguousarray(boxes[:, bi]))
    pp = points[:, [project_axis]].T
    bp = boxes[:, [project_axis]]
    bd = boxes[:, [3 + project_axis]] / 2
    return mask_2d & ((pp - bd < bp) & (bp < pp + bd))


def _box_rows(boxes):
    """[M,7] (x,y,z,lx,ly,lz,rz) or the [n,9] rows of Target3DArray.to_numpy (label, score, then the 7) -> (array, stride, offset)"""
    bx = np.ascontiguousarray(boxes, dtype=np.float32)
    if bx.ndim != 2 or bx.shape[1] not in (7, 9):
        raise ValueError("boxes should be [M,7] or [M,9]")
    return bx, bx.shape[1], 0 if bx.shape[1] == 7 else 2


def crop_points(boxes, cloud):
    """Target3DArray.crop_points (abstraction.pyx:684-687; per pair box3dr_contains, dgal_wrap.h:6-19): bool[M,N]"""
    bx, bs, bo = _box_rows(boxes)
    pts = np.ascontiguousarray(cloud, dtype=np.float32)
    n, m = pts.shape[0], bx.shape[0]
    out = np.zeros((m, n), np.uint8)
    f = lib().oracle_crop_3dr
    f.restype = None
    f(_p(pts), ctypes.c_int64(n), ctypes.c_int(pts.shape[1]), _p(bx), ctypes.c_int64(m), ctypes.c_int(bs), ctypes.c_int(bo), _p(out))
    return out.astype(bool)


def paint_label(boxes, cloud, semantics, labels=None):
    """Target3DArray.paint_label (abstraction.pyx:662-682): uint16[N]; labels default to column 0 of [n,9] rows"""
    bx, bs, bo = _box_rows(boxes)
    if labels is None:
        if bs != 9:
            raise ValueError("labels are needed with [M,7] boxes")
        labels = bx[:, 0]
    lab = np.ascontiguousarray(labels).astype(np.uint8)
    pts = np.ascontiguousarray(cloud, dtype=np.float32)
    sem = np.ascontiguousarray(semantics, dtype=np.uint8)
    n, m = pts.shape[0], bx.shape[0]
    out = np.zeros((n,), np.uint16)
    f = lib().oracle_paint_label
    f.restype = None
    f(_p(pts), ctypes.c_int64(n), ctypes.c_int(pts.shape[1]), _p(sem), _p(bx), ctypes.c_int64(m), ctypes.c_int(bs), ctypes.c_int(bo),
      _p(lab), _p(out))
    return out


# --------------------------------------------------------------------------- point (next row: d3d.point)
ALIGN_TYPE = {"DROP": 0, "MEAN": 1, "LINEAR": 2, "MAX": 3, "NEAREST": 4}   # point/scatter.h:37


def _scatter_args(coord, image):
    dt = np.float64 if image.dtype == np.float64 else np.float32
    coord = np.ascontiguousarray(coord, dtype=dt)
    image = np.ascontiguousarray(image, dtype=dt)
    dim = coord.shape[1] - 1
    if dim not in (1, 2, 3) or image.ndim != dim + 2:
        raise ValueError("Unsupported dimension size: %d" % dim)          # scatter.h:33
    dims = np.ascontiguousarray(image.shape[2:], dtype=np.int64)
    return coord, image, dim, dims, dt


def aligned_scatter_forward(coord, image, atype):
    """scatter.cpp:182-194"""
    at = ALIGN_TYPE[atype.upper()] if isinstance(atype, str) else int(atype)
    if at not in (1, 2):
        raise ValueError("Unsupported align type!")                       # scatter.h:18
    coord, image, dim, dims, dt = _scatter_args(coord, image)
    n, C = coord.shape[0], image.shape[1]
    out = np.empty((n, C), dt)
    f = lib().oracle_aligned_scatter_forward_f64 if dt == np.float64 else lib().oracle_aligned_scatter_forward_f32
    f.restype = None
    f(_p(coord), ctypes.c_int64(n), ctypes.c_int(dim), _p(image), ctypes.c_int64(C), _p(dims), ctypes.c_int(at), _p(out))
    return out


def aligned_scatter_backward(coord, grad, atype, image_shape):
    """scatter.cpp:196-200 into a zero image of `image_shape`; returns the image gradient"""
    at = ALIGN_TYPE[atype.upper()] if isinstance(atype, str) else int(atype)
    if at not in (1, 2):
        raise ValueError("Unsupported align type!")
    dt = np.float64 if grad.dtype == np.float64 else np.float32
    img = np.zeros(image_shape, dt)
    coord, img, dim, dims, dt = _scatter_args(coord, img)
    grad = np.ascontiguousarray(grad, dtype=dt)
    n, C = coord.shape[0], img.shape[1]
    f = lib().oracle_aligned_scatter_backward_f64 if dt == np.float64 else lib().oracle_aligned_scatter_backward_f32
    f.restype = None
    f(_p(coord), ctypes.c_int64(n), ctypes.c_int(dim), _p(grad), ctypes.c_int64(C), _p(dims), ctypes.c_int(at), _p(img))
    return img


# --------------------------------------------------------------------------- matcher / evaluator (reference Cython, restated)
def prepare_boxes(src_arr, dst_arr, rotated=True):
    """BaseMatcher.prepare_boxes (matcher.pyx:46-80), metrics IoU / RIoU: f32[n,m] = 1 - box3d[r]_iou after the +-1e3 clip"""
    src = np.array(src_arr, dtype=np.float32).reshape(-1, 9)
    dst = np.array(dst_arr, dtype=np.float32).reshape(-1, 9)
    src[:, 5:8] = np.clip(src[:, 5:8], -1e3, 1e3)
    dst[:, 5:8] = np.clip(dst[:, 5:8], -1e3, 1e3)
    if len(src) == 0 or len(dst) == 0:
        return np.zeros((len(src), len(dst)), np.float32)
    return (np.float32(1) - iou3d(src[:, 2:9], dst[:, 2:9], "rbox" if rotated else "box")).astype(np.float32)


def score_match(cache, src_arr, dst_arr, src_subset, dst_subset, distance_threshold, literal=False):
    """ScoreMatcher.match + match_by_order (matcher.pyx:90-162) -> (src_assignment, dst_assignment) dicts.
    literal=False (what the product's default implements): every source walks ITS OWN row of the distance matrix, nearest
    destination first -- the behaviour the reference's class docstring describes.
    literal=True: the reference's loop as written -- the k-th best source is paired with the distance order of the k-th ROW OF
    THE SUBSET (`dst_order[src_idx, dst_idx]` next to `src_order[src_idx]`, matcher.pyx:155-158: the loop counter indexes the
    unsorted row), so a source may take a destination that is merely the first acceptable one in ANOTHER box's order.  The two
    agree whenever no source has more than one acceptable destination (the reference's own test_calc_stats) or the subset
    happens to be sorted by score.
    literal=True makes the reference's own call for the score order -- np.flip(np.argsort(scores)), so equal scores come out as they
    do there -- and hands a destination tag missing from `distance_threshold` the 0.0 that unordered_map::operator[] inserts (:112).
    (otherwise: stable sorts, i.e. equal scores in index order; equal distances go to the lower index in both modes -- the reference's
    unstable np.argsort leaves that open)"""
    src_assign, dst_assign = {}, {}
    src_subset, dst_subset = list(src_subset), list(dst_subset)
    if not src_subset or not dst_subset:
        return src_assign, dst_assign
    scores = np.array([src_arr[i, 1] for i in src_subset])
    src_order = np.flip(np.argsort([float(x) for x in scores])) if literal else np.argsort(-scores, kind="stable")
    sub = cache[np.ix_(src_subset, dst_subset)]
    dst_order = np.argsort(sub, axis=1, kind="stable")
    for si in range(len(src_subset)):
        s = src_subset[src_order[si]]
        for di in range(len(dst_subset)):
            d = dst_subset[dst_order[si if literal else src_order[si], di]]
            if s in src_assign:
                continue
            if d in dst_assign:
                continue
            if int(src_arr[s, 0]) != int(dst_arr[d, 0]):
                continue
            if cache[s, d] <= (distance_threshold.get(int(dst_arr[d, 0]), 0.0) if literal else distance_threshold[int(dst_arr[d, 0])]):
                src_assign[s] = d
                dst_assign[d] = s
    return src_assign, dst_assign


def calc_stats(gt, dt, classes, max_distance, thresholds, literal=False):
    """DetectionEvaluator.calc_stats (benchmarks.pyx:178-283) on [n,9] arrays, one matching PER threshold like the reference;
    literal: with ScoreMatcher.match's pairing as written (score_match(literal=True)) -- the reference's own numbers"""
    gt = np.asarray(gt, np.float32).reshape(-1, 9)
    dt = np.asarray(dt, np.float32).reshape(-1, 9)
    T = len(thresholds)
    cache = prepare_boxes(dt, gt, True)
    st = Dict(ngt={c: 0 for c in classes}, ndt={c: [0] * T for c in classes}, tp={c: [0] * T for c in classes},
              fp={c: [0] * T for c in classes}, fn={c: [0] * T for c in classes})
    acc = {k: [dict() for _ in range(T)] for k in ("iou", "angular", "dist", "box")}
    gt_idx = []
    for g in range(len(gt)):
        if int(gt[g, 0]) in classes:
            st.ngt[int(gt[g, 0])] += 1
            gt_idx.append(g)
    for t in range(T):
        dt_idx = []
        for d in range(len(dt)):
            if int(dt[d, 0]) not in classes or dt[d, 1] < thresholds[t]:
                continue
            st.ndt[int(dt[d, 0])][t] += 1
            dt_idx.append(d)
        sa, da = score_match(cache, dt, gt, dt_idx, gt_idx, max_distance, literal=literal)
        for g in gt_idx:
            c = int(gt[g, 0])
            if g not in da:
                st.fn[c][t] += 1
                continue
            st.tp[c][t] += 1
            d = da[g]
            acc["iou"][t][g] = 1 - cache[d, g]
            acc["dist"][t][g] = float(np.linalg.norm(gt[g, 2:5] - dt[d, 2:5]))
            acc["box"][t][g] = float(np.linalg.norm(gt[g, 5:8] - dt[d, 5:8]))
            dy = float(gt[g, 8] - dt[d, 8])
            acc["angular"][t][g] = abs((dy + np.pi) % (2 * np.pi) - np.pi) / np.pi
        for d in dt_idx:
            if d not in sa:
                st.fp[int(dt[d, 0])][t] += 1
    for name in acc:
        agg = {c: [float("nan")] * T for c in classes}
        for t in range(T):
            for c in classes:
                vals = [v for g, v in acc[name][t].items() if int(gt[g, 0]) == c]
                if vals:
                    agg[c][t] = float(np.sum(np.asarray(vals, np.float32)) / len(vals))
        st["acc_" + name] = agg
    return st


def score_match_rows(cache, src_arr, dst_arr, distance_threshold):
    """the same association over ALL boxes, one numpy step per src row instead of match_by_order's n x m pair loop:
    -> (src_match[n], dst_match[m]) index arrays, -1 = none (equality with score_match is tested on the CPU)"""
    n, m = cache.shape
    src_match, dst_match = np.full((n,), -1, np.int64), np.full((m,), -1, np.int64)
    stag, dtag = src_arr[:, 0].astype(np.int64), dst_arr[:, 0].astype(np.int64)
    thr = np.array([distance_threshold.get(int(t), -np.inf) for t in dtag], np.float32)
    for s in np.argsort(-src_arr[:, 1], kind="stable"):
        if int(stag[s]) not in distance_threshold:
            continue
        ok = np.nonzero((dtag == stag[s]) & (cache[s] <= thr) & (dst_match < 0))[0]
        if len(ok):
            d = ok[np.argmin(cache[s, ok])]            # first minimum = lowest index among equal distances
            src_match[s], dst_match[d] = d, s
    return src_match, dst_match
