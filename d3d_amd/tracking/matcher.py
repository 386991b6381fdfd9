"""d3d_amd.tracking.matcher -- BaseMatcher.prepare_boxes + ScoreMatcher of the reference (d3d/tracking/matcher.pyx:12-162)
on arrays: boxes are [n,9] float32 rows (label, score, x, y, z, lx, ly, lz, yaw), the layout Target3DArray.to_numpy produces
(d3d/abstraction.pyx:263-272) and prepare_boxes consumes (matcher.pyx:46-51).  The container classes of d3d.abstraction are
outside this library's scope; everything that is arithmetic runs in HIP kernels (d3d_match_distance, d3d_score_match).
"""
import ctypes
import enum

import numpy as np
import torch

from .. import _lib


class DistanceTypes(enum.IntEnum):      # matcher.pxd:5-8
    IoU = 1
    RIoU = 2
    Position = 3


def _boxes(a, what):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)) if isinstance(a, np.ndarray) else a.to(torch.float32)
    if t.dim() != 2 or t.shape[1] != 9:
        raise ValueError("%s should be [n,9] rows of (label, score, x, y, z, lx, ly, lz, yaw)" % what)
    return t


def prepare_boxes(src_arr, dst_arr, distance_metric):
    """the distance cache of BaseMatcher.prepare_boxes (matcher.pyx:25-80): f32[n,m] on the device.
    IoU -> 1 - box3d_iou, RIoU -> 1 - box3dr_iou (dimensions clipped to +-1e3 first, :49-51), Position -> euclidean distance
    of the centres (the reference's cdist over columns 0:3 of these arrays, :82, compares (label, score, x): a slip)."""
    lib = _lib.load()
    metric = DistanceTypes(int(distance_metric))
    src, dst = _boxes(src_arr, "src_boxes"), _boxes(dst_arr, "dst_boxes")
    dev = src.device if src.is_cuda else (dst.device if dst.is_cuda else _lib.require_gpu())
    src, dst = src.to(dev).contiguous(), dst.to(dev).contiguous()
    n, m = src.shape[0], dst.shape[0]
    with torch.cuda.device(dev):
        if metric == DistanceTypes.Position:
            return torch.cdist(src[:, 2:5], dst[:, 2:5], compute_mode="donot_use_mm_for_euclid_dist")
        cache = torch.empty((n, m), dtype=torch.float32, device=dev)
        ws = _lib.workspace(lib.d3d_iou3d_workspace_bytes(n, m), dev)
        rc = lib.d3d_match_distance(_lib.ptr(src), n, _lib.ptr(dst), m, 1 if metric == DistanceTypes.RIoU else 0, _lib.ptr(cache),
                                    _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "match_distance")
    return cache


def score_match(distance, src_scores, src_tags, dst_tags, distance_threshold):
    """ScoreMatcher.match over ALL boxes (matcher.pyx:142-162 + match_by_order :90-121): src rows from the best score down,
    each takes the nearest unassigned dst of its own tag with distance <= distance_threshold[tag].  Tags absent from
    `distance_threshold` (or negative) take no part.  Returns (src_match[n], dst_match[m]) int32 tensors on the device,
    -1 = unmatched.  The matching of a score threshold t is this result restricted to the src with score >= t: a box's choice
    only depends on the boxes before it (benchmarks.pyx:218-238 recomputes it per threshold).  Any threshold is exact (rows with
    more than 64 candidates list their 64 nearest and fall back to a sweep of the row).  Ties: equal scores are taken
    in index order, equal distances go to the lower dst index (the reference's argsorts leave both unspecified)."""
    lib = _lib.load()
    dev = distance.device
    n, m = distance.shape
    def host(a):                       # tags / scores: numpy arrays, lists, or tensors on any device
        return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    # host-side preparation in numpy (a frame has a few hundred boxes: tensor ops would each cost more than the arithmetic),
    # shipped to the device as ONE buffer: [src scores f32 | src tags i32 | dst tags i32 | dst thresholds f32]
    st = host(src_tags).astype(np.int32, copy=True).reshape(-1)
    dt = host(dst_tags).astype(np.int32, copy=True).reshape(-1)
    thr = np.full((m,), np.nan, np.float32)
    known = np.zeros((max(int(st.max()) if n else 0, int(dt.max()) if m else 0, 0) + 2,), bool)
    for tag, v in distance_threshold.items():
        if 0 <= int(tag) < known.size:
            known[int(tag)] = True
        thr[dt == int(tag)] = float(v)
    st[(st < 0) | ~known[np.clip(st, 0, None)]] = -1
    dt[(dt < 0) | ~known[np.clip(dt, 0, None)]] = -1
    # the score order: a stable descending sort ON THE DEVICE (equal scores in index order, as numpy's stable argsort of the
    # negated scores gives them; 20 k scores: 1 ms of numpy on the host against ~50 us) -- the scores ride in the one buffer
    sc = host(src_scores).astype(np.float32).reshape(-1)
    packed = torch.from_numpy(np.concatenate([sc.view(np.uint8), st.view(np.uint8), dt.view(np.uint8), thr.view(np.uint8)]))
    with torch.cuda.device(dev):
        packed = packed.to(dev)
        order = torch.sort(packed[:4 * n].view(torch.float32), descending=True, stable=True).indices
        st, dt = packed[4 * n:8 * n].view(torch.int32), packed[8 * n:8 * n + 4 * m].view(torch.int32)
        thr = packed[8 * n + 4 * m:].view(torch.float32)
        src_match = torch.empty((n,), dtype=torch.int32, device=dev)
        dst_match = torch.empty((m,), dtype=torch.int32, device=dev)
        status = torch.zeros((1,), dtype=torch.int32, device=dev)
        ws = _lib.workspace(lib.d3d_score_match_workspace_bytes(n, m), dev)
        rc = lib.d3d_score_match(_lib.ptr(distance.contiguous()), n, m, _lib.ptr(st), _lib.ptr(dt), _lib.ptr(thr), _lib.ptr(order),
                                 _lib.ptr(src_match), _lib.ptr(dst_match), _lib.ptr(status), _lib.ptr(ws), ws.numel(),
                                 _lib.stream_ptr())
        _lib.check(rc, "score_match")
    return src_match, dst_match


def score_match_reference_compat(distance, src_scores, src_tags, dst_tags, distance_threshold, src_subset, dst_subset):
    """ScoreMatcher.match exactly as matcher.pyx:142-162 pairs the two orders -- INCLUDING its row mix-up: the k-th best source
    (`src_subset[src_order[k]]`, :157) walks the destinations in the distance order of the k-th row OF THE SUBSET
    (`dst_order[src_idx, ...]` with src_idx the loop counter, :158), i.e. of another box whenever the subset is not already
    sorted by score, and takes the first one that is free, of its tag and within the threshold (match_by_order :96-115 skips a
    failing pair and goes on).  `score_match` (the default) walks the source's OWN row -- nearest first -- which is what the
    docstring of the reference's class describes; this function exists so that the difference can be measured
    (tests/test_gpu_boxloss.py: a detection within the threshold of two ground truths).  A sequential loop of small tensor
    operations on the device of `distance` (one source per step): a parity mode, not a fast path.
    Returns (src_match[n], dst_match[m]) int32 tensors, -1 = unmatched."""
    dev = distance.device
    n, m = distance.shape
    src_subset = torch.as_tensor(list(src_subset), dtype=torch.int64, device=dev)
    dst_subset = torch.as_tensor(list(dst_subset), dtype=torch.int64, device=dev)
    src_match = torch.full((n,), -1, dtype=torch.int32, device=dev)
    dst_match = torch.full((m,), -1, dtype=torch.int32, device=dev)
    if src_subset.numel() == 0 or dst_subset.numel() == 0:
        return src_match, dst_match
    scores = torch.as_tensor(np.asarray(src_scores.detach().cpu() if isinstance(src_scores, torch.Tensor) else src_scores, np.float32)).to(dev)
    stags = torch.as_tensor(np.asarray(src_tags.detach().cpu() if isinstance(src_tags, torch.Tensor) else src_tags, np.int64)).to(dev)
    dtags = torch.as_tensor(np.asarray(dst_tags.detach().cpu() if isinstance(dst_tags, torch.Tensor) else dst_tags, np.int64)).to(dev)
    thr = torch.full((m,), float("nan"), dtype=torch.float32, device=dev)
    for tag, v in distance_threshold.items():
        thr[dtags == int(tag)] = float(v)
    # np.flip(np.argsort(scores)) (:146); ties: the stable descending order of this library's other sorts
    src_order = torch.argsort(-scores[src_subset], stable=True)
    sub = distance[src_subset][:, dst_subset]
    dst_order = torch.argsort(sub, dim=1, stable=True)                       # (:148) row k = subset row k
    free = torch.ones((m,), dtype=torch.bool, device=dev)
    for k in range(src_subset.numel()):
        s = src_subset[src_order[k]]
        cand = dst_subset[dst_order[k]]                                      # (:158) the LOOP COUNTER's row, not the source's
        ok = free[cand] & (dtags[cand] == stags[s]) & (distance[s, cand] <= thr[cand])
        first = torch.argmax(ok.to(torch.int8))
        hit = ok[first]
        d = cand[first]
        src_match[s] = torch.where(hit, d.to(torch.int32), src_match[s])
        dst_match[d] = torch.where(hit, s.to(torch.int32), dst_match[d])
        free[d] = free[d] & ~hit
    return src_match, dst_match


class ScoreMatcher:
    """array-level ScoreMatcher (matcher.pyx:138-162): prepare_boxes, match on subsets, query_* -- same call sequence as the
    reference's evaluator uses (benchmarks.pyx:188-238).  `reference_compat=True` reproduces the reference's pairing of the
    k-th best source with the k-th subset row's distance order (matcher.pyx:155-158); the default pairs
    every source with its own nearest destinations."""

    def __init__(self, reference_compat=False):
        self._cache = None
        self._src = self._dst = None
        self._src_assignment, self._dst_assignment = {}, {}
        self.reference_compat = bool(reference_compat)

    def clear_match(self):
        self._src_assignment, self._dst_assignment = {}, {}

    def prepare_boxes(self, src_arr, dst_arr, distance_metric):
        self.clear_match()
        self._src, self._dst = _boxes(src_arr, "src_boxes"), _boxes(dst_arr, "dst_boxes")
        if len(self._src) == 0 or len(self._dst) == 0:
            self._cache = torch.zeros((len(self._src), len(self._dst)), dtype=torch.float32)      # matcher.pyx:41-43
            return
        self._cache = prepare_boxes(self._src, self._dst, distance_metric)

    @property
    def distance_cache(self):
        return self._cache

    def match(self, src_subset, dst_subset, distance_threshold):
        self.clear_match()
        src_subset, dst_subset = list(src_subset), list(dst_subset)
        if not src_subset or not dst_subset or self._cache.numel() == 0:
            return
        n, m = self._cache.shape
        stags = np.full((n,), -1, np.int64)
        dtags = np.full((m,), -1, np.int64)
        stags[src_subset] = self._src[src_subset, 0].cpu().numpy().astype(np.int64)
        dtags[dst_subset] = self._dst[dst_subset, 0].cpu().numpy().astype(np.int64)
        if self.reference_compat:
            sm, dm = score_match_reference_compat(self._cache, self._src[:, 1].cpu().numpy(), stags, dtags, distance_threshold,
                                                  src_subset, dst_subset)
        else:
            sm, dm = score_match(self._cache, self._src[:, 1].cpu().numpy(), stags, dtags, distance_threshold)
        sm, dm = sm.cpu().numpy(), dm.cpu().numpy()
        self._src_assignment = {int(i): int(j) for i, j in enumerate(sm) if j >= 0}
        self._dst_assignment = {int(j): int(i) for j, i in enumerate(dm) if i >= 0}

    def query_src_match(self, src_idx):
        return self._src_assignment.get(int(src_idx), -1)

    def query_dst_match(self, dst_idx):
        return self._dst_assignment.get(int(dst_idx), -1)

    def num_of_matches(self):
        return len(self._src_assignment)
