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
    failing pair and goes on).  `score_match` walks the source's OWN row -- nearest first -- which is what the docstring of the
    reference's class describes.
    On the device (round 6; a sequential loop of tensor operations before): slot k's acceptable destinations are those of the
    k-th best source (its tag, ITS distance within the threshold, :103-112), their order is that of row k of the subset -- so the
    matrix P[k, j] = distance[subset row k, j] where (k-th best source, j) is acceptable, +inf elsewhere, handed to
    d3d_score_match with the slots in order, reproduces the loop: every slot takes its smallest free entry.
    The score order is the reference's own call on the same values, np.flip(np.argsort(scores)) on the host (so equal scores
    come out as they do there); equal DISTANCES go to the destination earlier in `dst_subset` (np.argsort's unstable default
    leaves that open in the reference); a destination tag missing from `distance_threshold` gets the threshold 0.0, as
    unordered_map::operator[] hands out (:112).
    Returns (src_match[n], dst_match[m]) int32 tensors, -1 = unmatched."""
    return ReferenceAssociation(distance, src_scores, src_tags, dst_tags, distance_threshold, dst_subset).match(src_subset)


_BATCH_MAX_ROWS = 1 << 21              # stacked rows of one batched call (its candidate lists: 512 B of workspace per row)


def _index_array(subset):
    return np.asarray(subset if isinstance(subset, np.ndarray) else list(subset), dtype=np.int64).reshape(-1)


class ReferenceAssociation:
    """score_match_reference_compat for MANY source subsets against one destination subset (the evaluator: 40 score thresholds,
    benchmarks.pyx:218-238): what does not depend on the sources' subset is prepared once -- the destinations' columns of the
    distance matrix, and for every (source, destination) whether the pair is acceptable (tag, the source's OWN distance within the
    destination tag's threshold) -- and `match` queues its work on the stream without reading anything back: the caller
    synchronises once, when it fetches the results."""

    def __init__(self, distance, src_scores, src_tags, dst_tags, distance_threshold, dst_subset):
        def host(a):
            return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        self.dev = dev = distance.device
        self.n, self.m = distance.shape
        self.scores = host(src_scores).astype(np.float32).reshape(-1)
        stags, dtags = host(src_tags).astype(np.int64).reshape(-1), host(dst_tags).astype(np.int64).reshape(-1)
        self.dsub = dsub = _index_array(dst_subset)
        if dsub.size == 0 or self.n == 0:
            return
        # a destination tag missing from `distance_threshold` gets 0.0, as unordered_map::operator[] hands out (matcher.pyx:112)
        thr = np.zeros((dsub.size,), np.float32)
        dt_sub = dtags[dsub]
        for tag in np.unique(dt_sub):
            thr[dt_sub == tag] = np.float32(float(distance_threshold.get(int(tag), 0.0)))
        with torch.cuda.device(dev):
            self.dsub_t = torch.from_numpy(dsub).to(dev)
            self.cols = distance.index_select(1, self.dsub_t)                                   # [n, md]
            self.ok = (self.cols <= torch.from_numpy(thr).to(dev)[None, :]) & \
                      (torch.from_numpy(stags).to(dev)[:, None] == torch.from_numpy(dt_sub).to(dev)[None, :])
            md = dsub.size
            self._dtag0 = torch.zeros((md,), dtype=torch.int32, device=dev)
            self._dthr = torch.full((md,), 3.0e38, dtype=torch.float32, device=dev)
            self._inf = torch.full((), float("inf"), dtype=self.cols.dtype, device=dev)
            self._ok_u8 = self.ok.view(torch.uint8)

    def match(self, src_subset):
        """-> (src_match[n], dst_match[m]) int32 tensors on the device, -1 = unmatched (nothing is read back here)"""
        dev, n, m = self.dev, self.n, self.m
        src_match = torch.full((n,), -1, dtype=torch.int32, device=dev)
        dst_match = torch.full((m,), -1, dtype=torch.int32, device=dev)
        ssub = _index_array(src_subset)
        if ssub.size == 0 or self.dsub.size == 0:
            return src_match, dst_match
        # matcher.pyx:145-146: np.flip(np.argsort([scores of the subset])) -- the same call on the same float64 values (a Python
        # list of floats becomes a float64 array), so equal scores come out as they do there
        src_order = np.flip(np.argsort(self.scores[ssub].astype(np.float64)))
        best = ssub[src_order]                                                                # slot k's source
        lib = _lib.load()
        k, md = int(ssub.size), int(self.dsub.size)
        with torch.cuda.device(dev):
            both = torch.from_numpy(np.concatenate([ssub, best])).to(dev)
            ssub_t, best_t = both[:k], both[k:]
            # P[k, j] = distance[subset row k, j] where (k-th best source, j) is acceptable, +inf elsewhere
            pref = torch.where(self.ok.index_select(0, best_t), self.cols.index_select(0, ssub_t), self._inf)
            sm = torch.empty((k,), dtype=torch.int32, device=dev)
            dm = torch.empty((md,), dtype=torch.int32, device=dev)
            status = torch.zeros((1,), dtype=torch.int32, device=dev)
            order = torch.arange(k, dtype=torch.int64, device=dev)                            # the slots are in order already
            stag0 = torch.zeros((k,), dtype=torch.int32, device=dev)
            ws = _lib.workspace(lib.d3d_score_match_workspace_bytes(k, md), dev)
            rc = lib.d3d_score_match(_lib.ptr(pref), k, md, _lib.ptr(stag0), _lib.ptr(self._dtag0), _lib.ptr(self._dthr), _lib.ptr(order),
                                     _lib.ptr(sm), _lib.ptr(dm), _lib.ptr(status), _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
            _lib.check(rc, "score_match")
            neg = torch.full((), -1, dtype=torch.int32, device=dev)
            src_match[best_t] = torch.where(sm >= 0, self.dsub_t[sm.clamp_min(0).long()].to(torch.int32), neg)
            dst_match[self.dsub_t] = torch.where(dm >= 0, best_t[dm.clamp_min(0).long()].to(torch.int32), neg)
        return src_match, dst_match


    def match_many(self, src_subsets):
        """`match` for a list of source subsets -> (src_match[T, n], dst_match[T, m]) int32 tensors on the device, row t = the
        association of src_subsets[t].  ONE d3d_score_match_batched call (a few above _BATCH_MAX_ROWS stacked rows): the problems'
        rows stacked, and no stacked matrix -- slot k of a problem reads its distances from the k-th subset row of the prepared
        columns and its acceptable pairs from the k-th best source's row of the prepared mask, by index."""
        dev, n, m = self.dev, self.n, self.m
        T = len(src_subsets)
        src_all = torch.full((T, n), -1, dtype=torch.int32, device=dev)
        dst_all = torch.full((T, m), -1, dtype=torch.int32, device=dev)
        md = int(self.dsub.size)
        if md == 0 or n == 0:
            return src_all, dst_all
        subs = [_index_array(x) for x in src_subsets]
        lib = _lib.load()
        t0 = 0
        while t0 < T:
            t1, rows = t0, 0
            while t1 < T and (t1 == t0 or rows + subs[t1].size <= _BATCH_MAX_ROWS):
                rows += subs[t1].size
                t1 += 1
            if rows == 0:
                t0 = t1
                continue
            B = t1 - t0
            bests = [ss[np.flip(np.argsort(self.scores[ss].astype(np.float64)))] for ss in subs[t0:t1]]      # matcher.pyx:145-146 per subset
            sizes = [ss.size for ss in subs[t0:t1]]
            off = np.zeros((B + 1,), np.int64)
            off[1:] = np.cumsum(sizes)
            local = np.concatenate([np.arange(k, dtype=np.int64) for k in sizes])
            bid = np.repeat(np.arange(B, dtype=np.int64), sizes)
            with torch.cuda.device(dev):
                d = torch.from_numpy(np.concatenate([np.concatenate(subs[t0:t1]), np.concatenate(bests), off, local, bid])).to(dev)
                ssub_t, best_t = d[:rows], d[rows:2 * rows]
                off_t, order_t, bid_t = d[2 * rows:2 * rows + B + 1], d[2 * rows + B + 1:3 * rows + B + 1], d[3 * rows + B + 1:]
                sm = torch.empty((rows,), dtype=torch.int32, device=dev)
                dm = torch.empty((B, md), dtype=torch.int32, device=dev)
                status = torch.zeros((1,), dtype=torch.int32, device=dev)
                stag0 = torch.zeros((rows,), dtype=torch.int32, device=dev)
                ws = _lib.workspace(lib.d3d_score_match_batched_workspace_bytes(rows, md, B), dev)
                # P[k, j] = cols[subset row k, j] where ok[k-th best source, j], +inf elsewhere -- read through the two index arrays
                rc = lib.d3d_score_match_batched(_lib.ptr(self.cols), _lib.ptr(ssub_t), _lib.ptr(self._ok_u8), _lib.ptr(best_t),
                                                 _lib.ptr(off_t), B, rows, md, _lib.ptr(stag0), _lib.ptr(self._dtag0),
                                                 _lib.ptr(self._dthr), _lib.ptr(order_t), _lib.ptr(sm), _lib.ptr(dm), _lib.ptr(status),
                                                 _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
                _lib.check(rc, "score_match_batched")
                neg = torch.full((), -1, dtype=torch.int32, device=dev)
                # slot k of problem b matched destination slot sm: source best[k] <-> destination dsub[sm]
                src_all[t0:t1][bid_t, best_t] = torch.where(sm >= 0, self.dsub_t[sm.clamp_min(0).long()].to(torch.int32), neg)
                who = best_t[(dm.clamp_min(0).long() + off_t[:B, None]).clamp_max(rows - 1)].to(torch.int32)
                dst_all[t0:t1, self.dsub_t] = torch.where(dm >= 0, who, neg)
            t0 = t1
        return src_all, dst_all


class ScoreMatcher:
    """array-level ScoreMatcher (matcher.pyx:138-162): prepare_boxes, match on subsets, query_* -- same call sequence as the
    reference's evaluator uses (benchmarks.pyx:188-238).  The default reproduces the reference's results: its pairing of the
    k-th best source with the k-th subset row's distance order (matcher.pyx:155-158);
    `reference_compat=False` pairs every source with its own nearest destinations (what the class docstring describes)."""

    def __init__(self, reference_compat=True):
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
        else:      # (boxes outside the subsets carry the tag -1 and take no part)
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
