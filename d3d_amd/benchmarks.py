"""d3d_amd.benchmarks -- DetectionEvaluator.calc_stats of the reference (d3d/benchmarks.pyx:60-300) on arrays.

Boxes are [n,9] float32 rows (label, score, x, y, z, lx, ly, lz, yaw) -- Target3DArray.to_numpy's layout
(d3d/abstraction.pyx:263-272); the container classes of d3d.abstraction and the dataset-specific class enums are outside
this library's scope, classes are plain integers here.  The pairwise distance matrix (d3d_match_distance) and the
score-ordered association (d3d_score_match) run on the GPU; the per-threshold counts and means over the n + m matched indices
are numpy.  Two associations:
* the default, `reference_compat=True`: the reference's own, result for result -- one association PER score threshold
  (benchmarks.pyx:218-238) with ScoreMatcher.match's literal pairing of the two orders (matcher.pyx:142-162,
  d3d_amd.tracking.matcher.score_match_reference_compat);
* `reference_compat=False`: every detection walks its own nearest ground truths -- then ONE association serves all thresholds
  (a box's choice only depends on the boxes of higher score), 40 times less work.  Equal to the first wherever no detection
  has two acceptable ground truths.
"""
import numpy as np
import torch

from .tracking.matcher import DistanceTypes, ReferenceAssociation, prepare_boxes, score_match
from .utils import Dict


class DetectionEvaluator:
    """Benchmark for object detection; targets association is done by score sorting (benchmarks.pyx:84-149)."""

    def __init__(self, classes, min_overlaps, pr_sample_count=40, min_score=0, pr_sample_scale="log10", reference_compat=True):
        self.reference_compat = bool(reference_compat)
        classes = list(classes) if isinstance(classes, (list, tuple)) else [classes]
        assert len(classes) > 0
        self._classes = [int(getattr(c, "value", c)) for c in classes]
        if isinstance(min_overlaps, (list, tuple)):
            self._max_distance = {c: 1 - v for c, v in zip(self._classes, min_overlaps)}            # :114-115
        elif isinstance(min_overlaps, (int, float)):
            self._max_distance = {c: 1 - min_overlaps for c in self._classes}
        else:
            raise ValueError("min_overlaps should be a list or a single value")
        self._pr_nsamples = int(pr_sample_count)
        self._min_score = float(min_score)
        if pr_sample_scale == "lin":                                                                 # :125-134
            thresholds = np.linspace(min_score, 1, pr_sample_count, endpoint=False, dtype=np.float32)
        elif pr_sample_scale.startswith("log"):
            logstart, logend = 1, int(pr_sample_scale[3:] or "10")
            thresholds = np.geomspace(logstart, logend, pr_sample_count + 1, dtype=np.float32)
            thresholds = (thresholds - logstart) * (1 - min_score) / (logend - logstart)
            thresholds = (1 - thresholds)[:0:-1]
        else:
            raise ValueError("Unrecognized PR sample type")
        self._pr_thresholds = np.asarray(thresholds, dtype=np.float32)

    @property
    def score_thresholds(self):
        return self._pr_thresholds

    def calc_stats(self, gt_boxes, dt_boxes):
        """-> Dict(ngt{c}, ndt{c}[T], tp, fp, fn, acc_iou{c}[T], acc_angular, acc_dist, acc_box, acc_var) as
        DetectionEvalStats (benchmarks.pyx:60-82, 178-283); both box sets must be in the same frame"""
        gt = np.ascontiguousarray(gt_boxes, dtype=np.float32).reshape(-1, 9)
        dt = np.ascontiguousarray(dt_boxes, dtype=np.float32).reshape(-1, 9)
        T, classes = self._pr_nsamples, self._classes
        thr = self._pr_thresholds
        gt_tag, dt_tag = gt[:, 0].astype(np.int64), dt[:, 0].astype(np.int64)
        dt_score = dt[:, 1]
        out = Dict(ngt={}, ndt={}, tp={}, fp={}, fn={}, acc_iou={}, acc_angular={}, acc_dist={}, acc_box={}, acc_var={})
        if self.reference_compat:
            return self._calc_stats_per_threshold(gt, dt, out)
        if len(gt) and len(dt):
            cache = prepare_boxes(dt, gt, DistanceTypes.RIoU)                                        # :188-189
            sm, dm = score_match(cache, dt_score, dt_tag, gt_tag, self._max_distance)
            dm = dm.cpu().numpy().astype(np.int64)
            sm = sm.cpu().numpy().astype(np.int64)
            matched = dm >= 0
            iou = np.zeros((len(gt),), np.float32)
            iou[matched] = (1 - cache[dm[matched], np.nonzero(matched)[0]]).cpu().numpy()            # :243
        else:
            sm, dm = np.full((len(dt),), -1, np.int64), np.full((len(gt),), -1, np.int64)
            matched = dm >= 0
            iou = np.zeros((len(gt),), np.float32)
        partner = np.where(matched, dm, 0)
        # a ground-truth box is a true positive at threshold t iff its detection is selected there (score >= t)  (:220-238)
        gscore = np.where(matched, dt_score[partner] if len(dt) else 0.0, -np.inf)
        dist = np.linalg.norm(gt[:, 2:5] - dt[partner, 2:5], axis=1) if len(dt) else np.zeros(len(gt))          # :244
        box = np.linalg.norm(gt[:, 5:8] - dt[partner, 5:8], axis=1) if len(dt) else np.zeros(len(gt))           # :245
        dyaw = (gt[:, 8] - dt[partner, 8]) if len(dt) else np.zeros(len(gt))
        ang = np.abs((dyaw + np.pi) % (2 * np.pi) - np.pi) / np.pi                                     # quatdiff of two yaw rotations / pi (:247-248)
        dmatched = sm >= 0
        vals = np.stack([iou, ang, dist, box]).astype(np.float64)                                      # [4, m]

        def at_least(scores):
            """#(scores >= t) for every threshold t: one sort instead of a [len, T] comparison (:224-225: `score < thres` is
            skipped, so a NaN score is selected at EVERY threshold there -- counted apart, np.sort files NaNs last)"""
            nan = int(np.isnan(scores).sum())
            srt = np.sort(scores[~np.isnan(scores)])
            return (len(srt) - np.searchsorted(srt, thr, side="left") + nan).astype(np.int64)
        for c in classes:
            g, d = gt_tag == c, dt_tag == c
            out.ngt[c] = int(g.sum())
            out.ndt[c] = at_least(dt_score[d]).tolist()
            out.fp[c] = at_least(dt_score[d & ~dmatched]).tolist()
            # the true positives of threshold t are the matched boxes with gscore >= t: sorted by that score, every threshold is
            # a prefix -- counts by searchsorted, the sums of the accuracy terms by one cumulative sum (float64, rounded once)
            gs = gscore[g]
            o = np.argsort(-gs, kind="stable")
            gs_sorted = gs[o]
            tp = np.searchsorted(-gs_sorted, -thr, side="right").astype(np.int64)                      # #(gscore >= t)
            out.tp[c] = tp.tolist()
            out.fn[c] = (out.ngt[c] - tp).tolist()
            csum = np.concatenate([np.zeros((4, 1)), np.cumsum(vals[:, g][:, o], axis=1)], axis=1)      # [4, mc + 1]
            with np.errstate(invalid="ignore", divide="ignore"):
                means = np.where(tp[None, :] > 0, csum[:, tp] / tp[None, :], np.nan).astype(np.float32)
            out.acc_iou[c], out.acc_angular[c] = means[0].tolist(), means[1].tolist()
            out.acc_dist[c], out.acc_box[c] = means[2].tolist(), means[3].tolist()
            # no variances travel in the [n,9] arrays: orientation_var = 0 -> -inf per match (:250-258), NaN without one
            out.acc_var[c] = np.where(tp > 0, -np.inf, np.nan).astype(np.float32).tolist()
        return out


    def _calc_stats_per_threshold(self, gt, dt, out):
        """benchmarks.pyx:188-283 as written: select the detections of a threshold, associate (the literal pairing), count"""
        T, classes, thr = self._pr_nsamples, self._classes, self._pr_thresholds
        gt_tag, dt_tag = gt[:, 0].astype(np.int64), dt[:, 0].astype(np.int64)
        dt_score = dt[:, 1]
        gt_idx = np.nonzero(np.isin(gt_tag, classes))[0]                                              # :205-212
        dt_in = np.isin(dt_tag, classes)
        for c in classes:
            out.ngt[c] = int((gt_tag == c).sum())
            for k in ("ndt", "tp", "fp", "fn"):
                out[k][c] = [0] * T
            for k in ("acc_iou", "acc_angular", "acc_dist", "acc_box", "acc_var"):
                out[k][c] = [float("nan")] * T
        cache = prepare_boxes(dt, gt, DistanceTypes.RIoU) if len(gt) and len(dt) else None             # :188-189
        # the 40 associations share what does not depend on the threshold (the ground truths' columns of the cache, which pairs are
        # acceptable) and go to the device as batched calls (ReferenceAssociation.match_many: one for a frame, a few for config 4's
        # 20 k x 5 k), nothing read back in between; the results are fetched together
        assoc = gt_idx_t = None
        if cache is not None and len(gt_idx):
            assoc = ReferenceAssociation(cache, dt_score, dt_tag, gt_tag, self._max_distance, gt_idx)
            gt_idx_t = torch.from_numpy(gt_idx).to(cache.device)
        sel = dt_in[None, :] & ~(dt_score[None, :] < thr[:, None])                                   # [T, n]: :219-228 (`score < thres`: skip)
        dt_idxs = [np.nonzero(sel[t])[0] for t in range(T)]
        md = len(gt_idx)
        sm_all = np.full((T, len(dt)), -1, np.int32)
        dm_g = np.full((T, md), -1, np.int32)                                                       # dst_match over the ground truths taking part
        iou_all = np.zeros((T, md), np.float32)
        if assoc is not None:
            sm_t, dm_t = assoc.match_many(dt_idxs)                                                     # :231-232, all thresholds
            dmg_t = dm_t.index_select(1, gt_idx_t)
            iou_t = 1 - cache[dmg_t.long().clamp_min(0), gt_idx_t[None, :]]                            # :243 (entries of the unmatched: unused)
            sm_all, dm_g, iou_all = sm_t.cpu().numpy(), dmg_t.cpu().numpy(), iou_t.cpu().numpy()
        # the counts and means of :236-283 for all thresholds at once: the K matched (threshold, ground truth) pairs as flat arrays,
        # per-threshold counts and float64 sums by bincount (the arithmetic of one pair is unchanged: float32 terms, their sum in
        # float64, one division, rounded to float32)
        tt, jj = np.nonzero(dm_g >= 0)
        terms, gcls = {}, np.zeros((0,), np.int64)
        if len(tt):
            gi, d_of = gt_idx[jj], dm_g[tt, jj]
            ga, da = gt[gi], dt[d_of]                                                                  # [K, 9]
            dp, db = ga[:, 2:5] - da[:, 2:5], ga[:, 5:8] - da[:, 5:8]
            dyaw = ga[:, 8] - da[:, 8]
            terms = dict(acc_iou=iou_all[tt, jj],
                         acc_angular=np.abs((dyaw + np.pi) % (2 * np.pi) - np.pi) / np.pi,              # :247-248
                         # (np.linalg.norm(., axis=1) spelled out -- sqrt(add.reduce(x * x)) over three terms, the same bits)
                         acc_dist=np.sqrt((dp[:, 0] * dp[:, 0] + dp[:, 1] * dp[:, 1]) + dp[:, 2] * dp[:, 2]),   # :244
                         acc_box=np.sqrt((db[:, 0] * db[:, 0] + db[:, 1] * db[:, 1]) + db[:, 2] * db[:, 2]))    # :245
            gcls = gt_tag[gi]
        unmatched = sm_all < 0
        for c in classes:
            dc = sel & (dt_tag == c)[None, :]
            of_c = gcls == c
            tc = tt[of_c]
            tp = np.bincount(tc, minlength=T)
            out.ndt[c] = dc.sum(1).tolist()
            out.tp[c] = tp.tolist()
            out.fn[c] = (out.ngt[c] - tp).tolist()                                                     # :236-240
            out.fp[c] = (dc & unmatched).sum(1).tolist()                                               # :262-265
            for name, v in terms.items():                                                              # :150-174 (sum / count, fp32)
                ssum = np.bincount(tc, weights=v[of_c].astype(np.float64), minlength=T)
                with np.errstate(invalid="ignore", divide="ignore"):
                    out[name][c] = np.where(tp > 0, (ssum / tp).astype(np.float32), np.float32(np.nan)).tolist()
            # no variances travel in the [n,9] arrays: orientation_var = 0 -> -inf per match (:250-258), NaN without one
            out.acc_var[c] = np.where(tp > 0, -np.inf, np.nan).tolist()
        return out


__all__ = ["DetectionEvaluator"]
