"""CPU: the C box oracle against the reference's known-answer tests (test/test_box.py, test_benchmark.py)
and an exact-rational clipper.  (dgal is not vendored -> value-level pin; DESIGN.md.)"""
import numpy as np
import pytest

import box_cases as bc
import oracle
from exact_clip import iou_exact


def test_iou_aa_boxes():          # test_box.py:12-37
    assert np.allclose(oracle.box2d_iou(bc.AA_B1, bc.AA_B2, "box"), bc.AA_EXPECTED, atol=bc.eps)
    assert np.allclose(oracle.box2d_iou(bc.AA_B1, bc.AA_B2, "rbox"), bc.AA_EXPECTED, atol=4 * bc.eps)


def test_iou_rotated_boxes():     # test_box.py:39-72
    assert np.allclose(oracle.box2d_iou(bc.ROT_B1, bc.ROT_B2, "box"), bc.ROT_BOX_EXPECTED, atol=2 * bc.eps)
    assert np.allclose(oracle.box2d_iou(bc.ROT_B1, bc.ROT_B2, "rbox"), bc.ROT_RBOX_EXPECTED, atol=4 * bc.eps)


def test_iou_apart_boxes():       # test_box.py:74-100
    ious = oracle.box2d_iou(bc.APART_BOX, bc.APART_BOX, "box")
    assert np.allclose(ious - np.eye(4), 0, atol=1e-6)
    ious = oracle.box2d_iou(bc.APART_RBOX, bc.APART_RBOX, "rbox")
    assert np.allclose(ious - np.eye(5), 0, atol=1e-6)


def test_nms():                   # test_box.py:102-123
    for m in ["box", "rbox"]:
        assert np.array_equal(oracle.box2d_nms(bc.NMS_BOXES, bc.NMS_SCORES, iou_method=m), bc.NMS_EXPECTED)


def test_iou_large_array_range():  # test_box.py:125-138
    b, _ = bc.random_boxes_like_reference(500, 0)
    for m in ["box", "rbox"]:
        r = oracle.box2d_iou(b, b, m)
        assert np.all(r >= -bc.eps) and np.all(r <= 1 + bc.eps)
        assert np.allclose(np.diag(r), 1, atol=1e-5)


def test_nms_large_array():       # test_box.py:140-155
    b, s = bc.random_boxes_like_reference(500, 1)
    for m in ["box", "rbox"]:
        for thr in [0, 0.2, 0.5, 0.8, 0.99]:
            keep = oracle.box2d_nms(b, s, iou_method=m, iou_threshold=0.3, score_threshold=thr)
            assert np.all(keep[s <= thr] == False)  # noqa: E712


def test_softnms_all_kept():      # test_box.py:157-179
    for m in ["box", "rbox"]:
        for sup in ["linear", "gaussian"]:
            keep = oracle.box2d_nms(bc.SOFT_BOXES, bc.NMS_SCORES, iou_method=m, supression_method=sup)
            assert np.all(keep)


def test_box3dr_iou_evaluator_vectors():   # test_benchmark.py:31-39, 45-71
    v = oracle.iou3d(bc.EVAL_DT, bc.EVAL_GT, "rbox")[0, 0]
    assert v > 0.1 and abs(v - bc.EVAL_IOU) < 1e-4
    self_iou = oracle.iou3d(bc.EVAL_DT, bc.EVAL_DT, "rbox")[0, 0]
    assert np.isclose(self_iou, 1)


def test_against_exact_rational_clipper():
    rng = np.random.default_rng(3)
    n = 40
    b1 = np.stack([(rng.random(n) - .5) * 10, (rng.random(n) - .5) * 10, rng.random(n) * 5 + .1, rng.random(n) * 5 + .1,
                   (rng.random(n) - .5) * 10], 1)
    b2 = np.stack([(rng.random(n) - .5) * 10, (rng.random(n) - .5) * 10, rng.random(n) * 5 + .1, rng.random(n) * 5 + .1,
                   (rng.random(n) - .5) * 10], 1)
    got = oracle.iou2d_forward(b1, b2, "rbox")
    for i in range(n):
        for j in range(0, n, 3):
            assert abs(got[i, j] - iou_exact(b1[i], b2[j])) < 1e-12


def test_threads_do_not_change_results():
    b, _ = bc.random_boxes_like_reference(300, 4)
    a = oracle.iou2d_forward(b.astype(np.float64), b.astype(np.float64), "rbox", nthreads=1)
    c = oracle.iou2d_forward(b.astype(np.float64), b.astype(np.float64), "rbox", nthreads=4)
    assert np.array_equal(a, c)


def test_crop_reference_case():   # test_box.py:191-205
    rng = np.random.default_rng(40)
    cloud = (rng.random((100, 2)) * 2 - 1).astype(np.float32)
    boxes = np.array([[0, 0, 1, 1, 0], [0, 0, 1, 1, bc.d90]], np.float32)
    res = oracle.crop_2dr(cloud, boxes)
    a = np.abs(cloud)
    assert np.array_equal(np.where(res[0])[0], np.where(np.all(a < 0.5, 1))[0])
    assert np.array_equal(np.where(res[1])[0], np.where(np.abs(a[:, 0] + a[:, 1]) < bc.sq2 / 2)[0])


def test_candidate_restricted_forms_equal_the_plain_loops():
    """the two accelerated oracle forms the full-size GPU tests rely on: IoU over the pairs of an AABB sweep reproduces the
    dense matrix (zero elsewhere), hard NMS over those pairs reproduces nms.cpp's greedy loop"""
    from d3d_amd import synth
    b, s = synth.boxes2d_sparse(4000, 11)
    b2, _ = synth.boxes2d_dense(300, 12)
    b2[:, :2] = b2[:, :2] * 30 + 300
    b2[:, 2:4] *= 8
    for x, y in ((b, None), (b[:700], b2)):
        pi, pj = oracle.aabb_candidate_pairs(x, y)
        yy = x if y is None else y
        for method in ("rbox", "box"):
            full = oracle.box2d_iou(x, yy, method, nthreads=4)
            m = np.zeros_like(full)
            m[pi, pj] = oracle.iou2d_pairs(x, yy, pi, pj, method)
            assert np.array_equal(m, full) and (full > 0).sum() > 300
    for method, thr, sthr in (("rbox", 0.5, 0.0), ("box", 0.0, 0.3), ("rbox", 0.25, 0.1)):
        assert np.array_equal(oracle.box2d_nms(b, s, iou_method=method, iou_threshold=thr, score_threshold=sthr),
                              oracle.box2d_nms_hard_candidates(b, s, method, thr, sthr))
    d, ds = synth.boxes2d_dense(1500, 13)                     # the heavily overlapping distribution
    assert np.array_equal(oracle.box2d_nms(d, ds, iou_method="rbox", iou_threshold=0.3),
                          oracle.box2d_nms_hard_candidates(d, ds, "rbox", 0.3))


def _loss_cases():
    """pairs for GIoU / DIoU: random overlapping and disjoint ones plus the degenerate configurations of the hull logic"""
    rng = np.random.default_rng(8)
    n = 24
    b1 = np.stack([(rng.random(n) - .5) * 8, (rng.random(n) - .5) * 8, rng.random(n) * 5 + .1, rng.random(n) * 5 + .1,
                   (rng.random(n) - .5) * 10], 1)
    b2 = np.stack([(rng.random(n) - .5) * 8, (rng.random(n) - .5) * 8, rng.random(n) * 5 + .1, rng.random(n) * 5 + .1,
                   (rng.random(n) - .5) * 10], 1)
    special1 = np.array([[0, 0, 2, 2, 0], [0, 0, 2, 2, 0], [0, 0, 2, 2, 0], [0, 0, 4, 4, 0], [0, 0, 2, 2, 0], [0, 0, 2, 2, 0],
                         [0, 0, 2, 2, 0.5], [0, 0, 2, 1, 0], [1, 1, 2, 2, np.pi / 4]], np.float64)
    special2 = np.array([[0, 0, 2, 2, 0],          # identical
                         [2, 0, 2, 2, 0],          # sharing a whole edge
                         [2, 2, 2, 2, 0],          # touching in one corner
                         [0.5, 0.25, 1, 1, 0],     # contained
                         [1, 0, 2, 2, 0],          # two collinear sides each, overlapping
                         [5, 0, 2, 2, 0],          # apart, collinear sides
                         [0, 0, 2, 2, 0.5],        # identical, rotated
                         [0, 1.5, 2, 2, 0],        # collinear vertical sides, different sizes
                         [1, 1, 2, 2, np.pi / 4 + np.pi / 2]], np.float64)     # the same square under another angle
    return np.concatenate([b1, special1]), np.concatenate([b2, special2])


def test_giou_diou_against_exact_rational():
    from exact_clip import loss_iou_exact
    b1, b2 = _loss_cases()
    for kind in ("grbox", "drbox"):
        got = oracle.loss_iou2dr(b1, b2, kind)
        for i in range(len(b1)):
            for j in list(range(0, len(b2), 5)) + [i]:
                assert abs(got[i, j] - loss_iou_exact(b1[i], b2[j], kind)) < 1e-12, (kind, i, j)
    g = oracle.loss_iou2dr(b1, b2, "grbox")
    assert np.all(g <= 1 + 1e-12) and np.all(g > -1) and abs(g[len(b1) - 9, len(b2) - 9] - 1) < 1e-15     # identical boxes -> 1
    assert np.all(oracle.loss_iou2dr(b1, b2, "drbox") <= 1 + 1e-12)
    # the public wrapper routes the two methods (reference box/__init__.py:207-216)
    assert np.array_equal(oracle.box2d_iou(b1, b2, "grbox"), g)
    # a rectangle without area gives 0, never NaN
    z = np.array([[0, 0, 0, 2, 0.3]])
    assert oracle.loss_iou2dr(z, b2[:3], "grbox").tolist() == [[0, 0, 0]] and oracle.loss_iou2dr(b1[:2], z, "drbox").tolist() == [[0], [0]]


def test_flags_describe_the_intersection_polygon():
    """xflags name where every vertex of the intersection comes from: rebuilding the polygon from them gives the
    intersection area of the IoU; hull flags list the hull's corners; far is the farthest corner pair"""
    from exact_clip import corners
    rng = np.random.default_rng(9)
    n = 30
    b1 = np.stack([(rng.random(n) - .5) * 6, (rng.random(n) - .5) * 6, rng.random(n) * 4 + .5, rng.random(n) * 4 + .5,
                   (rng.random(n) - .5) * 6], 1)
    f = oracle.iou2dr_flags(b1, b1[::-1].copy())
    b2 = b1[::-1]
    iou = oracle.box2d_iou(b1, b2, "rbox")

    def line_x(p, q, r, s):
        d = (q[0] - p[0]) * (s[1] - r[1]) - (q[1] - p[1]) * (s[0] - r[0])
        t = ((r[0] - p[0]) * (s[1] - r[1]) - (r[1] - p[1]) * (s[0] - r[0])) / d
        return (p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1]))
    for i in range(n):
        for j in range(n):
            c1, c2 = corners(*b1[i]), corners(*b2[j])
            k = int(f.nx[i, j])
            assert (k == 0) == (iou[i, j] == 0) or k == 0
            poly = []
            for fl in f.xflags[i, j, :k]:
                if fl < 0x10:
                    poly.append(c1[fl])
                elif fl < 0x20:
                    poly.append(c2[fl & 3])
                else:
                    e1, e2 = (fl >> 2) & 3, fl & 3
                    poly.append(line_x(c1[e1], c1[(e1 + 1) & 3], c2[e2], c2[(e2 + 1) & 3]))
            assert np.all(f.xflags[i, j, k:] == 0xff)
            a = 0.5 * sum(poly[t][0] * poly[(t + 1) % k][1] - poly[(t + 1) % k][0] * poly[t][1] for t in range(k)) if k else 0.0
            a1, a2 = b1[i, 2] * b1[i, 3], b2[j, 2] * b2[j, 3]
            inter = iou[i, j] * (a1 + a2) / (1 + iou[i, j])
            assert abs(a - inter) < 1e-9, (i, j, k, a, inter)
            pts = np.array(c1 + c2)
            hull = f.mflags[i, j, :f.nm[i, j]]
            assert 3 <= len(hull) <= 8 and len(set(hull.tolist())) == len(hull)
            d = np.linalg.norm(pts[:, None] - pts[None], axis=2)
            assert np.isclose(d[f.far[i, j, 0], f.far[i, j, 1]], d.max()) and f.far[i, j, 0] < f.far[i, j, 1]
            hull_pts = {tuple(pts[h]) for h in hull}                     # (identical boxes: a hull vertex stands for both copies)
            assert all(tuple(pts[v]) in hull_pts for v in f.far[i, j])   # the diameter is attained at hull vertices


def test_pdist_sign_and_nearest_feature():
    rng = np.random.default_rng(10)
    pts = (rng.random((400, 2)) - 0.5) * 8
    boxes = np.array([[0, 0, 2, 2, 0], [0.5, -0.3, 3, 1, 0.7], [-1, 1, 1, 4, -2.0]])
    d, e = oracle.pdist2dr(pts, boxes)
    inside = oracle.crop_2dr(pts, boxes)
    assert np.array_equal(d > 0, inside & (d != 0))                       # positive exactly inside (box/__init__.py:370-381)
    # axis-aligned square: closed form
    ax, ay = np.abs(pts[:, 0]), np.abs(pts[:, 1])
    inside0 = (ax <= 1) & (ay <= 1)
    ref = np.where(inside0, np.minimum(1 - ax, 1 - ay), -np.hypot(np.maximum(ax - 1, 0), np.maximum(ay - 1, 0)))
    assert np.allclose(d[0], ref, atol=1e-12)
    assert np.all(e[0][inside0] < 4) and np.all(e[0][(ax > 1) & (ay > 1)] >= 4)      # corner regions name a corner


def test_evaluator_restatement_reproduces_the_reference_test():
    """test/test_benchmark.py:10-84 on arrays (classes Car = 1, Van = 2, Pedestrian = 3)"""
    thr = np.array([0.05, 0.75, 0.9], np.float32)
    dt = np.array([[1, 0.8, 0, 0, 0, 2, 2, 2, 0], [2, 0.7, 1, 1, 1, 2, 2, 2, 0], [3, 0.8, -1, -1, -1, 2, 2, 2, 0]], np.float32)
    r = oracle.calc_stats(dt, dt, [1, 2], {1: 0.9, 2: 0.8}, thr)
    for c in (1, 2):
        assert r.ngt[c] == 1 and r.ndt[c][0] == 1 and r.ndt[c][-1] == 0 and r.tp[c][0] == 1 and r.tp[c][-1] == 0
        assert r.fp[c][0] == 0 and r.fp[c][-1] == 0 and r.fn[c][0] == 0 and r.fn[c][-1] == 1
        assert np.isclose(r.acc_iou[c][0], 1) and np.isnan(r.acc_iou[c][-1]) and np.isclose(r.acc_dist[c][0], 0)
    gt = np.array([[2, 0, 0, 0, 0, 2.1, 2.1, 2.1, 0.01], [1, 0, -1, 1, 0, 2.1, 2.1, 2.1, 0.01],
                   [3, 0, 1, -1, 0, 2.1, 2.1, 2.1, 0.01]], np.float32)
    r = oracle.calc_stats(gt, dt, [1, 2], {1: 0.9, 2: 0.8}, thr)
    assert r.tp[1][0] == 1 and r.fp[1][0] == 0 and r.fn[1][0] == 0 and r.acc_iou[1][0] > 0.1 and r.acc_dist[1][0] > 1
    assert r.acc_angular[1][0] > 0 and r.acc_box[1][0] > 0 and abs(r.acc_iou[1][0] - bc.EVAL_IOU) < 1e-4
    assert r.tp[2][0] == 0 and r.fp[2][0] == 1 and r.fn[2][0] == 1 and np.isnan(r.acc_iou[2][0])


def test_literal_association_order_of_the_reference_and_where_it_matters():
    """matcher.pyx:155-158 walks, for the k-th best source, the distance order of the k-th ROW of the subset.  oracle.score_match
    restates both: literal=True (that loop as written) and the default (every source its own row).  They agree on the
    reference's own test scene (test/test_benchmark.py: one acceptable ground truth per detection) and differ on a crowded
    one, where the literal loop gives the best detection the ground truth nearest to ANOTHER detection"""
    dt = np.array([[1, 0.8, 0, 0, 0, 2, 2, 2, 0], [2, 0.7, 1, 1, 1, 2, 2, 2, 0], [3, 0.8, -1, -1, -1, 2, 2, 2, 0]], np.float32)
    gt = np.array([[2, 0, 0, 0, 0, 2.1, 2.1, 2.1, 0.01], [1, 0, -1, 1, 0, 2.1, 2.1, 2.1, 0.01],
                   [3, 0, 1, -1, 0, 2.1, 2.1, 2.1, 0.01]], np.float32)
    cache = oracle.prepare_boxes(dt, gt)
    for thr in ({1: 0.9, 2: 0.8}, {1: 0.9, 2: 0.8, 3: 0.9}):
        src = [i for i in range(3) if int(dt[i, 0]) in thr]
        dst = [j for j in range(3) if int(gt[j, 0]) in thr]
        assert oracle.score_match(cache, dt, gt, src, dst, thr, literal=True) == oracle.score_match(cache, dt, gt, src, dst, thr)
    g9 = np.array([[1, 0, 0.0, 0.0, 0, 4, 2, 2, 0], [1, 0, 1.2, 0.0, 0, 4, 2, 2, 0]], np.float32)
    d9 = np.array([[1, 0.6, -0.4, 0.0, 0, 4, 2, 2, 0], [1, 0.9, 0.9, 0.0, 0, 4, 2, 2, 0]], np.float32)
    c = oracle.prepare_boxes(d9, g9)
    lit, _ = oracle.score_match(c, d9, g9, [0, 1], [0, 1], {1: 0.8}, literal=True)
    own, _ = oracle.score_match(c, d9, g9, [0, 1], [0, 1], {1: 0.8})
    assert own == {1: 1, 0: 0} and lit == {1: 0, 0: 1}
    # a subset that IS in score order: the loop counter then names the source's own row, both forms agree
    assert oracle.score_match(c, d9, g9, [1, 0], [0, 1], {1: 0.8}, literal=True)[0] == own


def test_row_wise_association_equals_the_pair_loop():
    from d3d_amd import synth
    pred, gt = synth.boxes3d_eval(60, 4, 5)
    rng = np.random.default_rng(6)
    dt = np.concatenate([rng.integers(1, 4, (len(pred), 1)), rng.random((len(pred), 1)), pred], 1).astype(np.float32)
    g9 = np.concatenate([rng.integers(1, 4, (len(gt), 1)), np.zeros((len(gt), 1)), gt], 1).astype(np.float32)
    cache = oracle.prepare_boxes(dt, g9)
    thr = {1: 0.7, 2: 0.5}
    sa, da = oracle.score_match(cache, dt, g9, [i for i in range(len(dt)) if int(dt[i, 0]) in thr],
                                [j for j in range(len(g9)) if int(g9[j, 0]) in thr], thr)
    sm, dm = oracle.score_match_rows(cache, dt, g9, thr)
    assert {i: int(j) for i, j in enumerate(sm) if j >= 0} == sa and {j: int(i) for j, i in enumerate(dm) if i >= 0} == da
    assert 10 < len(sa) < len(g9)


@pytest.mark.parametrize("sup,thr,sthr,param", [("linear", 0.1, 0.3, 1.0), ("gaussian", 0.0, 0.2, 0.5), ("linear", 0.3, 0.0, 2.0)])
def test_soft_nms_over_candidate_pairs_equals_the_literal_loop(sup, thr, sthr, param):
    """oracle.box2d_nms_soft_candidates (rescaling only the neighbours whose bounding boxes touch) == the literal restatement
    of nms.cpp:60-94: what lets a GPU test check soft-NMS beyond 65 536 boxes"""
    from d3d_amd import synth
    for n, gen in ((300, synth.boxes2d_dense), (1200, synth.boxes2d_sparse)):
        b, s = gen(n, 3)
        for method in ("rbox", "box"):
            a = oracle.box2d_nms(b, s, iou_method=method, supression_method=sup, iou_threshold=thr, score_threshold=sthr,
                                 supression_param=param)
            c = oracle.box2d_nms_soft_candidates(b, s, method, sup, thr, sthr, param)
            assert np.array_equal(a, c)
