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
