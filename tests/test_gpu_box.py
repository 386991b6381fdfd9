"""GPU parity (through the C ABI): d3d_amd.box vs the reference's known answers and the CPU oracle.
Tolerances (north_star): IoU 1e-6 in fp64, 1e-3 in fp32; NMS keep masks bit-exact."""
import numpy as np
import pytest
import torch

import box_cases as bc
from call_opts import cur_opts, set_opts
import oracle

pytestmark = pytest.mark.gpu


def pytest_generate_tests(metafunc):
    # every NMS test runs four ways: as the operator picks (sets of up to 4096 boxes take the small-set path), on the general
    # path with either broad phase (the uniform grid, the sweep along x), and on the grid with its level kernels forced (they
    # run by themselves only on dense grids: clusters of detections)
    # (soft-NMS is one kernel of its own -- k_softnms -- and takes none of the broad phases: once is enough, 4 x 33 s saved)
    name = metafunc.function.__name__
    if "nms" in name and "soft" not in name and "nms_broad" in metafunc.fixturenames:
        metafunc.parametrize("nms_broad", ["auto", "grid", "sweep", "levels"], indirect=True)


@pytest.fixture(autouse=True)
def nms_broad(request, monkeypatch):
    from d3d_amd import _lib, box
    mode = getattr(request, "param", "auto")
    set_opts(nms_flags={"auto": 0, "grid": _lib.NMS_GENERAL, "sweep": _lib.NMS_BROAD_SWEEP, "levels": _lib.NMS_FORCE_LEVELS}[mode],
             poison=True)                               # every IoU / iou3d result buffer of this module starts as NaN
    yield mode


def T(a, cuda=True):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.cuda() if cuda else t


@pytest.mark.parametrize("cuda", [True, False])
def test_iou_aa_boxes(cuda):          # test_box.py:12-37
    from d3d_amd.box import box2d_iou
    exp = torch.from_numpy(bc.AA_EXPECTED)
    ious = box2d_iou(T(bc.AA_B1, cuda), T(bc.AA_B2, cuda), method="box")
    assert ious.is_cuda == cuda and ious.dtype == torch.float32
    assert torch.allclose(ious.cpu(), exp, atol=bc.eps)
    ious = box2d_iou(T(bc.AA_B1, cuda), T(bc.AA_B2, cuda), method="rbox")
    assert torch.allclose(ious.cpu(), exp, atol=4 * bc.eps)


def test_iou_rotated_boxes():         # test_box.py:39-72
    from d3d_amd.box import box2d_iou, iou2d
    for precise in [True, False]:
        ious = box2d_iou(T(bc.ROT_B1), T(bc.ROT_B2), method="box", precise=precise)
        assert torch.allclose(ious.cpu(), torch.from_numpy(bc.ROT_BOX_EXPECTED), atol=2 * bc.eps)
        ious = iou2d(T(bc.ROT_B1), T(bc.ROT_B2), method="rbox", precise=precise)
        assert torch.allclose(ious.cpu(), torch.from_numpy(bc.ROT_RBOX_EXPECTED), atol=4 * bc.eps)


def test_iou_apart_boxes():           # test_box.py:74-100
    from d3d_amd.box import box2d_iou
    for precise in [True, False]:
        ious = box2d_iou(T(bc.APART_BOX), T(bc.APART_BOX), method="box", precise=precise)
        assert np.allclose(ious.cpu().numpy() - np.eye(4), 0, atol=1e-6)
        ious = box2d_iou(T(bc.APART_RBOX), T(bc.APART_RBOX), method="rbox", precise=precise)
        assert np.allclose(ious.cpu().numpy() - np.eye(5), 0, atol=1e-6)


def test_numpy_ingress_and_errors():
    from d3d_amd.box import box2d_iou, box2d_nms
    r = box2d_iou(bc.AA_B1, bc.AA_B2, method="rbox")
    assert isinstance(r, np.ndarray) and r.dtype == np.float32
    with pytest.raises(ValueError):
        box2d_iou(T(bc.AA_B1)[:, :4], T(bc.AA_B2), method="box")
    with pytest.raises(ValueError):
        box2d_iou(T(bc.AA_B1)[0], T(bc.AA_B2), method="box")
    with pytest.raises(ValueError):
        box2d_nms(T(bc.NMS_BOXES), T(bc.NMS_SCORES)[:3])
    with pytest.raises(ValueError):
        box2d_nms(T(bc.NMS_BOXES), T(bc.NMS_SCORES), iou_method="grbox")
    assert box2d_nms(torch.zeros((0, 5)), torch.zeros((0,))).shape == (0,)
    assert box2d_iou(torch.zeros((0, 5)).cuda(), T(bc.AA_B2), method="rbox").shape == (0, 3)


def test_nms_known_answer():          # test_box.py:102-123
    from d3d_amd.box import box2d_nms, nms
    for m in ["box", "rbox"]:
        for cuda in [True, False]:
            mask = box2d_nms(T(bc.NMS_BOXES, cuda), T(bc.NMS_SCORES, cuda), iou_method=m)
            assert mask.dtype == torch.bool and mask.is_cuda == cuda
            assert np.array_equal(mask.cpu().numpy(), bc.NMS_EXPECTED)
    assert np.array_equal(nms(bc.NMS_BOXES, bc.NMS_SCORES, iou_method="rbox"), bc.NMS_EXPECTED)
    s2 = np.stack([bc.NMS_SCORES * 0.5, bc.NMS_SCORES], 1)      # [N,K] scores -> class max (box/__init__.py:253)
    assert np.array_equal(box2d_nms(T(bc.NMS_BOXES), T(s2)).cpu().numpy(), bc.NMS_EXPECTED)


def test_iou_large_array_and_nms_property():   # test_box.py:125-155
    from d3d_amd.box import box2d_iou, box2d_nms
    b, s = bc.random_boxes_like_reference(500, 0)
    for m in ["box", "rbox"]:
        r = box2d_iou(T(b), T(b), method=m)
        assert torch.all(r >= -bc.eps) and torch.all(r <= 1 + bc.eps)
        for thr in [0, 0.2, 0.5, 0.8, 0.99]:
            keep = box2d_nms(T(b), T(s), iou_method=m, iou_threshold=0.3, score_threshold=thr).cpu().numpy()
            assert np.all(keep[s <= thr] == False)  # noqa: E712
            assert np.array_equal(keep, oracle.box2d_nms(b, s, iou_method=m, iou_threshold=0.3, score_threshold=thr))


@pytest.mark.parametrize("method", ["box", "rbox"])
@pytest.mark.parametrize("gen", ["sparse", "dense"])
def test_iou_vs_oracle_fp64(method, gen):
    from d3d_amd import synth
    from d3d_amd.box import box2d_iou
    mk = synth.boxes2d_sparse if gen == "sparse" else synth.boxes2d_dense
    b1, _ = mk(700, 21)
    b2, _ = mk(333, 22)
    got = box2d_iou(T(b1), T(b2), method=method).cpu().numpy()
    exp = oracle.box2d_iou(b1, b2, method, nthreads=8)
    assert got.dtype == np.float64 and got.shape == (700, 333)
    assert np.max(np.abs(got - exp)) < 1e-6, np.max(np.abs(got - exp))   # north_star: 1e-6 in fp64
    assert np.max(np.abs(got - exp)) < 1e-9                               # what we actually achieve


@pytest.mark.parametrize("method", ["box", "rbox"])
def test_iou_vs_oracle_fp32(method):
    from d3d_amd import synth
    from d3d_amd.box import box2d_iou
    b1, _ = synth.boxes2d_dense(500, 23, np.float32)
    b2, _ = synth.boxes2d_dense(400, 24, np.float32)
    got = box2d_iou(T(b1), T(b2), method=method, precise=False).cpu().numpy()
    ref64 = oracle.box2d_iou(b1.astype(np.float64), b2.astype(np.float64), method, nthreads=8)
    assert got.dtype == np.float32
    assert np.max(np.abs(got - ref64)) < 1e-3                             # north_star: 1e-3 in fp32


def test_degenerate_boxes():
    from d3d_amd.box import box2d_iou
    b = np.array([[0, 0, 2, 2, 0.3], [0, 0, 2, 2, 0.3],       # identical -> 1
                  [0, 0, 0, 0, 0], [5, 5, 0, 3, 0.2],          # zero area -> 0, no NaN
                  [2, 0, 2, 2, 0],                             # shares an edge with box 5
                  [0, 0, 2, 2, 0], [0, 0, 1, 1, 0],            # contained: 1/4
                  [0, 0, 2, 2, np.pi / 2]], np.float64)        # same square rotated 90 deg
    got = box2d_iou(T(b), T(b), method="rbox").cpu().numpy()
    exp = oracle.box2d_iou(b, b, "rbox")
    assert np.all(np.isfinite(got))
    assert abs(got[0, 1] - 1) < 1e-12 and abs(got[5, 6] - 0.25) < 1e-12 and abs(got[4, 5]) < 1e-12
    assert abs(got[5, 7] - 1) < 1e-9
    assert np.allclose(got, exp, atol=1e-9)


@pytest.mark.parametrize("method", ["box", "rbox"])
@pytest.mark.parametrize("gen,n,thr", [("sparse", 5000, 0.5), ("dense", 1500, 0.3), ("sparse", 5000, 0.0),
                                        ("dense", 63, 0.1), ("dense", 64, 0.1), ("dense", 65, 0.1), ("dense", 1, 0.1)])
def test_nms_vs_oracle_bit_exact(method, gen, n, thr):
    from d3d_amd import synth
    from d3d_amd.box import box2d_nms
    mk = synth.boxes2d_sparse if gen == "sparse" else synth.boxes2d_dense
    b, s = mk(n, 31)
    assert len(np.unique(s)) == n          # no score ties -> the order is unambiguous
    for sthr in [0.0, 0.3]:
        keep = box2d_nms(T(b), T(s), iou_method=method, iou_threshold=thr, score_threshold=sthr).cpu().numpy()
        exp = oracle.box2d_nms(b, s, iou_method=method, iou_threshold=thr, score_threshold=sthr)
        assert np.array_equal(keep, exp), (np.sum(keep != exp), n)


@pytest.mark.parametrize("hook,value", [("force_dense", 1), ("cand_cap", 100), ("cand_cap", 3000)])
@pytest.mark.parametrize("gen,n,thr", [("sparse", 5000, 0.3), ("dense", 1500, 0.3)])
def test_nms_dense_path_hooks(monkeypatch, hook, value, gen, n, thr):
    """the gated dense path (bit matrix + single-workgroup sweep) gives the same keep set as the list path: forced,
    and reached through a candidate list that is too small (with and without the LDS batch spilling)"""
    from d3d_amd import synth
    from d3d_amd.box import box2d_nms
    mk = synth.boxes2d_sparse if gen == "sparse" else synth.boxes2d_dense
    b, s = mk(n, 77)
    exp = oracle.box2d_nms(b, s, iou_method="rbox", iou_threshold=thr, score_threshold=0.1)
    from d3d_amd import _lib, box
    base = cur_opts().nms_flags
    set_opts(nms_flags=base | (_lib.NMS_FORCE_DENSE if hook == "force_dense" else _lib.nms_cand_cap(value)))
    keep = box2d_nms(T(b), T(s), iou_method="rbox", iou_threshold=thr, score_threshold=0.1).cpu().numpy()
    assert np.array_equal(keep, exp)
    set_opts(nms_flags=base)
    keep = box2d_nms(T(b), T(s), iou_method="rbox", iou_threshold=thr, score_threshold=0.1).cpu().numpy()
    assert np.array_equal(keep, exp)


@pytest.mark.parametrize("scan", ["grid cells", "incoming lists"])
def test_nms_chained_scan_gives_up_and_hands_over(monkeypatch, scan):
    """the one-launch scans poll their predecessors' totals; a predecessor that never publishes (test hook: ticket 0
    withholds) makes every later workgroup give up after ~0.1 s, nothing that depends on the void prefix is stored, and the
    dense path recomputes the call: same keep mask.  Both scans: the grid's cell scan, the scan of the incoming-list sizes."""
    from d3d_amd import _lib, box, synth
    from d3d_amd.box import box2d_nms
    # (the grid's cell scan only has a second workgroup to wait above 1024 cells: 20 k boxes)
    b, s = synth.boxes2d_sparse(20000 if scan == "grid cells" else 6000, 81)
    exp = oracle.box2d_nms_hard_candidates(b, s, "rbox", 0.3, 0.1)
    hook = _lib.NMS_TEST_WITHHOLD | (_lib.NMS_GENERAL if scan == "grid cells" else _lib.NMS_BROAD_SWEEP)
    set_opts(nms_flags=hook)
    for _ in range(2):
        keep = box2d_nms(T(b), T(s), iou_method="rbox", iou_threshold=0.3, score_threshold=0.1).cpu().numpy()
        assert np.array_equal(keep, exp)
    # the give-up is reported, not silent (d3d_nms2d_status)
    sup, status = box.nms2d(T(b), T(s), box.IouType.RBOX, 0, 0.3, 0.1, 0.0, return_status=True)
    assert status == (box.NMS_STATUS_DENSE_PATH | box.NMS_STATUS_SCAN_GAVE_UP)
    set_opts(nms_flags=hook & ~_lib.NMS_TEST_WITHHOLD)         # and the next call is healthy again
    keep = box2d_nms(T(b), T(s), iou_method="rbox", iou_threshold=0.3, score_threshold=0.1).cpu().numpy()
    assert np.array_equal(keep, exp)
    sup2, status = box.nms2d(T(b), T(s), box.IouType.RBOX, 0, 0.3, 0.1, 0.0, return_status=True)
    assert status == 0 and torch.equal(sup, sup2)


@pytest.mark.parametrize("n", [128, 3000])
def test_nms_sweep_and_prune_edge_geometry(n):
    """broad phases (uniform grid / sort by AABB xmin + walk): negative coordinates, many identical xmin, boxes spanning the
    whole scene (more cells than one box may register in: the grid hands over to the dense path), zero-size boxes, a
    non-finite box -- keep set bit-exact with the oracle"""
    from d3d_amd.box import box2d_nms
    rng = np.random.default_rng(n)
    b = np.empty((n, 5))
    b[:, 0] = rng.uniform(-300, 300, n)
    b[:, 1] = rng.uniform(-300, 300, n)
    b[:, 2] = rng.uniform(1, 40, n)
    b[:, 3] = rng.uniform(1, 40, n)
    b[:, 4] = rng.uniform(-3.2, 3.2, n)
    b[: n // 8, 0] = np.round(b[: n // 8, 0] / 50) * 50          # columns of boxes sharing x
    b[: n // 8, 2:4] = 20.0
    b[: n // 8, 4] = 0.0                                         # ... and therefore the same AABB xmin
    b[n // 8: n // 8 + 5, 2:4] = 2000.0                          # a few boxes covering everything
    b[n // 4: n // 4 + 5, 2] = 0.0                               # degenerate
    b[n // 2, 0] = np.nan                                        # non-finite boxes: never overlap anything
    b[n // 2 + 1, 2] = np.inf
    b[n // 2 + 2, 1] = -np.inf
    s = rng.permutation(n) / n + 0.001
    for method in ["rbox", "box"]:
        keep = box2d_nms(T(b), T(s), iou_method=method, iou_threshold=0.25).cpu().numpy()
        exp = oracle.box2d_nms(b, s, iou_method=method, iou_threshold=0.25)
        assert np.array_equal(keep, exp), (method, int(np.sum(keep != exp)))


def test_nms_list_path_equals_dense_path_at_scale(monkeypatch):
    """30k boxes at the cfg3 density: candidate-list path == all-pairs bit-matrix path (the reference's structure)"""
    from d3d_amd import synth
    from d3d_amd.box import box2d_nms
    b, s = synth.boxes2d_sparse(30000, 5)
    keep = box2d_nms(T(b), T(s), iou_method="rbox", iou_threshold=0.3).cpu().numpy()
    from d3d_amd import _lib, box
    set_opts(nms_flags=cur_opts().nms_flags | _lib.NMS_FORCE_DENSE)
    dense = box2d_nms(T(b), T(s), iou_method="rbox", iou_threshold=0.3).cpu().numpy()
    assert np.array_equal(keep, dense) and 0 < keep.sum() < len(keep)


def test_nms_score_ties_are_stable():
    from d3d_amd.box import box2d_nms
    b, _ = bc.random_boxes_like_reference(400, 8)
    s = np.round(np.random.default_rng(9).random(400) * 10) / 10   # many ties
    keep = box2d_nms(T(b.astype(np.float64)), T(s), iou_method="rbox", iou_threshold=0.2).cpu().numpy()
    assert np.array_equal(keep, oracle.box2d_nms(b.astype(np.float64), s, iou_method="rbox", iou_threshold=0.2))


def test_nms_top_box_below_score_threshold():
    """CPU semantics (nms.cpp:23): sorted position 0 is never pre-suppressed by the score threshold."""
    from d3d_amd.box import box2d_nms
    b, s = bc.random_boxes_like_reference(50, 10)
    keep = box2d_nms(T(b), T(s * 0.5), iou_method="box", score_threshold=0.9).cpu().numpy()
    exp = oracle.box2d_nms(b, s * 0.5, iou_method="box", score_threshold=0.9)
    assert np.array_equal(keep, exp) and keep.sum() == 1


@pytest.mark.parametrize("method", ["rbox", "box"])
def test_iou3d_vs_oracle(method):
    from d3d_amd import synth
    from d3d_amd.box import iou3d
    pred, gt = synth.boxes3d_eval(300, 4, 2)
    got = iou3d(T(pred), T(gt), method).cpu().numpy()
    exp = oracle.iou3d(pred, gt, method, nthreads=8)
    assert got.shape == (1200, 300) and got.dtype == np.float32
    assert np.max(np.abs(got - exp)) < 1e-3
    # evaluator vectors (test_benchmark.py:31-39, 45-71)
    v = iou3d(T(bc.EVAL_DT), T(bc.EVAL_GT))[0, 0].item()
    assert v > 0.1 and abs(v - bc.EVAL_IOU) < 1e-4
    assert np.isclose(iou3d(T(bc.EVAL_DT), T(bc.EVAL_DT))[0, 0].item(), 1)


def test_full_size_cfg4_properties():
    """BASELINE config 4 at full size (20k x 5k): symmetry / range / matched-pair properties."""
    from d3d_amd import synth
    from d3d_amd.box import iou3d
    pred, gt = synth.boxes3d_eval(5000, 4, 2)
    m = iou3d(T(pred), T(gt))
    assert m.shape == (20000, 5000)
    assert float(m.min()) >= 0 and float(m.max()) <= 1 + 1e-3
    mt = iou3d(T(gt), T(pred))
    assert torch.allclose(m, mt.t(), atol=2e-3)            # IoU is symmetric
    best = m.argmax(1).cpu().numpy()
    assert np.mean(best == np.repeat(np.arange(5000), 4)) > 0.9   # each pred matches its own GT
    # EVERY pair of the 20 k x 5 k matrix against the oracle (1e8 pairs, NaN-poisoned result buffer), both methods; the
    # non-zero pattern too: a pair is non-zero here iff it is in the oracle, up to pairs whose overlap is rounding noise
    for method, mat in (("rbox", m), ("box", iou3d(T(pred), T(gt), "box"))):
        got = mat.cpu().numpy()
        exp = oracle.iou3d(pred, gt, method, nthreads=8)
        assert not np.isnan(got).any()
        assert np.max(np.abs(got - exp)) < 1e-3
        differ = (got != 0) != (exp != 0)          # zero on one side only: overlaps at the scale of fp32 rounding (1e-3 bar)
        assert np.count_nonzero(exp) > 150000 and np.max(np.maximum(got, exp)[differ], initial=0.0) < 2e-4
        # (4 % of the oracle's non-zeros lie below 2e-4 -- BEV overlap times z overlap of boxes that barely touch; only there
        # may rounding decide between zero and not)
        assert int(differ.sum()) <= int(((exp > 0) & (exp < 2e-4)).sum()) + int(((got > 0) & (got < 2e-4)).sum())
        assert abs(np.count_nonzero(got) - np.count_nonzero(exp)) <= np.count_nonzero(exp) // 100


def test_box_crop_reference_case():
    """reference test/test_box.py:191-205 (with the function's real name and return type)"""
    from d3d_amd.box import box2dr_crop
    rng = np.random.default_rng(40)
    cloud = (rng.random((100, 2)) * 2 - 1).astype(np.float32)
    boxes = np.array([[0, 0, 1, 1, 0], [0, 0, 1, 1, bc.d90]], np.float32)
    res = box2dr_crop(T(cloud), T(boxes)).cpu().numpy()
    a = np.abs(cloud)
    assert res.shape == (2, 100) and res.dtype == np.bool_
    assert np.array_equal(np.where(res[0])[0], np.where(np.all(a < 0.5, 1))[0])
    assert np.array_equal(np.where(res[1])[0], np.where(np.abs(a[:, 0] + a[:, 1]) < bc.sq2 / 2)[0])


@pytest.mark.parametrize("method", ["rbox", "box"])
def test_iou_backward_row_reduction_is_additive(method):
    """k_iou_grad sums the row gradients of a wavefront's candidates across the lanes (a segmented scan over runs of equal row)
    before its atomics.  The gradient is additive over any split of the columns, and a split changes every run: rows with
    hundreds of candidates (runs longer than a wavefront), rows that come back in later batches, weighted upstream gradients
    with zeros in them -- full == left + right in fp64"""
    from d3d_amd import synth
    from d3d_amd.box import box2d_iou
    b, _ = synth.boxes2d_dense(900, 5)
    b1, b2 = b[:400], b[400:]
    rng = np.random.default_rng(0)
    w = rng.random((400, 500))
    w[rng.random((400, 500)) < 0.3] = 0.0
    wt = torch.from_numpy(w).cuda()

    def grads(cols):
        t1 = torch.from_numpy(b1).cuda().requires_grad_(True)
        t2 = torch.from_numpy(np.ascontiguousarray(b2[cols])).cuda().requires_grad_(True)
        (box2d_iou(t1, t2, method=method) * wt[:, cols]).sum().backward()
        return t1.grad.cpu().numpy(), t2.grad.cpu().numpy()
    full1, full2 = grads(slice(0, 500))
    l1, l2 = grads(slice(0, 137))
    r1, r2 = grads(slice(137, 500))
    assert np.abs(full1 - (l1 + r1)).max() < 1e-9 * max(np.abs(full1).max(), 1.0)
    assert np.abs(full2 - np.concatenate([l2, r2])).max() < 1e-9 * max(np.abs(full2).max(), 1.0)
    assert np.abs(full1).max() > 0 and np.abs(full2).max() > 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [1000, 1003])
def test_crop_vs_oracle(dtype, n):
    from d3d_amd.box import box2dr_crop, box3dp_crop
    rng = np.random.default_rng(41)
    pts = ((rng.random((n, 2)) - 0.5) * 12).astype(dtype)
    boxes, _ = bc.random_boxes_like_reference(70, 42)
    boxes = boxes.astype(dtype)
    boxes[:, :2] = (rng.random((70, 2)) - 0.5) * 10
    boxes[:, 2:4] *= 0.2
    got = box2dr_crop(T(pts), T(boxes)).cpu().numpy()
    exp = oracle.crop_2dr(pts, boxes)
    assert got.shape == (70, n)
    assert np.array_equal(got, exp)     # bit for bit: quad_contains forms corners and products in the oracle's order
    assert exp.sum() > 100
    p3 = np.concatenate([pts, ((rng.random((n, 1)) - 0.5) * 4).astype(dtype)], 1)
    b3 = np.stack([boxes[:, 0], boxes[:, 1], np.zeros(70), boxes[:, 2], boxes[:, 3], np.full(70, 2.0), boxes[:, 4]], 1).astype(dtype)
    for ax in [2]:
        g3 = box3dp_crop(T(p3), T(b3), ax).cpu().numpy()
        e3 = oracle.box3dp_crop(p3, b3, ax)
        assert np.array_equal(g3, e3) and e3.sum() > 10


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_crop_2dr_2k_boxes_x_1m_points_bit_exact(dtype):
    """box2dr_crop at the size of a GT-sampling pass, the WHOLE 2 GB mask against the oracle (250 box rows at a time), and
    box3dp_crop (box/__init__.py:289-315) on the same scene: array_equal, no mismatch budget"""
    from d3d_amd.box import box2dr_crop, box3dp_crop
    m, n = 2000, 1000000
    rng = np.random.default_rng(43)
    pts = ((rng.random((n, 3)) - 0.5) * np.array([140.0, 140.0, 4.0])).astype(dtype)
    b3 = np.stack([(rng.random(m) - 0.5) * 130, (rng.random(m) - 0.5) * 130, (rng.random(m) - 0.5) * 2,
                   3.5 + 1.5 * rng.random(m), 1.6 + 0.5 * rng.random(m), 1.4 + 0.5 * rng.random(m),
                   (rng.random(m) - 0.5) * 2 * np.pi], 1).astype(dtype)
    b2 = np.ascontiguousarray(b3[:, [0, 1, 3, 4, 6]])
    p2 = np.ascontiguousarray(pts[:, :2])
    got2 = box2dr_crop(T(p2), T(b2))
    got3 = box3dp_crop(T(pts), T(b3), 2)
    assert got2.shape == (m, n) and got2.dtype == torch.bool and got3.shape == (m, n)
    total = 0
    for i0 in range(0, m, 250):
        e2 = oracle.crop_2dr(p2, b2[i0:i0 + 250])
        assert torch.equal(got2[i0:i0 + 250], torch.from_numpy(e2).cuda()), i0
        if i0 < 500:                    # (numpy temporaries of 250 x 1 M per comparison: two row blocks are enough)
            e3 = oracle.box3dp_crop(pts, b3[i0:i0 + 250], 2)
            assert torch.equal(got3[i0:i0 + 250], torch.from_numpy(e3).cuda()), i0
        total += int(e2.sum())
    assert total > 100000


@pytest.mark.parametrize("method", ["rbox", "box"])
def test_iou3d_candidate_list_overflow_takes_every_pair(method):
    """pairwise 3D IoU: more overlapping pairs than the candidate list holds (2^27) -- 11 700 x 11 700 boxes crowded into a few
    metres -- and k_iou3d_clip computes every pair itself (round 6: no fallback launch); row / column samples against the oracle"""
    from d3d_amd.box import iou3d
    rng = np.random.default_rng(77)
    k = 11700

    def mk():
        return np.concatenate([rng.normal(0, 0.7, (k, 2)), rng.uniform(-1.5, -0.5, (k, 1)), rng.uniform(3.5, 5, (k, 1)),
                               rng.uniform(3.4, 4.2, (k, 1)), rng.uniform(1.4, 1.9, (k, 1)), rng.uniform(-3.14, 3.14, (k, 1))], 1).astype(np.float32)
    a, b = mk(), mk()
    got = iou3d(T(a), T(b), method=method)
    assert got.shape == (k, k)
    assert int((got > 0).sum()) > (1 << 27)                       # the list did overflow
    ri, ci = rng.choice(k, 40, replace=False), rng.choice(k, 300, replace=False)
    exp = oracle.iou3d(a[ri], b[ci], method)
    sub = got[torch.from_numpy(ri).cuda()][:, torch.from_numpy(ci).cuda()].cpu().numpy()
    assert np.max(np.abs(sub - exp)) < 1e-3                       # fp32 tolerance of BASELINE.json
    exp = oracle.iou3d(a[:3], b, method)
    assert np.max(np.abs(got[:3].cpu().numpy() - exp)) < 1e-3


def test_iou_candidate_list_overflow_falls_back(monkeypatch):
    """two-phase rbox IoU: when the candidate list is too small the single-kernel path recomputes the matrix"""
    from d3d_amd import synth
    from d3d_amd.box import box2d_iou
    b1, _ = synth.boxes2d_dense(300, 51)
    b2, _ = synth.boxes2d_dense(200, 52)
    exp = oracle.box2d_iou(b1, b2, "rbox", nthreads=4)
    from d3d_amd import _lib, box
    set_opts(iou_flags=_lib.iou_list_cap(64))
    got = box2d_iou(T(b1), T(b2), method="rbox").cpu().numpy()
    assert np.max(np.abs(got - exp)) < 1e-9
    set_opts(iou_flags=0)
    got = box2d_iou(T(b1), T(b2), method="rbox").cpu().numpy()
    assert np.max(np.abs(got - exp)) < 1e-9


def _numeric_grads(b1, b2, g, method, h=1e-6):
    """central differences of the fp64 oracle IoU: d sum(g * iou) / d boxes"""
    def f(x1, x2):
        return float(np.sum(g * oracle.iou2d_forward(x1, x2, method)))
    g1, g2 = np.zeros_like(b1), np.zeros_like(b2)
    for arr, out, first in ((b1, g1, True), (b2, g2, False)):
        for i in range(arr.shape[0]):
            for k in range(5):
                p, q = arr.copy(), arr.copy()
                p[i, k] += h
                q[i, k] -= h
                out[i, k] = (f(p, b2) - f(q, b2)) / (2 * h) if first else (f(b1, p) - f(b1, q)) / (2 * h)
    return g1, g2


@pytest.mark.parametrize("method", ["rbox", "box"])
def test_iou_backward_vs_finite_differences(method):
    """"next" row 2: analytic gradients of the HIP path against central differences of the fp64 oracle"""
    from d3d_amd.box import box2d_iou
    rng = np.random.default_rng(61)
    n, m = 14, 11
    mk = lambda k: np.stack([(rng.random(k) - .5) * 6, (rng.random(k) - .5) * 6, rng.random(k) * 3 + 1, rng.random(k) * 3 + 1,
                             (rng.random(k) - .5) * 6], 1)   # noqa: E731
    b1, b2 = mk(n), mk(m)
    g = rng.random((n, m))
    t1 = torch.from_numpy(b1).cuda().requires_grad_(True)
    t2 = torch.from_numpy(b2).cuda().requires_grad_(True)
    iou = box2d_iou(t1, t2, method=method)
    assert (iou > 0).sum() > 20
    (iou * torch.from_numpy(g).cuda()).sum().backward()
    e1, e2 = _numeric_grads(b1, b2, g, method)
    assert np.max(np.abs(t1.grad.cpu().numpy() - e1)) < 2e-6 * max(1.0, np.abs(e1).max())
    assert np.max(np.abs(t2.grad.cpu().numpy() - e2)) < 2e-6 * max(1.0, np.abs(e2).max())
    # disjoint boxes: zero gradient; identical boxes: finite
    far = torch.tensor([[100., 100, 1, 1, 0]], dtype=torch.float64, device="cuda", requires_grad=True)
    box2d_iou(far, t2.detach(), method=method).sum().backward()
    assert torch.all(far.grad == 0)


def test_iou_backward_fp32_and_large():
    from d3d_amd import synth
    from d3d_amd.box import box2d_iou
    b, _ = synth.boxes2d_dense(600, 71, np.float32)
    t1 = torch.from_numpy(b).cuda().requires_grad_(True)
    t2 = torch.from_numpy(b[::-1].copy()).cuda().requires_grad_(True)
    box2d_iou(t1, t2, method="rbox", precise=False).sum().backward()
    d1 = torch.from_numpy(b.astype(np.float64)).cuda().requires_grad_(True)
    d2 = torch.from_numpy(b[::-1].astype(np.float64).copy()).cuda().requires_grad_(True)
    box2d_iou(d1, d2, method="rbox").sum().backward()
    assert torch.isfinite(t1.grad).all() and t1.grad.dtype == torch.float32
    rel = (t1.grad.double() - d1.grad).abs().max() / d1.grad.abs().max()
    assert rel < 5e-3        # fp32 clip vs fp64 clip of the same boxes


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_box_method_ragged_shapes_and_degenerate_rectangles(dtype):
    """method "box" goes through the same zero-fill + candidate path as "rbox": odd matrix shapes, and rectangles
    without area whose bounding box has one (w = 0, rotated) keep their AABB IoU"""
    from d3d_amd.box import box2d_iou
    rng = np.random.default_rng(3)
    b1 = bc.random_boxes_like_reference(131, 21)[0].astype(dtype)
    b2 = bc.random_boxes_like_reference(77, 22)[0].astype(dtype)
    b1[:5, 2] = 0.0                       # zero width, rotated: AABB still has an area
    b1[:5, 4] = 0.7
    b2[:3, 3] = 0.0
    b2[:3, 4] = -0.4
    b2[10:15, :2] = b1[:5, :2]            # full boxes centred on the degenerate ones
    got = box2d_iou(T(b1), T(b2), method="box", precise=False).cpu().numpy()
    exp = oracle.box2d_iou(b1.astype(np.float64), b2.astype(np.float64), "box")
    assert got.shape == (131, 77)
    assert np.max(np.abs(got - exp)) < (1e-9 if dtype == np.float64 else 1e-3)
    assert np.any(exp[:5] > 0)


@pytest.mark.parametrize("sup,param", [("linear", 1.0), ("linear", 2.0), ("gaussian", 0.5), ("gaussian", 0.1)])
@pytest.mark.parametrize("method", ["rbox", "box"])
def test_softnms_vs_oracle(method, sup, param, monkeypatch):
    """soft-NMS (nms.cpp:60-94) on the device: same keep mask as the literal restatement, incl. the insertion re-sort;
    state in LDS and (hook) in global scratch"""
    from d3d_amd import synth
    from d3d_amd.box import box2d_nms
    for n, seed, gen in [(300, 41, synth.boxes2d_dense), (1200, 42, synth.boxes2d_sparse), (1, 43, synth.boxes2d_dense)]:
        b, s = gen(n, seed)
        for sthr in [0.0, 0.25]:
            kw = dict(iou_method=method, supression_method=sup, iou_threshold=0.2, score_threshold=sthr, supression_param=param)
            exp = oracle.box2d_nms(b, s, **kw)
            keep = box2d_nms(T(b), T(s), **kw).cpu().numpy()
            assert np.array_equal(keep, exp), (n, sthr, int(np.sum(keep != exp)))
            if n == 300:
                from d3d_amd import _lib, box
                set_opts(nms_flags=cur_opts().nms_flags | _lib.NMS_SOFT_NO_LDS)
                keep = box2d_nms(T(b), T(s), **kw).cpu().numpy()
                set_opts(nms_flags=cur_opts().nms_flags & ~_lib.NMS_SOFT_NO_LDS)
                assert np.array_equal(keep, exp)


def test_softnms_reference_case_and_fp32():
    """test_box.py:157-179: nothing is suppressed at score_threshold 0; fp32 path (precise=False)"""
    from d3d_amd.box import box2d_nms
    for m in ["box", "rbox"]:
        for sup in ["linear", "gaussian"]:
            assert bool(box2d_nms(T(bc.SOFT_BOXES), T(bc.NMS_SCORES), iou_method=m, supression_method=sup).all())
    b, s = bc.random_boxes_like_reference(500, 6)
    kw = dict(iou_method="rbox", supression_method="gaussian", iou_threshold=0.1, score_threshold=0.3, supression_param=0.3)
    keep = box2d_nms(T(b), T(s), precise=False, **kw).cpu().numpy()
    exp = oracle.box2d_nms(b, s, precise=False, **kw)
    assert np.mean(keep == exp) > 0.995          # fp32 powf / expf may differ from libm in the last ulp


def test_softnms_beyond_65536_boxes():
    """soft-NMS has no size limit (nms.cpp:60-94 has none): 70 000 boxes, position-indexed state in global scratch.  5000 boxes
    score above the threshold, 65 000 below it -- those are suppressed before the loop starts (nms.cpp:23-29), are never
    processed (a suppressed box ends the loop, :38) and only sink; so the keep mask of the 5000 must be that of the 5000 on
    their own plus ONE low box (which gives the insertion pass the same extent, :75), from the literal loop restricted to
    touching pairs (equal to the literal loop: tests/test_oracle_box.py)."""
    from d3d_amd import synth
    from d3d_amd.box import box2d_nms
    rng = np.random.default_rng(77)
    b, _ = synth.boxes2d_sparse(70000, 31)
    s = np.concatenate([0.5 + 0.5 * rng.random(5000), 0.3 * rng.random(65000)])
    perm = rng.permutation(70000)
    b, s = np.ascontiguousarray(b[perm]), np.ascontiguousarray(s[perm])
    top = np.flatnonzero(s >= 0.5)
    low = int(np.flatnonzero(s < 0.5)[np.argmax(s[s < 0.5])])                 # the best of the low boxes: first behind the 5000
    sub = np.concatenate([top, [low]])
    for method, sup, param in (("rbox", "linear", 1.0), ("box", "gaussian", 0.5)):
        keep = box2d_nms(T(b), T(s), iou_method=method, supression_method=sup, iou_threshold=0.05, score_threshold=0.4,
                         supression_param=param).cpu().numpy()
        exp = oracle.box2d_nms_soft_candidates(b[sub], s[sub], method, sup, 0.05, 0.4, param)
        assert not keep[s < 0.5].any()
        assert np.array_equal(keep[top], exp[:-1]) and not exp[-1]
        assert 1000 < keep.sum() < 5000


def test_nms_fp32_scores_sorted_narrow():
    """fp32 scores (precise=True promotes them to fp64): the argsort runs on the fp32 keys -- same keep mask, ties included"""
    from d3d_amd.box import box2d_nms
    b, s = bc.random_boxes_like_reference(3000, 12)
    s = (np.round(s * 50) / 50).astype(np.float32)                    # many ties
    keep = box2d_nms(T(b), T(s), iou_method="rbox", iou_threshold=0.2, score_threshold=0.1).cpu().numpy()
    exp = oracle.box2d_nms(b, s, iou_method="rbox", iou_threshold=0.2, score_threshold=0.1)
    assert np.array_equal(keep, exp)
    s2 = np.stack([s, s * 0.5], 1)                                    # class scores: max over classes
    keep2 = box2d_nms(T(b), T(s2), iou_method="rbox", iou_threshold=0.2, score_threshold=0.1).cpu().numpy()
    assert np.array_equal(keep2, exp)


@pytest.mark.parametrize("nobj,per", [(30, 100), (8, 400), (60, 300), (3, 3000)])
def test_nms_detector_like_clusters(nobj, per):
    """clusters of heavily overlapping boxes around each object (hundreds of hitters per box; cell lists spanning many
    wavefronts of the level kernels): the levels decide most of a cluster before any pair is listed, incoming lists of any
    length keep the rest on the list path; keep mask bit-exact with the oracle"""
    from d3d_amd.box import box2d_nms
    rng = np.random.default_rng(nobj)
    c = np.stack([rng.random(nobj) * 300, rng.random(nobj) * 300, rng.random(nobj) * 20 + 10, rng.random(nobj) * 20 + 10,
                  rng.random(nobj) * 6.28], 1)
    b = np.repeat(c, per, 0) + rng.normal(0, 1, (nobj * per, 5)) * [1.5, 1.5, 1.0, 1.0, 0.05]
    s = rng.permutation(nobj * per) / (nobj * per)
    from d3d_amd import _lib, box
    base = cur_opts().nms_flags
    for method, thr in [("rbox", 0.5), ("box", 0.3)]:
        exp = oracle.box2d_nms(b, s, iou_method=method, iou_threshold=thr)
        for extra in (0, _lib.NMS_ONE_LEVEL):             # (two levels of the level kernels, or one)
            set_opts(nms_flags=base | extra)              # (unwound by conftest)
            keep = box2d_nms(T(b), T(s), iou_method=method, iou_threshold=thr).cpu().numpy()
            assert np.array_equal(keep, exp), (method, extra, int(np.sum(keep != exp)))
        assert keep.sum() < 20 * nobj


def test_nms_launch_guess_alternating_inputs():
    """whether the grid's level kernels are LAUNCHED is a guess from the grid density of the calls before (one host-mapped word
    + a streak counter): scattered boxes after clusters launch them for nothing (they test the density on the device and
    return), clusters after a STREAK of calls on scattered boxes run without them (every pair is listed, the round-2 route),
    alternating inputs always launch -- the keep mask never depends on it"""
    from d3d_amd import synth
    from d3d_amd.box import box2d_nms
    rng = np.random.default_rng(3)
    c = np.stack([rng.random(30) * 300, rng.random(30) * 300, rng.random(30) * 20 + 10, rng.random(30) * 20 + 10, rng.random(30) * 6.28], 1)
    bc = np.repeat(c, 300, 0) + rng.normal(0, 1, (9000, 5)) * [1.5, 1.5, 1.0, 1.0, 0.05]
    sc = rng.permutation(9000) / 9000
    bs, ss = synth.boxes2d_sparse(6000, 17)
    exp_c = oracle.box2d_nms(bc, sc, iou_method="rbox", iou_threshold=0.5)
    exp_s = oracle.box2d_nms(bs, ss, iou_method="rbox", iou_threshold=0.5)
    seq = [(bs, ss, exp_s)] * 6 + [(bc, sc, exp_c)] * 2 + [(bs, ss, exp_s), (bc, sc, exp_c)] * 2 + [(bs, ss, exp_s)] * 5 + [(bc, sc, exp_c)]
    for b, s, exp in seq:
        keep = box2d_nms(T(b), T(s), iou_method="rbox", iou_threshold=0.5).cpu().numpy()
        assert np.array_equal(keep, exp)


def test_c_abi_error_codes():
    """the C ABI reports bad arguments / unsupported options with status codes (nothing throws, nothing exits:
    reference common.h:25,33-46 exit()s on CUDA errors and throws py::value_error on unsupported enums)"""
    import ctypes
    from d3d_amd import _lib
    lib = _lib.load()
    z = ctypes.c_void_p(0)
    b = torch.zeros((4, 5), dtype=torch.float64, device="cuda")
    out = torch.zeros((4, 4), dtype=torch.float64, device="cuda")
    sup = torch.zeros((4,), dtype=torch.uint8, device="cuda")
    order = torch.arange(4, dtype=torch.int64, device="cuda")
    ws = torch.zeros((1 << 20,), dtype=torch.uint8, device="cuda")
    p = _lib.ptr
    assert lib.d3d_iou2d_forward(z, 4, p(b), 4, 2, _lib.F64, p(out), z, 0, z, 0) == _lib.ERR_BAD_ARG            # null boxes
    assert lib.d3d_iou2d_forward(p(b), -1, p(b), 4, 2, _lib.F64, p(out), z, 0, z, 0) == _lib.ERR_BAD_ARG       # negative size
    assert lib.d3d_iou2d_forward(p(b), 4, p(b), 4, 3, _lib.F64, p(out), z, 0, z, 0) == _lib.ERR_UNSUPPORTED    # GBOX
    assert lib.d3d_iou2d_forward(p(b), 0, p(b), 4, 2, _lib.F64, z, z, 0, z, 0) == 0                             # empty: ok
    assert lib.d3d_iou2d_forward(p(b), 4, p(b), 4, 4, _lib.F64_M32, p(out), z, 0, z, 0) == _lib.ERR_UNSUPPORTED  # GRBOX with an fp32 matrix
    assert lib.d3d_iou2d_forward(p(b), 4, p(b), 4, 2, 7, p(out), z, 0, z, 0) == _lib.ERR_BAD_ARG                # unknown dtype code
    big = torch.zeros((300, 5), dtype=torch.float64, device="cuda")
    o32 = torch.zeros((300, 300), dtype=torch.float32, device="cuda")
    assert lib.d3d_iou2d_forward(p(big), 300, p(big), 300, 2, _lib.F64_M32, p(o32), z, 0, z, 0) == _lib.ERR_WORKSPACE   # mixed form: list path only
    assert lib.d3d_iou2d_workspace_bytes(300, 300, _lib.F64_M32) == lib.d3d_iou2d_workspace_bytes(300, 300, _lib.F64)
    assert lib.d3d_nms2d(p(b), p(b[:, 0].contiguous()), p(order), 4, 3, 0, _lib.F64, 0.5, 0.0, 0.0, p(sup), p(ws), ws.numel(),
                         z, 0) == _lib.ERR_UNSUPPORTED                                                          # GBOX in NMS
    assert lib.d3d_nms2d(p(b), p(b[:, 0].contiguous()), p(order), 4, 2, 7, _lib.F64, 0.5, 0.0, 0.0, p(sup), p(ws), ws.numel(),
                         z, 0) == _lib.ERR_UNSUPPORTED                                                          # suppression enum
    assert lib.d3d_nms2d(p(b), p(b[:, 0].contiguous()), p(order), 4, 2, 0, _lib.F64, 0.5, 0.0, 0.0, p(sup), p(ws), 16,
                         z, 0) == _lib.ERR_WORKSPACE                                                            # workspace too small
    assert lib.d3d_iou3d_forward(z, 4, z, 4, 1, z, z, 0, z) == _lib.ERR_BAD_ARG                                     # null boxes
    assert lib.d3d_status_string(_lib.ERR_WORKSPACE) == b"workspace too small"
    torch.cuda.synchronize()


@pytest.mark.parametrize("n", [20000, 3000, 800])
def test_nms_is_graph_capturable(n):
    """no host synchronisation, no allocation inside the library: box2d_nms records into a HIP graph and replays -- the general
    path and the small-set path (3000 fp64 boxes: more than 64 KB of LDS for the sort inside its first kernel)"""
    from d3d_amd import synth
    from d3d_amd.box import box2d_nms
    b, s = synth.boxes2d_sparse(n, 3)
    b[:, :2] *= (n / 100000.0) ** 0.5                     # same density at every size
    bt, st = T(b), T(s)
    ref = box2d_nms(bt, st, iou_method="rbox", iou_threshold=0.4)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # warm-up off the capture stream (workspace allocation)
        box2d_nms(bt, st, iou_method="rbox", iou_threshold=0.4)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = box2d_nms(bt, st, iou_method="rbox", iou_threshold=0.4)
    st.copy_(torch.from_numpy(s[::-1].copy()).cuda())    # new scores, same buffers
    g.replay()
    torch.cuda.synchronize()
    exp = oracle.box2d_nms(b, s[::-1].copy(), iou_method="rbox", iou_threshold=0.4)
    assert np.array_equal(out.cpu().numpy(), exp) and not np.array_equal(exp, ref.cpu().numpy())


def test_soft_nms_trailing_zero_scores_come_back():
    """score_threshold 0 with scores that are exactly 0: the reference first suppresses the trailing boxes whose score is not
    ABOVE the threshold (nms.cpp:23-29) and later ASSIGNS suppressed = score < threshold to every box it rescales
    (nms.cpp:58,62), which brings them back -- and reorders the tail (fuzz seed 3025)"""
    from d3d_amd.box import box2d_nms
    rng = np.random.default_rng(3025)
    n = 1370
    b = np.stack([rng.random(n) * 1000, rng.random(n) * 1000, rng.random(n) * 20 + 0.5, rng.random(n) * 20 + 0.5,
                  (rng.random(n) - 0.5) * 8], 1)
    b[: n // 3] = b[0] + rng.normal(0, 0.3, (n // 3, 5))
    s = np.round(rng.random(n) * 20) / 20
    assert (s == 0).sum() > 10
    for sup in ("gaussian", "linear"):
        for method in ("box", "rbox"):
            kw = dict(iou_method=method, supression_method=sup, iou_threshold=0.1, score_threshold=0.0, supression_param=2.0)
            keep = box2d_nms(torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda(), **kw).cpu().numpy()
            assert np.array_equal(keep, oracle.box2d_nms(b, s, **kw)), (sup, method)


@pytest.mark.parametrize("n", [1, 2, 63, 4095, 4096, 4097, 100000, 131073])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_argsort_desc_stable_all_sizes(n, dtype):
    """d3d_argsort_desc: descending, equal keys (+0 and -0 included) in index order, infinities at the ends"""
    from d3d_amd.box import argsort_desc
    rng = np.random.default_rng(n)
    s = rng.random(n).astype(dtype)
    if n > 10:
        s[rng.integers(0, n, n // 5)] = s[rng.integers(0, n, n // 5)]          # ties
        s[rng.integers(0, n, 3)] = 0.0
        s[rng.integers(0, n, 3)] = -0.0
        s[rng.integers(0, n, 2)] = -np.inf
        s[rng.integers(0, n, 2)] = np.inf
        s[::7] *= -1
    got = argsort_desc(T(s)).cpu().numpy()
    assert got.dtype == np.int64 and np.array_equal(got, np.argsort(-s, kind="stable"))


def _expected_order(s):
    """torch's descending order (nms.cpp:103): NaN first, -0 == +0, ties in index order"""
    nan = np.isnan(s)
    val = np.where(nan, 0, s)
    return np.lexsort((np.arange(len(s)), -val, ~nan))


@pytest.mark.parametrize("n", [8191, 8192, 8193, 50000, 131072, 131073, (1 << 20) + 3])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_argsort_desc_bucket_path(n, dtype):
    """the 8 k .. 128 k range takes the bucket (sample) sort, the sizes around it the library: same order on random keys with
    ties, on constant / sorted / few-valued inputs (the splitters cut runs of equal keys by index), and on NaN, +-0, +-inf;
    the library path at the same size agrees entry by entry"""
    import ctypes
    from d3d_amd import _lib
    from d3d_amd.box import argsort_desc
    lib = _lib.load()
    rng = np.random.default_rng(n)
    cases = {"random": rng.random(n), "constant": np.full(n, 0.25), "ascending": np.arange(n) / n, "descending": -np.arange(n) / n,
             "few": rng.integers(0, 5, n) / 4.0, "normal": rng.normal(0, 1e3, n)}
    special = rng.random(n)
    special[rng.integers(0, n, n // 5)] = special[rng.integers(0, n, n // 5)]
    special[rng.integers(0, n, 50)] = np.nan
    special[rng.integers(0, n, 50)] = 0.0
    special[rng.integers(0, n, 50)] = -0.0
    special[rng.integers(0, n, 5)] = np.inf
    special[rng.integers(0, n, 5)] = -np.inf
    special[::5] *= -1
    cases["special"] = special
    lib.d3d_internal_argsort_desc_radix.restype = ctypes.c_int
    lib.d3d_internal_argsort_desc_radix.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_size_t, ctypes.c_void_p]
    for name, s in cases.items():
        s = s.astype(dtype)
        t = T(s)
        got = argsort_desc(t).cpu().numpy()
        exp = _expected_order(s)
        assert np.array_equal(got, exp), (name, int(np.sum(got != exp)))
        if name in ("special", "few"):
            code = 1 if dtype == np.float64 else 0
            order = torch.empty(n, dtype=torch.int64, device="cuda")
            ws = torch.empty(lib.d3d_argsort_desc_workspace_bytes(n, code), dtype=torch.uint8, device="cuda")
            rc = lib.d3d_internal_argsort_desc_radix(t.data_ptr(), n, code, order.data_ptr(), ws.data_ptr(), ws.numel(), None)
            torch.cuda.synchronize()
            assert rc == 0 and np.array_equal(order.cpu().numpy(), exp), name


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097])
def test_nms_small_set_path_sizes(n):
    """the small-set path (<= 4096 boxes: scores sorted inside the first kernel, all-pairs candidates, one-workgroup fixed point
    up to 1024 boxes) at its size boundaries, fp64 and fp32 scores with ties, clustered and scattered boxes: bit-exact with the
    oracle and with the general path"""
    from d3d_amd import _lib
    from d3d_amd.box import box2d_nms, nms2d, IouType
    rng = np.random.default_rng(n)
    nobj = max(n // 40, 1)
    c = np.stack([rng.random(nobj) * 300, rng.random(nobj) * 300, rng.random(nobj) * 20 + 10, rng.random(nobj) * 20 + 10,
                  rng.random(nobj) * 6.28], 1)
    b = c[rng.integers(0, nobj, n)] + rng.normal(0, 1, (n, 5)) * [2.0, 2.0, 1.5, 1.5, 0.08]
    s = np.round(rng.random(n) * 200) / 200                       # ties
    for method, thr, sthr in (("rbox", 0.5, 0.0), ("rbox", 0.2, 0.3), ("box", 0.4, 0.1)):
        exp = oracle.box2d_nms(b, s, iou_method=method, iou_threshold=thr, score_threshold=sthr)
        keep = box2d_nms(T(b), T(s), iou_method=method, iou_threshold=thr, score_threshold=sthr).cpu().numpy()
        assert np.array_equal(keep, exp), (method, thr, int(np.sum(keep != exp)))
        sup = nms2d(T(b), T(s), IouType[method.upper()], 0, thr, sthr, 0.0, flags=_lib.NMS_GENERAL).cpu().numpy()
        assert np.array_equal(~sup, exp), ("general", method)
    b32, s32 = b.astype(np.float32), s.astype(np.float32)
    k32 = box2d_nms(T(b32), T(s32), iou_method="rbox", iou_threshold=0.5, precise=False).cpu().numpy()
    g32 = nms2d(T(b32), T(s32), IouType.RBOX, 0, 0.5, 0.0, 0.0, flags=_lib.NMS_GENERAL).cpu().numpy()
    assert np.array_equal(k32, ~g32)                              # same fp32 arithmetic on both paths


def test_nms_c_abi_order_optional():
    """d3d_nms2d with order = NULL sorts the scores itself (any size, hard and soft); with the caller's order it follows it --
    a different valid order of tied scores gives the mask of THAT order (nms.cpp:103 leaves ties to the sort)"""
    import ctypes
    from d3d_amd import _lib
    lib = _lib.load()
    for n in (700, 6000):
        b, s = bc.random_boxes_like_reference(n, n)
        b, s = b.astype(np.float64), np.round(s.astype(np.float64) * 20) / 20
        bt, st = T(b), T(s)
        ws = torch.empty(lib.d3d_nms2d_workspace_bytes(n), dtype=torch.uint8, device="cuda")
        for sup_type, param in ((0, 0.0), (1, 0.5)):
            out = {}
            for name, order in (("internal", None), ("given", T(np.argsort(-s, kind="stable").astype(np.int64)))):
                sup = torch.empty(n, dtype=torch.uint8, device="cuda")
                rc = lib.d3d_nms2d(bt.data_ptr(), st.data_ptr(), order.data_ptr() if order is not None else None, n, 2, sup_type, 1,
                                   0.3, 0.1, float(param), sup.data_ptr(), ws.data_ptr(),
                                   ws.numel(), None, 0)
                torch.cuda.synchronize()
                assert rc == 0
                out[name] = sup.cpu().numpy()
            assert np.array_equal(out["internal"], out["given"])
        rev = np.lexsort((-np.arange(n), -s)).astype(np.int64)           # ties in DESCENDING index order: also a valid order
        sup = torch.empty(n, dtype=torch.uint8, device="cuda")
        rc = lib.d3d_nms2d(bt.data_ptr(), st.data_ptr(), T(rev).data_ptr(), n, 2, 0, 1, 0.3, 0.1,
                           0.0, sup.data_ptr(), ws.data_ptr(), ws.numel(), None, 0)
        torch.cuda.synchronize()
        assert rc == 0
        if n <= 1000:                                                      # the greedy loop of nms.cpp:23-59 over THAT order
            iou = oracle.box2d_iou(b, b, "rbox")
            supp = np.zeros(n, bool)
            supp[rev[1:]] = ~(s[rev[1:]] > np.float32(0.1))
            for a, i in enumerate(rev):
                if not supp[i]:
                    later = rev[a + 1:]
                    supp[later[iou[i, later] > np.float32(0.3)]] = True
            assert np.array_equal(sup.cpu().numpy().astype(bool), supp)


def test_small_matrix_kernels_equal_the_two_phase_path():
    """matrices of up to 65536 pairs take a one-launch kernel (one pair per lane); larger ones geometry + fill + candidate list
    + clip.  Same candidates, same per-pair function: a 100 x 300 block computed on its own equals the first 100 rows of the
    300 x 300 matrix bit for bit -- IoU (rbox / box, fp64 / fp32), iou3d (rbox / box) and the matcher's distance; degenerate
    boxes and identical boxes included"""
    from d3d_amd import synth
    from d3d_amd.box import box2d_iou, iou3d
    from d3d_amd.tracking import DistanceTypes, prepare_boxes
    b, _ = synth.boxes2d_dense(300, 9)
    b[5, 2] = 0.0                                        # degenerate
    b[7] = b[6]                                          # identical
    for method in ("rbox", "box"):
        for dt in (np.float64, np.float32):
            x = T(b.astype(dt))
            big = box2d_iou(x, x, method=method, precise=dt == np.float64).cpu().numpy()
            small = box2d_iou(x[:100], x, method=method, precise=dt == np.float64).cpu().numpy()
            assert np.array_equal(big[:100], small), (method, dt)
            assert np.max(np.abs(big - oracle.box2d_iou(b.astype(dt).astype(np.float64), b.astype(dt).astype(np.float64), method))) < (1e-9 if dt == np.float64 else 2e-3)
    p, g = synth.boxes3d_eval(75, 4, 5)                  # 300 predictions around 75 ground truths
    p[3, 3] = 0.0
    for method in ("rbox", "box"):
        big = iou3d(T(p), T(p), method=method).cpu().numpy()
        small = iou3d(T(p[:100]), T(p), method=method).cpu().numpy()
        assert np.array_equal(big[:100], small), method
    d9 = np.concatenate([np.ones((300, 1)), np.linspace(1, 0, 300)[:, None], p], 1).astype(np.float32)
    for metric in (DistanceTypes.RIoU, DistanceTypes.IoU):
        big = prepare_boxes(d9, d9, metric).cpu().numpy()
        small = prepare_boxes(d9[:100], d9, metric).cpu().numpy()
        assert np.array_equal(big[:100], small), metric


def test_nms_two_threads_two_streams_mixed_inputs():
    """nothing is shared between NMS callers any more (the launch decision for the level kernels comes from
    the call's own grid through the calling thread's host word).  Two threads on two streams, one alternating clustered and
    scattered sets, the other the reverse, ten rounds: every mask equals the oracle's."""
    import threading
    from d3d_amd import synth
    from d3d_amd.box import box2d_nms
    rc = np.random.default_rng(3)
    cc = np.stack([rc.random(60) * 600, rc.random(60) * 600, rc.random(60) * 20 + 10, rc.random(60) * 20 + 10, rc.random(60) * 6.28], 1)
    bclu = np.repeat(cc, 150, 0) + rc.normal(0, 1, (9000, 5)) * [1.5, 1.5, 1.0, 1.0, 0.05]
    sclu = rc.random(9000)
    bspa, sspa = synth.boxes2d_sparse(9000, 4)
    exp = {"clu": oracle.box2d_nms_hard_candidates(bclu, sclu, "rbox", 0.5, 0.0),
           "spa": oracle.box2d_nms_hard_candidates(bspa, sspa, "rbox", 0.5, 0.0)}
    errors = []

    def worker(first):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                sets = {"clu": (T(bclu), T(sclu)), "spa": (T(bspa), T(sspa))}
                seq = ["clu", "spa"] if first == "clu" else ["spa", "clu"]
                for it in range(10):
                    name = seq[it & 1]
                    keep = box2d_nms(*sets[name], iou_method="rbox", iou_threshold=0.5).cpu().numpy()
                    if not np.array_equal(keep, exp[name]):
                        errors.append((first, it, name, int(np.sum(keep != exp[name]))))
        except Exception as e:      # pragma: no cover
            errors.append((first, repr(e)))
    threads = [threading.Thread(target=worker, args=(f,)) for f in ("clu", "spa")]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors


@pytest.mark.gpu
def test_crop_2dr_fp32_point_on_a_long_edge_follows_the_hosts_sine():
    """found by tests/fuzz.py (seed 20267): for this angle the device's sincosf is one ulp off glibc's sinf -- the correctly
    rounded float, which the reference's host code and the oracle use -- and with it the point, exactly ON the 54-unit edge,
    fell outside.  fp32 angles now take the host's own sinf / cosf operation for operation (geom.hpp: HostSinCos, checked
    against libm in tests/test_host_sincos.py).  Both kernels: all pairs (few points) and the box grid (>= 4096 points)."""
    from d3d_amd.box import crop_2dr
    box = np.array([[16.303468704223633, 5.941005706787109, 54.238182067871094, 3.4432148933410645, -2.555604934692383]], np.float32)
    pt = np.array([[7.366237163543701, 2.075172185897827]], np.float32)
    assert oracle.crop_2dr(pt, box)[0, 0]
    assert bool(crop_2dr(T(pt), T(box)).cpu().numpy()[0, 0])
    rng = np.random.default_rng(3)
    many = np.concatenate([pt, (rng.random((5000, 2)) * 40 - 10).astype(np.float32)])
    got = crop_2dr(T(many), T(box)).cpu().numpy()
    assert got[0, 0] and np.array_equal(got, oracle.crop_2dr(many, box))


@pytest.mark.parametrize("npts", [1500, 6000])
def test_crop_2dr_fp32_points_within_an_ulp_of_an_edge(npts):
    """every point is planted on an edge of its own rotated box (computed in double, rounded to fp32): which side it falls on is
    decided by the last bit of the box's sine and cosine, so the mask equals the oracle's only if the device's fp32 angles are the
    host's bit for bit -- seed 61633 of tests/fuzz.py was a box whose glibc sinf is not even the correctly rounded float."""
    from d3d_amd.box import crop_2dr
    rng = np.random.default_rng(npts)
    m = min(npts, 3000)                               # 6000 points on 3000 boxes take the box grid, 1500 on 1500 all pairs
    box = np.stack([rng.random(m) * 80 - 40, rng.random(m) * 80 - 40, rng.random(m) * 50 + 5, rng.random(m) * 4 + 1,
                    (rng.random(m) - 0.5) * 12], 1).astype(np.float32)
    which = np.arange(npts) % m
    b = box.astype(np.float64)[which]
    t = rng.random(npts) * 2 - 1
    side = rng.integers(0, 4, npts)
    lx = np.where(side < 2, t * b[:, 2] / 2, np.where(side == 2, b[:, 2] / 2, -b[:, 2] / 2))
    ly = np.where(side == 0, b[:, 3] / 2, np.where(side == 1, -b[:, 3] / 2, t * b[:, 3] / 2))
    cs, sn = np.cos(b[:, 4]), np.sin(b[:, 4])
    pts = np.stack([b[:, 0] + lx * cs - ly * sn, b[:, 1] + lx * sn + ly * cs], 1).astype(np.float32)
    exp = oracle.crop_2dr(pts, box)
    own = exp[which, np.arange(npts)]
    assert 0.2 < own.mean() < 0.8                     # the planted points really straddle their edges
    assert np.array_equal(crop_2dr(T(pts), T(box)).cpu().numpy(), exp)


@pytest.mark.parametrize("method", ["rbox", "box"])
@pytest.mark.parametrize("shape", [(7, 9), (1, 1), (300, 400), (257, 1023), (3, 70001), (2100, 33)])
def test_precise_on_fp32_boxes_rounds_where_the_matrix_is_stored(method, shape):
    """box2d_iou(precise=True) on fp32 boxes (the reference's default call, box/__init__.py:204-205, 224: boxes.double(), fp64
    kernels, ious.to(float32)) runs as fp64 arithmetic with an fp32 matrix (D3D_F64_M32): the values are those of the explicit
    chain bit for bit -- small matrices (one launch), the two-phase path, an overflowed candidate list -- and within an fp32
    ulp of the oracle's fp64 values; the gradients are those of the chain (fp64 atomics: the order of the sums is not fixed)"""
    from d3d_amd import _lib, synth
    from d3d_amd.box import IouType, Iou2D, Iou2DR, box2d_iou
    n, m = shape
    gen = synth.boxes2d_dense if n * m < 200000 else synth.boxes2d_sparse
    b1 = gen(n, 61)[0].astype(np.float32)
    b2 = gen(m, 62)[0].astype(np.float32)
    if n > 4 and m > 4:
        b2[:4] = b1[:4]                                          # identical boxes
        b1[4, 2] = 0.0                                           # a degenerate rectangle
    fn = Iou2DR if method == "rbox" else Iou2D
    for cap in (0, 8):
        if cap and n * m <= 65536:
            continue                                             # (one launch, no list)
        set_opts(iou_flags=_lib.iou_list_cap(cap) if cap else 0)
        t1, t2 = T(b1).requires_grad_(True), T(b2).requires_grad_(True)
        got = box2d_iou(t1, t2, method=method)
        assert got.dtype == torch.float32 and got.shape == (n, m)
        c1, c2 = T(b1).requires_grad_(True), T(b2).requires_grad_(True)
        chain = fn.apply(c1.double(), c2.double()).to(torch.float32)
        assert torch.equal(got, chain)
        from d3d_amd.box import _iou_forward
        assert torch.equal(_iou_forward(T(b1).double(), T(b2).double(), IouType[method.upper()], matrix32=True), got.detach())   # D3D_F64_M32
        exp = oracle.box2d_iou(b1.astype(np.float64), b2.astype(np.float64), method, nthreads=4)
        assert np.max(np.abs(got.detach().cpu().numpy().astype(np.float64) - exp)) <= 6.1e-8
        w = torch.from_numpy((np.random.default_rng(5).random((n, m)) - 0.3).astype(np.float32)).cuda()
        (got * w).sum().backward()
        (chain * w).sum().backward()
        for a, b in ((t1.grad, c1.grad), (t2.grad, c2.grad)):
            assert a.dtype == torch.float32 and torch.allclose(a, b, rtol=2e-6, atol=1e-7)
        from d3d_amd.box import _iou_backward
        m1, m2 = _iou_backward(T(b1).double(), T(b2).double(), w, IouType[method.upper()], matrix32=True)      # D3D_F64_M32: fp32 grad, fp64 boxes
        assert m1.dtype == torch.float64 and torch.allclose(m1.float(), c1.grad, rtol=2e-6, atol=1e-7) and torch.allclose(m2.float(), c2.grad, rtol=2e-6, atol=1e-7)
    set_opts(iou_flags=0)
    # numpy in, numpy out, on the same path
    gn = box2d_iou(b1, b2, method=method)
    assert isinstance(gn, np.ndarray) and gn.dtype == np.float32 and np.array_equal(gn, got.detach().cpu().numpy())


def test_precise_on_fp32_boxes_allocates_no_fp64_matrix():
    """4096 x 4096: the fp32 matrix is 64 MB; the chain around fp64 kernels held 128 MB + 64 MB at its peak"""
    from d3d_amd import synth
    from d3d_amd.box import box2d_iou
    b = T(synth.boxes2d_sparse(4096, 3)[0].astype(np.float32))
    box2d_iou(b, b, method="rbox")                               # (workspace arena allocated)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    before = torch.cuda.memory_allocated()
    out = box2d_iou(b, b, method="rbox")
    torch.cuda.synchronize()
    assert out.dtype == torch.float32
    assert torch.cuda.max_memory_allocated() - before < 100 * (1 << 20)


@pytest.mark.parametrize("n", [100, 1500, 3000, 20000])
def test_nms_keep_mask_is_the_inverted_suppressed_mask(n, nms_broad):
    """D3D_NMS_KEEP_MASK (what box2d_nms asks for: the reference returns ~suppressed, box/__init__.py:272): every kernel that
    decides a box writes the inverted bit -- the small-set resolve, the fixed point, the dense sweep, soft-NMS"""
    from d3d_amd import _lib, synth
    from d3d_amd.box import IouType, box2d_nms, nms2d
    b, s = synth.boxes2d_dense(n, 71) if n <= 3000 else synth.boxes2d_sparse(n, 71)
    bt, st = T(b), T(s)
    base = cur_opts().nms_flags
    for extra in (0, _lib.NMS_FORCE_DENSE):
        sup = nms2d(bt, st, IouType.RBOX, 0, 0.3, 0.1, 0.0, flags=base | extra)
        keep = nms2d(bt, st, IouType.RBOX, 0, 0.3, 0.1, 0.0, flags=base | extra, keep_mask=True)
        assert keep.dtype == torch.bool and torch.equal(keep, ~sup) and 0 < int(keep.sum()) < n
    assert torch.equal(box2d_nms(bt, st, iou_method="rbox", iou_threshold=0.3, score_threshold=0.1), keep)
    assert np.array_equal(keep.cpu().numpy(), oracle.box2d_nms(b, s, iou_method="rbox", iou_threshold=0.3, score_threshold=0.1))
    if n <= 3000 and nms_broad == "auto":
        for sm in (1, 2):
            sup = nms2d(bt, st, IouType.RBOX, sm, 0.3, 0.1, 0.5)
            assert torch.equal(nms2d(bt, st, IouType.RBOX, sm, 0.3, 0.1, 0.5, keep_mask=True), ~sup)


@pytest.mark.parametrize("n", [100, 1500, 3000, 20000])
def test_nms_precise_on_fp32_tensors_widens_inside_the_kernels(n, nms_broad):
    """box2d_nms(precise=True) on fp32 boxes and scores (reference box/__init__.py:254-255: boxes.double(), scores.double()) runs
    as D3D_F32_WIDE -- fp64 arithmetic on values widened where they are loaded, the score order from the fp32 keys: the masks
    of the explicit fp64 call, ties in the scores included; soft-NMS too"""
    from d3d_amd import synth
    from d3d_amd.box import box2d_nms
    b, s = synth.boxes2d_dense(n, 81) if n <= 3000 else synth.boxes2d_sparse(n, 81)
    b32 = b.astype(np.float32)
    for ties in (False, True):
        s32 = (np.round(s * 50) / 50 if ties else s).astype(np.float32)
        bt, st = T(b32), T(s32)
        keep = box2d_nms(bt, st, iou_method="rbox", iou_threshold=0.3, score_threshold=0.1)
        wide = box2d_nms(bt.double(), st.double(), iou_method="rbox", iou_threshold=0.3, score_threshold=0.1)
        assert torch.equal(keep, wide) and 0 < int(keep.sum()) < n
        if not ties:
            assert np.array_equal(keep.cpu().numpy(), oracle.box2d_nms(b32.astype(np.float64), s32.astype(np.float64), iou_method="rbox",
                                                                       iou_threshold=0.3, score_threshold=0.1))
    assert torch.equal(box2d_nms(bt, st, iou_method="box", iou_threshold=0.5), box2d_nms(bt.double(), st.double(), iou_method="box", iou_threshold=0.5))
    if n <= 1500 and nms_broad == "auto":
        for sup, prm in (("linear", 1.0), ("gaussian", 0.5)):
            kw = dict(iou_method="rbox", supression_method=sup, iou_threshold=0.3, score_threshold=0.2, supression_param=prm)
            assert torch.equal(box2d_nms(bt, st, **kw), box2d_nms(bt.double(), st.double(), **kw))
    # [N,K] class scores: the class maximum commutes with the widening
    sk = torch.stack([st, st * 0.5], dim=1)
    assert torch.equal(box2d_nms(bt, sk, iou_method="rbox", iou_threshold=0.3), box2d_nms(bt, st, iou_method="rbox", iou_threshold=0.3))


@pytest.mark.parametrize("n,m", [(120000, 50), (5000, 1), (70001, 700), (4096, 4096)])
def test_box3dp_crop_one_launch_equals_the_composition(n, m):
    """box3dp_crop along z as one launch (d3d_crop_3dp) against the reference's composition (box/__init__.py:289-315: crop_2dr on
    gathered columns & the interval test as [M,N] tensor operations), bit for bit: points ON box faces and edges (copies of box
    centres shifted by exactly half an extent), NaN points, a degenerate box, extra point columns; numpy / CPU ingress"""
    from d3d_amd import synth
    from d3d_amd.box import box3dp_crop, crop_2dr
    rng = np.random.default_rng(n + m)
    _, g = synth.boxes3d_eval(max(m, 2), 1, 5)
    boxes = g[:m].astype(np.float32).copy()
    pts = np.stack([rng.random(n) * 150, rng.random(n) * 150, rng.random(n) * 4 - 3, rng.random(n)], 1).astype(np.float32)
    k = min(m, n // 8)
    pts[:k, :3] = boxes[:k, :3]                                                   # centres
    pts[k:2 * k, :3] = boxes[:k, :3]; pts[k:2 * k, 2] += boxes[:k, 5] / 2        # on the top face (z - d / 2 == b: outside, strict)
    pts[2 * k:3 * k, :3] = boxes[:k, :3]; pts[2 * k:3 * k, 2] -= boxes[:k, 5] / 2
    pts[3 * k, 0] = np.nan
    if m > 3:
        boxes[3, 3] = 0.0                                                        # a degenerate rectangle
    for cols in (4, 3):
        p, b = T(np.ascontiguousarray(pts[:, :cols])), T(boxes)
        got = box3dp_crop(p, b)
        pz, bz, hd = p[:, [2]].t(), b[:, [2]], b[:, [5]] / 2
        exp = crop_2dr(p[:, [0, 1]], b[:, [0, 1, 3, 4, 6]]) & ((pz - hd < bz) & (bz < pz + hd))
        assert got.dtype == torch.bool and got.shape == (m, n) and torch.equal(got, exp)
        assert int(got.sum()) >= k                                                # the centres are inside
    host = box3dp_crop(torch.from_numpy(pts[:, :3].copy()), torch.from_numpy(boxes))
    assert not host.is_cuda and torch.equal(host, got.cpu())
    for axis in (0, 1):                                                           # the other axes: the composition, as before
        assert box3dp_crop(p, b, project_axis=axis).shape == (m, n)
    with pytest.raises(ValueError):
        box3dp_crop(p, b, project_axis=3)
