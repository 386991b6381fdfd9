"""Target3DArray.crop_points / paint_label (reference d3d/abstraction.pyx:308-324, 654-687; per pair
box3dr_contains, d3d/dgal_wrap.h:6-19).  CPU: the oracle's restatement on hand-made cases (closed z interval, closed edges,
paint order).  GPU: d3d_amd.abstraction against the oracle, bit for bit, up to 2 k boxes x 1 M points."""
import numpy as np
import pytest
import torch

import oracle


def _scene(m, n, seed, classes=3):
    rng = np.random.default_rng(seed)
    boxes = np.stack([rng.random(m) * 100 - 50, rng.random(m) * 100 - 50, rng.random(m) * 2 - 1.5,
                      rng.random(m) * 4 + 1.5, rng.random(m) * 1.5 + 1.2, rng.random(m) * 1.5 + 1.0,
                      rng.random(m) * 6.28 - 3.14], 1).astype(np.float32)
    pts = np.stack([rng.random(n) * 110 - 55, rng.random(n) * 110 - 55, rng.random(n) * 4 - 2.5, rng.random(n)], 1).astype(np.float32)
    # a third of the points inside boxes on purpose (random points rarely are), some exactly on top faces
    k = n // 3
    which = rng.integers(0, m, k)
    c, s = np.cos(boxes[which, 6]), np.sin(boxes[which, 6])
    u, v = (rng.random(k) - 0.5) * boxes[which, 3] * 1.1, (rng.random(k) - 0.5) * boxes[which, 4] * 1.1
    pts[:k, 0] = boxes[which, 0] + c * u - s * v
    pts[:k, 1] = boxes[which, 1] + s * u + c * v
    pts[:k, 2] = boxes[which, 2] + (rng.random(k) - 0.5) * boxes[which, 5] * 1.1
    top = np.arange(0, k, 17)
    pts[top, 2] = boxes[which[top], 2] + boxes[which[top], 5] / 2           # z == z + lz/2 in fp32: inside (closed)
    labels = rng.integers(1, classes + 1, m).astype(np.uint8)
    sem = rng.integers(0, classes + 1, n).astype(np.uint8)
    scores = np.sort(rng.random(m))[::-1]
    rows9 = np.concatenate([labels[:, None], scores[:, None], boxes], 1).astype(np.float32)
    return boxes, rows9, pts, labels, sem


def test_oracle_box3dr_contains_is_closed_in_z_and_on_edges():
    """dgal_wrap.h:12-13 rejects only zq > z + lz/2 or zq < z - lz/2 -- unlike box3dp_crop's strict test
    (box/__init__.py:313); edges and corners of the rectangle count as inside"""
    b = np.array([[0, 0, 0, 2, 1, 2, 0.0], [5, 5, 0, 2, 2, 2, np.pi / 4]], np.float32)
    pts = np.array([[0.9, 0.4, 1.0], [1.0, 0.5, 1.0], [1.0, 0.5, -1.0], [1.01, 0, 0], [0, 0, 1.0001], [0, 0, -1.0001],
                    [5, 5 + 1.41, 0], [5, 5 + 1.42, 0], [5.9, 5.9, 0]], np.float32)
    got = oracle.crop_points(b, pts)
    assert got[0].tolist() == [True, True, True, False, False, False, False, False, False]
    assert got[1].tolist() == [False, False, False, False, False, False, True, False, False]
    strict = oracle.box3dp_crop(pts, b)
    assert not strict[0, 1] and got[0, 1]                 # the face point: strict test out, box3dr_contains in


def test_oracle_paint_label_order_and_classes():
    """abstraction.pyx:662-673: painted from the last box to the first, so the lowest index wins; class must match"""
    boxes, rows9, pts, labels, sem = _scene(40, 3000, 3)
    boxes[1] = boxes[0]                                     # two identical boxes ...
    rows9[1, 2:] = rows9[0, 2:]
    labels[1] = labels[0]; rows9[1, 0] = rows9[0, 0]        # ... of the same class: index 0 must win
    mask = oracle.crop_points(rows9, pts)
    ids = oracle.paint_label(rows9, pts, sem)
    assert ids.dtype == np.uint16 and mask.shape == (40, 3000)
    exp = np.zeros(3000, np.uint16)
    for ib in range(39, -1, -1):
        exp[mask[ib] & (sem == labels[ib])] = ib + 1
    assert np.array_equal(ids, exp)
    assert not np.any(ids == 2) and np.any(ids == 1)
    assert np.array_equal(oracle.paint_label(boxes, pts, sem, labels=labels), exp)     # [M,7] + explicit labels
    assert np.array_equal(oracle.crop_points(boxes, pts), mask)


@pytest.mark.gpu
@pytest.mark.parametrize("m,n", [(1, 1), (3, 1001), (64, 4096), (130, 20000)])
def test_crop_points_and_paint_label_vs_oracle(m, n):
    from d3d_amd.abstraction import crop_points, paint_label
    boxes, rows9, pts, labels, sem = _scene(m, n, 100 + m)
    exp_mask, exp_ids = oracle.crop_points(boxes, pts), oracle.paint_label(rows9, pts, sem)
    for bx in (boxes, rows9):
        got = crop_points(torch.from_numpy(bx).cuda(), torch.from_numpy(pts).cuda())
        assert got.dtype == torch.bool and np.array_equal(got.cpu().numpy(), exp_mask)
    assert exp_mask.sum() > 0.2 * n or n < 10
    got = paint_label(torch.from_numpy(rows9).cuda(), torch.from_numpy(pts).cuda(), torch.from_numpy(sem).cuda())
    assert np.array_equal(got.cpu().numpy().astype(np.uint16), exp_ids)
    got = paint_label(boxes, pts[:, :3].copy(), sem, labels=labels)             # numpy in -> numpy out, xyz-only cloud
    assert got.dtype == np.uint16 and np.array_equal(got, exp_ids)
    assert np.array_equal(crop_points(boxes, pts), exp_mask)


@pytest.mark.gpu
def test_crop_points_2k_boxes_x_1m_points():
    """2000 boxes x 1 M points (the size of a GT-sampling / label-painting pass): 30 k sampled points against the oracle,
    every output; the whole 2 GB mask against paint_label (the painted id is the first set row of the class-filtered
    mask) and against its own column sums"""
    from d3d_amd.abstraction import crop_points, paint_label
    m, n = 2000, 1000000
    boxes, rows9, pts, labels, sem = _scene(m, n, 77)
    pt, bt, st = torch.from_numpy(pts).cuda(), torch.from_numpy(rows9).cuda(), torch.from_numpy(sem).cuda()
    mask = crop_points(bt, pt)
    ids = paint_label(bt, pt, st)
    assert mask.shape == (m, n) and ids.shape == (n,)
    cols = np.random.default_rng(1).choice(n, 30000, replace=False)
    sub = np.ascontiguousarray(pts[cols])
    assert np.array_equal(mask[:, torch.from_numpy(cols).cuda()].cpu().numpy(), oracle.crop_points(rows9, sub))
    assert np.array_equal(ids[torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.uint16), oracle.paint_label(rows9, sub, sem[cols]))
    # whole arrays, GPU-side: first box row (ascending index) whose class matches the point's label and that contains it
    lab = torch.from_numpy(labels).cuda()
    first = torch.zeros((n,), dtype=torch.int32, device="cuda")
    for i0 in range(0, m, 250):                               # 250 x 1 M bools at a time
        hit = mask[i0:i0 + 250] & (lab[i0:i0 + 250, None] == st[None, :])
        any_hit = hit.any(0)
        idx = hit.int().argmax(0).int() + i0 + 1
        first = torch.where((first == 0) & any_hit, idx, first)
    assert torch.equal(first, ids.int())
    assert int(mask.sum()) > 300000


@pytest.mark.gpu
def test_paint_label_refuses_class_ids_beyond_uint8():
    """class ids travel as uint8; a value that would wrap in the cast is an error, not a silent mismatch"""
    from d3d_amd.abstraction import paint_label
    boxes, rows9, pts, labels, sem = _scene(5, 100, 3)
    bad = labels.astype(np.int64).copy()
    bad[2] = 300
    with pytest.raises(ValueError):
        paint_label(boxes, pts, sem, labels=bad)
    neg = sem.astype(np.int32)
    neg[0] = -1
    with pytest.raises(ValueError):
        paint_label(boxes, pts, neg, labels=labels)
    ok = paint_label(boxes, pts, sem.astype(np.int64), labels=labels.astype(np.int64))      # in range: any integer dtype
    assert np.array_equal(ok, oracle.paint_label(rows9, pts, sem))
