#!/usr/bin/env python3
"""seeded fuzzing of the HIP path against the CPU oracle beyond the seeds of the test-suite (development aid):
python tools/fuzz.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
import test_gpu_voxel as tv
from d3d_amd.box import box2d_iou, box2d_nms

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for seed in range(first, first + count):
    try:
        tv.test_randomized_configs_vs_oracle(seed)
    except AssertionError as e:
        bad += 1; print("VOXEL seed", seed, "FAILED", str(e)[:200])
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 1500))
    scale = float(rng.choice([20, 100, 1000]))
    b = np.stack([rng.random(n) * scale, rng.random(n) * scale, rng.random(n) * 20 + 0.5, rng.random(n) * 20 + 0.5,
                  (rng.random(n) - 0.5) * 8], 1)
    if seed % 4 == 0:
        b[: n // 3] = b[0] + rng.normal(0, 0.3, (n // 3, 5))              # a cluster
    s = rng.random(n)
    if seed % 5 == 0:
        s = np.round(s * 20) / 20                                          # ties
    method = str(rng.choice(["box", "rbox"]))
    thr, sthr = float(rng.choice([0.0, 0.1, 0.3, 0.5, 0.7])), float(rng.choice([0.0, 0.2, 0.6]))
    sup = str(rng.choice(["hard", "hard", "linear", "gaussian"]))
    kw = dict(iou_method=method, supression_method=sup, iou_threshold=thr, score_threshold=sthr, supression_param=float(rng.choice([0.3, 1.0, 2.0])))
    keep = box2d_nms(torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda(), **kw).cpu().numpy()
    exp = oracle.box2d_nms(b, s, **kw)
    if not np.array_equal(keep, exp):
        bad += 1; print("NMS seed", seed, kw, n, "FAILED", int(np.sum(keep != exp)))
    m = int(rng.integers(1, 400))
    b2 = b[rng.integers(0, n, m)] + rng.normal(0, 1.0, (m, 5))
    got = box2d_iou(torch.from_numpy(b).cuda(), torch.from_numpy(b2).cuda(), method=method).cpu().numpy()
    ref = oracle.box2d_iou(b, b2, method)
    err = float(np.max(np.abs(got - ref))) if got.size else 0.0
    if err > 1e-9:
        bad += 1; print("IOU seed", seed, method, n, m, "FAILED", err)
print("fuzz: %d seeds, %d failures" % (count, bad))
