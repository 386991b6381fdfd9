"""soft-NMS against the CPU oracle on ties, exact-zero and negative scores, score_threshold 0 (development aid)"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch, oracle
from d3d_amd.box import box2d_nms
bad = 0
for seed in range(5000, 5400):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 1200))
    scale = float(rng.choice([20, 100, 1000]))
    b = np.stack([rng.random(n) * scale, rng.random(n) * scale, rng.random(n) * 20 + 0.5, rng.random(n) * 20 + 0.5,
                  (rng.random(n) - 0.5) * 8], 1)
    if seed % 2 == 0:
        b[: n // 3] = b[0] + rng.normal(0, 0.3, (n // 3, 5))
    s = rng.random(n)
    if seed % 3 != 2:
        s = np.round(s * 10) / 10                 # ties, exact zeros
    if seed % 7 == 0:
        s = s - 0.3                               # negative scores
    method = str(rng.choice(["box", "rbox"]))
    thr, sthr = float(rng.choice([0.0, 0.1, 0.3, 0.5])), float(rng.choice([0.0, 0.0, 0.1, 0.2, 0.5]))
    sup = str(rng.choice(["linear", "gaussian"]))
    kw = dict(iou_method=method, supression_method=sup, iou_threshold=thr, score_threshold=sthr,
              supression_param=float(rng.choice([0.3, 1.0, 2.0])))
    keep = box2d_nms(torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda(), **kw).cpu().numpy()
    exp = oracle.box2d_nms(b, s, **kw)
    if not np.array_equal(keep, exp):
        bad += 1
        print("seed", seed, n, kw, "FAILED", int(np.sum(keep != exp)))
print("soft-NMS fuzz: 400 seeds,", bad, "failures")
