import sys, torch, numpy as np
sys.path.insert(0, ".")
import bench
from d3d_amd import synth
from d3d_amd.box import box2d_nms, box2d_iou
for n in (500, 2000, 8000, 30000):
    rng = np.random.default_rng(n)
    nobj = max(n // 50, 1)
    c = np.stack([rng.random(nobj) * 400, rng.random(nobj) * 400, rng.random(nobj) * 20 + 10, rng.random(nobj) * 20 + 10, rng.random(nobj) * 6.28], 1)
    b = np.repeat(c, n // nobj, 0) + rng.normal(0, 1, (nobj * (n // nobj), 5)) * [1.5, 1.5, 1.0, 1.0, 0.05]
    s = rng.random(len(b))
    for dt_ in (np.float64, np.float32):
        bt, st = torch.from_numpy(b.astype(dt_)).cuda(), torch.from_numpy(s.astype(dt_)).cuda()
        f = lambda: box2d_nms(bt, st, iou_method="rbox", iou_threshold=0.5, precise=dt_ == np.float64)
        t = bench.timed(f, 30, 5)
        f2 = lambda: box2d_iou(bt[:1000], bt[:1000], method="rbox", precise=dt_ == np.float64)
        t2 = bench.timed(f2, 30, 5)
        print("n=%6d %s clustered: nms %.1f us/call (stream)   iou 1000x1000 %.1f us" % (len(b), dt_.__name__, t / 30 * 1e6, t2 / 30 * 1e6))
