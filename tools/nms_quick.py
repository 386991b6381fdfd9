"""box2d_nms steady-state time per call: config 3's scattered 100 k boxes, clustered raw detections, one after the other (no history
is kept between calls since round 4).  usage (GPU box): python tools/nms_quick.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from bench import timed
from d3d_amd import synth
from d3d_amd.box import box2d_nms

b, s = synth.boxes2d_sparse(100000, 1)
bt, st = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
rc = np.random.default_rng(1)
cc = np.stack([rc.random(200) * 2000, rc.random(200) * 2000, rc.random(200) * 20 + 10, rc.random(200) * 20 + 10, rc.random(200) * 6.28], 1)
bc = torch.from_numpy(np.repeat(cc, 500, 0) + rc.normal(0, 1, (100000, 5)) * [1.5, 1.5, 1.0, 1.0, 0.05]).cuda()
sc = torch.from_numpy(rc.random(100000)).cuda()
f_s = lambda: box2d_nms(bt, st, iou_method="rbox", iou_threshold=0.5)
f_c = lambda: box2d_nms(bc, sc, iou_method="rbox", iou_threshold=0.5)
for rep in range(2):
    print("scattered 100k: %.1f us/call (%.0f M boxes/s)" % (1e6 * timed(f_s, 20, 1) / 20, 100000 * 20 / timed(f_s, 20, 0) / 1e6))
    print("clusters 200x500: %.1f us/call" % (1e6 * timed(f_c, 10, 1) / 10))
    torch.cuda.synchronize()
    print("  one clustered call right after scattered ones: %.1f us" % (1e6 * timed(f_c, 1, 0)), flush=True)
    f_s(); torch.cuda.synchronize()
