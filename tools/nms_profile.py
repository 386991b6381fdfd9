#!/usr/bin/env python3
"""per-kernel HIP-event timings of box2d_nms (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import kernel_profile, timed
from d3d_amd import synth
from d3d_amd.box import box2d_nms
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for name, mk, thr in [("sparse", synth.boxes2d_sparse, 0.5), ("dense", synth.boxes2d_dense, 0.3)]:
    nn = n if name == "sparse" else min(n, 20000)
    b, s = mk(nn, 1)
    bt, st = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
    f = lambda: box2d_nms(bt, st, iou_method="rbox", iou_threshold=thr)
    dt = timed(f, 3, 1)
    prof = kernel_profile(f, 3)
    print(name, nn, "%.2f ms" % (dt / 3 * 1e3), "kept", int(f().sum()),
          {k: round(v["total_ms"] / 3 * 1e3, 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])})
