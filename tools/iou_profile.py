#!/usr/bin/env python3
"""per-kernel HIP-event timings of iou3d (config 4) and rotated iou2d (config 3 row block, dense 5k) (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import kernel_profile, timed
from d3d_amd import synth
from d3d_amd.box import iou3d, iou2dr_forward

def show(tag, f, units, reps=10):
    dt = timed(f, reps, 2)
    prof = kernel_profile(f, reps)
    print(tag, "%.1f us/call  %.1f G/s" % (dt / reps * 1e6, units * reps / dt / 1e9),
          {k: round(v["avg_us"], 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])})

p, g = synth.boxes3d_eval(5000, 4, 2)
pt, gt = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
show("iou3d 20k x 5k fp32", lambda: iou3d(pt, gt), 1e8)
b, _ = synth.boxes2d_sparse(100000, 1)
bt = torch.from_numpy(b).cuda()
show("iou2dr 20k x 100k fp64", lambda: iou2dr_forward(bt[:20000], bt), 2e9, 5)
bd, _ = synth.boxes2d_dense(5000, 1)
bdt = torch.from_numpy(bd).cuda()
show("iou2dr dense 5k x 5k fp64", lambda: iou2dr_forward(bdt, bdt), 25e6)
show("iou2dr 20k x 5k fp64", lambda: iou2dr_forward(bt[:20000], bt[:5000]), 1e8)
show("iou2dr 20k x 10k fp64", lambda: iou2dr_forward(bt[:20000], bt[:10000]), 2e8)
bf = bt.float()
show("iou2dr 20k x 5k fp32", lambda: iou2dr_forward(bf[:20000], bf[:5000]), 1e8)
show("iou2dr 20k x 20k fp32", lambda: iou2dr_forward(bf[:20000], bf[:20000]), 4e8)
show("iou2dr 5k x 20k fp32", lambda: iou2dr_forward(bf[:5000], bf[:20000]), 1e8)
show("iou2dr 20k x 5001 fp32", lambda: iou2dr_forward(bf[:20000], bf[:5001]), 1.0002e8)
show("iou2dr 20001 x 4999 fp64", lambda: iou2dr_forward(bt[:20001], bt[:4999]), 1e8)
from d3d_amd.box import iou2d_forward
show("iou2d box 20k x 100k fp64", lambda: iou2d_forward(bt[:20000], bt), 2e9, 5)
show("iou2d box 20k x 5001 fp32", lambda: iou2d_forward(bf[:20000], bf[:5001]), 1.0002e8)
show("iou2d box 20k x 5000 fp32", lambda: iou2d_forward(bf[:20000], bf[:5000]), 1e8)
