#!/usr/bin/env python3
"""timings of the "next" rows: IoU backward, point-in-box crop, aligned_scatter (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import kernel_profile, timed
from d3d_amd import synth
from d3d_amd.box import iou2dr_backward, iou2d_backward, crop_2dr
from d3d_amd.point import aligned_scatter_forward, aligned_scatter_backward, AlignType

def show(tag, f, units, reps=10):
    dt = timed(f, reps, 2)
    prof = kernel_profile(f, reps)
    print(tag, "%.1f us/call  %.2f G/s" % (dt / reps * 1e6, units * reps / dt / 1e9),
          {k: round(v["avg_us"], 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])})

b, _ = synth.boxes2d_sparse(100000, 1)
bt = torch.from_numpy(b).cuda()
g = torch.rand(20000, 20000, dtype=torch.float64, device="cuda")
show("iou2dr_backward 20k x 20k fp64 (config-3 density)", lambda: iou2dr_backward(bt[:20000], bt[20000:40000], g), 4e8, 5)
show("iou2d_backward  20k x 20k fp64", lambda: iou2d_backward(bt[:20000], bt[20000:40000], g), 4e8, 5)
bd, _ = synth.boxes2d_dense(3000, 1)
bdt = torch.from_numpy(bd).cuda()
gd = torch.rand(3000, 3000, dtype=torch.float64, device="cuda")
show("iou2dr_backward dense 3k x 3k fp64", lambda: iou2dr_backward(bdt, bdt, gd), 9e6, 5)
pts = torch.rand(1000000, 2, device="cuda") * 100
boxes = torch.rand(2000, 5, device="cuda") * torch.tensor([100, 100, 5, 5, 3.14], device="cuda")
show("crop_2dr 2000 boxes x 1M points fp32", lambda: crop_2dr(pts, boxes), 2e9)
from d3d_amd.abstraction import crop_points, paint_label
cloud = torch.cat([torch.rand(1000000, 2, device="cuda") * 100, torch.rand(1000000, 1, device="cuda") * 4 - 2,
                   torch.rand(1000000, 1, device="cuda")], 1)
b3 = torch.cat([torch.rand(2000, 2, device="cuda") * 100, torch.rand(2000, 1, device="cuda") * 2 - 1,
                torch.rand(2000, 3, device="cuda") * 4 + 1, torch.rand(2000, 1, device="cuda") * 6.28], 1)
show("crop_points 2000 boxes x 1M points (bool[M,N])", lambda: crop_points(b3, cloud), 2e9)
sem = torch.randint(0, 4, (1000000,), device="cuda", dtype=torch.uint8)
lab = torch.randint(0, 4, (2000,), device="cuda", dtype=torch.uint8)
show("paint_label 2000 boxes x 1M points (uint16[N])", lambda: paint_label(b3, cloud, sem, lab), 2e9)
from d3d_amd.box import pdist2dr_forward, box2d_iou
show("pdist2dr_forward 2000 boxes x 1M points fp32 (10 GB out)", lambda: pdist2dr_forward(pts, boxes), 2e9, 5)
bg = bt[:10000].contiguous()
show("giou (grbox) forward 10k x 10k fp64", lambda: box2d_iou(bg, bg, method="grbox"), 1e8, 3)
img = torch.rand(2, 64, 200, 176, device="cuda")
coord = torch.cat([torch.randint(0, 2, (500000, 1), device="cuda").float(), torch.rand(500000, 1, device="cuda") * 199,
                   torch.rand(500000, 1, device="cuda") * 175], 1)
for at in (AlignType.MEAN, AlignType.LINEAR):
    show("aligned_scatter_forward %s 500k pts x 64 ch" % at.name, lambda: aligned_scatter_forward(coord, img, at), 500000 * 64)
    go = torch.rand(500000, 64, device="cuda")
    gi = torch.zeros_like(img)
    show("aligned_scatter_backward %s" % at.name, lambda: aligned_scatter_backward(coord, go, at, gi), 500000 * 64)
