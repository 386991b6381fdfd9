#!/usr/bin/env python3
"""per-kernel HIP-event timings of the dense voxelizer for a few configurations (development aid)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import kernel_profile, timed
from d3d_amd import synth
from d3d_amd.voxel import VoxelGenerator

def run(tag, cloud, **kw):
    gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, dense=True, max_voxels=len(cloud), **kw)
    pts = torch.from_numpy(cloud).cuda()
    dt = timed(lambda: gen(pts), 10, 3)
    prof = kernel_profile(lambda: gen(pts), 10)
    print(tag, "%.1f us/step" % (dt / 10 * 1e6), {k: round(v["avg_us"], 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])})

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
L = synth.lidar_like(n, 0)
run("lidar mean P32", L, reduction="mean", max_points=32)
run("lidar none P32", L, reduction=None, max_points=32)
run("lidar max  P32", L, reduction="max", max_points=32)
run("lidar mean P30", L, reduction="mean", max_points=30)
run("lidar mean P5 ", L, reduction="mean", max_points=5)
U = synth.uniform_cloud(n, 0)
run("unif  mean P32", U, reduction="mean", max_points=32)
if len(sys.argv) > 2:
    n2 = int(sys.argv[2])
    from d3d_amd.synth import WAYMO_BOUNDS, WAYMO_SHAPE
    big = synth.lidar_like(n2, 3, WAYMO_BOUNDS)
    gen = VoxelGenerator(WAYMO_BOUNDS, WAYMO_SHAPE, dense=True, max_voxels=n2, reduction="mean", max_points=32)
    pts = torch.from_numpy(big).cuda()
    dt = timed(lambda: gen(pts), 5, 2)
    prof = kernel_profile(lambda: gen(pts), 5)
    r = gen(pts)
    print("waymo %d pts -> %d voxels" % (n2, r.coords.shape[0]), "%.1f us/step  %.1f Mpts/s" % (dt / 5 * 1e6, n2 * 5 / dt / 1e6),
          {k: round(v["avg_us"], 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])})
