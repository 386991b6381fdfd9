#!/usr/bin/env python3
"""dense voxelizer on the same cloud in random order, in scan-line-like order (sorted by azimuth) and fully sorted by
voxel: real LiDAR frames arrive ordered, so neighbouring lanes often hit the same voxel (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import kernel_profile, timed
from d3d_amd import synth
from d3d_amd.voxel import VoxelGenerator

L = synth.lidar_like(1000000, 0)
az = np.arctan2(L[:, 1], L[:, 0])
key = (np.floor(L[:, 0] / 0.1) * 800 + np.floor((L[:, 1] + 40) / 0.1)) * 40 + np.floor((L[:, 2] + 3) / 0.1)
for tag, order in [("random", np.arange(len(L))), ("by azimuth", np.argsort(az, kind="stable")), ("by voxel", np.argsort(key, kind="stable"))]:
    pts = torch.from_numpy(np.ascontiguousarray(L[order])).cuda()
    for dense in (True, False):
        gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, dense=dense, max_voxels=len(L), max_points=32,
                             **(dict(reduction="mean") if dense else dict(max_points_filter="trim")))
        dt = timed(lambda: gen(pts), 10, 3)
        prof = kernel_profile(lambda: gen(pts), 5)
        print("%-11s %-6s %.1f us/step" % (tag, "dense" if dense else "sparse", dt / 10 * 1e6),
              {k: round(v["avg_us"], 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:5]})
