#!/usr/bin/env python3
"""per-kernel HIP-event timings of the sparse contract + filter (the reference's default VoxelGenerator mode) (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import kernel_profile, timed
from d3d_amd import synth
from d3d_amd.voxel import VoxelGenerator

def run(tag, cloud, **kw):
    gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, **kw)
    pts = torch.from_numpy(cloud).cuda()
    dt = timed(lambda: gen(pts), 10, 3)
    prof = kernel_profile(lambda: gen(pts), 10)
    print(tag, "%.1f us/step  %.0f Mpts/s" % (dt / 10 * 1e6, len(cloud) * 10 / dt / 1e6),
          {k: round(v["avg_us"] * v.get("calls", 10) / 10, 1) if "calls" in v else round(v["avg_us"], 1)
           for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])})

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
L = synth.lidar_like(n, 0)
run("sparse trim P32", L, max_points=32, max_points_filter="trim")
run("sparse none", L, max_points=32)
run("sparse trim P5 maxvox trim 20000", L, max_points=5, max_points_filter="trim", max_voxels=20000, max_voxels_filter="trim")
run("sparse trim P32 maxvox descending", L, max_points=32, max_points_filter="trim", max_voxels=200000, max_voxels_filter="descending")
