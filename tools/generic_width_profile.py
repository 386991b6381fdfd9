#!/usr/bin/env python3
"""dense voxelizer on rows of C != 4 columns (config 2's cloud with extra feature columns): per-kernel HIP-event timings of the
one-launch output kernel for C-float rows against the generic kernels (flag VOXEL_SPLIT_FILL) (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import kernel_profile, timed
from d3d_amd import _lib, synth
from d3d_amd.voxel import VoxelGenerator

base = synth.lidar_like(1000000, 0)
for c in (4, 5, 8, 3):
    rng = np.random.default_rng(c)
    cloud = base[:, :c] if c <= 4 else np.concatenate([base, rng.random((len(base), c - 4), dtype=np.float32)], 1)
    pts = torch.from_numpy(np.ascontiguousarray(cloud)).cuda()
    for tag, fl in (("emit", 0), ("generic kernels", _lib.VOXEL_SPLIT_FILL)):
        gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, max_points=32, max_voxels=1000000, dense=True, reduction="mean")
        dt = timed(lambda: gen(pts, flags=fl), 10, 3)
        prof = kernel_profile(lambda: gen(pts, flags=fl), 10)
        print("C = %d [%s]: %.1f us/step" % (c, tag, dt / 10 * 1e6),
              {k: round(v["avg_us"], 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])})
