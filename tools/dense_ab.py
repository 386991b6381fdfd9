#!/usr/bin/env python3
"""A/B of the dense operator's output stage (development aid): rows staged + two launches (k_meta_first + k_fill_c4) against
the fused k_emit.  Two different clouds alternate.  usage: dense_ab.py [points] [waymo]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import kernel_profile, timed
from d3d_amd import _lib, synth
from d3d_amd.voxel import VoxelGenerator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
big = len(sys.argv) > 2 and sys.argv[2] == "waymo"
bounds, shape = (synth.WAYMO_BOUNDS, synth.WAYMO_SHAPE) if big else (synth.KITTI_BOUNDS, synth.KITTI_SHAPE)
A = torch.from_numpy(synth.lidar_like(n, 3 if big else 0, bounds)).cuda()
B = torch.from_numpy(synth.lidar_like(n, 4 if big else 1, bounds)).cuda()


def run(tag, flags, steps=20):
    gen = VoxelGenerator(bounds, shape, dense=True, max_voxels=n, reduction="mean", max_points=32)
    k = [0]

    def step():
        k[0] += 1
        return gen(A if k[0] & 1 else B, flags=flags)
    dt = timed(step, steps, 4)
    prof = kernel_profile(step, steps)
    print("%-34s %7.1f us/step  sum %.1f " % (tag, dt / steps * 1e6, sum(v["avg_us"] for v in prof.values())),
          {k: round(v["avg_us"], 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}, flush=True)


lib = _lib.load()
run("split (k_meta_first + k_fill_c4)", _lib.VOXEL_SPLIT_FILL)
run("emit", 0)
if hasattr(lib, "d3d_internal_emit_dbg"):
    for m, what in ((4, "no per-voxel outputs"), (32, "per-voxel outputs by meta_voxel (plain stores)"), (14, "zeros stretch only")):
        lib.d3d_internal_emit_dbg(m)
        run("emit dbg %d: %s" % (m, what), 0)
    lib.d3d_internal_emit_dbg(0)
run("emit again", 0)
