#!/usr/bin/env python3
"""A/B of the dense operator's output stage at config 2 (development aid): two-launch form (k_meta_first + k_fill_c4),
fused k_emit without a stream state, k_emit with the stream state (speculative pre-zeroing) for a few splits of the zero
stores over the index launches.  Two different clouds alternate, so the speculation is never exact."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import kernel_profile, timed
from d3d_amd import _lib, synth, voxel
from d3d_amd.voxel import VoxelGenerator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
big = len(sys.argv) > 2 and sys.argv[2] == "waymo"
bounds, shape = (synth.WAYMO_BOUNDS, synth.WAYMO_SHAPE) if big else (synth.KITTI_BOUNDS, synth.KITTI_SHAPE)
A = torch.from_numpy(synth.lidar_like(n, 3 if big else 0, bounds)).cuda()
B = torch.from_numpy(synth.lidar_like(n, 4 if big else 1, bounds)).cuda()
lib = _lib.load()
lib.d3d_internal_set_prezero_shares.argtypes = [ctypes.c_uint32] * 4


def run(tag, flags, use_state, shares=None, steps=20):
    voxel.default_flags = flags
    if shares:
        assert lib.d3d_internal_set_prezero_shares(*shares) == 0
    gen = VoxelGenerator(bounds, shape, dense=True, max_voxels=n, reduction="mean", max_points=32)
    if not use_state:
        gen._speculate = False
    k = [0]

    def step():
        k[0] += 1
        return gen(A if k[0] & 1 else B)
    dt = timed(step, steps, 4)
    prof = kernel_profile(step, steps)
    print("%-34s %7.1f us/step  sum %.1f " % (tag, dt / steps * 1e6, sum(v["avg_us"] for v in prof.values())),
          {k: round(v["avg_us"], 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}, flush=True)


run("split (k_meta_first + k_fill_c4)", _lib.VOXEL_SPLIT_FILL, False)
run("emit, no state", 0, False)
for sh in ((20, 44, 90, 236), (0, 0, 0, 256), (0, 30, 60, 230), (30, 70, 120, 230), (10, 60, 90, 200), (40, 100, 160, 226),
           (0, 128, 128, 128), (0, 60, 60, 200)):
    run("emit + state, cuts %s" % (sh,), 0, True, sh)
run("split again", _lib.VOXEL_SPLIT_FILL, False)
