"""A/B of experiment knobs (make -C d3d_amd/csrc TUNE=1 -> libd3d_hip_tune.so; voxel.hip D3D_TUNE_VAL) on ONE box in ONE process:
every setting alternates with the default, per-kernel HIP-event times, and its outputs are compared with the default's bit for bit.
usage (GPU box): python tools/tune_ab.py steps n "name:k=v,k=v" ..."""
import os
import sys
import torch
sys.path.insert(0, ".")
from d3d_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libd3d_hip_tune.so")
import bench
from d3d_amd import synth
from d3d_amd.voxel import VoxelGenerator

steps, n = int(sys.argv[1]), int(sys.argv[2])
variants = [("default", {})]
for a in sys.argv[3:]:
    name, kv = a.split(":")
    variants.append((name, {int(k): int(v) for k, v in (x.split("=") for x in kv.split(","))}))
lib = _lib.load()


def apply(kv):
    for k in range(16):
        lib.d3d_debug_set_tune(k, -1)
    for k, v in kv.items():
        lib.d3d_debug_set_tune(k, v)


big = n > 2000000
bounds, shape, seed = (synth.WAYMO_BOUNDS, synth.WAYMO_SHAPE, 3) if big else (synth.KITTI_BOUNDS, synth.KITTI_SHAPE, 0)
make_cloud = synth.uniform_cloud if os.environ.get("TUNE_DIST", "lidar") == "uniform" else synth.lidar_like
cloud = torch.from_numpy(make_cloud(n, seed, bounds)).cuda()
mode = os.environ.get("TUNE_MODE", "dense")
if mode == "dense":
    gen = VoxelGenerator(bounds, shape, dense=True, reduction=os.environ.get("TUNE_RED", "mean") or None, max_points=32, max_voxels=n)
else:
    gen = VoxelGenerator(bounds, shape, max_points=32, max_voxels=n, max_points_filter="trim")
apply({})
ref = {k: v.clone() for k, v in gen(cloud).items()}
for rep in range(3):
    for name, kv in variants:
        apply(kv)
        got = gen(cloud, poison=True) if mode == 'dense' else gen(cloud)
        same = all(torch.equal(got[k], ref[k]) for k in ref)
        dt = bench.timed(lambda: gen(cloud), steps, 3)
        prof = bench.kernel_profile(lambda: gen(cloud), steps)
        ks = sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])
        print("n=%d %-6s %-16s %s %7.1f us/call | " % (n, mode, name, "same" if same else "DIFF", 1e6 * dt / steps) +
              " ".join("%s %.1f" % (k.replace("k_", ""), p["avg_us"]) for k, p in ks[:8]), flush=True)
