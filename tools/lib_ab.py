"""one build of the library against another on ONE box: each (library, frame size) in a process of its own, alternating.
usage (GPU box): python tools/lib_ab.py libd3d_hip.so libd3d_hip_tune.so [n ...]   (files in d3d_amd/)"""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, torch
sys.path.insert(0, ".")
from d3d_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[1])
import bench
from d3d_amd import synth
from d3d_amd.voxel import VoxelGenerator
n = int(sys.argv[2])
big = n > 2000000
bounds, shape, seed = (synth.WAYMO_BOUNDS, synth.WAYMO_SHAPE, 3) if big else (synth.KITTI_BOUNDS, synth.KITTI_SHAPE, 0)
cloud = torch.from_numpy(synth.lidar_like(n, seed, bounds)).cuda()
gen = VoxelGenerator(bounds, shape, dense=True, reduction="mean", max_points=32, max_voxels=n)
steps = 200 if n <= 2000000 else 30
gen(cloud)
dt = bench.timed(lambda: gen(cloud), steps, 5)
prof = bench.kernel_profile(lambda: gen(cloud), steps)
print("%-22s n=%d %8.1f us/call | " % (sys.argv[1], n, 1e6 * dt / steps) + " ".join("%s %.1f" % (k.replace("k_", ""), p["avg_us"]) for k, p in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])), flush=True)
'''
libs = sys.argv[1:3]
sizes = [int(x) for x in sys.argv[3:]] or [1000000]
for n in sizes:
    for rep in range(3):
        for lib in libs:
            subprocess.run([sys.executable, "-c", CHILD, lib, str(n)], env=dict(os.environ), check=False)
