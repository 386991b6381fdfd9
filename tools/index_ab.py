"""A/B of the binned index's bucket kernel on ONE box in ONE call: round 5's (packed first-point entries, dense cells ranked by one
wavefront) against round 4's (D3D_VOXEL_INDEX_V1), alternating, per-kernel HIP-event times.
usage (GPU box): python tools/index_ab.py [steps] [n ...]"""
import os
import sys
import torch
sys.path.insert(0, ".")
import bench
from d3d_amd import _lib, synth
from d3d_amd.voxel import VoxelGenerator

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sizes = [int(x) for x in sys.argv[2:]] or [1000000]
for n in sizes:
    big = n > 2000000
    bounds, shape, seed = (synth.WAYMO_BOUNDS, synth.WAYMO_SHAPE, 3) if big else (synth.KITTI_BOUNDS, synth.KITTI_SHAPE, 0)
    cloud = torch.from_numpy(synth.lidar_like(n, seed, bounds)).cuda()
    gens = {"dense": VoxelGenerator(bounds, shape, dense=True, reduction="mean", max_points=32, max_voxels=n),
            "dense-none": VoxelGenerator(bounds, shape, dense=True, max_points=32, max_voxels=n),
            "sparse+trim": VoxelGenerator(bounds, shape, max_points=32, max_voxels=n, max_points_filter="trim"),
            "sparse": VoxelGenerator(bounds, shape, max_voxels=n)}
    only = os.environ.get("AB_MODES")
    for mode, gen in gens.items():
        if only and mode not in only.split(","):
            continue
        for rep in range(3):
            for name, fl in (("r5", 0), ("v1", _lib.VOXEL_INDEX_V1)):
                dt = bench.timed(lambda: gen(cloud, flags=fl), steps, 3)
                prof = bench.kernel_profile(lambda: gen(cloud, flags=fl), steps)
                ks = sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])
                print("n=%d %-11s %-4s %7.1f us/call | " % (n, mode, name, 1e6 * dt / steps) +
                      " ".join("%s %.1f" % (k.replace("k_", ""), p["avg_us"]) for k, p in ks[:9]), flush=True)
    del cloud, gens
    torch.cuda.empty_cache()
