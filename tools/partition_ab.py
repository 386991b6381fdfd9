"""A/B of the binned index's partition on ONE box in ONE call: the one-launch tile sort (round 4) against the three-pass
partition (D3D_VOXEL_PARTITION_3PASS), alternating, per-kernel HIP-event times.
usage (GPU box): python tools/partition_ab.py [steps] [n ...]"""
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
            "sparse+trim": VoxelGenerator(bounds, shape, max_points=32, max_voxels=n, max_points_filter="trim")}
    for mode, gen in gens.items():
        for rep in range(2):
            for name, fl in (("tile-sort", 0), ("3-pass", _lib.VOXEL_PARTITION_3PASS)):
                dt = bench.timed(lambda: gen(cloud, flags=fl), steps, 3)
                prof = bench.kernel_profile(lambda: gen(cloud, flags=fl), steps)
                ks = sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])
                print("n=%d %-11s %-9s %7.1f us/call | " % (n, mode, name, 1e6 * dt / steps) +
                      " ".join("%s %.1f" % (k.replace("k_", ""), p["avg_us"]) for k, p in ks[:9]), flush=True)
    del cloud, gens
    torch.cuda.empty_cache()
