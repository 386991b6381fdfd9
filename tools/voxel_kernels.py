"""per-kernel HIP-event times of the dense voxelizer at config 2 (1 M points) and config 5's frame on one GPU (8 M points).
usage (GPU box): python tools/voxel_kernels.py [steps]"""
import sys
import torch
sys.path.insert(0, ".")
import bench
from d3d_amd import synth
from d3d_amd.voxel import VoxelGenerator

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for name, n, seed, bounds, shape in (("config2", 1000000, 0, synth.KITTI_BOUNDS, synth.KITTI_SHAPE),
                                     ("config5-1gpu", 8000000, 3, synth.WAYMO_BOUNDS, synth.WAYMO_SHAPE)):
    cloud = torch.from_numpy(synth.lidar_like(n, seed, bounds)).cuda()
    gen = VoxelGenerator(bounds, shape, dense=True, reduction="mean", max_points=32, max_voxels=n)
    v = int(gen(cloud).coords.shape[0])
    dt = bench.timed(lambda: gen(cloud), steps, 3)
    prof = bench.kernel_profile(lambda: gen(cloud), steps)
    print("%s: V=%d  %.1f us per call" % (name, v, 1e6 * dt / steps))
    for k, p in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
        print("   %-18s %8.2f us" % (k, p["avg_us"]))
    del cloud, gen
    torch.cuda.empty_cache()
