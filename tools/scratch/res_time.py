import sys
sys.path.insert(0, "/root/repo")
import torch
from bench import kernel_profile, timed
from d3d_amd import synth
from d3d_amd.voxel import VoxelGenerator
frames = [torch.from_numpy(synth.lidar_like(1000000, s)).cuda() for s in range(4)]
for res in (False, True):
    gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, max_points=32, max_voxels=1000000, reduction="mean", dense=True, resident=res)
    for same in (True, False):
        k = [0]
        def step():
            f = frames[0] if same else frames[k[0] % 4]
            k[0] += 1
            return gen(f)
        for rep in range(3):
            dt = timed(step, 50, 5)
            print("resident" if res else "plain   ", "same frame" if same else "4 frames  ", "%.1f us/step" % (dt / 50 * 1e6))
        prof = kernel_profile(step, 12)
        print({k2: round(v["avg_us"], 1) for k2, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])})
