"""where the HOST time of one sparse + trim VoxelGenerator call goes: cProfile over a stream of calls (development aid).
usage (GPU box): python tools/host_profile_sparse.py [calls]"""
import cProfile
import pstats
import sys
import time
sys.path.insert(0, ".")
import torch
from d3d_amd import synth
from d3d_amd.voxel import VoxelGenerator

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = 1000000
pts = torch.from_numpy(synth.lidar_like(n, 0)).cuda()
gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, max_points=32, max_points_filter="trim")
for _ in range(20):
    gen(pts)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    gen(pts)
torch.cuda.synchronize()
print("%.1f us per call" % ((time.perf_counter() - t0) / calls * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(calls):
    gen(pts)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(25)
