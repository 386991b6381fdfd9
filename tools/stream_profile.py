"""VoxelGenerator.stream (pipelined frames) against the plain per-frame loop, and where the host's time goes (cProfile).
usage (GPU box): python tools/stream_profile.py"""
import cProfile
import pstats
import sys
import time
import torch
sys.path.insert(0, ".")
from d3d_amd import synth
from d3d_amd.voxel import VoxelGenerator

n, nfr = 1000000, 60
ca = torch.from_numpy(synth.lidar_like(n, 0)).cuda()
cb = torch.from_numpy(synth.lidar_like(n, 1)).cuda()
gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, dense=True, reduction="mean", max_points=32, max_voxels=n)


def frames():
    return ((ca if k & 1 else cb) for k in range(nfr))


def run_stream():
    return sum(r.coords.shape[0] for r in gen.stream(frames(), pipelined=True))


def run_loop():
    return sum(gen(f).coords.shape[0] for f in frames())


for name, fn in (("loop", run_loop), ("stream", run_stream), ("loop", run_loop), ("stream", run_stream)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    print("%-7s %.1f us per frame" % (name, 1e6 * (time.perf_counter() - t0) / nfr), flush=True)
pr = cProfile.Profile()
pr.enable()
run_stream()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
