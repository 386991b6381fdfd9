"""where the host time of one dense voxelizer call goes (development aid)"""
import ctypes, sys, time
sys.path.insert(0, ".")
import torch
from d3d_amd import _lib, synth
from d3d_amd.voxel import voxelize_3d_dense, VoxelGenerator

lib = _lib.load()
n, P = 1000000, 32
pts = torch.from_numpy(synth.lidar_like(n, 0)).cuda()
dev = pts.device


def t(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def allocs():
    torch.empty((n, P, 4), dtype=torch.float32, device=dev)
    torch.empty((n, 3), dtype=torch.int64, device=dev)
    torch.empty((n, P), dtype=torch.uint8, device=dev)
    torch.empty((n,), dtype=torch.int32, device=dev)
    torch.empty((n, 4), dtype=torch.float32, device=dev)
    torch.empty((4,), dtype=torch.int64, device=dev)


counts = torch.zeros((4,), dtype=torch.int64, device=dev)
pinned = torch.empty((4,), dtype=torch.int64, pin_memory=True)


def read_cpu():
    return counts.cpu()


def read_pinned():
    pinned.copy_(counts, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return pinned[0].item()


def read_item():
    return counts[0].item()


print("6 x torch.empty:            %6.1f us" % t(allocs))
print("counts.cpu() (idle GPU):    %6.1f us" % t(read_cpu))
print("pinned copy + stream sync:  %6.1f us" % t(read_pinned))
print("counts[0].item():           %6.1f us" % t(read_item))
gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, dense=True, reduction="mean", max_points=P, max_voxels=n)
print("VoxelGenerator call:        %6.1f us" % t(lambda: gen(pts), 50))
print("voxelize_3d_dense call:     %6.1f us" % t(lambda: voxelize_3d_dense(pts, synth.KITTI_SHAPE, synth.KITTI_BOUNDS, P, n, 1), 50))
