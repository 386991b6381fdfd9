"""Where a workgroup of k_bucket_index / k_tile_sort spends its cycles (diagnostic build: make -C d3d_amd/csrc PHASE_CLOCKS=1
-> d3d_amd/libd3d_hip_phase.so).  Prints mean cycles per workgroup and phase at config 2, dense and sparse + trim.
usage (GPU box): python tools/phase_clocks.py [steps]"""
import ctypes
import os
import sys
import torch
sys.path.insert(0, ".")
from d3d_amd import _lib
KV = os.environ.get("D3D_TUNE_KV", "")              # "k=v,k=v": experiment knobs (make PHASE_TUNE=1 -> libd3d_hip_phase_tune.so)
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libd3d_hip_phase_tune.so" if KV else "libd3d_hip_phase.so")
from d3d_amd import synth
from d3d_amd.voxel import VoxelGenerator

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
lib = _lib.load()
for kv in filter(None, KV.split(",")):
    lib.d3d_debug_set_tune(int(kv.split("=")[0]), int(kv.split("=")[1]))
buf = (ctypes.c_ulonglong * 64)()
names = {0: ("k_bucket_index", ["prologue", "clear+locate+issue", "insert", "records-loop", "segments", "rank+stores", "overflow", "records-scan", "entries-wait"]),
         1: ("k_tile_sort", ["load+clear", "keys+hist", "scan+table", "place", "copy-out"]),
         2: ("k_emit (wave 0 of each WG)", ["prefix+firstmap", "records", "rows", "reduce", "stretch", "per-voxel"])}
n = 1000000
target = int(os.environ.get("D3D_TUNE_BUCKET", "512"))
nb = 1
while nb < 8192 and nb * target < n:
    nb *= 2
cloud = torch.from_numpy(synth.lidar_like(n, 0, synth.KITTI_BOUNDS)).cuda()
for mode, gen in (("dense", VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, dense=True, reduction="mean", max_points=32, max_voxels=n)),
                  ("sparse+trim", VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, max_points=32, max_voxels=n, max_points_filter="trim"))):
    for name, fl in (("tile-sort", 0), ("3-pass", _lib.VOXEL_PARTITION_3PASS)):
        gen(cloud, flags=fl)
        lib.d3d_debug_phase_clocks(buf)
        for _ in range(steps):
            gen(cloud, flags=fl)
        lib.d3d_debug_phase_clocks(buf)
        for k, (kname, phases) in names.items():
            wgs = {0: nb, 1: (n + 8191) // 8192, 2: (n + 16383) // 16384 * 64}[k] * steps
            vals = [buf[k * 16 + p] / wgs for p in range(len(phases))]
            if sum(vals) == 0:
                continue
            print("%-11s %-9s %-15s total %7.0f cycles/WG | " % (mode, name, kname, sum(vals)) +
                  "  ".join("%s %.0f" % (ph, v) for ph, v in zip(phases, vals)), flush=True)
