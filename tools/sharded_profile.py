#!/usr/bin/env python3
"""per-kernel HIP-event timings of the sharded voxelizer's COMPUTE at world size W, emulated on one GPU with W virtual
ranks (threads; collectives through shared memory) -- what every rank runs besides the RCCL collectives (development aid)"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from bench import kernel_profile
from d3d_amd import synth
from d3d_amd.voxel.sharded import HipOps, ShardedVoxelGenerator
from sharded_helpers import LockedOps, ThreadWorld

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
cfg5 = len(sys.argv) > 3 and sys.argv[3] == "config5"        # config 5: Waymo range, 0.05 m voxels
exchange = sys.argv[4] if len(sys.argv) > 4 else "owner"      # owner | auto | keys | bitmap
replicate = not (len(sys.argv) > 5 and sys.argv[5] == "noreplicate")
max_points = int(sys.argv[6]) if len(sys.argv) > 6 else None   # with the dense contract (owner exchange)
BOUNDS, SHAPE = (synth.WAYMO_BOUNDS, synth.WAYMO_SHAPE) if cfg5 else (synth.KITTI_BOUNDS, synth.KITTI_SHAPE)
if cfg5:
    frame = synth.lidar_like(W * n, 3, BOUNDS)
    clouds = [torch.from_numpy(frame[r * n:(r + 1) * n].copy()).cuda() for r in range(W)]
else:
    clouds = [torch.from_numpy(synth.lidar_like(n, r)).cuda() for r in range(W)]

def step():
    tw, lock = ThreadWorld(W), threading.Lock()
    outs = [None] * W
    global stats
    stats = [None] * W
    def run(rank):
        torch.cuda.set_device(0)
        gen = ShardedVoxelGenerator(BOUNDS, SHAPE, reduction="mean", comm=tw.comm(rank), ops=LockedOps(HipOps(), lock),
                                    exchange=exchange, replicate=replicate, max_points=max_points)
        outs[rank] = gen(clouds[rank])
        stats[rank] = gen.last_stats
    ts = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    [t.start() for t in ts]; [t.join() for t in ts]
    return outs

outs = step()
print("world", W, "points/rank", n, "exchange", exchange, "replicate", replicate, "max_points", max_points, "global voxels", stats[0]["voxels"])
print("  collective bytes of rank 0:", {k: v for k, v in stats[0].items() if "bytes" in k})
prof = kernel_profile(step, 3)
tot = 0.0
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
    per_rank = v["total_ms"] / 3 / W * 1e3
    tot += per_rank
    print("  %-36s %8.1f us per rank and step" % (k, per_rank))
print("  total kernel time per rank and step: %.1f us (collectives not included)" % tot)
