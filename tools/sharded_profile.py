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

resident = len(sys.argv) > 7 and sys.argv[7] == "resident"    # dense contract into buffers the generators keep (replicate off)
tw, lock = ThreadWorld(W), threading.Lock()
gens = [ShardedVoxelGenerator(BOUNDS, SHAPE, reduction="mean", comm=tw.comm(r), ops=LockedOps(HipOps(), lock), exchange=exchange,
                              replicate=replicate, max_points=max_points, resident=resident) for r in range(W)]

def step():
    outs = [None] * W
    global stats
    stats = [None] * W
    def run(rank):
        torch.cuda.set_device(0)
        gen = gens[rank]
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
# (round 6) every figure above is an EVENT PAIR around a launch, which also measures the wait for the launch before it and the
# dispatch: an empty launch comes out at ~6 us that way (bench.event_overhead_us, measured here the same way), and a step is 15 or
# more launches per rank -- while the single-GPU bases below are wall-clock.  Net of that, the kernels of a rank:
import bench as _bench
launches = sum(v["calls"] for v in prof.values()) / 3 / W
ov = _bench.event_overhead_us()
tot_events = tot
GAP_US = 2.0          # what stays between two dependent launches on one stream (an ASSUMPTION like the link constants below)
tot = tot - launches * ov + launches * GAP_US
print("  %.1f launches per rank and step x %.2f us of event-pair overhead each, %.1f us of gap put back per launch -> %.1f us per rank "
      "(the modelled step below uses this; the event sum was %.1f)" % (launches, ov, GAP_US, tot, tot_events))

# ---- modelled step time on 8 MI355X over xGMI: what one GPU cannot measure, priced with public figures --
# every rank has 7 links of ~153 GB/s to its 7 peers, a direct all-to-all keeps all of them busy; a collective costs a launch +
# rendezvous latency on top (RCCL on MI300-class nodes: ~25 us for small messages); every host read-back a stream drain (~20 us)
LINK_GBS, LINKS, COLL_LAT_US, HOST_SYNC_US = 153.0, 7, 25.0, 20.0
st0 = stats[0]
a2a = max(st0.get("all_to_all_bytes_sent", 0), st0.get("all_to_all_bytes_received", 0)) + st0.get("reply_bytes_sent", 0) + \
    st0.get("rows_all_to_all_bytes_sent", 0)
# round 5: the shard sizes ride in the count matrix -- counts, records, bitmap, reply [, rows] [, sizes + gather x2]; two host
# synchronisations (the count matrix, the output sizes) [+ one for the replicated result's sizes]
ncoll = 1 + 1 + 1 + 1 + (1 if max_points else 0) + (3 if replicate else 0)
nsync = 2 + (1 if replicate else 0)
t_wire = a2a / (LINKS * LINK_GBS * 1e3) + 2 * (W - 1) / W * st0.get("all_reduce_bytes", 0) / (LINK_GBS * 1e3) + \
    (W - 1) * st0.get("all_gather_bytes_per_rank", 0) / (LINKS * LINK_GBS * 1e3)
model = tot + t_wire + ncoll * COLL_LAT_US + nsync * HOST_SYNC_US
print("  modelled step at world %d: %.0f us = kernels %.0f + wire %.0f + %d collectives x %.0f + %d host syncs x %.0f  (constants are "
      "ASSUMPTIONS: no multi-GPU box was available to any round)" % (W, model, tot, t_wire, ncoll, COLL_LAT_US, nsync, HOST_SYNC_US))
try:
    from d3d_amd.voxel import VoxelGenerator
    import bench
    whole = torch.cat(clouds)
    if max_points:
        g1 = VoxelGenerator(BOUNDS, SHAPE, dense=True, reduction="mean", max_points=max_points, max_voxels=len(whole), resident=resident)
    else:
        from d3d_amd.voxel.sharded import LocalComm
        g1 = ShardedVoxelGenerator(BOUNDS, SHAPE, reduction="mean", comm=LocalComm(), replicate=False)
    t1 = bench.timed(lambda: g1(whole), 5, 2) / 5 * 1e6
    if max_points:
        print("  single GPU, whole %d-point frame, same contract (VoxelGenerator, dense): %.0f us  ->  predicted speed-up at world %d: %.2fx"
              % (len(whole), t1, W, t1 / model))
    else:
        from d3d_amd.voxel.sharded import voxelize_reduce
        tb = bench.timed(lambda: voxelize_reduce(whole, SHAPE, BOUNDS, "mean"), 5, 2) / 5 * 1e6
        print("  single GPU, whole %d-point frame: best (voxelize_reduce: the contract with NO exchange) %.0f us -> predicted speed-up at "
              "world %d: %.2fx;  the sharded operator at world 1 (pack / merge / number / reply on one rank) %.0f us -> %.2fx"
              % (len(whole), tb, W, tb / model, t1, t1 / model))
except Exception as e:      # pragma: no cover
    print("  (single-GPU base not measured: %r)" % (e,))
