#!/usr/bin/env python3
"""seeded fuzzing of the point-sharded voxelizer end to end: K virtual ranks as threads on one GPU with the real kernels (the
collectives on the host, tests/sharded_helpers.ThreadWorld), random worlds, ragged and empty shards, clouds from sparse to a few
dense cells, all reductions, with and without the dense contract, replicated or not -- every rank's result against the oracle's
grid of the whole frame (tests/test_sharded.py's checks).
python tests/sharded_fuzz.py [first_seed] [count]"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_sharded as ts
from sharded_helpers import LockedOps, ThreadWorld
from d3d_amd import _lib
from d3d_amd.voxel.sharded import HipOps, ShardedVoxelGenerator

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = checked = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    world = int(rng.choice([1, 2, 3, 5, 8]))
    n = int(rng.choice([50, 3000, 40000, 150000]))
    reduction = str(rng.choice(["mean", "max", "min"]))
    P = int(rng.choice([0, 0, 1, 4, 32, 70]))
    replicate = bool(rng.integers(0, 2)) and P == 0
    mflags = int(rng.choice([0, 0, _lib.OWNER_MERGE_CHAINS, _lib.OWNER_MERGE_TEST_TINY]))
    cloud = ts._cloud(n, seed)
    if seed % 3 == 0:                                   # a dense blob: rows of several ranks compete for a voxel's slots
        k = max(n // 5, 1)
        cloud[:k, :3] = cloud[:k, :3] * 0.02 + np.array([30, 0, -1], np.float32)
        cloud = cloud[rng.permutation(n)]
    cuts = np.sort(rng.integers(0, n + 1, world - 1)).tolist() if world > 1 else []
    cuts = [0] + cuts + [n]
    # the dense contract into resident buffers (half of the eligible seeds): a frame with the points in reverse order first --
    # other owners, other ids, rows left in every rank's buffer -- then the frame that is checked
    resident = P > 0 and not replicate and bool(np.random.default_rng(seed + 10 ** 6).integers(0, 2))
    before = cloud[::-1].copy()
    tw, lock = ThreadWorld(world), threading.Lock()
    out, errs = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            gen = ShardedVoxelGenerator(ts.BOUNDS, ts.SHAPE, reduction=reduction, comm=tw.comm(rank), exchange="owner", replicate=replicate,
                                        ops=LockedOps(HipOps(), lock), max_points=P or None, merge_flags=mflags, resident=resident)
            if resident:
                gen(torch.from_numpy(before[cuts[rank]:cuts[rank + 1]]).cuda())
            out[rank] = gen(torch.from_numpy(cloud[cuts[rank]:cuts[rank + 1]]).cuda())
        except Exception:
            import traceback
            errs.append(traceback.format_exc())
            tw.barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    tag = "seed %d world %d n %d %s P %d replicate %s resident %s merge %d cuts %s" % (seed, world, n, reduction, P, replicate, resident, mflags, cuts)
    if errs:
        bad += 1; print("SHARDED", tag, "RAISED", errs[0][-400:]); continue
    try:
        ids = []
        for r in range(world):
            sl = slice(cuts[r], cuts[r + 1])
            if replicate:
                exp = ts._expected(cloud, reduction)
                assert np.array_equal(out[r].coords.cpu().numpy(), exp["coords"]) and np.array_equal(out[r].voxel_npoints.cpu().numpy(), exp["voxel_npoints"])
                if reduction == "mean":
                    np.testing.assert_allclose(out[r].aggregates.cpu().numpy(), exp["aggregates"], rtol=1e-5, atol=1e-6)
                else:
                    assert np.array_equal(out[r].aggregates.cpu().numpy(), exp["aggregates"])
            else:
                ids.append(ts._check_owned(out[r], cloud, sl, reduction, max_points=P))
        if not replicate:
            assert np.array_equal(np.sort(np.concatenate(ids)), np.arange(out[0].num_voxels))
        checked += int(out[0].num_voxels) if not replicate else int(out[0].coords.shape[0])
    except AssertionError as e:
        bad += 1; print("SHARDED", tag, "FAILED", str(e)[:300])
print("sharded fuzz: %d seeds, %d failures (%d voxels checked against the oracle)" % (count, bad, checked))
