"""Point-sharded voxelizer: (a) CPU, world_size 2 over gloo with oracle-backed compute steps -- checks the
distributed orchestration (offsets, variable-size all-gather, all-reduce, first-seen renumbering);
(b) GPU, K virtual ranks as threads on one device with the real HIP kernels -- checks the kernels of the
sharded path against the single-GPU dense contract and the oracle."""
import os
import socket
import threading

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import oracle
from sharded_helpers import LockedOps, NumpyOps, ThreadWorld

BOUNDS = [0, 70.4, -40, 40, -3, 1]
SHAPE = [88, 100, 4]


def _cloud(n, seed):
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.random(n) * 75 - 2, rng.random(n) * 84 - 42, rng.random(n) * 4.4 - 3.2, rng.random(n)], 1)
    return pts.astype(np.float32)


def _expected(cloud, reduction):
    r = oracle.voxelize_3d_dense(cloud, SHAPE, BOUNDS, 1, len(cloud), reduction)
    return r


def _check(res, cloud_all, my_slice, reduction):
    exp = _expected(cloud_all, reduction)
    assert np.array_equal(res.coords.cpu().numpy(), exp["coords"])
    assert np.array_equal(res.voxel_npoints.cpu().numpy(), exp["voxel_npoints"])
    if reduction == "mean":
        np.testing.assert_allclose(res.aggregates.cpu().numpy(), exp["aggregates"], rtol=1e-5, atol=1e-6)
    else:
        assert np.array_equal(res.aggregates.cpu().numpy(), exp["aggregates"])
    # point -> voxel map of the rank's own points
    m = res.points_mapping.cpu().numpy()
    pts = cloud_all[my_slice]
    inside = m >= 0
    size = np.float32(0.8), np.float32(0.8), np.float32(1.0)
    lo = np.array([0, -40, -3], np.float32)
    cc = ((pts[inside, :3] - lo) / np.array(size, np.float32)).astype(np.int64)
    assert np.array_equal(exp["coords"][m[inside]], cc)
    assert len(m) == 0 or (inside.sum() > 0 and (~inside).sum() > 0)   # in- and out-of-range points (unless the shard is empty)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _check_owned(res, cloud_all, my_slice, reduction, max_points=0):
    """replicate=False: this rank's OWNED voxels, in global id order, equal those rows of the whole frame's grid"""
    exp = _expected(cloud_all, reduction)
    ids = res.voxel_ids.cpu().numpy()
    if max_points:      # the dense contract of the owned voxels, bit for bit
        dense = oracle.voxelize_3d_dense(cloud_all, SHAPE, BOUNDS, max_points, len(cloud_all), reduction)
        assert np.array_equal(res.voxels.cpu().numpy(), dense["voxels"][ids])
        pm = np.arange(max_points)[None, :] < np.minimum(dense["voxel_npoints"][ids], max_points)[:, None]
        assert np.array_equal(res.voxel_pmask.cpu().numpy(), pm)
    assert res.num_voxels == len(exp["coords"])
    assert np.all(np.diff(ids) > 0) if len(ids) > 1 else True
    assert np.array_equal(res.coords.cpu().numpy(), exp["coords"][ids])
    assert np.array_equal(res.voxel_npoints.cpu().numpy(), exp["voxel_npoints"][ids])
    if reduction == "mean":
        np.testing.assert_allclose(res.aggregates.cpu().numpy(), exp["aggregates"][ids], rtol=1e-5, atol=1e-6)
    else:
        assert np.array_equal(res.aggregates.cpu().numpy(), exp["aggregates"][ids])
    m = res.points_mapping.cpu().numpy()
    pts = cloud_all[my_slice]
    inside = m >= 0
    lo = np.array([0, -40, -3], np.float32)
    cc = ((pts[inside, :3] - lo) / np.array([0.8, 0.8, 1.0], np.float32)).astype(np.int64)
    assert np.array_equal(exp["coords"][m[inside]], cc)
    return ids


def _gloo_worker(rank, world, port, reduction, exchange, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from d3d_amd.voxel.sharded import ShardedVoxelGenerator
        cloud = _cloud(3000, 5)
        cuts = [0, 1100, 3000] if world == 2 else np.linspace(0, 3000, world + 1).astype(int).tolist()
        sl = slice(cuts[rank], cuts[rank + 1])
        gen = ShardedVoxelGenerator(BOUNDS, SHAPE, reduction=reduction, ops=NumpyOps(), exchange=exchange)
        res = gen(torch.from_numpy(cloud[sl]))
        _check(res, cloud, sl, reduction)
        if exchange == "owner":     # ... and without the final all-gather: the owned voxels of the ranks partition the grid
            own = ShardedVoxelGenerator(BOUNDS, SHAPE, reduction=reduction, ops=NumpyOps(), exchange="owner", replicate=False,
                                        debug_checks=True)
            ids = _check_owned(own(torch.from_numpy(cloud[sl])), cloud, sl, reduction)
            allids = [None] * world
            dist.all_gather_object(allids, ids.tolist())
            assert sorted(sum(allids, [])) == list(range(len(_expected(cloud, reduction)["coords"])))
            # the dense contract through the same exchange: owned blocks, and the replicated tensor
            for rep in (False, True):
                dg = ShardedVoxelGenerator(BOUNDS, SHAPE, reduction=reduction, ops=NumpyOps(), exchange="owner", replicate=rep,
                                           max_points=3, debug_checks=True)
                dres = dg(torch.from_numpy(cloud[sl]))
                if not rep:
                    _check_owned(dres, cloud, sl, reduction, max_points=3)
                else:
                    _check(dres, cloud, sl, reduction)
                    dense = oracle.voxelize_3d_dense(cloud, SHAPE, BOUNDS, 3, len(cloud), reduction)
                    assert np.array_equal(dres.voxels.numpy(), dense["voxels"])
        # next frame through the SAME generator: only the last rank's shard size changes (rank 0 keeps its 1100 points).
        # Every rank must still enter the same sequence of collectives and see the new offsets / totals.
        cloud2 = _cloud(2600, 6)
        cuts2 = [0, 1100, 2600] if world == 2 else cuts[:-1] + [2600]
        sl2 = slice(cuts2[rank], cuts2[rank + 1])
        res2 = gen(torch.from_numpy(cloud2[sl2]))
        _check(res2, cloud2, sl2, reduction)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("reduction,exchange", [("mean", "owner"), ("min", "owner"), ("mean", "keys"), ("max", "bitmap"), ("mean", "bitmap")])
def test_sharded_orchestration_gloo_world2(reduction, exchange):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, reduction, exchange, q)) for r in range(world)]
    [p.start() for p in procs]
    results = [q.get(timeout=180) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(msg == "ok" for _, msg in results), results


def test_comm_of_the_replicated_grid_protocol_keeps_working():
    """a caller's comm that implements only the round 1-2 protocol (all_gather_int / all_gather_var / all_reduce)
    still constructs with the default arguments -- the generator then stays on the replicated-grid exchange -- and asking
    for what needs the owner-computes collectives is a TypeError at construction"""
    from d3d_amd.voxel.sharded import ShardedVoxelGenerator

    class OldComm:
        rank, world = 0, 1

        def all_gather_int(self, v, dev):
            return [int(v)]

        def all_gather_var(self, t, sizes):
            return t

        def all_reduce(self, t, op):
            return t
    gen = ShardedVoxelGenerator(BOUNDS, SHAPE, comm=OldComm(), ops=NumpyOps())
    assert gen._exchange == "auto"
    cloud = _cloud(500, 9)
    _check(gen(torch.from_numpy(cloud)), cloud, slice(0, 500), "mean")
    for kw in (dict(replicate=False), dict(max_points=4)):
        with pytest.raises(TypeError):
            ShardedVoxelGenerator(BOUNDS, SHAPE, comm=OldComm(), ops=NumpyOps(), **kw)


@pytest.fixture(params=["auto", "hash"])
def index_path(request):
    """both index paths of the per-rank voxelizer (automatic = binned whenever eligible, hash table)"""
    from d3d_amd import _lib
    from call_opts import set_opts
    set_opts(voxel_flags=_lib.VOXEL_PATH_HASH if request.param == "hash" else 0)
    yield request.param


@pytest.mark.gpu
@pytest.mark.parametrize("world,reduction,exchange", [(2, "mean", "owner"), (3, "min", "owner"), (4, "max", "owner"),
                                                     (8, "mean", "owner"), (3, "mean", "owner-empty"),
                                                     (2, "mean", "keys"), (3, "min", "bitmap"), (4, "max", "keys"),
                                                     (8, "mean", "bitmap"), (2, "mean", "auto"), (3, "mean", "bitmap-empty"),
                                                     (3, "max", "keys-empty")])
def test_sharded_hip_kernels_virtual_ranks(world, reduction, exchange, index_path):
    from d3d_amd.voxel import VoxelGenerator
    from d3d_amd.voxel.sharded import HipOps, ShardedVoxelGenerator
    cloud = _cloud(40000, 9)
    cuts = np.linspace(0, len(cloud), world + 1).astype(int)
    cuts[1] = max(cuts[1] // 3, 1)                      # ragged shards
    if exchange.endswith("-empty"):                     # ... and a rank without any point
        exchange = exchange.split("-")[0]
        cuts[2] = cuts[1]
    tw, lock = ThreadWorld(world), threading.Lock()
    out, errs = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            gen = ShardedVoxelGenerator(BOUNDS, SHAPE, reduction=reduction, comm=tw.comm(rank), exchange=exchange,
                                        ops=LockedOps(HipOps(), lock))
            out[rank] = gen(torch.from_numpy(cloud[cuts[rank]:cuts[rank + 1]]).cuda())
        except Exception as e:  # pragma: no cover
            import traceback
            errs.append(traceback.format_exc())
            tw.barrier.abort()
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[0]
    for r in range(world):
        _check(out[r], cloud, slice(cuts[r], cuts[r + 1]), reduction)
    # identical to the single-GPU dense contract on the concatenated frame
    single = VoxelGenerator(BOUNDS, SHAPE, dense=True, reduction=reduction, max_points=4, max_voxels=len(cloud))(
        torch.from_numpy(cloud).cuda())
    assert torch.equal(single.coords, out[0].coords) and torch.equal(single.voxel_npoints, out[0].voxel_npoints)


@pytest.mark.gpu
@pytest.mark.parametrize("merge", ["buckets", "chains", "tiny buckets"])
@pytest.mark.parametrize("world,reduction,empty,P", [(2, "mean", False, 0), (3, "max", True, 4), (8, "min", False, 32),
                                                      (5, "mean", True, 5), (4, "mean", False, 70)])
def test_owner_computes_without_replication_virtual_ranks(world, reduction, empty, P, merge):
    """replicate=False on the real kernels: every rank returns its owned voxels in global id order; together they are the
    single-GPU grid.  Ragged shards, optionally a rank without points; repeated calls give identical bits (the merge runs in
    rank order whatever the arrival order of the records).  P > 0: with the dense contract (voxels / voxel_pmask of the owned
    voxels, bit-exact against the oracle's dense contract of the whole frame; a dense blob so that candidate rows of several
    ranks compete for a voxel's P slots).  merge: the owner's merge on LDS buckets (default), on the global hash table with
    record chains (the general path), and with buckets that overflow at 4 records (test hook): every rank learns of the
    overflow from the all-reduced status word and they all repeat on the general path."""
    from d3d_amd import _lib
    from d3d_amd.voxel.sharded import HipOps, ShardedVoxelGenerator
    mflags = {"buckets": 0, "chains": _lib.OWNER_MERGE_CHAINS, "tiny buckets": _lib.OWNER_MERGE_TEST_TINY}[merge]
    cloud = _cloud(50000, 19)
    cloud[::7, :3] = cloud[::7, :3] * 0.02 + np.array([30, 0, -1], np.float32)      # ~7000 points in a handful of cells
    cuts = np.linspace(0, len(cloud), world + 1).astype(int)
    cuts[1] = max(cuts[1] // 3, 1)
    if empty:
        cuts[2] = cuts[1]
    tw, lock = ThreadWorld(world), threading.Lock()
    out, errs = [[None] * world, [None] * world], []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            gen = ShardedVoxelGenerator(BOUNDS, SHAPE, reduction=reduction, comm=tw.comm(rank), exchange="owner", replicate=False,
                                        ops=LockedOps(HipOps(), lock), max_points=P or None, merge_flags=mflags)
            for it in range(2):
                out[it][rank] = gen(torch.from_numpy(cloud[cuts[rank]:cuts[rank + 1]]).cuda())
        except Exception:  # pragma: no cover
            import traceback
            errs.append(traceback.format_exc())
            tw.barrier.abort()
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[0]
    ids = []
    for r in range(world):
        ids.append(_check_owned(out[0][r], cloud, slice(cuts[r], cuts[r + 1]), reduction, max_points=P))
        for k in ("coords", "voxel_npoints", "aggregates", "voxel_ids", "points_mapping") + (("voxels", "voxel_pmask") if P else ()):
            assert torch.equal(out[0][r][k], out[1][r][k]), k
    allids = np.concatenate(ids)
    assert np.array_equal(np.sort(allids), np.arange(out[0][0].num_voxels))


@pytest.mark.gpu
@pytest.mark.parametrize("world,P,merge", [(3, 4, "buckets"), (8, 32, "buckets"), (4, 16, "chains")])
def test_owner_dense_resident_over_a_sequence_of_frames(world, P, merge):
    """ShardedVoxelGenerator(..., replicate=False, max_points=P, resident=True): every rank's voxels[Vo, P, 4] is a view of a
    buffer the generator keeps (d3d_owner_dense with row_state: rows with points + stale rows stored, padding kept).  Five
    frames of different size and shape through the same generators -- the owned voxels change completely from frame to
    frame -- each equal to the oracle's dense contract of the whole frame, bit for bit; the buffers' invariant afterwards."""
    from d3d_amd import _lib
    from d3d_amd.voxel.sharded import HipOps, ShardedVoxelGenerator
    mflags = {"buckets": 0, "chains": _lib.OWNER_MERGE_CHAINS}[merge]
    frames = []
    for k, (n, blob) in enumerate([(50000, (30, 0, -1)), (9000, (5, 10, 0)), (80000, (50, -20, -2)), (300, (30, 0, -1)), (40000, (10, 5, 0))]):
        cl = _cloud(n, 100 + k)
        cl[::5, :3] = cl[::5, :3] * 0.03 + np.array(blob, np.float32)
        frames.append(cl)
    tw, lock = ThreadWorld(world), threading.Lock()
    out, errs, gens = [[None] * world for _ in frames], [], [None] * world

    def run(rank):
        try:
            torch.cuda.set_device(0)
            gen = gens[rank] = ShardedVoxelGenerator(BOUNDS, SHAPE, reduction="mean", comm=tw.comm(rank), exchange="owner", replicate=False,
                                                     ops=LockedOps(HipOps(), lock), max_points=P, merge_flags=mflags, resident=True)
            for f, cl in enumerate(frames):
                cuts = np.linspace(0, len(cl), world + 1).astype(int)
                res = gen(torch.from_numpy(cl[cuts[rank]:cuts[rank + 1]]).cuda())
                assert res.voxels.untyped_storage().data_ptr() == gen._resident_buf.voxels.untyped_storage().data_ptr()
                out[f][rank] = type(res)({k: (v.cpu() if hasattr(v, "cpu") else v) for k, v in res.items()})   # before the next frame
        except Exception:  # pragma: no cover
            import traceback
            errs.append(traceback.format_exc())
            tw.barrier.abort()
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[0]
    for f, cl in enumerate(frames):
        cuts = np.linspace(0, len(cl), world + 1).astype(int)
        ids = [_check_owned(out[f][r], cl, slice(cuts[r], cuts[r + 1]), "mean", max_points=P) for r in range(world)]
        assert np.array_equal(np.sort(np.concatenate(ids)), np.arange(out[f][0].num_voxels))
    for g in gens:
        buf = g._resident_buf
        state = buf.row_state.cpu().numpy().astype(np.int64) & 0xffff
        nz = (buf.voxels != 0).any(dim=2).cpu().numpy()
        assert not (nz & (np.arange(P)[None, :] >= state[:, None])).any()


@pytest.mark.gpu
def test_voxelize_reduce_single_gpu(index_path):
    from d3d_amd import synth
    from d3d_amd.voxel.sharded import voxelize_reduce
    cloud = synth.lidar_like(200000, 3)
    r = voxelize_reduce(torch.from_numpy(cloud).cuda(), synth.KITTI_SHAPE, synth.KITTI_BOUNDS, "mean")
    exp = oracle.voxelize_3d_dense(cloud, synth.KITTI_SHAPE, synth.KITTI_BOUNDS, 1, len(cloud), 1)
    assert np.array_equal(r.coords.cpu().numpy(), exp["coords"])
    assert np.array_equal(r.voxel_npoints.cpu().numpy(), exp["voxel_npoints"])
    fit = exp["voxel_npoints"] <= 32
    got = r.aggregates.cpu().numpy()
    assert np.array_equal(got[fit], exp["aggregates"][fit])          # sequential in point order: bit-exact
    np.testing.assert_allclose(got[~fit], exp["aggregates"][~fit], rtol=1e-5, atol=1e-6)
    m = r.points_mapping.cpu().numpy()
    assert m.min() >= 0 and np.array_equal(np.bincount(m, minlength=len(exp["coords"])), exp["voxel_npoints"])
    f = r.voxel_first.cpu().numpy()
    assert np.array_equal(m[f], np.arange(len(f))) and np.all(np.diff(f) > 0)


@pytest.mark.gpu
def test_rccl_world1_child_process():
    """the sharded voxelizer through TorchComm on backend `nccl` (RCCL) with world_size 1, in a fresh child
    process started before that child touches the GPU (tests/nccl_world1_child.py): every exchange, every collective
    signature, results against the oracle"""
    import subprocess
    import sys
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nccl_world1_child.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, child, str(port)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CHILD_OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.gpu
def test_rccl_every_visible_gpu():
    """lights up by itself on a multi-GPU box -- one FRESH child process per visible GPU (started before
    anything in it touches the GPU), backend `nccl`: config-5 geometry sharded over the ranks against the oracle, replicated and
    owner-only, the dense contract, the lock-step repeat after a bucket overflow, and the rank count RCCL reports
    (tests/nccl_worldN_child.py).  One GPU (every box of this pool so far): skipped -- test_rccl_world1_child_process covers the
    collective signatures there."""
    import subprocess
    import sys
    import socket
    world = torch.cuda.device_count()
    if world < 2:
        pytest.skip("needs >= 2 GPUs (this box shows %d)" % world)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nccl_worldN_child.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, child, str(r), str(world), str(port)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=900))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("CHILD_OK %d" % r) in so, (r, p.returncode, so[-2000:], se[-4000:])
