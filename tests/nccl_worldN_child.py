"""Child process of tests/test_sharded.py::test_rccl_every_visible_gpu -- one per visible GPU, started before anything in it has
touched the GPU (never a re-exec of a process that did): the sharded voxelizer through TorchComm on backend `nccl` (= RCCL on
ROCm) with world_size = the number of GPUs.  Rank k holds points [k n, (k + 1) n) of a frame in config 5's geometry (0.05 m
voxels over 150 m x 150 m x 6 m); the owner-computes exchange replicated and not, the dense contract of the owned voxels, the
lock-step repeat after a bucket overflow of the owners' merge (OWNER_MERGE_TEST_TINY) -- every rank checks what it holds
against the CPU oracle of the whole frame -- and the number of ranks RCCL itself reports.
usage: python tests/nccl_worldN_child.py <rank> <world> <port> [points per rank]; prints CHILD_OK <rank> on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
per_rank = int(sys.argv[4]) if len(sys.argv) > 4 else 150000
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np          # noqa: E402
import torch                # noqa: E402
import torch.distributed as dist   # noqa: E402


def main():
    import oracle
    from d3d_amd import _lib, synth
    from d3d_amd.voxel.sharded import ShardedVoxelGenerator, TorchComm
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        comm = TorchComm()
        ones = torch.ones((1,), dtype=torch.int64, device=dev)
        dist.all_reduce(ones)
        assert int(ones.item()) == world, "RCCL summed %d ranks, the launcher started %d" % (int(ones.item()), world)
        ids = [None] * world
        props = torch.cuda.get_device_properties(rank)
        dist.all_gather_object(ids, str(getattr(props, "uuid", "")) or "%s#%d" % (props.name, rank))
        assert len(set(ids)) == world, "ranks share a device: %s" % ids
        bounds, shape = synth.WAYMO_BOUNDS, synth.WAYMO_SHAPE
        frame = synth.lidar_like(world * per_rank, 3, bounds)
        # ragged shards: rank 0 holds a third of an even share, the last rank the rest
        cuts = np.linspace(0, len(frame), world + 1).astype(int)
        if world > 1:
            cuts[1] = max(cuts[1] // 3, 1)
        mine = slice(cuts[rank], cuts[rank + 1])
        pts = torch.from_numpy(np.ascontiguousarray(frame[mine])).to(dev)
        P = 8
        for reduction in ("mean", "max"):
            exp = oracle.voxelize_3d_dense(frame, shape, bounds, P, len(frame), reduction)
            res = ShardedVoxelGenerator(bounds, shape, reduction=reduction, comm=comm, exchange="owner", debug_checks=True)(pts)
            assert np.array_equal(res.coords.cpu().numpy(), exp["coords"]) and np.array_equal(res.voxel_npoints.cpu().numpy(), exp["voxel_npoints"])
            if reduction == "mean":
                np.testing.assert_allclose(res.aggregates.cpu().numpy(), exp["aggregates"], rtol=1e-5, atol=1e-6)
            else:
                assert np.array_equal(res.aggregates.cpu().numpy(), exp["aggregates"])
            m = res.points_mapping.cpu().numpy()
            inside = m >= 0
            lut = {tuple(c): v for v, c in enumerate(exp["coords"].tolist())}
            lo = np.asarray(bounds, np.float32)[0::2]
            size = ((np.asarray(bounds, np.float32)[1::2] - lo) / np.asarray(shape, np.float32)).astype(np.float32)
            cc = ((frame[mine][inside, :3] - lo) / size).astype(np.int64)
            assert all(lut[tuple(c)] == v for c, v in zip(cc[:2000].tolist(), m[inside][:2000].tolist()))
            for mflags in (0, _lib.OWNER_MERGE_TEST_TINY):       # (tiny buckets: every rank learns of the overflow and they all repeat)
                own = ShardedVoxelGenerator(bounds, shape, reduction=reduction, comm=comm, exchange="owner", replicate=False, max_points=P,
                                            debug_checks=True, merge_flags=mflags)(pts)
                vid = own.voxel_ids.cpu().numpy()
                assert own.num_voxels == len(exp["coords"]) and (len(vid) < 2 or np.all(np.diff(vid) > 0))
                assert np.array_equal(own.coords.cpu().numpy(), exp["coords"][vid])
                assert np.array_equal(own.voxel_npoints.cpu().numpy(), exp["voxel_npoints"][vid])
                assert np.array_equal(own.voxels.cpu().numpy(), exp["voxels"][vid])
                total = torch.tensor([len(vid)], dtype=torch.int64, device=dev)
                dist.all_reduce(total)
                assert int(total.item()) == len(exp["coords"])          # the owned voxels of the ranks partition the grid
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    print("CHILD_OK %d" % rank)


if __name__ == "__main__":
    main()
