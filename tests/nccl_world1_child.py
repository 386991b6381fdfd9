"""Child process of tests/test_sharded.py::test_rccl_world1_child_process: ShardedVoxelGenerator through TorchComm on backend
`nccl` (= RCCL on ROCm) with world_size 1, every exchange and every collective signature -- all_to_all_single with split
sizes, all_reduce SUM / MAX, all_gather_into_tensor -- checked against the CPU oracle.  Started before anything in this
process has touched the GPU.  Prints CHILD_OK on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29571")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np          # noqa: E402
import torch                # noqa: E402
import torch.distributed as dist   # noqa: E402


def main():
    import oracle
    from d3d_amd import synth
    from d3d_amd.voxel.sharded import ShardedVoxelGenerator, TorchComm
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        comm = TorchComm()
        # the collective signatures themselves, on the nccl backend
        dev = torch.device("cuda", 0)
        assert comm.all_gather_int(7, dev) == [7]
        t = torch.arange(10, dtype=torch.int32, device=dev).view(5, 2)
        assert torch.equal(comm.all_to_all(t, [5], [5]), t)
        assert torch.equal(comm.all_reduce(torch.tensor([3, 4], dtype=torch.int64, device=dev), "sum").cpu(), torch.tensor([3, 4]))
        assert comm.exchange_counts(torch.tensor([1, 2, 3], dtype=torch.int64, device=dev)) == [[1, 2, 3]]
        bounds, shape = synth.KITTI_BOUNDS, [176, 200, 10]
        cloud = synth.lidar_like(60000, 11)
        pts = torch.from_numpy(cloud).cuda()
        for reduction in ("mean", "max"):
            exp = oracle.voxelize_3d_dense(cloud, shape, bounds, 4, len(cloud), reduction)
            for exchange in ("owner", "keys", "bitmap"):
                gen = ShardedVoxelGenerator(bounds, shape, reduction=reduction, comm=comm, exchange=exchange, debug_checks=True)
                res = gen(pts)
                assert np.array_equal(res.coords.cpu().numpy(), exp["coords"]), (reduction, exchange)
                assert np.array_equal(res.voxel_npoints.cpu().numpy(), exp["voxel_npoints"]), (reduction, exchange)
                got, want = res.aggregates.cpu().numpy(), exp["aggregates"]
                if reduction == "mean":
                    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
                else:
                    assert np.array_equal(got, want)
            # owner-computes without replication, and the dense contract through it
            own = ShardedVoxelGenerator(bounds, shape, reduction=reduction, comm=comm, exchange="owner", replicate=False, max_points=4,
                                        debug_checks=True)(pts)
            ids = own.voxel_ids.cpu().numpy()
            assert np.array_equal(ids, np.arange(len(exp["coords"])))
            assert np.array_equal(own.voxels.cpu().numpy(), exp["voxels"])
            assert own.num_voxels == len(exp["coords"])
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    print("CHILD_OK")


if __name__ == "__main__":
    main()
