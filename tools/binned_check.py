"""dense voxelizer: binned index path vs hash-table path -- identical outputs, and the time of each (GPU)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from d3d_amd import _lib, synth
from d3d_amd.voxel import voxelize_3d_dense

lib = _lib.load()


def run(path, pts, shape, bound, P, mv, red):
    return voxelize_3d_dense(pts, shape, bound, P, mv, red, flags=_lib.VOXEL_PATH_HASH if path == 1 else 0)


def same(a, b, what):
    bad = []
    for k in a:
        x, y = a[k], b[k]
        if x.shape != y.shape:
            bad.append("%s shape %s vs %s" % (k, tuple(x.shape), tuple(y.shape)))
        elif k == "aggregates":
            if not torch.allclose(x, y, rtol=1e-6, atol=1e-6, equal_nan=True):
                bad.append("%s max diff %g" % (k, float((x - y).abs().max())))
        elif not torch.equal(x, y):
            bad.append("%s differs in %d entries" % (k, int((x != y).sum())))
    print("%-60s %s" % (what, "OK (V=%d)" % a["coords"].shape[0] if not bad else "MISMATCH " + "; ".join(bad)))
    return not bad


ok = True
for n, seed in ((1000, 1), (5000, 2), (40000, 3), (300000, 4), (1000000, 0)):
    cloud = torch.from_numpy(synth.lidar_like(n, seed)).cuda()
    for red in (0, 1, 2):
        for P, mv in ((32, n), (5, n), (32, max(n // 20, 10)), (16, n), (1, n)):
            a = run(1, cloud, synth.KITTI_SHAPE, synth.KITTI_BOUNDS, P, mv, red)
            b = run(2, cloud, synth.KITTI_SHAPE, synth.KITTI_BOUNDS, P, mv, red)
            ok &= same(a, b, "n=%d red=%d P=%d max_voxels=%d" % (n, red, P, mv))
# out-of-range points, NaNs, duplicates
g = torch.Generator().manual_seed(5)
pts = torch.rand((200000, 4), generator=g) * torch.tensor([90.0, 100.0, 6.0, 1.0]) - torch.tensor([10.0, 50.0, 4.0, 0.0])
pts[::97, 0] = float("nan")
pts[1000:3000] = pts[0:2000].clone()
pts = pts.cuda()
for red in (0, 1):
    a = run(1, pts, synth.KITTI_SHAPE, synth.KITTI_BOUNDS, 32, 200000, red)
    b = run(2, pts, synth.KITTI_SHAPE, synth.KITTI_BOUNDS, 32, 200000, red)
    ok &= same(a, b, "out-of-range + NaN + duplicates, red=%d" % red)
# coarse grid: heavy voxels (bucket overflow -> automatic retry on the hash path)
a = run(1, cloud, [8, 8, 2], synth.KITTI_BOUNDS, 32, 1000, 1)
b = run(2, cloud, [8, 8, 2], synth.KITTI_BOUNDS, 32, 1000, 1)
ok &= same(a, b, "1 M points in 128 cells (overflow -> retry)")

cloud = torch.from_numpy(synth.lidar_like(1000000, 0)).cuda()
for path, name in ((1, "hash"), (2, "binned")):
    fl = _lib.VOXEL_PATH_HASH if path == 1 else 0
    for _ in range(3):
        voxelize_3d_dense(cloud, synth.KITTI_SHAPE, synth.KITTI_BOUNDS, 32, 1000000, 1, flags=fl)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        voxelize_3d_dense(cloud, synth.KITTI_SHAPE, synth.KITTI_BOUNDS, 32, 1000000, 1, flags=fl)
    torch.cuda.synchronize()
    print("%s: %.1f us per call (operator level, incl. allocation + host sync)" % (name, (time.perf_counter() - t) / 20 * 1e6))
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
