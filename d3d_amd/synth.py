"""Synthetic workloads of BASELINE.json's configs; numpy, seeded, host side."""
import numpy as np

KITTI_BOUNDS = [0, 70.4, -40, 40, -3, 1]
KITTI_SHAPE = [704, 800, 40]          # 0.1 m voxels
WAYMO_BOUNDS = [-75.2, 75.2, -75.2, 75.2, -2, 4]
WAYMO_SHAPE = [3008, 3008, 120]       # 0.05 m voxels


def lidar_like(n, seed=0, bounds=KITTI_BOUNDS):
    """LiDAR-like cloud [n,4] f32: r = rmax*u^2 + 2, uniform azimuth, 70 % ground z = -1.73 + 0.03 N(0,1),
    30 % uniform z, uniform intensity; rejection-filtered to `bounds`, first n kept."""
    xr, yr, zr = bounds[0:2], bounds[2:4], bounds[4:6]
    rng = np.random.default_rng(seed)
    rmax = np.hypot(max(abs(xr[0]), abs(xr[1])), max(abs(yr[0]), abs(yr[1])))
    chunks, have = [], 0
    while have < n:
        m = max(2 * (n - have), 1024)
        r = rmax * rng.random(m) ** 2 + 2
        az = rng.random(m) * 2 * np.pi
        x, y = r * np.cos(az), r * np.sin(az)
        ground = rng.random(m) < 0.7
        z = np.where(ground, -1.73 + 0.03 * rng.standard_normal(m), zr[0] + (zr[1] - zr[0]) * rng.random(m))
        pts = np.stack([x, y, z, rng.random(m)], 1).astype(np.float32)
        ok = (pts[:, 0] >= xr[0]) & (pts[:, 0] < xr[1]) & (pts[:, 1] >= yr[0]) & (pts[:, 1] < yr[1]) & \
             (pts[:, 2] >= zr[0]) & (pts[:, 2] < zr[1])
        chunks.append(pts[ok])
        have += int(ok.sum())
    return np.ascontiguousarray(np.concatenate(chunks)[:n])


def uniform_cloud(n, seed=0, bounds=KITTI_BOUNDS):
    rng = np.random.default_rng(seed)
    lo = np.array([bounds[0], bounds[2], bounds[4], 0], np.float64)
    hi = np.array([bounds[1], bounds[3], bounds[5], 1], np.float64)
    return (lo + (hi - lo) * rng.random((n, 4))).astype(np.float32)


def boxes2d_sparse(n, seed=1, dtype=np.float64):
    """cfg3: same density as reference test/test_box.py:126-131 (500 boxes in 200 x 400)."""
    rng = np.random.default_rng(seed)
    side = np.sqrt(n / 500.0)
    b = np.stack([rng.random(n) * 200 * side, rng.random(n) * 400 * side, rng.random(n) * 20 + 10,
                  rng.random(n) * 30 + 5, rng.random(n) * 2 - 1], 1)
    scores = rng.random(n)
    return b.astype(dtype), scores.astype(dtype)


def boxes2d_dense(n, seed=1, dtype=np.float64):
    """the reference's own benchmark boxes (test/compare/benchmark_riou.py:70-74)"""
    rng = np.random.default_rng(seed)
    b = np.stack([(rng.random(n) - 0.5) * 10, (rng.random(n) - 0.5) * 10, rng.random(n) * 5, rng.random(n) * 5,
                  (rng.random(n) - 0.5) * 10], 1)
    return b.astype(dtype), rng.random(n).astype(dtype)


def boxes3d_eval(n_gt=5000, rep=4, seed=2):
    """cfg4: GT [n_gt,7] car-like boxes over 150 x 150 m; preds = each GT repeated `rep` times with noise."""
    rng = np.random.default_rng(seed)
    gt = np.stack([rng.random(n_gt) * 150, rng.random(n_gt) * 150, rng.random(n_gt) * 2 - 2,
                   rng.random(n_gt) * 1.5 + 3.5, rng.random(n_gt) * 0.5 + 1.6, rng.random(n_gt) * 0.5 + 1.4,
                   rng.random(n_gt) * 2 * np.pi - np.pi], 1)
    pred = np.repeat(gt, rep, axis=0)
    pred[:, 0:3] += 0.3 * rng.standard_normal((len(pred), 3))
    pred[:, 3:6] += 0.1 * rng.standard_normal((len(pred), 3))
    pred[:, 6] += 0.1 * rng.standard_normal(len(pred))
    return pred.astype(np.float32), gt.astype(np.float32)
