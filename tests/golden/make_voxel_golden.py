"""Generate tests/golden/voxel_*.npz by running the REAL reference (cmpute/d3d) in this container.

Needs /root/reference and oracle/_ref/voxel_impl.so (python oracle/build_ref.py).  The
reference's own Python layer (d3d/voxel/__init__.py) is imported from where it lies with a
stand-in for the missing `addict` dependency; only inputs + outputs (data) are written here.

Run:  python tests/golden/make_voxel_golden.py
"""
import importlib.util
import os
import shutil
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def load_reference_voxel():
    from oracle.build_ref import build, load_ref
    assert build() is not None, "cannot build the reference voxel_impl"
    impl = load_ref()

    class _D(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    addict = types.ModuleType("addict")
    addict.Dict = _D
    sys.modules["addict"] = addict
    pkg = types.ModuleType("d3d")
    pkg.__path__ = []
    sys.modules["d3d"] = pkg
    sys.modules["d3d.voxel.voxel_impl"] = impl
    spec = importlib.util.spec_from_file_location(
        "d3d.voxel", os.path.join(REF, "d3d/voxel/__init__.py"),
        submodule_search_locations=[os.path.join(REF, "d3d/voxel")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["d3d.voxel"] = mod
    spec.loader.exec_module(mod)
    return mod, impl


def lidar_like(n, seed, xr=(0, 70.4), yr=(-40, 40), zr=(-3, 1)):
    """SURVEY.md 8(d) generator (same as d3d_amd.synth.lidar_like, duplicated so this
    script has no dependency on the product package)."""
    rng = np.random.default_rng(seed)
    rmax = np.hypot(max(abs(xr[0]), abs(xr[1])), max(abs(yr[0]), abs(yr[1])))
    out = np.empty((0, 4), np.float32)
    while len(out) < n:
        m = 2 * n
        r = rmax * rng.random(m) ** 2 + 2
        az = rng.random(m) * 2 * np.pi
        x, y = r * np.cos(az), r * np.sin(az)
        ground = rng.random(m) < 0.7
        z = np.where(ground, -1.73 + 0.03 * rng.standard_normal(m), zr[0] + (zr[1] - zr[0]) * rng.random(m))
        pts = np.stack([x, y, z, rng.random(m)], 1).astype(np.float32)
        ok = (pts[:, 0] >= xr[0]) & (pts[:, 0] < xr[1]) & (pts[:, 1] >= yr[0]) & (pts[:, 1] < yr[1]) & \
             (pts[:, 2] >= zr[0]) & (pts[:, 2] < zr[1])
        out = np.concatenate([out, pts[ok]])
    return np.ascontiguousarray(out[:n])


def to_np(d):
    return {k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}


def main():
    vox, impl = load_reference_voxel()
    cases = {}

    def dense_case(name, cloud, bounds, shape, **kw):
        gen = vox.VoxelGenerator(bounds, shape, dense=True, **kw)
        r = to_np(gen(torch.from_numpy(cloud)))
        # reference pmask is torch::empty -> only its True entries are defined (voxelize.cpp:58,131)
        mp = kw.get("max_points", 30)
        derived = np.arange(mp)[None, :] < np.minimum(r["voxel_npoints"], mp)[:, None]
        assert np.all(r["voxel_pmask"][derived]), name
        r.pop("voxel_pmask")
        cases[name] = dict(kind="dense", cloud=cloud, bounds=np.asarray(bounds, np.float64),
                           shape=np.asarray(shape), kw=kw, out=r)

    def sparse_case(name, cloud, bounds, shape, **kw):
        gen = vox.VoxelGenerator(bounds, shape, **kw)
        r = to_np(gen(torch.from_numpy(cloud)))
        cases[name] = dict(kind="sparse", cloud=cloud, bounds=np.asarray(bounds, np.float64),
                           shape=np.asarray(shape), kw=kw, out=r)

    rng = np.random.default_rng(1234)
    unit = [0, 1, 0, 1, 0, 1]
    # (1) the reference's own fixture (test/test_voxel.py:80-88)
    fx = np.load(os.path.join(REF, "test/voxel_data.npz"))
    dense_case("fixture", fx["cloud"], unit, [10, 10, 10], max_points=5, max_points_filter="trim")
    assert np.array_equal(cases["fixture"]["out"]["voxels"], fx["voxels"])
    assert np.array_equal(cases["fixture"]["out"]["coords"], fx["coords"])

    # (2) test_generate_voxel-like: 2000 random + 2 outliers (test_voxel.py:11-31)
    cloud = rng.random((2000, 4), dtype=np.float32)
    cloud = np.concatenate([cloud, np.array([[-1, -1, -1, -100], [-2, -2, -2, 100]], np.float32)])
    for red in ["none", "mean", "max", "min"]:
        dense_case("unit_" + red, cloud, unit, [10, 10, 10], reduction=red, max_points=5, max_voxels=20000,
                   max_points_filter="trim", max_voxels_filter="trim")
    # max_voxels cap: later voxels dropped, existing keep accumulating (voxelize.cpp:116-117)
    dense_case("unit_cap", cloud, unit, [10, 10, 10], reduction="mean", max_points=3, max_voxels=100)
    # heavy overflow: 5000 points into 4x4x4
    c2 = rng.random((5000, 5), dtype=np.float32)
    dense_case("overflow_c5", c2, unit, [4, 4, 4], reduction="mean", max_points=8, max_voxels=50)
    dense_case("overflow_max_c5", c2, unit, [4, 4, 4], reduction="max", max_points=8, max_voxels=64)
    # edge cases: boundaries, slightly negative (truncation toward zero keeps them, App. A.1), nan/inf/huge
    e = np.array([
        [0, 0, 0, 1], [-0.05, 0.5, 0.5, 2], [-0.0999, 0.5, 0.5, 3], [-0.1001, 0.5, 0.5, 4],
        [1.0, 0.5, 0.5, 5], [0.99999994, 0.99999994, 0.99999994, 6], [0.5, 0.5, np.nan, 7],
        [np.inf, 0.5, 0.5, 8], [-np.inf, 0.5, 0.5, 9], [3e9, 0.5, 0.5, 10], [-3e9, 0.5, 0.5, 11],
        [0.3, 0.3, 0.3, 12], [0.30000001, 0.3, 0.3, 13], [0.7, 0.7, 0.7, 14], [-0.0, -0.0, -0.0, 15],
        [0.1, 0.2, 0.3, 16], [0.2, 0.4, 0.6, 17], [0.6, 0.6, 0.6, 18], [0.9, 0.9, 0.9, np.nan],
    ], np.float32)
    dense_case("edges", e, unit, [10, 10, 10], reduction="mean", max_points=4, max_voxels=100)
    dense_case("edges_min", e, unit, [10, 10, 10], reduction="min", max_points=4, max_voxels=100)
    # KITTI-like, small
    kc = lidar_like(20000, 7)
    kb = [0, 70.4, -40, 40, -3, 1]
    dense_case("kitti_mean", kc, kb, [704, 800, 40], reduction="mean", max_points=32, max_voxels=20000)
    dense_case("kitti_capped", kc, kb, [704, 800, 40], reduction="none", max_points=2, max_voxels=5000)
    dense_case("kitti_coarse", kc, kb, [88, 100, 4], reduction="mean", max_points=32, max_voxels=20000)
    # C = 3 features only
    dense_case("c3", np.ascontiguousarray(kc[:5000, :3]), kb, [176, 200, 8], reduction="max", max_points=6, max_voxels=4000)

    # sparse + filter (test_voxel.py:40-78)
    sparse_case("sp_unit", cloud, unit, [10, 10, 10])
    c3 = ((rng.random((2000, 3), dtype=np.float32) - 0.5) * 4).astype(np.float32)
    sparse_case("sp_bounds", c3, [-1, 1, -1, 1, -1, 1], [20, 20, 20])
    sparse_case("sp_trimvox", c3, unit, [10, 10, 10], max_voxels=10, max_voxels_filter="trim")
    sparse_case("sp_minmax", c3, unit, [10, 10, 10], min_points=2, max_points=4, max_points_filter="trim")
    sparse_case("sp_kitti", kc, kb, [704, 800, 40], max_points=32, max_points_filter="trim")
    sparse_case("sp_kitti_coarse", kc, kb, [88, 100, 4], max_points=5, max_points_filter="trim",
                max_voxels=3000, max_voxels_filter="trim", min_points=2)
    sparse_case("sp_offset", (c3 * 3).astype(np.float32), [-4, 4, -2, 6, 1, 5], [16, 16, 8], max_points=3,
                max_points_filter="trim")
    # descending: make counts tie-free among the kept ones is impossible in general -> store, compare as sets
    sparse_case("sp_desc", c3, unit, [10, 10, 10], max_voxels=10, max_voxels_filter="descending")

    # far outliers on every axis (sparse coordinates up to +-9e5): exercises the widest keys
    c4 = c3.copy()
    c4[:6] = [[9e4, 0, 0], [-9e4, 0, 0], [0, 9e4, 0], [0, -9e4, 0], [0, 0, 9e4], [0, 0, -9e4]]
    sparse_case("sp_outliers", c4, [-1, 1, -1, 1, -1, 1], [20, 20, 20], max_points=4, max_points_filter="trim")
    # C = 8 features, max_points not a multiple of 16, scan-ordered input (sorted by azimuth), a 600-point voxel
    c8 = rng.random((6000, 8), dtype=np.float32)
    c8[:600, :3] = 0.55 + 0.01 * rng.random((600, 3), dtype=np.float32)
    c8 = c8[np.argsort(np.arctan2(c8[:, 1] - 0.5, c8[:, 0] - 0.5), kind="stable")]
    dense_case("c8_ordered", np.ascontiguousarray(c8), unit, [12, 12, 12], reduction="mean", max_points=30, max_voxels=2000)
    sparse_case("sp_c8_minpts", np.ascontiguousarray(c8), unit, [12, 12, 12], min_points=3, max_points=7,
                max_points_filter="trim", max_voxels=400, max_voxels_filter="trim")

    # non-finite and far-out-of-range points in the default (sparse) mode: the reference's (int)floor(NaN) lands in an
    # INT_MIN voxel (voxelize.cpp:309) that the coordinate-bound filter drops (:376-384) -- the frame still works
    c5 = c3.copy()
    c5[3, 0], c5[7, 1], c5[9, 2], c5[12, :3], c5[20, 1], c5[31, 0] = np.nan, np.inf, -np.inf, np.nan, -2.5e5, 3e6
    sparse_case("sp_nonfinite", c5, [-1, 1, -1, 1, -1, 1], [20, 20, 20], max_points=4, max_points_filter="trim")
    sparse_case("sp_nonfinite_none", c5, unit, [10, 10, 10], min_points=1)

    # raw-function cases (boundary functions called directly, voxelize.h:9-25)
    sp = to_np(impl.voxelize_3d_sparse(torch.from_numpy(kc), torch.tensor([0.1, 0.1, 0.1]), 3))
    cases["raw_sparse"] = dict(kind="raw_sparse", cloud=kc, size=np.array([0.1, 0.1, 0.1], np.float32), out=sp)

    # round 5 -- voxel coordinates of ANY size (voxelize.cpp:309 keys a voxel by three ints, whatever they are): +-3e6 cells on
    # every axis at once (far beyond 3 x 21 bits; together beyond 64), several points per far voxel, far voxels that differ in ONE
    # coordinate only (in z; in y by one), non-finite points (INT_MIN) next to them
    cw = c3.copy()
    far = np.array([[3e5, 3e5, 3e5], [-3e5, -3e5, -3e5], [3e5, -3e5, 2.5e5], [3e5, -3e5, -2.5e5], [2.9e5, 1.0, 3e5], [2.9e5, 1.1, 3e5],
                    [2.9e5, 1.0, -3e5], [-1.5e5, 2e5, 0.05], [1.2e5, 0.05, 0.05], [0.05, -1.2e5, 0.05]], np.float32)
    for k in range(60):
        cw[5 * k, :3] = far[k % len(far)] + (0.01 if k >= 30 else 0.0)        # (0.01 stays inside the far cell at 0.1 m)
    cw[7, 0], cw[13, 1], cw[19, :3] = np.nan, np.inf, np.nan
    spw = to_np(impl.voxelize_3d_sparse(torch.from_numpy(cw), torch.tensor([0.1, 0.1, 0.1]), 3))
    cases["raw_sparse_wide"] = dict(kind="raw_sparse", cloud=cw, size=np.array([0.1, 0.1, 0.1], np.float32), out=spw)
    # ... and through the generator, its coordinate bounds reaching that far (so the far voxels are KEPT by the filter)
    sparse_case("sp_wide", cw, [-3.2e5, 3.2e5, -3.2e5, 3.2e5, -3.2e5, 3.2e5], [6400000, 6400000, 6400000], max_points=3,
                max_points_filter="trim", min_points=1)

    flat = {}
    import json
    meta = {}
    for name, cs in cases.items():
        meta[name] = {"kind": cs["kind"], "kw": cs.get("kw", {}),
                      "bounds": None if "bounds" not in cs else [float(x) for x in cs["bounds"]],
                      "shape": None if "shape" not in cs else [int(x) for x in cs["shape"]]}
        flat[name + "/cloud"] = cs["cloud"]
        if "size" in cs:
            flat[name + "/size"] = cs["size"]
        for k, v in cs["out"].items():
            flat[name + "/out/" + k] = v
    flat["__meta__"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "voxel_ref_cases.npz"), **flat)
    shutil.copyfile(os.path.join(REF, "test/voxel_data.npz"), os.path.join(HERE, "voxel_data_ref.npz"))
    print("wrote", len(cases), "cases;", os.path.getsize(os.path.join(HERE, "voxel_ref_cases.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()
