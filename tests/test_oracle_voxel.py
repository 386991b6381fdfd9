"""CPU: the C oracle (oracle/voxel_oracle.c) against outputs of the REAL reference
(tests/golden/voxel_ref_cases.npz) and the reference's own fixture (test/test_voxel.py:80-88)."""
import os

import numpy as np
import pytest

import oracle
from golden_io import GOLDEN, derived_pmask, load_voxel_cases

CASES = load_voxel_cases()


def check_dense(ret, exp, max_points, exact_mean=True):
    assert np.array_equal(ret["coords"], exp["coords"])
    assert ret["coords"].dtype == np.int64
    assert np.array_equal(ret["voxel_npoints"], exp["voxel_npoints"])
    assert ret["voxel_npoints"].dtype == np.int32
    assert np.array_equal(ret["voxels"], exp["voxels"], equal_nan=True)
    assert np.array_equal(ret["voxel_pmask"], derived_pmask(exp["voxel_npoints"], max_points))
    assert ("aggregates" in ret) == ("aggregates" in exp)
    if "aggregates" in exp:
        if exact_mean:
            assert np.array_equal(ret["aggregates"], exp["aggregates"], equal_nan=True)
        else:
            np.testing.assert_allclose(ret["aggregates"], exp["aggregates"], rtol=1e-5, atol=1e-6)


def check_sparse(ret, exp, descending=False):
    if descending:
        # unstable argsort in the reference (voxelize.cpp:406): compare as sets of (coord, count)
        a = sorted(map(tuple, np.concatenate([ret["coords"], ret["voxel_npoints"][:, None]], 1).tolist()))
        b = sorted(map(tuple, np.concatenate([exp["coords"], exp["voxel_npoints"][:, None]], 1).tolist()))
        assert len(a) == len(b)
        # counts kept must be the same multiset; ties may pick different voxels
        assert sorted(x[3] for x in a) == sorted(x[3] for x in b)
        return
    for k in ["points", "points_mask", "points_mapping", "voxel_npoints", "coords"]:
        assert np.array_equal(ret[k], exp[k]), k
        assert ret[k].dtype == exp[k].dtype, k


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if c["meta"]["kind"] == "dense"])
def test_oracle_dense_matches_reference(name):
    c = CASES[name]
    kw = dict(c["meta"]["kw"])
    gen = oracle.VoxelGenerator(c["meta"]["bounds"], c["meta"]["shape"], dense=True, **kw)
    check_dense(gen(c["cloud"]), c["out"], kw.get("max_points", 30))


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if c["meta"]["kind"] == "sparse"])
def test_oracle_sparse_matches_reference(name):
    c = CASES[name]
    kw = dict(c["meta"]["kw"])
    gen = oracle.VoxelGenerator(c["meta"]["bounds"], c["meta"]["shape"], **kw)
    check_sparse(gen(c["cloud"]), c["out"], descending=kw.get("max_voxels_filter") == "descending")


@pytest.mark.parametrize("name", ["raw_sparse", "raw_sparse_wide"])
def test_oracle_raw_sparse(name):
    """(raw_sparse_wide: voxel coordinates of +-3e6 cells on every axis, voxels that differ in one coordinate only, INT_MIN)"""
    c = CASES[name]
    r = oracle.voxelize_3d_sparse(c["cloud"], c["size"], 3)
    for k in ["points_mapping", "coords", "voxel_npoints"]:
        assert np.array_equal(r[k], c["out"][k]) and r[k].dtype == c["out"][k].dtype


def test_reference_fixture_spconv():
    """reference test/test_voxel.py:80-88 (fixture generated with spconv VoxelGeneratorV2)"""
    data = np.load(os.path.join(GOLDEN, "voxel_data_ref.npz"))
    gen = oracle.VoxelGenerator([0, 1, 0, 1, 0, 1], [10, 10, 10], max_points=5, max_points_filter="trim", dense=True)
    ret = gen(data["cloud"])
    assert np.array_equal(ret.voxels, data["voxels"])
    assert np.array_equal(ret.coords, data["coords"])


def test_oracle_invariants_like_reference_tests():
    """reference test/test_voxel.py:11-51 invariants, on the oracle"""
    rng = np.random.default_rng(0)
    cloud = rng.random((2000, 4), dtype=np.float32)
    cloud = np.concatenate([cloud, np.array([[-1, -1, -1, -100], [-2, -2, -2, 100]], np.float32)])
    gen = oracle.VoxelGenerator([0, 1, 0, 1, 0, 1], [10, 10, 10], reduction="mean", max_points=5, max_voxels=20000,
                                max_points_filter="trim", max_voxels_filter="trim", dense=True)
    d = gen(cloud)
    assert len(d.voxels) == len(d.coords) <= 1000
    assert np.all((d.voxels >= 0) & (d.voxels <= 1))
    for i in range(len(d.voxels)):
        for j in range(min(d.voxel_npoints[i], 5)):
            assert np.array_equal(d.coords[i], (d.voxels[i, j, :3] * 10).astype(np.int64))
    gen = oracle.VoxelGenerator([0, 1, 0, 1, 0, 1], [10, 10, 10])
    d = gen(cloud)
    assert len(d.points) == 2000
    assert "aggregates" not in d


def test_oracle_errors():
    with pytest.raises(ValueError):
        oracle.VoxelGenerator([0.05, 1, 0, 1, 0, 1], [10, 10, 10])
    with pytest.raises(ValueError):
        oracle.VoxelGenerator([0, 1, 0, 1, 0, 1], [10, 10, 10], reduction="mean")
    with pytest.raises(NotImplementedError):
        oracle.VoxelGenerator([0, 1, 0, 1, 0, 1], [10, 10, 10], min_points=1, dense=True)
