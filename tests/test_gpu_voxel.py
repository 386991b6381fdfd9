"""GPU parity (through the C ABI): d3d_amd.voxel vs outputs of the real reference (golden) and vs the
CPU oracle on seeded random clouds.  Bit-exact: coords, counts, voxels, pmask, mapping, MAX/MIN, and MEAN
for voxels that do not overflow max_points; MEAN of overflow voxels within 1e-5 rel (atomic order)."""
import numpy as np
import pytest
import torch

import oracle
from call_opts import cur_opts, set_opts
from golden_io import GOLDEN, derived_pmask, load_voxel_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["auto", "hash", "split", "3pass"])
def index_path(request):
    """every test of this module runs on both index paths of the voxelizer: automatic (binned whenever eligible; dense
    contract: the fused output kernel k_emit) and the hash table (tests that pick a path themselves -- `voxel_path` --
    override this); "split" = binned index with the two-launch output stage (k_meta_first + k_fill_c4: what max_points above 256
    takes by itself); "3pass" = the partition of frames above 8 M points.  (Round 5's fifth run on D3D_VOXEL_INDEX_V1 went with the
    flag: the bucket kernel's record form stays under test through the contracts that use it -- rows of 3 / 5 .. 8 floats, the
    reduce contract with a point map, DESCENDING.)  The dense operator's output buffers are poisoned before every call."""
    from d3d_amd import _lib
    set_opts(voxel_flags={"hash": _lib.VOXEL_PATH_HASH, "split": _lib.VOXEL_SPLIT_FILL,
                          "3pass": _lib.VOXEL_PARTITION_3PASS}.get(request.param, 0), poison=True)
    yield request.param


CASES = load_voxel_cases()


def _np(d):
    return {k: v.detach().cpu().numpy() for k, v in d.items()}


def check_dense(ret, exp, max_points):
    assert np.array_equal(ret["coords"], exp["coords"]) and ret["coords"].dtype == np.int64
    assert np.array_equal(ret["voxel_npoints"], exp["voxel_npoints"]) and ret["voxel_npoints"].dtype == np.int32
    assert np.array_equal(ret["voxels"], exp["voxels"], equal_nan=True)
    assert ret["voxel_pmask"].dtype == np.bool_
    assert np.array_equal(ret["voxel_pmask"], derived_pmask(exp["voxel_npoints"], max_points))
    assert ("aggregates" in ret) == ("aggregates" in exp)
    if "aggregates" in exp:
        fit = exp["voxel_npoints"] <= max_points
        assert np.array_equal(ret["aggregates"][fit], exp["aggregates"][fit], equal_nan=True)
        # overflow voxels: MEAN / sums run over all points in fp64 (arrival order); the reference's sequential fp32 sum
        # carries a rounding error that grows with the number of points (~count * 2^-24 relative) and bounds the difference
        cnt = exp["voxel_npoints"][~fit].astype(np.float64)[:, None]
        tol = np.maximum(1e-5, cnt * 2.0 ** -23)
        got, want = ret["aggregates"][~fit].astype(np.float64), exp["aggregates"][~fit].astype(np.float64)
        assert np.all(np.abs(got - want) <= 1e-6 + tol * np.abs(want)), float(np.max(np.abs(got - want)))


def check_sparse(ret, exp):
    for k in ["points", "points_mask", "points_mapping", "voxel_npoints", "coords"]:
        assert np.array_equal(ret[k], exp[k]), k
        assert ret[k].dtype == exp[k].dtype, k


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if c["meta"]["kind"] == "dense"])
@pytest.mark.parametrize("on_gpu", [True, False])
def test_dense_matches_reference(name, on_gpu):
    from d3d_amd.voxel import VoxelGenerator
    c = CASES[name]
    kw = dict(c["meta"]["kw"])
    gen = VoxelGenerator(c["meta"]["bounds"], c["meta"]["shape"], dense=True, **kw)
    pts = torch.from_numpy(c["cloud"])
    ret = gen(pts.cuda() if on_gpu else pts)
    assert ret.voxels.is_cuda == on_gpu
    check_dense(_np(ret), c["out"], kw.get("max_points", 30))


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if c["meta"]["kind"] == "sparse"])
def test_sparse_matches_reference(name):
    from d3d_amd.voxel import VoxelGenerator
    c = CASES[name]
    kw = dict(c["meta"]["kw"])
    gen = VoxelGenerator(c["meta"]["bounds"], c["meta"]["shape"], **kw)
    ret = _np(gen(torch.from_numpy(c["cloud"]).cuda()))
    if kw.get("max_voxels_filter") == "descending":
        # reference argsort is unstable (voxelize.cpp:406): our spec is the stable order -> compare with the oracle
        exp = oracle.VoxelGenerator(c["meta"]["bounds"], c["meta"]["shape"], **kw)(c["cloud"])
        check_sparse(ret, exp)
        assert sorted(ret["voxel_npoints"].tolist()) == sorted(c["out"]["voxel_npoints"].tolist())
    else:
        check_sparse(ret, c["out"])


@pytest.mark.parametrize("name", ["raw_sparse", "raw_sparse_wide"])
def test_raw_sparse_function(name):
    """raw_sparse_wide (round 5): voxel coordinates of +-3e6 cells on every axis -- beyond the 3 x 21-bit key: the call comes
    back with COORD_OVERFLOW and is repeated on the table that compares all 96 bits (D3D_VOXEL_WIDE_KEYS); same voxels, ids and
    counts as the compiled reference, INT_MIN voxels of the non-finite points included"""
    from d3d_amd.voxel import voxelize_3d_sparse
    c = CASES[name]
    r = _np(voxelize_3d_sparse(torch.from_numpy(c["cloud"]).cuda(), torch.from_numpy(c["size"]), 3))
    for k in ["points_mapping", "coords", "voxel_npoints"]:
        assert np.array_equal(r[k], c["out"][k]) and r[k].dtype == c["out"][k].dtype


def test_reference_fixture_spconv():
    """reference test/test_voxel.py:80-88"""
    from d3d_amd.voxel import VoxelGenerator
    data = np.load(GOLDEN + "/voxel_data_ref.npz")
    gen = VoxelGenerator([0, 1, 0, 1, 0, 1], [10, 10, 10], max_points=5, max_points_filter="trim", dense=True)
    ret = gen(torch.tensor(data["cloud"]))
    assert np.allclose(ret.voxels.numpy(), data["voxels"]) and np.array_equal(ret.voxels.numpy(), data["voxels"])
    assert np.array_equal(ret.coords.numpy(), data["coords"])


@pytest.mark.parametrize("reduction", ["none", "mean", "max", "min"])
@pytest.mark.parametrize("dist", ["lidar", "uniform"])
def test_dense_vs_oracle_100k(reduction, dist):
    from d3d_amd import synth
    from d3d_amd.voxel import VoxelGenerator
    cloud = synth.lidar_like(100000, 11) if dist == "lidar" else synth.uniform_cloud(100000, 12)
    kw = dict(reduction=reduction, max_points=8, max_voxels=60000, dense=True)
    exp = oracle.VoxelGenerator(synth.KITTI_BOUNDS, [352, 400, 20], **kw)(cloud)
    ret = _np(VoxelGenerator(synth.KITTI_BOUNDS, [352, 400, 20], **kw)(torch.from_numpy(cloud).cuda()))
    check_dense(ret, exp, 8)


@pytest.mark.parametrize("c", [3, 5, 6, 7, 8])
@pytest.mark.parametrize("P,reduction", [(32, "mean"), (8, "max"), (12, "min"), (32, "none"), (5, "mean")])
def test_dense_rows_of_other_widths_vs_oracle(c, P, reduction):
    """C = 3, 5 .. 8 columns take the one-launch output kernel for C-float rows (when P * C is a multiple of 4; P = 5 with an odd C
    takes the generic kernels): voxels, pmask, coords bit-exact, aggregates bit-exact within max_points, overflow voxels (the
    400-point voxels next to the sensor and a planted one of 3000 points) to the fp32 bound"""
    from d3d_amd import synth
    from d3d_amd.voxel import VoxelGenerator
    base = synth.lidar_like(150000, 21)
    rng = np.random.default_rng(c)
    cloud = np.concatenate([base[:, :3], rng.random((len(base), c - 3), dtype=np.float32) * 10 - 5], 1).astype(np.float32)
    cloud[1000:4000, :3] = cloud[999, :3] + rng.random((3000, 3), dtype=np.float32) * 0.01       # one heavy voxel
    kw = dict(reduction=reduction, max_points=P, max_voxels=90000, dense=True)
    exp = oracle.VoxelGenerator(synth.KITTI_BOUNDS, [352, 400, 20], **kw)(cloud)
    ret = _np(VoxelGenerator(synth.KITTI_BOUNDS, [352, 400, 20], **kw)(torch.from_numpy(cloud).cuda()))
    check_dense(ret, exp, P)
    assert exp["voxel_npoints"].max() > 2000 and len(exp["coords"]) > 30000


def test_sparse_vs_oracle_100k():
    from d3d_amd import synth
    from d3d_amd.voxel import VoxelGenerator
    cloud = synth.lidar_like(100000, 13)
    for kw in [dict(max_points=4, max_points_filter="trim"),
               dict(max_points=4, max_points_filter="trim", min_points=2, max_voxels=5000, max_voxels_filter="trim"),
               dict(max_voxels=3000, max_voxels_filter="descending", max_points=3, max_points_filter="trim"),
               dict()]:
        exp = oracle.VoxelGenerator(synth.KITTI_BOUNDS, [352, 400, 20], **kw)(cloud)
        ret = _np(VoxelGenerator(synth.KITTI_BOUNDS, [352, 400, 20], **kw)(torch.from_numpy(cloud).cuda()))
        check_sparse(ret, exp)


def test_empty_and_tiny_inputs():
    from d3d_amd.voxel import VoxelGenerator
    unit = [0, 1, 0, 1, 0, 1]
    e = torch.zeros((0, 4), dtype=torch.float32)
    d = VoxelGenerator(unit, [10, 10, 10], dense=True, reduction="mean", max_points=5)(e.cuda())
    assert d.voxels.shape == (0, 5, 4) and d.coords.shape == (0, 3) and d.aggregates.shape == (0, 4)
    s = VoxelGenerator(unit, [10, 10, 10])(e.cuda())
    assert s.points.shape == (0, 4) and s.coords.shape == (0, 3)
    one = torch.tensor([[0.55, 0.15, 0.95, 7.0]])
    d = VoxelGenerator(unit, [10, 10, 10], dense=True, reduction="max", max_points=2)(one.cuda())
    assert d.coords.tolist() == [[5, 1, 9]] and d.voxel_npoints.tolist() == [1]
    assert d.aggregates.cpu().tolist() == [[0.55, 0.15, 0.95, 7.0]] or np.allclose(d.aggregates.cpu(), one)
    # all points out of range
    far = torch.full((100, 4), 5.0)
    d = VoxelGenerator(unit, [10, 10, 10], dense=True)(far.cuda())
    assert d.voxels.shape[0] == 0


def test_all_points_in_one_voxel_and_errors():
    from d3d_amd.voxel import VoxelGenerator
    unit = [0, 1, 0, 1, 0, 1]
    rng = np.random.default_rng(5)
    cloud = (0.5 + 0.01 * rng.random((50000, 4))).astype(np.float32)
    exp = oracle.VoxelGenerator(unit, [10, 10, 10], dense=True, reduction="mean", max_points=16, max_voxels=10)(cloud)
    ret = _np(VoxelGenerator(unit, [10, 10, 10], dense=True, reduction="mean", max_points=16, max_voxels=10)(
        torch.from_numpy(cloud).cuda()))
    check_dense(ret, exp, 16)
    exp = oracle.VoxelGenerator(unit, [10, 10, 10], max_points=16, max_points_filter="trim")(cloud)
    ret = _np(VoxelGenerator(unit, [10, 10, 10], max_points=16, max_points_filter="trim")(torch.from_numpy(cloud).cuda()))
    check_sparse(ret, exp)
    with pytest.raises(ValueError):
        VoxelGenerator([0.05, 1, 0, 1, 0, 1], [10, 10, 10])
    with pytest.raises(ValueError):
        VoxelGenerator(unit, [10, 10, 10], reduction="mean")
    with pytest.raises(ValueError):
        VoxelGenerator(unit, [10, 10, 10], max_points_filter="bogus")
    with pytest.raises(NotImplementedError):
        VoxelGenerator(unit, [10, 10, 10], min_points=1, dense=True)
    # non-finite points: the reference's (int)floor(NaN) = INT_MIN voxel (voxelize.cpp:309) is reproduced by the raw sparse
    # function and dropped by the coordinate-bound filter (goldens sp_nonfinite*); a FINITE coordinate beyond the 21-bit key
    # range is reported by the raw function and dropped by VoxelGenerator
    from d3d_amd.voxel import voxelize_3d_sparse
    bad = np.array([[np.nan, 0, 0, 0], [0.55, 0.55, 0.55, 1], [0.05, np.inf, -np.inf, 2], [np.nan, 0.01, 0.02, 3]], np.float32)
    got = voxelize_3d_sparse(torch.from_numpy(bad).cuda(), [0.1, 0.1, 0.1])
    exp = oracle.voxelize_3d_sparse(bad, np.array([0.1, 0.1, 0.1], np.float32))
    for k in ("points_mapping", "coords", "voxel_npoints"):
        assert np.array_equal(got[k].cpu().numpy(), exp[k]), k
    assert exp["coords"].min() == -2147483648 and len(exp["coords"]) == 3
    ret = VoxelGenerator(unit, [10, 10, 10])(torch.from_numpy(bad).cuda())
    assert ret.points_mask.tolist() == [1] and ret.coords.tolist() == [[5, 5, 5]] and ret.points_mapping.tolist() == [0]
    far = torch.tensor([[2.0e5, 0, 0, 0], [0.5, 0.5, 0.5, 1]]).cuda()            # coordinate 2e6 > 2^20: any int is a voxel
    got = voxelize_3d_sparse(far, [0.1, 0.1, 0.1])                               # (voxelize.cpp:309; ValueError until round 5)
    assert got["coords"].tolist() == [[2000000, 0, 0], [5, 5, 5]] and got["points_mapping"].tolist() == [0, 1]
    ret = VoxelGenerator(unit, [10, 10, 10])(far)
    assert ret.points_mask.tolist() == [1] and ret.coords.tolist() == [[5, 5, 5]]


def test_full_size_cfg2_properties():
    """BASELINE config 2 at full size: size-independent properties + oracle on indices."""
    from d3d_amd import synth
    from d3d_amd.voxel import VoxelGenerator
    cloud = synth.lidar_like(1000000, 0)
    pts = torch.from_numpy(cloud).cuda()
    gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, dense=True, reduction="mean", max_points=32,
                         max_voxels=1000000)
    d = gen(pts)
    V = d.coords.shape[0]
    assert int(d.voxel_npoints.sum()) == 1000000            # every generated point is in range
    assert int(d.voxel_pmask.sum()) == int(torch.clamp(d.voxel_npoints, max=32).sum())
    lin = (d.coords[:, 0] * 800 + d.coords[:, 1]) * 40 + d.coords[:, 2]
    assert torch.unique(lin).numel() == V                    # voxels are unique
    # stored points fall in their voxel (test_voxel.py:23-26 invariant)
    first = d.voxels[:, 0, :3]
    cc = ((first - torch.tensor([0, -40, -3.0], device="cuda")) / 0.1).long()
    assert (torch.abs(cc - d.coords) <= 1).all()
    # checksum of checksums: sum of all stored features == sum over points kept
    # voxel mean * count == feature sum (fp64 reference)
    tot = (d.aggregates.double() * d.voxel_npoints.double()[:, None]).sum(0)
    assert torch.allclose(tot, pts.double().sum(0), rtol=1e-6)
    # ... and EVERY output against the oracle at the stated parameters (max_points = 32, MEAN): 0.5 s of CPU
    exp = oracle.voxelize_3d_dense(cloud, synth.KITTI_SHAPE, synth.KITTI_BOUNDS, 32, 1000000, 1)
    check_dense(_np(d), exp, 32)
    # a second frame through the same generator
    cloud2 = synth.lidar_like(1000000, 1)
    exp2 = oracle.voxelize_3d_dense(cloud2, synth.KITTI_SHAPE, synth.KITTI_BOUNDS, 32, 1000000, 1)
    check_dense(_np(gen(torch.from_numpy(cloud2).cuda())), exp2, 32)
    check_dense(_np(gen(pts)), exp, 32)


@pytest.mark.parametrize("P", [32, 5, 1, 70])
def test_sequence_of_frames_through_one_generator(P):
    """one generator, frames whose voxel counts grow, shrink, vanish and come back; outputs poisoned before every call (the
    fixture), so a row the fused output kernel skips shows as NaN; every frame equals the oracle bit for bit.  P = 70:
    voxels with more rows than a wavefront has lanes."""
    from d3d_amd import synth
    from d3d_amd.voxel import VoxelGenerator
    shape = [352, 400, 20]
    kw = dict(reduction="mean", max_points=P, max_voxels=150000, dense=True)
    gen = VoxelGenerator(synth.KITTI_BOUNDS, shape, **kw)
    ora = oracle.VoxelGenerator(synth.KITTI_BOUNDS, shape, **kw)
    frames = [synth.lidar_like(120000, 31), synth.lidar_like(200000, 32), synth.lidar_like(3000, 34), np.zeros((0, 4), np.float32),
              synth.uniform_cloud(150000, 35), synth.lidar_like(150000, 36)]
    for cloud in frames:
        check_dense(_np(gen(torch.from_numpy(cloud).cuda())), ora(cloud), P)


def test_plain_slot_layout_matches(monkeypatch):
    """the general (unpacked) hash-slot layout used for n >= 2^24 points gives the same results"""
    from d3d_amd import _lib, synth, voxel
    from d3d_amd.voxel import VoxelGenerator
    set_opts(voxel_flags=cur_opts().voxel_flags | _lib.VOXEL_PLAIN_SLOTS)
    cloud = synth.lidar_like(60000, 17)
    kw = dict(reduction="mean", max_points=6, max_voxels=30000, dense=True)
    exp = oracle.VoxelGenerator(synth.KITTI_BOUNDS, [352, 400, 20], **kw)(cloud)
    ret = _np(VoxelGenerator(synth.KITTI_BOUNDS, [352, 400, 20], **kw)(torch.from_numpy(cloud).cuda()))
    check_dense(ret, exp, 6)


def test_packed_slot_count_overflow_falls_back():
    """grid 2000^3 (33 key bits) x 2^20 points (20 index bits) leaves 11 count bits: a voxel with 5000 points
    overflows the packed slot and the operator transparently repeats with the general layout"""
    from d3d_amd.voxel import VoxelGenerator
    rng = np.random.default_rng(21)
    n = 1 << 20
    cloud = rng.random((n, 4), dtype=np.float32)
    cloud[:5000, :3] = 0.25 + 1e-5 * rng.random((5000, 3), dtype=np.float32)   # one heavy voxel
    unit = [0, 1, 0, 1, 0, 1]
    kw = dict(dense=True, reduction="max", max_points=2, max_voxels=n)
    ret = _np(VoxelGenerator(unit, [2000, 2000, 2000], **kw)(torch.from_numpy(cloud).cuda()))
    exp = oracle.VoxelGenerator(unit, [2000, 2000, 2000], **kw)(cloud)
    assert exp["voxel_npoints"].max() >= 4000
    check_dense(ret, exp, 2)


@pytest.mark.parametrize("seed", range(12))
def test_randomized_configs_vs_oracle(seed):
    """seeded sweep over C, max_points (incl. non powers of two and 0-ish sizes), max_voxels caps, reductions,
    grid shapes and both contracts -- every output compared with the oracle"""
    from d3d_amd.voxel import VoxelGenerator
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 30000))
    c = int(rng.integers(3, 9))
    shape = [int(rng.integers(1, 40)) for _ in range(3)]
    size = [float(rng.choice([0.25, 0.5, 1.0, 2.0])) for _ in range(3)]
    lo = [float(rng.integers(-5, 5)) * size[d] for d in range(3)]        # grid aligned with the origin
    bounds = [lo[0], lo[0] + size[0] * shape[0], lo[1], lo[1] + size[1] * shape[1], lo[2], lo[2] + size[2] * shape[2]]
    span = np.array([bounds[1] - bounds[0], bounds[3] - bounds[2], bounds[5] - bounds[4]])
    cloud = rng.random((n, c)).astype(np.float32)
    cloud[:, :3] = (np.array(lo) - 0.1 * span + cloud[:, :3] * 1.2 * span).astype(np.float32)   # 20 % outside
    if seed % 3 == 0:
        cloud[: n // 2, :3] = cloud[0, :3] + 0.01 * rng.random((n // 2, 3))                   # a heavy voxel
    P = int(rng.choice([1, 2, 3, 5, 16, 30, 32, 40]))
    mv = int(rng.choice([n + 5, max(1, n // 50), 7]))
    red = str(rng.choice(["none", "mean", "max", "min"]))
    pts = torch.from_numpy(cloud).cuda()
    kw = dict(reduction=red, max_points=P, max_voxels=mv, dense=True)
    exp = oracle.VoxelGenerator(bounds, shape, **kw)(cloud)
    ret = _np(VoxelGenerator(bounds, shape, **kw)(pts))
    check_dense(ret, exp, P)
    if True:            # the resident output (d3d_voxelize_3d_dense_resident), rows of 3 .. 8 floats: five frames through ONE buffer, each vs the oracle
        rgen, ogen = VoxelGenerator(bounds, shape, resident=True, **kw), oracle.VoxelGenerator(bounds, shape, **kw)
        for frame in (cloud, cloud[::3].copy(), cloud[::-1].copy(), cloud[: max(1, n // 7)].copy(), cloud):
            check_dense(_np(rgen(torch.from_numpy(frame).cuda())), ogen(frame), P)
    kw = dict(max_points=P, max_points_filter=str(rng.choice(["trim", "none"])), min_points=int(rng.integers(0, 3)),
              max_voxels=mv, max_voxels_filter=str(rng.choice(["trim", "none", "descending"])))
    exp = oracle.VoxelGenerator(bounds, shape, **kw)(cloud)
    ret = _np(VoxelGenerator(bounds, shape, **kw)(pts))
    check_sparse(ret, exp)


@pytest.mark.parametrize("kw", [dict(max_points=4, max_points_filter="trim"),
                                dict(max_points=3, max_points_filter="trim", min_points=2, max_voxels=700, max_voxels_filter="trim"),
                                dict(max_points=4, max_points_filter="trim", min_points=2, max_voxels=900, max_voxels_filter="descending"),
                                dict(max_voxels=100000, max_voxels_filter="descending"),
                                dict()])
def test_chained_sparse_filter_equals_the_two_calls(kw):
    """VoxelGenerator's sparse mode chains sparse -> filter with one read-back (d3d_voxelize_3d_filter_chained, the voxel
    count stays on the device): same result as the reference's two separate calls (voxel/__init__.py:93-102)"""
    from d3d_amd.voxel import VoxelGenerator, voxelize_3d_filter, voxelize_3d_sparse
    rng = np.random.default_rng(5)
    cloud = np.concatenate([rng.random((30000, 4)), rng.random((4000, 4)) * 0.1 + 0.45]).astype(np.float32)
    pts = torch.from_numpy(cloud).cuda()
    gen = VoxelGenerator([0, 1, 0, 1, 0, 1], [20, 20, 20], **kw)
    one = _np(gen(pts))
    sp = voxelize_3d_sparse(pts, gen._size_h, 3)
    two = _np(voxelize_3d_filter(pts, sp["points_mapping"], sp["coords"], sp["voxel_npoints"], gen._vbounds, gen._min_points,
                                 gen._max_points, gen._max_voxels, gen._max_points_filter, gen._max_voxels_filter))
    two["coords"] = two["coords"] - gen._offset.numpy()
    assert set(one) == set(two)
    for k in one:
        assert np.array_equal(one[k], two[k]), k


@pytest.mark.parametrize("max_voxels", [7, 300, 5000, 10 ** 6])
@pytest.mark.parametrize("max_points_filter", ["trim", "none"])
def test_descending_filter_with_many_crowded_voxels(max_voxels, max_points_filter):
    """the fused DESCENDING filter sorts the voxels' counts in one counting pass on min(count, 255) and ranks the voxels of 255
    points or more among themselves (sort.hip: k_cs_scatter, k_cs_rank_big): 700 crowded voxels with counts around the class
    boundary (253 .. 257) and far above it, most of them tied (voxelize.cpp:406 keeps ties in first-seen order), 60 k ordinary
    ones; the cut falls inside the crowded class, behind it, nowhere."""
    from d3d_amd.voxel import VoxelGenerator
    rng = np.random.default_rng(max_voxels % 97)
    shape, bounds = [64, 64, 16], [0, 64, 0, 64, 0, 16]
    cells = rng.choice(64 * 64 * 16, 700, replace=False)
    counts = rng.choice([253, 254, 255, 256, 257, 300, 300, 300, 511, 512, 1200], 700)
    rows = []
    for cell, k in zip(cells, counts):
        corner = np.array([cell // (64 * 16), (cell // 16) % 64, cell % 16], np.float32)
        rows.append(np.concatenate([corner + 0.05 + 0.9 * rng.random((k, 3)), rng.random((k, 1))], 1))
    rows.append(rng.random((120000, 4)) * np.array([64, 64, 16, 1]))
    cloud = np.concatenate(rows).astype(np.float32)
    cloud = cloud[rng.permutation(len(cloud))]
    kw = dict(max_points=5, max_points_filter=max_points_filter, min_points=1, max_voxels=max_voxels, max_voxels_filter="descending")
    exp = oracle.VoxelGenerator(bounds, shape, **kw)(cloud)
    ret = _np(VoxelGenerator(bounds, shape, **kw)(torch.from_numpy(cloud).cuda()))
    check_sparse(ret, exp)
    assert (exp["voxel_npoints"] >= 255).sum() >= min(max_voxels, 400) or max_points_filter == "trim"


@pytest.mark.parametrize("red", ["none", "mean", "max"])
@pytest.mark.parametrize("P,C", [(4, 4), (32, 4), (32, 5), (8, 3), (12, 7)])
def test_resident_dense_output_over_a_sequence_of_frames(red, P, C, index_path):
    """VoxelGenerator(dense=True, resident=True) (d3d_voxelize_3d_dense_resident): `voxels` is a view of a buffer kept on the
    device, of which a call stores only the rows with points and zeros over what the previous frame left under the same voxel
    id.  Ten frames of very different shape -- dense, sparse, one crowded voxel, a single point, empty, growing past the
    buffer's capacity, max_voxels cutting -- each equal to the oracle (= the reference's fresh, zero-padded tensor) bit for
    bit on EVERY output, and the buffer's invariant (rows at and beyond a voxel's count are zero, everywhere) after each."""
    from d3d_amd.voxel import VoxelGenerator
    rng = np.random.default_rng(P)
    bounds, shape = [0, 8, 0, 8, 0, 2], [40, 40, 8]
    gen = VoxelGenerator(bounds, shape, max_points=P, max_voxels=9000, reduction=red, dense=True, resident=True)
    ref = oracle.VoxelGenerator(bounds, shape, max_points=P, max_voxels=9000, reduction=red, dense=True)
    span = np.array([8, 8, 2] + [1] * (C - 3), np.float32)

    def cloud(n, clump=0.0, where=0.5):
        c = rng.random((n, C)).astype(np.float32) * span
        k = int(n * clump)
        c[:k, :3] = (np.array([8, 8, 2]) * where + 0.15 * rng.random((k, 3))).astype(np.float32)
        return c[rng.permutation(n)]

    frames = [cloud(6000), cloud(60000, 0.3), cloud(900), cloud(1), cloud(20000, 0.9, 0.2), np.zeros((0, C), np.float32),
              cloud(5000, 0.5, 0.8), cloud(200000), cloud(7000, 0.2), cloud(40000, 0.1, 0.3)]
    for k, f in enumerate(frames):
        exp = ref(f)
        got = gen(torch.from_numpy(f).cuda())
        check_dense(_np(got), exp, P)
        buf = gen._resident_buf
        if buf is not None and len(f):
            # (the hash path and the two-launch output stage write whole tensors: a fresh one then, the buffer is left alone)
            aliased = got["voxels"].untyped_storage().data_ptr() == buf.voxels.untyped_storage().data_ptr()
            assert aliased == (index_path in ("auto", "3pass", "v1")), k
            state = buf.row_state.cpu().numpy().astype(np.int64) & 0xffff
            nz = (buf.voxels != 0).any(dim=2).cpu().numpy()                     # [capacity, P]: rows that hold anything
            assert not (nz & (np.arange(P)[None, :] >= state[:, None])).any(), k
    assert gen._resident_buf.capacity == 9000                                   # (grew from 6000 when the 60000-point frame came)


def test_resident_dense_c_abi_contract():
    """d3d_voxelize_3d_dense_resident through ctypes: rows of 9 floats, the hash path and SPLIT_FILL are D3D_ERR_UNSUPPORTED
    and touch nothing; a missing row_state is a bad argument; two frames through ONE buffer equal the plain entry point."""
    import ctypes
    from d3d_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    shape = (ctypes.c_int32 * 3)(32, 32, 4)
    bound = (ctypes.c_float * 6)(0, 1, 0, 1, 0, 1)
    P, MV = 16, 5000
    vox = torch.zeros((MV, P, 4), device="cuda")
    state = torch.zeros((MV,), dtype=torch.int16, device="cuda")

    def outs(c=4):
        return (torch.full((MV, 3), -5, dtype=torch.int64, device="cuda"), torch.full((MV, P), 7, dtype=torch.uint8, device="cuda"),
                torch.full((MV,), -5, dtype=torch.int32, device="cuda"), torch.full((MV, c), -5.0, device="cuda"),
                torch.zeros((_lib.NUM_COUNTS,), dtype=torch.int64, device="cuda"))

    def call(pts, c, flags=0, row_state=state, resident=True, voxels=vox):
        coords, pmask, npts, agg, counts = outs(c)
        n = pts.shape[0]
        ws = _lib.workspace(lib.d3d_voxelize_workspace_bytes(n, 0), pts.device)
        a = [_lib.ptr(pts), n, c, ctypes.cast(shape, ctypes.c_void_p), ctypes.cast(bound, ctypes.c_void_p), P, MV, 1, _lib.ptr(voxels)]
        b = [_lib.ptr(coords), _lib.ptr(pmask), _lib.ptr(npts), _lib.ptr(agg), _lib.ptr(counts), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()]
        if resident:
            rc = lib.d3d_voxelize_3d_dense_resident(*a, _lib.ptr(row_state) if row_state is not None else None, *b, None, flags)
        else:
            rc = lib.d3d_voxelize_3d_dense(*a, *b, flags)
        torch.cuda.synchronize()
        return rc, coords, pmask, npts, agg, counts

    p9 = torch.from_numpy(rng.random((3000, 9)).astype(np.float32)).cuda()
    p4 = torch.from_numpy(rng.random((3000, 4)).astype(np.float32)).cuda()
    for rc, *o in (call(p9, 9), call(p4, 4, _lib.VOXEL_PATH_HASH), call(p4, 4, _lib.VOXEL_SPLIT_FILL)):
        assert rc == _lib.ERR_UNSUPPORTED
        assert int(vox.count_nonzero()) == 0 and int(state.count_nonzero()) == 0 and int((o[2] != -5).sum()) == 0
    assert call(p4, 4, row_state=None)[0] == _lib.ERR_BAD_ARG
    for n in (3000, 700):
        pts = torch.from_numpy(rng.random((n, 4)).astype(np.float32)).cuda()
        rc, coords, pmask, npts, agg, counts = call(pts, 4)
        assert rc == _lib.OK
        fresh = torch.full((MV, P, 4), float("nan"), device="cuda")
        rc2, coords2, pmask2, npts2, agg2, counts2 = call(pts, 4, resident=False, voxels=fresh)
        nv = int(counts[_lib.COUNT_VOXELS])
        assert rc2 == _lib.OK and nv == int(counts2[_lib.COUNT_VOXELS]) and nv > 0
        assert torch.equal(vox[:nv], fresh[:nv]) and torch.equal(coords[:nv], coords2[:nv]) and torch.equal(npts[:nv], npts2[:nv])
        assert torch.equal(pmask[:nv], pmask2[:nv]) and torch.equal(agg[:nv], agg2[:nv])


def test_descending_filter_when_every_voxel_is_crowded():
    """the adversarial frame for the counting sort of the DESCENDING filter: 15 000 voxels of 255 .. 269 points each (3.9 M
    points) -- ALL of them in the class the counting pass hands to the all-pairs ranking (k_cs_rank_big), whose side list is
    sized by points / 255; ties everywhere, the cut in the middle"""
    from d3d_amd.voxel import VoxelGenerator
    rng = np.random.default_rng(1)
    shape, bounds = [64, 64, 16], [0, 64, 0, 64, 0, 16]
    cells = rng.choice(64 * 64 * 16, 15000, replace=False)
    rows = []
    for cell, k in zip(cells, rng.integers(255, 270, 15000)):
        corner = np.array([cell // (64 * 16), (cell // 16) % 64, cell % 16], np.float32)
        rows.append(np.concatenate([corner + 0.05 + 0.9 * rng.random((k, 3)), rng.random((k, 1))], 1))
    cloud = np.concatenate(rows).astype(np.float32)
    cloud = cloud[rng.permutation(len(cloud))]
    kw = dict(max_points=5, max_points_filter="trim", min_points=1, max_voxels=7000, max_voxels_filter="descending")
    check_sparse(_np(VoxelGenerator(bounds, shape, **kw)(torch.from_numpy(cloud).cuda())), oracle.VoxelGenerator(bounds, shape, **kw)(cloud))


def test_sparse_bounding_box_key_and_its_fallback():
    """sparse contract: one-word hash slots keyed inside the frame's bounding box (k_bbox); a box too large for the key
    field (far outliers on every axis) raises PACK_OVERFLOW and the call is repeated with plain slots -- same result"""
    from d3d_amd.voxel import voxelize_3d_sparse
    rng = np.random.default_rng(17)
    cloud = (rng.random((20000, 4)) * [40, 40, 4, 1] - [20, 20, 2, 0]).astype(np.float32)
    size = np.array([0.1, 0.1, 0.1], np.float32)
    for outliers in (False, True):
        pts = cloud.copy()
        if outliers:
            pts[:6, :3] = [[9e4, 0, 0], [-9e4, 0, 0], [0, 9e4, 0], [0, -9e4, 0], [0, 0, 9e4], [0, 0, -9e4]]
        exp = oracle.voxelize_3d_sparse(pts, size)
        got = _np(voxelize_3d_sparse(torch.from_numpy(pts).cuda(), torch.from_numpy(size), 3))
        for k in ("points_mapping", "coords", "voxel_npoints"):
            assert np.array_equal(got[k], exp[k]), (outliers, k)
    one = _np(voxelize_3d_sparse(torch.from_numpy(cloud[:1]).cuda(), torch.from_numpy(size), 3))
    assert one["coords"].shape == (1, 3) and one["voxel_npoints"].tolist() == [1]


def test_results_are_reproducible_run_to_run():
    """atomics decide arrival order, never results: five runs of the dense (MEAN, incl. overflow voxels summed in fp64),
    sparse + trim and reduce paths are bit-identical"""
    from d3d_amd import synth
    from d3d_amd.voxel import VoxelGenerator
    from d3d_amd.voxel.sharded import voxelize_reduce
    pts = torch.from_numpy(synth.lidar_like(300000, 21)).cuda()
    dense = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, dense=True, reduction="mean", max_points=8, max_voxels=300000)
    sparse = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, max_points=8, max_points_filter="trim")
    first = None
    for _ in range(5):
        d, s = dense(pts), sparse(pts)
        r = voxelize_reduce(pts, synth.KITTI_SHAPE, synth.KITTI_BOUNDS, "mean")
        cur = [d.voxels, d.coords, d.voxel_npoints, d.aggregates, s.points, s.points_mask, s.points_mapping, s.coords,
               r.coords, r.aggregates, r.points_mapping]
        if first is None:
            first = [t.clone() for t in cur]
            assert int((d.voxel_npoints > 8).sum()) > 100          # overflow voxels are present
        else:
            for idx, (a, b) in enumerate(zip(first, cur)):
                if idx in (3, 9):       # aggregates: overflow voxels are summed in fp64 in arrival order (1e-16 relative)
                    fit = (d.voxel_npoints if idx == 3 else r.voxel_npoints) <= (8 if idx == 3 else 32)
                    assert torch.equal(a[fit], b[fit]) and torch.allclose(a, b, rtol=1e-6, atol=0)
                else:
                    assert torch.equal(a, b)


# ---------------------------------------------------------------- dense contract: binned index path
@pytest.fixture
def voxel_path():
    """setter: 1 = hash table, 0 / 2 = binned index wherever eligible (the default)"""
    from d3d_amd import _lib, voxel

    def set_path(path):
        set_opts(voxel_flags=_lib.VOXEL_PATH_HASH if path == 1 else 0)
    yield set_path


@pytest.mark.parametrize("n,P,mv,reduction", [(1, 4, 10, "mean"), (7, 1, 7, "max"), (300, 3, 100, "min"), (5000, 32, 5000, "mean"),
                                              (40000, 5, 1500, "none"), (70000, 2, 70000, "mean")])
def test_binned_index_vs_oracle(voxel_path, n, P, mv, reduction):
    """forced onto the binned path at every size (automatic from 32 k points): same outputs as the reference's loop"""
    from d3d_amd import synth
    from d3d_amd.voxel import VoxelGenerator
    cloud = synth.lidar_like(n, 31 + n)
    kw = dict(reduction=reduction, max_points=P, max_voxels=mv, dense=True)
    exp = oracle.VoxelGenerator(synth.KITTI_BOUNDS, [352, 400, 20], **kw)(cloud)
    voxel_path(2)
    ret = _np(VoxelGenerator(synth.KITTI_BOUNDS, [352, 400, 20], **kw)(torch.from_numpy(cloud).cuda()))
    check_dense(ret, exp, P)


@pytest.mark.parametrize("reduction", ["none", "mean", "max"])
def test_binned_and_hash_index_paths_agree(voxel_path, reduction):
    """250 k points with NaNs, out-of-range points and exact duplicates: every output tensor identical on both paths"""
    from d3d_amd import synth
    from d3d_amd.voxel import VoxelGenerator
    g = torch.Generator().manual_seed(5)
    pts = torch.rand((250000, 4), generator=g) * torch.tensor([90.0, 100.0, 6.0, 1.0]) - torch.tensor([10.0, 50.0, 4.0, 0.0])
    pts[::97, 0] = float("nan")
    pts[1000:3000] = pts[0:2000].clone()
    pts = pts.cuda()
    out = []
    for path in (1, 2):
        voxel_path(path)
        out.append(_np(VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, reduction=reduction, max_points=3, max_voxels=60000,
                                      dense=True)(pts)))
    assert out[0]["coords"].shape[0] == 60000
    for k in out[0]:
        if k == "aggregates":
            assert np.allclose(out[0][k], out[1][k], rtol=1e-6, atol=1e-6, equal_nan=True), k
        else:
            assert np.array_equal(out[0][k], out[1][k], equal_nan=True), k


def _dense_call(lib, pts, shape3, P, red, path_setter, path):
    """raw C-ABI call -> (status bits, voxel count)"""
    import ctypes
    from d3d_amd import _lib, synth
    n = pts.shape[0]
    shape = (ctypes.c_int32 * 3)(*shape3)
    bound = (ctypes.c_float * 6)(*synth.KITTI_BOUNDS)
    voxels = torch.empty((n, P, 4), device="cuda")
    coords = torch.empty((n, 3), dtype=torch.int64, device="cuda")
    pmask = torch.empty((n, P), dtype=torch.uint8, device="cuda")
    npts = torch.empty((n,), dtype=torch.int32, device="cuda")
    agg = torch.empty((n, 4), device="cuda")
    counts = torch.empty((_lib.NUM_COUNTS,), dtype=torch.int64, device="cuda")
    ws = _lib.workspace(lib.d3d_voxelize_workspace_bytes(n, 0), pts.device)
    rc = lib.d3d_voxelize_3d_dense(_lib.ptr(pts), n, 4, ctypes.cast(shape, ctypes.c_void_p), ctypes.cast(bound, ctypes.c_void_p),
                                   P, n, red, _lib.ptr(voxels), _lib.ptr(coords), _lib.ptr(pmask), _lib.ptr(npts), _lib.ptr(agg),
                                   _lib.ptr(counts), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(),
                                   _lib.VOXEL_PATH_HASH if path == 1 else 0)
    assert rc == 0
    host = counts.cpu()
    return int(host[_lib.COUNT_STATUS]), int(host[_lib.COUNT_VOXELS])


def test_binned_big_buckets_are_indexed_in_place(voxel_path):
    """300 k points in 128 cells: every bucket of the partition holds thousands of points; the bucket kernel switches to
    its looping mode (state in global memory) instead of giving up -- no status bit, same result as the reference"""
    from d3d_amd import _lib, synth
    from d3d_amd.voxel import voxelize_3d_dense
    lib = _lib.load()
    cloud = synth.lidar_like(300000, 41)
    pts = torch.from_numpy(cloud).cuda()
    status, nv = _dense_call(lib, pts, [8, 8, 2], 4, 1, voxel_path, 2)
    assert status == 0 and nv == 128
    for red in (1, 2, 0):
        voxel_path(2)
        ret = {k: v.cpu().numpy() for k, v in voxelize_3d_dense(pts, [8, 8, 2], synth.KITTI_BOUNDS, 4, 300000, red).items()}
        exp = oracle.voxelize_3d_dense(cloud, [8, 8, 2], synth.KITTI_BOUNDS, 4, 300000, red)
        check_dense(ret, exp, 4)


def test_binned_bucket_with_too_many_cells_repeats_on_the_hash_path(voxel_path):
    """3000 distinct cells chosen to fall into ONE bucket (same low bits of the cell hash): more cells than the bucket's
    LDS table has slots -> the C ABI reports BIN_OVERFLOW and the operator repeats the call with the hash-table index"""
    from d3d_amd import _lib, synth
    from d3d_amd.voxel import voxelize_3d_dense
    lib = _lib.load()
    n, nbins = 40000, 128                                   # 40 k points -> 128 buckets (voxel.hip binned_eligible)
    keys = np.arange(704 * 800 * 40, dtype=np.uint32)
    h = keys.copy()
    h ^= h >> 16; h *= np.uint32(0x85ebca6b); h ^= h >> 13; h *= np.uint32(0xc2b2ae35); h ^= h >> 16    # DenseKey::bin_hash
    cells = keys[(h & (nbins - 1)) == 5][:3000].astype(np.int64)
    cx, cy, cz = cells // (800 * 40), (cells // 40) % 800, cells % 40
    cloud = synth.lidar_like(n, 47)
    cloud[:3000, 0] = (cx + 0.5) * 0.1
    cloud[:3000, 1] = (cy + 0.5) * 0.1 - 40.0
    cloud[:3000, 2] = (cz + 0.5) * 0.1 - 3.0
    pts = torch.from_numpy(cloud).cuda()
    status, _ = _dense_call(lib, pts, synth.KITTI_SHAPE, 4, 1, voxel_path, 2)
    assert status & _lib.STATUS_BIN_OVERFLOW
    voxel_path(0)                                           # automatic: binned first, then the retry
    ret = {k: v.cpu().numpy() for k, v in voxelize_3d_dense(pts, synth.KITTI_SHAPE, synth.KITTI_BOUNDS, 4, n, 1).items()}
    exp = oracle.voxelize_3d_dense(cloud, synth.KITTI_SHAPE, synth.KITTI_BOUNDS, 4, n, 1)
    check_dense(ret, exp, 4)


@pytest.mark.parametrize("P,reduction", [(32, "mean"), (1, "max"), (5, "none"), (64, "min"), (100, "mean"), (256, "max"), (72, "none")])
@pytest.mark.parametrize("layout", ["shuffled", "front", "back", "runs"])
def test_crowded_cells_ranked_by_one_wavefront(P, reduction, layout):
    """cells of 65 .. 2048 points in a register bucket: round 5's bucket kernel ranks each of them with ONE wavefront (radix-64
    select of the max_points smallest point indices, all-pairs ranks among those) and leaves a record only from 255 points
    on.  Crowded cells of every size class (64 / 65, 254 / 255 / 256, a full bucket), their points spread over the frame, at
    its front, at its end or in runs of consecutive indices (the select's bins then hold everything or nothing), for max_points
    below, at and above the cells' sizes -- all outputs against the oracle"""
    from d3d_amd import synth
    from d3d_amd.voxel import VoxelGenerator
    n = 120000
    cloud = synth.lidar_like(n, 77)
    rng = np.random.default_rng(78)
    sizes = [64, 65, 66, 100, 128, 129, 254, 255, 256, 257, 400, 1000, 1500, 1900]
    homes = rng.choice(n, len(sizes), replace=False)
    at = 0
    order = rng.permutation(n) if layout == "shuffled" else np.arange(n) if layout in ("front", "runs") else np.arange(n)[::-1]
    for sz, h in zip(sizes, homes):
        lo3 = np.array(synth.KITTI_BOUNDS[0::2])
        centre = ((np.floor((cloud[h, :3] - lo3) / 0.1) + 0.3) * 0.1 + lo3).astype(np.float32)      # inside one 0.1 m cell
        if layout == "runs":
            start = int(rng.integers(0, n - sz))
            idx = np.arange(start, start + sz)
        else:
            idx = order[at:at + sz]
        at += sz
        cloud[idx, :3] = centre + 0.03 * rng.random((sz, 3), dtype=np.float32)
    kw = dict(reduction=reduction, max_points=P, max_voxels=n, dense=True)
    exp = oracle.VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, **kw)(cloud)
    assert exp["voxel_npoints"].max() >= 1900
    ret = _np(VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, **kw)(torch.from_numpy(cloud).cuda()))
    check_dense(ret, exp, P)


def test_sparse_contract_heavy_voxel_in_a_big_bucket():
    """100 k points, 30 k of them in one voxel: that bucket is indexed by the bucket kernel's looping mode"""
    from d3d_amd import synth
    from d3d_amd.voxel import VoxelGenerator
    cloud = synth.lidar_like(100000, 43)
    rng = np.random.default_rng(44)
    cloud[20000:50000, :3] = cloud[7, :3] + 0.01 * rng.random((30000, 3), dtype=np.float32)
    for kw in (dict(max_points=8, max_points_filter="trim", min_points=2), dict()):
        exp = oracle.VoxelGenerator(synth.KITTI_BOUNDS, [352, 400, 20], **kw)(cloud)
        ret = _np(VoxelGenerator(synth.KITTI_BOUNDS, [352, 400, 20], **kw)(torch.from_numpy(cloud).cuda()))
        assert exp["voxel_npoints"].max() >= 8 if kw else exp["voxel_npoints"].max() >= 30000
        check_sparse(ret, exp)


@pytest.mark.parametrize("n,path", [(0, 0), (500, 1), (500, 2), (60000, 1), (60000, 2)])
def test_dense_notify_publishes_the_counts_to_pinned_host_memory(voxel_path, n, path):
    """d3d_voxelize_3d_dense_notify: host_counts[0..NUM_COUNTS) == counts[], flag word set, on both index paths and for the
    empty input; outputs identical to the plain entry point"""
    import ctypes
    from d3d_amd import _lib, synth
    lib = _lib.load()
    cloud = synth.lidar_like(max(n, 1), 51)[:n]
    pts = torch.from_numpy(cloud).cuda()
    P, cap = 8, max(n, 1)
    shape = (ctypes.c_int32 * 3)(352, 400, 20)
    bound = (ctypes.c_float * 6)(*synth.KITTI_BOUNDS)
    ws = _lib.workspace(lib.d3d_voxelize_workspace_bytes(n, 0), torch.device("cuda", 0))
    outs = []
    for notify in (False, True):
        voxels = torch.zeros((cap, P, 4), device="cuda")
        coords = torch.zeros((cap, 3), dtype=torch.int64, device="cuda")
        pmask = torch.zeros((cap, P), dtype=torch.uint8, device="cuda")
        npts = torch.zeros((cap,), dtype=torch.int32, device="cuda")
        agg = torch.zeros((cap, 4), device="cuda")
        counts = torch.empty((_lib.NUM_COUNTS,), dtype=torch.int64, device="cuda")
        args = [_lib.ptr(pts), n, 4, ctypes.cast(shape, ctypes.c_void_p), ctypes.cast(bound, ctypes.c_void_p), P, cap, 1,
                _lib.ptr(voxels), _lib.ptr(coords), _lib.ptr(pmask), _lib.ptr(npts), _lib.ptr(agg), _lib.ptr(counts), _lib.ptr(ws),
                ws.numel(), _lib.stream_ptr()]
        fl = _lib.VOXEL_PATH_HASH if path == 1 else 0
        if notify:
            note = _lib.NotifyBuffer.get()
            note.arm()
            assert lib.d3d_voxelize_3d_dense_notify(*args, note.ptr, fl) == 0
            host = note.wait(counts, spin_s=5.0)
            assert note.arr[_lib.NUM_COUNTS] == 1                       # the flag itself, not the fallback read
            assert host == counts.cpu().tolist()
            assert lib.d3d_voxelize_3d_dense_notify(*args, None, fl) != 0   # the buffer is mandatory here
            assert lib.d3d_voxelize_3d_dense_notify(*args, note.ptr, 0x100) == _lib.ERR_BAD_ARG   # unknown option bit
        else:
            assert lib.d3d_voxelize_3d_dense(*args, fl) == 0
        nv = int(counts.cpu()[_lib.COUNT_VOXELS])
        outs.append((nv, voxels[:nv].cpu(), coords[:nv].cpu(), pmask[:nv].cpu(), npts[:nv].cpu(), agg[:nv].cpu()))
    assert outs[0][0] == outs[1][0] and (n == 0) == (outs[0][0] == 0)
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n", [511, 512, 513, 65536, 65537])
def test_binned_bucket_count_boundaries(voxel_path, n):
    """point counts around the powers of two that change the number of buckets (n / 512 rounded up to a power of two)"""
    from d3d_amd import synth
    from d3d_amd.voxel import VoxelGenerator
    cloud = synth.uniform_cloud(n, 61 + n)
    kw = dict(reduction="max", max_points=3, max_voxels=n, dense=True)
    exp = oracle.VoxelGenerator(synth.KITTI_BOUNDS, [88, 100, 8], **kw)(cloud)
    voxel_path(2)
    check_dense(_np(VoxelGenerator(synth.KITTI_BOUNDS, [88, 100, 8], **kw)(torch.from_numpy(cloud).cuda())), exp, 3)
    kw = dict(max_points=2, max_points_filter="trim", max_voxels=max(n // 3, 1), max_voxels_filter="trim")
    exp = oracle.VoxelGenerator(synth.KITTI_BOUNDS, [88, 100, 8], **kw)(cloud)
    check_sparse(_np(VoxelGenerator(synth.KITTI_BOUNDS, [88, 100, 8], **kw)(torch.from_numpy(cloud).cuda())), exp)


def test_no_point_inside_the_grid_and_large_max_points(voxel_path):
    """every point outside the bounds (dense contract: zero voxels; sparse + filter: zero kept points), and max_points larger
    than any voxel / than the 64-lane wavefront"""
    from d3d_amd import synth
    from d3d_amd.voxel import VoxelGenerator
    cloud = synth.lidar_like(5000, 71)
    far = cloud.copy()
    far[:, 0] += 1000.0
    for path in (1, 2):
        voxel_path(path)
        ret = VoxelGenerator(synth.KITTI_BOUNDS, [352, 400, 20], max_points=4, max_voxels=100, dense=True, reduction="mean")(
            torch.from_numpy(far).cuda())
        assert ret.coords.shape[0] == 0 and ret.voxels.shape == (0, 4, 4)
        ret = VoxelGenerator(synth.KITTI_BOUNDS, [352, 400, 20], max_points=4, max_points_filter="trim")(torch.from_numpy(far).cuda())
        assert ret.points.shape[0] == 0 and ret.coords.shape[0] == 0
        kw = dict(reduction="mean", max_points=300, max_voxels=5000, dense=True)
        exp = oracle.VoxelGenerator(synth.KITTI_BOUNDS, [44, 50, 4], **kw)(cloud)
        check_dense(_np(VoxelGenerator(synth.KITTI_BOUNDS, [44, 50, 4], **kw)(torch.from_numpy(cloud).cuda())), exp, 300)


@pytest.mark.parametrize("n", [4194304, 8388608, 8388609, 16777216, 16777217])
def test_millions_of_points_up_to_the_binned_limit(index_path, n):
    """16 M points is the largest frame the binned index takes (16384 buckets of 1024 on average; 8 M = config 5's frame on
    ONE GPU was the limit until round 3); one more goes to the hash table -- all bit-exact with the oracle"""
    if index_path == "hash" and n != 4194304:
        pytest.skip("one hash-table run of this size class is enough")
    from d3d_amd import synth
    from d3d_amd.voxel import VoxelGenerator
    cloud = synth.lidar_like(n, 81, synth.WAYMO_BOUNDS)
    P = 5 if n in (8388608, 16777216) else 4          # (frames this large take the group-per-wavefront fill: both of its row -> voxel forms)
    kw = dict(reduction="mean", max_points=P, max_voxels=n, dense=True)
    exp = oracle.VoxelGenerator(synth.WAYMO_BOUNDS, [752, 752, 30], **kw)(cloud)
    ret = _np(VoxelGenerator(synth.WAYMO_BOUNDS, [752, 752, 30], **kw)(torch.from_numpy(cloud).cuda()))
    check_dense(ret, exp, P)


def test_fused_sparse_filter_entry_error_codes():
    """d3d_voxelize_3d_sparse_filter: the DESCENDING voxel filter off the binned index needs the voxel count on the host (unsupported in the fused
    call, VoxelGenerator then issues the two calls), sparse_counts is mandatory, a short workspace is reported"""
    import ctypes
    from d3d_amd import _lib, synth
    lib = _lib.load()
    n = 5000
    pts = torch.from_numpy(synth.lidar_like(n, 91)).cuda()
    size = (ctypes.c_float * 3)(0.2, 0.2, 0.2)
    bound = (ctypes.c_int64 * 6)(0, 352, -200, 200, -15, 5)
    dev = pts.device
    i64 = lambda *s: torch.empty(s, dtype=torch.int64, device=dev)   # noqa: E731
    mapping, coords, o_mask, o_map, o_crd = i64(n), i64(n, 3), i64(n), i64(n), i64(n, 3)
    npts, o_cnt = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
    o_feats = torch.empty((n, 4), device=dev)
    counts = i64(2, _lib.NUM_COUNTS)
    ws = _lib.workspace(lib.d3d_voxelize_workspace_bytes(n, n), dev)

    def call(vf, sparse_counts, ws_bytes, fl=0):
        return lib.d3d_voxelize_3d_sparse_filter(
            _lib.ptr(pts), n, 4, ctypes.cast(size, ctypes.c_void_p), ctypes.cast(bound, ctypes.c_void_p), 0, 8, 100, 1, vf,
            _lib.ptr(mapping), _lib.ptr(coords), _lib.ptr(npts), sparse_counts, _lib.ptr(o_feats), _lib.ptr(o_mask),
            _lib.ptr(o_map), _lib.ptr(o_cnt), _lib.ptr(o_crd), _lib.ptr(counts[1]), _lib.ptr(ws), ws_bytes, _lib.stream_ptr(), None, fl, None)
    assert call(2, _lib.ptr(counts[0]), ws.numel(), _lib.VOXEL_PATH_HASH) == _lib.ERR_UNSUPPORTED   # DESCENDING off the binned index
    assert call(2, _lib.ptr(counts[0]), ws.numel()) == 0                             # ... fused on it (round 4)
    torch.cuda.synchronize()
    assert call(1, None, ws.numel()) == _lib.ERR_BAD_ARG
    assert call(1, _lib.ptr(counts[0]), 1024) == _lib.ERR_WORKSPACE
    assert call(1, _lib.ptr(counts[0]), ws.numel()) == 0
    host = counts.cpu()
    assert host[0, _lib.COUNT_STATUS] == 0 and 0 < host[1, _lib.COUNT_VOXELS] <= 100 and host[1, _lib.COUNT_POINTS] > 0


def test_pipelined_stream_of_frames_bit_exact():
    """VoxelGenerator.stream (round 4): six different frames of different sizes through the pipelined mode -- frame k + 1's
    index on a side stream under frame k's output, two frames in flight -- with poisoned outputs: every output of every frame
    equals the oracle's; then the same generator on a frame the staged path refuses (five columns) falls back to the plain call"""
    from d3d_amd import synth
    from d3d_amd.voxel import VoxelGenerator
    sizes = [60000, 200000, 1000, 350000, 200000, 77777]
    clouds = [synth.lidar_like(n, 40 + k) for k, n in enumerate(sizes)]
    kw = dict(dense=True, reduction="mean", max_points=16, max_voxels=300000)
    gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, **kw)
    ora = oracle.VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, **kw)
    frames = [torch.from_numpy(c).cuda() for c in clouds]
    got = list(gen.stream(iter(frames), poison=True, pipelined=True))
    plain = list(gen.stream(iter(frames[:2]), poison=True))                  # (default: the plain loop)
    assert torch.equal(plain[1].voxels, got[1].voxels)
    assert len(got) == len(clouds)
    for cloud, ret in zip(clouds, got):
        check_dense(_np(ret), ora(cloud), 16)
    # interleaved use: results consumed while later frames are in flight, on a stream of the caller's
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        sums = [float(r.voxels.sum()) for r in gen.stream(frames[:4], poison=True, pipelined=True)]
    for cloud, sm in zip(clouds[:4], sums):
        assert abs(sm - float(ora(cloud)["voxels"].astype(np.float64).sum())) <= 1e-3 * max(abs(sm), 1.0)
    wide = np.concatenate([clouds[0], np.ones((len(clouds[0]), 1), np.float32)], 1)
    (one,) = list(gen.stream([torch.from_numpy(wide).cuda()], poison=True, pipelined=True))
    check_dense(_np(one), ora(wide), 16)


@pytest.mark.parametrize("n,c,P", [(1000000, 4, 32), (150000, 4, 4), (120000, 5, 8), (3000, 3, 1)])
def test_exact_mean_flag_is_bit_exact_on_overflow_voxels(n, c, P):
    """D3D_VOXEL_EXACT_MEAN: MEAN of the voxels with more than max_points points summed sequentially in
    fp32 in point order, bit for bit like voxelize.cpp:142,164 -- config 2 itself (1 M points, max_points 32: ~3 k overflow
    voxels of up to 407 points), a setting where most voxels overflow, rows of 5 and 3 columns; on every index path of the
    module's fixture.  Without the flag those voxels differ from the reference by rounding (checked: the flag matters)."""
    from d3d_amd import _lib, synth
    from d3d_amd.voxel import VoxelGenerator
    base = synth.lidar_like(n, 7)
    rng = np.random.default_rng(n)
    cloud = base[:, :c] if c <= 4 else np.concatenate([base, rng.random((n, c - 4), dtype=np.float32)], 1)
    cloud = np.ascontiguousarray(cloud)
    kw = dict(dense=True, reduction="mean", max_points=P, max_voxels=n)
    exp = oracle.VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, **kw)(cloud)
    gen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, **kw)
    got = _np(gen(torch.from_numpy(cloud).cuda(), flags=cur_opts().voxel_flags | _lib.VOXEL_EXACT_MEAN))
    over = exp["voxel_npoints"] > P
    assert over.sum() > 10
    assert np.array_equal(got["coords"], exp["coords"]) and np.array_equal(got["voxels"], exp["voxels"])
    assert np.array_equal(got["aggregates"], exp["aggregates"]), int((got["aggregates"] != exp["aggregates"]).any(1).sum())
    plain = _np(gen(torch.from_numpy(cloud).cuda()))["aggregates"]
    assert np.array_equal(plain[~over], exp["aggregates"][~over])
    if n >= 100000:
        assert not np.array_equal(plain[over], exp["aggregates"][over])       # (fp64 sums: within rounding, not bit-equal)
    # ... and with the resident output (the exact-mean pass only rewrites aggregates): a different frame first, then this one
    rgen = VoxelGenerator(synth.KITTI_BOUNDS, synth.KITTI_SHAPE, resident=True, **kw)
    rgen(torch.from_numpy(np.ascontiguousarray(cloud[::-2])).cuda(), flags=cur_opts().voxel_flags | _lib.VOXEL_EXACT_MEAN)
    res = _np(rgen(torch.from_numpy(cloud).cuda(), flags=cur_opts().voxel_flags | _lib.VOXEL_EXACT_MEAN))
    assert np.array_equal(res["voxels"], exp["voxels"]) and np.array_equal(res["aggregates"], exp["aggregates"])
    assert np.array_equal(res["coords"], exp["coords"]) and np.array_equal(res["voxel_npoints"], exp["voxel_npoints"])
