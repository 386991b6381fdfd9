"""CPU: the C-ABI library loads and exports every symbol include/d3d_hip.h declares (no compute calls);
the host-side layer validates arguments like the reference without touching a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "d3d_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(d3d_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from d3d_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "libd3d_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "d3d_amd/_lib.py does not bind %s" % n
    assert set(_lib.SIGNATURES) <= set(names)
    assert _lib.load().d3d_abi_version() == 13
    assert _lib.load().d3d_status_string(-2) == b"unsupported option"


def test_workspace_queries_are_pure():
    from d3d_amd import _lib
    lib = _lib.load()
    assert lib.d3d_voxelize_workspace_bytes(1000000, 0) > 16 * 2 * 1000000
    assert lib.d3d_nms2d_workspace_bytes(100000) > 100000 * (100000 // 64) * 8
    assert lib.d3d_argsort_desc_workspace_bytes(1000, 1) > 0
    assert lib.d3d_grid_compact_workspace_bytes(704 * 800 * 40) > 704 * 800 * 40 // 8


def test_host_side_validation_without_gpu():
    import torch
    from d3d_amd.box import IouType, SupressionType, box2d_iou, box2d_nms
    from d3d_amd.voxel import MaxPointsFilterType, ReductionType, VoxelGenerator
    assert ReductionType.MIN == 3 and MaxPointsFilterType.TRIM == 1 and IouType.DRBOX == 6 and SupressionType.GAUSSIAN == 2
    g = VoxelGenerator([0, 70.4, -40, 40, -3, 1], [704, 800, 40], max_points=32)
    assert g._vbounds.tolist() == [[0, 704], [-400, 400], [-30, 10]] and g._offset.tolist() == [0, -400, -30]
    with pytest.raises(ValueError):
        VoxelGenerator([0.05, 1, 0, 1, 0, 1], [10, 10, 10])
    with pytest.raises(ValueError):
        VoxelGenerator([0, 1, 0, 1, 0, 1], [10, 10, 10], reduction="median", dense=True)
    with pytest.raises(ValueError):              # the resident output is an option of the dense contract ...
        VoxelGenerator([0, 1, 0, 1, 0, 1], [10, 10, 10], resident=True)
    from d3d_amd.voxel.sharded import ShardedVoxelGenerator
    from sharded_helpers import NumpyOps
    from d3d_amd.voxel.sharded import LocalComm
    with pytest.raises(ValueError):              # ... and, sharded, of the owners' own voxels (no replication)
        ShardedVoxelGenerator([0, 1, 0, 1, 0, 1], [10, 10, 10], comm=LocalComm(), ops=NumpyOps(), max_points=4, replicate=True, resident=True)
    with pytest.raises(ValueError):
        ShardedVoxelGenerator([0, 1, 0, 1, 0, 1], [10, 10, 10], comm=LocalComm(), ops=NumpyOps(), replicate=False, resident=True)
    with pytest.raises(ValueError):
        box2d_iou(torch.zeros(3, 4), torch.zeros(3, 5))
    with pytest.raises(ValueError):
        box2d_nms(torch.zeros(3, 5), torch.zeros(2))
    assert box2d_nms(torch.zeros(0, 5), torch.zeros(0)).numel() == 0
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):          # no silent CPU fallback
            g(torch.zeros(4, 4))


def test_evaluator_and_matcher_host_logic_without_gpu():
    """score thresholds as benchmarks.pyx:125-134 derives them, class / overlap parsing, array validation"""
    import numpy as np
    import torch
    from d3d_amd.benchmarks import DetectionEvaluator
    from d3d_amd.tracking import DistanceTypes, prepare_boxes
    ev = DetectionEvaluator([3, 7], [0.7, 0.5])
    assert abs(ev._max_distance[3] - 0.3) < 1e-12 and ev._max_distance[7] == 0.5 and len(ev.score_thresholds) == 40
    t = ev.score_thresholds
    assert np.all(np.diff(t) > 0) and t[0] == 0 and t[-1] < 1 and abs(t[-1] - (1 - (10 ** (1 / 40) - 1) / 9)) < 1e-6
    lin = DetectionEvaluator(5, 0.5, pr_sample_count=4, min_score=0.2, pr_sample_scale="lin").score_thresholds
    assert np.allclose(lin, [0.2, 0.4, 0.6, 0.8])
    assert DetectionEvaluator([1], 0.5, pr_sample_scale="log100", pr_sample_count=8).score_thresholds.shape == (8,)
    with pytest.raises(ValueError):
        DetectionEvaluator([1], 0.5, pr_sample_scale="cubic")
    with pytest.raises(ValueError):
        DetectionEvaluator([1], "0.5")
    assert DistanceTypes.IoU == 1 and DistanceTypes.RIoU == 2 and DistanceTypes.Position == 3
    with pytest.raises(ValueError):
        prepare_boxes(np.zeros((3, 7), np.float32), np.zeros((3, 9), np.float32), DistanceTypes.RIoU)
    # no boxes on either side: the reference's early return (benchmarks.pyx / matcher.pyx:41-43), nothing touches a device
    r = ev.calc_stats(np.zeros((0, 9), np.float32), np.zeros((0, 9), np.float32))
    assert r.ngt == {3: 0, 7: 0} and r.tp[3] == [0] * 40 and np.all(np.isnan(r.acc_iou[7]))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="HIP device"):
            prepare_boxes(np.zeros((3, 9), np.float32), np.zeros((3, 9), np.float32), DistanceTypes.IoU)


def test_segment_helpers_are_plain_tensor_arithmetic():
    """seg1d_iou / seg1d_pdist (reference box/__init__.py:152-178, 317-331): no kernel behind them, they run on CPU tensors"""
    import torch
    from d3d_amd.box import seg1d_iou, seg1d_pdist
    a = torch.tensor([[0.0, 2.0], [0.0, 2.0], [5.0, 1.0], [0.0, 4.0]])
    b = torch.tensor([[1.0, 2.0], [3.0, 2.0], [5.0, 1.0], [0.0, 2.0]])
    assert torch.allclose(seg1d_iou(a, b, reference_compat=False), torch.tensor([1.0 / 3.0, 0.0, 1.0, 0.5]))
    assert torch.allclose(seg1d_iou(a, b), torch.tensor([1.0 / 3.0, 0.0, 1.0, 1.0]))      # the reference's (seg1's width twice, :164)
    d = seg1d_pdist(torch.tensor([[0.5], [3.0]]), torch.tensor([[0.0, 2.0]]))
    assert torch.allclose(d, torch.tensor([[0.5], [-2.0]]))


def test_bucket_kernel_keeps_its_first_point_store_separate(tmp_path):
    """with two plain conditional stores next to each other in k_bucket_index's record phase hipcc 7.2 once emitted
    a MERGED store that wrote the record position at firstmap[record position] -- the other store's index.  The firstmap
    store is an agent-scope atomic store since.  This pins the shape: in the gfx950 assembly every instantiation of the
    kernel carries one `global_store_dword ... sc1` per record slot of a lane (kBucketSlots / kBucketThreads = 4) in each of
    the two copies of the record phase (register buckets, big buckets) -- if a compiler bump folds them into something else,
    look at the record phase again before trusting the big-bucket tests alone."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "d3d_amd", "csrc", "voxel.hip")
    asm = str(tmp_path / "voxel.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           "-S", "--cuda-device-only", src, "-o", asm], stderr=subprocess.DEVNULL)
    scoped, cur = {}, None
    for line in open(asm):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            cur = m.group(1)
        elif cur and "k_bucket_index" in cur and re.search(r"global_store_dword\b.*\bsc1\b", line):
            scoped[cur] = scoped.get(cur, 0) + 1
    # round 5's instantiation (V2, the last template argument): its register path stores a voxel's entry from the point of rank 0
    # -- one store per item of a lane (4) + one in the crowded cells' wavefront loop -- next to the big-bucket copy (4)
    v2 = {k: v for k, v in scoped.items() if "Lb1EEE" in k.split("vT_")[0]}
    v1 = {k: v for k, v in scoped.items() if k not in v2}
    assert len(v1) == 6 and set(v1.values()) == {8}, scoped
    assert len(v2) == 2 and set(v2.values()) == {9}, scoped            # (DenseKey: the index for k_emit; BoundKey: for k_sparse_finish)


def test_sparse_call_block_layout_matches_the_header(tmp_path):
    """D3DSparseFilterCall: the ctypes Structure the Python layer fills (d3d_amd.voxel._SparseFilterCall) against the struct of
    include/d3d_hip.h as gcc lays it out -- size and the offset of every field; and the layout query: aligned, ordered pieces that
    hold their rows (a pure function: no GPU)"""
    import subprocess
    from d3d_amd import _lib
    from d3d_amd.voxel import _SparseFilterCall
    fields = [f[0] for f in _SparseFilterCall._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "d3d_hip.h"\nint main(void) {\n'
                   '    printf("%zu\\n", sizeof(D3DSparseFilterCall));\n' +
                   "".join('    printf("%%zu\\n", offsetof(D3DSparseFilterCall, %s));\n' % f for f in fields) + "    return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", str(src), "-I" + os.path.join(ROOT, "include"), "-o", str(exe)])
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert got[0] == ctypes.sizeof(_SparseFilterCall)
    assert got[1:] == [getattr(_SparseFilterCall, f).offset for f in fields]
    lib = _lib.load()
    off = (ctypes.c_size_t * 5)()
    for n, c in ((1, 3), (1000, 4), (1000003, 7)):
        total = lib.d3d_voxelize_3d_sparse_filter_call_layout(n, c, off)
        sizes = [n * c * 4, n * 8, n * 8, n * 4, n * 24]
        assert list(off)[0] == 0 and all(o % 256 == 0 for o in off)
        assert all(off[k] + sizes[k] <= (off[k + 1] if k < 4 else total) for k in range(5))
        assert lib.d3d_voxelize_3d_sparse_filter_call_workspace_bytes(n) > lib.d3d_voxelize_workspace_bytes(n, n)
