"""Build container only: the reference's OWN Python layers (d3d/box/__init__.py, d3d/voxel/__init__.py) are imported from
where they lie under /root/reference with `d3d_amd.box.box_impl` / `d3d_amd.voxel.voxel_impl` standing in for the compiled
modules they import (box/impl.cpp, voxel/impl.cpp).  Proves the drop-in boundary name by name and signature by signature;
nothing of the reference is copied or shipped (the GPU box has no /root/reference: skipped there)."""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "d3d")), reason="needs the reference checkout")


@pytest.fixture
def ref_pkg():
    saved = {k: v for k, v in sys.modules.items() if k == "d3d" or k.startswith("d3d.") or k == "addict"}
    for k in saved:
        del sys.modules[k]

    class _D(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    addict = types.ModuleType("addict")            # the one third-party import of voxel/__init__.py:1, not installed here
    addict.Dict = _D
    sys.modules["addict"] = addict
    pkg = types.ModuleType("d3d")                  # (the real d3d/__init__.py eagerly imports the Cython parts)
    pkg.__path__ = []
    sys.modules["d3d"] = pkg

    def load(sub, impl_name, impl):
        sys.modules["d3d.%s.%s" % (sub, impl_name)] = impl
        spec = importlib.util.spec_from_file_location("d3d." + sub, os.path.join(REF, "d3d", sub, "__init__.py"),
                                                      submodule_search_locations=[os.path.join(REF, "d3d", sub)])
        mod = importlib.util.module_from_spec(spec)
        sys.modules["d3d." + sub] = mod
        spec.loader.exec_module(mod)
        return mod
    yield load
    for k in [k for k in sys.modules if k == "d3d" or k.startswith("d3d.") or k == "addict"]:
        del sys.modules[k]
    sys.modules.update(saved)


def test_reference_box_layer_imports_against_box_impl(ref_pkg):
    from d3d_amd.box import box_impl
    mod = ref_pkg("box", "box_impl", box_impl)
    for name in ("Iou2D", "Iou2DR", "GIou2DR", "DIou2DR", "PDist2DR", "box2d_iou", "box2d_nms", "box2dr_crop", "box3dp_crop",
                 "box2dr_pdist", "box3dr_pdist"):
        assert hasattr(mod, name), name
    # every compiled name the reference binds resolves to this library's function
    assert mod.iou2dr_forward is box_impl.iou2dr_forward and mod.nms2d_cuda is box_impl.nms2d_cuda
    assert mod.giou2dr_backward_cuda is box_impl.giou2dr_backward and mod.pdist2dr_forward is box_impl.pdist2dr_forward
    assert mod.IouType.DRBOX == 6 and mod.SupressionType.GAUSSIAN == 2 and mod.cuda_available is True
    # argument validation of the reference layer runs before anything reaches the device
    with pytest.raises(ValueError):
        mod.box2d_iou(torch.zeros(3, 4), torch.zeros(3, 5))
    with pytest.raises(ValueError):
        mod.box2d_nms(torch.zeros(3, 5), torch.zeros(2))
    assert mod.box2d_nms(torch.zeros(0, 5), torch.zeros(0)).numel() == 0
    if not torch.cuda.is_available():              # ... and a real call lands in this library (no CPU fallback to fall into)
        for method in ("box", "rbox", "grbox", "drbox"):
            with pytest.raises(RuntimeError, match="HIP device"):
                mod.box2d_iou(np.zeros((2, 5)), np.zeros((3, 5)), method=method)
        with pytest.raises(RuntimeError, match="HIP device"):
            mod.box2d_nms(torch.zeros(4, 5), torch.zeros(4), iou_method="rbox")
        with pytest.raises(RuntimeError, match="HIP device"):
            mod.box2dr_pdist(torch.zeros(4, 2), torch.ones(3, 5))


def test_reference_voxel_layer_imports_against_voxel_impl(ref_pkg):
    from d3d_amd.voxel import voxel_impl
    mod = ref_pkg("voxel", "voxel_impl", voxel_impl)
    gen = mod.VoxelGenerator([0, 70.4, -40, 40, -3, 1], [704, 800, 40], max_points=32, reduction="mean", dense=True)
    assert gen._reduction == voxel_impl.ReductionType.MEAN and gen._vbounds.tolist() == [[0, 704], [-400, 400], [-30, 10]]
    with pytest.raises(ValueError):
        mod.VoxelGenerator([0.05, 1, 0, 1, 0, 1], [10, 10, 10])
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="HIP device"):
            gen(torch.zeros(4, 4))
        with pytest.raises(RuntimeError, match="HIP device"):
            mod.VoxelGenerator([0, 1, 0, 1, 0, 1], [10, 10, 10], max_points=5, max_points_filter="trim")(torch.zeros(4, 4))


def test_intentional_deviations_are_pinned(ref_pkg):
    """where this library deliberately differs from the reference's Python layer, pinned against the
    reference's own code: (1) seg1d_iou -- the reference halves seg1's width for BOTH segments (box/__init__.py:163-164);
    the default reproduces its values bit for bit, `reference_compat=False` uses seg2's own width and agrees wherever the widths
    agree; (2) PDist2DR -- the reference hands (boxes, points) to compiled functions declared (points, boxes) and returns the
    backward's (grad_boxes, grad_points) for the inputs (points, boxes); this library's PDist2DR passes (points, boxes) and
    returns the gradients in input order."""
    from d3d_amd import box as ours
    from d3d_amd.box import box_impl
    mod = ref_pkg("box", "box_impl", box_impl)
    g = torch.Generator().manual_seed(5)
    a = torch.rand(200, 2, generator=g) + 0.1
    b = torch.rand(200, 2, generator=g) + 0.1
    ref = mod.seg1d_iou(a, b)
    assert torch.equal(ours.seg1d_iou(a, b), ref) and torch.equal(ours.seg1d_iou(a, b, reference_compat=True), ref)
    fixed = ours.seg1d_iou(a, b, reference_compat=False)
    assert not torch.allclose(fixed, ref)                      # the widths differ: so do the results
    lo = torch.maximum(a[:, 0] - a[:, 1] / 2, b[:, 0] - b[:, 1] / 2)
    hi = torch.minimum(a[:, 0] + a[:, 1] / 2, b[:, 0] + b[:, 1] / 2)
    inter = torch.clamp_min(hi - lo, 0)
    assert torch.allclose(fixed, inter / (a[:, 1] + b[:, 1] - inter), atol=1e-6)
    b2 = torch.stack([b[:, 0], a[:, 1]], 1)                    # equal widths: no deviation
    assert torch.equal(ours.seg1d_iou(a, b2), mod.seg1d_iou(a, b2))

    calls = []

    def fwd(x, y):
        calls.append(("fwd", tuple(x.shape), tuple(y.shape)))
        return torch.zeros(y.shape[0] if x.shape[1] == 2 else x.shape[0], 1), torch.zeros(1, dtype=torch.uint8)

    def bwd(x, y, grad, *rest):
        calls.append(("bwd", tuple(x.shape), tuple(y.shape)))
        return torch.zeros_like(x), torch.zeros_like(y)
    pts, boxes = torch.zeros(7, 2, requires_grad=True), torch.zeros(3, 5, requires_grad=True)
    mod.pdist2dr_forward, mod.pdist2dr_backward = fwd, bwd      # the reference's PDist2DR: (boxes, points) reach the functions
    with pytest.raises(RuntimeError, match="invalid gradient"):    # ... and their gradients come back in THAT order, for the
        mod.PDist2DR.apply(pts, boxes).sum().backward()            # inputs (points, boxes): autograd rejects it when N != M
    assert calls == [("fwd", (3, 5), (7, 2)), ("bwd", (3, 5), (7, 2))]
    calls.clear()
    saved = ours.pdist2dr_forward, ours.pdist2dr_backward
    try:
        ours.pdist2dr_forward = lambda p, b: (calls.append(("fwd", tuple(p.shape), tuple(b.shape))) or torch.zeros(3, 7), None)
        ours.pdist2dr_backward = lambda p, b, g: (calls.append(("bwd", tuple(p.shape), tuple(b.shape))) or torch.zeros_like(b),
                                                  torch.zeros_like(p))
        p2, b2_ = torch.zeros(7, 2, requires_grad=True), torch.zeros(3, 5, requires_grad=True)
        ours.PDist2DR.apply(p2, b2_).sum().backward()
    finally:
        ours.pdist2dr_forward, ours.pdist2dr_backward = saved
    assert calls == [("fwd", (7, 2), (3, 5)), ("bwd", (7, 2), (3, 5))]   # ours: (points, boxes), as dist.h:7-13 declares
    assert p2.grad.shape == (7, 2) and b2_.grad.shape == (3, 5)
