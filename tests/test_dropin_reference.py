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
