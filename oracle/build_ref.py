"""Build the *real* reference voxelizer into oracle/_ref/ (TEST INFRASTRUCTURE ONLY).

Compiles /root/reference/d3d/voxel/{impl,voxelize}.cpp *where they lie* (nothing is
copied into this repo) with torch.utils.cpp_extension.  Output: oracle/_ref/voxel_impl.so
(git-ignored; travels to the GPU box with gpurun).  The box/iou/nms sources cannot be
built: they include the un-vendored third-party header dgal/geometry.hpp
(reference d3d/box/utils.h:5) -> treated as unbuildable, see DESIGN.md.

Usage: python oracle/build_ref.py   (no-op when /root/reference is absent or the .so exists)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("D3D_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")


def ref_so_path():
    return os.path.join(OUT, "voxel_impl.so")


def build(verbose=False):
    so = ref_so_path()
    if os.path.exists(so):
        return so
    srcs = [os.path.join(REF, "d3d/voxel/impl.cpp"), os.path.join(REF, "d3d/voxel/voxelize.cpp")]
    if not all(os.path.exists(s) for s in srcs):
        return None
    os.makedirs(OUT, exist_ok=True)
    os.environ.setdefault("CXX", "g++")
    os.environ.setdefault("MAX_JOBS", "2")
    from torch.utils.cpp_extension import load
    load(name="voxel_impl", sources=srcs, extra_include_paths=[REF],
         extra_cflags=["-O2", "-Wno-deprecated-declarations"],
         build_directory=OUT, verbose=verbose)
    return so if os.path.exists(so) else None


def load_ref():
    """Import the built reference module (or None if it is not available)."""
    so = ref_so_path()
    if not os.path.exists(so):
        return None
    import importlib.util
    import torch  # noqa: F401  (the .so links against libtorch)
    spec = importlib.util.spec_from_file_location("voxel_impl", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    p = build(verbose=True)
    print("reference voxel_impl:", p)
    sys.exit(0 if p else 1)
