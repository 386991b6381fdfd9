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


MODULES = {   # name -> reference sources (compiled where they lie)
    "voxel_impl": ["d3d/voxel/impl.cpp", "d3d/voxel/voxelize.cpp"],
    "point_impl": ["d3d/point/impl.cpp", "d3d/point/scatter.cpp"],     # "next" row: d3d.point.aligned_scatter
}


def ref_so_path(name="voxel_impl"):
    return os.path.join(OUT, name, name + ".so") if name != "voxel_impl" else os.path.join(OUT, "voxel_impl.so")


def build(verbose=False, name="voxel_impl"):
    so = ref_so_path(name)
    if os.path.exists(so):
        return so
    srcs = [os.path.join(REF, s) for s in MODULES[name]]
    if not all(os.path.exists(s) for s in srcs):
        return None
    outdir = os.path.dirname(so)
    os.makedirs(outdir, exist_ok=True)
    os.environ.setdefault("CXX", "g++")
    os.environ.setdefault("MAX_JOBS", "2")
    from torch.utils.cpp_extension import load
    load(name=name, sources=srcs, extra_include_paths=[REF],
         extra_cflags=["-O2", "-Wno-deprecated-declarations"],
         build_directory=outdir, verbose=verbose)
    return so if os.path.exists(so) else None


def load_ref(name="voxel_impl"):
    """Import a built reference module (or None if it is not available)."""
    so = ref_so_path(name)
    if not os.path.exists(so):
        return None
    import importlib.util
    import torch  # noqa: F401  (the .so links against libtorch)
    spec = importlib.util.spec_from_file_location(name, so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    ok = True
    for name in MODULES:
        p = build(verbose=True, name=name)
        print("reference %s:" % name, p)
        ok = ok and bool(p)
    sys.exit(0 if ok else 1)
