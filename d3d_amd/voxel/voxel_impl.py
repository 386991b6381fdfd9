"""d3d_amd.voxel.voxel_impl -- stands in for the reference's compiled module `d3d.voxel.voxel_impl` (voxel/impl.cpp:3-21):
the three functions and three enums reference d3d/voxel/__init__.py:6-9 imports, with the compiled signatures
(voxelize.h:9-25).  Dropped in as d3d/voxel/voxel_impl.py it runs the reference's own VoxelGenerator on the HIP kernels."""
from . import (MaxPointsFilterType, MaxVoxelsFilterType, ReductionType, voxelize_3d_dense, voxelize_3d_filter,
               voxelize_3d_sparse)

__all__ = ["voxelize_3d_dense", "voxelize_3d_sparse", "voxelize_3d_filter", "ReductionType", "MaxPointsFilterType",
           "MaxVoxelsFilterType"]
