"""d3d_amd -- MI355X-native implementation of the d3d.voxel / d3d.box hot path.

    from d3d_amd.voxel import VoxelGenerator
    from d3d_amd.box import box2d_iou, box2d_nms, iou2d, iou3d, nms
"""
__version__ = "0.1.0"
