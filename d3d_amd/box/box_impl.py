"""d3d_amd.box.box_impl -- stands in for the reference's compiled module `d3d.box.box_impl` (box/impl.cpp:8-54).

Exports EXACTLY the names reference d3d/box/__init__.py:5-36 imports, with the compiled functions' signatures and return
shapes (iou.h, nms.h, dist.h, utils.h): dropping this file in as d3d/box/box_impl.py lets the reference's own Python layer
run unchanged on the HIP kernels.  The `_cuda` twins are the same functions (tensors live where they live: HIP tensors are
`is_cuda` on PyTorch-ROCm, CPU tensors are staged through the GPU and come back on the CPU).

The flag tensors the reference's autograd saves between forward and backward (nx, xflags, ...) are produced at ANY size, in the
declared shapes (iou.cpp:125-141: nx[N,M], xflags[N,M,8], ...; 9-18 bytes per pair -- torch raises its out-of-memory error where
they do not fit, as the reference's allocation would).  This library's backward recomputes the geometry analytically and never
reads them: a caller that owns both ends of the call and wants to skip the Sutherland-Hodgman pass over every pair sets
`box_impl.compute_flags = False` and gets EMPTY tensors in their place -- an explicit opt-out, never a size-dependent default.
"""
import torch

from . import (IouType, SupressionType, crop_2dr, cuda_available, diou2dr_backward as _diou_bwd, diou2dr_forward as _diou_fwd,
               giou2dr_backward as _giou_bwd, giou2dr_forward as _giou_fwd, iou2d_backward, iou2d_forward,
               iou2dr_backward as _riou_bwd, iou2dr_flags, iou2dr_forward as _riou_fwd, nms2d as _nms2d,
               pdist2dr_backward as _pdist_bwd, pdist2dr_forward as _pdist_fwd)

compute_flags = True


def _flags(boxes1, boxes2, which):
    if not compute_flags:
        return {k: torch.empty((0,), dtype=torch.uint8, device=boxes1.device) for k in which}
    return iou2dr_flags(boxes1, boxes2, which)


def iou2dr_forward(boxes1, boxes2):
    """iou.h:25-31 -> (ious[N,M], nx[N,M] u8, xflags[N,M,8] u8)"""
    f = _flags(boxes1, boxes2, ("nx", "xflags"))
    return _riou_fwd(boxes1, boxes2), f["nx"], f["xflags"]


def iou2dr_backward(boxes1, boxes2, grad, nx=None, xflags=None):
    """iou.h:32-40 -> (grad_boxes1[N,5], grad_boxes2[M,5])"""
    return _riou_bwd(boxes1, boxes2, grad)


def giou2dr_forward(boxes1, boxes2):
    """iou.h:41-47 -> (ious[N,M], nxm[N,M,2] u8 = (nx, nm), xmflags[N,M,16] u8 = (xflags, mflags))"""
    f = _flags(boxes1, boxes2, ("nx", "xflags", "nm", "mflags"))
    if f["nx"].numel() == 0 and boxes1.shape[0] * boxes2.shape[0] > 0:
        return _giou_fwd(boxes1, boxes2), f["nx"], f["xflags"]
    return (_giou_fwd(boxes1, boxes2), torch.stack([f["nx"], f["nm"]], dim=-1), torch.cat([f["xflags"], f["mflags"]], dim=-1))


def giou2dr_backward(boxes1, boxes2, grad, nxm=None, xmflags=None):
    return _giou_bwd(boxes1, boxes2, grad)


def diou2dr_forward(boxes1, boxes2):
    """iou.h:56-62 -> (ious[N,M], nxd[N,M,3] u8 = (nx, far0, far1), xflags[N,M,8] u8)"""
    f = _flags(boxes1, boxes2, ("nx", "xflags", "far"))
    if f["nx"].numel() == 0 and boxes1.shape[0] * boxes2.shape[0] > 0:
        return _diou_fwd(boxes1, boxes2), f["nx"], f["xflags"]
    return _diou_fwd(boxes1, boxes2), torch.cat([f["nx"].unsqueeze(-1), f["far"]], dim=-1), f["xflags"]


def diou2dr_backward(boxes1, boxes2, grad, nxd=None, xflags=None):
    return _diou_bwd(boxes1, boxes2, grad)


def _points_boxes(a, b):
    # dist.h:7-13 declares (points, boxes); the reference's Python layer passes (boxes, points) (box/__init__.py:137-139,
    # SURVEY App. E) -- told apart by their shapes so that either caller works
    if a.dim() == 2 and b.dim() == 2 and a.shape[1] == 5 and b.shape[1] == 2:
        return b, a
    return a, b


def pdist2dr_forward(points, boxes):
    """dist.h:7-9 -> (distance[M,N], iedge[M,N] u8)"""
    points, boxes = _points_boxes(points, boxes)
    return _pdist_fwd(points, boxes)


def pdist2dr_backward(points, boxes, grad, iedge=None):
    """dist.h:10-13 -> (grad_boxes[M,5], grad_points[N,2])"""
    points, boxes = _points_boxes(points, boxes)
    return _pdist_bwd(points, boxes, grad)


def nms2d(boxes, scores, iou_type, supression_type, iou_threshold, score_threshold, supression_param):
    """nms.h:6-11 -> suppressed bool[N]"""
    return _nms2d(boxes, scores, iou_type, supression_type, iou_threshold, score_threshold, supression_param)


iou2d_forward_cuda = iou2d_forward
iou2d_backward_cuda = iou2d_backward
iou2dr_forward_cuda = iou2dr_forward
iou2dr_backward_cuda = iou2dr_backward
giou2dr_forward_cuda = giou2dr_forward
giou2dr_backward_cuda = giou2dr_backward
diou2dr_forward_cuda = diou2dr_forward
diou2dr_backward_cuda = diou2dr_backward
pdist2dr_forward_cuda = pdist2dr_forward
pdist2dr_backward_cuda = pdist2dr_backward
nms2d_cuda = nms2d

__all__ = ["cuda_available", "iou2d_forward", "iou2d_backward", "iou2dr_forward", "iou2dr_backward", "giou2dr_forward",
           "giou2dr_backward", "diou2dr_forward", "diou2dr_backward", "pdist2dr_forward", "pdist2dr_backward", "nms2d", "crop_2dr",
           "IouType", "SupressionType", "iou2d_forward_cuda", "iou2d_backward_cuda", "iou2dr_forward_cuda", "iou2dr_backward_cuda",
           "giou2dr_forward_cuda", "giou2dr_backward_cuda", "diou2dr_forward_cuda", "diou2dr_backward_cuda",
           "pdist2dr_forward_cuda", "pdist2dr_backward_cuda", "nms2d_cuda"]
