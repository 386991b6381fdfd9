"""d3d_amd.point -- drop-in for d3d.point.aligned_scatter (reference d3d/point/__init__.py), a "next" row of the
hot-path scope: gather features of a dense map at fractional coordinates (drop / mean / linear) with autograd."""
import ctypes
import enum

import torch

from .. import _lib


class AlignType(enum.IntEnum):      # point/scatter.h:37
    DROP = 0
    MEAN = 1
    LINEAR = 2
    MAX = 3
    NEAREST = 4


cuda_available = True


def _call(fn_name, coord, a, atype, out, image_shape):
    lib = _lib.load()
    dim = coord.shape[1] - 1
    dims_h = (ctypes.c_int64 * max(dim, 1))(*[int(x) for x in image_shape[2:]])
    code = _lib.F64 if a.dtype == torch.float64 else _lib.F32
    dims_p = ctypes.cast(dims_h, ctypes.c_void_p)
    batch, channels = int(image_shape[0]), int(image_shape[1])
    ws = _lib.workspace(lib.d3d_aligned_scatter_workspace_bytes(batch, channels, dims_p, dim, code), coord.device)
    rc = getattr(lib, fn_name)(_lib.ptr(coord), coord.shape[0], dim, _lib.ptr(a), batch, channels, dims_p, int(atype), code,
                               _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
    if rc == _lib.ERR_UNSUPPORTED:
        raise ValueError("Unsupported align type!" if dim in (1, 2, 3) else "Unsupported dimension size: %d" % dim)
    _lib.check(rc, fn_name)


def _check(coord, feat):
    if coord.dim() != 2 or feat.dim() != coord.shape[1] + 1:
        raise RuntimeError("coordinates must be N x (m+1) and the feature map B x C x D1 .. x Dm")
    if feat.dtype not in (torch.float32, torch.float64) or coord.dtype != feat.dtype:
        raise RuntimeError("coordinates and features must share a floating dtype")   # accessor<scalar_t> in the reference


def aligned_scatter_forward(coord, image_feature, atype):
    """aligned_scatter_forward[_cuda] (scatter.h:39-41): [N, C] features"""
    _check(coord, image_feature)
    odev = image_feature.device
    dev = odev if image_feature.is_cuda else _lib.require_gpu()
    c, f = coord.to(dev).contiguous(), image_feature.detach().to(dev).contiguous()
    with torch.cuda.device(dev):
        out = torch.empty((c.shape[0], f.shape[1]), dtype=f.dtype, device=dev)
        _call("d3d_aligned_scatter_forward", c, f, atype, out, f.shape)
    return out.to(odev) if odev != dev else out


def aligned_scatter_backward(coord, grad, atype, image_grad):
    """aligned_scatter_backward[_cuda] (scatter.h:42-45): accumulates into image_grad (in place)"""
    odev = image_grad.device
    dev = odev if image_grad.is_cuda else _lib.require_gpu()
    c, g = coord.to(dev).contiguous(), grad.to(dev).contiguous()
    target = image_grad if (odev == dev and image_grad.is_contiguous()) else image_grad.to(dev).contiguous()
    with torch.cuda.device(dev):
        _call("d3d_aligned_scatter_backward", c, g, atype, target, image_grad.shape)
    if target is not image_grad:
        image_grad.copy_(target)


aligned_scatter_forward_cuda, aligned_scatter_backward_cuda = aligned_scatter_forward, aligned_scatter_backward


class AlignedScatter(torch.autograd.Function):      # point/__init__.py:13-40
    @staticmethod
    def forward(ctx, image_feature, coords, atype):
        ctx.save_for_backward(coords)
        ctx.atype = atype
        ctx.image_shape, ctx.image_dtype, ctx.image_device = image_feature.shape, image_feature.dtype, image_feature.device
        return aligned_scatter_forward(coords, image_feature, atype)

    @staticmethod
    def backward(ctx, grad):
        coords, = ctx.saved_tensors
        image_grad = torch.zeros(ctx.image_shape, dtype=ctx.image_dtype, device=ctx.image_device)
        aligned_scatter_backward(coords, grad, ctx.atype, image_grad)
        return image_grad, None, None


def aligned_scatter(coordinates, feature_map, method="drop"):
    """Gather the values of `feature_map` [B, C, D1..Dm] at `coordinates` [N, m+1] (batch index first) --
    reference point/__init__.py:43-67.  method: drop | mean | linear."""
    method = (method or "DROP").upper()
    if method == "DROP":
        coordinates = coordinates.long()
        _, ndim = coordinates.shape
        assert len(feature_map.shape) == ndim + 1
        indexing = (coordinates[:, 0], slice(None)) + tuple(coordinates[:, i] for i in range(1, ndim))
        return feature_map[indexing]
    align_type = getattr(AlignType, method)
    return AlignedScatter.apply(feature_map, coordinates, align_type)


__all__ = ["aligned_scatter", "AlignedScatter", "AlignType", "aligned_scatter_forward", "aligned_scatter_backward",
           "cuda_available"]
