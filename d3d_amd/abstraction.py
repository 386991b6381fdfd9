"""d3d_amd.abstraction -- the point-in-box batch operators of the reference's d3d.abstraction containers on MI355X, at the
ARRAY level (the containers themselves -- ObjectTarget3D, Target3DArray -- are out of scope):

  crop_points(boxes, cloud)              Target3DArray.crop_points   (abstraction.pyx:684-687; one box: :321-324)
  paint_label(boxes, cloud, semantics)   Target3DArray.paint_label   (abstraction.pyx:662-682)

`boxes` is either [M,7] rows (x, y, z, lx, ly, lz, rz) or the [n,9] rows of Target3DArray.to_numpy (label, score, x, y, z,
lx, ly, lz, yaw; abstraction.pyx:263-272), read in place.  Per pair the test is box3dr_contains (dgal_wrap.h:6-19): closed z
interval, fp32.  numpy in -> numpy out; torch in -> torch out on the same device.  There is no CPU path: CPU inputs are
staged through the current HIP device.
"""
import numpy as np
import torch

from . import _lib


def _ingress(boxes, cloud):
    convert = isinstance(cloud, np.ndarray)
    bx = torch.from_numpy(boxes) if isinstance(boxes, np.ndarray) else boxes
    pts = torch.from_numpy(cloud) if isinstance(cloud, np.ndarray) else cloud
    if bx.dim() != 2 or bx.shape[1] not in (7, 9):
        raise ValueError("boxes should be [M,7] (x,y,z,lx,ly,lz,rz) or [M,9] (label,score,x,y,z,lx,ly,lz,yaw)")
    if pts.dim() != 2 or pts.shape[1] < 3:
        raise ValueError("cloud should be [N,>=3] (x, y, z first)")
    dev = pts.device if pts.is_cuda else (bx.device if bx.is_cuda else _lib.require_gpu())
    odev = pts.device
    bx = bx.to(dev, torch.float32).contiguous()       # the reference's memoryviews are float32 (abstraction.pyx:310)
    pts = pts.to(dev, torch.float32).contiguous()
    return bx, pts, dev, odev, convert


def crop_points(boxes, cloud):
    """bool[M,N]: [i, j] = box i contains point j (Target3DArray.crop_points; a single box: pass one row)."""
    lib = _lib.load()
    bx, pts, dev, odev, convert = _ingress(boxes, cloud)
    m, n = bx.shape[0], pts.shape[0]
    with torch.cuda.device(dev):
        out = torch.empty((m, n), dtype=torch.uint8, device=dev)
        rc = lib.d3d_crop_3dr(_lib.ptr(pts), n, pts.shape[1], _lib.ptr(bx), m, bx.shape[1], 0 if bx.shape[1] == 7 else 2,
                              _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "crop_3dr")
    out = out.view(torch.bool)
    if odev != dev:
        out = out.to(odev)
    return out.numpy() if convert else out


def paint_label(boxes, cloud, semantics, labels=None):
    """uint16[N]: 1 + index of the first box (the best score of a descendingly sorted array) that contains the point and
    whose class equals the point's semantic label, 0 where there is none (Target3DArray.paint_label).  `labels`: class per
    box; defaults to column 0 of [n,9] rows (tag.labels[0], abstraction.pyx:667)."""
    lib = _lib.load()
    bx, pts, dev, odev, convert = _ingress(boxes, cloud)
    def as_u8(t, what):
        # class ids travel as uint8 (abstraction.pyx:662-682 compares uint8 label arrays): a value outside [0, 255] would
        # wrap silently in the cast -- refuse it
        t = torch.from_numpy(np.ascontiguousarray(t)) if isinstance(t, np.ndarray) else t
        if t.numel() and t.dtype != torch.uint8:
            lo, hi = float(t.min()), float(t.max())
            if lo < 0 or hi > 255 or (t.is_floating_point() and bool((t != t.round()).any())):
                raise ValueError("%s must be integers in [0, 255]" % what)
        return t.to(dev, torch.uint8).contiguous()
    if labels is None:
        if bx.shape[1] != 9:
            raise ValueError("labels are needed with [M,7] boxes")
        lab = as_u8(bx[:, 0], "labels")
    else:
        lab = as_u8(labels, "labels")
    sem = as_u8(semantics, "semantics")
    m, n = bx.shape[0], pts.shape[0]
    if sem.numel() != n or lab.numel() != m:
        raise ValueError("semantics needs one entry per point, labels one per box")
    with torch.cuda.device(dev):
        idarr = torch.empty((n,), dtype=torch.int16, device=dev)      # uint16 bits (torch has no uint16 arithmetic type)
        rc = lib.d3d_paint_label(_lib.ptr(pts), n, pts.shape[1], _lib.ptr(sem), _lib.ptr(bx), m, bx.shape[1],
                                 0 if bx.shape[1] == 7 else 2, _lib.ptr(lab), _lib.ptr(idarr), _lib.stream_ptr())
    _lib.check(rc, "paint_label")
    if convert or odev != dev:
        out = idarr.cpu().numpy().view(np.uint16)
        return out if convert else torch.from_numpy(out.astype(np.int32))
    return idarr.to(torch.int32) & 0xffff


__all__ = ["crop_points", "paint_label"]
