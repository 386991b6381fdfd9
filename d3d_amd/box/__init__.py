"""d3d_amd.box -- drop-in for d3d.box on MI355X.

Mirrors reference d3d/box/__init__.py: `box2d_iou` (:180, methods box / rbox / grbox / drbox), `box2d_nms` (:226),
`box2dr_crop` / `box3dp_crop` (:278-315), `box2dr_pdist` / `box3dr_pdist` (:333-381), the autograd functions `Iou2D`,
`Iou2DR`, `GIou2DR`, `DIou2DR`, `PDist2DR` (:41-150) and the enums `IouType`, `SupressionType` (box/common.h:5-10).
The compiled module's own names (box/impl.cpp:8-54) live in `d3d_amd.box.box_impl`.  north_star's names `iou2d`, `iou3d`,
`nms` are thin aliases (the reference reaches "iou3d" only through Cython: d3d/dgal_wrap.h:45-91,
d3d/tracking/matcher.pyx:57-80).  All compute runs in HIP kernels behind include/d3d_hip.h.
"""
import ctypes
import enum

import numpy as np
import torch

from .. import _lib, options


class IouType(enum.IntEnum):            # box/common.h:5-9
    NA = 0
    BOX = 1
    RBOX = 2
    GBOX = 3
    GRBOX = 4
    DBOX = 5
    DRBOX = 6


class SupressionType(enum.IntEnum):     # box/common.h:10  (sic: the reference spells it this way)
    HARD = 0
    LINEAR = 1
    GAUSSIAN = 2


cuda_available = True   # box/impl.cpp:9-13: this build always has its device path

# (per-call options -- flags=, poison= -- or the calling context's: d3d_amd.options)


def _dtype_code(t):
    if t.dtype == torch.float64:
        return _lib.F64
    if t.dtype == torch.float32:
        return _lib.F32
    raise RuntimeError("boxes must be float32 or float64")   # AT_DISPATCH_FLOATING_TYPES (iou.cpp:132)


def _to_device(*ts):
    dev = None
    for t in ts:
        if t.is_cuda:
            dev = t.device
            break
    if dev is None:
        dev = _lib.require_gpu()
    return [t.to(dev).contiguous() for t in ts], dev


_MAX_ROWS = 65535 * 64      # rows of boxes1 per d3d_iou2d_forward launch (grid.y limit x 64-row tiles)


def _iou_forward(boxes1, boxes2, iou_type, flags=None, matrix32=False, wide32=False):
    """matrix32 (fp64 boxes, BOX / RBOX): the arithmetic in fp64, `ious` stored as fp32 (D3D_F64_M32);
    wide32 (fp32 boxes, BOX / RBOX): the same with the boxes widened where the kernels load them (D3D_F32_WIDE)"""
    lib = _lib.load()
    odev = boxes1.device
    if boxes1.dtype != boxes2.dtype:
        raise RuntimeError("boxes1 and boxes2 must have the same dtype")
    (b1, b2), dev = _to_device(boxes1, boxes2)
    n, m = b1.shape[0], b2.shape[0]
    fl = options.current().iou_flags if flags is None else int(flags)
    with torch.cuda.device(dev):
        ious = torch.empty((n, m), dtype=torch.float32 if matrix32 else b1.dtype, device=dev)
        if options.current().poison:
            ious.fill_(float("nan"))
        code = _dtype_code(b1)
        if matrix32 or wide32:
            if code != (_lib.F32 if wide32 else _lib.F64) or int(iou_type) not in (IouType.BOX, IouType.RBOX):
                raise ValueError("matrix32 takes fp64 boxes, wide32 fp32 boxes, both the box / rbox methods")
            code = _lib.F32_WIDE if wide32 else _lib.F64_M32
        # one launch covers 65535 tiles of 64 rows; taller inputs go in row blocks into the same output
        for r0 in range(0, max(n, 1), _MAX_ROWS):
            r1 = min(n, r0 + _MAX_ROWS)
            ws = _lib.workspace(lib.d3d_iou2d_workspace_bytes(r1 - r0, m, code), dev)
            rc = lib.d3d_iou2d_forward(_lib.ptr(b1[r0:r1]), r1 - r0, _lib.ptr(b2), m, int(iou_type), code, _lib.ptr(ious[r0:r1]),
                                       _lib.ptr(ws), ws.numel() if ws is not None else 0, _lib.stream_ptr(), fl)
            _lib.check(rc, "iou2d_forward")
    return _lib.to_caller(ious, odev, dev)


def iou2d_forward(boxes1, boxes2):
    """iou2d_forward / iou2d_forward_cuda (iou.h:7-13; iou.cpp:35-46): AABB-of-rotated-box IoU [N,M]."""
    return _iou_forward(boxes1, boxes2, IouType.BOX)


def iou2dr_forward(boxes1, boxes2):
    """`ious` of iou2dr_forward / iou2dr_forward_cuda (iou.h:25-31; iou.cpp:125-141).  The reference's 3-tuple form
    (ious, nx, xflags) is `d3d_amd.box.box_impl.iou2dr_forward`; the flags alone: `iou2dr_flags`."""
    return _iou_forward(boxes1, boxes2, IouType.RBOX)


def giou2dr_forward(boxes1, boxes2):
    """`ious` of giou2dr_forward[_cuda] (iou.h:41-47; iou.cpp:243-258): GIoU = IoU - (hull - union) / hull"""
    return _iou_forward(boxes1, boxes2, IouType.GRBOX)


def diou2dr_forward(boxes1, boxes2):
    """`ious` of diou2dr_forward[_cuda] (iou.h:56-62; iou.cpp:352-367): DIoU = IoU - centre distance^2 / hull diameter^2"""
    return _iou_forward(boxes1, boxes2, IouType.DRBOX)


def iou2dr_flags(boxes1, boxes2, which=("nx", "xflags")):
    """the autograd bookkeeping of the rotated IoU family (include/d3d_hip.h: d3d_iou2dr_flags) -> dict of uint8 tensors:
    nx[N,M], xflags[N,M,8] (iou.cpp:125-141), nm[N,M], mflags[N,M,8] (giou, iou.cpp:243-258), far[N,M,2] (diou, :352-367)"""
    lib = _lib.load()
    odev = boxes1.device
    if boxes1.dtype != boxes2.dtype:
        raise RuntimeError("boxes1 and boxes2 must have the same dtype")
    (b1, b2), dev = _to_device(boxes1.detach(), boxes2.detach())
    n, m = b1.shape[0], b2.shape[0]
    shapes = dict(nx=(n, m), xflags=(n, m, 8), nm=(n, m), mflags=(n, m, 8), far=(n, m, 2))
    with torch.cuda.device(dev):
        out = {k: torch.empty(shapes[k], dtype=torch.uint8, device=dev) for k in which}
        rc = lib.d3d_iou2dr_flags(_lib.ptr(b1), n, _lib.ptr(b2), m, _dtype_code(b1), _lib.ptr(out.get("nx")),
                                  _lib.ptr(out.get("xflags")), _lib.ptr(out.get("nm")), _lib.ptr(out.get("mflags")),
                                  _lib.ptr(out.get("far")), _lib.stream_ptr())
    _lib.check(rc, "iou2dr_flags")
    return {k: (v.to(odev) if odev != dev else v) for k, v in out.items()}


def _iou_backward(boxes1, boxes2, grad, iou_type, matrix32=False, wide32=False):
    """matrix32 (fp64 boxes, BOX / RBOX): `grad` is read as fp32 and widened in the kernels (D3D_F64_M32);
    wide32 (fp32 boxes, BOX / RBOX): the boxes too (D3D_F32_WIDE) -- the sums are kept in fp64 and come back rounded to fp32"""
    lib = _lib.load()
    odev = boxes1.device
    mixed = matrix32 or wide32
    (b1, b2, g), dev = _to_device(boxes1.detach(), boxes2.detach(), grad.to(torch.float32 if mixed else boxes1.dtype))
    n, m = b1.shape[0], b2.shape[0]
    code = _dtype_code(b1)
    if mixed:
        if code != (_lib.F32 if wide32 else _lib.F64) or int(iou_type) not in (IouType.BOX, IouType.RBOX):
            raise ValueError("matrix32 takes fp64 boxes, wide32 fp32 boxes, both the box / rbox methods")
        code = _lib.F32_WIDE if wide32 else _lib.F64_M32
    with torch.cuda.device(dev):
        # both results in ONE buffer: the library then clears them with one launch, and the wide form rounds them with one
        gg = torch.empty((n + m, 5), dtype=torch.float64 if wide32 else b1.dtype, device=dev)
        g1, g2 = gg[:n], gg[n:]
        ws = _lib.workspace(lib.d3d_iou2d_workspace_bytes(n, m, code), dev)
        rc = lib.d3d_iou2d_backward(_lib.ptr(b1), n, _lib.ptr(b2), m, _lib.ptr(g), int(iou_type), code, _lib.ptr(g1),
                                    _lib.ptr(g2), _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "iou2d_backward")
        if wide32:
            gg = gg.to(torch.float32)
            g1, g2 = gg[:n], gg[n:]
    return (g1.to(odev), g2.to(odev)) if odev != dev else (g1, g2)


def iou2d_backward(boxes1, boxes2, grad):
    """iou2d_backward[_cuda] (iou.h:14-24; iou.cpp:75-93): (grad_boxes1[N,5], grad_boxes2[M,5])"""
    return _iou_backward(boxes1, boxes2, grad, IouType.BOX)


def iou2dr_backward(boxes1, boxes2, grad, nx=None, xflags=None):
    """iou2dr_backward[_cuda] (iou.h:32-40; iou.cpp:191-211).  `nx` / `xflags` (the reference's saved clip flags)
    are accepted for signature compatibility and ignored: the clip is recomputed."""
    return _iou_backward(boxes1, boxes2, grad, IouType.RBOX)


def giou2dr_backward(boxes1, boxes2, grad, nxm=None, xmflags=None):
    """giou2dr_backward[_cuda] (iou.h:48-55; iou.cpp:303-320); the saved flags are ignored (analytic gradients)"""
    return _iou_backward(boxes1, boxes2, grad, IouType.GRBOX)


def diou2dr_backward(boxes1, boxes2, grad, nxd=None, xflags=None):
    """diou2dr_backward[_cuda] (iou.h:63-69; iou.cpp:402-419); the saved flags are ignored (analytic gradients)"""
    return _iou_backward(boxes1, boxes2, grad, IouType.DRBOX)


class Iou2D(torch.autograd.Function):
    """Differentiable axis aligned IoU function for 2D boxes -- reference box/__init__.py:41-61"""

    @staticmethod
    def forward(ctx, boxes1, boxes2):
        ctx.save_for_backward(boxes1, boxes2)
        return iou2d_forward(boxes1, boxes2)

    @staticmethod
    def backward(ctx, grad):
        boxes1, boxes2 = ctx.saved_tensors
        return iou2d_backward(boxes1, boxes2, grad.contiguous())


class Iou2DR(torch.autograd.Function):
    """Differentiable rotated IoU function for 2D boxes -- reference box/__init__.py:63-84"""

    @staticmethod
    def forward(ctx, boxes1, boxes2):
        ctx.save_for_backward(boxes1, boxes2)
        return iou2dr_forward(boxes1, boxes2)

    @staticmethod
    def backward(ctx, grad):
        boxes1, boxes2 = ctx.saved_tensors
        return iou2dr_backward(boxes1, boxes2, grad.contiguous())


class GIou2DR(torch.autograd.Function):
    """Differentiable rotated GIoU function for 2D boxes -- reference box/__init__.py:86-107"""

    @staticmethod
    def forward(ctx, boxes1, boxes2):
        ctx.save_for_backward(boxes1, boxes2)
        return giou2dr_forward(boxes1, boxes2)

    @staticmethod
    def backward(ctx, grad):
        boxes1, boxes2 = ctx.saved_tensors
        return giou2dr_backward(boxes1, boxes2, grad.contiguous())


class DIou2DR(torch.autograd.Function):
    """Differentiable rotated DIoU function for 2D boxes -- reference box/__init__.py:109-130"""

    @staticmethod
    def forward(ctx, boxes1, boxes2):
        ctx.save_for_backward(boxes1, boxes2)
        return diou2dr_forward(boxes1, boxes2)

    @staticmethod
    def backward(ctx, grad):
        boxes1, boxes2 = ctx.saved_tensors
        return diou2dr_backward(boxes1, boxes2, grad.contiguous())


class _IouPrecise32(torch.autograd.Function):
    """box2d_iou(precise=True) on fp32 boxes, 'box' / 'rbox': the reference widens the boxes, computes in fp64 and casts the matrix
    back (box/__init__.py:204-205, 224).  Same numbers -- fp64 arithmetic on the widened boxes, every value rounded once -- with
    the widening where the boxes are loaded, the rounding where the matrix is stored, and in backward the widening where the
    incoming gradient is read: no fp64 copy of an [N,M] matrix in either direction, no cast launches in front of the forward."""

    @staticmethod
    def forward(ctx, boxes1, boxes2, iou_type):
        ctx.save_for_backward(boxes1, boxes2)
        ctx.iou_type = iou_type
        return _iou_forward(boxes1, boxes2, iou_type, wide32=True)

    @staticmethod
    def backward(ctx, grad):
        boxes1, boxes2 = ctx.saved_tensors
        g1, g2 = _iou_backward(boxes1, boxes2, grad.contiguous(), ctx.iou_type, wide32=True)
        return g1, g2, None


_IOU_FUNCTIONS = {IouType.BOX: Iou2D, IouType.RBOX: Iou2DR, IouType.GRBOX: GIou2DR, IouType.DRBOX: DIou2DR}


def _ingress(first, *rest):
    """numpy arrays in, numpy arrays out: -> (tensors, was_numpy).  A mix of the two is refused with the reference's message
    (box/__init__.py:191-192, 237-238)."""
    if not isinstance(first, np.ndarray):
        return (first,) + rest, False
    assert all(isinstance(a, np.ndarray) for a in rest), "Input should be both numpy tensor or pytorch tensor!"
    return tuple(torch.from_numpy(a) for a in (first,) + rest), True


def _egress(t, was_numpy):
    return t.numpy() if was_numpy else t


def box2d_iou(boxes1, boxes2, method="box", precise=True):
    """Differentiable IoU on axis-aligned ('box') or rotated ('rbox') 2D boxes, and the loss variants 'grbox' / 'drbox' --
    signature, validation order, messages and dtype handling of reference box/__init__.py:180-224.

    :param boxes1: N x 5 (x,y,w,h,r), torch tensor or numpy array
    :param boxes2: M x 5
    :param precise: compute in float64 and cast back to the input dtype
    """
    (boxes1, boxes2), was_numpy = _ingress(boxes1, boxes2)
    dtype_in = boxes1.dtype
    # fp32 boxes, precise: the fp64 arithmetic with an fp32 matrix (_IouPrecise32) instead of .double() ... .to(float32) around it
    fused32 = precise and dtype_in == torch.float32 and boxes2.dtype == torch.float32
    if precise and not fused32:
        boxes1, boxes2 = boxes1.double(), boxes2.double()
    if boxes1.dim() != 2 or boxes2.dim() != 2:
        raise ValueError("Input of rbox_2d_iou should be Nx2 tensors!")
    if boxes1.shape[1] != 5 or boxes2.shape[1] != 5:
        raise ValueError("Input boxes should have 5 fields: x, y, w, h, r")
    iou_type = getattr(IouType, method.upper())                   # AttributeError for unknown names, like the reference
    fn = _IOU_FUNCTIONS.get(iou_type)
    if fn is None:
        raise ValueError("Unrecognized iou type!")
    if fused32 and iou_type in (IouType.BOX, IouType.RBOX):
        return _egress(_IouPrecise32.apply(boxes1, boxes2, iou_type), was_numpy)
    if fused32:
        boxes1, boxes2 = boxes1.double(), boxes2.double()
    ious = fn.apply(boxes1, boxes2)
    return _egress(ious.to(dtype_in) if precise else ious, was_numpy)


def argsort_desc(scores):
    """stable descending argsort on the device (nms.cpp:103 uses torch's unstable argsort)."""
    lib = _lib.load()
    n = scores.numel()
    dev = scores.device
    order = torch.empty((n,), dtype=torch.int64, device=dev)
    code = _dtype_code(scores)
    with torch.cuda.device(dev):
        ws = _lib.workspace(lib.d3d_argsort_desc_workspace_bytes(n, code), dev)
        rc = lib.d3d_argsort_desc(_lib.ptr(scores), n, code, _lib.ptr(order), _lib.ptr(ws), ws.numel(),
                                  _lib.stream_ptr())
    _lib.check(rc, "argsort_desc")
    return order


NMS_STATUS_DENSE_PATH, NMS_STATUS_SCAN_GAVE_UP = 1, 2


def nms2d(boxes, scores, iou_type, supression_type, iou_threshold, score_threshold, supression_param, sort_keys=None,
          flags=None, return_status=False, keep_mask=False, wide32=False):
    """nms2d / nms2d_cuda (nms.h:6-18; nms.cpp:98-119): returns the SUPPRESSED mask (bool[N]).
    Follows the CPU control flow of the reference (nms.cpp:23-59).  sort_keys: optional fp32 tensor that orders like
    `scores` (the scores before their promotion to fp64): half the radix passes of the argsort, same order.
    return_status: (mask, status) with d3d_nms2d_status's bits -- which route decided the mask (waits for the stream).
    keep_mask: the kernels write the KEEP mask instead (D3D_NMS_KEEP_MASK): box2d_nms's `~suppressed` without the extra pass.
    wide32 (fp32 boxes and scores): the arithmetic in fp64, the values widened where the kernels load them (D3D_F32_WIDE): what
    box2d_nms(precise=True) computes for fp32 tensors, without the .double() copies."""
    lib = _lib.load()
    iou_type, supression_type = int(iou_type), int(supression_type)
    if iou_type not in (IouType.BOX, IouType.RBOX):
        raise ValueError("Unsupported iou type!")                   # common.h:25
    if supression_type not in (0, 1, 2):
        raise ValueError("Unsupported supression type!")            # common.h:40
    odev = boxes.device
    if boxes.dtype != scores.dtype:
        raise RuntimeError("boxes and scores must have the same dtype")
    (b, s), dev = _to_device(boxes, scores)
    n = b.shape[0]
    code = _dtype_code(b)
    if wide32:
        if code != _lib.F32:
            raise ValueError("wide32 takes fp32 boxes and scores")
        code = _lib.F32_WIDE
    with torch.cuda.device(dev):
        # order = None: the library sorts the scores itself (inside the first kernel for up to 4096 boxes)
        order = None
        if sort_keys is not None and sort_keys.dtype == torch.float32 and sort_keys.numel() == n and s.dtype != torch.float32 and n > 4096:
            order = argsort_desc(sort_keys.to(dev).contiguous())
        sup = torch.empty((n,), dtype=torch.uint8, device=dev)
        # both workspaces are carved from one arena; the sort finished with it (same stream)
        ws = _lib.workspace(lib.d3d_nms2d_workspace_bytes(n), dev)
        # (the per-thread host word: the call decides from ITS OWN grid whether the level kernels are worth launching)
        rc = lib.d3d_nms2d_notify(_lib.ptr(b), _lib.ptr(s), _lib.ptr(order) if order is not None else None, n, iou_type, supression_type,
                                  code, float(iou_threshold), float(score_threshold), float(supression_param), _lib.ptr(sup),
                                  _lib.ptr(ws), ws.numel(), _lib.stream_ptr(),
                                  (options.current().nms_flags if flags is None else int(flags)) | (_lib.NMS_KEEP_MASK if keep_mask else 0),
                                  _lib.HostWord.get().ptr)
        _lib.check(rc, "nms2d")
        status = ctypes.c_uint32(0)
        if return_status and n > 0:
            _lib.check(lib.d3d_nms2d_status(_lib.ptr(ws), supression_type, _lib.stream_ptr(), ctypes.byref(status)), "nms2d_status")
    sup = sup.view(torch.bool)
    sup = _lib.to_caller(sup, odev, dev)
    return (sup, int(status.value)) if return_status else sup


nms2d_cuda = nms2d


def box2d_nms(boxes, scores, iou_method="box", supression_method="hard",
              iou_threshold=0, score_threshold=0, supression_param=0, precise=True):
    """NMS on axis-aligned or rotated 2D boxes; returns the KEEP mask -- signature, validation order and messages of reference
    box/__init__.py:226-276 (fp64 promotion of boxes AND scores, class-max of [N,K] scores, the empty case's CPU tensor)."""
    (boxes, scores), was_numpy = _ingress(boxes, scores)
    # (the order of fp32 scores survives the promotion -- fp32 -> fp64 is monotone and injective -- so the sort may use the narrow keys)
    keys = scores if scores.dtype == torch.float32 else None
    # fp32 boxes AND scores, precise: fp64 arithmetic on values widened inside the kernels (D3D_F32_WIDE) instead of .double() copies
    wide32 = precise and boxes.dtype == torch.float32 and scores.dtype == torch.float32
    if precise and not wide32:
        boxes, scores = boxes.double(), scores.double()
    if len(boxes) != len(scores):
        raise ValueError("Numbers of boxes and scores are inconsistent!")
    if scores.dim() == 2:
        scores = scores.max(axis=1).values
        keys = None if keys is None else keys.max(axis=1).values
    if boxes.numel() == 0:
        return torch.tensor([], dtype=torch.bool)
    # (the reference returns ~suppressed, box/__init__.py:272: here the kernels that decide the mask write it inverted)
    keep = nms2d(boxes, scores, getattr(IouType, iou_method.upper()), getattr(SupressionType, supression_method.upper()),
                 iou_threshold, score_threshold, supression_param, sort_keys=None if wide32 else keys, keep_mask=True, wide32=wide32)
    return _egress(keep, was_numpy)


def iou3d(boxes1, boxes2, method="rbox"):
    """Pairwise "3D IoU" of [N,7] x [M,7] boxes (x,y,z,lx,ly,lz,rz) -> f32[N,M]: BEV IoU (rotated for
    'rbox' = box3dr_iou, AABB for 'box' = box3d_iou) times the 1-D z-interval IoU, all in fp32
    (reference d3d/dgal_wrap.h:45-91; the pair loop of BaseMatcher.prepare_boxes,
    d3d/tracking/matcher.pyx:57-80, stores 1 - this value)."""
    lib = _lib.load()
    convert_numpy = False
    if isinstance(boxes1, np.ndarray):
        assert isinstance(boxes2, np.ndarray), "Input should be both numpy tensor or pytorch tensor!"
        boxes1, boxes2 = torch.from_numpy(boxes1), torch.from_numpy(boxes2)
        convert_numpy = True
    key = method.upper()
    if key not in ("RBOX", "BOX"):
        raise ValueError("Unrecognized iou type!")
    if len(boxes1.shape) != 2 or len(boxes2.shape) != 2 or boxes1.shape[1] != 7 or boxes2.shape[1] != 7:
        raise ValueError("Input boxes should have 7 fields: x, y, z, lx, ly, lz, rz")
    odev = boxes1.device
    (b1, b2), dev = _to_device(boxes1.to(torch.float32), boxes2.to(torch.float32))
    n, m = b1.shape[0], b2.shape[0]
    with torch.cuda.device(dev):
        out = torch.empty((n, m), dtype=torch.float32, device=dev)
        if options.current().poison:
            out.fill_(float("nan"))
        ws = _lib.workspace(lib.d3d_iou3d_workspace_bytes(n, m), dev)
        rc = lib.d3d_iou3d_forward(_lib.ptr(b1), n, _lib.ptr(b2), m, 1 if key == "RBOX" else 0, _lib.ptr(out),
                                   _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "iou3d_forward")
    out = _lib.to_caller(out, odev, dev)
    return out.numpy() if convert_numpy else out


def crop_2dr(points, boxes):
    """crop_2dr of the reference (utils.cpp:38-47; box_impl.crop_2dr): bool[M,N] indicators, [i,j] = point j is
    inside rotated box i.  points [N,2], boxes [M,5], same floating dtype."""
    lib = _lib.load()
    if len(points.shape) != 2 or points.shape[1] != 2 or len(boxes.shape) != 2 or boxes.shape[1] != 5:
        raise ValueError("points should be Nx2 and boxes Mx5")
    if points.dtype != boxes.dtype:
        raise RuntimeError("points and boxes must have the same dtype")
    odev = points.device
    (p, b), dev = _to_device(points, boxes)
    n, m = p.shape[0], b.shape[0]
    with torch.cuda.device(dev):
        out = torch.empty((m, n), dtype=torch.uint8, device=dev)
        rc = lib.d3d_crop_2dr(_lib.ptr(p), n, _lib.ptr(b), m, _dtype_code(p), _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "crop_2dr")
    out = out.view(torch.bool)
    return _lib.to_caller(out, odev, dev)


def box2dr_crop(points, boxes):
    """Crop points by rotated boxes -- reference box/__init__.py:278-287 (returns the [M,N] indicator matrix,
    as the reference's code does)."""
    return crop_2dr(points, boxes)


def box3dp_crop(points, boxes, project_axis=2):
    """Crop points [N,3] by boxes [M,7] projected along `project_axis` -- reference box/__init__.py:289-315.  fp32, along z, a cloud
    of 4096 points and more against up to 4096 boxes: ONE launch (d3d_crop_3dp: the same tests on the same float expressions);
    anything else: the reference's composition around crop_2dr."""
    if project_axis not in (0, 1, 2):
        raise ValueError("The projection axis can only be 0-x, 1-y and 2-z!")
    if (project_axis == 2 and torch.is_tensor(points) and torch.is_tensor(boxes) and points.dtype == torch.float32 and
            boxes.dtype == torch.float32 and points.dim() == 2 and boxes.dim() == 2 and points.shape[1] >= 3 and boxes.shape[1] == 7
            and points.shape[0] >= 4096 and 0 < boxes.shape[0] <= 4096):
        lib = _lib.load()
        odev = points.device
        (p, b), dev = _to_device(points.detach(), boxes.detach())
        n, m = p.shape[0], b.shape[0]
        with torch.cuda.device(dev):
            out = torch.empty((m, n), dtype=torch.uint8, device=dev)
            rc = lib.d3d_crop_3dp(_lib.ptr(p), n, p.shape[1], _lib.ptr(b), m, 7, 2, _lib.ptr(out), _lib.stream_ptr())
        if rc != _lib.ERR_UNSUPPORTED:
            _lib.check(rc, "crop_3dp")
            return _lib.to_caller(out.view(torch.bool), odev, dev)
    if project_axis == 0:
        points_2d, boxes_2d = points[:, [1, 2]], boxes[:, [1, 2, 4, 5, 6]]
    elif project_axis == 1:
        points_2d, boxes_2d = points[:, [0, 2]], boxes[:, [0, 2, 3, 5, 6]]
    elif project_axis == 2:
        points_2d, boxes_2d = points[:, [0, 1]], boxes[:, [0, 1, 3, 4, 6]]
    else:
        raise ValueError("The projection axis can only be 0-x, 1-y and 2-z!")
    mask_2d = crop_2dr(points_2d, boxes_2d)
    points_p = points[:, [project_axis]].t()
    boxes_p = boxes[:, [project_axis]]
    boxes_pd = boxes[:, [3 + project_axis]] / 2
    mask_p = (points_p - boxes_pd < boxes_p) & (boxes_p < points_p + boxes_pd)
    return mask_2d & mask_p


def pdist2dr_forward(points, boxes):
    """pdist2dr_forward[_cuda] (dist.h:7-9; dist.cpp:36-52): (distance[M,N], iedge[M,N] uint8) for points[N,2], boxes[M,5];
    signed distance to the box boundary, positive inside"""
    lib = _lib.load()
    if len(points.shape) != 2 or points.shape[1] != 2 or len(boxes.shape) != 2 or boxes.shape[1] != 5:
        raise ValueError("points should be Nx2 and boxes Mx5")
    if points.dtype != boxes.dtype:
        raise RuntimeError("points and boxes must have the same dtype")
    odev = points.device
    (p, b), dev = _to_device(points.detach(), boxes.detach())
    n, m = p.shape[0], b.shape[0]
    with torch.cuda.device(dev):
        dist = torch.empty((m, n), dtype=p.dtype, device=dev)
        iedge = torch.empty((m, n), dtype=torch.uint8, device=dev)
        rc = lib.d3d_pdist2dr_forward(_lib.ptr(p), n, _lib.ptr(b), m, _dtype_code(p), _lib.ptr(dist), _lib.ptr(iedge),
                                      _lib.stream_ptr())
    _lib.check(rc, "pdist2dr_forward")
    return (dist.to(odev), iedge.to(odev)) if odev != dev else (dist, iedge)


def pdist2dr_backward(points, boxes, grad, iedge=None):
    """pdist2dr_backward[_cuda] (dist.h:10-13; dist.cpp:90-110): (grad_boxes[M,5], grad_points[N,2]); `iedge` is accepted
    for signature compatibility and ignored (the nearest feature is recomputed)"""
    lib = _lib.load()
    odev = points.device
    (p, b, g), dev = _to_device(points.detach(), boxes.detach(), grad.to(points.dtype))
    n, m = p.shape[0], b.shape[0]
    with torch.cuda.device(dev):
        gb = torch.empty((m, 5), dtype=p.dtype, device=dev)
        gp = torch.empty((n, 2), dtype=p.dtype, device=dev)
        rc = lib.d3d_pdist2dr_backward(_lib.ptr(p), n, _lib.ptr(b), m, _lib.ptr(g), _dtype_code(p), _lib.ptr(gb), _lib.ptr(gp),
                                       _lib.stream_ptr())
    _lib.check(rc, "pdist2dr_backward")
    return (gb.to(odev), gp.to(odev)) if odev != dev else (gb, gp)


class PDist2DR(torch.autograd.Function):
    """reference box/__init__.py:132-150 -- with its argument mix-up fixed: the compiled functions take (points, boxes)
    (dist.h:7-13) while the reference passes (boxes, points) and returns the two gradients in the wrong order"""

    @staticmethod
    def forward(ctx, points, boxes):
        dist, iedge = pdist2dr_forward(points, boxes)
        ctx.save_for_backward(points, boxes)
        return dist

    @staticmethod
    def backward(ctx, grad):
        points, boxes = ctx.saved_tensors
        grad_boxes, grad_points = pdist2dr_backward(points, boxes, grad.contiguous())
        return grad_points, grad_boxes


def seg1d_iou(seg1, seg2, reference_compat=True):
    """IoU of 1-D segments, row by row: seg1, seg2 [N,2] = (centre, width) -> [N] -- reference box/__init__.py:152-178 (plain
    tensor arithmetic there too).  The default reproduces the reference's values bit for bit, INCLUDING its slip: the half-width
    of seg2 is taken from seg1 (:164).  `reference_compat=False` uses seg2's own width -- INTEGRATION.md section 5"""
    assert torch.all(seg1[:, 1] > 0)
    assert torch.all(seg2[:, 1] > 0)
    d1, d2 = seg1[:, 1] / 2, (seg1 if reference_compat else seg2)[:, 1] / 2
    s1max, s1min, s2max, s2min = seg1[:, 0] + d1, seg1[:, 0] - d1, seg2[:, 0] + d2, seg2[:, 0] - d2
    i = torch.clamp_min(torch.minimum(s1max, s2max) - torch.maximum(s1min, s2min), 0)
    u = torch.clamp_min(torch.maximum(s1max, s2max) - torch.minimum(s1min, s2min), 1e-6)
    return i / u


def seg1d_pdist(points, segs):
    """distance from points [N,1] to 1-D segments [M,2] = (centre, width) -- reference box/__init__.py:317-331
    (positive inside; broadcasts to the [N,M] / [M] shapes the reference's callers use)"""
    assert torch.all(segs[:, 1] > 0)
    dsegs = segs[:, 1] / 2
    smax, smin = segs[:, 0] + dsegs, segs[:, 0] - dsegs
    return torch.where(points > segs[:, 0], smax - points, points - smin)


def box2dr_pdist(points, boxes, method="rbox"):
    """Signed distance from points [N,2] to rotated 2D boxes [M,5] -> [M,N] -- reference box/__init__.py:333-349"""
    if len(boxes.shape) != 2:
        raise ValueError("Input boxes should be Nx2 tensors!")
    if boxes.shape[1] != 5:
        raise ValueError("Input boxes should have 5 fields: x, y, w, h, r")
    if method != "rbox":
        raise ValueError("Only supported rotated boxes by now!")
    return PDist2DR.apply(points, boxes)


def box3dr_pdist(points, boxes, project_axis=2):
    """Signed distance from points [N,3] to 3D boxes [M,7] (surfaces) -> [M,N] -- reference box/__init__.py:351-381.
    (The reference combines a [M,N] planar distance with a [N,M] axial one, which only broadcasts for N == M; here the
    axial distance is laid out [M,N] as well.)"""
    if project_axis == 0:
        points_2d, boxes_2d = points[:, [1, 2]], boxes[:, [1, 2, 4, 5, 6]]
    elif project_axis == 1:
        points_2d, boxes_2d = points[:, [0, 2]], boxes[:, [0, 2, 3, 5, 6]]
    elif project_axis == 2:
        points_2d, boxes_2d = points[:, [0, 1]], boxes[:, [0, 1, 3, 4, 6]]
    else:
        raise ValueError("The projection axis can only be 0-x, 1-y and 2-z!")
    dist_2d = box2dr_pdist(points_2d.contiguous(), boxes_2d.contiguous())
    dist_p = seg1d_pdist(points[:, [project_axis]], boxes[:, [project_axis, 3 + project_axis]]).t()
    return torch.where(
        dist_p > 0,
        torch.where(dist_2d > 0, torch.min(dist_p, dist_2d), dist_2d),
        torch.where(dist_2d > 0, dist_p, -torch.sqrt(dist_2d.square() + dist_p.square())))


# north_star operator names
iou2d = box2d_iou
nms = box2d_nms

__all__ = ["Iou2D", "Iou2DR", "GIou2DR", "DIou2DR", "PDist2DR", "iou2d_backward", "iou2dr_backward", "giou2dr_forward",
           "giou2dr_backward", "diou2dr_forward", "diou2dr_backward", "iou2dr_flags", "pdist2dr_forward", "pdist2dr_backward",
           "box2dr_crop", "box3dp_crop", "box2dr_pdist", "box3dr_pdist", "seg1d_pdist", "seg1d_iou", "crop_2dr", "box2d_iou", "box2d_nms",
           "iou2d", "iou3d", "nms", "iou2d_forward", "iou2dr_forward", "nms2d", "nms2d_cuda", "argsort_desc", "IouType",
           "SupressionType", "cuda_available"]
