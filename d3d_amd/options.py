"""Per-call options of the operators: which of several equivalent internal paths a call takes, and the test hook that
poisons output buffers.  They are arguments, not library state -- every operator takes them as keywords (`flags=`,
`poison=`), and a caller that cannot reach the keyword (the call sits behind an autograd Function, a VoxelGenerator, the
reference's own test code) scopes them with

    with d3d_amd.options.scope(voxel_flags=_lib.VOXEL_PATH_HASH, poison=True):
        ...

which binds them to the CALLING CONTEXT (a contextvars.ContextVar: per thread and per asyncio task), never to the module:
two threads or streams can run different options at the same time, and nothing in the package writes them.  The C ABI
receives them as the `flags` argument of each call (include/d3d_hip.h).
"""
import contextlib
import contextvars


class CallOptions:
    __slots__ = ("voxel_flags", "nms_flags", "iou_flags", "poison")

    def __init__(self, voxel_flags=0, nms_flags=0, iou_flags=0, poison=False):
        self.voxel_flags, self.nms_flags, self.iou_flags, self.poison = int(voxel_flags), int(nms_flags), int(iou_flags), bool(poison)

    def replace(self, **kw):
        vals = {k: getattr(self, k) for k in self.__slots__}
        for k, v in kw.items():
            if k not in vals:
                raise TypeError("unknown option %r" % k)
            vals[k] = v
        return CallOptions(**vals)


_DEFAULT = CallOptions()
_current = contextvars.ContextVar("d3d_amd_call_options", default=_DEFAULT)


def current():
    """the options of the calling context (defaults: automatic paths, no poisoning)"""
    return _current.get()


def push(**kw):
    """bind options for the calling context; returns the token for pop()"""
    return _current.set(_current.get().replace(**kw))


def pop(token):
    _current.reset(token)


@contextlib.contextmanager
def scope(**kw):
    token = push(**kw)
    try:
        yield
    finally:
        pop(token)
