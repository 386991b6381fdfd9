"""ctypes binding of libd3d_hip.so (C ABI: include/d3d_hip.h).

There is no CPU fallback: if the HIP library is missing or no GPU is visible the
operators raise -- they never silently compute somewhere else.
"""
import ctypes
import threading
import time
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libd3d_hip.so")

OK, ERR_BAD_ARG, ERR_UNSUPPORTED, ERR_WORKSPACE, ERR_HIP = 0, -1, -2, -3, -4
COUNT_VOXELS, COUNT_POINTS, COUNT_STATUS, COUNT_AUX, NUM_COUNTS = 0, 1, 2, 3, 4
STATUS_COORD_OVERFLOW, STATUS_TABLE_FULL, STATUS_PACK_OVERFLOW, STATUS_BIN_OVERFLOW = 1, 2, 4, 8
F32, F64, F64_M32, F32_WIDE = 0, 1, 2, 3      # (F64_M32: d3d_iou2d_forward / _backward -- fp64 boxes and arithmetic, fp32 [n,m] matrix)
# per-call option bits (include/d3d_hip.h)
VOXEL_PATH_HASH, VOXEL_PARTITION_3PASS, VOXEL_PLAIN_SLOTS, VOXEL_SPLIT_FILL, VOXEL_EXACT_MEAN, VOXEL_WIDE_KEYS = 1, 2, 4, 8, 16, 64
OWNER_MERGE_CHAINS, OWNER_MERGE_TEST_TINY = 1, 2
NMS_BROAD_SWEEP, NMS_FORCE_DENSE, NMS_SOFT_NO_LDS, NMS_GENERAL, NMS_TEST_WITHHOLD, NMS_FORCE_LEVELS, NMS_ONE_LEVEL = 1, 2, 4, 8, 16, 32, 64
NMS_KEEP_MASK = 128


def nms_cand_cap(k):
    return int(k) << 8


def iou_list_cap(k):
    return int(k) << 8


_vp, _i64, _i32, _sz, _f32, _u32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_size_t, ctypes.c_float, ctypes.c_uint32

# name -> (restype, argtypes); must list every symbol include/d3d_hip.h declares
SIGNATURES = {
    "d3d_abi_version": (ctypes.c_int, []),
    "d3d_last_hip_error": (ctypes.c_int, []),
    "d3d_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "d3d_voxelize_workspace_bytes": (_sz, [_i64, _i64]),
    "d3d_voxelize_3d_dense": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _i32, _i32, _i32,
                                             _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _u32]),
    "d3d_voxelize_3d_dense_notify": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _i32, _i32, _i32,
                                             _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _u32]),
    "d3d_voxelize_3d_dense_resident": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _i32, _i32, _i32,
                                             _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _u32]),
    "d3d_voxelize_3d_dense_staged": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _i32, _i32, _i32,
                                             _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _u32, _i32]),
    "d3d_voxelize_3d_sparse": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _u32]),
    "d3d_voxelize_3d_filter": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32,
                                              _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "d3d_voxelize_3d_sparse_filter": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                                     _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _u32, _vp]),
    "d3d_voxelize_3d_filter_chained": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32,
                                                      _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "d3d_voxelize_reduce_rows": (_sz, [_i64]),
    "d3d_voxelize_3d_reduce": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp,
                                              _i32, _vp, _vp, _vp, _vp, _sz, _vp, _u32]),
    "d3d_sharded_scatter": (ctypes.c_int, [_vp, _i64, _i64, _i64, _i64, _vp, _sz, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i32,
                                           _vp, _vp, _vp, _vp, _vp]),
    "d3d_sharded_finalize": (ctypes.c_int, [_i64, _i32, _vp, _i64, _vp, _vp, _sz, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp,
                                            _vp, _vp, _vp]),
    "d3d_sharded_map": (ctypes.c_int, [_i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "d3d_grid_compact_workspace_bytes": (_sz, [_i64]),
    "d3d_grid_compact_index": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "d3d_grid_bitmap_mark": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "d3d_grid_owner_workspace_bytes": (_sz, [_i64]),
    "d3d_grid_compact_from_bitmaps": (ctypes.c_int, [_vp, _i64, _i32, _i64, _vp, _vp, _sz, _i32, _vp, _sz, _vp]),
    "d3d_sharded_scatter_owned": (ctypes.c_int, [_vp, _i64, _i64, _vp, _sz, _vp, _sz, _i32, _i64, _i32, _i32, _vp, _vp, _vp,
                                                 _i32, _vp, _vp, _vp, _sz, _vp]),
    "d3d_sharded_finalize_owned": (ctypes.c_int, [_i64, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "d3d_grid_compact_keys": (ctypes.c_int, [_i64, _vp, _sz, _vp, _vp]),
    "d3d_grid_compact_lookup": (ctypes.c_int, [_vp, _i64, _i64, _vp, _sz, _i64, _vp, _vp]),
    "d3d_owner_record_words": (ctypes.c_int, [_i32]),
    "d3d_owner_pack_workspace_bytes": (_sz, [_i64, _i32]),
    "d3d_owner_pack": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _vp, _sz, _vp, _vp, _i64]),
    "d3d_owner_merge_workspace_bytes": (_sz, [_i64, _i32]),
    "d3d_owner_merge": (ctypes.c_int, [_vp, _i64, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _u32, _vp]),
    "d3d_owner_dense": (ctypes.c_int, [_vp, _i64, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _sz, _vp, _vp, _vp, _u32, _vp]),
    "d3d_owner_mark_first": (ctypes.c_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "d3d_owner_number_workspace_bytes": (_sz, [_i64]),
    "d3d_owner_number": (ctypes.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "d3d_owner_reply": (ctypes.c_int, [_i64, _vp, _vp, _vp, _vp]),
    "d3d_owner_map": (ctypes.c_int, [_i64, _vp, _vp, _vp, _vp, _vp]),
    "d3d_owner_replicate": (ctypes.c_int, [_i64, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _sz]),
    "d3d_aligned_scatter_workspace_bytes": (_sz, [_i64, _i64, _vp, _i32, _i32]),
    "d3d_aligned_scatter_forward": (ctypes.c_int, [_vp, _i64, _i32, _vp, _i64, _i64, _vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "d3d_aligned_scatter_backward": (ctypes.c_int, [_vp, _i64, _i32, _vp, _i64, _i64, _vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "d3d_profile_enable": (ctypes.c_int, [ctypes.c_int]),
    "d3d_profile_report": (ctypes.c_int, [ctypes.c_char_p, _sz]),
    "d3d_voxelize_dense_last_plan": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int64)]),
    "d3d_voxelize_3d_sparse_filter_call_layout": (_sz, [_i64, ctypes.c_int32, ctypes.POINTER(ctypes.c_size_t)]),
    "d3d_voxelize_3d_sparse_filter_call_workspace_bytes": (_sz, [_i64]),
    "d3d_voxelize_3d_sparse_filter_call": (ctypes.c_int, [ctypes.c_void_p]),
    "d3d_stream_probe": (ctypes.c_int, [ctypes.c_int, _vp, _sz, _vp]),
    "d3d_iou2d_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "d3d_iou2d_forward": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _sz, _vp, _u32]),
    "d3d_iou2d_backward": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "d3d_iou2dr_flags": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "d3d_pdist2dr_forward": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp]),
    "d3d_pdist2dr_backward": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _vp, _vp]),
    "d3d_iou3d_workspace_bytes": (_sz, [_i64, _i64]),
    "d3d_iou3d_forward": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _sz, _vp]),
    "d3d_match_distance": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _sz, _vp]),
    "d3d_score_match_workspace_bytes": (_sz, [_i64, _i64]),
    "d3d_score_match": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "d3d_score_match_batched_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "d3d_score_match_batched": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "d3d_crop_2dr": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp]),
    "d3d_crop_3dp": (ctypes.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _vp, _vp]),
    "d3d_crop_3dr": (ctypes.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _vp, _vp]),
    "d3d_paint_label": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "d3d_argsort_desc_workspace_bytes": (_sz, [_i64, _i32]),
    "d3d_argsort_desc": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _sz, _vp]),
    "d3d_nms2d_workspace_bytes": (_sz, [_i64]),
    "d3d_nms2d": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _f32, _f32, _vp, _vp, _sz, _vp, _u32]),
    "d3d_nms2d_notify": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _f32, _f32, _vp, _vp, _sz, _vp, _u32, _vp]),
    "d3d_nms2d_status": (ctypes.c_int, [_vp, _i32, _vp, _vp]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises ImportError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "Cannot find compiled library! %s is missing: build it with "
                "`make -C d3d_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError here = header/library mismatch: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("d3d_amd needs a HIP device (MI355X); there is no CPU fallback in the product path")
    return torch.device("cuda", torch.cuda.current_device())


def check(status, what):
    """Map a d3d_status to the exception type the reference raises for the same condition."""
    if status == OK:
        return
    lib = load()
    msg = "%s: %s" % (what, lib.d3d_status_string(status).decode())
    if status in (ERR_BAD_ARG, ERR_UNSUPPORTED):
        raise ValueError(msg)                       # reference: py::value_error
    if status == ERR_HIP:
        msg += " (hipError %d)" % lib.d3d_last_hip_error()
    raise RuntimeError(msg)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else ctypes.c_void_p(0)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_raw(device_index=None):
    """the current stream of a device (default: the current device) as an integer handle.  torch.cuda.current_stream() builds
    a Stream object through several Python layers (8 us per call, three calls per operator call: more than the 3 kernel launches
    of the sparse operator); the raw query underneath is a fraction of a microsecond."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device() if device_index is None else device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


def stream_ptr():
    return ctypes.c_void_p(stream_raw())


_ws_local = threading.local()       # the arenas of a host thread die with it (a global registry kept every short-lived
                                    # worker thread's arena -- up to GBs for a 100 k-box NMS -- allocated for good)
_WS_MAX_STREAMS = 8                 # arenas a thread keeps (one per (device, stream) it used last)


def workspace(nbytes, device):
    """Reusable scratch arena per (device, stream, host thread); grows geometrically.  Per thread as well: two threads that
    issue multi-kernel operators on the SAME stream interleave their launches, and each operator must keep its scratch to
    itself for that to be harmless (the C ABI is re-entrant given distinct workspaces).  A thread keeps the arenas of the
    last few (device, stream) pairs it used; they are freed when the thread ends."""
    cache = getattr(_ws_local, "cache", None)
    if cache is None:
        cache = _ws_local.cache = {}
    key = (device.index, stream_raw(device.index))
    buf = cache.pop(key, None)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
    cache[key] = buf                 # most recently used last
    while len(cache) > _WS_MAX_STREAMS:
        cache.pop(next(iter(cache)))
    return buf


_PINNED_RESULT_MIN_BYTES = 1 << 20


def to_caller(results, odev, dev):
    """results (a tensor, or a dict / tuple of tensors on `dev`) on the CALLER's device `odev` -- the reference's operators take
    and return CPU tensors (voxelize.cpp), so a drop-in caller may hand over host tensors and expects host tensors back.  To the
    host, tensors of 1 MiB and more go through torch's pinned host allocator (cached blocks) with asynchronous copies and ONE
    wait on the stream: config 2's 328 MB of dense outputs take 6.5 ms that way instead of 47 ms for pageable destinations
    (tensor.to("cpu") allocates and faults in fresh pages, then copies through a staging buffer).  The tensors that come back
    are ordinary CPU tensors whose memory happens to be page-locked."""
    if odev == dev:
        return results
    if isinstance(results, dict):
        keys = list(results)
        vals = to_caller(tuple(results[k] for k in keys), odev, dev)
        return {k: v for k, v in zip(keys, vals)}
    single = torch.is_tensor(results)
    seq = (results,) if single else tuple(results)
    if odev.type != "cpu":
        out = tuple(t.to(odev) for t in seq)
    else:
        out, pending = [], False
        for t in seq:
            if t.numel() * t.element_size() >= _PINNED_RESULT_MIN_BYTES:
                h = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
                h.copy_(t, non_blocking=True)
                out.append(h)
                pending = True
            else:
                out.append(t.to(odev))
        if pending:
            torch.cuda.current_stream(dev).synchronize()
        out = tuple(out)
    return out[0] if single else out


class HostWord:
    """host-mapped pinned int32 word, one per thread: d3d_nms2d_notify's density verdict lands here"""
    _local = threading.local()

    def __init__(self):
        self.tensor = torch.zeros((16,), dtype=torch.int32).pin_memory()
        self.ptr = ctypes.c_void_p(self.tensor.data_ptr())

    @classmethod
    def get(cls):
        buf = getattr(cls._local, "buf", None)
        if buf is None:
            buf = cls._local.buf = cls()
        return buf


class NotifyBuffer:
    """host-mapped pinned int64[NUM_COUNTS + 1] that d3d_voxelize_3d_dense_notify fills as soon as the output sizes are
    final (flag word last); one per thread.  wait() polls the flag -- the GPU keeps writing the outputs meanwhile."""
    _local = threading.local()

    def __init__(self):
        self.tensor = torch.zeros((2 * NUM_COUNTS + 1,), dtype=torch.int64).pin_memory()
        self.arr = self.tensor.numpy()
        self.ptr = ctypes.c_void_p(self.tensor.data_ptr())

    @classmethod
    def get(cls):
        buf = getattr(cls._local, "buf", None)
        if buf is None:
            buf = cls._local.buf = cls()
        return buf

    def arm(self):
        self.arr[NUM_COUNTS] = 0

    def wait(self, counts, spin_s=0.05):
        """-> list of the NUM_COUNTS values (counts[2, NUM_COUNTS]: the two rows, flattened).  Falls back to a blocking read
        of the device counters (which also surfaces HIP errors) if the flag has not appeared after spin_s."""
        arr, t0 = self.arr, time.perf_counter()
        while arr[NUM_COUNTS] == 0:
            if time.perf_counter() - t0 > spin_s:
                return counts.cpu().reshape(-1).tolist()
        if counts.dim() == 2:
            return arr[:NUM_COUNTS].tolist() + arr[NUM_COUNTS + 1:].tolist()
        return arr[:NUM_COUNTS].tolist()
