"""dense voxelizer at the C ABI: eager launches vs the same call captured into a HIP graph (development aid)"""
import ctypes, sys, time
sys.path.insert(0, ".")
import torch
from d3d_amd import _lib, synth

lib = _lib.load()
n, P = 1000000, 32
pts = torch.from_numpy(synth.lidar_like(n, 0)).cuda()
shape = (ctypes.c_int32 * 3)(*synth.KITTI_SHAPE)
bound = (ctypes.c_float * 6)(*synth.KITTI_BOUNDS)
voxels = torch.empty((n, P, 4), device="cuda")
coords = torch.empty((n, 3), dtype=torch.int64, device="cuda")
pmask = torch.empty((n, P), dtype=torch.uint8, device="cuda")
npts = torch.empty((n,), dtype=torch.int32, device="cuda")
agg = torch.empty((n, 4), device="cuda")
counts = torch.empty((_lib.NUM_COUNTS,), dtype=torch.int64, device="cuda")
ws = _lib.workspace(lib.d3d_voxelize_workspace_bytes(n, 0), pts.device)


def call():
    rc = lib.d3d_voxelize_3d_dense(_lib.ptr(pts), n, 4, ctypes.cast(shape, ctypes.c_void_p), ctypes.cast(bound, ctypes.c_void_p),
                                   P, n, 1, _lib.ptr(voxels), _lib.ptr(coords), _lib.ptr(pmask), _lib.ptr(npts), _lib.ptr(agg),
                                   _lib.ptr(counts), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(), 0)
    assert rc == 0


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


print("eager C call, back to back: %.1f us" % timeit(call))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    call()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        call()
torch.cuda.synchronize()
print("graph replay:               %.1f us" % timeit(g.replay))
print("voxels", int(counts.cpu()[0]))
