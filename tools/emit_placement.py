"""Does k_emit's duration depend on WHERE its outputs lie?  The same frame through d3d_voxelize_3d_dense with the five output
arrays carved from one arena at different relative skews (multiples of 256 bytes up to a few MB), per-kernel HIP-event times.
usage (GPU box): python tools/emit_placement.py"""
import ctypes, sys
sys.path.insert(0, ".")
import torch
import bench
from d3d_amd import _lib, synth

lib = _lib.load()
n, P = 1000000, 32
pts = torch.from_numpy(synth.lidar_like(n, 0)).cuda()
shape = (ctypes.c_int32 * 3)(*synth.KITTI_SHAPE)
bound = (ctypes.c_float * 6)(*synth.KITTI_BOUNDS)
ws = torch.empty((lib.d3d_voxelize_workspace_bytes(n, 0),), dtype=torch.uint8, device="cuda")
counts = torch.empty((4,), dtype=torch.int64, device="cuda")
sizes = [n * P * 16, n * 24, n * P, n * 4, n * 16]          # voxels, coords, pmask, npoints, aggregates
arena = torch.empty((sum(sizes) + (64 << 20),), dtype=torch.uint8, device="cuda")
base = arena.data_ptr()
base += (-base) % (2 << 20)


def run(skews):
    ptrs, off = [], 0
    for s, k in zip(sizes, skews):
        ptrs.append(ctypes.c_void_p(base + off + k))
        off += s + (8 << 20)
        off += (-off) % (2 << 20)

    def call():
        rc = lib.d3d_voxelize_3d_dense(_lib.ptr(pts), n, 4, ctypes.cast(shape, ctypes.c_void_p), ctypes.cast(bound, ctypes.c_void_p), P, n, 1,
                                       ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], _lib.ptr(counts), _lib.ptr(ws), ws.numel(),
                                       _lib.stream_ptr(), 0)
        assert rc == 0, rc
    prof = bench.kernel_profile(call, 30)
    return prof["k_emit"]["avg_us"]


for name, sk in [("all 2 MB aligned", (0, 0, 0, 0, 0)), ("+256 B steps", (0, 256, 512, 768, 1024)), ("+4 KB steps", (0, 4096, 8192, 12288, 16384)),
                 ("+64 KB steps", (0, 65536, 131072, 196608, 262144)), ("+1 MB +  odd", (0, (1 << 20) + 256, (1 << 19) + 1024, (1 << 18) + 4096, 768)),
                 ("voxels + 1 KB", (1024, 0, 0, 0, 0)), ("voxels + 128 KB", (131072, 0, 0, 0, 0))]:
    print("%-18s k_emit %.1f us  %.1f us" % (name, run(sk), run(sk)), flush=True)
