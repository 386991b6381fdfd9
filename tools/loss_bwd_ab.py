"""GIoU / DIoU backward: ms per call of d(sum(w * iou)) on config 3's boxes, dense weights and weights on one pair per row
(a loss on matched pairs), 2 k x 2 k (bench.py's figure) and 6 k x 6 k.  usage: python tools/loss_bwd_ab.py"""
import ctypes
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from d3d_amd import synth  # noqa: E402
from d3d_amd.box import box2d_iou  # noqa: E402

b, _ = synth.boxes2d_sparse(100000, 1)
for n in (2000, 6000):
    b1 = torch.from_numpy(b[:n]).cuda().requires_grad_(True)
    b2 = torch.from_numpy(b[n:2 * n]).cuda().requires_grad_(True)
    dense = torch.ones((n, n), dtype=torch.float64, device="cuda")
    picked = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    picked[torch.arange(n), torch.randint(0, n, (n,))] = 1.0
    for method in ("grbox", "drbox"):
        for name, w in (("dense", dense), ("picked", picked)):
            out = box2d_iou(b1, b2, method=method)
            best = 1e9
            for _ in range(6):
                b1.grad = b2.grad = None
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out.backward(w, retain_graph=True)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            from d3d_amd import _lib
            lib = _lib.load()
            torch.cuda.synchronize()
            lib.d3d_profile_enable(1)
            b1.grad = b2.grad = None
            out.backward(w, retain_graph=True)
            torch.cuda.synchronize()
            lib.d3d_profile_enable(0)
            buf = ctypes.create_string_buffer(1 << 16)
            lib.d3d_profile_report(buf, len(buf))
            ks = " ".join("%s %.0f" % (ln.rsplit(",", 2)[0], 1e3 * float(ln.rsplit(",", 2)[2])) for ln in buf.value.decode().strip().splitlines())
            print(f"{method} {n}x{n} backward, {name} weights: {best * 1e3:8.3f} ms  ({n * n / best / 1e9:6.2f} G pairs/s) | us: {ks}", flush=True)

# a crowded scene (a third of the pairs overlap: most pairs take the complete routine) -- the two-kernel path against the
# one-kernel path (no workspace), which is what ran before round 5
from d3d_amd import _lib as _L  # noqa: E402
dense_boxes, _ = synth.boxes2d_dense(2000, 5)
for method in ("grbox", "drbox"):
    for label, patched in (("two kernels", False), ("one kernel", True)):
        keep = _L.workspace
        if patched:
            _L.workspace = lambda nbytes, dev: torch.empty((0,), dtype=torch.uint8, device=dev)
        try:
            b1 = torch.from_numpy(dense_boxes[:2000]).cuda().requires_grad_(True)
            b2 = torch.from_numpy(dense_boxes[:2000].copy()).cuda().requires_grad_(True)
            w = torch.ones((2000, 2000), dtype=torch.float64, device="cuda")
            out = box2d_iou(b1, b2, method=method)
            best = 1e9
            for _ in range(5):
                b1.grad = b2.grad = None
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out.backward(w, retain_graph=True)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            print(f"{method} crowded 2000x2000 backward, {label}: {best * 1e3:8.3f} ms", flush=True)
        finally:
            _L.workspace = keep
