"""Rotated IoU backward on the reference's benchmark boxes (test/compare/benchmark_riou.py: centres in +-5, sizes in [0, 5), angles
in +-5 rad -- 28 % of the pairs overlap), per kernel (HIP events).  usage: python tools/riou_bwd_ab.py"""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from d3d_amd import _lib  # noqa: E402
from d3d_amd.box import box2d_iou  # noqa: E402

import os  # noqa: E402
if os.environ.get("D3D_X_LIB"):          # development aid: an experimental build of the library
    _lib.LIB_PATH = os.environ["D3D_X_LIB"]
rng = np.random.default_rng(11)
lib = _lib.load()
for n in (1000, 2000, 5000):
    mk = lambda: np.stack([(rng.random(n) - 0.5) * 10, (rng.random(n) - 0.5) * 10, rng.random(n) * 5, rng.random(n) * 5,  # noqa: E731
                           (rng.random(n) - 0.5) * 10], 1).astype(np.float32)
    p1, p2 = torch.from_numpy(mk()).cuda().requires_grad_(True), torch.from_numpy(mk()).cuda().requires_grad_(True)
    out = box2d_iou(p1, p2, method="rbox")
    torch.cuda.synchronize()
    lib.d3d_profile_enable(1)
    box2d_iou(p1, p2, method="rbox")
    torch.cuda.synchronize()
    lib.d3d_profile_enable(0)
    buf = ctypes.create_string_buffer(1 << 16)
    lib.d3d_profile_report(buf, len(buf))
    print(f"rbox {n}x{n} forward kernels | us: " + " ".join("%s %.0f" % (ln.rsplit(",", 2)[0], 1e3 * float(ln.rsplit(",", 2)[2])) for ln in buf.value.decode().strip().splitlines()), flush=True)
    w = torch.ones_like(out)
    best = 1e9
    for _ in range(6):
        p1.grad = p2.grad = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out.backward(w, retain_graph=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    lib.d3d_profile_enable(1)
    p1.grad = p2.grad = None
    out.backward(w, retain_graph=True)
    torch.cuda.synchronize()
    lib.d3d_profile_enable(0)
    buf = ctypes.create_string_buffer(1 << 16)
    lib.d3d_profile_report(buf, len(buf))
    ks = " ".join("%s %.0f" % (ln.rsplit(",", 2)[0], 1e3 * float(ln.rsplit(",", 2)[2])) for ln in buf.value.decode().strip().splitlines())
    print(f"rbox {n}x{n} backward: {best * 1e3:8.3f} ms ({int((out > 0).sum())} overlapping pairs) | us: {ks}", flush=True)
    g1 = p1.grad.clone()
    print("   |grad| sums", float(p1.grad.abs().sum()), float(p2.grad.abs().sum()), flush=True)
