"""GIoU / DIoU forward: G pairs/s of k_loss_iou at config 3's box density (10 k x 10 k fp64, 20 k x 20 k fp32), the figure
VERDICT r04 item 7 asks for, plus parity of the result against the oracle on a 300 x 300 corner.  usage: python tools/giou_ab.py"""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import oracle  # noqa: E402
from d3d_amd import synth  # noqa: E402
from d3d_amd.box import box2d_iou  # noqa: E402

b, _ = synth.boxes2d_sparse(100000, 1)
for dtype, n in ((torch.float64, 10000), (torch.float32, 20000)):
    bl = torch.from_numpy(b[:n]).cuda().to(dtype)
    for method in ("grbox", "drbox"):
        box2d_iou(bl, bl, method=method)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            out = box2d_iou(bl, bl, method=method)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        sub = bl[:300].cpu().numpy()
        ref = oracle.box2d_iou(sub, sub, method, precise=(dtype == torch.float64))
        err = float(np.abs(out[:300, :300].double().cpu().numpy() - ref.astype(np.float64)).max())
        from d3d_amd import _lib
        lib = _lib.load()
        lib.d3d_profile_enable(1)
        box2d_iou(bl, bl, method=method)
        torch.cuda.synchronize()
        lib.d3d_profile_enable(0)
        buf = ctypes.create_string_buffer(1 << 16)
        lib.d3d_profile_report(buf, len(buf))
        ks = " ".join("%s %.0f" % (ln.rsplit(",", 2)[0], 1e3 * float(ln.rsplit(",", 2)[2])) for ln in buf.value.decode().strip().splitlines())
        print(f"{method} {str(dtype)[6:]} {n}x{n}: {n * n / best / 1e9:7.2f} G pairs/s   max |err| vs oracle {err:.2e} | us: {ks}", flush=True)
