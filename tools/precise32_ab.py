"""box2d_iou(precise=True) on fp32 boxes: the fused form (fp64 arithmetic, fp32 matrix: D3D_F64_M32) against the explicit chain the
reference's Python layer spells out (boxes.double() -> fp64 kernels -> ious.to(float32); backward: the gradient widened to fp64
first) -- forward and forward + backward, the reference's benchmark boxes and config 3's density.  usage: python tools/precise32_ab.py"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from d3d_amd import synth
from d3d_amd.box import Iou2DR, box2d_iou


def wall(fn, k):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / k


for name, gen, n, k in (("dense 5 k x 5 k", synth.boxes2d_dense, 5000, 20), ("dense 2 k x 2 k", synth.boxes2d_dense, 2000, 50),
                        ("sparse 10 k x 10 k", synth.boxes2d_sparse, 10000, 20), ("sparse 30 k x 30 k", synth.boxes2d_sparse, 30000, 5)):
    b1 = torch.from_numpy(gen(n, 1)[0].astype(np.float32)).cuda()
    b2 = torch.from_numpy(gen(n, 2)[0].astype(np.float32)).cuda()
    fused = lambda: box2d_iou(b1, b2, method="rbox")                                      # noqa: E731
    chain = lambda: Iou2DR.apply(b1.double(), b2.double()).to(torch.float32)              # noqa: E731
    same = torch.equal(fused(), chain())

    def fb(f):
        def run():
            t2 = b2.detach().requires_grad_(True)
            if f == "fused":
                r = box2d_iou(b1, t2, method="rbox")
            else:
                r = Iou2DR.apply(b1.double(), t2.double()).to(torch.float32)
            r.sum().backward()
        return run
    for rep in range(2):
        print("%-20s forward: fused %8.1f us  chain %8.1f us  %s | forward + backward: fused %8.1f us  chain %8.1f us" % (
            name, wall(fused, k), wall(chain, k), "same" if same else "DIFF", wall(fb("fused"), k), wall(fb("chain"), k)), flush=True)
    del b1, b2
    torch.cuda.empty_cache()
