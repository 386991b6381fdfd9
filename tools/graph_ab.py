"""the operators replayed from a captured HIP graph against the same calls issued eagerly (the library launches on the caller's
current stream, so a caller can capture it: torch.cuda.graph): wall-clock per call over `steps` back-to-back calls, outputs compared
bit for bit.  usage (GPU box): python tools/graph_ab.py [steps]"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from d3d_amd import synth
from d3d_amd.box import box2d_iou, box2d_nms, iou3d

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def flat(out):
    if isinstance(out, dict):
        return [out[k] for k in sorted(out) if torch.is_tensor(out[k])]
    return list(out) if isinstance(out, (tuple, list)) else [out]


def wall(fn, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / k


def case(name, fn):
    ref = [t.clone() for t in flat(fn())]
    eager = min(wall(fn, steps) for _ in range(3))
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        g.replay()
        torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(flat(out), ref))
        replay = min(wall(g.replay, steps) for _ in range(3))
        print("%-34s eager %8.1f us   graph replay %8.1f us   %s" % (name, eager, replay, "same" if same else "DIFF"), flush=True)
    except Exception as e:      # noqa: BLE001
        torch.cuda.synchronize()
        print("%-34s eager %8.1f us   capture failed: %s" % (name, eager, str(e).splitlines()[0][:150]), flush=True)


# (the voxel operators return tensors cut to the voxel count, which the host reads inside the call: not capturable by contract)
b, s = synth.boxes2d_sparse(100000, 1)
bt, st = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
case("box2d_nms rbox, 100 k boxes", lambda: box2d_nms(bt, st, iou_method="rbox", iou_threshold=0.3))
p3, g3 = synth.boxes3d_eval(5000, 4, 2)
pt, gt = torch.from_numpy(p3).cuda(), torch.from_numpy(g3).cuda()
case("iou3d, config 4", lambda: iou3d(pt, gt))
d5 = torch.from_numpy(synth.boxes2d_dense(5000, 1)[0]).cuda().float()
case("box2d_iou rbox fp32 5 k x 5 k", lambda: box2d_iou(d5, d5, method="rbox", precise=False))
small = torch.from_numpy(synth.boxes2d_sparse(2000, 1)[0]).cuda()
case("box2d_iou rbox fp64 2 k x 2 k", lambda: box2d_iou(small, small, method="rbox"))
