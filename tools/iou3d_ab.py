"""Timing of the pairwise 3D IoU (config 4: 20 k x 5 k fp32, and other shapes) with its launches' HIP-event times.  (Round 6 used it
as an A/B of two one-launch forms against the library's launches, through a debug switch that is gone with them:
profiles/r06_iou3d_fused_ab.txt.)
usage (GPU box): python tools/iou3d_ab.py [steps]"""
import sys
import torch
sys.path.insert(0, ".")
import bench
from d3d_amd import _lib, synth
from d3d_amd.box import iou3d
from d3d_amd.tracking import DistanceTypes, prepare_boxes
import numpy as np

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
if len(sys.argv) > 2:      # another build of the library (e.g. libd3d_hip_tune.so left from before a change): same-box comparison
    import os
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[2])
lib = _lib.load()
for n_gt, rep in ((5000, 4), (20000, 1), (2000, 4), (10000, 2)):
    p, g = synth.boxes3d_eval(n_gt, rep, 2)
    pt, gt = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
    res = {}
    for r in range(3):
        for name, on in (("default", 0), ("rows8", 8), ("rows16", 16), ("rows32", 32), ("rows64", 64)):
            lib.d3d_debug_set_pre_rows(on)
            for method in ("rbox", "box"):
                out = iou3d(pt, gt, method=method)
                res.setdefault((name, method), out.clone())
                dt = bench.timed(lambda: iou3d(pt, gt, method=method), steps, 3)
                prof = bench.kernel_profile(lambda: iou3d(pt, gt, method=method), steps)
                same = torch.equal(out, res[(name, method)])
                print("%d x %d %-5s %-15s %s %8.1f us/call %7.1f Gpairs/s | " % (len(p), len(g), method, name, "same" if same else "DIFF",
                      1e6 * dt / steps, len(p) * len(g) * steps / dt / 1e9) +
                      " ".join("%s %.1f" % (k.replace("k_", ""), v["avg_us"]) for k, v in prof.items()), flush=True)
