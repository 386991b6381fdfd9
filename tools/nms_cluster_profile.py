#!/usr/bin/env python3
"""box2d_nms on detector-like input: clusters of heavily overlapping boxes around each object (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import kernel_profile, timed
from d3d_amd.box import box2d_nms

def clustered(nobj, per, seed):
    rng = np.random.default_rng(seed)
    c = np.stack([rng.random(nobj) * 2000, rng.random(nobj) * 2000, rng.random(nobj) * 20 + 10, rng.random(nobj) * 20 + 10,
                  rng.random(nobj) * 6.28], 1)
    b = np.repeat(c, per, 0) + rng.normal(0, 1, (nobj * per, 5)) * [1.5, 1.5, 1.0, 1.0, 0.05]
    return b, rng.random(nobj * per)

from d3d_amd import _lib
from d3d_amd.box import nms2d, IouType, SupressionType
modes = {"default": 0, "1 level": _lib.NMS_ONE_LEVEL}
for nobj, per in [(1000, 100), (5000, 20), (200, 500), (20000, 5)]:
    b, s = clustered(nobj, per, 1)
    bt, st = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
    for tag, fl in modes.items():
        f = lambda fl=fl: ~nms2d(bt, st, IouType.RBOX, SupressionType.HARD, 0.5, 0.0, 0.0, flags=fl)     # = box2d_nms, per-call flags
        dt = timed(f, 5, 1)
        prof = kernel_profile(f, 3)
        print("%d objects x %d boxes [%s]: %.2f ms, kept %d" % (nobj, per, tag, dt / 5 * 1e3, int(f().sum())),
              {k: round(v["total_ms"] / 3 * 1e3, 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:7]})
