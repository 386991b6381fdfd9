"""d3d_argsort_desc above the sample sort's range (the radix path, round 4): time per call and per kernel for fp64 / fp32 keys.
usage (GPU box): python tools/sort_profile.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from d3d_amd.box import argsort_desc

rng = np.random.default_rng(0)
for name, keys in (("1M fp64 random", rng.random(1000000)), ("1M fp32 random", rng.random(1000000).astype(np.float32)),
                   ("1M fp64 promoted from fp32", rng.random(1000000).astype(np.float32).astype(np.float64)),
                   ("300k fp64 random", rng.random(300000)), ("4M fp32 random", rng.random(4000000).astype(np.float32))):
    t = torch.from_numpy(keys).cuda()
    dt = bench.timed(lambda: argsort_desc(t), 20, 3)
    prof = bench.kernel_profile(lambda: argsort_desc(t), 10)
    ref = bench.timed(lambda: torch.argsort(t, descending=True, stable=True), 20, 3)
    print("%-28s %7.1f us/call (torch.argsort stable: %7.1f) | " % (name, 1e6 * dt / 20, 1e6 * ref / 20) +
          " ".join("%s %.1fx%d" % (k.replace("k_rs_", ""), p["avg_us"], p["calls"] // 10) for k, p in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])), flush=True)
