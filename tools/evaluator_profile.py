"""DetectionEvaluator.calc_stats on config 4 (20 k detections x 5 k ground truths, 40 score thresholds): ms per call and the
association's kernels (HIP events).  usage: python tools/evaluator_profile.py"""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from d3d_amd import _lib, synth  # noqa: E402
from d3d_amd.benchmarks import DetectionEvaluator  # noqa: E402

p, g = synth.boxes3d_eval(5000, 4, 2)
rng = np.random.default_rng(5)
gt9 = np.concatenate([rng.integers(1, 3, (len(g), 1)), np.zeros((len(g), 1)), g], 1).astype(np.float32)
dt9 = np.concatenate([np.repeat(gt9[:, :1], 4, axis=0), rng.random((len(p), 1)), p], 1).astype(np.float32)
ev = DetectionEvaluator([1, 2], [0.7, 0.5])
for _ in range(3):
    ev.calc_stats(gt9, dt9)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    ev.calc_stats(gt9, dt9)
torch.cuda.synchronize()
print("calc_stats 20 k x 5 k: %.2f ms per call" % ((time.perf_counter() - t0) / 10 * 1e3))
lib = _lib.load()
lib.d3d_profile_enable(1)
ev.calc_stats(gt9, dt9)
torch.cuda.synchronize()
lib.d3d_profile_enable(0)
buf = ctypes.create_string_buffer(1 << 16)
lib.d3d_profile_report(buf, len(buf))
print(" ".join("%s %.0f us" % (ln.rsplit(",", 2)[0], 1e3 * float(ln.rsplit(",", 2)[2])) for ln in buf.value.decode().strip().splitlines()))
