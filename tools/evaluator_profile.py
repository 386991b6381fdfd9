"""cProfile of DetectionEvaluator.calc_stats on config 4 (20 k detections x 5 k ground truths, 40 score thresholds) in the default
mode (the reference's association per threshold): where the host spends the call.  usage: python tools/evaluator_profile.py"""
import cProfile
import pstats
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from d3d_amd import synth
from d3d_amd.benchmarks import DetectionEvaluator

p, g = synth.boxes3d_eval(5000, 4, 2)
rng = np.random.default_rng(5)
gt9 = np.concatenate([rng.integers(1, 3, (len(g), 1)), np.zeros((len(g), 1)), g], 1).astype(np.float32)
dt9 = np.concatenate([np.repeat(gt9[:, :1], 4, axis=0), rng.random((len(p), 1)), p], 1).astype(np.float32)
ev = DetectionEvaluator([1, 2], [0.7, 0.5])
ev.calc_stats(gt9, dt9)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    r = ev.calc_stats(gt9, dt9)
torch.cuda.synchronize()
print("calc_stats: %.1f ms per call" % (1e3 * (time.perf_counter() - t0) / 3))
pr = cProfile.Profile()
pr.enable()
ev.calc_stats(gt9, dt9)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
