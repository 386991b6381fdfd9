#!/usr/bin/env python3
"""d3d_argsort_desc on random sizes across its three size ranges and on adversarial key distributions, against numpy's stable
order (development aid, not collected by pytest): python tests/sort_fuzz.py [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from d3d_amd.box import argsort_desc


def expected(s):
    nan = np.isnan(s)
    return np.lexsort((np.arange(len(s)), -np.where(nan, 0, s), ~nan))


count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for it in range(count):
    rng = np.random.default_rng(it)
    n = int(rng.choice([rng.integers(1, 8192), rng.integers(8192, 131073), rng.integers(131073, 400000)], p=[0.2, 0.7, 0.1]))
    kind = it % 8
    if kind == 0: s = rng.random(n)
    elif kind == 1: s = rng.integers(0, 3, n).astype(np.float64)                 # three distinct values
    elif kind == 2: s = np.sort(rng.random(n))                                  # ascending
    elif kind == 3: s = -np.sort(rng.random(n))                                 # descending, negative
    elif kind == 4: s = np.tile(rng.random(max(n // 1024, 1) + 1), 1024)[:n]    # periodic with the sample stride
    elif kind == 5: s = rng.standard_cauchy(n)                                  # heavy tails
    elif kind == 6: s = np.where(rng.random(n) < 0.01, np.nan, rng.random(n)); s[rng.integers(0, n, 5)] = -0.0
    else: s = np.exp(-rng.random(n) * 700)                                      # 300 binades
    for dt in (np.float64, np.float32):
        x = s.astype(dt)
        got = argsort_desc(torch.from_numpy(x).cuda()).cpu().numpy()
        if not np.array_equal(got, expected(x)):
            bad += 1
            print("SORT it", it, "n", n, "kind", kind, dt.__name__, "FAILED", int(np.sum(got != expected(x))))
print("sort fuzz: %d cases, %d failures" % (count, bad))
