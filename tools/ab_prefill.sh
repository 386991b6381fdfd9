cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_boxloss.py -x -q 2>&1 | tail -5
python - <<'PY'
import time, numpy as np, torch, sys
sys.path.insert(0, ".")
from d3d_amd import synth
from d3d_amd.benchmarks import DetectionEvaluator
p, g = synth.boxes3d_eval(5000, 4, 2)
rng = np.random.default_rng(5)
gt9 = np.concatenate([rng.integers(1, 3, (len(g), 1)), np.zeros((len(g), 1)), g], 1).astype(np.float32)
dt9 = np.concatenate([np.repeat(gt9[:, :1], 4, axis=0), rng.random((len(p), 1)), p], 1).astype(np.float32)
for compat in (True, False):
    ev = DetectionEvaluator([1, 2], [0.7, 0.5], reference_compat=compat)
    r = ev.calc_stats(gt9, dt9)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): r = ev.calc_stats(gt9, dt9)
    torch.cuda.synchronize()
    print("calc_stats 20k x 5k compat=%s: %.1f ms  tp[1][:4]=%s" % (compat, (time.perf_counter() - t0) / 3 * 1e3, r.tp[1][:4]))
PY
