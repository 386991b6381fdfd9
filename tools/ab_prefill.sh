cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python - <<'PY' 2>&1 | grep -v amdgpu
import sys, torch
sys.path.insert(0, ".")
import bench
from d3d_amd import _lib, synth
from d3d_amd.box import box2d_iou
import os
for libname in ("libd3d_hip.so", "libd3d_hip_tune.so", "libd3d_hip.so", "libd3d_hip_tune.so"):
    pass
b = synth.boxes2d_dense(5000, 1)[0]
for dt in (torch.float64, torch.float32):
    t = torch.from_numpy(b).cuda().to(dt)
    for rep in range(3):
        d = bench.timed(lambda: box2d_iou(t, t, method="rbox", precise=False), 20, 3)
        prof = bench.kernel_profile(lambda: box2d_iou(t, t, method="rbox", precise=False), 20)
        print("dense5k %s %8.1f us | " % (str(dt)[-7:], 1e6 * d / 20) + " ".join("%s %.1f" % (k.replace("k_", ""), v["avg_us"]) for k, v in prof.items()), flush=True)
PY
timeout 600 python -m pytest tests/test_gpu_box.py -x -q -k "iou" 2>&1 | tail -2
