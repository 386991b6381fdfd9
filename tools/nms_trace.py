"""box2d_nms on cfg3, for rocprofv3 (development aid; no per-launch events).  Run it with the interpreter named after `--`
(a script with an env shebang would be an exec after the profiler has initialised the GPU, which this pool forbids):

    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d out -- python3 $GRAFT_REPO_ROOT/tools/nms_trace.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from d3d_amd import synth
from d3d_amd.box import box2d_nms
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
b, s = synth.boxes2d_sparse(n, 1)
bt, st = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
for _ in range(20):
    box2d_nms(bt, st, iou_method="rbox", iou_threshold=0.5)
torch.cuda.synchronize()
