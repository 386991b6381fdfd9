import sys, torch
sys.path.insert(0, ".")
import bench
from d3d_amd import synth
from d3d_amd.box import box2d_nms
b, s = synth.boxes2d_sparse(100000, 1)
bt, st = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
f = lambda: box2d_nms(bt, st, iou_method="rbox", iou_threshold=0.5)
dt = bench.timed(f, 50, 5)
print("nms stream: %.1f us/call" % (dt / 50 * 1e6))
