"""GIoU / DIoU forward on 10 k x 10 k fp64 boxes of config 3 a few times, for rocprofv3 --pmc passes over k_loss_iou (development aid;
see tools/alu_roofline.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from d3d_amd import synth
from d3d_amd.box import box2d_iou

b, _ = synth.boxes2d_sparse(100000, 1)
bt = torch.from_numpy(b[:10000]).cuda()
method = sys.argv[1] if len(sys.argv) > 1 else "grbox"
for _ in range(5):
    box2d_iou(bt, bt, method=method)
torch.cuda.synchronize()
