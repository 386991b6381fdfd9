"""the clip-bound IoU case (the reference's benchmark boxes: 28 % of the pairs overlap, 5 k x 5 k, fp64) a few times, for a
rocprofv3 --pmc pass over k_iou_clip (VALU activity; development aid):
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d out -- python3 tools/iou_dense_pmc.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from d3d_amd import synth
from d3d_amd.box import box2d_iou

b, _ = synth.boxes2d_dense(5000, 2)
bt = torch.from_numpy(b).cuda()
for _ in range(5):
    box2d_iou(bt, bt, method="rbox")
torch.cuda.synchronize()
