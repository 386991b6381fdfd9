"""rows per workgroup of k_iou_pre (d3d_debug_set_pre_rows; 0 = the library's choice) on the rotated 2D IoU: the reference's
benchmark boxes (5 k x 5 k, 28 % overlap) and config 3's density at several sizes, fp64 and fp32.  usage: python tools/pre_rows_ab.py"""
import sys
import torch
sys.path.insert(0, ".")
import bench
from d3d_amd import _lib, synth
from d3d_amd.box import box2d_iou

lib = _lib.load()
cases = [("dense5k", synth.boxes2d_dense(5000, 1)[0]), ("dense2k", synth.boxes2d_dense(2000, 1)[0]),
         ("sparse10k", synth.boxes2d_sparse(10000, 1)[0]), ("sparse30k", synth.boxes2d_sparse(30000, 1)[0])]
for name, b in cases:
    for dt in (torch.float64, torch.float32):
        t = torch.from_numpy(b).cuda().to(dt)
        ref = None
        for rep in range(2):
            for rows in (0, 8, 16, 32, 64):
                lib.d3d_debug_set_pre_rows(rows)
                out = box2d_iou(t, t, method="rbox", precise=False)
                ref = out.clone() if ref is None else ref
                d = bench.timed(lambda: box2d_iou(t, t, method="rbox", precise=False), 20, 3)
                prof = bench.kernel_profile(lambda: box2d_iou(t, t, method="rbox", precise=False), 20)
                print("%-10s %s rows %2d %s %8.1f us | " % (name, str(dt)[-7:], rows, "same" if torch.equal(out, ref) else "DIFF", 1e6 * d / 20) +
                      " ".join("%s %.1f" % (k.replace("k_", ""), v["avg_us"]) for k, v in prof.items()), flush=True)
lib.d3d_debug_set_pre_rows(0)
