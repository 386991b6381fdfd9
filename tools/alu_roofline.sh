#!/bin/bash
# VALU counters + durations of the clip-bound IoU kernels: k_iou_clip<double> on the reference's benchmark
# boxes (5 k x 5 k, 28 % of the pairs overlap) and k_giou_main<double, 0> (GIoU's pair kernel, 10 k x 10 k; round 5: was k_loss_iou).  One rocprofv3 pass per counter
# (--kernel-trace + --pmc only).  usage (GPU box): bash tools/alu_roofline.sh <tag>  ->  gpurun_out/<tag>/alu_*.txt
tag=${1:-r05_alu}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for job in "k_iou_clip iou_dense_pmc.py" "k_giou_main iou_loss_pmc.py"; do
  set -- $job
  pat=$1; script=$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$pat -o t -- python3 $GRAFT_REPO_ROOT/tools/$script > /dev/null 2> $out/trace_$pat.err
  for c in SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${pat}_$c -o pmc -- python3 $GRAFT_REPO_ROOT/tools/$script > /dev/null 2> $out/pmc_${pat}_$c.err
  done
done
python3 $GRAFT_REPO_ROOT/tools/alu_roofline_summary.py $out > $out/alu_summary.txt 2>&1
cat $out/alu_summary.txt
