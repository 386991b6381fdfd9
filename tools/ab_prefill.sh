cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for i in 1 2; do
python tools/iou3d_ab.py 50 2>&1 | grep -v amdgpu.ids | grep rbox
python tools/iou3d_ab.py 50 libd3d_hip_tune.so 2>&1 | grep -v amdgpu.ids | grep rbox | sed 's/library/OLD_lib/'
done > gpurun_out/r06/iou3d_union.txt; cat gpurun_out/r06/iou3d_union.txt
python bench.py --iou-only 2>&1 | grep -v amdgpu | cut -c1-400
