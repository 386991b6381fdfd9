cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python tools/iou3d_ab.py 50 2>&1 | grep -v amdgpu.ids | grep rbox > gpurun_out/r06/iou3d_roles.txt; cat gpurun_out/r06/iou3d_roles.txt
