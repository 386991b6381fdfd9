cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python tools/iou3d_ab.py 50 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/iou3d_now.txt; grep rbox gpurun_out/r06/iou3d_now.txt | tail -12
timeout 900 python -m pytest tests/test_gpu_box.py tests/test_gpu_boxloss.py -x -q 2>&1 | tail -3
