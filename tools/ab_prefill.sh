cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for leg in iou3d nms iou; do python bench.py --$leg-only --steps 5 --warmup 1 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06/legs.txt
cat gpurun_out/r06/legs.txt
