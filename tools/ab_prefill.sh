cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_voxel.py -x -q -k "sparse or filter or default or trim or descending" 2>&1 | tail -4
python tools/host_profile_sparse.py 2000 2>&1 | grep -v amdgpu.ids | head -30 > gpurun_out/r06/host_sparse2.txt; head -22 gpurun_out/r06/host_sparse2.txt
python bench.py --sparse-only --steps 200 --warmup 10 2>&1 | grep -v amdgpu
