cd $GRAFT_REPO_ROOT
python tools/tune_ab.py 200 1000000 "unfused:12=0" 2>&1 | grep -v amdgpu.ids | tail -6
for n in 750000 1300000 2000000 100000; do python tools/tune_ab.py 50 $n "unfused:12=0" 2>&1 | grep -v amdgpu.ids | tail -4; done
TUNE_DIST=uniform python tools/tune_ab.py 100 1000000 "unfused:12=0" 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python -m pytest tests/test_gpu_voxel.py -x -q 2>&1 | tail -3
