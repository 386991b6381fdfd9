cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
( timeout 900 python tests/fuzz.py 600000 400 2>&1 | grep -v amdgpu.ids | tail -5
  timeout 400 python tests/sharded_fuzz.py 7000 80 2>&1 | grep -v amdgpu.ids | tail -3
  timeout 300 python tests/owner_merge_fuzz.py 10000 100 2>&1 | grep -v amdgpu.ids | tail -3
  timeout 300 python tests/softnms_fuzz.py 8000 150 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/r06/fuzz_final.txt 2>&1
cat gpurun_out/r06/fuzz_final.txt
