cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
( python tools/sharded_profile.py 8 1000000 config5 owner noreplicate; python tools/sharded_profile.py 8 1000000 config5 owner noreplicate 32; python tools/sharded_profile.py 8 1000000 config5 owner ) 2>&1 | grep -v amdgpu > gpurun_out/r06/sharded_w8.txt; cat gpurun_out/r06/sharded_w8.txt
