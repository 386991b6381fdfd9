cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for n in 2000000 8000000 500000 200000; do python tools/tune_ab.py 50 $n "wg512:1=1" 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06/ab13_wg512_sizes.txt
cat gpurun_out/r06/ab13_wg512_sizes.txt
