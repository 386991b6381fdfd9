cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_p1
python bench.py > gpurun_out/r06_p1/bench.json 2> gpurun_out/r06_p1/bench.err; tail -3 gpurun_out/r06_p1/bench.err; head -c 600 gpurun_out/r06_p1/bench.json
timeout 600 python -m pytest tests/test_gpu_boxloss.py -x -q 2>&1 | tail -2
