cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python bench.py > gpurun_out/r06/bench_a.json 2> gpurun_out/r06/bench_a.err; tail -3 gpurun_out/r06/bench_a.err; head -c 2500 gpurun_out/r06/bench_a.json; echo
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
