cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -22 > gpurun_out/r06/suite_durations.txt; cat gpurun_out/r06/suite_durations.txt
