cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_cabi.py -x -q 2>&1 | grep -v "^$" | tail -30
