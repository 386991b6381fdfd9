cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -k "roles_and_fillers" 2>&1 | tail -15
