cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_box.py tests/test_gpu_boxloss.py tests/test_gpu_fullsize.py -x -q -k "iou or overflow or cfg3 or cfg4" 2>&1 | tail -3
python bench.py --iou-only 2>&1 | grep -v amdgpu | cut -c1-330
