cd $GRAFT_REPO_ROOT
python tools/riou_bwd_ab.py 2>&1 | grep -v amdgpu
