cd $GRAFT_REPO_ROOT
python tools/lib_ab.py libd3d_hip.so libd3d_hip_tune.so 1000000 8000000 2000000 750000 2>&1 | grep -v amdgpu
