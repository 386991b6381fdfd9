cd $GRAFT_REPO_ROOT
for red in mean max ""; do echo "== reduction '$red'"; TUNE_RED=$red python tools/tune_ab.py 200 1000000 2>&1 | grep -v amdgpu.ids | tail -2; done
