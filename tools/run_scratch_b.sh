cd $GRAFT_REPO_ROOT
for e in "X=1" "DIR_TRAIN_FUSE_BN=0" "IMG=256"; do echo "== $e"; env $e python tools/dbg_relu_mask.py 2>&1 | grep -v amdgpu | tail -4 | cut -c1-400; done
