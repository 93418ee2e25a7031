cd $GRAFT_REPO_ROOT
for e in "DIR_TRAIN_STATS_IN_EPILOGUE=0" "DIR_TRAIN_FUSE_BN=0" "X=1"; do
  echo "=== $e"
  env $e python -m pytest "tests/test_gpu_hrnet_train.py::test_hrnet_module_trains_like_torch_autograd" -q -s 2>&1 | grep -E "HRNet-W48 training|passed|failed|forward c1"
done
