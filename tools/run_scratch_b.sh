cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_hrnet_train.py tests/test_gpu_bn_fused.py tests/test_gpu_full_bwd.py -x -q 2>&1 | tail -4
BACKBONE=hrnet_w48 NSTEP=4 python tools/bench_train_graphed.py 32 2>&1 | tail -1
DIR_TRAIN_FUSE_RELU_BWD=0 BACKBONE=hrnet_w48 NSTEP=4 python tools/bench_train_graphed.py 32 2>&1 | tail -1
