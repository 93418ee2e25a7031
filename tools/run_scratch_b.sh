cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_conv_bwd.py tests/test_gpu_bn_fused.py tests/test_gpu_blocks_bwd.py tests/test_gpu_full_bwd.py -x -q 2>&1 | tail -4
TOP=60 python tools/bench_train.py 32 7 2>&1 | grep -E "wgrad|library calls|train step" | head -30
python tools/bench_train_graphed.py 2>&1 | tail -1
BACKBONE=hrnet_w48 NSTEP=4 python tools/bench_train_graphed.py 32 2>&1 | tail -1
