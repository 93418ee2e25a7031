cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_bn_fused.py tests/test_gpu_blocks_bwd.py tests/test_gpu_hrnet_train.py tests/test_gpu_full_bwd.py -x -q 2>&1 | tail -4
BACKBONE=hrnet_w48 TOP=3 python tools/bench_train.py 32 5 2>&1 | grep -v amdgpu.ids | head -4
TOP=3 python tools/bench_train.py 32 7 2>&1 | grep -v amdgpu.ids | head -4
