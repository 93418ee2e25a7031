set -x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r05_e
mkdir -p $out
python -m pytest tests/test_gpu_bn_fused.py tests/test_gpu_blocks_bwd.py tests/test_gpu_full_bwd.py -x -q 2>&1 | tail -5
TOP=200 python tools/bench_train.py 32 7 2>&1 | grep -v amdgpu.ids > $out/train_calls_full.txt
DIR_TRAIN_STATS_IN_EPILOGUE=0 TOP=200 python tools/bench_train.py 32 7 2>&1 | grep -v amdgpu.ids > $out/train_calls_nostats.txt
python tools/bench_train_graphed.py 2>&1 | tail -1
DIR_TRAIN_STATS_IN_EPILOGUE=0 python tools/bench_train_graphed.py 2>&1 | tail -1
DIR_TRAIN_FUSE_BN=0 python tools/bench_train_graphed.py 2>&1 | tail -1
python tools/bench_train_graphed.py 2>&1 | tail -1
( cd /tmp && AUTOTUNE=0 rocprofv3 --kernel-trace -d $out/trace5 -o r -- python $GRAFT_REPO_ROOT/tools/profile_config5.py 32 > $out/trace5.log 2>&1 )
python tools/prof_summary.py $(find $out/trace5 -name "*.db" | head -1) 60 > $out/config5_kernel_stats_noautotune.txt
rm -rf $out/trace5
