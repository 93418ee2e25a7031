set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${TAG:-r06_c}
mkdir -p $out
cd $R
timeout 600 python -m pytest tests/test_gpu_conv_as.py -m gpu -q -x 2>&1 | tail -30 > $out/tests_as.txt
timeout 600 python tools/bench_as.py f16 > $out/bench_as_f16.txt 2>&1
