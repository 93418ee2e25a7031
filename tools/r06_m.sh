export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${TAG:-r06_m}
mkdir -p $out
cd $R
timeout 600 python -m pytest tests/test_gpu_conv_as.py -m gpu -q 2>&1 | tail -15 > $out/tests_as.txt
