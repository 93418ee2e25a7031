export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${TAG:-r06_k}
mkdir -p $out
cd $R
timeout 600 python -m pytest tests/test_gpu_conv_as.py -m gpu -q 2>&1 | tail -3 > $out/tests_as.txt
DIR_STAMPS=conv_as timeout 300 python tools/stamps_as.py 2>&1 | grep -v amdgpu > $out/stamps.txt
timeout 600 python tools/bench_as.py f16 > $out/bench_as_f16.txt 2>&1
