set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${TAG:-r06_b}
mkdir -p $out
cd $R
timeout 300 python tools/probe_sweep.py > $out/probe_sweep.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_dir.py tests/test_gpu_soak.py -m gpu -q -s -k "full_size or odd_batch or forward_pipeline or shipped or four_forwards or foreign" 2>&1 | grep -v "^$" | grep "rows\|passed\|failed\|FAILED\|Error" | cut -c1-250 > $out/tests_f16.txt
