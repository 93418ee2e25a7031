set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${TAG:-r05_a}
mkdir -p $out
cd $R
t0=$SECONDS
python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $out/gpu_tests.txt
echo "pytest -m gpu: $((SECONDS - t0)) s" > $out/durations.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1
echo "smoke(): rc $? in $((SECONDS - t0)) s (incl. pytest)" >> $out/durations.txt
t0=$SECONDS
python bench.py > $out/bench_stdout.txt 2> $out/bench_stderr.txt
echo "python bench.py (default flags): $((SECONDS - t0)) s" >> $out/durations.txt
tail -1 $out/bench_stdout.txt > $out/bench_line.txt
cp bench_detail.json $out/bench_detail.json
TOP=180 python tools/bench_train.py 32 7 2>&1 | grep -v amdgpu.ids > $out/train_step_library_calls.txt
( cd /tmp && rocprofv3 --kernel-trace -d $out/trace -o r -- python $R/tools/bench_train.py 32 9 > $out/trace.log 2>&1 )
python tools/prof_summary.py $(find $out/trace -name "*.db" | head -1) 60 > $out/train_step_b32_kernel_stats.txt
rm -rf $out/trace
