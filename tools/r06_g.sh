set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${TAG:-r06_g}
mkdir -p $out
cd $R
t0=$SECONDS
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $out/gpu_tests.txt
echo "pytest -m gpu: $((SECONDS - t0)) s" > $out/durations.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1
echo "smoke rc $?" >> $out/durations.txt
t0=$SECONDS
python bench.py --detail-out $out/bench_detail.json > $out/bench_stdout.txt 2> $out/bench_stderr.txt
echo "bench: $((SECONDS - t0)) s" >> $out/durations.txt
tail -1 $out/bench_stdout.txt > $out/bench_line.txt
