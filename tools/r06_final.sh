set -x
export TMPDIR=/tmp
export DIR_HEAD=597df1b
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r06_d_final
mkdir -p $out
cd $R
t0=$SECONDS
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $out/gpu_tests.txt
echo "pytest -m gpu: $((SECONDS - t0)) s" > $out/durations.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1
echo "smoke rc $?" >> $out/durations.txt
t0=$SECONDS
python bench.py --detail-out $out/bench_detail.json > $out/bench_stdout.txt 2> $out/bench_stderr.txt
echo "bench default: $((SECONDS - t0)) s" >> $out/durations.txt
tail -1 $out/bench_stdout.txt > $out/bench_line.txt
t0=$SECONDS
python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $out/bench_detail_driver_cmd.json > $out/bench_driver_cmd.txt 2> /dev/null
echo "bench driver cmd: $((SECONDS - t0)) s" >> $out/durations.txt
bash tools/profile_round.sh r06_d_prof
bash tools/profile_four_in_flight.sh
bash tools/pmc_fwd_sq.sh r06_d_sq conv_as_kernel
