# round 6, first GPU call: the changed / new tests, then a short bench (parity record, rebuilt HBM probe, trained-like vs plain weights)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${TAG:-r06_a}
mkdir -p $out
cd $R
t0=$SECONDS
timeout 900 python -m pytest tests/test_gpu_loss.py tests/test_gpu_jpeg.py tests/test_gpu_fromdisk.py -m gpu -q -x 2>&1 | tail -8 > $out/tests_small.txt
echo "small tests: $((SECONDS - t0)) s" > $out/durations.txt
t0=$SECONDS
timeout 1500 python -m pytest tests/test_gpu_dir.py tests/test_gpu_soak.py -m gpu -q -x -s -k "full_size or odd_batch or forward_pipeline or shipped or four_forwards or foreign" 2>&1 | grep -v "^$" | tail -60 > $out/tests_f16.txt
echo "f16 headline tests: $((SECONDS - t0)) s" >> $out/durations.txt
t0=$SECONDS
timeout 900 python bench.py --steps 100 --no-train --no-config5 --no-cpu-baseline --detail-out $out/bench_detail_cond.json > $out/bench_cond.txt 2> $out/bench_cond.err
echo "bench cond: $((SECONDS - t0)) s" >> $out/durations.txt
t0=$SECONDS
timeout 600 python bench.py --steps 100 --weights plain --no-train --no-config5 --no-cpu-baseline --no-fp32-mode --no-pgcn --detail-out $out/bench_detail_plain.json > $out/bench_plain.txt 2> $out/bench_plain.err
echo "bench plain: $((SECONDS - t0)) s" >> $out/durations.txt
tail -1 $out/bench_cond.txt > $out/bench_line_cond.txt
tail -1 $out/bench_plain.txt > $out/bench_line_plain.txt
