export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${TAG:-r06_o}
mkdir -p $out
cd $R
timeout 900 python bench.py --no-train --no-config5 --no-cpu-baseline --no-fp32-mode --detail-out $out/bench_detail.json > $out/bench.txt 2> $out/bench.err
tail -1 $out/bench.txt > $out/bench_line.txt
timeout 900 python tools/autotune_report.py > $out/autotune_report_f16.txt 2>&1
