export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof4
mkdir -p $out
rm -f /tmp/dir_autotune.json
python $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-fp32-mode --no-train --no-proj-feat-variant --no-power --no-time-table-pass --force-table --no-config5 --no-ceiling-probe --no-other-half --no-pgcn --autotune-cache /tmp/dir_autotune.json > $out/tune.log 2>&1
cmd="python $R/bench.py --inflight 4 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-fp32-mode --no-train --no-proj-feat-variant --no-power --no-time-table-pass --force-table --no-config5 --no-ceiling-probe --no-other-half --no-pgcn --autotune-cache /tmp/dir_autotune.json"
( cd /tmp && rocprofv3 --kernel-trace --stats -d $out/trace -o r -- $cmd > $out/trace.log 2>&1 )
cd $R
python tools/prof_summary.py $(find $out/trace -name "*.db" | head -1) 120 > $out/kernel_stats.txt
tail -1 $out/trace.log | cut -c1-300
head -30 $out/kernel_stats.txt | cut -c1-160
rm -rf $out/trace
