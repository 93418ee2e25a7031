#!/bin/bash
# usage (on the GPU box, via gpurun): tools/profile_round.sh <tag>      e.g. r01_f
# (one forward in flight: uncontended kernel durations, as in the live roofline pass of bench.py)
# kernel trace + the two PMC passes (FETCH_SIZE, WRITE_SIZE) of the bench command; results under gpurun_out/<tag>/
tag=$1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
# the per-layer kernel choice is made once OUTSIDE the profiler and re-used, so the traces hold only the timed configuration
rm -f /tmp/dir_autotune.json
python $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-fp32-mode --no-train --no-proj-feat-variant --no-power --no-time-table-pass --force-table --no-config5 --no-ceiling-probe --no-other-half --no-pgcn --autotune-cache /tmp/dir_autotune.json > $out/tune.log 2>&1
cmd="python $R/bench.py --inflight 1 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-fp32-mode --no-train --no-proj-feat-variant --no-power --no-time-table-pass --force-table --no-config5 --no-ceiling-probe --no-other-half --no-pgcn --dump-conv --autotune-cache /tmp/dir_autotune.json"
( cd /tmp && rocprofv3 --kernel-trace --stats -d $out/trace -o r -- $cmd > $out/trace.log 2>&1 )
# counter passes: the same kernels launched eagerly (--no-graph) -- under --pmc the HIP-graph RNG bookkeeping kernel of torch.cuda.graph
# segfaulted inside the profiler on this stack (r02); counters are per dispatch, so graph or no graph makes no difference to them
pcmd="python $R/bench.py --no-graph --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-fp32-mode --no-train --no-proj-feat-variant --no-power --no-time-table-pass --force-table --no-config5 --no-ceiling-probe --no-other-half --no-pgcn --autotune-cache /tmp/dir_autotune.json"
( cd /tmp && rocprofv3 --pmc FETCH_SIZE -d $out/pmcF -o r -- $pcmd > $out/pmcF.log 2>&1 )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE -d $out/pmcW -o r -- $pcmd > $out/pmcW.log 2>&1 )
( cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $out/pmcS -o r -- $pcmd > $out/pmcS.log 2>&1 )
cd $R
python tools/prof_summary.py $(find $out/trace -name "*.db" | head -1) 60 > $out/kernel_stats.txt
python tools/pmc_traffic.py $(find $out/pmcF -name "*.db" | head -1) $(find $out/pmcW -name "*.db" | head -1) $out/pmc_traffic.json
python tools/pmc_per_kernel.py $(find $out/trace -name "*.db" | head -1) $(find $out/pmcF -name "*.db" | head -1) $(find $out/pmcW -name "*.db" | head -1) $(find $out/pmcS -name "*.db" | head -1) > $out/per_kernel.txt 2>&1
grep -h "_kernel \|^{" $out/trace.log > $out/bench_line.txt
rm -rf $out/trace $out/pmcF $out/pmcW $out/pmcS
