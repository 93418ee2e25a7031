#!/bin/bash
# usage (on the GPU box): tools/pgcn_sweep.sh <out.txt>   -- B-sweep of the rebuilt P-GCN kernel with counted HBM bytes
out=${1:-gpurun_out/r03_pgcn_sweep.txt}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > $out
for B in 64 256 1024 4096; do
  d=/tmp/pgcn_$B; rm -rf $d; mkdir -p $d
  cmd="python $R/tools/pgcn_sweep.py $B 20"
  ( cd /tmp && rocprofv3 --kernel-trace -d $d/t -o r -- $cmd > $d/t.log 2>&1 )
  ( cd /tmp && rocprofv3 --pmc FETCH_SIZE -d $d/f -o r -- $cmd > $d/f.log 2>&1 )
  ( cd /tmp && rocprofv3 --pmc WRITE_SIZE -d $d/w -o r -- $cmd > $d/w.log 2>&1 )
  echo "== B=$B  ($(grep algorithmic $d/t.log))" >> $out
  python $R/tools/pmc_per_kernel.py $(find $d/t -name "*.db" | head -1) $(find $d/f -name "*.db" | head -1) $(find $d/w -name "*.db" | head -1) 2>&1 | grep -E "kernel|pgcn" >> $out
done
cat $out
