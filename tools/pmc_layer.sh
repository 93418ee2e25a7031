#!/bin/bash
# usage: tools/pmc_layer.sh "<ONLY pattern>" <tag> [env assignments...]   (run on the GPU box via gpurun)
# collects SQ counters for one layer of tools/bench_layers.py in separate rocprofv3 --pmc passes
pat="$1"; tag="$2"; shift 2
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM" ; do
  i=$((i+1))
  ( cd /tmp && env "$@" ONLY="$pat" rocprofv3 --pmc $set -d $out/p$i -o r -- python $GRAFT_REPO_ROOT/tools/bench_layers.py > $out/p$i.log 2>&1 )
  db=$(find $out/p$i -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $db conv_ 2>&1 | tee -a $out/summary.txt
done
