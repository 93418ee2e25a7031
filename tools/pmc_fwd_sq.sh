#!/bin/bash
# usage (on the GPU box via gpurun): tools/pmc_fwd_sq.sh <tag> [kernel patterns...]
# SQ wait / issue / LDS / MFMA counters of the forward convolution kernels inside the bench forward (eager launches of the timed
# configuration, one forward in flight), as tools/pmc_wgrad.sh does for the weight-gradient kernel.  Result: gpurun_out/<tag>/sq_counters.txt
tag=$1; shift
pats="${@:-conv_patch_kernel conv_big_kernel stream1x1_kernel conv_pipe_kernel conv_igemm_kernel tail_chain_kernel bneck_chain_kernel conv_as_kernel}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
rm -f /tmp/dir_autotune.json
common="--no-cpu-baseline --no-fp32-mode --no-train --no-proj-feat-variant --no-power --no-time-table-pass --force-table --no-config5 --no-ceiling-probe --autotune-cache /tmp/dir_autotune.json"
python $R/bench.py --steps 2 --warmup 1 --repeats 1 $common > $out/tune.log 2>&1
pcmd="python $R/bench.py --no-graph --steps 4 --warmup 2 --repeats 1 $common"
i=0
: > $out/sq_counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_SMEM" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_ANY SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $set -d $out/p$i -o r -- $pcmd > $out/p$i.log 2>&1 )
  db=$(find $out/p$i -name "*.db" | head -1)
  if [ -n "$db" ]; then
    for p in $pats; do python $R/tools/pmc_summary.py $db $p >> $out/sq_counters.txt 2>&1; done
  else
    echo "pass $i: no database (see p$i.log)" >> $out/sq_counters.txt; tail -5 $out/p$i.log >> $out/sq_counters.txt
  fi
  rm -rf $out/p$i
done
wc -l $out/sq_counters.txt
