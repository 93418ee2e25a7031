#!/bin/bash
# usage (on the GPU box via gpurun): tools/pmc_wgrad.sh "<ONLY pattern>" <tag>  -- SQ counters of conv_wgrad_x3_kernel on one shape of tools/bench_wgrad.py
pat="$1"; tag="$2"
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_SMEM" ; do
  i=$((i+1))
  ( cd /tmp && ONLY="$pat" rocprofv3 --pmc $set -d $out/p$i -o r -- python $GRAFT_REPO_ROOT/tools/bench_wgrad.py 5 > $out/p$i.log 2>&1 )
  db=$(find $out/p$i -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $db conv_wgrad_x3 2>&1 | tee -a $out/summary.txt
done
rm -rf $out/p1 $out/p2 $out/p3
