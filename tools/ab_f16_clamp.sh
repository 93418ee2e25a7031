#!/bin/bash
# usage (via gpurun): tools/ab_f16_clamp.sh [rounds]  -- same-box A/B of the f16-storage mode with / without the store clamp (lib/libdir_hip_noclamp.so:
# DIR_BUILD_TAG=noclamp DIR_HIPCC_EXTRA=-DDIR_F16_NOCLAMP=1 python -m dir_amd.build) and of the bf16 mode, alternating
R=${1:-2}
common="--steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-fp32-mode --no-train --no-proj-feat-variant --no-config5 --no-power --no-other-half --no-ceiling-probe --no-time-table-pass"
for i in $(seq $R); do
  for v in "bf16 clamp" "f16 clamp" "f16 noclamp"; do
    set -- $v
    lib=""; [ "$2" = noclamp ] && lib="$GRAFT_REPO_ROOT/dir_amd/lib/libdir_hip_noclamp.so"
    DIR_LIB_PATH=$lib python bench.py --dtype $1 $common --autotune-cache /tmp/at_$1.json 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1 $2', d['ms_per_step'], d['value'], d['config']['ms_per_forward_one_in_flight'])"
  done
done
