#!/bin/bash
# usage (via gpurun): tools/ab_lib.sh [rounds]   -- same-box A/B of tools/_ubench/lib_old.so vs lib_new.so on bench.py
R=${1:-2}
for i in $(seq $R); do for v in old new; do
  cp tools/_ubench/lib_$v.so dir_amd/lib/libdir_hip.so
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --autotune-cache /tmp/at_$v.json 2>/dev/null | tail -1 | python -c "import sys,json; print('$v', json.loads(sys.stdin.readlines()[-1])['ms_per_step'])"
done; done
