# same-lease A/B of two builds of the library (tools/_ubench/lib_old.so, lib_new.so) through bench.py, alternating
R=${1:-3}
for i in $(seq $R); do for v in old new; do
  cp tools/_ubench/lib_$v.so dir_amd/lib/libdir_hip.so
  python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fp32-mode --no-train --no-proj-feat-variant --no-time-table-pass 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); p=d['power'] or {}
print('%-4s %8.1f images/s  %.3f ms/step | one in flight %.3f | %s W %s J/step' % ('$v', d['value'], d['ms_per_step'], d['config']['ms_per_forward_one_in_flight'], p.get('socket_w'), p.get('joules_per_step')))"
done; done
cp tools/_ubench/lib_old.so dir_amd/lib/libdir_hip.so
