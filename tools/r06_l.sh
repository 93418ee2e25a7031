export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${TAG:-r06_l}
mkdir -p $out
cd $R
common="--steps 100 --repeats 3 --no-cpu-baseline --no-fp32-mode --no-train --no-proj-feat-variant --no-power --no-time-table-pass --no-config5 --no-ceiling-probe --no-other-half --no-pgcn --detail-out /tmp/d.json"
for ov in 0 1; do for q in 4 8; do
  DIR_OVERLAP=$ov GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --inflight 1 $common 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('DIR_OVERLAP=$ov queues=$q inflight 1: %.3f ms per forward' % d['ms_per_step'])" >> $out/overlap.txt
done; done
DIR_OVERLAP=1 timeout 300 python bench.py --inflight 4 $common 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('DIR_OVERLAP=1 inflight 4: %.3f ms per step, one in flight %s' % (d['ms_per_step'], d['config']['ms_per_forward_one_in_flight']))" >> $out/overlap.txt
