cd $GRAFT_REPO_ROOT
X="--steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-fp32-mode --no-train --no-proj-feat-variant --no-power --no-config5 --no-ceiling-probe --no-other-half --no-pgcn --tuning time"
for ov in 0 1 0 1; do
  echo "=== DIR_OVERLAP=$ov inflight 1"
  DIR_OVERLAP=$ov python bench.py --inflight 1 $X 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('ms_per_forward_one_in_flight'))"
done
for ov in 0 1; do
  echo "=== DIR_OVERLAP=$ov inflight 2"
  DIR_OVERLAP=$ov python bench.py --inflight 2 $X 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
