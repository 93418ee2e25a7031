export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${TAG:-r06_f}
mkdir -p $out
cd $R
timeout 1500 python tools/energy_tune.py > $out/energy_tune.txt 2>&1
