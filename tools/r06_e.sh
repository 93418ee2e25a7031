export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${TAG:-r06_e}
mkdir -p $out
cd $R
timeout 900 python tools/autotune_report.py > $out/autotune_report_f16.txt 2>&1
