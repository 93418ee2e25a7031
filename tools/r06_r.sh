export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${TAG:-r06_r}
mkdir -p $out
cd $R
MIN_SAVING=0.08 timeout 1500 python tools/energy_tune.py > $out/energy_tune_ms08.txt 2>&1
cp gpurun_out/tuning/gfx950_bf16_b64_throughput.json $out/table_ms08.json
timeout 600 python tools/ab_tables.py dir_amd/tuning/gfx950_bf16_b64_throughput.json $out/table_ms08.json 2>&1 | grep -v amdgpu | tail -4 > $out/ab.txt
