export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${TAG:-r06_s}
mkdir -p $out
cd $R
MIN_SAVING=0.05 timeout 1500 python tools/energy_tune.py > $out/energy_tune_ms05.txt 2>&1
cp gpurun_out/tuning/gfx950_bf16_b64_throughput.json $out/table_ms05.json
timeout 600 python tools/ab_tables.py dir_amd/tuning/gfx950_bf16_b64_throughput.json tools/tmp_tables/ms08.json $out/table_ms05.json 2>&1 | grep -v amdgpu | tail -5 > $out/ab.txt
