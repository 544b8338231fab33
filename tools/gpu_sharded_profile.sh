# Developer utility: kernel profile of the sharded code path with one rank (no neighbours) under rocprofv3.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 WGS_BENCH_FORCE_SHARDED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
rm -rf gpurun_out/shp; mkdir -p gpurun_out/shp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/shp -- python3 bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/shp/bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/shp/bench.log
python3 tools/show_stats.py gpurun_out/shp | head -20
