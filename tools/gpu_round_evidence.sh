# Everything committed under profiles/rNN_* in one call (ROUND=rNN): rocprofv3 kernel stats + PMC passes + three bench runs
# (gpu_round_profile.sh), the -m gpu suite with its parity margins (gpu_tests_all.sh), the sharded substep with one rank as its
# own neighbours over RCCL, un-profiled and under rocprofv3 (1 M slab and the 2 M slab of the 16 M configuration).
R=${ROUND:-r04}
ROUND=$R bash tools/gpu_round_profile.sh
cd $GRAFT_REPO_ROOT
TAG=${R}_tests TMO=2400 PYARGS="--deselect tests/test_multi_gpu.py" bash tools/gpu_tests_all.sh | tail -3
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/${R}_sharded
for cfg in c2 c5; do
  timeout 400 python tools/gpu_native_host_cost.py $cfg 2>&1 | grep substeps | tee gpurun_out/${R}_sharded/cost_$cfg.log
  CFG=$cfg bash tools/gpu_native_profile.sh > gpurun_out/${R}_sharded/kstats_$cfg.log 2>&1
  cp $(find gpurun_out/nprof -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_sharded/kernel_stats_$cfg.csv
done
