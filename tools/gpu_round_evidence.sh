# Everything committed under profiles/rNN_* in one call (ROUND=rNN): rocprofv3 kernel stats + PMC passes + three bench runs of the headline
# (gpu_round_profile.sh), kernel stats of C3 / C4 / C5 / the stirred cube / the reference's sand3 and the byte + SQ counters of C3, C4, C5,
# sand3 and the stirred cube, the -m gpu suite with its parity margins (gpu_tests_all.sh), the sharded substep with one
# rank as its own neighbours over RCCL, un-profiled and under rocprofv3 (1 M slab and the 2 M slab of the 16 M configuration).
# Afterwards, here: tools/summarize_profiles.py rNN; tools/summarize_margins.py rNN_tests rNN; tools/summarize_extra.py rNN; tools/check_profiles.py
R=${ROUND:-r06}
ROUND=$R bash tools/gpu_round_profile.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${R}_extra
for cfg in c3 c4 c5; do CFG=$cfg bash tools/gpu_kstats_cfg.sh > gpurun_out/${R}_extra/kstats_$cfg.log 2>&1; cp gpurun_out/kstats_$cfg/kernel_stats.csv gpurun_out/${R}_extra/kernel_stats_$cfg.csv; done
for sc in stirred sand3; do SCENE=$sc bash tools/gpu_scene_kstats.sh > gpurun_out/${R}_extra/kstats_$sc.log 2>&1; cp $(find gpurun_out/kstats_scene -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_extra/kernel_stats_$sc.csv; done
for cfg in c3 c4 c5; do CFG=$cfg STEPS=12 bash tools/gpu_pmc_cfg.sh > gpurun_out/${R}_extra/pmc_$cfg.log 2>&1; cp gpurun_out/pmc_$cfg/summary.json gpurun_out/${R}_extra/pmc_summary_$cfg.json; done
for sc in sand3 stirred; do SCENE=$sc bash tools/gpu_scene_pmc.sh > gpurun_out/${R}_extra/pmc_$sc.log 2>&1; cp gpurun_out/pmc_scene_$sc/summary.json gpurun_out/${R}_extra/pmc_summary_$sc.json; done
TAG=${R}_tests TMO=2400 PYARGS="--deselect tests/test_multi_gpu.py" bash tools/gpu_tests_all.sh | tail -3
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/${R}_sharded
for cfg in c2 c5; do
  timeout 400 python tools/gpu_native_host_cost.py $cfg 2>&1 | grep substeps | tee gpurun_out/${R}_sharded/cost_$cfg.log
  CFG=$cfg bash tools/gpu_native_profile.sh > gpurun_out/${R}_sharded/kstats_$cfg.log 2>&1
  cp $(find gpurun_out/nprof -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_sharded/kernel_stats_$cfg.csv
done
