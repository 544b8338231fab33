# round 4, step C: incremental block totals in the G2P binning + swizzled P2G tile: parity, then A/B against k_rebin (bit 20)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/r04_c; mkdir -p $D
timeout 1200 python -m pytest tests -m gpu -q -x -k "binning_inside or steady_state_rebinning or determinism or golden or c1_configs0 or grid_update_inside or fast_translation or grid_grows or checkpoint_restart or reference_sand3 or random_scenes_match or long_near or plastic_pair or large_one_way or dense_blocks or visit_list or 2d_block or multi_substep" 2>&1 | tail -8 > $D/pytest.log
cat $D/pytest.log
for rep in 1 2; do
BITS="1048576" CFGS="c2 c3 c5" STEPS=60 bash tools/gpu_ab_debug.sh 2>&1 | tee -a $D/ab.log
done
ARGS="" bash tools/gpu_kstats.sh 2>&1 | tee $D/kstats_c2.log
