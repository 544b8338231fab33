# round 4, step A: binning inside the fused G2P — parity tests, then A/B against the k_rebin launch (WGS_DEBUG bit 20), then kernel stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/r04_a; mkdir -p $D
timeout 900 python -m pytest tests -m gpu -q -x -k "binning_inside or steady_state_rebinning or determinism or golden or c1_configs0 or grid_update_inside or fast_translation or grid_grows or checkpoint_restart_is_bit or reference_sand3" 2>&1 | tail -8 > $D/pytest.log
cat $D/pytest.log
for rep in 1 2; do
BITS=1048576 CFGS="c2 c3 c5" STEPS=60 bash tools/gpu_ab_debug.sh 2>&1 | tee -a $D/ab.log
done
ARGS="" bash tools/gpu_kstats.sh 2>&1 | tee $D/kstats_c2.log
