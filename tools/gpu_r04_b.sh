# round 4, step B: attribution of the binning's cost inside the fused G2P (ablation variants), A/B against k_rebin (bit 20), unpaired G2P (bit 4096)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/r04_b; mkdir -p $D
timeout 600 python -m pytest tests -m gpu -q -x -k "binning_inside" 2>&1 | tail -3 > $D/pytest.log
cat $D/pytest.log
for rep in 1 2; do
CFGS="c2" bash tools/gpu_variants_multi.sh 2>&1 | tee -a $D/variants.log
BITS="1048576 4096 1052672" CFGS="c2" STEPS=60 bash tools/gpu_ab_debug.sh 2>&1 | tee -a $D/ab.log
done
BITS=1048576 CFGS="c3 c5" STEPS=60 bash tools/gpu_ab_debug.sh 2>&1 | tee -a $D/ab.log
