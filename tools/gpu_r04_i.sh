cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/r04_i; mkdir -p $D
for sc in sand3 stirred; do SCENE=$sc bash tools/gpu_scene_kstats.sh 2>&1 | grep -v amdgpu.ids | tee $D/kstats_$sc.log; done
