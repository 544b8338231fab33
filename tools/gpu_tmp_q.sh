cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do CFGS="c2 c3 c5" bash tools/gpu_variants_multi.sh 2>&1 | grep -v amdgpu; done
