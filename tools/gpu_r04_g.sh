cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/r04_g; mkdir -p $D
for sc in stirred; do
LIBDIR=tools/tmp_prof LIB=ablate CMD="python tools/gpu_sort_prof.py $sc" bash tools/gpu_with_lib.sh 2>&1 | grep -v amdgpu.ids | tee $D/sort_prof_$sc.log
done
