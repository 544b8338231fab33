# usage: VARS="name:ENV=val,ENV2=val ..." ABLK=regex bash tools/gpu_ablate2.sh   (developer utility)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in ${VARS:-base:}; do
  name=${spec%%:*}; envs=${spec#*:}
  rm -rf gpurun_out/abl_$name; mkdir -p gpurun_out/abl_$name
  ( IFS=,; for e in $envs; do [ -n "$e" ] && export "$e"; done
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$name -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $BENCHARGS > gpurun_out/abl_$name/bench.log 2>&1 )
  f=$(find gpurun_out/abl_$name -name "*kernel_stats.csv" | head -1); echo "== $name ($envs) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/abl_$name/bench.log)"; grep -E "${ABLK:-p2g}" $f | cut -d, -f1-4
done
