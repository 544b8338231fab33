# developer utility: where does k_regroup's time go? (needs a -DWGS_ABLATE build in the tree)
export WGS_BENCH_ALLOW_ABLATE=1
for dbg in 0 1048576 2097152 3145728; do
  echo "== WGS_DEBUG=$dbg"; WGS_DEBUG=$dbg ARGS="--allow-debug-switches" bash tools/gpu_kstats.sh 2>&1 | grep -E "regroup|rebin" | cut -c1-140
done
