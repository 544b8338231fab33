cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for dbg in ${ABL:-0}; do
rm -rf gpurun_out/abl$dbg; mkdir -p gpurun_out/abl$dbg
WGS_DEBUG=$dbg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl$dbg -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/abl$dbg/bench.log 2>&1
f=$(find gpurun_out/abl$dbg -name "*kernel_stats.csv" | head -1); echo "dbg=$dbg"; grep -E "${ABLK:-canon}" $f | cut -d, -f1-4
done
