# developer utility: rocprofv3 kernel stats of one bench configuration (CFG = c2|c3|c5, ARGS = more bench arguments) -> gpurun_out/kstats_$CFG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/kstats_${TAG:-$CFG}; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --config ${CFG:-c2} --steps ${STEPS:-50} --warmup 10 --no-cpu-baseline --no-extra --no-live-pmc $ARGS > $O/bench.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f} us  {r['Percentage']}%")
PY
