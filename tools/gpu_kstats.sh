# rocprofv3 kernel stats of a short bench run (developer utility): ARGS = bench arguments
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kstats; mkdir -p gpurun_out/kstats
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra $ARGS > gpurun_out/kstats/bench.log 2>&1
f=$(find gpurun_out/kstats -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']}%")
PY
