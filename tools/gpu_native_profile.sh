# developer utility: rocprofv3 kernel stats of tools/gpu_native_host_cost.py (sharded substeps without / with self-neighbours over RCCL)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -rf gpurun_out/nprof; mkdir -p gpurun_out/nprof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nprof -- python3 tools/gpu_native_host_cost.py ${CFG:-c2} > gpurun_out/nprof/log 2>&1
grep "substeps" gpurun_out/nprof/log
f=$(find gpurun_out/nprof -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:28]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']}%")
PY
