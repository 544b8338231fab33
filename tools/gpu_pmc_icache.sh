# developer utility: instruction-cache counters of a bench configuration (ARGS), per kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_ic; mkdir -p gpurun_out/pmc_ic
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_ic -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-extra $ARGS > gpurun_out/pmc_ic/bench.log 2>&1
f=$(find gpurun_out/pmc_ic -name "*counter_collection.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'][:70]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    if r['Counter_Name']=='SQ_WAVES': n[k]+=1
for k,v in acc.items():
    c=max(n[k],1)
    print(f"{k:70s} launches {c:4d} " + " ".join(f"{a}={b/c:.3g}" for a,b in sorted(v.items())))
PY
