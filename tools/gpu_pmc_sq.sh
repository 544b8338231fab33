# developer utility: SQ counters per kernel for a bench configuration (ARGS)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcsq; mkdir -p gpurun_out/pmcsq
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmcsq -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra $ARGS > gpurun_out/pmcsq/log 2>&1
f=$(find gpurun_out/pmcsq -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0][-44:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in acc.items():
    if "wgs" not in k: continue
    m={c:sum(v[len(v)//2:])/len(v[len(v)//2:]) for c,v in cs.items()}
    w=max(m.get("SQ_WAVES",1),1)
    print(f"{k:46s} waves {w:8.0f} valu/wave {m.get('SQ_INSTS_VALU',0)/w:7.0f} salu/wave {m.get('SQ_INSTS_SALU',0)/w:6.0f} wait {m.get('SQ_WAIT_ANY',0)/max(m.get('SQ_WAVE_CYCLES',1),1):.2f} valu-active {m.get('SQ_ACTIVE_INST_VALU',0)/max(m.get('SQ_WAVE_CYCLES',1),1):.2f}")
PY
