cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/r04_pmcq; rm -rf $D; mkdir -p $D
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $D/pmc_SQ -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $D/pmc_SQ.log 2>&1
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/r04_pmcq/pmc_SQ/**/*_counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[-1])):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "wgs::" not in k: continue
    m = {c: sum(v[len(v)//2:]) / max(1, len(v[len(v)//2:])) for c, v in cs.items()}
    print(k[:70], {c: round(x) for c, x in m.items() if c in ("SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU", "SQ_WAVES")}, "conflict cycles / LDS instr", round(m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, m.get("SQ_INSTS_LDS", 1)), 2))
PY
