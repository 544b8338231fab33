# developer utility: FETCH_SIZE / WRITE_SIZE of the fused G2P launch (KiB per launch, raw), optionally with WGS_DEBUG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcx_$c; mkdir -p gpurun_out/pmcx_$c
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcx_$c -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra $ARGS > gpurun_out/pmcx_$c/log 2>&1
  f=$(find gpurun_out/pmcx_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "g2p" in k or "p2g" in k: v=v[len(v)//2:]; print(sys.argv[2], k, round(sum(v)/len(v)*1024/1e6,1), "MB raw")
PY
done
