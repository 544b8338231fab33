# developer utility: byte and SQ counters of tools/gpu_scene_prof.py (SCENE = sand3 | sand2 | stirred ...), per kernel, averaged over the second
# half of the launches -> gpurun_out/pmc_scene_$SCENE/summary.json (each counter set in a run of its own, --kernel-trace only, as the guide prescribes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=${SCENE:-sand3}
O=gpurun_out/pmc_scene_$S; rm -rf $O; mkdir -p $O
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- python3 tools/gpu_scene_prof.py $S > $O/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD
python3 - $O <<'PY'
import csv, glob, json, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    if not k.startswith("void wgs::k_") or "k_bin" in k or "k_bodies" in k: continue
    e = {}
    for cn, v in cs.items():
        v = v[len(v) // 2:]
        e[cn] = sum(v) / len(v)
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = e["FETCH_SIZE"] * 2048 + e["WRITE_SIZE"] * 1024   # FETCH x2 on gfx950 (MI355X_MICROARCH.md), units of 1 KiB
    if "SQ_WAVE_CYCLES" in e and "SQ_WAIT_ANY" in e: e["wait_frac_of_wave_cycles"] = e["SQ_WAIT_ANY"] / max(e["SQ_WAVE_CYCLES"], 1)
    out[k.split("(")[0]] = e
json.dump(out, open(O + "/summary.json", "w"), indent=1, sort_keys=True)
for k, e in out.items():
    print(k[:70], {a: (round(b, 3) if b < 10 else int(b)) for a, b in e.items() if a in ("hbm_bytes_per_launch", "wait_frac_of_wave_cycles", "SQ_INSTS_VALU")})
PY
