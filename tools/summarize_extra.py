"""gpurun_out/<round>_extra, <round>_sharded (tools/gpu_round_evidence.sh) -> profiles/<round>_kernel_stats_{c3,c4,c5,stirred,sand3}.csv,
profiles/<round>_pmc_summary_{c3,c4,c5,sand3,stirred}.json, profiles/<round>_sharded_cost.txt, profiles/<round>_sharded_self_neighbours_{1m,c5}_slab_kernel_stats.csv."""
import csv, json, os, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
CMD = {"c3": "rocprofv3 --kernel-trace --stats -- python3 bench.py --config c3 --steps 50 --warmup 10 --no-cpu-baseline --no-extra --no-live-pmc",
       "c4": "rocprofv3 --kernel-trace --stats -- python3 bench.py --config c4 --steps 50 --warmup 10 --no-cpu-baseline --no-extra --no-live-pmc",
       "c5": "rocprofv3 --kernel-trace --stats -- python3 bench.py --config c5 --steps 50 --warmup 10 --no-cpu-baseline --no-extra --no-live-pmc",
       "stirred": "rocprofv3 --kernel-trace --stats -- python3 tools/gpu_scene_prof.py stirred (bench.py's c2_stirred scene: 5 + 50 substeps)",
       "sand3": "rocprofv3 --kernel-trace --stats -- python3 tools/gpu_scene_prof.py sand3 (the reference's sand3 scene: 100 + 200 substeps)"}
def copy_stats(src, dst, cmd):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w") as f:
        f.write(f"# {cmd} (MI355X)\n")
        w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows: w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    top = sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:5]
    print(os.path.basename(dst)); [print(f"   {r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us") for r in top]
for k, cmd in CMD.items():
    src = f"gpurun_out/{R}_extra/kernel_stats_{k}.csv"
    if os.path.exists(src): copy_stats(src, f"profiles/{R}_kernel_stats_{k}.csv", cmd)
NOTE = {"c3": "tools/gpu_pmc_cfg.sh CFG=c3: bench.py --config c3 --steps 12 --warmup 4", "c4": "tools/gpu_pmc_cfg.sh CFG=c4: bench.py --config c4 --steps 12 --warmup 4",
        "c5": "tools/gpu_pmc_cfg.sh CFG=c5: bench.py --config c5 --steps 12 --warmup 4", "sand3": "tools/gpu_scene_pmc.sh SCENE=sand3: tools/gpu_scene_prof.py sand3 (100 + 200 substeps)",
        "stirred": "tools/gpu_scene_pmc.sh SCENE=stirred: tools/gpu_scene_prof.py stirred (bench.py's c2_stirred scene, 5 + 50 substeps)"}
for tag, what in NOTE.items():
    src = f"gpurun_out/{R}_extra/pmc_summary_{tag}.json"
    if not os.path.exists(src): continue
    d = json.load(open(src))
    d["_note"] = (what + "; rocprofv3 --kernel-trace --pmc, one counter set per run (FETCH_SIZE | WRITE_SIZE | two SQ sets); per kernel, mean over the second half of the "
                  "launches; hbm_bytes_per_launch = FETCH_SIZE x 2048 + WRITE_SIZE x 1024 (units of 1 KiB; FETCH doubled on gfx950 per MI355X_MICROARCH.md)")
    json.dump(d, open(f"profiles/{R}_pmc_summary_{tag}.json", "w"), indent=1, sort_keys=True)
    for k, e in d.items():
        if k.startswith("void wgs::k_g2p") or k.startswith("void wgs::k_p2g") or k.startswith("void wgs::k_regroup"):
            print(tag, k[:60], "MB/launch", round(e.get("hbm_bytes_per_launch", 0) / 1e6, 1), "wait frac", round(e.get("wait_frac_of_wave_cycles", 0), 3), "VALU insts", int(e.get("SQ_INSTS_VALU", 0)))
lines = []
for cfg, name in (("c2", "1m"), ("c5", "c5")):
    f = f"gpurun_out/{R}_sharded/cost_{cfg}.log"
    if os.path.exists(f): lines += [f"== tools/gpu_native_host_cost.py {cfg}"] + [l.rstrip() for l in open(f)]
    src = f"gpurun_out/{R}_sharded/kernel_stats_{cfg}.csv"
    if os.path.exists(src):
        copy_stats(src, f"profiles/{R}_sharded_self_neighbours_{name}_slab_kernel_stats.csv", f"rocprofv3 --kernel-trace --stats -- python3 tools/gpu_native_host_cost.py {cfg} (one rank as its own two neighbours over RCCL)")
if lines:
    open(f"profiles/{R}_sharded_cost.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
