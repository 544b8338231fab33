"""Turns gpurun_out/<round>/ (made by tools/gpu_round_profile.sh on the GPU box) into the small files
committed under profiles/: per-kernel rocprofv3 --stats table, PMC byte counters per launch with the
gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (x2 for 16-B-per-lane streams), the bench line."""
import csv, glob, json, os, sys, collections
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = f"gpurun_out/{R}"
os.makedirs("profiles", exist_ok=True)
st = sorted(glob.glob(f"{src}/stats/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(st)))
with open(f"profiles/{R}_kernel_stats.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra (MI355X)\n")
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows: w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
pmc = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ"):
    fs = sorted(glob.glob(f"{src}/pmc_{c}/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
    if not fs: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for cn, v in cs.items():
            v = v[len(v) // 2:]
            pmc[k][cn] = sum(v) / len(v)
out = {}
for k, cs in pmc.items():
    e = dict(cs)
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        # units: KiB-ish (1024 B) per the guide's `(FETCH_SIZE + WRITE_SIZE) * 1024`; FETCH x2 on gfx950 for 16-B/lane streams
        e["hbm_read_bytes_per_launch_corrected"] = cs["FETCH_SIZE"] * 1024 * 2
        e["hbm_write_bytes_per_launch"] = cs["WRITE_SIZE"] * 1024
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch_corrected"] + e["hbm_write_bytes_per_launch"]
    out[k] = e
json.dump(out, open(f"profiles/{R}_pmc_summary.json", "w"), indent=1, sort_keys=True)
g2p = [v for k, v in out.items() if ("k_g2p_update" in k or "k_g2p_pair" in k) and "hbm_bytes_per_launch" in v]
if g2p:
    best = max(g2p, key=lambda v: v["hbm_bytes_per_launch"])
    json.dump({"kernel": "k_g2p_pair / k_g2p_update (dominant instantiation)", "hbm_bytes_per_launch": best["hbm_bytes_per_launch"],
               "fetch_size_raw_kib": best["FETCH_SIZE"], "write_size_raw_kib": best["WRITE_SIZE"],
               "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); separate --pmc passes"},
              open(f"profiles/{R}_pmc_g2p.json", "w"), indent=1)
if os.path.exists(f"{src}/bench.json"):
    line = [l for l in open(f"{src}/bench.json") if l.startswith("{")]
    if line:
        line.sort(key=lambda l: json.loads(l)["value"])
        open(f"profiles/{R}_bench.json", "w").write(line[len(line) // 2])      # the median run
        open(f"profiles/{R}_bench_runs.jsonl", "w").write("".join(line))         # all of them
print(open(f"profiles/{R}_kernel_stats.csv").read()[:1800])
