cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0
D=gpurun_out/r04_t; rm -rf $D; mkdir -p $D
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/tr -- python3 tools/gpu_native_host_cost.py > $D/run.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04_t/tr/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the self-neighbour phase: find rccl kernels
idx = [i for i, r in enumerate(rows) if "ccl" in r["Kernel_Name"].lower() or "nccl" in r["Kernel_Name"].lower()]
print("kernels", len(rows), "rccl kernels", len(idx))
if idx:
    i0 = idx[len(idx) // 2]
    t0 = int(rows[i0 - 6]["Start_Timestamp"])
    for r in rows[i0 - 6:i0 + 14]:
        print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} .. {(int(r["End_Timestamp"]) - t0) / 1e3:9.1f} us  queue {r.get("Queue_Id", "?"):>3s}  {r["Kernel_Name"][:80]}')
PY
