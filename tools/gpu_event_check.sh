# Developer utility: per-pass HIP-event timings of bench.py next to rocprofv3's kernel durations of the same run.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ev; mkdir -p gpurun_out/ev
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ev -- python3 bench.py --steps 64 --warmup 10 --no-cpu-baseline $BENCHARGS > gpurun_out/ev/bench.log 2>&1
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/ev/bench.log") if l.startswith("{")][-1])
print("ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"])
print({k: round(v * 1e3, 1) for k, v in d["pass_ms_per_step"].items()})
PY
python3 tools/show_stats.py gpurun_out/ev | head -12
