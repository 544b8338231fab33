# round 4, step E: parity of the reworked binning, then the full bench line (new legs) with and without it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/r04_e; mkdir -p $D
timeout 1500 python -m pytest tests -m gpu -q -x -k "binning_inside or steady_state_rebinning or determinism or golden or c1_configs0 or grid_update_inside or fast_translation or grid_grows or checkpoint_restart or reference_sand3 or random_scenes_match or long_near or plastic_pair or large_one_way or dense_blocks or visit_list or 2d_block or multi_substep or dynamic or exploding or random_api" 2>&1 | tail -8 > $D/pytest.log
cat $D/pytest.log
timeout 900 python bench.py --no-cpu-baseline --no-live-pmc > $D/bench_default.json 2> $D/bench_default.err
WGS_DEBUG=1048576 timeout 900 python bench.py --no-cpu-baseline --no-live-pmc --allow-debug-switches > $D/bench_rebin.json 2> $D/bench_rebin.err
python - <<'PY'
import json
for name in ("default", "rebin"):
    try:
        d = json.load(open(f"gpurun_out/r04_e/bench_{name}.json"))
    except Exception as e:
        print(name, "no json", e); continue
    print(name, "c2", round(d["ms_per_step"]*1e3,1), {a: round(b*1e3,1) for a,b in d["pass_ms_per_step"].items() if b > 0.006})
    for k, v in d.get("extra", {}).items():
        if "ms_per_step" in v:
            print("  ", k, round(v["ms_per_step"]*1e3,1), "movers", v.get("mover_fraction"), {a: round(b*1e3,1) for a,b in v["pass_ms_per_step"].items() if b > 0.006})
        else:
            print("  ", k, v)
PY
tail -3 $D/bench_default.err
