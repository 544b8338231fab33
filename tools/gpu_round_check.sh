# developer utility: the whole -m gpu suite, rocprofv3 kernel stats of two moving scenes (the reference sand3, the stirred cube) and the
# default bench line with its extra legs, summarised — what a change to the kernels is checked with before it is committed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/check; mkdir -p $D
timeout 2400 python -X faulthandler -m pytest tests -m gpu -q -x --deselect tests/test_multi_gpu.py > $D/pytest_full.log 2>&1
grep -n "FAILED\|Fatal\|Error\|passed\|failed\|File \"/tmp" $D/pytest_full.log | tail -20
for sc in sand3 stirred; do SCENE=$sc bash tools/gpu_scene_kstats.sh 2>&1 | grep -v amdgpu.ids | grep "k_regroup\|k_g2p_pair\|k_p2g" | cut -c1-160 | tee -a $D/kstats.log; done
timeout 900 python bench.py --no-cpu-baseline --no-live-pmc > $D/bench_default.json 2> $D/bench_default.err
python - <<'PY'
import json
for name in ("default",):
    try:
        d = json.load(open(f"gpurun_out/check/bench_{name}.json"))
    except Exception as e:
        print(name, "no json", e); continue
    print(name, "c2", round(d["ms_per_step"]*1e3,1), {a: round(b*1e3,1) for a,b in d["pass_ms_per_step"].items() if b > 0.006})
    for k, v in d.get("extra", {}).items():
        if "ms_per_step" in v:
            print("  ", k, round(v["ms_per_step"]*1e3,1), "movers", v.get("mover_fraction"), {a: round(b*1e3,1) for a,b in v["pass_ms_per_step"].items() if b > 0.006})
        else:
            print("  ", k, v)
PY
