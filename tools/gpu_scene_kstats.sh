# rocprofv3 kernel stats of tools/gpu_scene_prof.py (developer utility): SCENE = sand3 | c1 | cube64
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kstats_scene; mkdir -p gpurun_out/kstats_scene
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_scene -- python3 tools/gpu_scene_prof.py ${SCENE:-sand3} > gpurun_out/kstats_scene/run.log 2>&1
tail -1 gpurun_out/kstats_scene/run.log
f=$(find gpurun_out/kstats_scene -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']}%")
PY
