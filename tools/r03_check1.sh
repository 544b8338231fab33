# round 3, first look: new tests, the sharded preflight with one rank, baseline numbers of this box
cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r03a
timeout 900 python -m pytest tests -m gpu -x -q -k "c1_configs0 or multi_gpu or native_lockstep" > gpurun_out/r03a/tests.log 2>&1; echo "tests rc=$?" 
tail -5 gpurun_out/r03a/tests.log
timeout 300 python - > gpurun_out/r03a/preflight.log 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
from wgsparkl_amd import MpmPipeline, selfcheck
from wgsparkl_amd.sharded import NativeComm
pipe = MpmPipeline(0, 3)
comm = NativeComm(pipe, None, 0, 1)
print(selfcheck.bar_check(pipe, None, comm, 1, 0))
PY
tail -3 gpurun_out/r03a/preflight.log
timeout 600 python bench.py > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03a/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['pass_ms_per_step'])
for k, v in d.get('extra', {}).items(): print(k, v.get('ms_per_step'), v.get('roofline_g2p', {}).get('frac'), v.get('pass_ms_per_step'))
PY
timeout 300 python tools/gpu_native_host_cost.py > gpurun_out/r03a/native.log 2>&1; cat gpurun_out/r03a/native.log | tail -12
