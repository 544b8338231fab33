cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r03e
timeout 1500 python -m pytest tests -m gpu -x -q -k "shard or lockstep or decomposition or rccl or slabs or launch_shapes or determinism" 2>&1 | tail -30
timeout 300 python tools/gpu_native_host_cost.py 2>&1 | grep substeps
timeout 600 python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['pass_ms_per_step'])"
bash tools/gpu_native_profile.sh 2>&1 | tail -22
timeout 400 python tools/gpu_native_host_cost.py c5 2>&1 | grep substeps
