# developer utility: the N > 1 code paths of bench.py on a 1-GPU box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show() { python3 -c "
import sys,json
lines=[l for l in open(sys.argv[1]) if l.strip()]
print('stdout lines:', len(lines), [l[:60] for l in lines[:-1]])
d=json.loads(lines[-1]); print(d['n_gpus'], d['scaling'], round(d['value']/1e9,3), round(d['ms_per_step'],4), d['config']['parallelism'], d['config']['global_particles']); print({k:(round(v['value']/1e9,2), round(v['ms_per_step']*1e3,1)) for k,v in d.get('extra',{}).items()})" $1; }
echo "== forced sharded path, one rank, RCCL communicator (wgs_sharded_step)"
WGS_BENCH_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bp1.out 2> gpurun_out/bp1.err; echo rc=$?; show gpurun_out/bp1.out; grep -v "amdgpu.ids\|socket.cpp" gpurun_out/bp1.err | tail -5
echo "== two ranks on one GPU over gloo (python-driven protocol, functional)"
WGS_BENCH_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 10 --warmup 3 --no-extra > gpurun_out/bp2.out 2> gpurun_out/bp2.err; echo rc=$?; show gpurun_out/bp2.out; grep -v "amdgpu.ids\|socket.cpp" gpurun_out/bp2.err | tail -5
echo "== --gpus 2 without a launcher (self-spawn), gloo"
WGS_BENCH_ONE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --no-extra --config c5 --scaling strong --n-side 48 > gpurun_out/bp3.out 2> gpurun_out/bp3.err; echo rc=$?; show gpurun_out/bp3.out; grep -v "amdgpu.ids\|socket.cpp" gpurun_out/bp3.err | tail -5
