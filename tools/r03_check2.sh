# round 3: the one-exchange sharded protocol on the GPU — sharded tests first, then the whole suite, then the self-neighbour cost
cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r03b
timeout 1500 python -m pytest tests -m gpu -x -q -k "shard or lockstep or decomposition or rccl or slabs or multi_gpu" > gpurun_out/r03b/tests_sharded.log 2>&1; echo "sharded tests rc=$?"
tail -25 gpurun_out/r03b/tests_sharded.log
timeout 300 python tools/gpu_native_host_cost.py > gpurun_out/r03b/native.log 2>&1; grep substeps gpurun_out/r03b/native.log
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r03b/tests_all.log 2>&1; echo "all tests rc=$?"
tail -15 gpurun_out/r03b/tests_all.log
