cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0
D=gpurun_out/r04_s; mkdir -p $D; rm -f $D/cost.log
echo "normal priority second stream" | tee -a $D/cost.log
WGS_STREAM2_NORMAL=1 timeout 300 python tools/gpu_native_host_cost.py 2>&1 | grep "substeps" | grep "self" | tee -a $D/cost.log
echo "GPU_MAX_HW_QUEUES=8" | tee -a $D/cost.log
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/gpu_native_host_cost.py 2>&1 | grep "substeps" | grep "self" | tee -a $D/cost.log
