cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r03c
timeout 600 python -m pytest tests -m gpu -x -q -k "kinematic_collider or uniform_material or mesh_colliders_on_sharded" 2>&1 | tail -5
bash tools/gpu_native_profile.sh 2>&1 | tail -45
