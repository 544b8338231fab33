"""-m gpu, boxes with two or more GPUs only: `wgs_sharded_step` between real ranks over RCCL (one process per GPU,
fresh child processes: nothing here touches a GPU in the pytest process) against the single-domain run of the same
scene — the elastic bar of the bench preflight, configs[3] with its kinematic rotating cuboid, a dynamic body (two-way
coupling: the ranks' fixed-point impulses are all-reduced) and mesh colliders. One-GPU boxes skip; there the same
protocol code runs as a lockstep group on one device (test_gpu_sharded.py) and over gloo on the CPU (test_sharded_cpu.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _gpus() -> int:
    try:
        import torch
        return int(torch.cuda.device_count())          # counts devices without initialising the GPU in this process
    except Exception:
        return 0


@pytest.mark.skipif(_gpus() < 2, reason="needs at least two GPUs (one process per GPU over RCCL)")
@pytest.mark.parametrize("case,world", [("bar", 2), ("bar", 4), ("c4", 2), ("dynamic_ball3d", 2), ("mesh_floor3d", 2), ("dynamic_ball2d", 2)])
def test_sharded_step_over_rccl_matches_single_domain(hip_libs, tmp_path, case, world):
    if _gpus() < world:
        pytest.skip(f"needs {world} GPUs")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "verdict.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "multi_gpu_worker.py"), case, out]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    log = r.stdout.decode()[-6000:]
    assert os.path.exists(out), log
    verdict = json.load(open(out))
    assert r.returncode == 0 and verdict["ok"], (verdict, log)
    assert verdict["ids_exact"]
    if case in ("bar", "c4"):
        assert verdict["migrated"], "the test scene must make particles cross the faces"
