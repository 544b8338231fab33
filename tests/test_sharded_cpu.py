"""-m "not gpu": the multi-GPU protocol (wgsparkl_amd/sharded.py) under torch.distributed with the gloo
backend, world_size 2, with the CPU oracle as the per-rank solver: the decomposed run must reproduce the
single-domain oracle run (fp64, so only the association of the interface sums differs)."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_partition_helpers():
    from wgsparkl_amd.sharded import INT_MAX, INT_MIN, SlabPartition
    p = SlabPartition([0, 4, 9, 12])
    assert p.world == 3
    assert p.block_range(0) == (INT_MIN, 4) and p.block_range(1) == (4, 9) and p.block_range(2) == (9, INT_MAX)
    assert p.owner_of_blocks(np.array([-5, 0, 3, 4, 8, 9, 100])).tolist() == [0, 0, 0, 1, 1, 2, 2]
    bx = np.repeat(np.arange(10), 100)
    q = SlabPartition.balanced(bx, 4)
    counts = np.bincount(q.owner_of_blocks(bx), minlength=4)
    assert counts.min() >= 200 and counts.sum() == 1000


@pytest.mark.parametrize("dim,k,pipelined", [(3, 20, False), (2, 20, False), (3, 20, True)])
def test_gloo_world2_matches_single_domain(oracle_libs, tmp_path, dim, k, pipelined):
    out = str(tmp_path / "shard")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29611 + dim + (7 if pipelined else 0)), WORLD_SIZE="2",
               OMP_NUM_THREADS="1", WGS_PIPELINED="1" if pipelined else "0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "shard_gloo_worker.py"), str(dim), str(k), out],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    logs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = [np.load(f"{out}.rank{r}.npz") for r in range(2)]

    sys.path.insert(0, HERE)
    from shard_gloo_worker import make_scene
    sc = make_scene(dim)
    ps = sc["particles"]
    st = oracle_libs.Oracle(dim, np.float64).new_state(ps, sc["params"], sc["colliders"], sc["cell_width"],
                                                       sc["grid_capacity"], sc.get("model", 0))
    st.step(k)
    ids = np.concatenate([r["ids"] for r in res])
    assert np.array_equal(np.sort(ids), np.arange(ps.n))                     # nobody lost, nobody duplicated
    assert [len(r["ids"]) for r in res] != [int(r["n0"][0]) for r in res]    # particles did migrate
    order = np.argsort(ids)
    for f in ("pos", "vel", "def_grad", "affine"):
        got = np.concatenate([r[f] for r in res])[order]
        ref = st.arr[f]
        scale = max(np.sqrt(np.mean(ref * ref)), 1e-300)
        err = np.sqrt(np.mean((got - ref) ** 2)) / scale
        assert err < 1e-9, (f, err)


def test_fixed_exchange_routing_gloo_world3():
    """FixedExchange (the transport bench.py uses over RCCL) routes lower/upper messages correctly; gloo, 3 ranks."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", WORLD_SIZE="3", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "fixed_exchange_worker.py")], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(3)]
    logs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
