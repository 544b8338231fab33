"""-m "not gpu": the multi-GPU substep protocol (wgsparkl_amd/csrc/kernels_shard.h) restated on the CPU oracle
(tests/shard_oracle.py) under torch.distributed with the gloo backend, world_size 2 and 3: the decomposed run must
reproduce the single-domain oracle run (fp64, so only the association of the interface sums differs)."""
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


@pytest.mark.parametrize("dim,world,k", [(3, 2, 20), (2, 2, 20), (3, 3, 20)])
def test_gloo_matches_single_domain(oracle_libs, tmp_path, dim, world, k):
    """world_size 2 and 3 (an interior slab with two neighbours) over gloo: the one-exchange protocol — guests transferred
    to the grid by their old owner, advanced by their new owner from the message — reproduces the single-domain run."""
    out = str(tmp_path / "shard")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29611 + dim + 10 * world), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "shard_gloo_worker.py"), str(dim), str(k), out],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = [np.load(f"{out}.rank{r}.npz") for r in range(world)]

    sys.path.insert(0, HERE)
    from shard_gloo_worker import make_scene
    sc = make_scene(dim, world)
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


def test_protocol_lockstep_four_slabs_and_mask_table(oracle_libs):
    """The same restated protocol with all ranks in one process (4 slabs: two interior ones), and the interface masks it
    shares with kernels_shard.h spelt out for one slab."""
    sys.path.insert(0, HERE)
    from shard_gloo_worker import make_scene, partition_of
    from shard_oracle import OracleShard, iface_masks, lockstep
    from wgsparkl_amd.sharded import split_scene
    # rank with core range [8, 12), both neighbours, 3D (2 x-layer pairs per block): what each layer is to it
    m = {bx: iface_masks(bx, 8, 12, True, True, 2) for bx in range(6, 15)}      # (recv, send_lo, send_hi)
    assert m[6] == (0, 0, 0) and m[14] == (0, 0, 0) and m[10] == (0, 0, 0)
    assert m[7] == (0, 3, 0)          # layer lo - 1: only guests write it; everything they wrote goes down
    assert m[8] == (3, 1, 0)          # layer lo: first pair shared with the lower neighbour's core; all pairs may receive
    assert m[9] == (1, 0, 0)          # layer lo + 1: first pair may receive (the lower neighbour's guests)
    assert m[11] == (3, 0, 0)         # layer hi - 1: may receive (the upper neighbour's guests)
    assert m[12] == (1, 0, 3)         # layer hi: own core reaches the first pair, own guests all of it: all pairs go up
    assert m[13] == (0, 0, 1)         # layer hi + 1: own guests reach the first pair
    world, k = 4, 12
    sc = make_scene(3, world)
    part = partition_of(sc, world)
    assert part.min_interior_width() >= 3
    shards = []
    for r, (sub, gids) in enumerate(split_scene(sc["particles"], part, sc["cell_width"])):
        lo, hi = part.block_range(r)
        shards.append(OracleShard(sc, sub, gids, lo, hi, r > 0, r < world - 1))
    n0 = [len(s.gids) for s in shards]
    lockstep(shards, k)
    st = oracle_libs.Oracle(3, np.float64).new_state(sc["particles"], sc["params"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc.get("model", 0))
    st.step(k)
    outs = [s.export() for s in shards]
    ids = np.concatenate([o["ids"] for o in outs])
    assert np.array_equal(np.sort(ids), np.arange(sc["particles"].n)) and [len(o["ids"]) for o in outs] != n0
    order = np.argsort(ids)
    for f in ("pos", "vel", "def_grad", "affine"):
        got, ref = np.concatenate([o[f] for o in outs])[order], st.arr[f]
        assert np.sqrt(np.mean((got - ref) ** 2)) / max(np.sqrt(np.mean(ref * ref)), 1e-300) < 1e-9, f
