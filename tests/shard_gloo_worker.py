"""One rank of the world_size-2 gloo test (spawned by tests/test_sharded_cpu.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from shard_oracle import OracleShard  # noqa: E402
from wgsparkl_amd import scenes  # noqa: E402
from wgsparkl_amd.sharded import (DistExchange, SlabPartition, associated_block_x, split_scene,  # noqa: E402
                                  finish_migration, pipelined_substep, substep_phases)


def make_scene(dim):
    if dim == 3:
        sc = scenes.neo_hookean_cube(n_side=16)
    else:
        sc = scenes.elastic_block_2d(nx=40, ny=24, with_floor=False)
    rng = np.random.default_rng(8)
    sc["particles"].vel[:] = rng.normal(0.0, 3.0, sc["particles"].vel.shape).astype(np.float32)
    sc["particles"].vel[:, 0] += 60.0   # ~1 cell in 20 substeps: particles cross the slab face
    return sc


def main():
    dim, k, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    sc = make_scene(dim)
    ps = sc["particles"]
    bx = associated_block_x(ps.pos, sc["cell_width"], dim)
    mid = int(np.median(bx))
    part = SlabPartition([int(bx.min()), max(mid, int(bx.min()) + 1), int(bx.max()) + 1])
    sub, gids = split_scene(ps, part, sc["cell_width"])[rank]
    lo, hi = part.block_range(rank)
    shard = OracleShard(sc, sub, gids, lo, hi, rank > 0, rank < world - 1)
    ex = DistExchange(dist, rank, world, torch.device("cpu"), dtype=torch.float64)
    n0 = len(shard.gids)
    if os.environ.get("WGS_PIPELINED") == "1":   # the order bench.py uses for N > 1
        pending = None
        for _ in range(k):
            pending = pipelined_substep(shard, ex, pending)
        finish_migration(shard, pending)
    else:
        for _ in range(k):
            substep_phases(shard, ex)
    res = shard.export()
    res["n0"] = np.array([n0])
    np.savez(f"{out}.rank{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
