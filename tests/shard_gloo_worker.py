"""One rank of the gloo tests of tests/test_sharded_cpu.py: the substep protocol of the multi-GPU path restated on
the CPU oracle (tests/shard_oracle.py), one process per rank over torch.distributed (gloo)."""
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from shard_oracle import DistExchange, OracleShard  # noqa: E402
from wgsparkl_amd import scenes  # noqa: E402
from wgsparkl_amd.sharded import SlabPartition, associated_block_x, split_scene  # noqa: E402


def make_scene(dim, world):
    """A bar along x, a few blocks per slab, pushed along x so that particles cross every cut."""
    if dim == 3:
        sc = scenes.config_scene("c2", world, None, "weak", n_side=12)      # 12 * world x 12 x 12 particles
        sc["colliders"] = []
        if world == 2:                                                         # 6 cells per rank would be 1.5 blocks: widen
            sc = scenes.config_scene("c2", 4, None, "weak", n_side=12)
            sc["colliders"] = []
    else:
        sc = scenes.elastic_block_2d(nx=48 * world, ny=16, with_floor=False)
    rng = np.random.default_rng(8)
    sc["particles"].vel[:] = rng.normal(0.0, 3.0, sc["particles"].vel.shape).astype(np.float32)
    sc["particles"].vel[:, 0] += 60.0   # ~1 cell in 20 substeps
    return sc


def partition_of(sc, world):
    ps = sc["particles"]
    return SlabPartition.balanced(associated_block_x(ps.pos, sc["cell_width"], ps.dim), world)


def main():
    dim, k, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    sc = make_scene(dim, world)
    part = partition_of(sc, world)
    sub, gids = split_scene(sc["particles"], part, sc["cell_width"])[rank]
    lo, hi = part.block_range(rank)
    shard = OracleShard(sc, sub, gids, lo, hi, rank > 0, rank < world - 1)
    ex = DistExchange(dist, rank, world)
    n0 = len(shard.gids)
    for _ in range(k):
        to_lower, to_upper = shard.begin()
        shard.end(*ex(to_lower, to_upper))
    res = shard.export()
    res["n0"] = np.array([n0])
    np.savez(f"{out}.rank{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
