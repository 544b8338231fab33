"""Rank of the FixedExchange routing test (spawned by tests/test_sharded_cpu.py): gloo, CPU tensors."""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgsparkl_amd.sharded import FixedExchange

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ex = FixedExchange(dist, rank, world)
for it in range(3):
    to_lower = torch.full((5,), float(100 * rank + it), dtype=torch.float32) if rank > 0 else None
    to_upper = torch.full((5,), float(100 * rank + 50 + it), dtype=torch.float32) if rank < world - 1 else None
    from_lower, from_upper = ex(to_lower, to_upper)
    if rank > 0:
        assert from_lower.shape == (5,)
        assert torch.all(from_lower == float(100 * (rank - 1) + 50 + it)), (rank, from_lower)
    else:
        assert from_lower is None
    if rank < world - 1:
        assert torch.all(from_upper == float(100 * (rank + 1) + it)), (rank, from_upper)
    else:
        assert from_upper is None
dist.barrier()
dist.destroy_process_group()
print("ok", rank)
