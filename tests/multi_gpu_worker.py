"""One rank of tests/test_multi_gpu.py (started by torch.distributed.run, one process per GPU): a scene through
`wgs_sharded_step` over real RCCL against rank 0's single-domain run (wgsparkl_amd/selfcheck.py). Rank 0 writes the
verdict as JSON to argv[2]."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    case, out = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from wgsparkl_amd import MpmPipeline, scenes, selfcheck
    from wgsparkl_amd.sharded import NativeComm, SlabPartition, associated_block_x, split_scene, uniform_material_of
    dim = 2 if case.endswith("2d") else 3
    pipe = MpmPipeline(local, dim)
    comm = NativeComm(pipe, dist, rank, world)
    if case == "bar":
        verdict = selfcheck.bar_check(pipe, dist, comm, world, rank, n_side=32, substeps=60)
    elif case == "c4":
        def c4(w, r):                               # the cube slides towards the paddle: particles cross the cut
            sc = scenes.config_scene("c4", w, r, "strong", n_side=48)
            sc["particles"].vel[:, 0] = (10.0 + 3.0 * np.sin(0.37 * sc["global_ids"].astype(np.float64))).astype(np.float32)
            return sc
        slab = c4(world, rank)
        slab["uniform_material"] = uniform_material_of(slab["particles"])
        verdict = selfcheck.compare_with_single_domain(pipe, dist, comm, world, rank, slab, lambda: c4(1, None), 60, vel_tol=5e-5)
    else:                                           # a golden collider scene cut into `world` slabs of equal particle count
        from golden_cases import CASES
        make, k = CASES[case]
        full = make()
        ps = full["particles"]
        part = SlabPartition.balanced(associated_block_x(ps.pos, full["cell_width"], dim), world)
        sub, gids = split_scene(ps, part, full["cell_width"])[rank]
        slab = dict(full, particles=sub, global_ids=gids, partition=part, uniform_material=uniform_material_of(ps))
        verdict = selfcheck.compare_with_single_domain(pipe, dist, comm, world, rank, slab, make, k, vel_tol=5e-5)
    if rank == 0:
        with open(out, "w") as f:
            json.dump(verdict, f)
        print(json.dumps(verdict))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()
    sys.exit(0 if verdict["ok"] else 3)


if __name__ == "__main__":
    main()
