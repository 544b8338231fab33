"""Self-validation of the multi-GPU path: the same scene through `wgs_sharded_step` on N ranks (one process per GPU,
RCCL between them) and through `wgs_step` on rank 0 alone, compared particle by particle.

NEW DESIGN (the reference is single-GPU, src/pipeline.rs:176-193): there is no reference behaviour to match for the
decomposition itself, so the contract is "the decomposed run IS the single-domain run": the same particle ids, none lost
or duplicated, positions / velocities to fp32 round-off (only the association of the interface node sums differs).
`bench.py --gpus N` runs `bar_check` as a preflight before it times anything and refuses to report a number if it
fails; tests/test_multi_gpu.py runs it and the golden collider scenes on boxes with two or more GPUs.

Nothing here touches the CPU oracle: both sides of the comparison are the HIP path.
"""
from __future__ import annotations

import numpy as np


def _rel_rms(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    den = float(np.sqrt(np.mean(b * b)))
    num = float(np.sqrt(np.mean((a - b) ** 2)))
    return num / den if den > 0 else num


def compare_with_single_domain(pipe, dist, comm, world, rank, slab, full_scene, substeps, calls=2, pos_tol=1e-5, vel_tol=1e-5,
                               body_tol=3e-4):
    """`slab` = this rank's part of the scene (dict with particles, global_ids, partition, params, colliders, cell_width,
    grid_capacity, model); `full_scene()` builds the whole scene (called on rank 0 only; its `global_ids`, if present,
    name the particles, else their index does). Advances the slab `substeps` substeps in `calls` calls of
    `wgs_sharded_step`, gathers every rank's particles on rank 0 and compares them with rank 0's single-domain run.
    Returns the same verdict dict on every rank ({"ok": bool, ...})."""
    from . import MpmData
    from .sharded import NativeShard
    ps = slab["particles"]
    lo, hi = slab["partition"].block_range(rank)
    ys = ps.pos[:, 1] if ps.n else np.zeros(1, np.float32)
    zs = ps.pos[:, 2] if ps.n and ps.dim == 3 else np.zeros(1, np.float32)
    face_blocks = (int((ys.max() - ys.min()) / slab["cell_width"]) // 4 + 4) * (int((zs.max() - zs.min()) / slab["cell_width"]) // 4 + 4)
    shard = NativeShard(pipe, slab["params"], ps, slab["global_ids"], slab["colliders"], slab["cell_width"], slab["grid_capacity"], lo, hi,
                        rank > 0, rank < world - 1, particle_capacity=int(ps.n * 1.5) + 4096, model=slab["model"],
                        halo_capacity_records=3 * face_blocks + 64, migrant_capacity=max(1024, ps.n // 8), comm=comm,
                        uniform_material=slab.get("uniform_material"))
    n0 = shard.num_particles()
    per = max(1, substeps // max(1, calls))
    done = 0
    while done < substeps:
        k = min(per, substeps - done)
        shard.step(k)
        done += k
    shard.sync()                                   # reports halo / migration / capacity overflows
    mine = shard.export()
    mine = {k: mine[k] for k in ("ids", "pos", "vel", "def_grad")}
    mine["n0"] = n0
    mine["bodies"] = shard.read_body_poses() if slab["colliders"] else []
    shard.close()
    gathered = [None] * world if rank == 0 else None
    if world > 1:
        dist.gather_object(mine, gathered, dst=0)
    else:
        gathered = [mine]
    verdict = [None]
    if rank == 0:
        full = full_scene()
        fp = full["particles"]
        gid = np.asarray(full.get("global_ids", np.arange(fp.n)), np.int64)
        data = MpmData.new(pipe, full["params"], fp, full["colliders"], full["cell_width"], full["grid_capacity"], full["model"])
        pipe.step(data, substeps)
        data.sync()
        ref = data.read_particles()
        ref_bodies = data.read_body_poses() if full["colliders"] else []
        data.close()
        ids = np.concatenate([g["ids"] for g in gathered]).astype(np.int64)
        v = {"ranks": world, "substeps": substeps, "particles": int(fp.n), "ids_exact": bool(np.array_equal(np.sort(ids), np.sort(gid))),
             "migrated": bool([len(g["ids"]) for g in gathered] != [g["n0"] for g in gathered]),
             "particles_per_rank_after": [int(len(g["ids"])) for g in gathered], "against": "wgs_step on the whole scene, rank 0"}
        if v["ids_exact"]:
            order, ref_order = np.argsort(ids), np.argsort(gid)
            for f, tol in (("pos", pos_tol), ("vel", vel_tol), ("def_grad", pos_tol)):
                got = np.concatenate([g[f] for g in gathered])[order]
                v[f + "_rel_rms"] = _rel_rms(got, getattr(ref, f)[ref_order])
            v["ok"] = bool(v["pos_rel_rms"] < pos_tol and v["vel_rel_rms"] < vel_tol and v["def_grad_rel_rms"] < pos_tol)
        else:
            v["ok"] = False
        if ref_bodies:
            worst = 0.0
            for g in gathered:                     # every rank integrates the same bodies
                for b, rb in zip(g["bodies"], ref_bodies):
                    for key in ("rotation", "translation", "linvel", "angvel"):
                        worst = max(worst, float(np.abs(np.asarray(b[key]) - np.asarray(rb[key])).max()))
            v["bodies_abs_err"] = worst
            v["ok"] = bool(v["ok"] and worst < body_tol)
        verdict = [v]
    if world > 1:
        dist.broadcast_object_list(verdict, src=0)
    return verdict[0]


def bar_check(pipe, dist, comm, world, rank, n_side=32, substeps=40):
    """The preflight of `bench.py --gpus N`: a small elastic bar over the floor (the weak-scaling workload at n_side^3
    particles per rank), pushed along x so that particles cross every face."""
    from . import scenes
    from .sharded import uniform_material_of

    def vx(gid):
        return (12.0 + 3.0 * np.sin(0.37 * np.asarray(gid, np.float64))).astype(np.float32)   # 0.3 - 0.5 cells in 40 substeps

    slab = scenes.neo_hookean_bar(n_side=n_side, world=world, rank=rank)
    slab["particles"].vel[:, 0] = vx(slab["global_ids"])
    slab["uniform_material"] = uniform_material_of(slab["particles"])     # one material on every rank (scenes.py)

    def full_scene():
        full = scenes.neo_hookean_bar(n_side=n_side, world=world, rank=None)
        full["particles"].vel[:, 0] = vx(full["global_ids"])
        full["grid_capacity"] *= world
        return full

    return compare_with_single_domain(pipe, dist, comm, world, rank, slab, full_scene, substeps)
