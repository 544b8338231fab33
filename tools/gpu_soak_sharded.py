"""Developer utility: long lockstep runs of a bar cut into slabs (particles sloshing across the cuts, table rebuilds inside the run);
no particle may be lost or doubled, nothing may overflow, everything stays finite."""
import os, sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
if os.environ.get("WGS_SOAK_NO_PERIOD") != "1": os.environ.setdefault("WGS_REHASH_PERIOD", "256")   # (WGS_SOAK_NO_PERIOD=1: no periodic rebuild — eviction keeps the slabs' tables instead)
import numpy as np
from helpers import pipeline
from gpu_common import _native_slabs
from wgsparkl_amd import scenes
from wgsparkl_amd.sharded import native_lockstep
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for world, dim, floor in ((4, 3, True), (3, 2, True), (8, 3, False)):
    if dim == 3:
        sc = scenes.config_scene("c2", world, None, "weak", n_side=24)
        if floor: sc["particles"].pos[:, 1] -= 5.6
        else: sc["colliders"] = []
    else:
        sc = scenes.elastic_block_2d(nx=48 * world, ny=40, with_floor=floor)
        sc["particles"].pos[:, 1] -= 4.6
    ps = sc["particles"]
    rng = np.random.default_rng(3)
    ps.vel[:] = rng.normal(0.0, 3.0, ps.vel.shape).astype(np.float32)
    ps.vel[:, 0] += 6.0 + 4.0 * np.sign(np.sin(np.arange(ps.n) * 0.01)).astype(np.float32)      # a drift along x (across the cuts) + bands moving against each other
    pipe = pipeline(dim)
    shards, part = _native_slabs(sc, world, pipe)
    t0 = time.time(); done = 0; crossed = 0
    n_prev = [s.num_particles() for s in shards]
    while done < steps:
        native_lockstep(pipe, shards, 100); done += 100
        for s in shards: s.sync()
        n_now = [s.num_particles() for s in shards]
        crossed += sum(abs(a - b) for a, b in zip(n_now, n_prev)); n_prev = n_now
        assert sum(n_now) == ps.n, (done, n_now)
    outs = [s.export() for s in shards]
    ids = np.sort(np.concatenate([o["ids"] for o in outs]))
    ok = np.array_equal(ids, np.arange(ps.n, dtype=np.uint32)) and all(np.isfinite(o["pos"]).all() and np.isfinite(o["vel"]).all() for o in outs)
    print(f"{world} slabs, {dim}D, floor={floor}: {steps} substeps, {ps.n} particles, {time.time()-t0:.1f}s, ids exact and finite: {ok}, net count changes {crossed}, per slab {n_prev}", flush=True)
    assert ok
    for s in shards: s.close()
