"""Developer utility: more seeds of tests/test_gpu_sharded.py::test_random_scenes_sharded_match_single_domain than the suite runs
(random clouds with kinematic colliders cut into 2-4 lockstep slabs against the single-domain run).  usage: gpu_fuzz_sharded.py LO HI [substeps]"""
import sys, traceback; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import pipeline, rel_rms, run_gpu
from gpu_common import _native_slabs
from wgsparkl_amd import scenes
from wgsparkl_amd.models import ParticlePhase
from wgsparkl_amd.sharded import native_lockstep
from wgsparkl_amd.solver import Collider, SimulationParams
lo, hi = int(sys.argv[1]), int(sys.argv[2])
k = int(sys.argv[3]) if len(sys.argv) > 3 else 30
bad, skipped = [], 0
for seed in range(lo, hi):
    try:
        rng = np.random.default_rng(500 + seed)
        dim = 3 if seed % 2 == 0 else 2
        world = int(rng.integers(2, 5))
        stretch = 3.0 if dim == 3 else 5.0
        ps = scenes.random_cloud(4000, dim=dim, seed=300 + seed, extent=22.0, young=1e6, phase=ParticlePhase(1.0, -1.0), vel_scale=2.5, perturb_F=0.02, perturb_C=0.2)
        ps.pos[:, 0] *= np.float32(stretch)
        ps.vel[:, 0] += np.float32(rng.uniform(-6.0, 6.0))
        cols = []
        for _ in range(int(rng.integers(0, 3))):
            pos = [float(x) for x in rng.uniform(2.0, 20.0, dim)]
            pos[0] *= stretch
            kw = dict(linvel=tuple(float(x) for x in rng.uniform(-1.0, 1.0, 3)), angvel=tuple(float(x) for x in rng.uniform(-0.5, 0.5, 3 if dim == 3 else 1)))
            cols.append(Collider.ball(float(rng.uniform(1.0, 3.0)), tuple(pos), **kw) if rng.random() < 0.5 else
                        Collider.cuboid(tuple(float(x) for x in rng.uniform(1.0, 4.0, dim)), tuple(pos), **kw))
        sc = dict(particles=ps, params=SimulationParams(gravity=(0.0, -9.81, 0.0)[:dim], dt=8e-4), colliders=cols, cell_width=1.0, grid_capacity=4096, model=int(rng.integers(0, 2)))
        ref_data, shards = None, []
        try:
            ref_data = run_gpu(sc, k)
            ref = ref_data.read_particles()
            pipe = pipeline(dim)
            shards, part = _native_slabs(sc, world, pipe)
            if part.min_interior_width() < 3:
                skipped += 1
                continue
            native_lockstep(pipe, shards, k)
            for s in shards: s.sync()
            outs = [s.export() for s in shards]
            ids = np.concatenate([o["ids"] for o in outs])
            assert np.array_equal(np.sort(ids), np.arange(ps.n, dtype=np.uint32)), "ids"
            order = np.argsort(ids)
            for f, tol in (("pos", 1e-5), ("vel", 2e-4)):
                err = rel_rms(np.concatenate([o[f] for o in outs])[order], getattr(ref, f))
                assert err < tol, (f, err)
        finally:   # (device memory of every seed is released whatever happened: a long range must not fail on leaked memory)
            for s in shards: s.close()
            if ref_data is not None: ref_data.close()
    except Exception as e:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        bad.append((seed, repr(e)))
print(f"sharded fuzz seeds {lo}..{hi}: failures {bad}, skipped (slab too narrow) {skipped}")
