"""Developer utility: the bench configurations run for thousands of substeps (free fall, landing, settling) — finite, inside the
domain, no error, and what the grid did meanwhile."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import pipeline
from wgsparkl_amd import MpmData, scenes
for cfg, total, per in (("c2", 20000, 500), ("c3", 4000, 250), ("c5", 1500, 250)):
    sc = scenes.config_scene(cfg)
    pipe = pipeline(3)
    data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
    t0 = time.perf_counter()
    for _ in range(total // per):
        pipe.step(data, per)
        data.sync()
    dt = time.perf_counter() - t0
    pos = data.read_positions()
    st = data.stats()
    print(f"{cfg}: {total} substeps in {dt:.1f} s ({1e6 * dt / total:.0f} us/substep), finite={bool(np.isfinite(pos).all())}, "
          f"y=[{pos[:,1].min():.2f},{pos[:,1].max():.2f}], x=[{pos[:,0].min():.2f},{pos[:,0].max():.2f}], {st}", flush=True)
    data.close()
