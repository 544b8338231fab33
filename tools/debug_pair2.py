"""Developer utility: paired vs separate P2G launches after a sync (test_long_near_collider_list_paths...)."""
import os, subprocess, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
if len(sys.argv) > 1:
    from helpers import pipeline
    from wgsparkl_amd import MpmData, scenes
    sc = scenes.corotated_cube_with_paddle(n_side=64)
    pipe = pipeline(3)
    data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
    pipe.step(data, 4); data.sync(); pipe.step(data, 4); data.sync()
    p = data.read_particles()
    np.savez(sys.argv[1], pos=p.pos, vel=p.vel, affine=p.affine, aff=p.cdf_affinity)
    print(data.stats())
    sys.exit(0)
res = {}
for name, flag in (("pair1", 0), ("pair2", 0), ("sep1", 8192), ("sep2", 8192)):
    out = f"/tmp/{name}.npz"
    subprocess.run([sys.executable, __file__, out], env=dict(os.environ, WGS_DEBUG=str(flag)), check=True)
    res[name] = np.load(out)
for a, b in (("pair1", "pair2"), ("sep1", "sep2"), ("pair1", "sep1")):
    for f in ("pos", "vel", "affine", "aff"):
        x, y = res[a][f], res[b][f]
        bad = np.nonzero((x != y).reshape(len(x), -1).any(1))[0]
        print(a, b, f, "identical" if len(bad) == 0 else f"{len(bad)} particles differ, max abs {np.abs(x.astype(np.float64) - y).max():.3e}, first {bad[:5]}")
