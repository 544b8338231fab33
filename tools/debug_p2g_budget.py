"""Developer utility: the two register budgets of the one-way P2G pair, and the separate launches, on 640 k particles."""
import os, subprocess, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
if len(sys.argv) > 1:
    from helpers import pipeline
    from wgsparkl_amd import MpmData, scenes
    sc = scenes.neo_hookean_cube(n_side=86, with_floor=True); sc["particles"].pos[:, 1] -= 5.7
    pipe = pipeline(3)
    data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
    pipe.step(data, 3); data.sync(); pipe.step(data, int(sys.argv[2])); data.sync()
    p = data.read_particles()
    np.savez(sys.argv[1], pos=p.pos, vel=p.vel, F=p.def_grad, aff=p.cdf_affinity)
    sys.exit(0)
res = {}
for k in (1, 5):
    for flag in (0, 32768, 8192):
        out = f"/tmp/pb_{flag}_{k}.npz"
        subprocess.run([sys.executable, __file__, out, str(k)], env=dict(os.environ, WGS_DEBUG=str(flag)), check=True, stderr=subprocess.DEVNULL)
        res[flag, k] = np.load(out)
    for flag in (32768, 8192):
        a, b = res[0, k], res[flag, k]
        print(f"k={k} default vs {flag}: " + ", ".join(f"{f} max|d|={np.abs(a[f].astype(np.float64) - b[f].astype(np.float64)).max():.3e} ndiff={(a[f] != b[f]).sum()}" for f in ("pos", "vel", "F", "aff")))
