"""Developer utility: the two register budgets of the plastic G2P pair kernel on a sand column between walls — bitwise?"""
import os, subprocess, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
if len(sys.argv) > 1:
    from helpers import pipeline
    from wgsparkl_amd import MpmData, scenes
    sc = scenes.sand_column(nx=40, ny=60, nz=40, with_walls=True); sc["particles"].pos[:, 1] -= 5.8
    pipe = pipeline(3)
    data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
    pipe.step(data, 4); data.sync(); print("stats", data.stats(), file=sys.stderr)
    pipe.step(data, 8); data.sync()
    p = data.read_particles()
    np.savez(sys.argv[1], pos=p.pos, vel=p.vel, F=p.def_grad, aff=p.cdf_affinity, dp=p.dp_state)
    sys.exit(0)
res = {}
for flag in (0, 16384):
    out = f"/tmp/wpe_{flag}.npz"
    r = subprocess.run([sys.executable, __file__, out], env=dict(os.environ, WGS_DEBUG=str(flag)), check=True, capture_output=True, text=True)
    print([l for l in r.stderr.split("\n") if l.startswith("stats")])
    res[flag] = np.load(out)
a, b = res[0], res[16384]
print(", ".join(f"{f} max|d|={np.abs(a[f].astype(np.float64) - b[f].astype(np.float64)).max():.3e} ndiff={(a[f] != b[f]).sum()}" for f in ("pos", "vel", "F", "aff", "dp")))
