"""Developer utility: difference between the paired and the separate near-collider launches (WGS_DEBUG bits 4096 / 8192)."""
import os, subprocess, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
if len(sys.argv) > 1:
    from helpers import pipeline
    from wgsparkl_amd import MpmData, scenes
    sc = scenes.corotated_cube_with_paddle(n_side=64)
    pipe = pipeline(3)
    data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
    pipe.step(data, 4); data.sync(); pipe.step(data, int(sys.argv[2])); data.sync()
    p = data.read_particles()
    np.savez(sys.argv[1], pos=p.pos, vel=p.vel, F=p.def_grad, aff=p.cdf_affinity)
    sys.exit(0)
res = {}
for flag in (0, 4096, 8192, 12288):
    for k in (1, 4):
        out = f"/tmp/pp_{flag}_{k}.npz"
        subprocess.run([sys.executable, __file__, out, str(k)], env=dict(os.environ, WGS_DEBUG=str(flag)), check=True, stderr=subprocess.DEVNULL)
        res[flag, k] = np.load(out)
for k in (1, 4):
    for flag in (4096, 8192, 12288):
        a, b = res[0, k], res[flag, k]
        print(f"k={k} flag={flag}: " + ", ".join(f"{f} max|d|={np.abs(a[f].astype(np.float64) - b[f].astype(np.float64)).max():.3e} ndiff={(a[f] != b[f]).sum()}" for f in ("pos", "vel", "F", "aff")))
