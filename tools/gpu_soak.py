"""Developer utility: long runs of the main scene families on the GPU; everything must stay finite and inside the grid."""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import pipeline
from wgsparkl_amd import MpmData, scenes
from golden_cases import dynamic_ball3d, mesh_floor3d, polyline2d, dynamic_ball2d

def run(name, sc, steps, chunk=200):
    dim = sc["particles"].dim
    pipe = pipeline(dim)
    data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc.get("model", 0))
    t0 = time.time()
    done = 0
    while done < steps:
        pipe.step(data, chunk); data.sync(); done += chunk
    ps = data.read_particles()
    st = data.stats()
    ok = np.isfinite(ps.pos).all() and np.isfinite(ps.vel).all() and np.isfinite(ps.def_grad).all() and st["overflow"] == 0
    bodies = data.read_body_poses() if sc["colliders"] else []
    print(f"{name}: {steps} substeps, {ps.n} particles, {time.time()-t0:.1f}s, finite={ok}, y=[{ps.pos[:,1].min():.2f},{ps.pos[:,1].max():.2f}], "
          f"|v|max={np.abs(ps.vel).max():.3f}, blocks={st['num_active_blocks']}, cpic={(ps.cdf_affinity != 0).sum()}"
          + (f", body0 y={bodies[0]['translation'][1]:.3f} vy={bodies[0]['linvel'][1]:.3f}" if bodies else ""))
    assert ok, name

sc = scenes.neo_hookean_cube(n_side=64, with_floor=True); sc["particles"].pos[:, 1] -= 5.0
run("cube on floor", sc, 3000)
sc = scenes.sand_column(nx=60, ny=120, nz=60, with_floor=True); sc["particles"].pos[:, 1] -= 5.8
run("sand column", sc, 2000)
run("dynamic ball 3d", dynamic_ball3d(), 3000)
run("dynamic ball 2d", dynamic_ball2d(), 3000)
run("mesh floor 3d", mesh_floor3d(), 3000)
run("polyline 2d", polyline2d(), 3000)
sc = scenes.corotated_cube_with_paddle(n_side=64)
run("paddle", sc, 2000)
