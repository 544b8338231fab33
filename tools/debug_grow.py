"""Developer utility: the grid-growth scenarios one by one."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import pipeline
from wgsparkl_amd import MpmData, _ffi, scenes
from wgsparkl_amd.solver import SimulationParams
def cube():
    sc = scenes.neo_hookean_cube(n_side=8)
    ps = sc["particles"]; c = ps.pos.mean(0)
    ps.vel[:] = ((ps.pos - c) * 25.0).astype(np.float32); ps.lambda_[:] = 1.0; ps.mu[:] = 1.0
    sc["params"] = SimulationParams(gravity=(0.0, 0.0, 0.0), dt=sc["params"].dt)
    return sc
pipe = pipeline(3)
cap, grow = int(sys.argv[1]), int(sys.argv[2])
sc = cube()
data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], cap, sc["model"])
_ffi.check(pipe.lib, pipe.lib.wgs_set_grid_growth(data._h, grow))
for f in range(30):
    pipe.step(data, 10)
    if len(sys.argv) > 3:
        try:
            data.sync()
        except Exception as e:
            print("frame", f, "error", e); break
        print("frame", f, data.stats(), flush=True)
try:
    data.sync(); print("final", data.stats())
except Exception as e:
    print("final error:", e)
