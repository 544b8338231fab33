"""Developer utility: substep time of a scene with many near-collider blocks (cube resting on the floor + paddle)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgsparkl_amd import MpmData, MpmPipeline, scenes
n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sc = scenes.corotated_cube_with_paddle(n_side=n_side)
pipe = MpmPipeline(0, 3)
data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
pipe.step(data, 30); data.sync()
for rep in range(int(os.environ.get('REPS', '3'))):
    t0 = time.perf_counter(); pipe.step(data, 200); data.sync(); t1 = time.perf_counter()
    print(f"WGS_DEBUG={os.environ.get('WGS_DEBUG')} n_side={n_side}: {1e6*(t1-t0)/200:.1f} us/substep", flush=True)
if not os.environ.get('NO_TS'):
  pipe.step(data, 32, timestamps=True); data.sync()
  print({k: round(v / 32 * 1e3, 1) for k, v in data.read_timings().items()}, data.stats())
