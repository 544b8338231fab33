"""Developer utility: substep time and per-pass timings of the main scene families at sizes where they matter
(looking for passes that are out of proportion, like the 290 us of body-impulse atomics this found)."""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import pipeline
from wgsparkl_amd import MpmData, scenes
from wgsparkl_amd.solver import Collider

def run(name, sc, warm=60, steps=100):
    dim = sc["particles"].dim
    pipe = pipeline(dim)
    data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc.get("model", 0))
    pipe.step(data, warm); data.sync()
    t0 = time.perf_counter(); pipe.step(data, steps); data.sync(); t1 = time.perf_counter()
    pipe.step(data, 32, timestamps=True); data.sync()
    tm = {k: round(v / 32 * 1e3, 1) for k, v in data.read_timings().items() if v > 0}
    st = data.stats()
    print(f"{name}: {sc['particles'].n} particles, {st['num_active_blocks']} blocks, {1e6*(t1-t0)/steps:.1f} us/substep  {tm}", flush=True)

which = sys.argv[1:] or ["c1", "c3", "mesh", "ball", "sandfloor"]
if "c1" in which:
    run("C1 2D elastic block + floor", scenes.elastic_block_2d())
if "c3" in which:
    run("C3 sand column 4M, free fall", scenes.sand_column())
if "sandfloor" in which:
    sc = scenes.sand_column(nx=100, ny=100, nz=100, with_floor=True); sc["particles"].pos[:, 1] -= 5.8
    run("sand 1M resting on the floor", sc, warm=200)
if "mesh" in which:
    sc = scenes.neo_hookean_cube(n_side=64, with_floor=False)
    sc["particles"].pos[:, 1] -= 0.4
    sc["particles"].vel[:, 1] = -3.0
    ii, jj = np.meshgrid(np.arange(25), np.arange(25), indexing="ij")
    heights = (0.25 * np.sin(0.9 * ii) * np.cos(0.7 * jj)).astype(np.float32)
    sc["colliders"] = [Collider.heightfield(heights, (60.0, 1.0, 60.0), (36.13, 7.2, 35.81))]
    run("mesh heightfield under a 262k cube", sc, warm=100)
if "ball" in which:
    sc = scenes.neo_hookean_cube(n_side=64, with_floor=True)
    sc["particles"].pos[:, 1] -= 5.6
    sc["colliders"].append(Collider.ball(4.0, (36.0, 40.0, 36.0), linvel=(0.0, -8.0, 0.0)).with_density(800.0, 3))
    run("dynamic ball on a 262k cube on the floor", sc, warm=300)
