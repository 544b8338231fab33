"""Developer utility: C3 (4 M Drucker-Prager sand) free fall and standing between the floor and four walls."""
import sys, time; sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import pipeline
from wgsparkl_amd import MpmData, scenes
for name, sc, drop in (("free fall", scenes.sand_column(), 0.0), ("floor + walls", scenes.sand_column(with_walls=True), 5.8)):
    sc["particles"].pos[:, 1] -= drop
    pipe = pipeline(3)
    d = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
    pipe.step(d, 20); d.sync()
    t0 = time.perf_counter(); pipe.step(d, 50); d.sync(); t1 = time.perf_counter()
    pipe.step(d, 16, timestamps=True); d.sync()
    print(f"C3 {name}: {1e6 * (t1 - t0) / 50:.1f} us/substep", {k: round(v / 16 * 1e3, 1) for k, v in d.read_timings().items() if v > 0.08},
          "near-collider blocks", d.stats()["num_near_collider_blocks"], flush=True)
    del d
