"""Developer utility: configs[0] (2D elastic block, 10k particles, 100 steps) on the GPU — wall time per substep."""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import pipeline
from wgsparkl_amd import MpmData, scenes
sc = scenes.elastic_block_2d(nx=100, ny=100, with_floor=True)
pipe = pipeline(2)
data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
pipe.step(data, 20); data.sync()
t0 = time.perf_counter(); pipe.step(data, 100); data.sync(); dt = time.perf_counter() - t0
print("C1: %d particles, 100 substeps in %.2f ms = %.1f us/substep = %.2f M particle-steps/s" % (sc["particles"].n, dt * 1e3, dt * 1e4, sc["particles"].n * 100 / dt / 1e6))
