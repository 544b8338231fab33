"""Developer utility: block statistics of a bench configuration after a few substeps."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import pipeline
from wgsparkl_amd import MpmData, scenes
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
sc = scenes.config_scene(cfg, n_side=int(sys.argv[2]) if len(sys.argv) > 2 else None)
pipe = pipeline(3)
data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
pipe.step(data, 20); data.sync()
print(data.stats())
