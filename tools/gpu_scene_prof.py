"""Developer utility: one reference-sized scene under rocprofv3 (tools/gpu_scene_kstats.sh): settle, sync, then the substeps."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import pipeline
from wgsparkl_amd import MpmData, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "sand3"
sc = (scenes.config_scene("c4") if name == "c4leg" else scenes.reference_sand3() if name == "sand3" else scenes.reference_sand2() if name == "sand2" else scenes.elastic_block_2d() if name == "c1" else
      scenes.neo_hookean_cube(n_side=100 if name == "stirred" else 64, with_floor=True))
if name == "stirred":       # bench.py's c2_stirred leg
    rel = sc["particles"].pos - sc["particles"].pos.mean(0)
    sc["particles"].vel[:, 0] = 48.0 + 1.5 * rel[:, 2]
    sc["particles"].vel[:, 1] = 48.0
    sc["particles"].vel[:, 2] = 48.0 - 1.5 * rel[:, 0]
pipe = pipeline(sc["particles"].dim)
data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc.get("model", 0))
pipe.step(data, 100 if name != 'stirred' else 5); data.sync()
pipe.step(data, 50 if name in ('stirred', 'c4leg') else 200); data.sync()
print(data.stats())
