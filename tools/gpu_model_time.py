"""developer utility: a bench configuration with another constitutive model (0 corotated, 1 neo-Hookean), event-free wall time per substep.
usage: gpu_model_time.py c3 1 [substeps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import pipeline
from wgsparkl_amd import MpmData, scenes
cfg, model = sys.argv[1], int(sys.argv[2])
ksub = int(sys.argv[3]) if len(sys.argv) > 3 else 100
sc = scenes.config_scene(cfg)
pipe = pipeline(3)
d = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], model)
pipe.step(d, 40); d.sync()
for rep in range(3):
    t0 = time.perf_counter(); pipe.step(d, ksub); d.sync()
    print(f"{cfg} model {model}: {(time.perf_counter() - t0) * 1e6 / ksub:.1f} us/substep", flush=True)
