"""Developer experiment: K substeps enqueued on the stream vs captured into one hipGraph (WGS_GRAPH=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgsparkl_amd import MpmData, MpmPipeline, scenes
K = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for name, dim, sc in (("C2 floor", 3, scenes.neo_hookean_cube(100, with_floor=True)), ("C1 2D", 2, scenes.elastic_block_2d())):
    pipe = MpmPipeline(0, dim)
    data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
    pipe.step(data, 20); data.sync()
    for rep in range(3):
        t0 = time.perf_counter(); pipe.step(data, K); t1 = time.perf_counter(); data.sync(); t2 = time.perf_counter()
        print(f"{name} graph={os.environ.get('WGS_GRAPH')} K={K}: enqueue {1e6*(t1-t0)/K:.1f} us/substep, total {1e6*(t2-t0)/K:.1f} us/substep", flush=True)
