"""developer utility: how much do the launches of a substep gain from running BESIDE each other? Two copies of a bench configuration on their own
streams, stepped together, against one copy alone (the kernels of different data overlap on the device; a launch of one fills what the other leaves idle).
usage: gpu_overlap_probe.py [c2|c3|c5|sand3|small] [substeps] (sand3 / small = scenes that leave most of the chip idle: do streams overlap at all?)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import pipeline
from wgsparkl_amd import MpmData, scenes
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
ksub = int(sys.argv[2]) if len(sys.argv) > 2 else 400
pipe = pipeline(3)
def make():
    sc = scenes.reference_sand3() if cfg == "sand3" else scenes.neo_hookean_cube(n_side=32, with_floor=True) if cfg == "small" else scenes.config_scene(cfg)
    return MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
datas = [make() for _ in range(int(os.environ.get("NDATA", "2")))]
for d in datas: pipe.step(d, 60); d.sync()
def run(ds, chunk=10):
    for d in ds: d.sync()
    t0 = time.perf_counter()
    for _ in range(ksub // chunk):
        for d in ds: pipe.step(d, chunk)
    for d in ds: d.sync()
    return (time.perf_counter() - t0) * 1e6 / ksub
for rep in range(3):
    alone = run(datas[:1])
    both = run(datas)
    print(f"{cfg}: one copy {alone:.1f} us/substep; {len(datas)} copies together {both:.1f} us per substep of all = {both / len(datas):.1f} per copy ({len(datas) * alone / both:.2f}x the throughput of one after the other)", flush=True)
