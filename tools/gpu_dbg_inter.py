"""developer utility: several wgs_data of the SAME scene stepped interleaved on their own streams (kernels of different data overlap
on the device): do they stay bit-identical? usage: gpu_dbg_inter.py SEED NDATA REPS SUBSTEPS BITS..."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["WGS_REHASH_PERIOD"] = "64"
from gpu_common import _random_scene
from helpers import pipeline
from wgsparkl_amd import MpmData
seed, nd, reps, ksub = (int(x) for x in sys.argv[1:5])
for bits in sys.argv[5:]:
    if int(bits): os.environ["WGS_DEBUG"] = bits
    else: os.environ.pop("WGS_DEBUG", None)
    bad_runs, first_bad = 0, []
    for r in range(reps):
        datas = []
        for i in range(nd):
            sc = _random_scene(seed)
            pipe = pipeline(sc["particles"].dim)
            datas.append(MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"]))
        bad = None
        for k in range(ksub):
            for dd in datas: pipe.step(dd, 1)
            for dd in datas: dd.sync()
            ps = [dd.read_particles().pos for dd in datas]
            if any(not np.array_equal(ps[0], p) for p in ps[1:]):
                bad = k + 1
                break
        if bad: bad_runs += 1; first_bad.append(bad)
        for dd in datas: dd.close()
    print("seed", seed, "datas", nd, "dbg", bits, "reps", reps, "runs that diverged", bad_runs, "at substeps", first_bad, flush=True)
