import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import *
from golden_cases import CASES
name = sys.argv[1] if len(sys.argv) > 1 else "floor3d"
make, k = CASES[name]
for steps in (1, 2, 3):
    sc = make()
    data = run_gpu(sc, steps)
    st = run_oracle(sc, steps, np.float32)
    got = data.read_particles()
    cells, vm, dist, aff, closest = data.read_grid()
    oc, ovm, odist, oaff, oclosest = st.grid_records()
    print("steps", steps, "cells eq", np.array_equal(cells, oc), "node aff eq", np.array_equal(aff, oaff), (aff != oaff).sum(),
          "closest eq", np.array_equal(closest, oclosest), "dist maxerr", np.abs(dist - odist)[oaff != 0].max() if (oaff != 0).any() else None)
    same = got.cdf_affinity == st.arr["cdf_affinity"]
    print("   particle aff agree", same.mean(), "nonzero gpu", (got.cdf_affinity != 0).sum(), "orc", (st.arr["cdf_affinity"] != 0).sum())
    bad = np.nonzero(~same)[0][:5]
    for i in bad:
        print("   i", i, "gpu aff", hex(got.cdf_affinity[i]), "orc", hex(st.arr["cdf_affinity"][i]), "gpu n/d", got.cdf_normal[i], got.cdf_dist[i], "orc n/d", st.arr["cdf_normal"][i], st.arr["cdf_dist"][i], "pos", got.pos[i])
    print("   vel err", rel_rms(got.vel, st.arr["vel"]), "gridv err", rel_rms(vm[:, :got.dim], ovm[:, :got.dim]) if cells.shape == oc.shape else None)
