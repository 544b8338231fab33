"""Developer utility: where does the small C5 scene on the floor differ from the oracle?"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import run_gpu, run_oracle, grid_of
from wgsparkl_amd import scenes
def make():
    sc = scenes.fluid_block(40, 24, 24, with_floor=True)
    ps = sc["particles"]
    rng = np.random.default_rng(55)
    ps.vel[:] = rng.normal(0, 0.4, ps.vel.shape).astype(np.float32)
    ps.def_grad[:] += rng.normal(0, 0.03, ps.def_grad.shape).astype(np.float32)
    ps.def_grad[:, [0, 4, 8]] *= np.float32(0.97)
    ps.pos[:, 1] -= 5.7
    return sc
for k in (1, 2, 3, 10):
    sc = make()
    data = run_gpu(sc, k)
    st = run_oracle(sc, k, np.float32)
    cells, vm, dist, aff, closest = data.read_grid()
    oc, omv, odist, oaff, oclosest = st.grid_records()
    assert np.array_equal(cells, oc)
    err = np.abs(vm[:, :3] - omv[:, :3]).max(1)
    bad = np.argsort(-err)[:8]
    print(f"k={k}: max node err {err.max():.3e}, nodes with err > 1e-3: {(err > 1e-3).sum()} of {len(err)}; aff equal {np.array_equal(aff, oaff)}")
    for b in bad[:5]:
        print("   cell", cells[b], "gpu", vm[b], "orc", omv[b], "aff", hex(aff[b]), hex(oaff[b]), "closest", closest[b], oclosest[b])
    got = data.read_particles()
    same = got.cdf_affinity == st.arr["cdf_affinity"]
    perr = np.abs(got.vel - st.arr["vel"]).max(1)
    print("   particles: affinity same", same.mean(), "max vel err", perr.max(), "count > 1e-3:", (perr > 1e-3).sum(),
          " pos y of worst", got.pos[np.argmax(perr)], "cdf dist", got.cdf_dist[np.argmax(perr)], st.arr["cdf_dist"][np.argmax(perr)] if "cdf_dist" in st.arr else None)
