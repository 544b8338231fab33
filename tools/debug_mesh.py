"""Developer utility: HIP vs oracle node cdf on the mesh-collider golden scenes."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import run_gpu, run_oracle
from golden_cases import CASES
for name in ('mesh_floor3d', 'polyline2d'):
    make, k = CASES[name]
    if len(sys.argv) > 1: k = int(sys.argv[1])
    sc = make()
    data = run_gpu(sc, k)
    s32, s64 = run_oracle(sc, k, np.float32), run_oracle(sc, k, np.float64)
    cells, vm, dist, aff, closest = data.read_grid()
    for nm, st in (('f32', s32), ('f64', s64)):
        oc, mv, odist, oaff, oclosest = st.grid_records()
        same_cells = cells.shape == oc.shape and np.array_equal(cells, oc)
        print(name, nm, 'cells equal', same_cells)
        if same_cells:
            bad = np.nonzero(aff != oaff)[0]
            print('   aff mismatches', len(bad), 'closest mismatches', int((closest != oclosest).sum()), 'dist max err', np.abs(np.where(oaff != 0, dist - odist, 0)).max())
            for b in bad[:6]:
                print('     cell', cells[b], 'gpu %08x' % aff[b], 'orc %08x' % oaff[b], 'dist', dist[b], odist[b])
