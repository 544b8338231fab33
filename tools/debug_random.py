"""Developer utility: details of a fuzz scene where HIP and oracle disagree."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import run_gpu, run_oracle, rel_rms
import test_gpu_parity as T
for seed in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else range(10)):
    sc = T._random_scene(seed)
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    data = run_gpu(sc, k)
    st, st64 = run_oracle(sc, k, np.float32), run_oracle(sc, k, np.float64)
    got = data.read_particles()
    same = got.cdf_affinity == st.arr["cdf_affinity"]
    err = rel_rms(got.vel[same], st64.arr["vel"][same]); err32 = rel_rms(st.arr["vel"][same], st64.arr["vel"][same])
    desc = [(c.shape_type, 'dyn' if any(c.inv_mass) else 'kin') for c in sc["colliders"]]
    print(seed, 'dim', sc["particles"].dim, 'model', sc["model"], 'plastic', bool(sc["particles"].has_plasticity.any()), 'phase', bool(sc["particles"].has_phase.any()), desc, 'vel err %.2e (f32 %.2e)' % (err, err32), 'aff same', same.mean())
    if err > 1e-3:
        d = np.linalg.norm(got.vel - st.arr["vel"], axis=1)
        bad = np.argsort(-d)[:8]
        print('   worst particles', bad, d[bad])
        print('   their affinity', got.cdf_affinity[bad], st.arr["cdf_affinity"][bad])
        print('   phase gpu', got.phase[bad], 'orc', st.arr["phase"][bad])
        print('   F gpu', got.def_grad[bad[0]], 'orc', st.arr["def_grad"][bad[0]])
        print('   vel gpu', got.vel[bad[0]], 'orc', st.arr["vel"][bad[0]], 'pos', got.pos[bad[0]], st.arr["pos"][bad[0]])
