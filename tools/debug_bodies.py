"""Developer utility: HIP vs oracle on the golden scenes with moving / dynamic bodies."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import oracle, pipeline, rel_rms
from wgsparkl_amd import MpmData
from golden_cases import CASES

for name in ('dynamic_ball2d', 'dynamic_ball3d', 'tilted_box2d'):
    make, n = CASES[name]
    sc = make()
    dim = sc['particles'].dim
    if len(sys.argv) > 1: n = int(sys.argv[1])
    args = (sc['particles'], sc['params'], sc['colliders'], sc['cell_width'], sc['grid_capacity'], sc.get('model', 0))
    st = oracle(dim, np.float64).new_state(*args)
    st32 = oracle(dim, np.float32).new_state(*args)
    pipe = pipeline(dim)
    data = MpmData.new(pipe, sc['params'], sc['particles'], *args[2:])
    st.step(n); st32.step(n); pipe.step(data, n); data.sync()
    st.update_world_mass_properties(); st32.update_world_mass_properties()
    got = data.read_body_poses(); ref = st.collider_states(); r32 = st32.collider_states()
    for i in range(len(got)):
        for k in ('translation', 'rotation', 'linvel', 'angvel', 'com'):
            print(name, i, k, 'gpu', got[i][k], 'f64', ref[i][k], 'err gpu %.2e' % np.abs(got[i][k] - ref[i][k]).max(), 'err f32 %.2e' % np.abs(r32[i][k] - ref[i][k]).max())
    gp = data.read_particles()
    same = gp.cdf_affinity == st.arr['cdf_affinity']
    print(name, 'affinity agreement', same.mean(), 'pos rel', rel_rms(gp.pos, st.arr['pos']), 'f32', rel_rms(st32.arr['pos'], st.arr['pos']), 'vel', rel_rms(gp.vel, st.arr['vel']), 'f32', rel_rms(st32.arr['vel'], st.arr['vel']))
