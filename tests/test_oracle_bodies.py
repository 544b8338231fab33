"""-m "not gpu": the rigid-body passes of the oracle (rigid_impulses.wgsl:95-149 restated; wgrapier's
Body::* functions are third party and not on disk, so these are invariants of the published rapier
algorithms they restate, plus the coupling behaviour the reference's P2G defines)."""
import dataclasses

import numpy as np
import pytest

from golden_cases import dynamic_ball2d, dynamic_ball3d
from wgsparkl_amd import scenes
from wgsparkl_amd.solver import Collider


def _state(oracle_libs, sc, dtype=np.float64):
    ps = sc["particles"]
    return oracle_libs.Oracle(ps.dim, dtype).new_state(ps, sc["params"], sc["colliders"], sc["cell_width"],
                                                       sc["grid_capacity"], sc.get("model", 0))


def _far_scene(dim, colliders):
    sc = scenes.neo_hookean_cube(n_side=4) if dim == 3 else scenes.elastic_block_2d(nx=8, ny=8, with_floor=False)
    sc["colliders"] = colliders
    return sc


@pytest.mark.parametrize("dim", [2, 3])
def test_kinematic_body_follows_its_velocity(oracle_libs, dim):
    """integrateVelocity: the centre of mass translates by v dt per substep, the body turns by |w| dt about it."""
    v = (0.7, -0.2, 0.4)[:dim]
    if dim == 3:
        c = Collider.cuboid((1, 1, 1), (60.0, 60.0, 60.0), linvel=v, angvel=(0.0, 0.0, 1.5), com=(61.0, 60.5, 60.0))
    else:
        c = Collider.cuboid((1, 1), (60.0, 60.0), linvel=v + (0.0,), angvel=(1.5,), com=(61.0, 60.5))
    sc = _far_scene(dim, [c])
    st = _state(oracle_libs, sc)
    k, dt = 50, sc["params"].dt
    st.step(k)
    st.update_world_mass_properties()
    b = st.collider_states()[0]
    com0 = np.array(c.com[:dim])
    assert np.allclose(b["com"], com0 + np.array(v) * k * dt, atol=1e-12)
    ang = 1.5 * k * dt
    if dim == 3:
        assert np.allclose(b["rotation"], [0.0, 0.0, np.sin(ang / 2), np.cos(ang / 2)], atol=1e-12)
    else:
        assert np.allclose(b["rotation"], [np.cos(ang), np.sin(ang)], atol=1e-12)
    arm0 = np.array(c.translation[:dim]) - com0          # the body origin turns about the centre of mass
    rot = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    arm = arm0.copy()
    arm[:2] = rot @ arm0[:2]
    assert np.allclose(b["translation"], b["com"] + arm, atol=1e-10)
    assert np.allclose(b["linvel"], v) and np.allclose(b["angvel"][-1], 1.5)   # no gravity on kinematic bodies


@pytest.mark.parametrize("dim", [2, 3])
def test_dynamic_body_free_fall(oracle_libs, dim):
    """rigid_impulses.wgsl:127-131: the pose is integrated with the velocity BEFORE gravity is added to it."""
    tr = (60.0,) * dim
    c = Collider.ball(1.0, tr, linvel=(0.0, 1.0, 0.0)).with_density(10.0, dim)
    sc = _far_scene(dim, [c])
    st = _state(oracle_libs, sc)
    k, dt = 40, sc["params"].dt
    g = np.array(sc["params"].gravity)
    st.step(k)
    b = st.collider_states()[0]
    v0 = np.array((0.0, 1.0, 0.0)[:dim])
    assert np.allclose(b["linvel"], v0 + g * k * dt, atol=1e-12)
    x = np.array(tr) + sum((v0 + g * j * dt) * dt for j in range(k))
    assert np.allclose(b["translation"], x, atol=1e-11)
    assert np.all(st.impulses() == 0)


def test_velocity_caps_apply_only_when_pushed(oracle_libs):
    """rigid_impulses.wgsl:112-125: |w| is capped to 1 and |v| to 0.1 h / dt only in substeps with a non-zero
    impulse — a spinning kinematic box keeps w = 2 in free space and drops to 1 once it touches particles."""
    free = _far_scene(2, [Collider.cuboid((1.0, 1.0), (60.0, 60.0), angvel=(2.0,))])
    st = _state(oracle_libs, free)
    st.step(20)
    assert st.collider_states()[0]["angvel"][0] == 2.0
    sc = dynamic_ball2d()
    st = _state(oracle_libs, sc)
    st.step(30)
    assert st.collider_states()[1]["angvel"][0] == 1.0


@pytest.mark.parametrize("make", [dynamic_ball2d, dynamic_ball3d])
def test_two_way_coupling_pushes_back(oracle_libs, make):
    """p2g.wgsl:200-228 + rigid_impulses.wgsl:95-111: the dynamic ball is decelerated by the block it lands on
    (a kinematic twin is not), and the heavier the ball the less it is decelerated."""
    sc = make()
    dim = sc["particles"].dim
    k, dt = 100, sc["params"].dt
    g = sc["params"].gravity[1]
    st = _state(oracle_libs, sc)
    st.step(k)
    vy = st.collider_states()[0]["linvel"][1]
    v0 = sc["colliders"][0].linvel[1]
    free_fall = v0 + g * k * dt
    assert vy > free_fall + 0.5, (vy, free_fall)
    heavy = dataclasses.replace(sc["colliders"][0]).with_density(5000.0, dim)
    sc2 = dict(sc, colliders=[heavy, sc["colliders"][1]])
    st2 = _state(oracle_libs, sc2)
    st2.step(k)
    assert free_fall < st2.collider_states()[0]["linvel"][1] < vy
    kin = dataclasses.replace(sc["colliders"][0], inv_mass=(0.0,) * 3, inv_inertia_local=(0.0,) * 9)
    st3 = _state(oracle_libs, dict(sc, colliders=[kin, sc["colliders"][1]]))
    st3.step(k)
    assert st3.collider_states()[0]["linvel"][1] == pytest.approx(v0)


@pytest.mark.parametrize("make", [dynamic_ball2d, dynamic_ball3d])
def test_fp32_and_fp64_oracles_agree_on_bodies(oracle_libs, make):
    sc = make()
    a, b = _state(oracle_libs, sc, np.float32), _state(oracle_libs, sc, np.float64)
    a.step(100); b.step(100)
    for x, y in zip(a.collider_states(), b.collider_states()):
        for key in ("rotation", "translation", "linvel", "angvel"):
            assert np.allclose(x[key], y[key], rtol=0.0, atol=2e-4), key


def test_fixed_point_impulses(oracle_libs):
    """rigid_impulses.wgsl:50-58: impulses are accumulated as i32(x * 1e5), one conversion per NODE total
    (p2g.wgsl:142-155); the accumulators are reset by integrate_bodies."""
    sc = dynamic_ball2d()
    st = _state(oracle_libs, sc)
    for _ in range(40):
        st.update_world_mass_properties()
        st.sort(); st.grid_update_cdf(); st.g2p_cdf(); st.p2g()
        imp = st.impulses()
        st.grid_update(); st.g2p(); st.particle_update()
        before = st.collider_states()[0]["linvel"].copy()
        st.integrate_bodies()
        assert np.all(st.impulses() == 0)
        if np.any(imp[:3] != 0):
            dv = st.collider_states()[0]["linvel"] - before - np.array(sc["params"].gravity) * sc["params"].dt
            inv_mass = sc["colliders"][0].inv_mass[0]
            assert np.allclose(dv, imp[:2] / 1e5 * inv_mass, atol=1e-9)
            return
    pytest.fail("the ball never touched the block")
