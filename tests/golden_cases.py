"""Seeded scenes shared by tests/golden/make_golden.py (which stores the oracle's answers) and the
tests that compare against the stored answers."""
import numpy as np

from wgsparkl_amd import scenes
from wgsparkl_amd.models import DruckerPrager, ParticlePhase
from wgsparkl_amd.solver import Collider, SimulationParams


def cloud3d():
    ps = scenes.random_cloud(1500, dim=3, seed=21, phase=ParticlePhase(1.0, -1.0))
    return dict(particles=ps, params=SimulationParams((0.0, -9.81, 0.0), 1e-3), colliders=[], cell_width=1.0,
                grid_capacity=2048, model=0)


def cloud2d():
    ps = scenes.random_cloud(1200, dim=2, seed=22, phase=ParticlePhase(1.0, -1.0))
    return dict(particles=ps, params=SimulationParams((0.0, -9.81), 1e-3), colliders=[], cell_width=1.0,
                grid_capacity=512, model=1)


def sand3d():
    ps = scenes.random_cloud(1200, dim=3, seed=23, young=1e6, plasticity=DruckerPrager.new(1e6, 0.25), phase=None)
    return dict(particles=ps, params=SimulationParams((0.0, -9.81, 0.0), 5e-4), colliders=[], cell_width=1.0,
                grid_capacity=2048, model=0)


def floor3d():
    sc = scenes.neo_hookean_cube(n_side=10, with_floor=True)
    sc["particles"].pos[:, 1] -= 5.9
    sc["particles"].vel[:, 1] = -3.0
    sc["grid_capacity"] = 1024
    return sc


def tilted_box2d():
    sc = scenes.elastic_block_2d(nx=24, ny=24, with_floor=False)
    sc["particles"].pos[:, 1] -= 3.8
    sc["particles"].vel[:, 1] = -2.0
    sc["colliders"] = [Collider.cuboid((50.0, 1.0), (10.0, 1.2), rotation=(0.15,)),
                       Collider.ball(2.0, (14.0, 6.0), linvel=(0.5, 0.0, 0.0), angvel=(1.0,))]
    return sc


def dynamic_ball2d():
    """Two-way coupling: a dynamic ball dropped on an elastic block next to a kinematic, spinning box."""
    sc = scenes.elastic_block_2d(nx=16, ny=16, with_floor=False)
    sc["colliders"] = [Collider.ball(1.5, (11.0, 16.6), linvel=(0.3, -2.0, 0.0)).with_density(500.0, 2),
                       Collider.cuboid((1.0, 1.0), (5.5, 10.0), linvel=(0.5, 0.0, 0.0), angvel=(2.0,), com=(5.0, 10.0))]
    return sc


def dynamic_ball3d():
    sc = scenes.neo_hookean_cube(n_side=8, with_floor=False)
    sc["colliders"] = [Collider.ball(1.5, (21.8, 13.3, 22.1), linvel=(0.2, -3.0, 0.0)).with_density(500.0, 3),
                       Collider.cuboid((1.0, 1.0, 1.0), (18.9, 10.0, 22.0), linvel=(0.5, 0.0, 0.0), angvel=(0.0, 0.0, 0.8),
                                       com=(18.7, 10.0, 22.0))]
    return sc


def mesh_floor3d():
    """Mesh colliders (rigid particles, p2g_cdf): an elastic cube lands on an undulating heightfield while a
    kinematic trimesh wedge pushes into its side."""
    sc = scenes.neo_hookean_cube(n_side=8, with_floor=False)
    sc["particles"].pos[:, 1] -= 0.4
    sc["particles"].vel[:, 1] = -3.0
    ii, jj = np.meshgrid(np.arange(7), np.arange(7), indexing="ij")
    heights = (0.25 * np.sin(0.9 * ii) * np.cos(0.7 * jj)).astype(np.float32)
    wedge_v = np.array([[0, -2, -2], [0, -2, 2], [0, 2, 2], [0, 2, -2], [-2, 0, 0]], np.float32)
    wedge_i = np.array([[0, 1, 2], [0, 2, 3], [4, 1, 0], [4, 2, 1], [4, 3, 2], [4, 0, 3]], np.uint32)
    sc["colliders"] = [Collider.heightfield(heights, (18.0, 1.0, 18.0), (22.13, 7.2, 21.81)),
                       Collider.trimesh(wedge_v, wedge_i, (19.9, 9.5, 22.0), linvel=(0.6, 0.0, 0.0), angvel=(0.0, 0.3, 0.0))]
    sc["grid_capacity"] = 1024
    return sc


def tied_mesh3d():
    """Mesh colliders whose triangles have TIED longest edges (particle3d.rs:348-362: bc, then ca, else ab): an elastic cube
    lands on a regular octahedron (every face equilateral) beside a square pyramid (isosceles sides) that moves into it."""
    sc = scenes.neo_hookean_cube(n_side=8, with_floor=False)
    sc["particles"].vel[:, 1] = -3.0
    octa_v = np.array([[3, 0, 0], [-3, 0, 0], [0, 3, 0], [0, -3, 0], [0, 0, 3], [0, 0, -3]], np.float32)
    octa_i = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], np.uint32)
    pyr_v = np.array([[-2, 0, -2], [2, 0, -2], [2, 0, 2], [-2, 0, 2], [0, 3, 0]], np.float32)
    pyr_i = np.array([[0, 1, 4], [1, 2, 4], [2, 3, 4], [3, 0, 4], [0, 2, 1], [0, 3, 2]], np.uint32)
    x0, z0 = (float(sc["particles"].pos[:, k].mean()) for k in (0, 2))
    y0 = float(sc["particles"].pos[:, 1].min())
    sc["colliders"] = [Collider.trimesh(octa_v, octa_i, (x0 + 0.37, y0 - 3.3, z0 - 0.21)),
                       Collider.trimesh(pyr_v, pyr_i, (x0 - 4.4, y0 - 0.6, z0 + 0.3), linvel=(0.7, 0.0, 0.0), angvel=(0.0, 0.4, 0.0))]
    sc["grid_capacity"] = 1024
    return sc


def polyline2d():
    """2D polyline collider: a V-shaped ground under an elastic block."""
    sc = scenes.elastic_block_2d(nx=20, ny=16, with_floor=False)
    sc["particles"].vel[:, 1] = -6.0
    verts = np.array([[-14.0, 3.0], [-4.0, 0.4], [0.0, 0.0], [5.0, 0.6], [14.0, 3.5]], np.float32)
    segs = np.array([[0, 1], [1, 2], [2, 3], [3, 4]], np.uint32)
    x0 = float(sc["particles"].pos[:, 0].mean())
    y0 = float(sc["particles"].pos[:, 1].min())
    sc["colliders"] = [Collider.polyline(verts, segs, (x0, y0 - 0.9))]
    return sc


CASES = {"cloud3d": (cloud3d, 3), "cloud2d": (cloud2d, 3), "sand3d": (sand3d, 2), "floor3d": (floor3d, 20),
         "tilted_box2d": (tilted_box2d, 20), "dynamic_ball2d": (dynamic_ball2d, 120),
         "dynamic_ball3d": (dynamic_ball3d, 120), "mesh_floor3d": (mesh_floor3d, 80), "polyline2d": (polyline2d, 140),
         "tied_mesh3d": (tied_mesh3d, 80)}
