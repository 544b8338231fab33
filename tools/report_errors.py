"""Prints the HIP-vs-oracle error table (not a test; used for DESIGN.md numbers).
Usage on the GPU box: python tools/report_errors.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import *
from wgsparkl_amd import scenes
import test_gpu_parity as T

cases = [("cloud3d corotated k=1", T.cloud_scene(), 1), ("cloud3d neo-hookean k=1", T.cloud_scene(model=1), 1),
         ("reference smoke k=3", scenes.reference_smoke_scene(), 3), ("cube24 k=10", scenes.neo_hookean_cube(24), 10),
         ("cube24 k=100", scenes.neo_hookean_cube(24), 100),
         ("2d block k=5", scenes.elastic_block_2d(40, 40, False), 5)]
for name, sc, k in cases:
    D = sc["particles"].dim
    data = run_gpu(sc, k)
    a = run_oracle(sc, k, np.float32)
    b = run_oracle(sc, k, np.float64)
    gg = data.read_grid()
    same = gg[0].shape == grid_of(b)[0].shape and np.array_equal(gg[0], grid_of(b)[0])
    row = [f"{name:26s} nodes_equal={same}"]
    if same:
        row.append(f"gridv gpu={rel_rms(gg[1][:, :D], grid_of(b)[1][:, :D]):.2e} o32={rel_rms(grid_of(a)[1][:, :D], grid_of(b)[1][:, :D]):.2e}")
    got = data.read_particles()
    for f in ("pos", "vel", "def_grad", "affine"):
        row.append(f"{f} gpu={rms(getattr(got, f) - b.arr[f]):.2e} o32={rms(a.arr[f] - b.arr[f]):.2e} scale={rms(b.arr[f]):.2e}")
    print(" | ".join(row))
