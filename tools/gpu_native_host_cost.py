"""Developer utility: host time per substep of wgs_sharded_step (the substep protocol driven from inside the library) and
the device time, for one slab without neighbours and — SELF — for a rank that is its own two neighbours over RCCL (every
ncclSend / ncclRecv group of an interior rank is issued; a 1-GPU box cannot hold a second rank), next to wgs_step on the
same particles as a single domain.  usage: gpu_native_host_cost.py [c2|c5]   (c2: a 1 M-particle slab of the weak-scaling
bar; c5: rank 3 of the 8-slab cut of the 16 M fluid block = 2 M particles)"""
import os, sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from wgsparkl_amd import MpmData, MpmPipeline, scenes
from wgsparkl_amd.sharded import NativeComm, NativeShard, associated_block_x, uniform_material_of
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
pipe = MpmPipeline(0, 3)
def slab_scene():
    if cfg == "c5":
        return scenes.config_scene("c5", 8, 3, "strong")
    sc = scenes.neo_hookean_bar(n_side=100, world=1, rank=0)
    return sc
for self_nb in (False, True):
    sc = slab_scene()
    ps = sc["particles"]
    bx = associated_block_x(ps.pos, sc["cell_width"], 3)
    lo, hi = (int(bx.min()), int(bx.max()) + 2) if self_nb else (-2 ** 31, 2 ** 31 - 1)
    ny = int(round((float(ps.pos[:, 1].max()) - float(ps.pos[:, 1].min())) * 2.0)) + 1
    nz = int(round((float(ps.pos[:, 2].max()) - float(ps.pos[:, 2].min())) * 2.0)) + 1
    face = (ny // 8 + 3) * (nz // 8 + 3)                      # bench.py's capacities
    comm = NativeComm(pipe, None, 0, 1, flags=1 if self_nb else 0)
    data = NativeShard(pipe, sc["params"], ps, sc["global_ids"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], lo, hi, self_nb, self_nb,
                       particle_capacity=int(ps.n * 1.25) + 4096, model=sc["model"], halo_capacity_records=face + face // 2 + 64,
                       migrant_capacity=max(512, (ny * nz) // 32), comm=comm, uniform_material=uniform_material_of(ps))
    data.step(30); data.sync()
    for k in (100, 300, 100):
        t0 = time.perf_counter(); data.step(k); t1 = time.perf_counter(); data.sync(); t2 = time.perf_counter()
        print(f"{cfg} {'self-neighbours over RCCL' if self_nb else 'no neighbours':26s} {k} substeps: host enqueue {1e6 * (t1 - t0) / k:.1f} us/substep, "
              f"total {1e6 * (t2 - t0) / k:.1f} us/substep", flush=True)
    assert data.num_particles() == ps.n
    data.close(); comm.close()
sc = slab_scene() if cfg == "c5" else scenes.neo_hookean_cube(n_side=100, with_floor=True)
d1 = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
pipe.step(d1, 30); d1.sync()
for k in (100, 300):
    t0 = time.perf_counter(); pipe.step(d1, k); t1 = time.perf_counter(); d1.sync(); t2 = time.perf_counter()
    print(f"{cfg} {'single domain (wgs_step)':26s} {k} substeps: host enqueue {1e6 * (t1 - t0) / k:.1f} us/substep, total {1e6 * (t2 - t0) / k:.1f} us/substep")
