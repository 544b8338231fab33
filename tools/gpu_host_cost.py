"""Developer utility: host time per substep of the Python-driven sharded loop (enqueue only) against the GPU time."""
import os, sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, torch.distributed as dist
from wgsparkl_amd import MpmPipeline, scenes
from wgsparkl_amd.sharded import FixedExchange, GpuShard, RcclExchange, finish_migration, pipelined_substep, substep_phases
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29551", rank=0, world_size=1, device_id=torch.device("cuda", 0))
work = torch.cuda.Stream() if os.environ.get("OWN_STREAM") == "1" else torch.cuda.current_stream()
torch.cuda.set_stream(work)
print("stream handle", work.cuda_stream)
pipe = MpmPipeline(0, 3)
# SELF_NEIGHBOURS=1: the single rank is its own lower and upper neighbour, so that every RCCL call of a real interior
# rank is issued (host cost of the transport; the physics of the interface layers is meaningless)
SELF = os.environ.get("SELF_NEIGHBOURS") == "1"
sc = scenes.neo_hookean_bar(n_side=100, world=1, rank=0)
lo, hi = sc["partition"].block_range(0)
data = GpuShard(pipe, sc["params"], sc["particles"], sc["global_ids"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], lo, hi,
                SELF, SELF, particle_capacity=int(sc["particles"].n * 1.25) + 4096, model=sc["model"], halo_capacity_blocks=int(os.environ.get("HALO_CAP", "450")), migrant_capacity=int(os.environ.get("MIG_CAP", "2048")))
ex = FixedExchange(dist, 0, 1) if os.environ.get("EXCH") == "torch" else RcclExchange(dist, 0, 1)
if SELF:
    ex.lower = ex.upper = 0
def run(k):
    if os.environ.get("PLAIN") == "1":
        for _ in range(k):
            substep_phases(data, ex)
        return
    if os.environ.get("CSTEP") == "1":      # the C++ loop on the same (sharded) data: no Python between substeps
        from wgsparkl_amd import _ffi
        _ffi.check(pipe.lib, pipe.lib.wgs_step(pipe._h, data._h, k, 0))
        return
    p = None
    for _ in range(k):
        p = pipelined_substep(data, ex, p)
    finish_migration(data, p)
run(20); data.sync()
for k in (100, 300, 100):
    t0 = time.perf_counter(); run(k); t1 = time.perf_counter(); data.sync(); t2 = time.perf_counter()
    print(f"{k} substeps: host enqueue {1e6 * (t1 - t0) / k:.1f} us/substep, total {1e6 * (t2 - t0) / k:.1f} us/substep")
if os.environ.get("LONG"):
    n0 = data.num_particles()
    run(int(os.environ["LONG"])); data.sync()
    print(f"after {os.environ['LONG']} more substeps: {data.num_particles()} particles (was {n0})")
    assert data.num_particles() == n0
dist.destroy_process_group()
