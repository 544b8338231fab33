#!/usr/bin/env python3
"""bench.py — particle-steps/s of the MLS-MPM substep on MI355X.

Workload (BASELINE.json configs[1]): wgsparkl3d neo-Hookean elastic
cube, 100^3 = 1M particles (8 per cell) in a 128^3-cell domain, fp32, synthetic
lattice + jitter. One "step" = one substep of MpmPipeline::queue_step
(sort -> P2G -> grid update -> fused G2P + particle update), inputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU (torch.distributed / RCCL for the barrier + max over
ranks). Round 1 runs one independent slab per rank (weak scaling, no exchange yet;
DESIGN.md §7 has the halo design).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def cpu_baseline(scene, substeps):
    """The oracle (CPU restatement of the reference algorithm, scalar C, 1 thread) timed on
    the GPU box's host cores, on a bounded sample of the same workload."""
    import numpy as np
    from oracle.orc import Oracle
    ps = scene["particles"]
    st = Oracle(3, np.float32).new_state(ps, scene["params"], scene["colliders"], scene["cell_width"],
                                         scene["grid_capacity"], scene["model"])
    t0 = time.perf_counter()
    st.step(substeps)
    dt = time.perf_counter() - t0
    return ps.n * substeps / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n-side", type=int, default=100, help="particles per cube edge (100 -> 1M, the named config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-floor", action="store_true", help="drop the floor cuboid of SURVEY 8d C2 (no CPIC passes)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev_index = local_rank if world > 1 else 0

    import numpy as np
    from wgsparkl_amd import MpmData, MpmPipeline, scenes

    scene = scenes.neo_hookean_cube(n_side=args.n_side, with_floor=not args.no_floor)
    ps = scene["particles"]
    n = ps.n
    pipe = MpmPipeline(dev_index, 3)
    data = MpmData.new(pipe, scene["params"], ps, scene["colliders"], scene["cell_width"],
                       scene["grid_capacity"], scene["model"])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    pipe.step(data, args.warmup)
    data.sync()
    barrier()
    t0 = time.perf_counter()
    pipe.step(data, args.steps)          # exactly K substeps, enqueued asynchronously
    data.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=f"cuda:{local_rank}", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Per-pass device times of the same K substeps, HIP events on the data's own stream.
    k_ts = min(args.steps, 64)
    pipe.step(data, k_ts, timestamps=True)
    data.sync()
    timings = data.read_timings()
    stats = data.stats()
    n_nodes = stats["num_active_blocks"] * 64

    if rank == 0:
        value = n * world * args.steps / elapsed
        g2p_ms = timings["g2p"] / k_ts
        # fused G2P + particle update, elastic: 160 B per particle + 16 B per active node
        algo_bytes = 160.0 * n + 16.0 * n_nodes
        achieved = algo_bytes / (g2p_ms * 1e-3) / 1e9 if g2p_ms > 0 else 0.0
        traffic = None
        prof = sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r*_pmc_g2p.json")))[-1:] or [""]
        prof = prof[0]
        if os.path.exists(prof):
            try:
                traffic = json.load(open(prof)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "particle-steps/sec", "value": value, "unit": "particle-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"wgsparkl3d neo-Hookean elastic cube, {n} particles/GPU, 128^3-cell domain, "
                                   f"h=1, dt=1/1200, 8 particles/cell, " + ("no collider" if args.no_floor else "floor cuboid (CPIC passes on)"),
                       "particles_per_gpu": n, "active_blocks": stats["num_active_blocks"],
                       "parallelism": "1 GPU" if world == 1 else f"{world} independent slabs (no halo exchange yet)"},
            "roofline": {"bound": "hbm", "kernel": "k_g2p_update (fused G2P + particle update)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": g2p_ms},
            "pass_ms_per_step": {k: v / k_ts for k, v in timings.items()},
        }
        if not args.no_cpu_baseline:
            sub = 2
            v, secs = cpu_baseline(scene, sub)
            out["cpu_baseline"] = {"value": v, "unit": "particle-steps/s", "cores": 1, "kind": "port",
                                   "sample": f"{sub} substeps of the same {n}-particle workload, scalar C oracle "
                                             f"(CPU restatement of the reference WGSL algorithm), {secs:.1f} s; "
                                             "reference WGSL via wgpu+lavapipe: unavailable (no cargo/rustc/Vulkan ICD)"}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
