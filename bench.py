#!/usr/bin/env python3
"""bench.py — particle-steps/s of the MLS-MPM substep on MI355X.

Workload (BASELINE.json configs[1]): wgsparkl3d neo-Hookean elastic
cube, 100^3 = 1M particles (8 per cell) in a 128^3-cell domain, fp32, synthetic
lattice + jitter. One "step" = one substep of MpmPipeline::queue_step
(sort -> P2G -> grid update -> fused G2P + particle update), inputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL). Weak scaling: N cubes side
by side form one bar, cut into x-slabs; per substep each rank swaps the partial node sums of
the interface layers and the migrating particles with its two neighbours (point-to-point over
xGMI, no collective on the data path; wgsparkl_amd/sharded.py).
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
    """The oracle (CPU restatement of the reference algorithm, C + OpenMP over the per-node / per-particle loops,
    the sort stays serial) timed on the GPU box's host cores, on a bounded sample of the same workload."""
    import numpy as np
    from oracle.orc import Oracle
    ps = scene["particles"]
    orc = Oracle(3, np.float32, omp=True)
    st = orc.new_state(ps, scene["params"], scene["colliders"], scene["cell_width"], scene["grid_capacity"], scene["model"])
    st.step(1)      # first touch of the arrays, thread pool start-up
    t0 = time.perf_counter()
    st.step(substeps)
    dt = time.perf_counter() - t0
    return ps.n * substeps / dt, dt, orc.num_threads


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
    # WGS_BENCH_FORCE_SHARDED=1: the N > 1 code path (RCCL process group, sharded data, collectives of the harness)
    # with however many ranks were launched, even one — a functional check for 1-GPU boxes
    sharded_path = world > 1 or os.environ.get("WGS_BENCH_FORCE_SHARDED") == "1"
    if sharded_path:
        import torch.distributed as dist
        # WGS_BENCH_ONE_GPU=1: functional test of the N > 1 path on a 1-GPU box (all ranks on cuda:0, gloo)
        one_gpu = os.environ.get("WGS_BENCH_ONE_GPU") == "1"
        if one_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev_index = local_rank if sharded_path else 0

    import numpy as np
    from wgsparkl_amd import MpmData, MpmPipeline, scenes

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    pipe = MpmPipeline(dev_index, 3)
    if not sharded_path:
        scene = scenes.neo_hookean_cube(n_side=args.n_side, with_floor=not args.no_floor)
        ps = scene["particles"]
        n = ps.n
        n_total = n
        data = MpmData.new(pipe, scene["params"], ps, scene["colliders"], scene["cell_width"],
                           scene["grid_capacity"], scene["model"])
        run = lambda k: pipe.step(data, k)          # K substeps enqueued asynchronously
        sync = data.sync
        parallelism = "1 GPU"
    else:
        # Weak scaling: `world` C2 cubes side by side along x form one elastic bar; x-slab domain
        # decomposition, one slab per GPU, halo + migration exchanges over RCCL point-to-point (sharded.py).
        from wgsparkl_amd.sharded import FixedExchange, GpuShard, RcclExchange, finish_migration, pipelined_substep, substep_phases
        scene = scenes.neo_hookean_bar(n_side=args.n_side, world=world, rank=rank)
        if args.no_floor:
            scene["colliders"] = []
        ps = scene["particles"]
        n = ps.n
        n_total = scene["global_particles"]
        lo, hi = scene["partition"].block_range(rank)
        data = GpuShard(pipe, scene["params"], ps, scene["global_ids"], scene["colliders"], scene["cell_width"],
                        scene["grid_capacity"], lo, hi, rank > 0, rank < world - 1,
                        particle_capacity=int(n * 1.25) + 4096, model=scene["model"],
                        # Messages travel at their full capacity (no size handshake), so the capacities are sized from
                        # the workload: a face of the bar touches at most (n_side / 8 + 3)^2 blocks (+ margin), and the
                        # bar falls along y — a handful of particles cross a cut per substep. An overflow is reported
                        # by wgs_sync and by the particle count checked below.
                        halo_capacity_blocks=(args.n_side // 8 + 3) ** 2 + 32, migrant_capacity=512)
        # transport: RCCL called directly (ctypes) unless WGS_EXCHANGE=torch or the process group is not RCCL (the
        # 1-GPU functional mode runs over gloo)
        use_rccl = os.environ.get("WGS_EXCHANGE", "rccl") == "rccl" and dist.get_backend() == "nccl"
        exch = None
        if use_rccl:
            try:
                exch = RcclExchange(dist, rank, world)
                if os.environ.get("WGS_BENCH_FORCE_SHARDED") == "1":
                    exch.selftest()
                exch.neighbour_test()
            except Exception as e:  # noqa: BLE001 — any failure here means "use the torch transport", on every rank
                print(f"[bench rank {rank}] direct RCCL transport unavailable ({e}); using torch.distributed p2p", file=sys.stderr)
                exch = None
            ok = torch.tensor([1 if exch is not None else 0], device=f"cuda:{local_rank}", dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)      # all ranks or none
            if int(ok.item()) == 0:
                exch, use_rccl = None, False
        if exch is None:
            exch = FixedExchange(dist, rank, world)
        transport = "RCCL send/recv (direct)" if use_rccl else "torch.distributed p2p"

        # Both messages of a substep are issued on the substep's own stream, in order (substep_phases). The pipelined
        # order (migration on a side stream while the next substep re-bins its residents, WGS_SHARD_ORDER=pipelined) is
        # SLOWER on this stack: measured with a rank that is its own two neighbours (every RCCL call of an interior
        # rank issued, tests/gpu_host_cost.py), 206 us per substep in order against 230 us pipelined — the two
        # cross-stream dependencies cost more than the overlap gains.
        pipelined = os.environ.get("WGS_SHARD_ORDER", "inorder") == "pipelined"

        def run(k):
            if not pipelined:
                for _ in range(k):
                    substep_phases(data, exch)
                return
            pending = None       # the migration of a substep stays in flight while the next one re-bins its residents
            for _ in range(k):
                pending = pipelined_substep(data, exch, pending)
            finish_migration(data, pending)
        sync = data.sync
        parallelism = f"{world} x-slabs, halo + migration over {transport}"

    run(args.warmup)
    sync()
    barrier()
    t0 = time.perf_counter()
    run(args.steps)                      # exactly K substeps
    sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=f"cuda:{local_rank}", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([data.num_particles()], device=f"cuda:{local_rank}", dtype=torch.int64)
        dist.all_reduce(cnt)
        assert int(cnt.item()) == n_total, f"particles lost in the exchange: {int(cnt.item())} != {n_total}"

    # Per-pass device times (HIP events on the data's own stream) of K more substeps of the local slab.
    k_ts = min(args.steps, 64)
    if not sharded_path:
        pipe.step(data, k_ts, timestamps=True)
        data.sync()
        timings = data.read_timings()
        stats = data.stats()
    else:
        from wgsparkl_amd import _ffi
        import ctypes as C
        _ffi.check(pipe.lib, pipe.lib.wgs_step(pipe._h, data._h, k_ts, 1))   # local slab only (no halo): kernel timing
        data.sync()
        ms = (C.c_float * _ffi.WGS_NUM_PASSES)()
        _ffi.check(pipe.lib, pipe.lib.wgs_read_timings(data._h, ms))
        timings = dict(zip(_ffi.PASS_NAMES, [float(x) for x in ms]))
        st = pipe.T.Stats()
        _ffi.check(pipe.lib, pipe.lib.wgs_get_stats(data._h, C.byref(st)))
        stats = {"num_active_blocks": int(st.num_active_blocks)}
    n_nodes = stats["num_active_blocks"] * 64
    import ctypes as _C
    _ovh = _C.c_float(0.0)
    pipe.lib.wgs_read_timing_overhead(data._h, _C.byref(_ovh))   # cost of one timing mark, measured in the same substeps
    mark_ms = float(_ovh.value)

    if rank == 0:
        value = n_total * args.steps / elapsed
        # one launch between the two marks of the "g2p" pass: event interval minus the cost of the closing mark
        # (two marks recorded back to back in the same substeps); rocprofv3's average duration of k_g2p_update
        # (profiles/) agrees with this, the raw interval is ~5 us longer
        g2p_interval_ms = timings["g2p"] / k_ts
        g2p_ms = max(g2p_interval_ms - mark_ms, 1e-9)
        # fused G2P + particle update, elastic: 160 B per particle + 16 B per active node
        algo_bytes = 160.0 * n + 16.0 * n_nodes
        achieved = algo_bytes / (g2p_ms * 1e-3) / 1e9 if g2p_ms > 0 else 0.0
        traffic = None
        prof = sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r*_pmc_g2p.json")))[-1:] or [""]
        prof = prof[0]
        if os.path.exists(prof) and args.n_side == 100:
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
                       "particles_per_gpu": n_total // world, "global_particles": n_total,
                       "active_blocks_rank0": stats["num_active_blocks"], "parallelism": parallelism},
            "roofline": {"bound": "hbm", "kernel": "k_g2p_pair (fused G2P + particle update; collider simulations run both bodies in this launch)" if not args.no_floor else "k_g2p_update (fused G2P + particle update)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": g2p_ms, "event_interval_ms": g2p_interval_ms, "event_mark_ms": mark_ms},
            "pass_ms_per_step": {k: v / k_ts for k, v in timings.items()},
        }
        if not args.no_cpu_baseline and world == 1:
            sub = 20
            v, secs, threads = cpu_baseline(scene, sub)
            out["cpu_baseline"] = {"value": v, "unit": "particle-steps/s", "cores": threads, "kind": "port",
                                   "sample": f"{sub} substeps of the same {n}-particle workload, C + OpenMP oracle "
                                             f"(CPU restatement of the reference WGSL algorithm, {threads} threads; the "
                                             f"hash-grid sort is serial), {secs:.1f} s; "
                                             "reference WGSL via wgpu+lavapipe: unavailable (no cargo/rustc/Vulkan ICD)"}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
