#!/usr/bin/env python3
"""bench.py — particle-steps/s of the MLS-MPM substep on MI355X.

Default workload (BASELINE.json configs[1]): wgsparkl3d neo-Hookean elastic cube, 100^3 = 1M
particles (8 per cell) in a 128^3-cell domain, floor cuboid, fp32, synthetic lattice + jitter. One "step" = one
substep of MpmPipeline::queue_step (sort -> P2G -> grid update -> fused G2P + particle update; three launches on a single
domain: the grid update rides in the P2G launch and the fused G2P bins its output for the next sort), inputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5] [--scaling weak|strong]

--config: c2 (default, the headline), c3 = Drucker-Prager sand column, 4M, standing between the floor and four walls,
c4 = corotated cube on the floor hit by a kinematic rotating cuboid, 8M, c5 = pressure-only neo-Hookean fluid block, 16M
(BASELINE.json configs[2], configs[3], configs[4]).
N > 1: one process per GPU, x-slab domain decomposition; per substep each rank swaps ONE message with each of its two
neighbours — the partial node sums of the interface layers and the particles changing owner — over RCCL point-to-point
issued from inside the library (wgs_sharded_step, include/wgsparkl_hip.h), no collective on the data path. Before
anything is timed the decomposition validates itself against a single-domain run (validation_sharded; exit 3 on a
mismatch). --scaling weak (default): N
copies of the config side by side along x (fixed work per GPU); strong: the named size cut into N slabs (north_star's
16M target: --config c5 --scaling strong). Without flags, the line also carries `extra` legs: the same cube after it
landed on the floor, c3, c4 on one GPU, the reference's own sand3 (202 k particles) and sand2 (490 k, 2D) scenes, c2_stirred (the cube
crossing the grid and spinning: > 10 % of its particles change cell per substep), c2_frames (20 substeps per wgs_step call
with the pose read-back between calls) and c5 (strong over the N GPUs), each with its pass times, its own G2P roofline figure
and `mover_fraction` (particles that changed cell per substep, counted on the device).
`python bench.py --gpus N` outside a launcher starts the N ranks itself (torch.distributed.run as a child process).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
KERNEL_ELASTIC = "k_g2p_pair (fused G2P + particle update + binning for the next sort; collider simulations run both bodies in this launch)"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=("c2", "c3", "c4", "c5"), default="c2")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--n-side", type=int, default=None, help="particles per cube edge (c2: 100 -> 1M, the named config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra legs (landed cube, c3, sand3, sand2, stirred, frames, c5)")
    ap.add_argument("--no-floor", action="store_true", help="c2 without the floor cuboid of SURVEY 8d (no CPIC passes)")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed rocprofv3 passes instead of two child runs under rocprofv3 --pmc")
    ap.add_argument("--no-prewarm", action="store_true", help="skip the ~0.1 s of throwaway substeps that bring the GPU clocks up before the measured leg")
    ap.add_argument("--allow-debug-switches", action="store_true", help="run although WGS_DEBUG is set (A/B of launch shapes)")
    return ap.parse_args()


def spawn_ranks(args):
    """`--gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run (never exec) and relay its exit
    code; rank 0 of the child prints the JSON line on the shared stdout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def cpu_baseline(scene, substeps):
    """The oracle (CPU restatement of the reference algorithm, C + OpenMP over the per-node / per-particle loops,
    the sort stays serial) timed on the GPU box's host cores, on a bounded sample of the same workload. Returns the
    oracle's final state too: bench.py checks the HIP path against it on the very bench data."""
    import numpy as np
    from oracle.orc import Oracle
    ps = scene["particles"]
    orc = Oracle(3, np.float32, omp=True)
    st = orc.new_state(ps, scene["params"], scene["colliders"], scene["cell_width"], scene["grid_capacity"], scene["model"])
    st.step(1)      # first touch of the arrays, thread pool start-up
    t0 = time.perf_counter()
    st.step(substeps)
    dt = time.perf_counter() - t0
    return ps.n * substeps / dt, dt, orc.num_threads, st


def live_pmc_traffic():
    """HBM bytes per launch of the fused G2P kernel, measured NOW: two child runs of this script (the headline workload, 20
    substeps) under `rocprofv3 --kernel-trace --pmc`, one counter each as MI355X_MICROARCH.md prescribes (FETCH_SIZE, WRITE_SIZE;
    units of 1 KiB; FETCH_SIZE doubled: gfx950 counts the 128-B requests of 16-B-per-lane streams as 64 B) — the very recipe of
    tools/gpu_round_profile.sh + tools/summarize_profiles.py behind profiles/rNN_pmc_g2p.json. None when rocprofv3 is missing,
    fails or times out (the caller then quotes the committed passes)."""
    import csv, glob, shutil, signal, tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None
    vals = {}
    # the children are plain single-GPU runs: nothing of a launcher's or this harness's environment goes along
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK")
           and not k.startswith(("MASTER_", "WGS_BENCH_", "TORCHELASTIC_"))}
    env["TMPDIR"] = "/tmp"
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="wgs_pmc_", dir="/tmp")
        try:
            cmd = [rp, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", tmp, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extra", "--no-live-pmc"]
            # a session of its own: on a timeout the whole group goes (rocprofv3 AND the bench process under it), never by pattern
            child = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                child.wait(timeout=150)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(child.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                child.wait()
                return None
            rows = []
            for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") == ctr and ("k_g2p_pair" in r.get("Kernel_Name", "") or "k_g2p_update" in r.get("Kernel_Name", "")):
                        rows.append(float(r["Counter_Value"]))
            if len(rows) < 8:
                return None
            rows = rows[len(rows) // 2:]          # the launches of the timed half: the cube is in steady free fall
            vals[ctr] = sum(rows) / len(rows)
        except Exception:  # noqa: BLE001 — any failure of the profiler means "not measured", never a failed bench
            return None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return {"bytes": vals["FETCH_SIZE"] * 1024.0 * 2.0 + vals["WRITE_SIZE"] * 1024.0,
            "raw_counters_per_launch": {"FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
                                        "note": "bytes = FETCH_SIZE x 2 (gfx950 counts the 128-B requests of 16-B-per-lane streams as 64 B: MI355X_MICROARCH.md) x 1 KiB + WRITE_SIZE x 1 KiB"}}


def g2p_roofline(timings, k_ts, mark_ms, n, n_nodes, bytes_per_particle, kernel):
    # One launch between the two marks of the "g2p" pass: the event interval IS the duration rocprofv3 reports for the
    # kernel (profiles/rNN_kernel_stats.csv; both contain the launch's dispatch gap, and the rocprofv3 durations of the
    # launches of a substep add up to the un-instrumented wall time per substep). Two marks recorded back to back
    # are 3.5 us apart (event_mark_ms, informational): that spacing is what an EMPTY pass costs the instrumented run, it
    # is not a cost inside an interval that holds a kernel — earlier rounds subtracted it and overstated the rate by 10 %.
    interval = timings["g2p"] / k_ts
    ms = max(interval, 1e-9)
    algo = bytes_per_particle * n + 16.0 * n_nodes     # 160 B (elastic) / 216 B (Drucker-Prager) per particle + 16 B per node
    achieved = algo / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "traffic_source": None, "algorithmic_bytes_per_launch": algo, "avg_launch_ms": ms,
            "event_interval_ms": interval, "event_mark_ms": mark_ms}


class Leg:
    """One timed workload on this rank: single-domain data (world == 1) or one slab of the decomposition."""

    def __init__(self, env, scene, world, rank, frame=0):
        from wgsparkl_amd import MpmData
        self.env, self.scene, self.world, self.rank = env, scene, world, rank
        self.frame = frame      # > 0: the caller's frame loop — wgs_step calls of `frame` substeps with the pose read-back between them
        ps = scene["particles"]
        if ps.dim == 2 and env.get("pipe2") is None:
            from wgsparkl_amd import MpmPipeline
            env["pipe2"] = MpmPipeline(env["dev_index"], 2)
        pipe = self.pipe = env["pipe2"] if ps.dim == 2 else env["pipe"]
        self.n = ps.n
        self.sharded = world > 1 or env["force_sharded"]
        self.n_total = scene.get("global_particles", ps.n) if self.sharded else ps.n
        self.bytes_per_particle = scene.get("bytes_per_particle", 160.0)
        if not self.sharded:
            self.data = MpmData.new(pipe, scene["params"], ps, scene["colliders"], scene["cell_width"], scene["grid_capacity"], scene["model"])
            self.handle = self.data._h
            self.parallelism = "1 GPU"
            return
        from wgsparkl_amd.sharded import NativeShard, uniform_material_of
        lo, hi = scene["partition"].block_range(rank)
        ny = int(round((float(ps.pos[:, 1].max()) - float(ps.pos[:, 1].min())) * 2.0 / scene["cell_width"])) + 1 if ps.n else 1
        nz = int(round((float(ps.pos[:, 2].max()) - float(ps.pos[:, 2].min())) * 2.0 / scene["cell_width"])) + 1 if ps.n else 1
        # Messages travel at their full capacity (no size handshake), so the capacities are sized from the workload: a
        # face touches at most (ny / 8 + 3) x (nz / 8 + 3) blocks, each sends one record (one x-layer pair) and the
        # blocks migrating particles touch a few more (+ 50 % + margin); a handful of particles cross a cut per substep
        # while the body falls along y. An overflow is reported by wgs_sync and by the particle count below.
        face = (ny // 8 + 3) * (nz // 8 + 3)
        # every rank generates particles of the same single material (scenes.py): the constants become kernel
        # arguments on sharded data too, like wgs_data_create decides by itself for a single domain
        self.data = NativeShard(pipe, scene["params"], ps, scene["global_ids"], scene["colliders"], scene["cell_width"], scene["grid_capacity"],
                                lo, hi, rank > 0, rank < world - 1, particle_capacity=int(ps.n * 1.25) + 4096, model=scene["model"],
                                halo_capacity_records=face + face // 2 + 64, migrant_capacity=max(512, (ny * nz) // 32), comm=env["comm"],
                                uniform_material=uniform_material_of(ps))
        self.parallelism = f"{world} x-slabs, one message per neighbour and substep (node sums + migrating particles) over RCCL send/recv inside wgs_sharded_step"
        self.handle = self.data._h

    def run(self, k):
        if self.frame:
            # one call per frame and the blocking pose read-back behind it (src_testbed/step.rs:122-132,175-176): the launch
            # shapes that follow "the host's last look" see a look every `frame` substeps
            for _ in range(k // self.frame):
                self.pipe.step(self.data, self.frame)
                self.data.read_body_poses()
            if k % self.frame:
                self.pipe.step(self.data, k % self.frame)
        elif not self.sharded:
            self.pipe.step(self.data, k)
        else:
            self.data.step(k)

    def timed(self, steps, warmup):
        env = self.env
        torch, dist = env["torch"], env["dist"]
        self.run(warmup)
        self.data.sync()
        st0 = None if self.sharded else self.data.stats()
        movers0 = None if st0 is None else st0["cell_changers"]
        env["barrier"]()
        t0 = time.perf_counter()
        self.run(steps)                      # exactly K substeps
        env["barrier"]()                     # torch.cuda.synchronize() (+ the process group's barrier): every stream of the device, the data's included
        elapsed = time.perf_counter() - t0
        self.data.sync()                     # (device-side errors of the timed substeps surface here: wgs_sync)
        # particles that changed their associated cell per substep of the timed region (device counter, wgs_stats.cell_changers)
        st1 = None if st0 is None else self.data.stats()
        self.mover_fraction = None if movers0 is None else (st1["cell_changers"] - movers0) / max(1, self.n * steps)
        # fixed-cost events INSIDE the timed region: substeps that rebuilt the table of block ids, growths of the grid (wgs_stats)
        self.events = None if st0 is None else {"table_rebuilds": st1["table_rebuilds"] - st0["table_rebuilds"], "grid_growths": st1["grid_growths"] - st0["grid_growths"]}
        if dist is not None:
            t = torch.tensor([elapsed], device=env["device"], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            cnt = torch.tensor([self.data.num_particles()], device=env["device"], dtype=torch.int64)
            dist.all_reduce(cnt)
            assert int(cnt.item()) == self.n_total, f"particles lost in the exchange: {int(cnt.item())} != {self.n_total}"
        return elapsed

    def result(self, steps, elapsed, kernel_name):
        """Throughput + per-pass device times (HIP events on the data's own stream) of up to 64 more substeps of the local
        data (a slab steps without its neighbours here: kernel timing only, after the timed region)."""
        from wgsparkl_amd import _ffi
        pipe = self.pipe
        k_ts = min(steps, 64)
        _ffi.check(pipe.lib, pipe.lib.wgs_step(pipe._h, self.handle, k_ts, 1))
        self.data.sync()
        ms = (C.c_float * _ffi.WGS_NUM_PASSES)()
        _ffi.check(pipe.lib, pipe.lib.wgs_read_timings(self.handle, ms))
        timings = dict(zip(_ffi.PASS_NAMES, [float(x) for x in ms]))
        st = pipe.T.Stats()
        _ffi.check(pipe.lib, pipe.lib.wgs_get_stats(self.handle, C.byref(st)))
        ovh = C.c_float(0.0)
        pipe.lib.wgs_read_timing_overhead(self.handle, C.byref(ovh))   # cost of one timing mark, measured in the same substeps
        nblocks = int(st.num_active_blocks)
        return {"value": self.n_total * steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "steps": steps,
                "global_particles": self.n_total, "active_blocks_rank0": nblocks, "near_collider_blocks_rank0": int(st.num_near_collider_blocks),
                "mover_fraction": self.mover_fraction, "events_in_timed_region": getattr(self, "events", None),
                "parallelism": self.parallelism,
                "roofline": g2p_roofline(timings, k_ts, float(ovh.value), self.n, nblocks * 64, self.bytes_per_particle, kernel_name),
                "pass_ms_per_step": {k: v / k_ts for k, v in timings.items()}}

    def close(self):
        self.data.close()


def measure(env, scene, world, rank, steps, warmup, kernel_name, settle=0, frame=0):
    leg = Leg(env, scene, world, rank, frame=frame)
    if settle:
        leg.run(settle)
    res = leg.result(steps, leg.timed(steps, warmup), kernel_name)
    leg.close()
    return res


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))                      # before anything touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    # stdout carries ONE line, the JSON of rank 0: everything else any library prints there (the RCCL version banner,
    # gloo's connection messages, ...) is sent to stderr by pointing file descriptor 1 at it until that line is due
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        try:
            C.CDLL(None).fflush(None)      # C stdio buffers of the libraries
        except Exception:
            pass
        os.write(json_fd, (json.dumps(obj) + "\n").encode())
    force_sharded = os.environ.get("WGS_BENCH_FORCE_SHARDED") == "1"
    if world != args.gpus and not force_sharded:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    dbg_env = os.environ.get("WGS_DEBUG")
    if os.environ.get("WGS_REHASH_PERIOD") and not args.allow_debug_switches:
        print("bench.py: WGS_REHASH_PERIOD is set (developer override); unset it or pass --allow-debug-switches", file=sys.stderr)
        sys.exit(2)
    if dbg_env not in (None, "", "0") and not args.allow_debug_switches:
        print(f"bench.py: WGS_DEBUG={dbg_env} is set (developer launch-shape switches); unset it or pass --allow-debug-switches", file=sys.stderr)
        sys.exit(2)

    import torch
    dist = None
    # WGS_BENCH_FORCE_SHARDED=1: the N > 1 code path (process group, sharded data, wgs_sharded_step) with however many
    # ranks were launched, even one — a functional check for 1-GPU boxes
    sharded_path = world > 1 or force_sharded
    if sharded_path:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev_index = local_rank if sharded_path else 0
    device = f"cuda:{dev_index}"

    import numpy as np
    from wgsparkl_amd import MpmPipeline, scenes

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    pipe = MpmPipeline(dev_index, 3)
    build_info = pipe.lib.wgs_build_info().decode()
    if "WGS_ABLATE" in build_info and os.environ.get("WGS_BENCH_ALLOW_ABLATE") != "1":
        print("bench.py: the library was built with -DWGS_ABLATE (ablation switches compiled in): not a product build", file=sys.stderr)
        sys.exit(2)
    env = dict(torch=torch, dist=dist, pipe=pipe, pipe2=None, dev_index=dev_index, barrier=barrier, device=device, force_sharded=force_sharded, comm=None)
    transport_note = None
    if sharded_path:
        # The substep protocol runs inside the library over RCCL (wgs_sharded_step); torch.distributed only makes the
        # process group of this harness and hands the 128-byte unique id around. No communicator, no run: on every rank.
        comm, why = None, ""
        try:
            from wgsparkl_amd.sharded import NativeComm
            comm = NativeComm(pipe, dist, rank, world)
        except Exception as e:  # noqa: BLE001 — reported below, after all ranks agreed
            why = str(e)
        ok = torch.tensor([1 if comm is not None else 0], device=device, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            print(f"[bench rank {rank}] wgs_comm_create failed on some rank ({why or 'not this one'}): no multi-GPU run", file=sys.stderr)
            dist.destroy_process_group()
            sys.exit(4)
        env["comm"] = comm

    # ---- N > 1: the decomposition validates itself before anything is timed (a small bar on all ranks through
    # wgs_sharded_step against the single-domain run of the same bar on rank 0: ids exact, pos / vel rel RMS < 1e-5)
    validation_sharded = None
    if sharded_path and world > 1:
        from wgsparkl_amd.selfcheck import bar_check
        validation_sharded = bar_check(pipe, dist, env["comm"], world, rank)
        if not validation_sharded["ok"]:
            if rank == 0:
                emit({"metric": "particle-steps/sec", "value": None, "unit": "particle-steps/s", "n_gpus": world, "validation_sharded": validation_sharded,
                      "error": "the sharded run disagrees with the single-domain run: nothing was timed"})
                print("bench.py: wgs_sharded_step on %d ranks disagrees with the single-domain run: %s" % (world, validation_sharded), file=sys.stderr)
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(3)

    # ---- the headline workload
    default_workload = args.config == "c2" and args.n_side in (None, 100) and args.scaling == "weak" and not args.no_floor
    if not sharded_path and args.config == "c2":
        scene = scenes.neo_hookean_cube(n_side=args.n_side or 100, with_floor=not args.no_floor)
        scene["name"] = (f"wgsparkl3d neo-Hookean elastic cube, {scene['particles'].n} particles, 128^3-cell domain, h=1, dt=1/1200, "
                         "8 particles/cell, " + ("no collider" if args.no_floor else "floor cuboid (CPIC passes on)") + ", free fall")
        scene["bytes_per_particle"] = 160.0
    elif default_workload:
        scene = scenes.neo_hookean_bar(n_side=100, world=world, rank=rank)          # N C2 cubes side by side = one elastic bar
        scene["name"] = f"wgsparkl3d neo-Hookean elastic cube x {world} side by side (one bar), 1000000 particles/GPU, floor cuboid"
        scene["bytes_per_particle"] = 160.0
    else:
        scene = scenes.config_scene(args.config, world, rank if sharded_path else None, args.scaling, n_side=args.n_side)
        if args.no_floor:
            scene["colliders"] = []
    def partition_ok(sc):
        """Every rank checks the cut it was given against what wgs_shard_attach accepts (a slab between two neighbours is at
        least MIN_INTERIOR_WIDTH blocks wide) and the ranks agree on the verdict BEFORE any of them builds a slab: one rank
        raising alone would leave the others blocked in their first ncclRecv."""
        if not sharded_path or "partition" not in sc:
            return True
        from wgsparkl_amd.sharded import MIN_INTERIOR_WIDTH
        ok = world < 3 or sc["partition"].min_interior_width() >= MIN_INTERIOR_WIDTH
        if dist is not None:
            t = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = int(t.item()) == 1
        return ok

    if not partition_ok(scene):
        if rank == 0:
            from wgsparkl_amd.sharded import MIN_INTERIOR_WIDTH
            emit({"metric": "particle-steps/sec", "value": None, "unit": "particle-steps/s", "n_gpus": world,
                  "error": f"--config {args.config} --scaling {args.scaling} cut into {world} x-slabs leaves a slab between two neighbours narrower than the "
                           f"{MIN_INTERIOR_WIDTH} blocks the one-message halo protocol needs (kernels_shard.h): nothing was timed"})
            print(f"bench.py: the decomposition of --config {args.config} into {world} slabs is too narrow; nothing was timed", file=sys.stderr)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(5)
    # Clock pre-warm (set-up, not part of the W + K substeps): a throwaway copy of the workload is stepped for ~0.1 s and released before
    # the measured data is created — on a box that has just been handed out the first milliseconds of GPU work run at ramping clocks
    # (one of three default runs read 98.6 us per substep where the other two read 93), and W = 10 warm-up substeps last one millisecond.
    # The measured leg starts from the same fresh state as without it.
    prewarm_substeps = 0
    if not args.no_prewarm:
        prewarm_substeps = int(min(1000, max(100, 500_000_000 // max(1, scene["particles"].n))))   # (C2: 500 substeps = 46 ms; C5: 100 = 0.1 s)
        pw = Leg(env, scene, world, rank)
        pw.run(prewarm_substeps)
        pw.data.sync()
        env["barrier"]()
        pw.close()
        del pw
    main_res = measure(env, scene, world, rank, args.steps, args.warmup,
                       ("k_g2p_pair<plastic>" if args.config == "c3" else KERNEL_ELASTIC) if scene["colliders"] else "k_g2p_update (fused G2P + particle update)")

    out = None
    if rank == 0:
        rl = main_res["roofline"]
        prof = sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r*_pmc_g2p.json")))[-1:]
        # (the live measurement — two child runs under rocprofv3 — comes after every timed leg, below: the children use this GPU;
        # until then the committed passes of this very command stand in)
        if prof and default_workload and world == 1:
            try:   # PMC byte counters of this very command, collected by rocprofv3 in its own passes and committed
                rl["traffic"] = json.load(open(prof[0])).get("hbm_bytes_per_launch")
                rl["traffic_source"] = os.path.relpath(prof[0], ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; not measured in this run)"
            except Exception:
                pass
        out = {
            "metric": "particle-steps/sec", "value": main_res["value"], "unit": "particle-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"],
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": scene["name"], "config": args.config, "particles_per_gpu": main_res["global_particles"] // world,
                       "global_particles": main_res["global_particles"], "active_blocks_rank0": main_res["active_blocks_rank0"],
                       "parallelism": main_res["parallelism"]},
            "roofline": rl,
            "pass_ms_per_step": main_res["pass_ms_per_step"],
            "events_in_timed_region": main_res["events_in_timed_region"], "mover_fraction": main_res["mover_fraction"],
            "prewarm_substeps": prewarm_substeps,
            "build": {"info": build_info, "WGS_DEBUG": dbg_env, "transport_note": transport_note},
            "validation_sharded": validation_sharded,
            "notes": "the table of block ids is built by the first substep (a k_bin launch in front of the sort, ~+0.25 ms once at this size); blocks "
                     "nobody activated for 8 substeps are evicted from it and their ids reused, so it is rebuilt again only when the grid grows "
                     "(events_in_timed_region counts such substeps per leg). "
                     "pass_ms_per_step: on single-domain data the grid update runs as workgroups of the P2G launch and the fused G2P bins its output for the "
                     "next substep's sort (DESIGN.md 4): 'p2g' holds P2G + grid update, 'g2p' holds the binning, 'grid sort' is the one regroup launch, "
                     "'grid_update' is then an empty interval between two event marks (~0.004 ms, like every pass without a launch)",        }

    # ---- CPU baseline + validation of the HIP path on the bench data itself (rank 0, N = 1)
    if rank == 0 and not args.no_cpu_baseline and not sharded_path:
        from wgsparkl_amd import MpmData
        sub = 20 if scene["particles"].n <= 2_000_000 else 4
        v, secs, threads, st = cpu_baseline(scene, sub)
        out["cpu_baseline"] = {"value": v, "unit": "particle-steps/s", "cores": threads, "kind": "port",
                               "sample": f"{sub} substeps of the same {scene['particles'].n}-particle workload, C + OpenMP oracle "
                                         f"(CPU restatement of the reference WGSL algorithm, {threads} threads; the "
                                         f"hash-grid sort is serial), {secs:.1f} s; "
                                         "reference WGSL via wgpu+lavapipe: unavailable (no cargo/rustc/Vulkan ICD)"}
        chk = MpmData.new(pipe, scene["params"], scene["particles"], scene["colliders"], scene["cell_width"], scene["grid_capacity"], scene["model"])
        pipe.step(chk, sub + 1)               # the oracle did 1 + sub substeps
        chk.sync()
        got = chk.read_particles()
        rel = lambda a, b: float(np.sqrt(np.mean((a.astype(np.float64) - b) ** 2)) / max(np.sqrt(np.mean(np.asarray(b, np.float64) ** 2)), 1e-30))
        gc, oc = chk.read_grid()[0], st.grid_records()[0]
        val = {"substeps": sub + 1, "against": "fp32 oracle (the reference's arithmetic) on the bench data",
               "pos_rel_rms": rel(got.pos, st.arr["pos"]), "vel_rel_rms": rel(got.vel, st.arr["vel"]),
               "def_grad_rel_rms": rel(got.def_grad, st.arr["def_grad"]),
               "active_cells_identical": bool(gc.shape == oc.shape and np.array_equal(gc, oc))}
        val["ok"] = bool(val["active_cells_identical"] and val["pos_rel_rms"] < 1e-5 and val["vel_rel_rms"] < 1e-4 and val["def_grad_rel_rms"] < 1e-5)
        out["validation"] = val
        chk.close()
        del st, got, chk
        if not val["ok"]:
            emit(out)
            print("bench.py: the HIP path disagrees with the oracle on the bench data", file=sys.stderr)
            sys.exit(3)

    # ---- extra legs (default command only): other states of the solver, each with its own G2P roofline figure
    if not args.no_extra and default_workload:
        extra = {}
        k, w = min(args.steps, 50), 5
        slim = lambda r, name: {"workload": name, "value": r["value"], "unit": "particle-steps/s", "ms_per_step": r["ms_per_step"],
                                "global_particles": r["global_particles"], "active_blocks_rank0": r["active_blocks_rank0"],
                                "near_collider_blocks_rank0": r["near_collider_blocks_rank0"], "steps": r["steps"], "mover_fraction": r["mover_fraction"],
                                "events_in_timed_region": r["events_in_timed_region"],
                                "roofline_g2p": {x: r["roofline"][x] for x in ("achieved", "frac", "avg_launch_ms", "algorithmic_bytes_per_launch")},
                                "pass_ms_per_step": r["pass_ms_per_step"], "parallelism": r["parallelism"]}
        if not sharded_path:
            sc = scenes.neo_hookean_cube(n_side=100, with_floor=True)
            sc["particles"].pos[:, 1] -= 5.7          # lowered onto the floor, impact at -3 cells/s: contact state after 200 substeps
            sc["particles"].vel[:, 1] = -3.0
            sc["bytes_per_particle"] = 160.0
            km = max(k, 200)   # scenes in motion: a timed region of 20 substeps swings by 15 % with one table rebuild inside it
            extra["c2_landed"] = slim(measure(env, sc, 1, 0, km, w, KERNEL_ELASTIC, settle=200 - w),
                                      "the C2 cube after it landed: lowered onto the floor with a -3 cells/s impact, 200 substeps before the timed region")
            sc = scenes.config_scene("c3")
            extra["c3"] = slim(measure(env, sc, 1, 0, k, w, "k_g2p_pair<plastic>"), sc["name"])
            # BASELINE.json configs[3] on ONE GPU (the config names four; it fits one: 8 M particles = 3.3 GB): corotated elasticity — an SVD per
            # particle, the reference's default model (src/models/linear_elasticity.wgsl:28-41) — with the floor and one kinematic
            # rotating cuboid pushed into the cube (a body whose velocity the host sets and whose pose the device integrates every
            # substep: crates/wgsparkl3d/examples/sand3.rs:95-103); 100 substeps before the timed region, so that the cuboid is in contact
            sc = scenes.config_scene("c4")
            extra["c4"] = slim(measure(env, sc, 1, 0, k, w, KERNEL_ELASTIC, settle=100), sc["name"] + ", one GPU, 100 substeps before the timed region")
            # the size the reference itself ships (its scenes hold 75 k - 490 k particles): latency-bound here, four dependent launches
            sc = scenes.reference_sand3()
            extra["sand3_202k"] = slim(measure(env, sc, 1, 0, k, w, "k_g2p_pair<plastic>", settle=100), sc["name"] + ", 100 substeps before the timed region")
            # the reference's largest shipping scene is 2D (crates/wgsparkl2d/examples/sand2.rs:33-50)
            sc = scenes.reference_sand2()
            extra["sand2_490k"] = slim(measure(env, sc, 1, 0, k, w, "k_g2p_pair<2D, plastic>", settle=100), sc["name"] + ", 100 substeps before the timed region")
            # particles that CHANGE CELLS: the headline cube is in free fall and moves 2e-4 cells in the timed region — the sort sees
            # no mover. Here it flies through the grid at (48, 48, 48) cells/s while spinning at 1.5 rad/s about its vertical axis:
            # more than one particle in ten changes its cell in every substep (mover_fraction: counted on the device)
            sc = scenes.neo_hookean_cube(n_side=100, with_floor=True)
            c = sc["particles"].pos.mean(0)
            rel = sc["particles"].pos - c
            sc["particles"].vel[:, 0] = 48.0 + 1.5 * rel[:, 2]
            sc["particles"].vel[:, 1] = 48.0
            sc["particles"].vel[:, 2] = 48.0 - 1.5 * rel[:, 0]
            sc["bytes_per_particle"] = 160.0
            extra["c2_stirred"] = slim(measure(env, sc, 1, 0, km, w, KERNEL_ELASTIC),
                                       "the C2 cube flying through the grid at (48, 48, 48) cells/s and spinning at 1.5 rad/s: particles change cells in every substep")
            # the caller's frame loop: 20 substeps per wgs_step call, the blocking pose read-back between the calls
            # (src_testbed/step.rs:122-132,175-176; sand3.rs runs 20 substeps per frame)
            sc = scenes.neo_hookean_cube(n_side=100, with_floor=True)
            sc["bytes_per_particle"] = 160.0
            extra["c2_frames"] = slim(measure(env, sc, 1, 0, 60 if k >= 50 else 20, 20, KERNEL_ELASTIC, frame=20),
                                      "the headline workload stepped as the testbed does: one wgs_step call of 20 substeps per frame, wgs_read_body_poses after each")
            del sc
        sc = scenes.config_scene("c5", world, rank if sharded_path else None, "strong")
        if partition_ok(sc):
            r = measure(env, sc, world, rank, k, w, KERNEL_ELASTIC)
            if rank == 0:
                extra["c5_strong"] = slim(r, sc["name"] + (f", cut into {world} x-slabs (strong scaling: fixed 16 M global)" if world > 1 else ", one GPU"))
        del sc
        if sharded_path and world > 1:
            # the headline's own 1 M cube cut N ways (BASELINE.json's metric read as strong scaling). A slab between two
            # neighbours must be at least MIN_INTERIOR_WIDTH = 3 blocks wide (kernels_shard.h, wgs_shard_attach): 12.5 block
            # columns cut 5 or more ways are not, so those N have no such leg
            sc = scenes.config_scene("c2", world, rank, "strong")
            if partition_ok(sc):
                r = measure(env, sc, world, rank, k, w, KERNEL_ELASTIC)
                if rank == 0:
                    extra["c2_strong"] = slim(r, sc["name"] + f", cut into {world} x-slabs (strong scaling: fixed 1 M global)")
            elif rank == 0:
                extra["c2_strong"] = {"skipped": f"the 50-cell cube cut into {world} x-slabs leaves slabs narrower than the three blocks the halo protocol needs"}
            del sc
        if sharded_path and world == 4:
            # BASELINE.json configs[3]: 8 M corotated + kinematic rotating cuboid on 4 GPUs (x-slabs)
            sc = scenes.config_scene("c4", world, rank, "strong")
            if partition_ok(sc):
                r = measure(env, sc, world, rank, k, w, KERNEL_ELASTIC)
                if rank == 0:
                    extra["c4_strong"] = slim(r, sc["name"] + ", cut into 4 x-slabs")
            del sc
        if rank == 0:
            out["extra"] = extra

    if rank == 0 and default_workload and world == 1 and not sharded_path and not args.no_live_pmc and not args.no_extra:
        live = live_pmc_traffic()   # two short child runs of the headline workload under rocprofv3 --pmc (one counter each)
        if live is not None:
            out["roofline"]["traffic"] = live["bytes"]
            out["roofline"]["traffic_raw"] = live["raw_counters_per_launch"]
            out["roofline"]["traffic_source"] = ("measured by this run, after its timed legs: two child runs of the headline workload under rocprofv3 --kernel-trace "
                                                 "--pmc (FETCH_SIZE, WRITE_SIZE; one counter per run), fused G2P launches only")
    if rank == 0:
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
