"""Developer utility: bench.py's stirred cube (1 M particles flying through the grid and spinning) for 2 000 substeps — wall time per substep and
the table events inside the run, with the eviction of long-inactive blocks (default) and without (WGS_DEBUG=1024: the table is rebuilt
whenever three quarters of the ids are handed out)."""
import os, subprocess, sys, time
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
    from helpers import pipeline
    from wgsparkl_amd import MpmData, scenes
    sc = scenes.neo_hookean_cube(n_side=100, with_floor=True)
    rel = sc["particles"].pos - sc["particles"].pos.mean(0)
    sc["particles"].vel[:, 0] = 48.0 + 1.5 * rel[:, 2]; sc["particles"].vel[:, 1] = 48.0; sc["particles"].vel[:, 2] = 48.0 - 1.5 * rel[:, 0]
    sc["params"].gravity = (0.0, 0.0, 0.0) if hasattr(sc["params"], "gravity") else None
    pipe = pipeline(3)
    data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
    pipe.step(data, 5); data.sync()
    s0 = data.stats(); t0 = time.perf_counter()
    for _ in range(20):
        pipe.step(data, 100)
    data.sync(); dt = time.perf_counter() - t0; s1 = data.stats()
    print(f"WGS_DEBUG={os.environ.get('WGS_DEBUG')}: {dt / 2000 * 1e6:.1f} us/substep over 2000 substeps;",
          {k: s1[k] - s0[k] for k in ("table_rebuilds", "table_refreshes", "grid_growths")}, {k: s1[k] for k in ("num_active_blocks", "grid_capacity", "block_ids", "block_ids_free", "table_marks", "overflow")})
    sys.exit(0)
for dbg in (None, "1024", None, "1024"):
    env = dict(os.environ)
    env.pop("WGS_DEBUG", None)
    if dbg: env["WGS_DEBUG"] = dbg
    subprocess.run([sys.executable, __file__, "--child"], env=env)
