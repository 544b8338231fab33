"""Developer utility: the same scenes stepped with two libraries (LIB_A, LIB_B = paths of libwgsparkl3d_hip.so variants; default: tools/tmp_libs/base.so
against the in-tree library), results compared bit for bit. A refactor that keeps every particle's arithmetic must print zeros."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TREE = os.path.join(ROOT, "wgsparkl_amd", "csrc", "libwgsparkl3d_hip.so")
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import pipeline
    from wgsparkl_amd import MpmData, scenes
    name, out, steps = sys.argv[2], sys.argv[3], int(sys.argv[4])
    if name == "sand3": sc = scenes.reference_sand3()
    elif name == "landed":
        sc = scenes.neo_hookean_cube(n_side=64, with_floor=True); sc["particles"].pos[:, 1] -= 5.7; sc["particles"].vel[:, 1] = -3.0
    elif name == "c3s": sc = scenes.config_scene("c3", n_side=64)
    else: sc = scenes.neo_hookean_cube(n_side=48, with_floor=True)
    pipe = pipeline(3)
    data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc.get("model", 0))
    pipe.step(data, steps); data.sync()
    p = data.read_particles()
    np.savez(out, pos=p.pos, vel=p.vel, F=p.def_grad, C=p.affine, dp=p.dp_state, aff=p.cdf_affinity, nrm=p.cdf_normal, dist=p.cdf_dist)
    sys.exit(0)
lib_a = os.environ.get("LIB_A", os.path.join(ROOT, "tools", "tmp_libs", "base.so"))
keep = tempfile.mktemp(suffix=".so"); subprocess.run(["cp", TREE, keep], check=True)
try:
    for name, steps in (("sand3", 150), ("landed", 260), ("c3s", 60), ("cube", 40)):
        outs = []
        for tag, lib in (("a", lib_a), ("b", keep)):
            subprocess.run(["cp", lib, TREE], check=True)
            out = tempfile.mktemp(suffix=".npz")
            subprocess.run([sys.executable, __file__, "--child", name, out, str(steps)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            outs.append(np.load(out))
        a, b = outs
        print(name, steps, {k: int((a[k] != b[k]).sum()) for k in a.files}, "max |dpos|", float(np.abs(a["pos"] - b["pos"]).max()))
finally:
    subprocess.run(["cp", keep, TREE], check=True)
