"""Developer utility (needs a -DWGS_ABLATE build): stage clocks of the plain P2G body, per block, last substep of a batch."""
import ctypes as C, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import pipeline
from wgsparkl_amd import MpmData, scenes
if len(sys.argv) > 1 and sys.argv[1] == "sand3":
    sc = scenes.reference_sand3()
elif len(sys.argv) > 1 and sys.argv[1] == "landed":       # bench.py's c2_landed leg
    sc = scenes.neo_hookean_cube(n_side=100, with_floor=True)
    sc["particles"].pos[:, 1] -= 5.7
    sc["particles"].vel[:, 1] = -3.0
elif len(sys.argv) > 1 and sys.argv[1] in ("c2", "c3", "c5"):
    sc = scenes.config_scene(sys.argv[1], n_side=int(sys.argv[2]) if len(sys.argv) > 2 else None)
else:
    n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    sc = scenes.neo_hookean_cube(n_side=n_side, with_floor=True)
pipe = pipeline(3)
data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
ROWS = 8192
buf = (C.c_ulonglong * (ROWS * 8))()
pipe.step(data, 200 if len(sys.argv) > 1 and sys.argv[1] == 'landed' else 20); data.sync()
names = ["meta (count, key, cell ranges)", "round-0 fetch issued", "round-0 in LDS", "accumulated", "slab stored"]
for rep in range(2):
    pipe.lib.wgs_debug_p2g_prof(buf)
    pipe.step(data, 10); data.sync()
    pipe.lib.wgs_debug_p2g_prof(buf)
    a = np.array(list(buf), np.float64).reshape(ROWS, 8)
    cp = a[ROWS // 2:]
    cp = cp[(cp[:, 0] > 0) & (cp[:, 5] > 0)]
    a = a[:ROWS // 2]
    if len(cp):
        print(f"CPIC body: {len(cp)} listed blocks recorded")
        for k, n in ((6, "particle cdf prologue done"), (7, "node affinities in"), (1, "round-0 fetch issued"), (2, "round-0 in LDS"), (4, "accumulated"), (5, "slab stored")):
            dd = (cp[:, k] - cp[:, 0]) / 100.0
            print(f"   {n:32s} mean {dd.mean():7.2f}  p10 {np.percentile(dd,10):7.2f}  p90 {np.percentile(dd,90):7.2f} us since the block's start")
    w = a[(a[:, 0] > 0) & (a[:, 5] > 0)]          # blocks with particles (the others leave after the count)
    t0 = a[a[:, 0] > 0][:, 0].min()
    print(f"rep {rep}: {len(w)} blocks with particles; first block starts at 0, last block starts at {(w[:,0].max()-t0)/100:.1f} us, last end {(w[:,5].max()-t0)/100:.1f} us")
    for k, n in enumerate(names):
        d = (w[:, 1 + k] - w[:, 0]) / 100.0
        print(f"   {n:32s} mean {d.mean():7.2f}  p10 {np.percentile(d,10):7.2f}  p90 {np.percentile(d,90):7.2f} us since the block's start")
    for k, n in ((6, "record + runs loaded"), (7, "runs in LDS, tile zeroed, longest run")):
        d = (w[:, k] - w[:, 0]) / 100.0
        print(f"   {n:32s} mean {d.mean():7.2f}  p10 {np.percentile(d,10):7.2f}  p90 {np.percentile(d,90):7.2f} us since the block's start")
    st = np.sort((w[:, 0] - t0) / 100.0)
    print("   block start times (us) deciles:", [round(float(x), 1) for x in np.percentile(st, [0, 10, 25, 50, 75, 90, 100])])
