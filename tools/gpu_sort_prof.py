"""Developer utility (needs a -DWGS_ABLATE build): stage clocks of k_regroup over the last substep of a batch."""
import ctypes as C, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import pipeline
from wgsparkl_amd import MpmData, scenes
if len(sys.argv) > 1 and sys.argv[1] == "sand3":
    sc = scenes.reference_sand3()
elif len(sys.argv) > 1 and sys.argv[1] == "stirred":      # bench.py's c2_stirred leg: > 10 % of the particles change cell per substep
    sc = scenes.neo_hookean_cube(n_side=100, with_floor=True)
    rel = sc["particles"].pos - sc["particles"].pos.mean(0)
    sc["particles"].vel[:, 0] = 48.0 + 1.5 * rel[:, 2]
    sc["particles"].vel[:, 1] = 48.0
    sc["particles"].vel[:, 2] = 48.0 - 1.5 * rel[:, 0]
elif len(sys.argv) > 1 and sys.argv[1] == "landed":       # bench.py's c2_landed leg
    sc = scenes.neo_hookean_cube(n_side=100, with_floor=True)
    sc["particles"].pos[:, 1] -= 5.7
    sc["particles"].vel[:, 1] = -3.0
elif len(sys.argv) > 1 and sys.argv[1] in ("c2", "c3", "c5"):
    sc = scenes.config_scene(sys.argv[1])
else:
    n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    sc = scenes.neo_hookean_cube(n_side=n_side, with_floor=True)
pipe = pipeline(3)
data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
ROWS = 8192
buf = (C.c_ulonglong * (ROWS * 8))()
pipe.step(data, 200 if len(sys.argv) > 1 and sys.argv[1] == 'landed' else 20); data.sync()
names = ["links", "staged", "pass1", "cdf", "pass2", "bstart", "end"]
for rep in range(2):
    pipe.lib.wgs_debug_prof(buf)          # reset
    pipe.step(data, 10); data.sync()      # rows hold the clocks of the LAST substep's launch
    pipe.lib.wgs_debug_prof(buf)
    a = np.array(list(buf), np.float64).reshape(ROWS, 8)
    scan = a[ROWS - 1, 0]
    w = a[:ROWS - 1][a[:ROWS - 1, 0] > 0]
    t0 = w[:, 0].min()
    late = np.argsort(-w[:, 7])[:5]
    print("   slowest waves (us since their start, per stage):", [[round(float(x), 1) for x in (w[i, 1:8] - w[i, 0]) / 100.0] for i in late])
    print(f"   scan wg 0: starts at {(a[ROWS-1,1]-t0)/100:.1f} us, has its total at {(a[ROWS-1,2]-t0)/100:.1f} us")
    print(f"rep {rep}: {len(w)} waves; first start 0, last start {(w[:,0].max()-t0)/100:.1f} us, scan published at {(scan-t0)/100:.1f} us, last end {(w[:,7].max()-t0)/100:.1f} us")
    for k, n in enumerate(names):
        d = (w[:, 1 + k] - w[:, 0]) / 100.0
        print(f"   {n:8s} mean {d.mean():7.2f}  p10 {np.percentile(d,10):7.2f}  p90 {np.percentile(d,90):7.2f} us since the wave's start")
