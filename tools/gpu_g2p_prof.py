"""Developer utility (needs a -DWGS_ABLATE build): stage clocks of the fused G2P main body, per wave, last substep of a batch."""
import ctypes as C, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import pipeline
from wgsparkl_amd import MpmData, scenes
n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sc = scenes.neo_hookean_cube(n_side=n_side, with_floor=True)
pipe = pipeline(3)
data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
ROWS = 16384
buf = (C.c_ulonglong * (ROWS * 8))()
pipe.step(data, 20); data.sync()
names = ["sort entries + state requested", "state + tile in", "stencil done", "update + stress done", "stores issued (end)"]
for rep in range(2):
    pipe.lib.wgs_debug_g2p_prof(buf)
    pipe.step(data, 10); data.sync()
    pipe.lib.wgs_debug_g2p_prof(buf)
    a = np.array(list(buf), np.float64).reshape(ROWS, 8)
    w = a[(a[:, 0] > 0) & (a[:, 5] > 0)]
    t0 = w[:, 0].min()
    print(f"rep {rep}: {len(w)} waves; last start {(w[:,0].max()-t0)/100:.1f} us, last end {(w[:,5].max()-t0)/100:.1f} us")
    for k, n in enumerate(names):
        d = (w[:, 1 + k] - w[:, 0]) / 100.0
        print(f"   {n:32s} mean {d.mean():7.2f}  p10 {np.percentile(d,10):7.2f}  p90 {np.percentile(d,90):7.2f} us since the wave's start")
    st = np.sort((w[:, 0] - t0) / 100.0)
    print("   wave start times (us) deciles:", [round(float(x), 1) for x in np.percentile(st, [0, 10, 25, 50, 75, 90, 100])])
    live = [(int(((w[:,0]-t0)/100 <= t) .sum() - ((w[:,5]-t0)/100 <= t).sum())) for t in (2, 5, 10, 15, 20, 25, 30)]
    print("   waves alive at t = 2, 5, 10, 15, 20, 25, 30 us:", live)
