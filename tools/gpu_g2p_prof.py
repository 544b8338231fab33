"""Developer utility (needs a -DWGS_ABLATE build): stage clocks of the fused G2P main body, per wave, last substep of a batch."""
import ctypes as C, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import pipeline
from wgsparkl_amd import MpmData, scenes
if len(sys.argv) > 1 and sys.argv[1] in ("c2", "c3", "c5"):
    sc = scenes.config_scene(sys.argv[1], n_side=int(sys.argv[2]) if len(sys.argv) > 2 else None)
elif len(sys.argv) > 1 and sys.argv[1] == "landed":       # bench.py's c2_landed leg
    sc = scenes.neo_hookean_cube(n_side=100, with_floor=True)
    sc["particles"].pos[:, 1] -= 5.7
    sc["particles"].vel[:, 1] = -3.0
elif len(sys.argv) > 1 and sys.argv[1] == "stirred":      # bench.py's c2_stirred leg: > 10 % of the particles change cell per substep
    sc = scenes.neo_hookean_cube(n_side=100, with_floor=True)
    rel = sc["particles"].pos - sc["particles"].pos.mean(0)
    sc["particles"].vel[:, 0] = 48.0 + 1.5 * rel[:, 2]
    sc["particles"].vel[:, 1] = 48.0
    sc["particles"].vel[:, 2] = 48.0 - 1.5 * rel[:, 0]
else:
    n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    sc = scenes.neo_hookean_cube(n_side=n_side, with_floor=True)
pipe = pipeline(3)
data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
ROWS = 16384
buf = (C.c_ulonglong * (ROWS * 8))()
pipe.step(data, 200 if len(sys.argv) > 1 and sys.argv[1] == 'landed' else 20); data.sync()
names = ["sort entries + state requested", "state + tile in", "stencil done", "update + stress done", "stores issued (end)", "plastic: parameters in", "plastic: SVD done"]
for rep in range(2):
    pipe.lib.wgs_debug_g2p_prof(buf)
    pipe.step(data, 10); data.sync()
    pipe.lib.wgs_debug_g2p_prof(buf)
    a = np.array(list(buf), np.float64).reshape(ROWS, 8)
    lw = a[ROWS * 3 // 4:]
    vis = a[ROWS // 2:ROWS * 3 // 4]
    xcd = np.arange(len(lw)) % 8
    keep = (lw[:, 0] > 0) & (lw[:, 5] > 0)
    lw, xcd = lw[keep], xcd[keep]
    a = a[:ROWS // 2]
    if len(lw):
        life = (lw[:, 5] - lw[:, 0]) / 100.0
        t00 = min(a[a[:, 0] > 0][:, 0].min(), lw[:, 0].min())
        print(f"list waves: {len(lw)}; life mean {life.mean():.1f} p90 {np.percentile(life, 90):.1f} max {life.max():.1f} us; chunks visited mean {lw[:,1].mean():.1f} (total {lw[:,1].sum():.0f}); "
              f"us per visit {life.sum() / max(lw[:,1].sum(), 1):.2f}; first start {(lw[:,0].min()-t00)/100:.1f} last end {(lw[:,5].max()-t00)/100:.1f} us")
    w = a[(a[:, 0] > 0) & (a[:, 5] > 0)]
    w = w[(w[:, 1:6] > 0).all(axis=1)]
    t0 = w[:, 0].min()
    print(f"rep {rep}: {len(w)} waves; last start {(w[:,0].max()-t0)/100:.1f} us, last end {(w[:,5].max()-t0)/100:.1f} us")
    for k, n in enumerate(names):
        ww = w[w[:, 1 + k] > 0]
        if len(ww) == 0: continue
        d = (ww[:, 1 + k] - ww[:, 0]) / 100.0
        print(f"   {n:32s} mean {d.mean():7.2f}  p10 {np.percentile(d,10):7.2f}  p90 {np.percentile(d,90):7.2f} us since the wave's start")
    if len(lw):
        for k in range(8):
            m = lw[xcd == k]
            if len(m): print(f"   XCD {k}: list waves {len(m)}, wave-us {((m[:,5]-m[:,0]).sum())/100:.0f}, visits {m[:,1].sum():.0f}, last end {(m[:,5].max()-t00)/100:.1f} us")
    vv = vis[(vis[:, 0] > 0) & (vis[:, 5] > 0) & (vis[:, 1:5] > 0).all(axis=1)]
    if len(vv):
        print(f"   visits of the list walk: {len(vv)} rows")
        for k, n in enumerate(names):
            ww = vv[vv[:, 1 + k] > 0]
            if len(ww):
                dd = (ww[:, 1 + k] - ww[:, 0]) / 100.0
                print(f"      {n:32s} mean {dd.mean():7.2f}  p10 {np.percentile(dd, 10):6.2f}  p50 {np.percentile(dd, 50):6.2f}  p90 {np.percentile(dd, 90):6.2f} us since the visit's start")
        o = np.argsort(vv[:, 0]); 
    st = np.sort((w[:, 0] - t0) / 100.0)
    print("   wave start times (us) deciles:", [round(float(x), 1) for x in np.percentile(st, [0, 10, 25, 50, 75, 90, 100])])
    live = [(int(((w[:,0]-t0)/100 <= t) .sum() - ((w[:,5]-t0)/100 <= t).sum())) for t in (2, 5, 10, 15, 20, 25, 30)]
    print("   waves alive at t = 2, 5, 10, 15, 20, 25, 30 us:", live)
