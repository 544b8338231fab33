"""CPU checker backend for the multi-GPU protocol of wgsparkl_amd/sharded.py: the same phases
(step_begin / pack_halo / add_halo / step_end / pack_migrants / add_migrants) on top of the C oracle,
so that the decomposition can be exercised under torch.distributed(gloo) without a GPU.
TEST INFRASTRUCTURE ONLY."""
import numpy as np
import torch

from oracle.orc import Oracle
from wgsparkl_amd.sharded import associated_block_x
from wgsparkl_amd.solver import ParticleSet

FIELDS = ("pos", "vel", "def_grad", "affine", "cdf_normal", "cdf_rigid_vel", "cdf_dist", "init_volume", "mass",
          "lambda_", "mu", "dp", "dp_state", "phase")


class OracleShard:
    def __init__(self, scene, sub: ParticleSet, gids, block_lo, block_hi, has_lower, has_upper, dtype=np.float64):
        self.scene, self.dtype = scene, np.dtype(dtype)
        self.dim = sub.dim
        self.orc = Oracle(self.dim, dtype)
        self.block_lo, self.block_hi, self.has_lower, self.has_upper = block_lo, block_hi, has_lower, has_upper
        self.bw = 4 if self.dim == 3 else 8
        self.gids = np.asarray(gids, np.int64)
        self._make_state({k: np.asarray(getattr(sub, k), dtype) for k in FIELDS}, sub.cdf_affinity.copy())

    def _make_state(self, arrs, aff):
        n = len(aff)
        ps = ParticleSet(dim=self.dim, cdf_affinity=aff.astype(np.uint32), init_radius=np.zeros(n, np.float32),
                         has_plasticity=np.ones(n, bool), has_phase=np.ones(n, bool), **arrs)
        sc = self.scene
        self.st = self.orc.new_state(ps, sc["params"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc.get("model", 0))

    # ---- protocol
    def bin_residents(self):
        """(an optimisation hook of the GPU backend: nothing to pre-compute for the CPU oracle)"""

    def step_begin(self):
        st = self.st
        st.sort(); st.grid_update_cdf(); st.g2p_cdf(); st.p2g()

    def _layer_nodes(self):
        t = np.arange(64)
        lx = t % self.bw
        return np.nonzero(lx < 2)[0]

    def pack_halo(self, layer_bx):
        st, D = self.st, self.dim
        nb = st.n_blocks
        vid = st.g["block_vid"][:nb]
        sel = np.nonzero(vid[:, 0] == layer_bx)[0]
        nodes = self._layer_nodes()
        mv = st.g["node_mv"].reshape(-1, 64, D + 1)
        rec = np.concatenate([vid[sel].astype(np.float64), mv[sel][:, nodes, :].reshape(len(sel), len(nodes) * (D + 1))], axis=1)
        return torch.from_numpy(np.ascontiguousarray(rec.reshape(-1)))

    def add_halo(self, recs):
        st, D = self.st, self.dim
        nodes = self._layer_nodes()
        w = D + len(nodes) * (D + 1)
        rec = recs.numpy().reshape(-1, w)
        nb = st.n_blocks
        index = {tuple(v): b for b, v in enumerate(st.g["block_vid"][:nb])}
        mv = st.g["node_mv"].reshape(-1, 64, D + 1)
        for r in rec:
            b = index.get(tuple(int(x) for x in r[:D]))
            if b is not None:
                mv[b, nodes, :] += r[D:].reshape(len(nodes), D + 1)

    def step_end(self):
        st = self.st
        st.grid_update(); st.g2p(); st.particle_update()

    def _records(self, idx):
        a = self.st.arr
        width = lambda k: int(np.prod(a[k].shape[1:])) if a[k].ndim > 1 else 1
        cols = [a[k][idx].reshape(len(idx), width(k)) for k in FIELDS]
        cols += [a["cdf_affinity"][idx].reshape(len(idx), 1).astype(np.float64), self.gids[idx].reshape(len(idx), 1).astype(np.float64)]
        return np.concatenate(cols, axis=1)

    def pack_migrants(self):
        bx = associated_block_x(self.st.arr["pos"].astype(np.float32), self.scene["cell_width"], self.dim)
        lo = np.nonzero(bx < self.block_lo)[0]
        hi = np.nonzero(bx >= self.block_hi)[0]
        out = (self._records(lo), self._records(hi))
        keep = np.nonzero((bx >= self.block_lo) & (bx < self.block_hi))[0]
        arrs = {k: self.st.arr[k][keep] for k in FIELDS}
        aff = self.st.arr["cdf_affinity"][keep]
        self.gids = self.gids[keep]
        self._make_state(arrs, aff)
        return tuple(torch.from_numpy(np.ascontiguousarray(o.reshape(-1))) for o in out)

    def add_migrants(self, in_lower, in_upper):
        for recs in (in_lower, in_upper):
            self._add_migrants(recs)

    def _add_migrants(self, recs):
        if recs is None or recs.numel() == 0:
            return
        a = self.st.arr
        widths = [int(np.prod(a[k].shape[1:])) if a[k].ndim > 1 else 1 for k in FIELDS]
        w = sum(widths) + 2
        rec = recs.numpy().reshape(-1, w)
        arrs, off = {}, 0
        for k, wd in zip(FIELDS, widths):
            new = rec[:, off:off + wd].reshape((len(rec),) + a[k].shape[1:])
            arrs[k] = np.concatenate([a[k], new.astype(a[k].dtype)])
            off += wd
        aff = np.concatenate([a["cdf_affinity"], rec[:, off].astype(np.uint32)])
        self.gids = np.concatenate([self.gids, rec[:, off + 1].astype(np.int64)])
        self._make_state(arrs, aff)

    def export(self):
        a = self.st.arr
        return dict(ids=self.gids.copy(), pos=a["pos"].copy(), vel=a["vel"].copy(), def_grad=a["def_grad"].copy(),
                    affine=a["affine"].copy())
