"""CPU checker of the multi-GPU substep protocol (wgsparkl_amd/csrc/kernels_shard.h, capi_sharded.inc) — the SAME
protocol restated on top of the C oracle's passes, so that the decomposition can be exercised under
torch.distributed (gloo) without a GPU, in fp64, against the single-domain oracle run:

  sort .. P2G over everything the rank holds (its core particles + the GUESTS that left the core range one substep ago)
  | message per neighbour = partial node sums of the shared x-layer pairs + the guests' records (state before this G2P)
  | add what the neighbours sent | grid update | G2P + particle update of the core particles AND of the arrivals (from
  the node values this rank holds, or — where a block is not active here — from the sender's partial sums) | guests dropped.

TEST INFRASTRUCTURE ONLY (imports the oracle)."""
import numpy as np
import torch

from oracle.orc import Oracle
from wgsparkl_amd.sharded import INT_MAX, INT_MIN, associated_block_x
from wgsparkl_amd.solver import ParticleSet

FIELDS = ("pos", "vel", "def_grad", "affine", "cdf_normal", "cdf_rigid_vel", "cdf_dist", "init_volume", "mass",
          "lambda_", "mu", "dp", "dp_state", "phase")


def iface_masks(bx, lo, hi, has_lo, has_hi, ntag):
    """kernels_shard.h iface_masks: (recv, send_lo, send_hi) bit masks over the x-layer pairs of block layer bx."""
    ALL = (1 << ntag) - 1
    rc = sl = sh = 0
    if has_lo:
        if bx == lo - 1: sl |= ALL
        if bx == lo: sl |= 1; rc |= ALL
        if bx == lo + 1: rc |= 1
    if has_hi:
        if bx == hi - 1: rc |= ALL
        if bx == hi: sh |= ALL; rc |= 1
        if bx == hi + 1: sh |= 1
    return rc, sl, sh


class OracleShard:
    def __init__(self, scene, sub: ParticleSet, gids, block_lo, block_hi, has_lower, has_upper, dtype=np.float64):
        self.scene, self.dtype = scene, np.dtype(dtype)
        self.dim = sub.dim
        assert not scene["colliders"], "the CPU checker covers the collider-free protocol"
        self.orc = Oracle(self.dim, dtype)
        self.lo, self.hi = int(block_lo), int(block_hi)
        self.has_lower, self.has_upper = has_lower, has_upper
        if has_lower and has_upper:
            assert self.hi - self.lo >= 3, "a slab with two neighbours must be at least 3 blocks wide"
        self.bw = 4 if self.dim == 3 else 8
        self.ntag = self.bw // 2
        t = np.arange(64)
        self.node_lx = t % self.bw                      # x coordinate of a node inside its block
        self.gids = np.asarray(gids, np.int64)
        self.st = self._make_state({k: np.asarray(getattr(sub, k), dtype) for k in FIELDS}, sub.cdf_affinity.copy())
        self.guest = np.zeros(len(self.gids), bool)     # residents that left the core range in the last G2P
        self.sent = 0

    def _make_state(self, arrs, aff):
        n = len(aff)
        ps = ParticleSet(dim=self.dim, cdf_affinity=aff.astype(np.uint32), init_radius=np.zeros(n, np.float32),
                         has_plasticity=np.ones(n, bool), has_phase=np.ones(n, bool), **arrs)
        sc = self.scene
        return self.orc.new_state(ps, sc["params"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc.get("model", 0))

    # ---- phase A: sort .. P2G, messages
    def begin(self):
        """Returns (to_lower, to_upper): flat float64 tensors, None where there is no neighbour."""
        st, D = self.st, self.dim
        st.sort(); st.grid_update_cdf(); st.g2p_cdf(); st.p2g()
        nb = st.n_blocks
        vid = st.g["block_vid"][:nb]
        mv = st.g["node_mv"].reshape(-1, 64, D + 1)
        halo = ([], [])
        for b in range(nb):
            _, sl, sh = iface_masks(int(vid[b, 0]), self.lo, self.hi, self.has_lower, self.has_upper, self.ntag)
            for f, send in ((0, sl), (1, sh)):
                for tag in range(self.ntag):
                    if not (send >> tag) & 1:
                        continue
                    nodes = np.nonzero(self.node_lx // 2 == tag)[0]
                    vals = mv[b, nodes, :]
                    regular = tag == 0 and int(vid[b, 0]) == (self.lo if f == 0 else self.hi)
                    if regular or np.any(vals != 0):
                        halo[f].append(np.concatenate([vid[b].astype(np.float64), [float(tag)], vals.reshape(-1)]))
        # the guests' records: state BEFORE this substep's G2P
        bx = associated_block_x(st.arr["pos"].astype(np.float32), self.scene["cell_width"], D)
        assert np.array_equal(self.guest, (bx < self.lo) | (bx >= self.hi)), "the guest list must be the residents outside the core range"
        out = []
        self.sent = int(self.guest.sum())
        for f, sel in ((0, bx < self.lo), (1, bx >= self.hi)):
            if not (self.has_lower if f == 0 else self.has_upper):
                assert not sel.any(), "a particle left the decomposition"
                out.append(None)
                continue
            recs = self._records(np.nonzero(sel)[0])
            h = np.stack(halo[f]) if halo[f] else np.zeros((0, 0))
            msg = np.concatenate([[float(len(h)), float(len(recs))], h.reshape(-1), recs.reshape(-1)])
            out.append(torch.from_numpy(np.ascontiguousarray(msg)))
        return tuple(out)

    def _records(self, idx):
        a = self.st.arr
        width = lambda k: int(np.prod(a[k].shape[1:])) if a[k].ndim > 1 else 1
        cols = [a[k][idx].reshape(len(idx), width(k)) for k in FIELDS]
        cols += [a["cdf_affinity"][idx].reshape(len(idx), 1).astype(np.float64), self.gids[idx].reshape(len(idx), 1).astype(np.float64)]
        return np.concatenate(cols, axis=1)

    def _parse(self, msg):
        D = self.dim
        nodes_per = 64 // self.ntag
        m = msg.numpy()
        nh, nm = int(m[0]), int(m[1])
        wh = D + 1 + nodes_per * (D + 1)
        halo = m[2:2 + nh * wh].reshape(nh, wh)
        a = self.st.arr
        widths = [int(np.prod(a[k].shape[1:])) if a[k].ndim > 1 else 1 for k in FIELDS]
        wr = sum(widths) + 2
        recs = m[2 + nh * wh:2 + nh * wh + nm * wr].reshape(nm, wr)
        return halo, recs, widths

    # ---- phase B: add, grid update, G2P of the core particles and the arrivals, guests dropped
    def end(self, from_lower, from_upper):
        st, D = self.st, self.dim
        nb = st.n_blocks
        vid = st.g["block_vid"][:nb]
        index = {tuple(int(x) for x in v): b for b, v in enumerate(vid)}
        mv = st.g["node_mv"].reshape(-1, 64, D + 1)
        nodes_per = 64 // self.ntag
        orphans = {}                                   # (block vid, pair) -> partial sums of a block not active here
        arrivals = []
        widths = None
        for msg in (from_lower, from_upper):
            if msg is None:
                continue
            halo, recs, widths = self._parse(msg)
            for r in halo:
                key, tag = tuple(int(x) for x in r[:D]), int(r[D])
                vals = r[D + 1:].reshape(nodes_per, D + 1)
                nodes = np.nonzero(self.node_lx // 2 == tag)[0]
                b = index.get(key)
                if b is not None:
                    rc, _, _ = iface_masks(key[0], self.lo, self.hi, self.has_lower, self.has_upper, self.ntag)
                    assert (rc >> tag) & 1, "a neighbour sent a pair this slab's grid update would not add"
                    mv[b, nodes, :] += vals
                else:
                    orphans[(key, tag)] = vals
            if len(recs):
                arrivals.append(recs)
        st.grid_update()
        # arrivals: a tiny state of their own whose grid is filled with this rank's node velocities (or the sender's totals)
        new_arr = None
        if arrivals:
            rec = np.concatenate(arrivals)
            arrs, off = {}, 0
            for k, wd in zip(FIELDS, widths):
                shape = (len(rec),) + st.arr[k].shape[1:]
                arrs[k] = rec[:, off:off + wd].reshape(shape).astype(self.dtype)
                off += wd
            aff = rec[:, off].astype(np.uint32)
            gid = rec[:, off + 1].astype(np.int64)
            tiny = self._make_state(arrs, aff)
            tiny.sort()
            tnb = tiny.n_blocks
            tvid = tiny.g["block_vid"][:tnb]
            tmv = tiny.g["node_mv"].reshape(-1, 64, D + 1)
            prm = self.scene["params"]
            g, dt, h = np.asarray(prm.gravity, np.float64), float(prm.dt), float(self.scene["cell_width"])
            for tb in range(tnb):
                key = tuple(int(x) for x in tvid[tb])
                b = index.get(key)
                if b is not None:
                    tmv[tb] = mv[b]                    # velocities: the grid update ran on the totals
                    continue
                tmv[tb] = 0.0
                for tag in range(self.ntag):
                    vals = orphans.get((key, tag))
                    if vals is None:
                        continue
                    nodes = np.nonzero(self.node_lx // 2 == tag)[0]
                    mass = vals[:, D]
                    with np.errstate(divide="ignore", invalid="ignore"):
                        inv = np.where(mass > 0, 1.0 / mass, 0.0)
                    vel = np.clip((vals[:, :D] + mass[:, None] * g[None, :] * dt) * inv[:, None], -h / dt, h / dt)
                    tmv[tb, nodes, :D] = vel
                    tmv[tb, nodes, D] = mass
            tiny.g2p(); tiny.particle_update()
            new_arr = (tiny, gid)
        st.g2p(); st.particle_update()
        keep = ~self.guest                              # the guests now live on the neighbour
        arrs = {k: st.arr[k][keep] for k in FIELDS}
        aff = st.arr["cdf_affinity"][keep]
        gids = self.gids[keep]
        if new_arr is not None:
            tiny, gid = new_arr
            arrs = {k: np.concatenate([arrs[k], tiny.arr[k]]) for k in FIELDS}
            aff = np.concatenate([aff, tiny.arr["cdf_affinity"]])
            gids = np.concatenate([gids, gid])
        self.gids = gids
        self.st = self._make_state(arrs, aff)
        bx = associated_block_x(self.st.arr["pos"].astype(np.float32), self.scene["cell_width"], D)
        self.guest = (bx < self.lo) | (bx >= self.hi)
        assert ((bx >= self.lo - 1) & (bx <= self.hi)).all(), "a particle moved more than one block in a substep"

    def export(self):
        a = self.st.arr
        return dict(ids=self.gids.copy(), pos=a["pos"].copy(), vel=a["vel"].copy(), def_grad=a["def_grad"].copy(),
                    affine=a["affine"].copy())


class DistExchange:
    """Neighbour exchange over torch.distributed point-to-point ops (gloo on CPU): sizes, then payloads."""

    def __init__(self, dist, rank: int, world: int):
        self.dist, self.rank, self.world = dist, rank, world

    def __call__(self, to_lower, to_upper):
        dist = self.dist
        lower = self.rank - 1 if self.rank > 0 else None
        upper = self.rank + 1 if self.rank < self.world - 1 else None
        sends = {lower: to_lower, upper: to_upper}
        ops, rsize = [], {}
        for peer in (lower, upper):
            if peer is None:
                continue
            n_out = torch.tensor([int(sends[peer].numel())], dtype=torch.int64)
            rsize[peer] = torch.zeros(1, dtype=torch.int64)
            ops += [dist.P2POp(dist.isend, n_out, peer), dist.P2POp(dist.irecv, rsize[peer], peer)]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        ops, recv = [], {}
        for peer in (lower, upper):
            if peer is None:
                continue
            recv[peer] = torch.empty(int(rsize[peer].item()), dtype=torch.float64)
            ops += [dist.P2POp(dist.isend, sends[peer].contiguous(), peer), dist.P2POp(dist.irecv, recv[peer], peer)]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return recv.get(lower), recv.get(upper)


def lockstep(shards, k):
    """All ranks inside one process (no transport): what the gloo run must equal, and a quick CPU check of the protocol."""
    n = len(shards)
    for _ in range(k):
        msgs = [s.begin() for s in shards]
        for r, s in enumerate(shards):
            s.end(msgs[r - 1][1] if r > 0 else None, msgs[r + 1][0] if r < n - 1 else None)
