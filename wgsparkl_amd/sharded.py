"""Multi-GPU driver: x-slab domain decomposition of the MLS-MPM substep, one process per GPU.

NEW DESIGN — the reference is single-GPU (one wgpu::Device, src/pipeline.rs:176-193).
The protocol is written against a small backend interface so that the very same code drives
  * `GpuShard` (this file): a sharded `wgs_data` behind the C ABI, torch tensors as exchange buffers,
    `torch.distributed` (backend "nccl" = RCCL over xGMI) as transport;
  * a CPU checker backend in tests/ (world_size-2 gloo test, no GPU needed).

Why x slabs and what crosses a face (see wgsparkl_amd/csrc/kernels_shard.h): a particle with associated cell c
touches nodes c..c+2 only, so a rank's particles reach the first two node layers of the block layer owned by
the NEXT rank and nothing on the lower side. Per substep:
  1. after P2G both neighbours swap the partial (momentum, mass) sums of those two node layers and add
     them (a + b == b + a bitwise, so both hold identical totals and update them redundantly);
  2. after the particle update, particles whose associated block left the rank's range move (full state).
Bytes per face and substep: 528 B per active interface block (3D) + ~200 B per migrating particle —
two point-to-point messages per neighbour, no collective on the data path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _ffi
from .models import MODEL_COROTATED
from .pipeline import MpmPipeline, _fill_collider, _pack_particles
from .solver import ParticleSet, SimulationParams

INT_MIN, INT_MAX = -(2 ** 31), 2 ** 31 - 1


@dataclass
class SlabPartition:
    """Contiguous block ranges along x: rank r owns bx in [cuts[r], cuts[r+1]); the two ends are open."""
    cuts: Sequence[int]          # world + 1 entries, in BLOCK units (4 cells in 3D, 8 in 2D)

    @property
    def world(self) -> int:
        return len(self.cuts) - 1

    def block_range(self, rank: int):
        lo = INT_MIN if rank == 0 else int(self.cuts[rank])
        hi = INT_MAX if rank == self.world - 1 else int(self.cuts[rank + 1])
        return lo, hi

    def owner_of_blocks(self, bx: np.ndarray) -> np.ndarray:
        inner = np.asarray(self.cuts[1:-1], np.int64)
        return np.searchsorted(inner, np.asarray(bx, np.int64), side="right")

    @staticmethod
    def balanced(block_x: np.ndarray, world: int) -> "SlabPartition":
        """Cuts at particle-count quantiles of the associated block x coordinate (balance by particle
        count, not by volume)."""
        bx = np.sort(np.asarray(block_x, np.int64))
        cuts = [int(bx[0])]
        for r in range(1, world):
            cuts.append(int(bx[(len(bx) * r) // world]))
        cuts.append(int(bx[-1]) + 1)
        for i in range(1, len(cuts)):          # strictly increasing
            cuts[i] = max(cuts[i], cuts[i - 1] + 1)
        return SlabPartition(cuts)


def associated_block_x(pos: np.ndarray, cell_width: float, dim: int) -> np.ndarray:
    """floor((round(x / h) - 1) / BW) with the reference's fp32 rule (particle3d.wgsl:41-49, grid.wgsl:284-292)."""
    bw = 4 if dim == 3 else 8
    c = np.rint(pos[:, 0].astype(np.float32) / np.float32(cell_width)) - np.float32(1.0)
    return np.floor(c / np.float32(bw)).astype(np.int64)


# ------------------------------------------------------------------------------------------------
# protocol (backend-agnostic)
# ------------------------------------------------------------------------------------------------
def _pack_halos(backend):
    if hasattr(backend, "pack_halos"):
        return backend.pack_halos()
    return (backend.pack_halo(backend.block_lo) if backend.has_lower else None,
            backend.pack_halo(backend.block_hi) if backend.has_upper else None)


def _add_halos(backend, from_lower, from_upper):
    if hasattr(backend, "add_halos"):
        backend.add_halos(from_lower, from_upper)
        return
    if from_lower is not None:
        backend.add_halo(from_lower)
    if from_upper is not None:
        backend.add_halo(from_upper)


def substep_phases(backend, exchange):
    """One substep of one rank. `exchange(to_lower, to_upper) -> (from_lower, from_upper)` moves opaque
    record buffers between neighbours (None where there is no neighbour)."""
    backend.step_begin()
    to_lower, to_upper = _pack_halos(backend)
    _add_halos(backend, *exchange(to_lower, to_upper))
    backend.step_end()
    out_lower, out_upper = backend.pack_migrants()
    in_lower, in_upper = exchange(out_lower if backend.has_lower else None, out_upper if backend.has_upper else None)
    backend.add_migrants(in_lower, in_upper)


def pipelined_substep(backend, exchange, pending):
    """`substep_phases` with the particle migration of the PREVIOUS substep still in flight while the residents
    are re-binned (`pending` = what the previous call returned, None at the start). Returns the handle of this
    substep's migration; `finish_migration` must absorb the last one before the state is read."""
    backend.bin_residents()                       # overlaps the messages in flight
    if pending is not None:
        finish_migration(backend, pending)
    backend.step_begin()
    to_lower, to_upper = _pack_halos(backend)
    _add_halos(backend, *exchange(to_lower, to_upper))
    backend.step_end()
    out_lower, out_upper = backend.pack_migrants()
    return exchange.start(out_lower if backend.has_lower else None, out_upper if backend.has_upper else None)


def finish_migration(backend, pending):
    in_lower, in_upper = pending.finish()
    backend.add_migrants(in_lower, in_upper)


def _transported(buf):
    """What a transport does: the receiver gets its own copy of the message (the sender resets and refills its
    outgoing buffer in its next substep, possibly before the receiver has consumed the message)."""
    return buf.clone() if hasattr(buf, "clone") else buf.copy()


def lockstep_substep(backends: List):
    """All ranks inside ONE process (tests, single-GPU emulation of the decomposition): runs the
    phases of every rank in lockstep and routes the messages directly."""
    n = len(backends)
    for b in backends:
        b.step_begin()
    packed = [_pack_halos(b) for b in backends]          # (to_lower, to_upper) of every rank
    for r, b in enumerate(backends):
        _add_halos(b, packed[r - 1][1] if r > 0 else None, packed[r + 1][0] if r < n - 1 else None)
    for b in backends:
        b.step_end()
    mig = [b.pack_migrants() for b in backends]
    for r, b in enumerate(backends):
        b.add_migrants(_transported(mig[r - 1][1]) if r > 0 else None, _transported(mig[r + 1][0]) if r < n - 1 else None)


def lockstep_pipelined_substep(backends: List, pending):
    """`lockstep_substep` in the order of `pipelined_substep`: residents re-binned before the previous substep's
    migrants are absorbed. `pending` = the return value of the previous call (None at the start); finish a run
    with `lockstep_finish(backends, pending)`."""
    n = len(backends)
    for b in backends:
        b.bin_residents()
    if pending is not None:
        lockstep_finish(backends, pending)
    for b in backends:
        b.step_begin()
    packed = [_pack_halos(b) for b in backends]          # (to_lower, to_upper) of every rank
    for r, b in enumerate(backends):
        _add_halos(b, packed[r - 1][1] if r > 0 else None, packed[r + 1][0] if r < n - 1 else None)
    for b in backends:
        b.step_end()
    mig = [b.pack_migrants() for b in backends]
    return [(_transported(mig[r - 1][1]) if r > 0 else None, _transported(mig[r + 1][0]) if r < n - 1 else None) for r in range(n)]


def lockstep_finish(backends: List, pending):
    for b, (in_lower, in_upper) in zip(backends, pending):
        b.add_migrants(in_lower, in_upper)


class _PendingExchange:
    def __init__(self, works, from_lower, from_upper):
        self.works, self.from_lower, self.from_upper = works, from_lower, from_upper

    def finish(self):
        for w in self.works:
            w.wait()
        self.works = []
        return self.from_lower, self.from_upper


class FixedExchange:
    """Neighbour exchange of FIXED-SIZE device buffers (the record count travels inside the buffer), one
    send + one receive per neighbour, no size handshake and no host synchronisation: the point-to-point
    ops are ordered on the current stream like the kernels. torch.distributed "nccl" = RCCL on ROCm; the
    two neighbours are distinct peers, so each message rides its own xGMI link."""

    def __init__(self, dist, rank: int, world: int):
        import torch
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self.lower = rank - 1 if rank > 0 else None
        self.upper = rank + 1 if rank < world - 1 else None
        # send buffers are persistent (GpuShard owns them), so the receive buffers and the P2POp lists are
        # built once per (lower buffer, upper buffer) pair and reused: two receive sets alternate, the
        # previous one may still be read by a kernel enqueued on the stream
        self._plans = {}

    def _plan(self, to_lower, to_upper):
        torch, dist = self.torch, self.dist
        key = (None if to_lower is None else to_lower.data_ptr(), None if to_upper is None else to_upper.data_ptr())
        plan = self._plans.get(key)
        if plan is None:
            sets = []
            for _ in range(2):
                ops, from_lower, from_upper = [], None, None
                # every rank uses the same buffer capacities, so both directions of a face carry same-sized messages
                if self.lower is not None and to_lower is not None:
                    from_lower = torch.empty_like(to_lower)
                    ops += [dist.P2POp(dist.isend, to_lower, self.lower), dist.P2POp(dist.irecv, from_lower, self.lower)]
                if self.upper is not None and to_upper is not None:
                    from_upper = torch.empty_like(to_upper)
                    ops += [dist.P2POp(dist.isend, to_upper, self.upper), dist.P2POp(dist.irecv, from_upper, self.upper)]
                sets.append((ops, from_lower, from_upper))
            plan = self._plans[key] = [sets, 0]
        sets, turn = plan
        plan[1] = turn ^ 1
        return sets[turn]

    def __call__(self, to_lower, to_upper):
        return self.start(to_lower, to_upper).finish()

    def start(self, to_lower, to_upper):
        """Issue the sends / receives and return at once; `.finish()` makes the current stream wait for them
        (it does not block the host with NCCL/RCCL) and hands out the received buffers."""
        ops, from_lower, from_upper = self._plan(to_lower, to_upper)
        works = self.dist.batch_isend_irecv(ops) if ops else []
        return _PendingExchange(works, from_lower, from_upper)


class _PendingOnStream:
    """Result of RcclExchange.start: `.finish()` makes the current stream wait for the side stream the transfer runs on."""
    __slots__ = ("torch", "side", "from_lower", "from_upper")

    def __init__(self, torch, side, from_lower, from_upper):
        self.torch, self.side, self.from_lower, self.from_upper = torch, side, from_lower, from_upper

    def finish(self):
        if self.side is not None:
            self.torch.cuda.current_stream().wait_stream(self.side)
            self.side = None
        return self.from_lower, self.from_upper


class RcclExchange:
    """The same fixed-size neighbour exchange as FixedExchange, issued straight on RCCL (`ncclSend` / `ncclRecv` inside
    one group per exchange, through ctypes on the librccl.so that torch itself loaded) instead of through
    torch.distributed's point-to-point wrappers, whose host-side cost (work objects, coalescing, watchdog) would
    otherwise bound a 200 us substep from the CPU side. Two communicators: the halo exchange runs on the substep's own
    stream (it is on the critical path anyway), the particle migration on a side stream so that it overlaps the next
    substep's re-binning. torch.distributed is only used to hand the unique ids around."""

    NCCL_FLOAT32 = 7

    def __init__(self, dist, rank: int, world: int):
        import os
        import torch
        self.torch, self.rank, self.world = torch, rank, world
        self.lower = rank - 1 if rank > 0 else None
        self.upper = rank + 1 if rank < world - 1 else None
        lib = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]

        lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        lib.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetErrorString.argtypes = [C.c_int]
        self.lib = lib
        dev = torch.device("cuda", torch.cuda.current_device())
        self.comms = []
        for _ in range(2):                                   # halo, migration
            uid = UniqueId()
            if rank == 0:
                self._check(lib.ncclGetUniqueId(C.byref(uid)))
            t = torch.tensor(list(C.string_at(C.byref(uid), 128)) if rank == 0 else [0] * 128, dtype=torch.uint8)  # all 128 bytes
            t = t.to(dev) if dist.get_backend() == "nccl" else t
            dist.broadcast(t, 0)
            raw = bytes(t.cpu().numpy().tobytes())
            uid = UniqueId()
            C.memmove(C.byref(uid), raw, 128)
            comm = C.c_void_p()
            self._check(lib.ncclCommInitRank(C.byref(comm), world, uid, rank))
            self.comms.append(comm)
        self.side = torch.cuda.Stream(device=dev)
        self._plans = {}

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("RCCL: " + self.lib.ncclGetErrorString(rc).decode())

    def _plan(self, to_lower, to_upper):
        torch = self.torch
        key = (None if to_lower is None else to_lower.data_ptr(), None if to_upper is None else to_upper.data_ptr())
        plan = self._plans.get(key)
        if plan is None:
            sets = []
            for _ in range(2):                               # two receive sets alternate (see FixedExchange)
                ops, from_lower, from_upper = [], None, None
                if self.lower is not None and to_lower is not None:
                    from_lower = torch.empty_like(to_lower)
                    ops.append((to_lower.data_ptr(), from_lower.data_ptr(), to_lower.numel(), self.lower))
                if self.upper is not None and to_upper is not None:
                    from_upper = torch.empty_like(to_upper)
                    ops.append((to_upper.data_ptr(), from_upper.data_ptr(), to_upper.numel(), self.upper))
                sets.append((ops, from_lower, from_upper))
            plan = self._plans[key] = [sets, 0]
        sets, turn = plan
        plan[1] = turn ^ 1
        return sets[turn]

    def _issue(self, ops, comm, stream_ptr):
        lib = self.lib
        if not ops:
            return
        self._check(lib.ncclGroupStart())
        for send_ptr, recv_ptr, count, peer in ops:
            self._check(lib.ncclSend(send_ptr, count, self.NCCL_FLOAT32, peer, comm, stream_ptr))
            self._check(lib.ncclRecv(recv_ptr, count, self.NCCL_FLOAT32, peer, comm, stream_ptr))
        self._check(lib.ncclGroupEnd())

    def __call__(self, to_lower, to_upper):
        """Halo exchange: on the current stream, ordered with the kernels before and after it."""
        ops, from_lower, from_upper = self._plan(to_lower, to_upper)
        self._issue(ops, self.comms[0], C.c_void_p(self.torch.cuda.current_stream().cuda_stream))
        return from_lower, from_upper

    def start(self, to_lower, to_upper):
        """Migration: on the side stream, after everything enqueued so far; `.finish()` makes the current stream wait."""
        torch = self.torch
        ops, from_lower, from_upper = self._plan(to_lower, to_upper)
        cur = torch.cuda.current_stream()
        if ops:
            self.side.wait_stream(cur)
            self._issue(ops, self.comms[1], C.c_void_p(self.side.cuda_stream))
        return _PendingOnStream(self.torch, self.side if ops else None, from_lower, from_upper)

    def selftest(self):
        """One send + receive to this very rank inside a group on both communicators (exercises every entry point
        this class binds; used on 1-GPU boxes where no second rank can exist)."""
        torch = self.torch
        a = torch.arange(1024, dtype=torch.float32, device="cuda")
        for comm, stream in ((self.comms[0], torch.cuda.current_stream()), (self.comms[1], self.side)):
            b = torch.zeros_like(a)
            stream.wait_stream(torch.cuda.current_stream())
            self._issue([(a.data_ptr(), b.data_ptr(), a.numel(), self.rank)], comm, C.c_void_p(stream.cuda_stream))
            torch.cuda.current_stream().wait_stream(stream)
            torch.cuda.synchronize()
            if not bool((a == b).all()):
                raise RuntimeError("RCCL self send/recv returned wrong data")
        return True

    def neighbour_test(self):
        """One real exchange with both neighbours on each communicator: every rank sends its own rank and must receive
        rank - 1 / rank + 1. Run once before the transport is trusted with particle data."""
        torch = self.torch
        mine = torch.full((256,), float(self.rank), dtype=torch.float32, device="cuda")
        for use_side in (False, True):
            if use_side:
                got_lo, got_hi = self.start(mine, mine).finish()
            else:
                got_lo, got_hi = self(mine, mine)
            torch.cuda.synchronize()
            for got, peer in ((got_lo, self.lower), (got_hi, self.upper)):
                if peer is not None and not bool((got == float(peer)).all()):
                    raise RuntimeError(f"RCCL neighbour exchange with rank {peer} returned wrong data")
        self._plans.clear()      # (plans are keyed by buffer address: the test tensor's address may be reused)
        return True

    def close(self):
        for comm in self.comms:
            self.lib.ncclCommDestroy(comm)
        self.comms = []


class DistExchange:
    """Neighbour exchange over torch.distributed point-to-point ops (nccl = RCCL on ROCm, gloo on CPU).
    Message = [count] then `count` records; neighbours are distinct peers, one xGMI link each."""

    def __init__(self, dist, rank: int, world: int, device, dtype=None):
        import torch
        self.torch, self.dist, self.rank, self.world, self.device = torch, dist, rank, world, device
        self.dtype = dtype or torch.float32

    def start(self, to_lower, to_upper):
        """Same interface as FixedExchange.start; this transport needs a size handshake, so it completes here."""
        return _PendingExchange([], *self(to_lower, to_upper))

    def __call__(self, to_lower, to_upper):
        torch, dist = self.torch, self.dist
        lower = self.rank - 1 if self.rank > 0 else None
        upper = self.rank + 1 if self.rank < self.world - 1 else None
        sends = {lower: to_lower, upper: to_upper}
        # 1. sizes
        ops, rsize = [], {}
        for peer in (lower, upper):
            if peer is None:
                continue
            t = sends[peer]
            n_out = torch.tensor([0 if t is None else int(t.numel())], dtype=torch.int64, device=self.device)
            rsize[peer] = torch.zeros(1, dtype=torch.int64, device=self.device)
            ops += [dist.P2POp(dist.isend, n_out, peer), dist.P2POp(dist.irecv, rsize[peer], peer)]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        # 2. payloads
        ops, recv = [], {}
        for peer in (lower, upper):
            if peer is None:
                continue
            t = sends[peer]
            n_in = int(rsize[peer].item())
            recv[peer] = torch.empty(n_in, dtype=self.dtype, device=self.device)
            if t is not None and t.numel() > 0:
                ops.append(dist.P2POp(dist.isend, t.contiguous(), peer))
            if n_in > 0:
                ops.append(dist.P2POp(dist.irecv, recv[peer], peer))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return recv.get(lower), recv.get(upper)


# ------------------------------------------------------------------------------------------------
# GPU backend
# ------------------------------------------------------------------------------------------------
class GpuShard:
    """One rank's slab on one MI355X: a sharded `wgs_data` + torch device buffers for the exchanges."""

    def __init__(self, pipeline: MpmPipeline, params: SimulationParams, particles: ParticleSet, global_ids: np.ndarray,
                 colliders, cell_width: float, grid_capacity: int, block_lo: int, block_hi: int, has_lower: bool,
                 has_upper: bool, particle_capacity: int, model: int = MODEL_COROTATED, force_plastic: bool = False,
                 halo_capacity_blocks: int = 0, migrant_capacity: int = 0):
        import torch
        self.torch = torch
        self.pipeline, self.lib, self.T = pipeline, pipeline.lib, pipeline.T
        T, D = self.T, pipeline.dim
        self.dim = D
        self.block_lo, self.block_hi = int(block_lo), int(block_hi)
        self.has_lower, self.has_upper = has_lower, has_upper
        sp = T.SimParams()
        sp.gravity = (C.c_float * D)(*params.gravity)
        sp.dt = params.dt
        raw = _pack_particles(T, particles)
        gids = np.ascontiguousarray(global_ids, np.uint32)
        cols = (T.Collider * max(1, len(colliders)))()
        for i, c in enumerate(colliders):
            _fill_collider(T, cols[i], c, D)
        h = C.c_void_p()
        cap = max(int(particle_capacity), particles.n)
        _ffi.check(self.lib, self.lib.wgs_data_create_sharded(
            pipeline._h, C.byref(sp), raw.ctypes.data_as(C.POINTER(T.Particle)), particles.n,
            gids.ctypes.data_as(C.POINTER(C.c_uint32)), cols, len(colliders), float(cell_width), int(grid_capacity),
            cap, max(self.block_lo, INT_MIN), min(self.block_hi, INT_MAX), 1 if force_plastic else 0, C.byref(h)))
        self._h = h
        if model != MODEL_COROTATED:
            _ffi.check(self.lib, self.lib.wgs_set_constitutive_model(self._h, int(model)))
        self.halo_rec = self.lib.wgs_shard_halo_record_bytes() // 4
        self.part_rec = self.lib.wgs_shard_particle_record_bytes() // 4
        self.hdr = self.lib.wgs_shard_buffer_header_bytes() // 4
        self.capacity = cap
        dev = torch.device("cuda", pipeline.device)
        self.device = dev
        # Same capacities on every rank: the messages are fixed-size (count in the header).
        self.halo_cap = int(halo_capacity_blocks) or 4096
        self.mig_cap = int(migrant_capacity) or 4096
        f32 = torch.float32
        self._halo_out = [torch.zeros(self.hdr + self.halo_cap * self.halo_rec, dtype=f32, device=dev) for _ in range(2)]
        self._mig_out = [torch.zeros(self.hdr + self.mig_cap * self.part_rec, dtype=f32, device=dev) for _ in range(2)]
        self._keep = []   # received tensors stay alive until the stream has consumed them
        # the outgoing buffers are reused every substep: their record counts are reset inside the substep
        ptr = lambda t, ok: C.c_void_p(t.data_ptr()) if ok else None
        _ffi.check(self.lib, self.lib.wgs_shard_register_buffers(
            self._h, ptr(self._halo_out[0], has_lower), ptr(self._halo_out[1], has_upper),
            ptr(self._mig_out[0], True), ptr(self._mig_out[1], True)))
        # kernels and RCCL messages are ordered on torch's current stream: no host sync inside a substep
        _ffi.check(self.lib, self.lib.wgs_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))

    # -- protocol (all asynchronous)
    def bin_residents(self):
        _ffi.check(self.lib, self.lib.wgs_shard_bin_residents(self.pipeline._h, self._h))

    def step_begin(self):
        self._keep = self._keep[-8:]
        _ffi.check(self.lib, self.lib.wgs_shard_step_begin(self.pipeline._h, self._h))

    def pack_halo(self, layer_bx: int):
        buf = self._halo_out[0 if layer_bx == self.block_lo else 1]
        _ffi.check(self.lib, self.lib.wgs_shard_pack_halo(self._h, int(layer_bx), C.c_void_p(buf.data_ptr()), self.halo_cap))
        return buf

    def pack_halos(self):
        """Both faces in one launch -> (to_lower, to_upper), None where there is no neighbour."""
        lo = self._halo_out[0] if self.has_lower else None
        hi = self._halo_out[1] if self.has_upper else None
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        _ffi.check(self.lib, self.lib.wgs_shard_pack_halos(self._h, ptr(lo), ptr(hi), self.halo_cap))
        return lo, hi

    def add_halos(self, from_lower, from_upper):
        """Both neighbours' partial sums in one launch."""
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self._keep += [t for t in (from_lower, from_upper) if t is not None]
        _ffi.check(self.lib, self.lib.wgs_shard_add_halos(self._h, ptr(from_lower), ptr(from_upper), self.halo_cap))

    def add_halo(self, buf):
        if buf is None:
            return
        self._keep.append(buf)
        _ffi.check(self.lib, self.lib.wgs_shard_add_halo(self._h, C.c_void_p(buf.data_ptr()), self.halo_cap))

    def step_end(self):
        _ffi.check(self.lib, self.lib.wgs_shard_step_end(self.pipeline._h, self._h))

    def pack_migrants(self):
        _ffi.check(self.lib, self.lib.wgs_shard_pack_migrants(self._h, C.c_void_p(self._mig_out[0].data_ptr()),
                                                               C.c_void_p(self._mig_out[1].data_ptr()), self.mig_cap))
        return self._mig_out[0], self._mig_out[1]

    def add_migrants(self, in_lower, in_upper):
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self._keep += [t for t in (in_lower, in_upper) if t is not None]
        _ffi.check(self.lib, self.lib.wgs_shard_add_migrants(
            self._h, ptr(in_lower), ptr(in_upper), ptr(self._mig_out[0]) if self.has_lower else None,
            ptr(self._mig_out[1]) if self.has_upper else None, self.mig_cap))

    # -- host side
    def sync(self):
        _ffi.check(self.lib, self.lib.wgs_sync(self._h))

    def num_particles(self) -> int:
        s = self.T.Stats()
        _ffi.check(self.lib, self.lib.wgs_get_stats(self._h, C.byref(s)))
        return int(s.num_particles)

    def export(self):
        """(global ids, pos, vel, def_grad, affine, mass) of the particles this rank owns now (blocking)."""
        torch = self.torch
        buf = torch.zeros(self.hdr + self.capacity * self.part_rec, dtype=torch.float32, device=self.device)
        cnt = C.c_uint32(0)
        _ffi.check(self.lib, self.lib.wgs_shard_export(self._h, C.c_void_p(buf.data_ptr()), self.capacity, C.byref(cnt)))
        rec = buf[self.hdr: self.hdr + cnt.value * self.part_rec].cpu().numpy().reshape(cnt.value, self.part_rec)
        return unpack_records(rec, self.dim)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.wgs_data_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# The substep driven from inside the library (include/wgsparkl_hip.h: wgs_comm_*, wgs_shard_attach,
# wgs_sharded_step): what bench.py --gpus N and a Rust caller use. The classes above remain as the
# protocol checker (gloo / CPU backends in tests/) and as the per-phase view of the same entry points.
# ------------------------------------------------------------------------------------------------
class NativeComm:
    """RCCL communicator owned by the library (`wgs_comm_create`). The 128-byte unique id is made by rank 0 and
    handed around with whatever the host has — here torch.distributed's broadcast."""

    def __init__(self, pipeline: MpmPipeline, dist, rank: int, world: int, flags: int = 0):
        import torch
        self.lib, self.rank, self.world = pipeline.lib, rank, world
        uid = C.create_string_buffer(128)
        failure = None
        if rank == 0:
            try:
                _ffi.check(self.lib, self.lib.wgs_comm_get_unique_id(uid))
            except Exception as e:  # noqa: BLE001 — the other ranks are waiting in the broadcast below: send them zeros
                failure = e
                uid = C.create_string_buffer(128)
        if world > 1:
            t = torch.tensor(list(uid.raw), dtype=torch.uint8)
            if dist.get_backend() == "nccl":
                t = t.to(torch.device("cuda", pipeline.device))
            dist.broadcast(t, 0)
            uid = C.create_string_buffer(bytes(t.cpu().numpy().tobytes()), 128)
        if failure is not None or not any(uid.raw):
            raise RuntimeError(f"no RCCL unique id from rank 0 ({failure})")   # on every rank alike
        h = C.c_void_p()
        _ffi.check(self.lib, self.lib.wgs_comm_create(pipeline._h, uid, rank, world, int(flags), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.wgs_comm_destroy(self._h)
            self._h = None


class NativeShard:
    """One slab as a Rust caller would hold it: a sharded `wgs_data` whose message buffers and substep protocol live
    in the library. `comm` = NativeComm (one process per GPU) or None (a slab of a lockstep group / no neighbours)."""

    def __init__(self, pipeline: MpmPipeline, params: SimulationParams, particles: ParticleSet, global_ids: np.ndarray,
                 colliders, cell_width: float, grid_capacity: int, block_lo: int, block_hi: int, has_lower: bool,
                 has_upper: bool, particle_capacity: int, model: int = MODEL_COROTATED, force_plastic: bool = False,
                 halo_capacity_blocks: int = 4096, migrant_capacity: int = 4096, comm: Optional[NativeComm] = None,
                 uniform_material=None):
        """`uniform_material` = (mass, init_volume, lambda, mu) shared by EVERY particle of EVERY rank (the caller's
        promise: a rank only sees its own), or None: the constants then travel with each particle."""
        self.pipeline, self.lib, self.T = pipeline, pipeline.lib, pipeline.T
        T, D = self.T, pipeline.dim
        self.dim = D
        self.uniform_material = uniform_material
        self.n_colliders = len(colliders)
        sp = T.SimParams()
        sp.gravity = (C.c_float * D)(*params.gravity)
        sp.dt = params.dt
        raw = _pack_particles(T, particles)
        gids = np.ascontiguousarray(global_ids, np.uint32)
        cols = (T.Collider * max(1, len(colliders)))()
        for i, c in enumerate(colliders):
            _fill_collider(T, cols[i], c, D)
        h = C.c_void_p()
        cap = max(int(particle_capacity), particles.n)
        _ffi.check(self.lib, self.lib.wgs_data_create_sharded(
            pipeline._h, C.byref(sp), raw.ctypes.data_as(C.POINTER(T.Particle)), particles.n,
            gids.ctypes.data_as(C.POINTER(C.c_uint32)), cols, len(colliders), float(cell_width), int(grid_capacity),
            cap, max(int(block_lo), INT_MIN), min(int(block_hi), INT_MAX), 1 if force_plastic else 0, C.byref(h)))
        self._h = h
        self.capacity = cap
        if model != MODEL_COROTATED:
            _ffi.check(self.lib, self.lib.wgs_set_constitutive_model(self._h, int(model)))
        if uniform_material is not None:
            _ffi.check(self.lib, self.lib.wgs_set_uniform_material(self._h, *[float(x) for x in uniform_material]))
        if any(any(c.inv_mass) or any(c.inv_inertia_local) for c in colliders):   # dynamic bodies: two-way coupling
            arr = (T.MassProperties * len(colliders))()
            for i, c in enumerate(colliders):
                arr[i].inv_mass = tuple(c.inv_mass)
                arr[i].inv_inertia_local = tuple(c.inv_inertia_local)
            _ffi.check(self.lib, self.lib.wgs_set_body_mass_properties(self._h, arr, len(colliders)))
        _ffi.check(self.lib, self.lib.wgs_shard_attach(self._h, comm._h if comm is not None else None, 1 if has_lower else 0,
                                                         1 if has_upper else 0, int(halo_capacity_blocks), int(migrant_capacity)))
        # mesh colliders: sampled on the host like GpuRigidParticles::from_rapier; every rank holds every sample
        from .sampling import build_rigid_particles
        rb = build_rigid_particles(colliders, D, float(cell_width))
        if rb is not None:
            F32 = np.float32
            pts, ids = np.ascontiguousarray(rb["local_pts"], F32), np.ascontiguousarray(rb["ids"], np.uint32)
            vtx, vcol = np.ascontiguousarray(rb["local_vtx"], F32), np.ascontiguousarray(rb["vtx_collider"], np.uint32)
            fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
            _ffi.check(self.lib, self.lib.wgs_set_rigid_particles(self._h, pts.ctypes.data_as(fp), ids.ctypes.data_as(C.c_void_p), len(pts),
                                                                  vtx.ctypes.data_as(fp), vcol.ctypes.data_as(up), len(vtx)))
        self.part_rec = self.lib.wgs_shard_particle_record_bytes() // 4
        self.hdr = self.lib.wgs_shard_buffer_header_bytes() // 4

    def step(self, num_substeps: int):
        """`wgs_sharded_step`: whole substeps incl. both neighbour exchanges, asynchronous."""
        _ffi.check(self.lib, self.lib.wgs_sharded_step(self.pipeline._h, self._h, int(num_substeps)))

    def sync(self):
        _ffi.check(self.lib, self.lib.wgs_sync(self._h))

    def stats(self):
        s = self.T.Stats()
        _ffi.check(self.lib, self.lib.wgs_get_stats(self._h, C.byref(s)))
        return {k: int(getattr(s, k)) for k, _ in s._fields_}

    def num_particles(self) -> int:
        return self.stats()["num_particles"]

    def read_body_poses(self):
        """Like MpmData.read_body_poses: every rank integrates the same bodies, any rank can be asked."""
        n, D = self.n_colliders, self.dim
        poses, vels = (self.T.Pose * max(1, n))(), (self.T.Velocity * max(1, n))()
        coms = (C.c_float * (3 * max(1, n)))()
        _ffi.check(self.lib, self.lib.wgs_read_body_poses(self._h, poses, vels, coms, n))
        return [dict(rotation=np.array(list(poses[i].rotation)[:(2 if D == 2 else 4)], np.float64),
                     translation=np.array(list(poses[i].translation)[:D], np.float64),
                     linvel=np.array(list(vels[i].linear)[:D], np.float64),
                     angvel=np.array(list(vels[i].angular)[:(1 if D == 2 else 3)], np.float64)) for i in range(n)]

    def export(self):
        """(global ids, pos, vel, def_grad, affine, mass) of the particles this rank owns now (blocking)."""
        import torch
        buf = torch.zeros(self.hdr + self.capacity * self.part_rec, dtype=torch.float32, device=torch.device("cuda", self.pipeline.device))
        cnt = C.c_uint32(0)
        _ffi.check(self.lib, self.lib.wgs_shard_export(self._h, C.c_void_p(buf.data_ptr()), self.capacity, C.byref(cnt)))
        rec = buf[self.hdr: self.hdr + cnt.value * self.part_rec].cpu().numpy().reshape(cnt.value, self.part_rec)
        return unpack_records(rec, self.dim, self.uniform_material if self.dim == 3 else None)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.wgs_data_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def native_lockstep(pipeline: MpmPipeline, shards: List[NativeShard], num_substeps: int):
    """`wgs_sharded_step_lockstep`: all slabs of one process on one device, in x order."""
    arr = (C.c_void_p * len(shards))(*[s._h for s in shards])
    _ffi.check(pipeline.lib, pipeline.lib.wgs_sharded_step_lockstep(pipeline._h, arr, len(shards), int(num_substeps)))


def uniform_material_of(particles: ParticleSet):
    """(mass, init_volume, lambda, mu) if all particles of the set share them bitwise, else None."""
    if particles.n == 0:
        return None
    cols = (particles.mass, particles.init_volume, particles.lambda_, particles.mu)
    if all(bool(np.all(c.view(np.uint32) == c.view(np.uint32)[0])) for c in cols):
        return tuple(float(c[0]) for c in cols)
    return None


def unpack_records(rec: np.ndarray, dim: int, uniform_material=None):
    """Particle records (quad layout of csrc/layout.h) -> dict of arrays. In uniform-material mode (3D) the record's
    XM.w slot holds F[8] and the mass is the shared one."""
    ids = rec[:, -2].copy().view(np.uint32)     # [..quads.., pid, cdf epoch]
    if dim == 3:
        q = lambda k: rec[:, 4 * k:4 * k + 4]
        pos, mass = q(0)[:, :3], q(0)[:, 3]
        C_ = np.concatenate([q(1), q(2), q(3)[:, :1]], 1)
        vel = q(3)[:, 1:4]
        F = np.concatenate([q(4), q(5), q(6)[:, :1]], 1)
        if uniform_material is not None:
            F = np.concatenate([q(4), q(5), q(0)[:, 3:4]], 1)
            mass = np.full(len(rec), np.float32(uniform_material[0]), np.float32)
    else:
        q = lambda k: rec[:, 4 * k:4 * k + 4]
        pos, mass = q(0)[:, :2], q(0)[:, 2]
        C_, vel, F = q(1), q(2)[:, :2], q(3)
    return dict(ids=ids, pos=pos.copy(), vel=vel.copy(), def_grad=F.copy(), affine=C_.copy(), mass=mass.copy())


def split_scene(particles: ParticleSet, partition: SlabPartition, cell_width: float):
    """Assign every particle to the rank owning its associated block; returns per-rank (ParticleSet, global ids)."""
    bx = associated_block_x(particles.pos, cell_width, particles.dim)
    owner = partition.owner_of_blocks(bx)
    out = []
    for r in range(partition.world):
        idx = np.nonzero(owner == r)[0]
        sub = ParticleSet(**{k: (v[idx] if isinstance(v, np.ndarray) else v) for k, v in particles.__dict__.items()})
        out.append((sub, idx.astype(np.uint32)))
    return out
