"""Domain decomposition of the block grid over the GPUs of one node.

Replaces the reference's rank partitioning + Synchronizer (main.cpp:6494-6504, 1971-2142) for uniform
grids with a px x py Cartesian split: every rank owns an nbx x nby patch of 8x8 blocks, keeps one ghost
block per boundary block on each interior side, and exchanges only face strips (3 cell layers for the
WENO5 stencil, 1 for the 5-point kernels) with at most four peers.  The HIP library packs/unpacks the
strips and sweeps inner blocks while they are in flight; this module supplies the plan and the two
communication callbacks (RCCL send/recv and all-reduce through torch.distributed; "nccl" IS RCCL on
ROCm).  With the "gloo" backend the same code stages through host memory, which is how the path is
exercised on CPU and on a single-GPU box.
"""
import ctypes

import numpy as np

from . import lib as _l
from .grid import BlockGrid
from .simulation import Simulation

OPPOSITE = (1, 0, 3, 2)  # W<->E, S<->N


def cartesian_dims(world):
    """px x py with py >= px and both powers of two where possible: 8 -> 2 x 4 (BASELINE.json configs[3])."""
    px = 1
    while px * px * 4 <= world and world % (px * 2) == 0:
        px *= 2
    if world % px:
        px = 1
    return px, world // px


class PatchTopology:
    """Which strips rank (cx, cy) of a px x py decomposition sends to / receives from whom.

    send_block/send_face and recv_block/recv_face are the arrays of cup2d_halo_plan, grouped by peer in
    side order W, E, S, N and by position along the side; both ends of a link enumerate positions in the
    same order, so strip k of a message lands in ghost slot k of the opposite side.
    """

    def __init__(self, nbx, nby, px, py, cx, cy, order="hilbert"):
        self.px, self.py, self.cx, self.cy = px, py, cx, cy
        self.rank = cy * px + cx
        sides = (cx > 0, cx < px - 1, cy > 0, cy < py - 1)
        self.grid = BlockGrid(nbx, nby, order=order, ghost_sides=sides)
        g = self.grid
        peer_of_side = (self.rank - 1, self.rank + 1, self.rank - px, self.rank + px)
        self.peers = []  # (peer_rank, send_offset, recv_offset, nstrips), offsets in strips
        sb, sf, rb, rf = [], [], [], []
        for side in range(4):
            if not sides[side]:
                continue
            npos = nby if side < 2 else nbx
            soff, roff = len(sb), len(rb)
            for pos in range(npos):
                if side == 0:
                    owned = g.index_of[pos, 0]
                elif side == 1:
                    owned = g.index_of[pos, nbx - 1]
                elif side == 2:
                    owned = g.index_of[0, pos]
                else:
                    owned = g.index_of[nby - 1, pos]
                sb.append(int(owned))
                sf.append(side)                          # my blocks' face on that side
                rb.append(int(g._ghost_id[(side, pos)]))
                rf.append(OPPOSITE[side])                # the peer's face that touches me
            self.peers.append((peer_of_side[side], soff, roff, npos))
        self.send_block = np.asarray(sb, dtype=np.int32)
        self.send_face = np.asarray(sf, dtype=np.int32)
        self.recv_block = np.asarray(rb, dtype=np.int32)
        self.recv_face = np.asarray(rf, dtype=np.int32)
        self.nsend, self.nrecv = len(sb), len(rb)


def strip_cells(face, width):
    """cell indices (iy*8+ix) of a strip in buffer order -- the layout of halo.hip's strip_cell()"""
    if face < 2:
        return [iy * 8 + (k if face == 0 else 8 - width + k) for iy in range(8) for k in range(width)]
    return [(j if face == 2 else 8 - width + j) * 8 + ix for j in range(width) for ix in range(8)]


class TorchComm:
    """Face-strip exchange and all-reduce over torch.distributed.

    mode "device": tensors are GPU tensors and the backend moves them directly (nccl/RCCL over xGMI);
                   the exchange runs on a dedicated communication stream and overlaps the inner-block
                   sweep the library launches between exchange() and wait().
    mode "staged": GPU tensors staged through host memory (gloo) -- functional path for boxes where
                   RCCL cannot be used (e.g. two ranks on one GPU).
    mode "host"  : tensors already live on the host (CPU tests of the plan).
    """

    MAX_STRIP = 192  # CUP2D_MAX_STRIP_DOUBLES: WENO halo 3 layers x 8 cells x 2 components = 48; Krylov ghost blocks: three whole scalar blocks

    def __init__(self, topo, mode, device=None, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.topo, self.mode = torch, dist, topo, mode
        self.group = group  # process group of the data path (None = the default group)
        dev = torch.device("cpu") if mode == "host" else torch.device("cuda", device)
        self.send = torch.zeros(max(1, topo.nsend) * self.MAX_STRIP, dtype=torch.float64, device=dev)
        self.recv = torch.zeros(max(1, topo.nrecv) * self.MAX_STRIP, dtype=torch.float64, device=dev)
        self.red = torch.zeros(8, dtype=torch.float64, device=dev)
        self._ops = {}
        self._red_views = {}
        if mode == "staged":
            self.h_send = torch.zeros_like(self.send, device="cpu").pin_memory()
            self.h_recv = torch.zeros_like(self.recv, device="cpu").pin_memory()
            self.h_red = torch.zeros(8, dtype=torch.float64).pin_memory()
        if mode == "device":
            self.compute_stream = torch.cuda.Stream(device=dev)
            self.comm_stream = torch.cuda.Stream(device=dev)
            self.ev_packed = torch.cuda.Event()
            self.ev_arrived = torch.cuda.Event()
        elif mode == "staged":
            self.compute_stream = torch.cuda.Stream(device=dev)
        else:
            self.compute_stream = None

    # ---- point-to-point -----------------------------------------------------------------------
    def _p2p(self, send, recv, strip_doubles, topo=None):
        """post all receives, then all sends, as one batch (one ncclGroup on RCCL).  The P2POp lists are built once
        per strip width: an exchange happens twice per BiCGSTAB iteration, its host cost is on the critical path of
        keeping the GPU fed."""
        dist = self.dist
        topo = topo or self.topo  # (a cell plan of an adapted grid brings its own offsets and counts: amr_dist.CellTopo)
        key = (send.data_ptr(), recv.data_ptr(), strip_doubles, id(topo))
        ops = self._ops.get(key)
        if ops is None:
            ops = []
            for pr in topo.peers:  # (peer, send offset, receive offset, strips out[, strips in])
                peer, roff, n = pr[0], pr[2], pr[4] if len(pr) > 4 else pr[3]
                if n:
                    ops.append(dist.P2POp(dist.irecv, recv[roff * strip_doubles:(roff + n) * strip_doubles], peer, self.group))
            for pr in topo.peers:
                peer, soff, n = pr[0], pr[1], pr[3]
                if n:
                    ops.append(dist.P2POp(dist.isend, send[soff * strip_doubles:(soff + n) * strip_doubles], peer, self.group))
            self._ops[key] = ops
        return dist.batch_isend_irecv(ops) if ops else []

    def exchange(self, strip_doubles, topo=None):
        """start moving the packed strips (called after the pack kernel was enqueued); topo: another unit and other per-peer
        counts than the block plan's (the cell plans of an adapted grid)"""
        torch = self.torch
        topo = topo or self.topo
        if self.mode == "device":
            self.ev_packed.record(self.compute_stream)
            self.comm_stream.wait_event(self.ev_packed)
            with torch.cuda.stream(self.comm_stream):
                for r in self._p2p(self.send, self.recv, strip_doubles, topo):
                    r.wait()
                self.ev_arrived.record(self.comm_stream)
        elif self.mode == "staged":
            ns, nr = topo.nsend * strip_doubles, topo.nrecv * strip_doubles
            with torch.cuda.stream(self.compute_stream):
                self.h_send[:ns].copy_(self.send[:ns], non_blocking=True)
            self.compute_stream.synchronize()
            for r in self._p2p(self.h_send, self.h_recv, strip_doubles, topo):
                r.wait()
            with torch.cuda.stream(self.compute_stream):
                self.recv[:nr].copy_(self.h_recv[:nr], non_blocking=True)
        else:
            for r in self._p2p(self.send, self.recv, strip_doubles, topo):
                r.wait()

    def wait(self):
        if self.mode == "device":
            self.compute_stream.wait_event(self.ev_arrived)

    # ---- reductions ------------------------------------------------------------------------------
    def allreduce(self, offset, count, op):
        torch, dist = self.torch, self.dist
        rop = dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX
        if self.mode == "device":
            view = self._red_views.get((offset, count))
            if view is None:
                view = self._red_views[(offset, count)] = self.red[offset:offset + count]
            with torch.cuda.stream(self.compute_stream):
                dist.all_reduce(view, op=rop, group=self.group)
        elif self.mode == "staged":
            with torch.cuda.stream(self.compute_stream):
                self.h_red[offset:offset + count].copy_(self.red[offset:offset + count], non_blocking=True)
            self.compute_stream.synchronize()
            dist.all_reduce(self.h_red[offset:offset + count], op=rop, group=self.group)
            with torch.cuda.stream(self.compute_stream):
                self.red[offset:offset + count].copy_(self.h_red[offset:offset + count], non_blocking=True)
        else:
            dist.all_reduce(self.red[offset:offset + count], op=rop, group=self.group)


class DistributedSimulation(Simulation):
    """One rank's patch of a px x py decomposition; same interface as Simulation.  Fields set/get
    through .vel/.pres/... are this rank's (nby*8, nbx*8) patch.

    comm "rccl"  (default on the "nccl" backend): the communicator inside the library (cup2d_comm_init, csrc/comm.hip) --
                 RCCL send/recv on the library's own communication stream, all-reduce / all-gather on its compute
                 stream; torch.distributed only carries the rendezvous token to the ranks.  No Python between a
                 cup2d_* call and its return.
    comm "torch": the callback interface (cup2d_set_comm) served by TorchComm -- the transport of the gloo tests
                 (mode "staged" / "host") and a second RCCL path (mode "device")."""

    def __init__(self, nbx, nby, px, py, extent=1.0, nu=1e-3, cfl=0.5, order="hilbert", device=0, mode=None, comm=None,
                 group=None):
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        assert world == px * py, "world size %d != %d x %d" % (world, px, py)
        self.topo = PatchTopology(nbx, nby, px, py, rank % px, rank // px, order=order)
        if comm is None:
            comm = "rccl" if (mode is None and dist.get_backend() == "nccl") else "torch"
        if comm not in ("rccl", "torch"):
            raise ValueError("comm must be 'rccl' or 'torch'")
        if mode is None:
            mode = "device" if dist.get_backend() == "nccl" else "staged"
        h = float(extent) / max(nbx * px, nby * py) / 8  # main.cpp:6338 on the GLOBAL grid
        super().__init__(nbx, nby, nu=nu, cfl=cfl, device=device, grid=self.topo.grid, h=h)
        t = self.topo
        vp = ctypes.c_void_p
        _l.check(self.L.cup2d_halo_plan(self._ctx, t.nsend, t.send_block.ctypes.data_as(vp), t.send_face.ctypes.data_as(vp),
                                        t.nrecv, t.recv_block.ctypes.data_as(vp), t.recv_face.ctypes.data_as(vp)), "halo_plan")
        self.comm_kind = comm
        self.comm_errors = []
        if comm == "rccl":
            self.comm = None
            token = ctypes.create_string_buffer(_l.COMM_ID_BYTES)
            if rank == 0:
                _l.check(self.L.cup2d_comm_unique_id(token), "comm_unique_id")
            box = [token.raw]
            dist.broadcast_object_list(box, src=0)  # the only thing torch.distributed moves for this simulation (default group)
            peers = np.asarray([tuple(p)[:4] for p in t.peers], dtype=np.int32).reshape(-1, 4)
            cols = [np.ascontiguousarray(peers[:, k]) for k in range(4)]
            _l.check(self.L.cup2d_comm_init(self._ctx, world, rank, box[0], len(t.peers), *[c.ctypes.data_as(vp) for c in cols], None),
                     "comm_init")
            # one checked round of everything the time loop will ask of the communicator (strips between all peers, an
            # all-gather, an all-reduce), bounded in time: a broken link or a missing peer ends HERE with a message
            self.comm_report = self.comm_selftest()
            return
        self.comm = TorchComm(self.topo, mode, device, group=group)
        self.set_stream(self.comm.compute_stream.cuda_stream)
        red_base = self.comm.red.data_ptr()
        comm = self.comm

        def _exchange(user, send, recv, strip_doubles, stream):
            try:
                comm.exchange(strip_doubles)
                return 0
            except Exception as e:  # noqa: BLE001 -- must not propagate into C
                self.comm_errors.append(repr(e))
                return -1

        def _wait(user, stream):
            try:
                comm.wait()
                return 0
            except Exception as e:  # noqa: BLE001
                self.comm_errors.append(repr(e))
                return -1

        def _allreduce(user, buf, count, op, stream):
            try:
                comm.allreduce((buf - red_base) // 8, count, op)
                return 0
            except Exception as e:  # noqa: BLE001
                self.comm_errors.append(repr(e))
                return -1

        self._cb = (_l.EXCHANGE_FN(_exchange), _l.WAIT_FN(_wait), _l.ALLREDUCE_FN(_allreduce))  # keep alive
        _l.check(self.L.cup2d_set_comm(self._ctx, self._cb[0], self._cb[1], self._cb[2], None,
                                       vp(comm.send.data_ptr()), vp(comm.recv.data_ptr()), vp(red_base)), "set_comm")
        _l.check(self.L.cup2d_set_comm_strip_capacity(self._ctx, TorchComm.MAX_STRIP), "set_comm_strip_capacity")

    def comm_selftest(self, timeout_s=20.0):
        """cup2d_comm_selftest: collective; returns the report string parsed into a dict"""
        buf = ctypes.create_string_buffer(1024)
        _l.check(self.L.cup2d_comm_selftest(self._ctx, float(timeout_s), buf, len(buf)), "comm_selftest")
        rep = {}
        for tok in buf.value.decode().split():
            k, _, v = tok.partition("=")
            rep[k] = v
        return rep

    def comm_stats(self):
        """ranks, peers of this rank and the collectives issued so far by the in-library communicator"""
        i, ll = ctypes.c_int, ctypes.c_longlong
        nr, np_, ex, ar, ag = i(), i(), ll(), ll(), ll()
        _l.check(self.L.cup2d_comm_stats(self._ctx, ctypes.byref(nr), ctypes.byref(np_), ctypes.byref(ex), ctypes.byref(ar),
                                         ctypes.byref(ag)), "comm_stats")
        return dict(nranks=nr.value, peers=np_.value, exchanges=ex.value, allreduces=ar.value, allgathers=ag.value)

    def halo_exchange(self, field, width):
        """sync1 of main.cpp:1971-2142 for one field: pack, exchange, unpack the ghost strips (cup2d_halo_exchange)"""
        _l.check(self.L.cup2d_halo_exchange(self._ctx, int(field), int(width)), "halo_exchange")


def self_periodic_simulation(nbx, nby, nu=1e-3, cfl=0.5, device=0, axes="x"):
    """One rank that is its own neighbour through the in-library communicator: ghost blocks on both x sides (axes "x": a domain
    periodic in x, walls in y; axes "y": the same in y) or on all four sides (axes "xy": doubly periodic -- what an INTERIOR rank of a decomposition has,
    and every rank of BASELINE.json configs[3]'s 2 x 4 layout has two or three of: halo-set patches that meet in corners, four
    peers in one ncclGroup, four consecutive ghost ranges received in place), each filled from the opposite edge by ncclSend /
    ncclRecv to self.  Everything a rank of an N-rank run does -- the halo set ordered last, pack kernels, whole ghost blocks of
    the Krylov vectors through RCCL, the MERGE 2 kernels, an all-gather and a one-wave kernel per reduction -- on ONE GPU: the
    tests of the communicator's data path (tests/test_comm.py) and the N-rank-path leg of bench.py use it.  Returns (Simulation,
    BlockGrid); the caller finalises the communicator (cup2d_comm_finalize) before closing."""
    if axes not in ("x", "y", "xy"):
        raise ValueError("axes must be 'x', 'y' or 'xy'")
    sides = {"x": (0, 1), "y": (2, 3), "xy": (0, 1, 2, 3)}[axes]
    g = BlockGrid(nbx, nby, ghost_sides=tuple(k in sides for k in range(4)))
    s = Simulation(nbx, nby, nu=nu, cfl=cfl, device=device, grid=g, h=1.0 / (8 * max(nbx, nby)))
    sb, sf, rb, rf = [], [], [], []
    for side in sides:  # W, E, S, N strips in the send list; W, E, S, N ghosts in the receive list
        for pos in range(nby if side < 2 else nbx):
            if side < 2:
                sb.append(int(g.index_of[pos, 0 if side == 0 else nbx - 1]))
            else:
                sb.append(int(g.index_of[0 if side == 2 else nby - 1, pos]))
            sf.append(side)
            rb.append(int(g._ghost_id[(side, pos)]))
            rf.append(OPPOSITE[side])
    arr = [np.asarray(a, dtype=np.int32) for a in (sb, sf, rb, rf)]
    vp = ctypes.c_void_p
    _l.check(s.L.cup2d_halo_plan(s.ctx, len(sb), arr[0].ctypes.data_as(vp), arr[1].ctypes.data_as(vp), len(rb),
                                 arr[2].ctypes.data_as(vp), arr[3].ctypes.data_as(vp)), "halo_plan")
    ids = ctypes.create_string_buffer(_l.COMM_ID_BYTES)
    _l.check(s.L.cup2d_comm_unique_id(ids), "comm_unique_id")
    # receive i pairs with send i (RCCL matches the operations of a pair of ranks in issue order): the W ghosts
    # (receive offset 0) take the E strips (send offset nby), the E ghosts the W strips; likewise S <- N, N <- S
    if axes == "x":
        soff, roff, cnt = [nby, 0], [0, nby], [nby, nby]
    elif axes == "y":  # (the S and N sides only: what the ranks of a 2 x 4 layout have towards their long sides)
        soff, roff, cnt = [nbx, 0], [0, nbx], [nbx, nbx]
    else:
        soff, roff, cnt = [nby, 0, 2 * nby + nbx, 2 * nby], [0, nby, 2 * nby, 2 * nby + nbx], [nby, nby, nbx, nbx]
    peer = np.zeros(len(cnt), dtype=np.int32)
    soff, roff, cnt = (np.asarray(a, dtype=np.int32) for a in (soff, roff, cnt))
    _l.check(s.L.cup2d_comm_init(s.ctx, 1, 0, ids, len(cnt), peer.ctypes.data_as(vp), soff.ctypes.data_as(vp), roff.ctypes.data_as(vp),
                                 cnt.ctypes.data_as(vp), None), "comm_init")
    return s, g
