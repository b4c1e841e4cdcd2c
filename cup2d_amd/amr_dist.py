"""Block-AMR grids on N ranks (BASELINE.json configs[4]: one process per GPU).

The reference gives every rank a contiguous range of the Hilbert-ordered leaf list (main.cpp:6494-6504), re-balances the
ranges after every regrid by migrating blocks (main.cpp:5055-5424) and reaches remote blocks through its synchroniser
(neighbour discovery over Hilbert ranges, fine->coarse averaging in pack main.cpp:2053-2125, remote unpack into the
coarse-fine labs 2582-2684, flux faces 1819-1825).  Here the same partition is planned on dense tables:

  AmrPartition   for one rank of `nranks`: its owned range, its GHOST blocks (every remote block its kernels read: face
                 neighbours of owned blocks, their tangential neighbours across a coarser block -- two rings -- and the
                 blocks the coarse-fine Poisson rows reference), the topology tables of owned + ghost blocks in local
                 numbering, who sends which whole blocks to whom (the halo plan of cup2d_halo_plan), and the rank's rows
                 of the Poisson matrix with ghost cells as halo columns.  Every rank derives every rank's plan from the
                 global leaf list (no negotiation round: the lists are deterministic), as the reference's Setup() does
                 from its tree.
  DistributedAmrSimulation   the device context of one rank on that plan; same block operators as AmrSimulation on
                 the owned blocks.  Regridding gathers the fields of all ranks (regrid-time work, every AdaptSteps
                 steps), regrids the global list on every rank with the library's host routines and re-partitions:
                 the new ranges ARE the load balance, a block that changes range has migrated.

The device side needs nothing AMR-specific for the exchange: ghost blocks travel whole through the face-strip kernels
(a strip of width 8 is the block), the kernels of csrc/amr.hip read them through the same tables as owned blocks.
"""
import ctypes

import numpy as np

from . import lib as _l
from .amr import AmrBlockGrid, AmrSimulation, BS, regrid, tag_states, validate_states, LEAVE


def partition_bounds(nblocks, nranks):
    """contiguous, equally filled ranges of the Hilbert-ordered leaf list (the reference's balance criterion: blocks per rank)"""
    return np.array([(nblocks * r) // nranks for r in range(nranks + 1)], dtype=np.int64)


class AmrPartition:
    def __init__(self, G, nranks, rank, coo=None):
        """G: the GLOBAL AmrBlockGrid (every rank holds the leaf list, 12 bytes per block).  coo: the global Poisson triplets
        (G.poisson_coo()) when the caller wants this rank's rows as triplets (row, col, val: the route through
        cup2d_set_matrix_coo, and the tests); without them the library assembles the rows from the local tables
        (cup2d_amr_install_poisson) -- a row's columns lie in the block itself and its face neighbours, all in the first
        ghost ring."""
        self.G, self.nranks, self.rank = G, int(nranks), int(rank)
        nb = G.nblocks
        self.bounds = partition_bounds(nb, nranks)
        self.owner = np.searchsorted(self.bounds, np.arange(nb), side="right") - 1
        self.coo = coo
        self._ghosts = [self._ghost_ids(r) for r in range(nranks)]  # every rank's ghost list (global ids, ascending)
        lo, hi = self.bounds[rank], self.bounds[rank + 1]
        self.lo, self.hi = int(lo), int(hi)
        self.nowned = int(hi - lo)
        ghosts = self._ghosts[rank]
        self.ghost_ids = ghosts
        self.nghost = len(ghosts)
        self.local_ids = np.concatenate([np.arange(lo, hi), ghosts]).astype(np.int64)  # local index -> global id
        self.local_of = -np.ones(nb, dtype=np.int64)
        self.local_of[self.local_ids] = np.arange(len(self.local_ids))
        # ---- tables in local numbering: a side whose neighbour this rank does not hold becomes a wall (never read) ----
        ids = self.local_ids
        kind = G.kind[ids].copy()
        nbr2 = G.nbr2[ids].astype(np.int64)
        have = np.where(nbr2 >= 0, self.local_of[np.maximum(nbr2, 0)], -1)
        need2 = kind == _l.AMR_FINER
        missing = (kind != _l.AMR_WALL) & ((have[:, :, 0] < 0) | (need2 & (have[:, :, 1] < 0)))
        assert not missing[:self.nowned].any(), "ghost closure does not cover the owned blocks' neighbours"
        kind[missing] = _l.AMR_WALL
        have[missing] = -1
        have[kind != _l.AMR_FINER, 1] = -1
        self.level = np.ascontiguousarray(G.level[ids], dtype=np.int32)
        self.kind = np.ascontiguousarray(kind, dtype=np.int32)
        self.nbr2 = np.ascontiguousarray(have, dtype=np.int32)
        self.half = np.ascontiguousarray(np.where(missing, 0, G.half[ids]), dtype=np.int32)
        self.nbr = np.ascontiguousarray(np.where(self.kind[:self.nowned] == _l.AMR_SAME, self.nbr2[:self.nowned, :, 0], -1), dtype=np.int32)
        # ---- who sends which blocks to whom: peers ascending, blocks ascending on both ends ----
        self.peers = []  # (peer, send offset, receive offset, blocks out, blocks in)
        send, recv = [], []
        for p in range(nranks):
            if p == rank:
                continue
            mine = self._ghosts[p]
            out = mine[self.owner[mine] == rank] if len(mine) else mine        # p's ghosts that I own, in p's order
            inn = ghosts[self.owner[ghosts] == p] if len(ghosts) else ghosts   # my ghosts that p owns
            if len(out) or len(inn):
                self.peers.append((p, len(send), len(recv), len(out), len(inn)))
                send.extend(self.local_of[out].tolist())
                recv.extend(self.local_of[inn].tolist())
        self.send_block = np.asarray(send, dtype=np.int32)
        self.recv_block = np.asarray(recv, dtype=np.int32)
        assert np.array_equal(self.recv_block, self.nowned + np.arange(self.nghost)), "ghosts are numbered in receive order"
        self.nsend, self.nrecv = len(send), len(recv)
        # ---- this rank's rows of the Poisson matrix; ghost cells are halo columns 64 * nowned + 64 * g + cell ----
        self.row = self.col = self.val = None
        if coo is not None:
            r, c, v = coo
            m = (r >= 64 * lo) & (r < 64 * hi)
            cb = self.local_of[c[m] // 64]
            assert (cb >= 0).all(), "ghost closure does not cover the matrix columns"
            self.row = (r[m] - 64 * lo).astype(np.int32)
            self.col = (cb * 64 + c[m] % 64).astype(np.int32)
            self.val = np.ascontiguousarray(v[m])
        self.gather = (self.send_block.astype(np.int64)[:, None] * 64 + np.arange(64)[None, :]).ravel().astype(np.int32)

    def _neighbours(self, ids):
        G = self.G
        k = G.kind[ids]
        n = G.nbr2[ids].astype(np.int64)
        a = n[:, :, 0][k != _l.AMR_WALL]
        b = n[:, :, 1][k == _l.AMR_FINER]
        return np.unique(np.concatenate([a, b]))

    def _ghost_ids(self, r):
        lo, hi = self.bounds[r], self.bounds[r + 1]
        owned = np.arange(lo, hi)
        ring1 = self._neighbours(owned)
        ring2 = self._neighbours(np.union1d(owned, ring1)) if len(ring1) else ring1
        g = np.union1d(ring1, ring2)
        if self.coo is not None:  # (a row's columns: the block and its face neighbours -- nothing beyond ring 1)
            rr, cc, _ = self.coo
            m = (rr >= 64 * lo) & (rr < 64 * hi)
            cols = np.unique(cc[m] // 64)
            assert np.isin(cols, np.union1d(owned, ring1)).all(), "a matrix column outside the first ghost ring"
        return g[(g < lo) | (g >= hi)].astype(np.int64)


class _OwnedGrid:
    """what AmrSimulation's field accessors need of a grid: the owned blocks"""

    def __init__(self, part):
        G = part.G
        self.nblocks = part.nowned
        self.blocks = G.blocks[part.lo:part.hi]
        self.level = G.level[part.lo:part.hi]
        self.kind = part.kind[:part.nowned]
        self.h0, self.bpdx, self.bpdy = G.h0, G.bpdx, G.bpdy


class DistributedAmrSimulation(AmrSimulation):
    """One rank of an adapted grid.  Fields are set and read as per-block arrays of the OWNED blocks, in global leaf
    order (rank r holds leaves [bounds[r], bounds[r+1])).  comm: "torch" (callbacks over torch.distributed; mode
    "staged" = host-staged gloo, the transport of the tests; "device" = nccl) or "rccl" (the communicator inside the
    library)."""

    def __init__(self, global_grid, nu=1e-3, cfl=0.5, device=0, comm="torch", mode=None, group=None, adapt_steps=20):
        import torch.distributed as dist
        from .distributed import TorchComm
        self.L = _l.load_library()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.global_grid = global_grid
        self.part = P = AmrPartition(global_grid, self.world, self.rank)
        self.grid = _OwnedGrid(P)
        self.nu, self.cfl, self.device, self.adapt_steps = float(nu), float(cfl), int(device), int(adapt_steps)
        self.step_count = getattr(self, "step_count", 0)
        self._strict = getattr(self, "_strict", None)
        self._solver = getattr(self, "_solver", None)
        self._timing = getattr(self, "_timing", None)
        self._comm_kind, self._comm_mode, self._group = comm, mode, group
        self._ctx = ctypes.c_void_p()
        vp = ctypes.c_void_p
        _l.check(self.L.cup2d_create(ctypes.byref(self._ctx), P.nowned, P.nghost, P.nowned, P.nbr.ctypes.data_as(vp), global_grid.h0,
                                     int(device)), "cup2d_create")
        zs, zr = np.zeros(max(1, P.nsend), dtype=np.int32), np.zeros(max(1, P.nrecv), dtype=np.int32)
        _l.check(self.L.cup2d_halo_plan(self._ctx, P.nsend, P.send_block.ctypes.data_as(vp), zs.ctypes.data_as(vp), P.nrecv,
                                        P.recv_block.ctypes.data_as(vp), zr.ctypes.data_as(vp)), "halo_plan")
        self.comm_errors = []
        if comm == "rccl":
            token = ctypes.create_string_buffer(_l.COMM_ID_BYTES)
            if self.rank == 0:
                _l.check(self.L.cup2d_comm_unique_id(token), "comm_unique_id")
            box = [token.raw]
            dist.broadcast_object_list(box, src=0)
            cols = [np.ascontiguousarray([p[k] for p in P.peers], dtype=np.int32) for k in range(5)]
            _l.check(self.L.cup2d_comm_init(self._ctx, self.world, self.rank, box[0], len(P.peers), *[c.ctypes.data_as(vp) for c in cols]),
                     "comm_init")
            self.comm = None
        else:
            if mode is None:
                mode = "device" if dist.get_backend() == "nccl" else "staged"
            self.comm = cm = TorchComm(P, mode, device, group=group)
            self.L.cup2d_set_stream(self._ctx, ctypes.c_void_p(cm.compute_stream.cuda_stream))
            red_base = cm.red.data_ptr()

            def guard(fn):
                def call(*a):
                    try:
                        fn(*a)
                        return 0
                    except Exception as e:  # noqa: BLE001 -- must not propagate into C
                        self.comm_errors.append(repr(e))
                        return -1
                return call
            # the assembled operator receives straight into the Krylov vector (device_recv = &vec[64 * nblocks]); TorchComm moves
            # its own buffers, so the arrived blocks are copied on behind the wait
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
            pending = []

            def exchange(user, snd, rcv, sd, st):
                cm.exchange(sd)
                if rcv and rcv != cm.recv.data_ptr():
                    pending.append((rcv, P.nrecv * sd * 8))

            def wait(user, st):
                cm.wait()
                while pending:
                    dst, nbytes = pending.pop()
                    if hip.hipMemcpyAsync(dst, cm.recv.data_ptr(), nbytes, 3, st) != 0:
                        raise RuntimeError("hipMemcpyAsync of the received ghost blocks failed")
            self._cb = (_l.EXCHANGE_FN(guard(exchange)), _l.WAIT_FN(guard(wait)),
                        _l.ALLREDUCE_FN(guard(lambda user, buf, n, op, st: cm.allreduce((buf - red_base) // 8, n, op))))
            _l.check(self.L.cup2d_set_comm(self._ctx, self._cb[0], self._cb[1], self._cb[2], None, vp(cm.send.data_ptr()),
                                           vp(cm.recv.data_ptr()), vp(red_base)), "set_comm")
        self._tables = [P.level, P.kind, P.nbr2, P.half]
        _l.check(self.L.cup2d_set_amr(self._ctx, global_grid.h0, *[a.ctypes.data_as(vp) for a in self._tables]), "cup2d_set_amr")
        _l.check(self.L.cup2d_amr_set_finest_level(self._ctx, int(global_grid.level.max())), "amr_set_finest_level")
        if self._strict is not None:
            self.set_math(self._strict)
        if self._solver is not None:
            self.set_solver(*self._solver)
        if self._timing is not None:
            self.set_timing(self._timing)

    def install_poisson_matrix(self):
        """this rank's rows of the operator of main.cpp:7034-7113, ghost cells as halo columns, whole sent blocks as the
        gather list (cuda.h's send_pack_idx_ with a block as the unit)"""
        P, vp = self.part, ctypes.c_void_p
        if P.val is not None:
            _l.check(self.L.cup2d_set_matrix_coo(self._ctx, 64 * P.nghost, len(P.val), P.row.ctypes.data_as(vp),
                                                 P.col.ctypes.data_as(vp), P.val.ctypes.data_as(vp)), "set_matrix_coo")
        else:  # assembled by the library from the local tables: rows only where a side is coarse-fine or a ghost block
            _l.check(self.L.cup2d_amr_install_poisson(self._ctx), "amr_install_poisson")
        _l.check(self.L.cup2d_set_gather(self._ctx, len(P.gather), P.gather.ctypes.data_as(vp)), "set_gather")

    # ---- regridding across the ranks -------------------------------------------------------------------------------------
    def _allgather_blocks(self, a):
        """per-block array of the owned blocks -> the same array for all leaves, on every rank"""
        import torch.distributed as dist
        parts = [None] * self.world
        dist.all_gather_object(parts, np.ascontiguousarray(a))
        return np.concatenate(parts, axis=0)

    def adapt(self, rtol, ctol, level_max):
        """adapt() of main.cpp:4657-5440 on N ranks: tags from this rank's blocks (vorticity on the GPU), gathered; the
        validated states, prolongation / restriction and the new leaf list are computed on every rank from the gathered
        fields with the library's host routines (regrid-time work); the new contiguous ranges re-balance the load --
        blocks that change range have migrated (main.cpp:5055-5424).  Returns True if the grid changed."""
        self.vorticity()
        linf = np.empty(self.part.nowned)
        _l.check(self.L.cup2d_block_linf(self._ctx, _l.TMP, linf.ctypes.data_as(ctypes.c_void_p)), "block_linf")
        G = self.global_grid
        linf = self._allgather_blocks(linf)
        st = validate_states(G.blocks, tag_states(linf, G.level, rtol, ctol, level_max), level_max, G.bpdx, G.bpdy)
        if not (st != LEAVE).any():
            return False
        names = {"chi": _l.CHI, "vel": _l.VEL, "vold": _l.VOLD, "pres": _l.PRES, "pold": _l.POLD}
        fields = {}
        for k, f in names.items():
            a = self._allgather_blocks(self.get_field(f).reshape(self.part.nowned, -1))
            fields[k] = (a, _l.FIELD_DIM[f], _l.FIELD_DIM[f] == 2)
        blocks, data = regrid(G.blocks, st, fields, level_max, G.bpdx, G.bpdy)
        new_grid = AmrBlockGrid(blocks, G.bpdx, G.bpdy, G.h0 * max(G.bpdx, G.bpdy) * BS)
        self.close()
        self.__init__(new_grid, nu=self.nu, cfl=self.cfl, device=self.device, comm=self._comm_kind, mode=self._comm_mode,
                      group=self._group, adapt_steps=self.adapt_steps)
        lo, hi = self.part.lo, self.part.hi
        for k, f in names.items():
            self.set_field(f, data[k][lo:hi])
        self.install_poisson_matrix()
        return True
