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
                 the owned blocks.  Regridding (every AdaptSteps steps): tags and the leaf list are replicated (8 + 12 bytes
                 per block), the FIELDS are not -- the new list is cut into contiguous ranges (the load balance: a block
                 whose range changes has migrated) and every rank fetches what its new range is made of from the old
                 owners and computes its own prolonged / restricted blocks (fetch_new_range, cup2d_amr_regrid_local).

The device side needs nothing AMR-specific for the exchange: a ghost block is a copy of the remote block, the kernels of
csrc/amr.hip read it through the same tables as owned blocks.  What TRAVELS is not the whole block but the cells of it the
receiving rank's kernels read (AmrPartition.cells: one cell plan per operator family, cup2d_halo_plan_cells) -- the strips of the
reference's synchroniser, found by running the kernels' own ghost expressions with a recording accessor
(cup2d_amr_trace_reads); CUP2D_AMR_STRIPS=0 sends whole blocks (the block plan alone).
"""
import ctypes

import numpy as np

from . import lib as _l
from .amr import AmrBlockGrid, AmrSimulation, BS, regrid, tag_states, validate_states, LEAVE


class CellTopo:
    """per-peer offsets and counts of one cell plan, in the shape TorchComm reads (peer, send offset, receive offset, cells out,
    cells in) -- the message unit is one cell"""

    def __init__(self, peers, nsend, nrecv):
        self.peers, self.nsend, self.nrecv = peers, int(nsend), int(nrecv)


def strips_enabled():
    import os
    return os.environ.get("CUP2D_AMR_STRIPS", "1") != "0"


def _mask_cells(mask_rows):
    """uint64 masks [n] -> (row index, cell) of every set bit, rows ascending, cells ascending"""
    if len(mask_rows) == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    bits = np.unpackbits(np.ascontiguousarray(mask_rows, dtype="<u8").view(np.uint8).reshape(-1, 8), axis=1, bitorder="little")
    return np.nonzero(bits)


def partition_bounds(nblocks, nranks):
    """contiguous, equally filled ranges of the Hilbert-ordered leaf list (the reference's balance criterion: blocks per rank)"""
    return np.array([(nblocks * r) // nranks for r in range(nranks + 1)], dtype=np.int64)


class AmrPartition:
    def __init__(self, G, nranks, rank, coo=None, strips=None):
        """G: the GLOBAL AmrBlockGrid (every rank holds the leaf list, 12 bytes per block).  coo: the global Poisson triplets
        (G.poisson_coo()) when the caller wants this rank's rows as triplets (row, col, val: the route through
        cup2d_set_matrix_coo, and the tests); without them the library assembles the rows from the local tables
        (cup2d_amr_install_poisson) -- a row's columns lie in the block itself and its face neighbours, all in the first
        ghost ring."""
        self.G, self.nranks, self.rank = G, int(nranks), int(rank)
        strips = strips_enabled() if strips is None else bool(strips)
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
        # ---- cell plans: of the blocks above, the cells the peer's kernels read (one list per operator family) ----
        self.cells = self._cell_plans() if (strips and self.peers) else None
        if self.cells is not None:
            self.gather = self.cells[_l.CELLS_MATRIX][0]

    def _cell_plans(self):
        """[set] -> (send_cell int32 [ns], recv_cell int32 [nr], CellTopo).  Link (reader rank a, owner rank b): the cells of b's
        blocks that a's blocks read, in (global block, cell) order on both ends.  The receiver traces its own blocks that some
        peer holds as ghosts (every block that reads a remote one is in that remote's two rings); the sender traces its ghost
        copies of the receiver's blocks -- the same readers as far as this link goes, so the same list, with no message."""
        G, L = self.G, _l.load_library()
        vp = ctypes.c_void_p
        kind = np.ascontiguousarray(G.kind, dtype=np.int32)
        nbr2 = np.ascontiguousarray(G.nbr2, dtype=np.int32)
        half = np.ascontiguousarray(G.half, dtype=np.int32)
        nb = G.nblocks

        def trace(readers, which):
            mask = np.zeros(nb, dtype=np.uint64)
            r = np.ascontiguousarray(readers, dtype=np.int32)
            _l.check(L.cup2d_amr_trace_reads(nb, kind.ctypes.data_as(vp), nbr2.ctypes.data_as(vp), half.ctypes.data_as(vp), len(r),
                                             r.ctypes.data_as(vp), which, mask.ctypes.data_as(vp)), "amr_trace_reads")
            return mask
        mine = [g[self.owner[g] == self.rank] for q, g in enumerate(self._ghosts) if q != self.rank and len(g)]
        readers_me = np.unique(np.concatenate(mine)) if mine else np.zeros(0, dtype=np.int64)
        plans = []
        for which in range(3):
            mask_me = trace(readers_me, which)
            mask_me[self.lo:self.hi] = 0  # (reads inside the own range)
            held = np.zeros(nb, dtype=bool)
            held[self.ghost_ids] = True
            assert held[mask_me != 0].all(), "a kernel reads a remote block outside the ghost list"
            send, recv, peers = [], [], []
            for (q, _so, _ro, _ns, _nr) in self.peers:
                g_in = self.ghost_ids[self.owner[self.ghost_ids] == q]          # my ghosts that q owns, ascending
                rows, cells = _mask_cells(mask_me[g_in])
                cin = self.local_of[g_in[rows]] * 64 + cells
                mask_q = trace(g_in, which)[self.lo:self.hi]                       # what q's blocks next to me read of mine
                rows, cells = _mask_cells(mask_q)
                cout = rows * 64 + cells
                listed = np.zeros(self.nowned, dtype=bool)
                listed[self.send_block[_so:_so + _ns]] = True
                assert listed[rows].all(), "a cell of a block the peer does not hold as a ghost"
                peers.append((q, len(send), len(recv), len(cout), len(cin)))
                send.extend(cout.tolist())
                recv.extend(cin.tolist())
            plans.append((np.asarray(send, dtype=np.int32), np.asarray(recv, dtype=np.int32), CellTopo(peers, len(send), len(recv))))
        return plans

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
        for which, (sc, rc, _t) in enumerate(P.cells or ()):
            _l.check(self.L.cup2d_halo_plan_cells(self._ctx, which, len(sc), sc.ctypes.data_as(vp), len(rc), rc.ctypes.data_as(vp)),
                     "halo_plan_cells")
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
            for which, (_sc, _rc, t) in enumerate(P.cells or ()):
                cc = [np.ascontiguousarray([x[k] for x in t.peers], dtype=np.int32) for k in (1, 3, 2, 4)]
                _l.check(self.L.cup2d_comm_set_cell_counts(self._ctx, which, len(t.peers), *[c.ctypes.data_as(vp) for c in cc]),
                         "comm_set_cell_counts")
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

            self.cell_exchanges = [0, 0, 0]  # per cell plan: exchanges issued (diagnostics, tests)
            self.block_exchanges = 0
            self.sent_doubles = [0, 0]         # through the cell plans: what went out, what the same exchanges move as whole blocks

            def exchange(user, snd, rcv, sd, st):
                cs = _l.cell_strip(sd)
                if cs is not None:  # a cell plan: the unit is one cell of cs[1] doubles, offsets and counts of that plan
                    self.cell_exchanges[cs[0]] += 1
                    self.sent_doubles[0] += P.cells[cs[0]][2].nsend * cs[1]
                    self.sent_doubles[1] += P.nsend * 64 * cs[1]
                    cm.exchange(cs[1], topo=P.cells[cs[0]][2])
                    return
                self.block_exchanges += 1
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
            _l.check(self.L.cup2d_set_comm_strip_capacity(self._ctx, cm.MAX_STRIP), "set_comm_strip_capacity")
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
        dist.all_gather_object(parts, np.ascontiguousarray(a), group=self._ctl_group())
        return np.concatenate(parts, axis=0)

    def _ctl_group(self):
        """host-side messages of a regrid (tags, block requests, migrating blocks): a gloo group -- the default one, or a
        second group next to an nccl default (created once, collectively: every rank regrids at the same step)"""
        import torch.distributed as dist
        if dist.get_backend() == "gloo":
            return None
        if _CTL.get("group") is None:
            _CTL["group"] = dist.new_group(backend="gloo")
        return _CTL["group"]

    FIELDS = (("chi", _l.CHI), ("vel", _l.VEL), ("vold", _l.VOLD), ("pres", _l.PRES), ("pold", _l.POLD))
    UNIT = 64 * (1 + 2 + 2 + 1 + 1)  # doubles of one migrating block: 448 (the reference moves 1040-byte
    #                                                      MPI_Blocks per field, main.cpp:5198-5424; here one message unit)

    def _download_units(self, local_blocks):
        """[n][UNIT]: the five fields of the listed owned blocks, field after field"""
        n = len(local_blocks)
        out = np.empty((n, self.UNIT))
        idx = np.ascontiguousarray(local_blocks, dtype=np.int32)
        o = 0
        for _, f in self.FIELDS:
            w = 64 * _l.FIELD_DIM[f]
            buf = np.empty((n, w))
            if n:
                _l.check(self.L.cup2d_download_blocks(self._ctx, f, n, idx.ctypes.data_as(ctypes.c_void_p), buf.ctypes.data_as(ctypes.c_void_p)),
                         "download_blocks")
            out[:, o:o + w] = buf
            o += w
        return out

    def adapt(self, rtol, ctol, level_max, gather_all=False):
        """adapt() of main.cpp:4657-5440 on N ranks.  Tags: max|vorticity| of this rank's blocks (GPU), one double per block
        gathered; the states are validated and the new leaf list is derived on every rank from the (replicated, 12 bytes per
        block) leaf list -- deterministic, no negotiation rounds.  The FIELDS are not gathered: the new list is cut into
        contiguous ranges again (the load balance: a block whose range changes has migrated, main.cpp:5055-5424), and a rank
        fetches exactly what its new range is made of -- the unchanged blocks it did not own before (whole blocks from their old
        owners) and the old blocks its prolonged / restricted blocks are computed from (cup2d_amr_regrid_local's needed_old:
        refined parents with their 3 x 3 neighbourhood, compressing siblings -- the reference gathers siblings on one rank the
        same way, 5120-5130).  Unchanged blocks that stay on the rank move between the old and the new context on the device.
        gather_all: the round-2 form (every field of every rank to every rank) -- the cross-check of the tests.
        Returns True if the grid changed; self.regrid_stats says what moved."""
        self.vorticity()
        linf = np.empty(self.part.nowned)
        _l.check(self.L.cup2d_block_linf(self._ctx, _l.TMP, linf.ctypes.data_as(ctypes.c_void_p)), "block_linf")
        G = self.global_grid
        linf = self._allgather_blocks(linf)
        st = validate_states(G.blocks, tag_states(linf, G.level, rtol, ctol, level_max), level_max, G.bpdx, G.bpdy)
        if not (st != LEAVE).any():
            return False
        names = dict(self.FIELDS)
        if gather_all:
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
        # ---- what this rank's new range is made of: fetched from the old owners, the changed blocks computed ----
        old, rank = self.part, self.rank
        R = fetch_new_range(G, old, st, level_max, rank, self.world, lambda ids: self._download_units(ids), self._ctl_group(),
                            self.FIELDS)
        new_blocks, n_new, lo, hi, my_src, kept, slot, comp, data = (R[k] for k in ("new_blocks", "n_new", "lo", "hi", "my_src", "kept",
                                                                                   "slot", "comp", "data"))
        self.regrid_stats = R["stats"]
        # ---- the new context; what stays on this rank unchanged moves on the device ----
        new_grid = AmrBlockGrid(new_blocks, G.bpdx, G.bpdy, G.h0 * max(G.bpdx, G.bpdy) * BS)
        old_ctx, old_lo = self._ctx, old.lo
        old_owner_of_src = R["old_owner_of_src"]
        self._ctx = ctypes.c_void_p()  # (the old context lives on until its blocks have been copied over)
        vp = ctypes.c_void_p
        try:
            self.__init__(new_grid, nu=self.nu, cfl=self.cfl, device=self.device, comm=self._comm_kind, mode=self._comm_mode,
                          group=self._group, adapt_steps=self.adapt_steps)
            stay = kept & (old_owner_of_src == rank)
            dst_dev = np.ascontiguousarray(np.flatnonzero(stay), dtype=np.int32)
            src_dev = np.ascontiguousarray(my_src[stay] - old_lo, dtype=np.int32)
            up_idx = np.ascontiguousarray(np.flatnonzero(~stay), dtype=np.int32)
            for k, f in self.FIELDS:
                if len(dst_dev):
                    _l.check(self.L.cup2d_copy_blocks(self._ctx, old_ctx, f, len(dst_dev), dst_dev.ctypes.data_as(vp), src_dev.ctypes.data_as(vp)),
                             "copy_blocks")
                if len(up_idx):
                    rows = data[k][up_idx]
                    moved = kept[up_idx]                       # unchanged blocks that migrated here: their received copy
                    if moved.any():
                        rows[moved] = comp[k][0][slot[my_src[up_idx[moved]]]]
                    if np.isnan(rows).any():
                        raise RuntimeError("regrid: field %s of a new block was not produced (plan and data disagree)" % k)
                    rows = np.ascontiguousarray(rows)
                    _l.check(self.L.cup2d_upload_blocks(self._ctx, f, len(up_idx), up_idx.ctypes.data_as(vp), rows.ctypes.data_as(vp)),
                             "upload_blocks")
        finally:
            self.L.cup2d_destroy(old_ctx)
        self.install_poisson_matrix()
        return True


FIELDS = (("chi", _l.CHI), ("vel", _l.VEL), ("vold", _l.VOLD), ("pres", _l.PRES), ("pold", _l.POLD))


def fetch_new_range(G, old, st, level_max, rank, world, download_units, group, fields=FIELDS):
    """The data side of a regrid on N ranks (main.cpp:5055-5424) for ONE rank: the new leaf list is cut into contiguous
    ranges; this rank fetches exactly what its new range [lo, hi) is made of -- unchanged blocks it did not own (whole
    blocks from their old owners: migration) and the old blocks its prolonged / restricted blocks are computed from
    (cup2d_amr_regrid_local's needed_old) -- and computes its changed blocks.  G: the old global grid (leaf list on every
    rank), old: this rank's AmrPartition of it, st: the validated states (the same on every rank), download_units(local old
    block indices) -> [n][UNIT] the rank's own block data, field after field (the device in production, numpy in the CPU
    tests), group: a gloo process group (or None: the default one).  Nothing outside `want` is read or received."""
    import torch
    import torch.distributed as dist
    from .amr import regrid_local_plan, regrid_local_compute
    unit = sum(64 * _l.FIELD_DIM[f] for _, f in fields)
    nb_old = G.nblocks
    n_new = regrid_local_plan(G.blocks, st, level_max, 0, 0, G.bpdx, G.bpdy)
    nbounds = partition_bounds(n_new, world)
    lo, hi = int(nbounds[rank]), int(nbounds[rank + 1])
    new_blocks, src, needed = regrid_local_plan(G.blocks, st, level_max, lo, hi, G.bpdx, G.bpdy, n_new=n_new)
    my_src = src[lo:hi]
    kept = my_src >= 0
    want = np.union1d(np.flatnonzero(needed), my_src[kept]).astype(np.int64)  # old blocks (global ids) this rank reads
    owner = old.owner[want]
    remote = want[owner != rank]
    # ---- who needs what from whom: the request lists travel (ids only), then the blocks ----
    reqs = [None] * world
    dist.all_gather_object(reqs, remote, group=group)
    send_ids = []
    for p in range(world):
        r = np.asarray(reqs[p], dtype=np.int64)
        send_ids.append(r[(r >= old.lo) & (r < old.hi)] if p != rank else np.zeros(0, np.int64))
    recv_ids = [remote[old.owner[remote] == p] for p in range(world)]
    ops, keep, recv_buf = [], [], {}
    for p in range(world):
        if len(send_ids[p]):
            t = torch.from_numpy(np.ascontiguousarray(download_units(send_ids[p] - old.lo)))
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, p, group=group))
        if len(recv_ids[p]):
            recv_buf[p] = torch.empty((len(recv_ids[p]), unit), dtype=torch.float64)
            ops.append(dist.P2POp(dist.irecv, recv_buf[p], p, group=group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    # ---- compact arrays of what this rank holds now: needed blocks of its own + everything that arrived ----
    local_need = want[owner == rank]
    local_need = local_need[needed[local_need]]          # own blocks that are only COPIED stay where they are
    held = np.concatenate([local_need] + [recv_ids[p] for p in range(world) if len(recv_ids[p])]).astype(np.int64)
    units = np.concatenate([download_units(local_need - old.lo).reshape(-1, unit)] +
                           [recv_buf[p].numpy() for p in range(world) if len(recv_ids[p])])
    slot = -np.ones(nb_old, dtype=np.int32)
    slot[held] = np.arange(len(held), dtype=np.int32)
    comp, o = {}, 0
    for k, f in fields:
        w = 64 * _l.FIELD_DIM[f]
        comp[k] = (np.ascontiguousarray(units[:, o:o + w]), _l.FIELD_DIM[f], _l.FIELD_DIM[f] == 2)
        o += w
    data = regrid_local_compute(G.blocks, st, level_max, lo, hi, n_new, slot, comp, G.bpdx, G.bpdy)
    stats = dict(old_blocks=nb_old, new_blocks=int(n_new), owned_before=old.nowned, owned_after=hi - lo,
                 blocks_received=int(sum(len(r) for r in recv_ids)), blocks_sent=int(sum(len(x) for x in send_ids)),
                 own_blocks_downloaded=int(len(local_need)), changed_blocks_computed=int((~kept).sum()),
                 host_blocks_held=int(len(held)))
    return dict(new_blocks=new_blocks, n_new=n_new, lo=lo, hi=hi, my_src=my_src, kept=kept, slot=slot, comp=comp, data=data,
                stats=stats, old_owner_of_src=old.owner[np.maximum(my_src, 0)])


_CTL = {}
