"""Block-AMR grids on the host (BASELINE.json configs[4]): which leaf blocks exist on which level and what lies
across every side of each -- the tables of cup2d_set_amr (include/cup2d_hip.h).  Replaces, for this purpose, the
reference's tree / Info::Znei / Zchild / Zparent lookups (main.cpp:672-738, 2197-2198) with dense arrays built once
per regrid.  Fields of an adapted grid are per-block arrays [nblocks][64 * dim] in the order of `blocks`."""
import ctypes
import os

import numpy as np

from . import lib as _l
from .simulation import BodyOps

BS = 8


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class AmrBlockGrid:
    """blocks: (nb, 3) int array of leaf blocks (level, i, j) of a bpdx x bpdy base grid (level-0 blocks), 2:1
    balanced (the reference's adapt() guarantees it, main.cpp:4734-4861).  h0 = extent / max(bpdx, bpdy) / 8
    (main.cpp:6338)."""

    def __init__(self, blocks, bpdx=1, bpdy=1, extent=1.0):
        self.blocks = np.ascontiguousarray(blocks, dtype=np.int64).reshape(-1, 3)
        self.nblocks = len(self.blocks)
        self.bpdx, self.bpdy = int(bpdx), int(bpdy)
        self.h0 = float(extent) / max(self.bpdx, self.bpdy) / BS
        nb = self.nblocks
        self.level = np.ascontiguousarray(self.blocks[:, 0], dtype=np.int32)
        self.kind = np.zeros((nb, 4), dtype=np.int32)
        self.nbr2 = -np.ones((nb, 4, 2), dtype=np.int32)
        self.half = np.zeros((nb, 4), dtype=np.int32)
        b32 = np.ascontiguousarray(self.blocks, dtype=np.int32)
        L = _l.load_library()
        rc = L.cup2d_amr_tables(nb, _p(b32), self.bpdx, self.bpdy, _p(self.kind), _p(self.nbr2), _p(self.half))
        if rc != 0:
            raise ValueError(L.cup2d_last_error().decode())
        # same-level neighbour table for cup2d_create (sides that are not same-level: wall)
        self.nbr = np.where(self.kind == _l.AMR_SAME, self.nbr2[:, :, 0], -1).astype(np.int32)
        self.nghost, self.n_inner = 0, nb

    def h(self, level):
        return self.h0 / (1 << int(level))

    # ---- Poisson matrix of main.cpp:7034-7112 on this adapted grid -------------------------------
    def poisson_coo(self):
        """COO triplets (row, col, val) of the matrix the reference assembles on this grid, assembled by the library's
        host routine cup2d_amr_poisson_coo (C++, include/cup2d_hip.h)"""
        L = _l.load_library()
        vp = ctypes.c_void_p
        tabs = [np.ascontiguousarray(a, dtype=np.int32) for a in (self.kind, self.nbr2, self.half)]
        ptr = [a.ctypes.data_as(vp) for a in tabs]
        nnz = L.cup2d_amr_poisson_coo(self.nblocks, *ptr, 0, None, None, None)
        if nnz < 0:
            _l.check(int(nnz), "amr_poisson_coo")
        r, c, v = np.empty(nnz, dtype=np.int32), np.empty(nnz, dtype=np.int32), np.empty(nnz, dtype=np.float64)
        got = L.cup2d_amr_poisson_coo(self.nblocks, *ptr, nnz, r.ctypes.data_as(vp), c.ctypes.data_as(vp), v.ctypes.data_as(vp))
        if got != nnz:
            _l.check(int(min(got, -1)), "amr_poisson_coo")
        return r, c, v

    def cell_centres(self):
        """x, y of every cell, (nb, 64) each, as the reference places them (origin main.cpp:695-696)"""
        l = self.blocks[:, 0]
        h = self.h0 / (1 << l)
        ox = self.blocks[:, 1] * BS * self.h0 / (1 << l)
        oy = self.blocks[:, 2] * BS * self.h0 / (1 << l)
        ix = np.tile(np.arange(BS), BS)
        iy = np.repeat(np.arange(BS), BS)
        return ox[:, None] + (ix[None, :] + 0.5) * h[:, None], oy[:, None] + (iy[None, :] + 0.5) * h[:, None]


class AmrSimulation(BodyOps):
    """Device-resident fields on an adapted grid + the halo-1 block operators in their AMR form.  Fields are set and
    read as per-block arrays (nb, 64) / (nb, 64, 2)."""

    def __init__(self, grid, nu=1e-3, cfl=0.5, device=0, adapt_steps=20):
        self.L = _l.load_library()
        self.grid, self.nu, self.cfl = grid, float(nu), float(cfl)
        self.device = int(device)
        self.adapt_steps = int(adapt_steps)  # sim.AdaptSteps (main.cpp:6603; run.sh passes 20)
        self.step_count = getattr(self, "step_count", 0)
        # settings that survive the context rebuild of adapt(): arithmetic policy, solver organisation, timing
        self._strict = getattr(self, "_strict", None)
        self._solver = getattr(self, "_solver", None)
        self._timing = getattr(self, "_timing", None)
        self._ctx = ctypes.c_void_p()
        vp = ctypes.c_void_p
        _l.check(self.L.cup2d_create(ctypes.byref(self._ctx), grid.nblocks, 0, grid.nblocks,
                                     np.ascontiguousarray(grid.nbr).ctypes.data_as(vp), grid.h0, int(device)), "cup2d_create")
        self._tables = [np.ascontiguousarray(a, dtype=np.int32) for a in (grid.level, grid.kind, grid.nbr2, grid.half)]
        _l.check(self.L.cup2d_set_amr(self._ctx, grid.h0, *[a.ctypes.data_as(vp) for a in self._tables]), "cup2d_set_amr")
        if self._strict is not None:
            self.set_math(self._strict)
        if self._solver is not None:
            self.set_solver(*self._solver)
        if self._timing is not None:
            self.set_timing(self._timing)

    def ctx_ptr(self):
        """the cup2d_ctx* of this simulation (for C-ABI calls that take two contexts)"""
        return self._ctx

    def set_solver(self, fused=False, finish_in_kernel=False, form=None):
        """form: 'auto' | 'full' | 'edge' | 'eab' (cup2d_set_solver_form; None leaves it as it is).  On the hybrid operator of an
        adapted grid 'auto' / 'eab' with the finish in the kernel = two sweeps + two rows launches per iteration (k_edge HYB),
        'full' = three sweeps + two (k_fused HYB)"""
        if form is None and self._solver is not None and len(self._solver) > 2:
            form = self._solver[2]  # (kept across the contexts a regrid creates)
        self._solver = (bool(fused), bool(finish_in_kernel), form)
        _l.check(self.L.cup2d_set_solver(self._ctx, _l.SOLVER_FUSED if fused else _l.SOLVER_SWEEPS, int(finish_in_kernel)), "set_solver")
        if form is not None:
            _l.check(self.L.cup2d_set_solver_form(self._ctx, ("auto", "full", "edge", "eab").index(form)), "set_solver_form")

    def last_solver_form(self):
        """(form, merge, handover mask) of the last fused solve (cup2d_get_last_solver_form): form 'full' | 'edge' | 'eab'"""
        f, m, h = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _l.check(self.L.cup2d_get_last_solver_form(self._ctx, ctypes.byref(f), ctypes.byref(m), ctypes.byref(h)), "get_last_solver_form")
        return ("none", "full", "edge", "eab")[f.value], m.value, h.value

    def last_solver(self):
        """'fused' or 'sweeps': what the last poisson_solve ran (the hybrid assembled operator takes the tile-fused sweeps)"""
        k = ctypes.c_int()
        _l.check(self.L.cup2d_get_last_solver(self._ctx, ctypes.byref(k)), "get_last_solver")
        return "fused" if k.value == _l.SOLVER_FUSED else "sweeps"

    def poisson_solve(self, tol=1e-9, rel_tol=0.0, max_restarts=100, max_iter=1000):
        """b = TMP, x0 = PRES -> PRES on the installed operator (cuda.cu:403-548)"""
        it, rs, e, e0 = ctypes.c_int(), ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        _l.check(self.L.cup2d_poisson_solve(self._ctx, tol, rel_tol, max_restarts, max_iter, ctypes.byref(it), ctypes.byref(rs),
                                            ctypes.byref(e), ctypes.byref(e0)), "poisson_solve")
        return dict(iters=it.value, restarts=rs.value, err=e.value, err_init=e0.value)

    def matrix_stats(self):
        """how the installed operator is applied (include/cup2d_hip.h cup2d_matrix_stats)"""
        a, b, e = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong()
        _l.check(self.L.cup2d_matrix_stats(self._ctx, ctypes.byref(a), ctypes.byref(b), ctypes.byref(e)), "matrix_stats")
        return dict(plain_blocks=a.value, general_tile_blocks=b.value, stored_entries=e.value)

    def set_timing(self, on=True):
        self._timing = int(on)
        _l.check(self.L.cup2d_set_timing(self._ctx, int(on)), "set_timing")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def should_adapt(self):
        """main.cpp:6603: the reference regrids on every one of the first eleven steps, then every AdaptSteps-th"""
        return self.step_count <= 10 or self.step_count % self.adapt_steps == 0

    def close(self):
        if self._ctx:
            self.L.cup2d_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_field(self, field, a):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(self.grid.nblocks, 64 * _l.FIELD_DIM[field])
        _l.check(self.L.cup2d_upload_slab(self._ctx, field, a.ctypes.data_as(ctypes.c_void_p)), "upload_slab")

    def get_field(self, field):
        dim = _l.FIELD_DIM[field]
        a = np.empty((self.grid.nblocks, 64 * dim))
        _l.check(self.L.cup2d_download_slab(self._ctx, field, a.ctypes.data_as(ctypes.c_void_p)), "download_slab")
        return a.reshape(self.grid.nblocks, 64, 2) if dim == 2 else a

    def set_math(self, strict):
        self._strict = bool(strict)
        _l.check(self.L.cup2d_set_math(self._ctx, _l.MATH_STRICT if strict else _l.MATH_FAST), "set_math")

    def advect_diffuse_rhs(self, dt):
        """prepare0 / computeA<VectorLab>(KernelAdvectDiffuse(), var.vel, 2) / fillcases (main.cpp:6611-6617): tmpV"""
        _l.check(self.L.cup2d_advect_diffuse_rhs(self._ctx, self.nu, float(dt), _l.BLOCKS_ALL), "advect_diffuse_rhs")

    def laplacian_sub(self):
        """prepare0 / computeA<ScalarLab>(pressure_rhs1(), var.pold, 1) / fillcases (main.cpp:7022-7027)"""
        _l.check(self.L.cup2d_laplacian_sub(self._ctx, _l.BLOCKS_ALL), "laplacian_sub")

    def vorticity(self):
        _l.check(self.L.cup2d_vorticity(self._ctx, _l.BLOCKS_ALL), "vorticity")

    def pressure_rhs(self, dt):
        """computeB<pressure_rhs,..> + fillcases (main.cpp:7007-7013): tmp from vel, tmpV (= udef), chi"""
        _l.check(self.L.cup2d_pressure_rhs(self._ctx, float(dt), 1, _l.BLOCKS_ALL), "pressure_rhs")

    def pressure_correction(self, dt):
        _l.check(self.L.cup2d_pressure_correction(self._ctx, float(dt), _l.BLOCKS_ALL), "pressure_correction")

    def apply_A(self, dst, src):
        _l.check(self.L.cup2d_apply_A(self._ctx, dst, src), "apply_A")

    def install_poisson_matrix(self, via_triplets=False):
        """the coarse-fine Poisson operator of this grid, what the reference assembles after every regrid
        (main.cpp:7034-7113): built by the library from the tables of cup2d_set_amr (cup2d_amr_install_poisson: rows only
        where a block has a coarse-fine side); via_triplets: the COO route (AmrBlockGrid.poisson_coo -> cup2d_set_matrix_coo),
        the same operator bit for bit"""
        if not via_triplets:
            _l.check(self.L.cup2d_amr_install_poisson(self._ctx), "amr_install_poisson")
            return
        r, c, v = self.grid.poisson_coo()
        vp = ctypes.c_void_p
        _l.check(self.L.cup2d_set_matrix_coo(self._ctx, 0, len(v), r.ctypes.data_as(vp), c.ctypes.data_as(vp), v.ctypes.data_as(vp)),
                 "set_matrix_coo")

    def step(self, tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=1000, dt=None):
        """one pass of the time-loop body on the (fixed) adapted grid: dt, RK2 WENO5 advect-diffuse with flux correction,
        Poisson rhs, BiCGSTAB on the assembled operator, volume-weighted mean removal + projection (main.cpp:6576-7187
        without adapt()).  dt: the reference computes dt BEFORE it regrids (main.cpp:6579-6603); a caller that adapts
        between the two passes the value compute_dt() gave on the old grid."""
        it, e = ctypes.c_int(), ctypes.c_double()
        if dt is None:
            d = ctypes.c_double()
            _l.check(self.L.cup2d_step(self._ctx, self.nu, self.cfl, float(tol), float(rel_tol), int(max_restarts), int(max_iter),
                                       ctypes.byref(d), ctypes.byref(it), ctypes.byref(e)), "step")
            if d.value > 2e-16:  # main.cpp:6596: a vanishing dt advances nothing
                self.step_count += 1
            return dict(dt=d.value, iters=it.value, err=e.value)
        dt = float(dt)
        if not dt > 2e-16:
            return dict(dt=dt, iters=0, err=0.0)
        self.step_count += 1
        _l.check(self.L.cup2d_advect_diffuse_rk2(self._ctx, self.nu, dt), "advect_diffuse_rk2")
        _l.check(self.L.cup2d_poisson_rhs(self._ctx, dt, 0), "poisson_rhs")
        _l.check(self.L.cup2d_poisson_solve(self._ctx, float(tol), float(rel_tol), int(max_restarts), int(max_iter), ctypes.byref(it),
                                            None, ctypes.byref(e), None), "poisson_solve")
        _l.check(self.L.cup2d_project(self._ctx, dt), "project")
        return dict(dt=dt, iters=it.value, err=e.value)

    def adapt(self, rtol, ctol, level_max, host_fields=None, route=None):
        """The reference's adapt() (main.cpp:4657-5440) for this simulation: tag by max|vorticity| per block (GPU),
        validate the states, prolong / restrict / copy the fields, rebuild the device context on the new grid and re-assemble
        the Poisson operator.  Returns True if the grid changed.  The caller decides WHEN (should_adapt() is the reference's
        rule, main.cpp:6603).  Body-free: the reference also runs GradChiOnTmp on chi before tagging (main.cpp:4660), which
        only matters with bodies.
        route: 'device' (default) -- prolongation (main.cpp:4981-5032) and restriction (5149-5166) as kernels between the old
               and the new context, no field crosses PCIe (cup2d_amr_regrid_device);
               'changed' -- the blocks that change are computed on the host, only they cross PCIe (round 2);
               'host' -- every field through host memory (round 1).  All three leave the same bits (tests/test_amr.py).
        host_fields (older spelling): True = 'host', False = 'changed'."""
        if route is None:
            route = "device" if host_fields is None else ("host" if host_fields else "changed")
        if route not in ("device", "changed", "host"):
            raise ValueError("route must be 'device', 'changed' or 'host'")
        host_fields = route == "host"
        import time as _t
        tm, t0 = {}, _t.perf_counter()

        def lap(name):
            nonlocal t0
            t1 = _t.perf_counter()
            tm[name] = tm.get(name, 0.0) + (t1 - t0) * 1e3
            t0 = t1
        self.adapt_stages_ms = tm  # where the last adapt() spent its time (host clock, ms)
        self.vorticity()
        linf = np.empty(self.grid.nblocks)
        _l.check(self.L.cup2d_block_linf(self._ctx, _l.TMP, _p(linf)), "block_linf")  # one double per block crosses PCIe
        lap("tags")
        st = validate_states(self.grid.blocks, tag_states(linf, self.grid.level, rtol, ctol, level_max), level_max,
                             self.grid.bpdx, self.grid.bpdy)
        lap("states")
        if not (st != LEAVE).any():
            return False
        G = self.grid
        nbk, vp = G.nblocks, ctypes.c_void_p
        names = {"chi": _l.CHI, "vel": _l.VEL, "vold": _l.VOLD, "pres": _l.PRES, "pold": _l.POLD}
        if host_fields:  # every field through host memory (the round-1 form; also the cross-check of the tests)
            fields = {k: (self.get_field(f).reshape(nbk, -1), _l.FIELD_DIM[f], _l.FIELD_DIM[f] == 2) for k, f in names.items()}
            blocks, data = regrid(G.blocks, st, fields, level_max, G.bpdx, G.bpdy)
            new_grid = AmrBlockGrid(blocks, G.bpdx, G.bpdy, G.h0 * max(G.bpdx, G.bpdy) * BS)
            self.close()
            # same device, same settings (the rebuilt context gets policy / solver / timing re-applied by __init__)
            self.__init__(new_grid, nu=self.nu, cfl=self.cfl, device=self.device, adapt_steps=self.adapt_steps)
            for k, f in names.items():
                self.set_field(f, data[k])
            self.install_poisson_matrix()
            return True
        # The fields stay on the device: the library's plan says which new blocks are unchanged copies of which old ones
        # (moved by a kernel between the old and the new context) and which old blocks the prolonged / restricted ones are
        # computed from; only those come to the host, only the changed blocks go back.
        b32 = np.ascontiguousarray(G.blocks, dtype=np.int32).reshape(-1, 3)
        st32 = np.ascontiguousarray(st, dtype=np.int32)
        cap = 4 * nbk  # every block refined: one call instead of count + fill
        new_blocks = np.empty((cap, 3), dtype=np.int32)
        src = np.empty(cap, dtype=np.int32)
        needed = np.empty(nbk, dtype=np.int32)
        n_new = self.L.cup2d_amr_regrid_plan(nbk, _p(b32), G.bpdx, G.bpdy, level_max, _p(st32), cap, _p(new_blocks), _p(src), _p(needed))
        if n_new < 0:
            _l.check(int(n_new), "amr_regrid_plan")
        new_blocks, src = new_blocks[:n_new], src[:n_new]
        if route == "device":
            lap("plan")
            new_grid = AmrBlockGrid(new_blocks.astype(np.int64), G.bpdx, G.bpdy, G.h0 * max(G.bpdx, G.bpdy) * BS)
            lap("tables")
            old_ctx = self._ctx
            self._ctx = ctypes.c_void_p()  # the old context lives on until the kernels have read it
            try:
                self.__init__(new_grid, nu=self.nu, cfl=self.cfl, device=self.device, adapt_steps=self.adapt_steps)
                self.adapt_stages_ms = tm
                lap("context")
                flds = np.array(list(names.values()), dtype=np.int32)
                _l.check(self.L.cup2d_amr_regrid_device(self._ctx, old_ctx, nbk, _p(b32), G.bpdx, G.bpdy, level_max, _p(st32), len(flds),
                                                        _p(flds)), "amr_regrid_device")
                lap("fields")
            finally:
                self.L.cup2d_destroy(old_ctx)
            lap("destroy_old")
            self.install_poisson_matrix()
            lap("operator")
            return True
        need_idx = np.ascontiguousarray(np.flatnonzero(needed), dtype=np.int32)
        changed = np.ascontiguousarray(np.flatnonzero(src < 0), dtype=np.int32)
        kept_new = np.ascontiguousarray(np.flatnonzero(src >= 0), dtype=np.int32)
        kept_old = np.ascontiguousarray(src[kept_new], dtype=np.int32)
        order = list(names)
        dims = np.array([_l.FIELD_DIM[names[k]] for k in order], dtype=np.int32)
        vec = (dims == 2).astype(np.int32)
        old_host, new_host = [], []
        poison = bool(os.environ.get("CUP2D_REGRID_POISON"))
        for k, d in zip(order, dims):  # full-size arrays the host routine indexes by block; only the needed blocks are filled
            # (untouched pages of np.empty cost nothing; CUP2D_REGRID_POISON=1 fills with NaN to catch a read outside the plan)
            a = np.full((nbk, 64 * int(d)), np.nan) if poison else np.empty((nbk, 64 * int(d)))
            got = np.empty((len(need_idx), 64 * int(d)))
            _l.check(self.L.cup2d_download_blocks(self._ctx, names[k], len(need_idx), _p(need_idx), _p(got)), "download_blocks")
            a[need_idx] = got
            old_host.append(a)
            new_host.append(np.empty((n_new, 64 * int(d))))
        srcp = (vp * len(order))(*[a.ctypes.data for a in old_host])
        dstp = (vp * len(order))(*[a.ctypes.data for a in new_host])
        chk = np.empty((n_new, 3), dtype=np.int32)
        if self.L.cup2d_amr_regrid_changed(nbk, _p(b32), G.bpdx, G.bpdy, level_max, _p(st32), len(order), srcp, _p(dims), _p(vec), n_new,
                                           _p(chk), dstp) != n_new:
            _l.check(-1, "amr_regrid_changed")
        new_grid = AmrBlockGrid(new_blocks.astype(np.int64), G.bpdx, G.bpdy, G.h0 * max(G.bpdx, G.bpdy) * BS)
        old_ctx = self._ctx
        self._ctx = ctypes.c_void_p()  # the old context lives on until its blocks have been copied over
        try:
            self.__init__(new_grid, nu=self.nu, cfl=self.cfl, device=self.device, adapt_steps=self.adapt_steps)
            for k, a in zip(order, new_host):
                up = np.ascontiguousarray(a[changed])
                if poison and np.isnan(up).any():  # a block the plan did not name was read: the plan is the library's, so this is a bug
                    raise RuntimeError("regrid: a prolonged / restricted block of %s was computed from a block that was not downloaded" % k)
                _l.check(self.L.cup2d_copy_blocks(self._ctx, old_ctx, names[k], len(kept_new), _p(kept_new), _p(kept_old)), "copy_blocks")
                _l.check(self.L.cup2d_upload_blocks(self._ctx, names[k], len(changed), _p(changed), _p(up)), "upload_blocks")
        finally:
            self.L.cup2d_destroy(old_ctx)
        self.install_poisson_matrix()
        return True

    def compute_dt(self):
        v = ctypes.c_double()
        _l.check(self.L.cup2d_compute_dt(self._ctx, self.nu, self.cfl, ctypes.byref(v)), "compute_dt")
        return v.value


# ---- regridding: the reference's adapt() (main.cpp:4657-5440) for one rank ---------------------------------------------
LEAVE, REFINE, COMPRESS = 0, 1, 2


def tag_states(linf, level, rtol, ctol, level_max):
    """main.cpp:4678-4690: Refine where max|vorticity| of the block exceeds Rtol, Compress where it is below Ctol;
    the finest level cannot refine, level 0 cannot compress"""
    linf = np.asarray(linf, dtype=np.float64)
    level = np.asarray(level)
    st = np.where(linf > rtol, REFINE, np.where(linf < ctol, COMPRESS, LEAVE)).astype(np.int32)
    st[(st == REFINE) & (level == level_max - 1)] = LEAVE
    st[(st == COMPRESS) & (level == 0)] = LEAVE
    return st


def validate_states(blocks, states, level_max, bpdx=1, bpdy=1):
    """The reference's state validation (main.cpp:4718-4861) by the library's host routine cup2d_amr_validate_states.
    Returns the final states."""
    b32 = np.ascontiguousarray(blocks, dtype=np.int32).reshape(-1, 3)
    st = np.array(states, dtype=np.int32)
    L = _l.load_library()
    _l.check(L.cup2d_amr_validate_states(len(b32), _p(b32), bpdx, bpdy, level_max, _p(st)), "amr_validate_states")
    return st


def regrid(blocks, states, fields, level_max, bpdx=1, bpdy=1):
    """Apply final states with the library's host routine cup2d_amr_regrid (include/cup2d_hip.h).  fields: {name: (array (nb, 64*dim), dim,
    is_vector)}.  Returns (new_blocks, new_fields)."""
    b32 = np.ascontiguousarray(blocks, dtype=np.int32).reshape(-1, 3)
    st = np.ascontiguousarray(states, dtype=np.int32)
    nb = len(b32)
    names = list(fields)
    src = [np.ascontiguousarray(fields[k][0], dtype=np.float64).reshape(nb, -1) for k in names]
    dims = np.array([fields[k][1] for k in names], dtype=np.int32)
    vec = np.array([1 if fields[k][2] else 0 for k in names], dtype=np.int32)
    L = _l.load_library()
    vp = ctypes.c_void_p
    n = len(names)
    srcp = (vp * max(n, 1))(*[a.ctypes.data for a in src])
    n_new = L.cup2d_amr_regrid(nb, _p(b32), bpdx, bpdy, level_max, _p(st), n, srcp, _p(dims), _p(vec), 0, None, None)
    if n_new < 0:
        _l.check(int(n_new), "amr_regrid")
    new_blocks = np.empty((n_new, 3), dtype=np.int32)
    dst = [np.empty((n_new, BS * BS * int(d))) for d in dims]
    dstp = (vp * max(n, 1))(*[a.ctypes.data for a in dst])
    got = L.cup2d_amr_regrid(nb, _p(b32), bpdx, bpdy, level_max, _p(st), n, srcp, _p(dims), _p(vec), n_new, _p(new_blocks), dstp)
    if got != n_new:
        _l.check(int(min(got, -1)), "amr_regrid")
    return new_blocks.astype(np.int64), dict(zip(names, dst))


def regrid_local_plan(blocks, states, level_max, new_lo, new_hi, bpdx=1, bpdy=1, n_new=None):
    """cup2d_amr_regrid_local, plan part, for the new blocks at positions [new_lo, new_hi): (new_blocks (n, 3) -- the whole
    new leaf list --, src_of_new (n,), needed_old (nb,) bool: what the prolonged / restricted blocks OF THE RANGE are
    computed from).  new_lo = new_hi = 0 with n_new None: the count only (returns n)."""
    b32 = np.ascontiguousarray(blocks, dtype=np.int32).reshape(-1, 3)
    st = np.ascontiguousarray(states, dtype=np.int32)
    L, nb = _l.load_library(), len(b32)
    if n_new is None:
        n = L.cup2d_amr_regrid_local(nb, _p(b32), bpdx, bpdy, level_max, _p(st), 0, 0, 0, None, None, None, 0, None, None, None, None, None)
        if n < 0:
            _l.check(int(n), "amr_regrid_local")
        return int(n)
    n = int(n_new)
    new_blocks, src, needed = np.empty((n, 3), dtype=np.int32), np.empty(n, dtype=np.int32), np.empty(nb, dtype=np.int32)
    got = L.cup2d_amr_regrid_local(nb, _p(b32), bpdx, bpdy, level_max, _p(st), int(new_lo), int(new_hi), n, _p(new_blocks), _p(src),
                                   _p(needed), 0, None, None, None, None, None)
    if got != n:
        _l.check(int(min(got, -1)), "amr_regrid_local")
    return new_blocks.astype(np.int64), src, needed.astype(bool)


def regrid_local_compute(blocks, states, level_max, new_lo, new_hi, n_new, slot_of_old, fields, bpdx=1, bpdy=1):
    """cup2d_amr_regrid_local, compute part: fields {name: (compact array (nslots, 64 * dim), dim, is_vector)} hold old block k at
    row slot_of_old[k]; returns {name: array (new_hi - new_lo, 64 * dim)} with the prolonged / restricted blocks of the range
    written (the rows of unchanged copies are left as NaN: the caller moves those)."""
    b32 = np.ascontiguousarray(blocks, dtype=np.int32).reshape(-1, 3)
    st = np.ascontiguousarray(states, dtype=np.int32)
    slot = np.ascontiguousarray(slot_of_old, dtype=np.int32)
    L, nb = _l.load_library(), len(b32)
    names = list(fields)
    src = [np.ascontiguousarray(fields[k][0], dtype=np.float64) for k in names]
    dims = np.array([fields[k][1] for k in names], dtype=np.int32)
    vec = np.array([1 if fields[k][2] else 0 for k in names], dtype=np.int32)
    vp = ctypes.c_void_p
    srcp = (vp * max(len(names), 1))(*[a.ctypes.data for a in src])
    dst = [np.full((int(new_hi - new_lo), BS * BS * int(d)), np.nan) for d in dims]
    dstp = (vp * max(len(names), 1))(*[a.ctypes.data for a in dst])
    new_blocks = np.empty((int(n_new), 3), dtype=np.int32)
    got = L.cup2d_amr_regrid_local(nb, _p(b32), bpdx, bpdy, level_max, _p(st), int(new_lo), int(new_hi), int(n_new), _p(new_blocks),
                                   None, None, len(names), srcp, _p(slot), _p(dims), _p(vec), dstp)
    if got != n_new:
        _l.check(int(min(got, -1)), "amr_regrid_local")
    return dict(zip(names, dst))


def regrid_plan(blocks, states, level_max, bpdx=1, bpdy=1):
    """cup2d_amr_regrid_plan: (new_blocks (n,3), src_of_new (n,) old block of an unchanged copy or -1, needed_old (nb,) bool:
    the old blocks the prolonged / restricted blocks are computed from)"""
    b32 = np.ascontiguousarray(blocks, dtype=np.int32).reshape(-1, 3)
    st = np.ascontiguousarray(states, dtype=np.int32)
    L, nb = _l.load_library(), len(b32)
    n = L.cup2d_amr_regrid_plan(nb, _p(b32), bpdx, bpdy, level_max, _p(st), 0, None, None, None)
    if n < 0:
        _l.check(int(n), "amr_regrid_plan")
    new_blocks, src, needed = np.empty((n, 3), dtype=np.int32), np.empty(n, dtype=np.int32), np.empty(nb, dtype=np.int32)
    if L.cup2d_amr_regrid_plan(nb, _p(b32), bpdx, bpdy, level_max, _p(st), n, _p(new_blocks), _p(src), _p(needed)) != n:
        _l.check(-1, "amr_regrid_plan")
    return new_blocks.astype(np.int64), src, needed.astype(bool)


def regrid_changed(blocks, states, fields, level_max, bpdx=1, bpdy=1):
    """cup2d_amr_regrid_changed: like regrid(), but only the prolonged / restricted blocks of the returned arrays are
    written (the rest is uninitialised) and only the needed blocks of `fields` are read"""
    b32 = np.ascontiguousarray(blocks, dtype=np.int32).reshape(-1, 3)
    st = np.ascontiguousarray(states, dtype=np.int32)
    nb, names = len(b32), list(fields)
    src = [np.ascontiguousarray(fields[k][0], dtype=np.float64).reshape(nb, -1) for k in names]
    dims = np.array([fields[k][1] for k in names], dtype=np.int32)
    vec = np.array([1 if fields[k][2] else 0 for k in names], dtype=np.int32)
    L, vp = _l.load_library(), ctypes.c_void_p
    n = L.cup2d_amr_regrid_plan(nb, _p(b32), bpdx, bpdy, level_max, _p(st), 0, None, None, None)
    if n < 0:
        _l.check(int(n), "amr_regrid_plan")
    new_blocks = np.empty((n, 3), dtype=np.int32)
    dst = [np.empty((n, BS * BS * int(d))) for d in dims]
    srcp = (vp * max(len(names), 1))(*[a.ctypes.data for a in src])
    dstp = (vp * max(len(names), 1))(*[a.ctypes.data for a in dst])
    if L.cup2d_amr_regrid_changed(nb, _p(b32), bpdx, bpdy, level_max, _p(st), len(names), srcp, _p(dims), _p(vec), n, _p(new_blocks), dstp) != n:
        _l.check(-1, "amr_regrid_changed")
    return new_blocks.astype(np.int64), dict(zip(names, dst))


def circle_band_grid(lfine, radius=0.25, width=0.06):
    """A three-level grid whose finest level (2^lfine blocks per side) is a band around a circle: the shape of
    BASELINE.json configs[4] (flow past a cylinder, finest level ~4096^2-equivalent at lfine = 9).  Built with the library's
    host routines (state validation with 2:1 balance, regrid), the way a run arrives at it."""
    l0 = lfine - 2
    blocks = np.array([(l0, i, j) for j in range(1 << l0) for i in range(1 << l0)], dtype=np.int64)
    for lvl in range(l0, lfine):
        cx = (blocks[:, 1] + 0.5) / (1 << blocks[:, 0]) - 0.5
        cy = (blocks[:, 2] + 0.5) / (1 << blocks[:, 0]) - 0.5
        d = np.abs(np.hypot(cx, cy) - radius)
        st = np.where((blocks[:, 0] == lvl) & (d < width), REFINE, LEAVE).astype(np.int32)
        st = validate_states(blocks, st, lfine + 1)
        blocks, _ = regrid(blocks, st, {}, lfine + 1)
    return AmrBlockGrid(blocks)
