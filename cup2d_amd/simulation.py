"""Host-side mirror of the reference's time loop (main.cpp:6576-7290, body-free path) on top of
the C-ABI.  Names follow the reference: fields tmp/chi/vel/vold/pres/pold/tmpV, computeA-style
block operators, sim.dt/sim.nu/sim.CFL.  Every method is one or a few C-ABI calls; no arithmetic
on field data happens in Python.
"""
import ctypes

import numpy as np

from . import lib as _l
from .grid import BlockGrid


class BodyOps:
    """Penalisation with host-supplied bodies (main.cpp:6643-7006; include/cup2d_hip.h cup2d_body_*): shared by the
    uniform and the block-AMR host mirrors (the kernels take a block's cell size from the context)."""

    def body_set(self, body, blocks, origin, chi, udef, centre):
        """blocks: ascending block indices the shape touches; origin (n, 2) = Info::origin; chi (n, 64); udef (n, 64, 2)"""
        vp = ctypes.c_void_p
        b = np.ascontiguousarray(blocks, dtype=np.int32)
        o = np.ascontiguousarray(origin, dtype=np.float64).reshape(len(b), 2)
        c = np.ascontiguousarray(chi, dtype=np.float64).reshape(len(b), 64)
        u = np.ascontiguousarray(udef, dtype=np.float64).reshape(len(b), 128)
        _l.check(self.L.cup2d_body_set(self._ctx, int(body), len(b), b.ctypes.data_as(vp), o.ctypes.data_as(vp), c.ctypes.data_as(vp),
                                       u.ctypes.data_as(vp), float(centre[0]), float(centre[1])), "body_set")

    def body_clear(self):
        _l.check(self.L.cup2d_body_clear(self._ctx), "body_clear")

    def body_momentum(self, body, lam, dt):
        """main.cpp:6643-6702: returns ((u, v, omega), the seven moments PM, PJ, PX, PY, UM, VM, AM)"""
        uvw, q = np.zeros(3), np.zeros(7)
        vp = ctypes.c_void_p
        _l.check(self.L.cup2d_body_momentum(self._ctx, int(body), float(lam), float(dt), uvw.ctypes.data_as(vp), q.ctypes.data_as(vp)),
                 "body_momentum")
        return uvw, q

    def penalize(self, lam, dt, uvw):
        """main.cpp:6944-7006: velocity blend + tmpV = u_def of the dominating bodies; uvw (nbodies, 3)"""
        a = np.ascontiguousarray(uvw, dtype=np.float64).reshape(-1, 3)
        _l.check(self.L.cup2d_penalize(self._ctx, float(lam), float(dt), a.ctypes.data_as(ctypes.c_void_p)), "penalize")


class Simulation(BodyOps):
    def __init__(self, nbx, nby=None, extent=1.0, nu=1e-3, cfl=0.5, order="hilbert", device=0, grid=None, h=None):
        """Uniform grid of nbx x nby blocks of 8x8 cells.  h = extent / max(nbx, nby) / 8 as
        main.cpp:6338 (sim.h0 at the level of the blocks)."""
        self.L = _l.load_library()
        self.grid = grid if grid is not None else BlockGrid(nbx, nby if nby is not None else nbx, order=order)
        g = self.grid
        self.h = float(h) if h is not None else float(extent) / max(g.nbx, g.nby) / 8
        self.nu, self.cfl = float(nu), float(cfl)
        self.time, self.step_count, self.dt = 0.0, 0, 0.0
        # sim.PoissonTol / PoissonTolRel / maxPoissonRestarts (command-line options without defaults, main.cpp:6333-6335;
        # these are the values the reference's run.sh passes): used from the eleventh step on (main.cpp:7028-7030)
        self.poisson_tol, self.poisson_tol_rel, self.max_poisson_restarts = 1e-3, 1e-2, 0
        self._ctx = ctypes.c_void_p()
        _l.check(self.L.cup2d_create(ctypes.byref(self._ctx), g.nblocks, g.nghost, g.n_inner,
                                     g.nbr.ctypes.data_as(ctypes.c_void_p), self.h, int(device)), "cup2d_create")

    # ---- lifetime ----------------------------------------------------------------------------
    def close(self):
        if self._ctx:
            self.L.cup2d_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def ctx(self):
        return self._ctx

    def set_math(self, strict):
        _l.check(self.L.cup2d_set_math(self._ctx, _l.MATH_STRICT if strict else _l.MATH_FAST), "set_math")

    def set_stream(self, raw_stream):
        _l.check(self.L.cup2d_set_stream(self._ctx, ctypes.c_void_p(raw_stream)), "set_stream")

    def synchronize(self):
        _l.check(self.L.cup2d_synchronize(self._ctx), "synchronize")

    def field_ptr(self, field):
        p = ctypes.c_void_p()
        _l.check(self.L.cup2d_field_ptr(self._ctx, field, ctypes.byref(p)), "field_ptr")
        return p.value

    # ---- data movement (global row-major arrays <-> device slabs) --------------------------------
    def set_field(self, field, a):
        slab = self.grid.to_blocks(a)
        assert slab.shape[1] == 64 * _l.FIELD_DIM[field]
        _l.check(self.L.cup2d_upload_slab(self._ctx, field, slab.ctypes.data_as(ctypes.c_void_p)), "upload_slab")

    def get_field(self, field):
        dim = _l.FIELD_DIM[field]
        slab = np.empty((self.grid.nblocks, 64 * dim))
        _l.check(self.L.cup2d_download_slab(self._ctx, field, slab.ctypes.data_as(ctypes.c_void_p)), "download_slab")
        return self.grid.from_blocks(slab, dim)

    def set_blocks(self, field, block_arrays):
        """reference-style upload: one separately allocated array per block (Info::block)"""
        ptrs = (ctypes.c_void_p * len(block_arrays))(*[b.ctypes.data for b in block_arrays])
        _l.check(self.L.cup2d_upload(self._ctx, field, ptrs), "upload")

    def get_blocks(self, field):
        dim = _l.FIELD_DIM[field]
        blocks = [np.empty(64 * dim) for _ in range(self.grid.nblocks)]
        ptrs = (ctypes.c_void_p * len(blocks))(*[b.ctypes.data for b in blocks])
        _l.check(self.L.cup2d_download(self._ctx, field, ptrs), "download")
        return blocks

    def fill(self, field, value=0.0):
        _l.check(self.L.cup2d_fill(self._ctx, field, float(value)), "fill")

    vel = property(lambda s: s.get_field(_l.VEL), lambda s, a: s.set_field(_l.VEL, a))
    pres = property(lambda s: s.get_field(_l.PRES), lambda s, a: s.set_field(_l.PRES, a))
    pold = property(lambda s: s.get_field(_l.POLD), lambda s, a: s.set_field(_l.POLD, a))
    tmp = property(lambda s: s.get_field(_l.TMP), lambda s, a: s.set_field(_l.TMP, a))
    tmpV = property(lambda s: s.get_field(_l.TMPV), lambda s, a: s.set_field(_l.TMPV, a))
    chi = property(lambda s: s.get_field(_l.CHI), lambda s, a: s.set_field(_l.CHI, a))

    # ---- block operators (names of the reference functors) -------------------------------------
    def advect_diffuse_rhs(self, dt, phase=_l.BLOCKS_ALL):
        """computeA<VectorLab>(KernelAdvectDiffuse(), var.vel, 2): tmpV <- rhs (main.cpp:6616)"""
        _l.check(self.L.cup2d_advect_diffuse_rhs(self._ctx, self.nu, float(dt), phase), "advect_diffuse_rhs")

    def advect_diffuse_rk2(self, dt):
        """main.cpp:6607-6642"""
        _l.check(self.L.cup2d_advect_diffuse_rk2(self._ctx, self.nu, float(dt)), "advect_diffuse_rk2")

    def vorticity(self):
        """computeA<VectorLab>(KernelVorticity(), var.vel, 2): tmp <- curl vel (main.cpp:4659)"""
        _l.check(self.L.cup2d_vorticity(self._ctx, _l.BLOCKS_ALL), "vorticity")

    def pressure_rhs(self, dt, use_bodies=False):
        """computeB<pressure_rhs,...>(.., var.vel, var.tmpV) (main.cpp:7011)"""
        _l.check(self.L.cup2d_pressure_rhs(self._ctx, float(dt), int(use_bodies), _l.BLOCKS_ALL), "pressure_rhs")

    def laplacian_sub(self):
        """computeA<ScalarLab>(pressure_rhs1(), var.pold, 1) (main.cpp:7026)"""
        _l.check(self.L.cup2d_laplacian_sub(self._ctx, _l.BLOCKS_ALL), "laplacian_sub")

    def poisson_rhs(self, dt, use_bodies=False):
        """main.cpp:7007-7026"""
        _l.check(self.L.cup2d_poisson_rhs(self._ctx, float(dt), int(use_bodies)), "poisson_rhs")

    def pressure_correction(self, dt):
        """computeA<ScalarLab>(pressureCorrectionKernel(), var.pres, 1) (main.cpp:7178)"""
        _l.check(self.L.cup2d_pressure_correction(self._ctx, float(dt), _l.BLOCKS_ALL), "pressure_correction")

    def add_correction(self):
        """main.cpp:7180-7187"""
        _l.check(self.L.cup2d_add_correction(self._ctx), "add_correction")

    def project(self, dt):
        """main.cpp:7120-7187"""
        _l.check(self.L.cup2d_project(self._ctx, float(dt)), "project")

    def apply_A(self, dst, src):
        _l.check(self.L.cup2d_apply_A(self._ctx, dst, src), "apply_A")

    def precond(self, dst, src):
        _l.check(self.L.cup2d_precond(self._ctx, dst, src), "precond")

    def P_inv(self):
        P = np.empty((64, 64))
        _l.check(self.L.cup2d_get_P_inv(self._ctx, P.ctypes.data_as(ctypes.c_void_p)), "get_P_inv")
        return P

    def set_P_inv(self, P):
        """the P_inv argument of LocalSpMatDnVec's constructor (cuda.h:28-29)"""
        P = np.ascontiguousarray(P, dtype=np.float64).reshape(64, 64)
        _l.check(self.L.cup2d_set_P_inv(self._ctx, P.ctypes.data_as(ctypes.c_void_p)), "set_P_inv")

    def set_precond(self, kind):
        _l.check(self.L.cup2d_set_precond(self._ctx, int(kind)), "set_precond")

    def set_solver(self, fused=False, finish_in_kernel=False, form=None):
        """organisation of a BiCGSTAB iteration (include/cup2d_hip.h cup2d_solver_kind); form: 'auto' | 'full' | 'edge' | 'eab'
        (cup2d_fused_form), None leaves it as it is"""
        _l.check(self.L.cup2d_set_solver(self._ctx, _l.SOLVER_FUSED if fused else _l.SOLVER_SWEEPS, int(finish_in_kernel)),
                 "set_solver")
        if form is not None:
            _l.check(self.L.cup2d_set_solver_form(self._ctx, ("auto", "full", "edge", "eab").index(form)), "set_solver_form")

    def keep_last_iterate(self, on=True):
        """diagnostic (cup2d_solver_keep_last): the next solves also keep their LAST iterate"""
        _l.check(self.L.cup2d_solver_keep_last(self._ctx, int(on)), "solver_keep_last")

    def last_iterate_to(self, field):
        """the last iterate of the previous solve -> a scalar field; returns max|r| of the recurrence after the last
        iteration (cup2d_solver_last_iterate)"""
        e = ctypes.c_double()
        _l.check(self.L.cup2d_solver_last_iterate(self._ctx, int(field), ctypes.byref(e)), "solver_last_iterate")
        return e.value

    def last_solver(self):
        """'fused' or 'sweeps': what the last poisson_solve ran"""
        k = ctypes.c_int()
        _l.check(self.L.cup2d_get_last_solver(self._ctx, ctypes.byref(k)), "get_last_solver")
        return "fused" if k.value == _l.SOLVER_FUSED else "sweeps"

    def set_nrank_organisation(self, deferred=-1, split=-1):
        """cup2d_set_nrank_organisation: how a reduction point of the N-rank solver is organised (-1 = the default)"""
        _l.check(self.L.cup2d_set_nrank_organisation(self._ctx, int(deferred), int(split)), "set_nrank_organisation")

    def last_solver_form(self):
        """(form, merge, handover mask) of the last fused solve (cup2d_get_last_solver_form): form 'full' | 'edge' | 'eab'"""
        f, m, h = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _l.check(self.L.cup2d_get_last_solver_form(self._ctx, ctypes.byref(f), ctypes.byref(m), ctypes.byref(h)), "get_last_solver_form")
        return ("none", "full", "edge", "eab")[f.value], m.value, h.value

    def placement(self):
        """the placement search of the solver's vectors (cup2d_get_placement): sets timed, us per iteration kept / slowest / first"""
        n, a, b, f = ctypes.c_int(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _l.check(self.L.cup2d_get_placement(self._ctx, ctypes.byref(n), ctypes.byref(a), ctypes.byref(b), ctypes.byref(f)), "get_placement")
        return dict(candidates=n.value, kept_us=a.value, slowest_us=b.value, first_us=f.value)

    def set_matrix_coo(self, row, col, val, halo=0):
        """Assembled Poisson operator (what main.cpp:7034-7112 pushes into LocalSpMatDnVec), local
        int32 indices in device block order; poisson_solve / apply_A use it instead of the stencil."""
        row = np.ascontiguousarray(row, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        assert row.shape == col.shape == val.shape and row.ndim == 1
        _l.check(self.L.cup2d_set_matrix_coo(self._ctx, int(halo), row.size, row.ctypes.data_as(ctypes.c_void_p),
                                             col.ctypes.data_as(ctypes.c_void_p), val.ctypes.data_as(ctypes.c_void_p)),
                 "set_matrix_coo")

    def clear_matrix(self):
        _l.check(self.L.cup2d_clear_matrix(self._ctx), "clear_matrix")

    def max_abs_vel(self):
        v = ctypes.c_double()
        _l.check(self.L.cup2d_max_abs_vel(self._ctx, ctypes.byref(v)), "max_abs_vel")
        return v.value

    def compute_dt(self):
        """main.cpp:6579-6595"""
        v = ctypes.c_double()
        _l.check(self.L.cup2d_compute_dt(self._ctx, self.nu, self.cfl, ctypes.byref(v)), "compute_dt")
        return v.value

    def poisson_solve(self, tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=1000):
        """sim.mat->solveWithUpdate/NoUpdate (main.cpp:7115-7118): b = tmp, x0 = pres, x -> pres"""
        it, rs = ctypes.c_int(), ctypes.c_int()
        e, e0 = ctypes.c_double(), ctypes.c_double()
        _l.check(self.L.cup2d_poisson_solve(self._ctx, float(tol), float(rel_tol), int(max_restarts), int(max_iter),
                                            ctypes.byref(it), ctypes.byref(rs), ctypes.byref(e), ctypes.byref(e0)),
                 "poisson_solve")
        return dict(iters=it.value, restarts=rs.value, err=e.value, err_init=e0.value)

    def block_linf(self, field=_l.TMP):
        """max |field| per block, in block order (cup2d_block_linf)"""
        out = np.empty(self.grid.nblocks)
        _l.check(self.L.cup2d_block_linf(self._ctx, int(field), out.ctypes.data_as(ctypes.c_void_p)), "block_linf")
        return out

    def jacobi_sweeps(self, nsweeps, omega=0.8):
        """nsweeps weighted-Jacobi sweeps on A pres = tmp (cup2d_jacobi_sweeps); returns max|tmp - A pres| of the
        iterate before the last sweep (left in pold)"""
        e = ctypes.c_double()
        _l.check(self.L.cup2d_jacobi_sweeps(self._ctx, float(omega), int(nsweeps), ctypes.byref(e)), "jacobi_sweeps")
        return e.value

    def poisson_residual(self):
        """pold = tmp - A pres; returns its max norm (cup2d_poisson_residual)"""
        e = ctypes.c_double()
        _l.check(self.L.cup2d_poisson_residual(self._ctx, ctypes.byref(e)), "poisson_residual")
        return e.value

    def step(self, tol=None, rel_tol=None, max_restarts=None, max_iter=1000):
        """One pass of the time-loop body.  Without explicit tolerances the reference's rule applies
        (main.cpp:7028-7030): steps 0..9 solve with zero tolerances and up to 100 restarts, later steps with
        poisson_tol / poisson_tol_rel / max_poisson_restarts.  Explicit arguments override it (max_restarts then
        defaults to the reference's 0).  A vanishing dt advances nothing (main.cpp:6596)."""
        if tol is None:
            early = self.step_count < 10
            tol = 0.0 if early else self.poisson_tol
            rel_tol = 0.0 if early else self.poisson_tol_rel
            max_restarts = 100 if early else self.max_poisson_restarts
        dt, it, e = ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
        _l.check(self.L.cup2d_step(self._ctx, self.nu, self.cfl, float(tol), float(rel_tol or 0.0),
                                   int(0 if max_restarts is None else max_restarts), int(max_iter),
                                   ctypes.byref(dt), ctypes.byref(it), ctypes.byref(e)), "step")
        self.dt = dt.value
        if self.dt > 2e-16:
            self.time += self.dt
            self.step_count += 1
        return dict(dt=dt.value, iters=it.value, err=e.value)

    # ---- output ----------------------------------------------------------------------------------
    def dump(self, path, time=None, level=0):
        """the reference's dump() (main.cpp:3367-3466, called at 6601 with "vel.%08d"): velocity in the
        XDMF2 + raw float32 layout post.py reads.  level: refinement level the grid stands for (h0 = h * 2^level)."""
        from . import dump as _d
        slab = np.empty((self.grid.nblocks, 128))
        _l.check(self.L.cup2d_download_slab(self._ctx, _l.VEL, slab.ctypes.data_as(ctypes.c_void_p)), "download_slab")
        _d.dump(path, self.time if time is None else time, self.grid, slab, self.h * (1 << level), level)

    # ---- instrumentation -----------------------------------------------------------------------
    def set_timing(self, on=True):
        """False/0 off, True/1 every launch, 2 sampled, 3 sampled with the launches outside the solver in every step (include/cup2d_hip.h)"""
        _l.check(self.L.cup2d_set_timing(self._ctx, int(on)), "set_timing")

    def debug_walk_knockout(self, knockout):
        """timing aid (include/cup2d_hip.h): 1 / 2 = the quad WENO5 stage without its arithmetic / without its memory traffic; wrong results"""
        _l.check(self.L.cup2d_debug_walk_knockout(self._ctx, int(knockout)), "debug_walk_knockout")

    def get_timing(self, timer):
        ms, n = ctypes.c_double(), ctypes.c_int()
        _l.check(self.L.cup2d_get_timing(self._ctx, timer, ctypes.byref(ms), ctypes.byref(n)), "get_timing")
        return ms.value, n.value
