"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It wraps

  * oracle/liboracle.so      -- plain-C restatement of the hot path (cup2d_oracle.c)
  * oracle/_ref/ref_harness  -- the reference's own main.cpp, compiled where it lies
                                under /root/reference (ref_harness.cpp), run as a subprocess

Arrays are global row-major float64: scalar (ny, nx), vector (ny, nx, 2).
"""
import ctypes
import os
import subprocess
import tempfile

# The restatement's OpenMP loops are tiny; on a 256-hardware-thread GPU host an unset OMP_NUM_THREADS
# makes every parallel region a 256-thread rendezvous (the solver tests then take minutes, not
# seconds).  libgomp reads this when it is first loaded, so set it before anything imports it.
os.environ.setdefault("OMP_NUM_THREADS", str(min(8, os.cpu_count() or 1)))

import numpy as np  # noqa: E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_HARNESS = os.path.join(_HERE, "_ref", "ref_harness")
# the same reference main.cpp linked against THIS repository's cuda.h implementation
# (cup2d_amd/libcup2d_spmat.so) instead of the CPU restatement: the drop-in test of seam B1 (needs a GPU)
REF_HARNESS_HIP = os.path.join(_HERE, "_ref", "ref_harness_hip")

_lib = None
_dp = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    """(Re)build liboracle.so and, when /root/reference exists, oracle/_ref/ref_harness."""
    if force or not os.path.exists(LIB_PATH) or (
        os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "cup2d_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.oracle_compute_dt.restype = ctypes.c_double
        _lib.oracle_max_abs.restype = ctypes.c_double
        _lib.oracle_step.restype = ctypes.c_double
        _lib.oracle_weno5_plus.restype = ctypes.c_double
        _lib.oracle_weno5_minus.restype = ctypes.c_double
        _lib.oracle_derivative.restype = ctypes.c_double
    return _lib


def _p(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def P_inv():
    P = np.empty((64, 64))
    lib().oracle_P_inv(_p(P))
    return P


def advect_diffuse_rhs(vel, h, nu, dt):
    vel = _c(vel)
    ny, nx, _ = vel.shape
    out = np.empty_like(vel)
    lib().oracle_advect_diffuse_rhs(nx, ny, ctypes.c_double(h), ctypes.c_double(nu), ctypes.c_double(dt), _p(vel), _p(out))
    return out


def rk2_advect_diffuse(vel, h, nu, dt):
    """returns (vel_new, stage1)"""
    v = _c(vel).copy()
    ny, nx, _ = v.shape
    s1 = np.empty_like(v)
    lib().oracle_rk2_advect_diffuse(nx, ny, ctypes.c_double(h), ctypes.c_double(nu), ctypes.c_double(dt), _p(v), _p(s1))
    return v, s1


def vorticity(vel, h):
    vel = _c(vel)
    ny, nx, _ = vel.shape
    out = np.empty((ny, nx))
    lib().oracle_vorticity(nx, ny, ctypes.c_double(h), _p(vel), _p(out))
    return out


def pressure_rhs(vel, h, dt, udef=None, chi=None):
    vel = _c(vel)
    ny, nx, _ = vel.shape
    out = np.empty((ny, nx))
    ud = _c(udef) if udef is not None else None
    ch = _c(chi) if chi is not None else None
    lib().oracle_pressure_rhs(nx, ny, ctypes.c_double(h), ctypes.c_double(dt), _p(vel),
                              _p(ud) if ud is not None else None, _p(ch) if ch is not None else None, _p(out))
    return out


def laplacian_sub(p, tmp):
    """returns tmp - Lap5(p) (Neumann ghosts)"""
    p = _c(p)
    t = _c(tmp).copy()
    ny, nx = p.shape
    lib().oracle_laplacian_sub(nx, ny, _p(p), _p(t))
    return t


def apply_A(x):
    x = _c(x)
    ny, nx = x.shape
    y = np.empty_like(x)
    lib().oracle_apply_A(nx, ny, _p(x), _p(y))
    return y


def poisson_diag(ny, nx):
    """diagonal of the assembled Poisson matrix on a same-level grid (main.cpp:7034-7112): -(number of neighbours)"""
    d = np.full((ny, nx), -4.0)
    d[0, :] += 1
    d[-1, :] += 1
    d[:, 0] += 1
    d[:, -1] += 1
    return d


def poisson_residual(x, b):
    """r = b - A x and its max norm (the reference's stopping norm, cuda.cu:303-311)"""
    r = _c(b) - apply_A(x)
    return r, float(np.abs(r).max())


def jacobi_sweeps(x, b, omega, nsweeps):
    """nsweeps times x <- x + (b - A x) * (omega / diag(A)) (the benchmark smoother; the reference
    has none).  Returns (x, max|b - A x| of the iterate before the last sweep)."""
    x, b = _c(x).copy(), _c(b)
    c = omega / poisson_diag(*x.shape)
    linf = 0.0
    for _ in range(nsweeps):
        r = b - apply_A(x)
        linf = float(np.abs(r).max())
        x = x + r * c
    return x, linf


def precond(x, P=None):
    x = _c(x)
    ny, nx = x.shape
    P = P_inv() if P is None else _c(P)
    y = np.empty_like(x)
    lib().oracle_precond(nx, ny, _p(P), _p(x), _p(y))
    return y


def pressure_correction(pres, h, dt):
    pres = _c(pres)
    ny, nx = pres.shape
    out = np.empty((ny, nx, 2))
    lib().oracle_pressure_correction(nx, ny, ctypes.c_double(h), ctypes.c_double(dt), _p(pres), _p(out))
    return out


def add_scaled(vel, tmpV, h):
    v = _c(vel).copy()
    ny, nx, _ = v.shape
    lib().oracle_add_scaled(nx, ny, ctypes.c_double(h), _p(_c(tmpV)), _p(v))
    return v


def compute_dt(h, nu, cfl, umax):
    return lib().oracle_compute_dt(ctypes.c_double(h), ctypes.c_double(nu), ctypes.c_double(cfl), ctypes.c_double(umax))


def bicgstab(b, x0=None, tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=1000, P=None):
    """returns (x_opt, dict(iters, restarts, err, err_init))"""
    b = _c(b)
    ny, nx = b.shape
    x = np.zeros_like(b) if x0 is None else _c(x0).copy()
    P = P_inv() if P is None else _c(P)
    info = np.zeros(4)
    lib().oracle_bicgstab(nx, ny, _p(P), _p(b), _p(x), ctypes.c_double(tol), ctypes.c_double(rel_tol),
                          int(max_restarts), int(max_iter), _p(info))
    return x, dict(iters=int(info[0]), restarts=int(info[1]), err=info[2], err_init=info[3])


def pressure_update(x, pold, h):
    x = _c(x)
    ny, nx = x.shape
    out = np.empty_like(x)
    lib().oracle_pressure_update(nx, ny, ctypes.c_double(h), _p(x), _p(_c(pold)), _p(out))
    return out


def step(vel, pres, h, nu, cfl, tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=1000, P=None):
    """one body-free time step (main.cpp:6576-7187). returns (vel, pres, dt, info)"""
    v = _c(vel).copy()
    p = _c(pres).copy()
    ny, nx = p.shape
    P = P_inv() if P is None else _c(P)
    info = np.zeros(4)
    dt = lib().oracle_step(nx, ny, ctypes.c_double(h), ctypes.c_double(nu), ctypes.c_double(cfl), _p(P), _p(v), _p(p),
                           ctypes.c_double(tol), ctypes.c_double(rel_tol), int(max_restarts), int(max_iter), _p(info))
    return v, p, dt, dict(iters=int(info[0]), restarts=int(info[1]), err=info[2], err_init=info[3])


# ----------------------------------------------------------------------------------------------
# synthetic inputs shared by oracle, tests and bench
# ----------------------------------------------------------------------------------------------
def taylor_green(n, noise=1e-3, seed=20250117, ny=None):
    """u = sin(2 pi x) cos(2 pi y), v = -cos(2 pi x) sin(2 pi y) at cell centres of the unit
    square (h = 1/n) plus uniform noise; returns (ny, nx, 2) float64."""
    nx = n
    ny = n if ny is None else ny
    h = 1.0 / max(nx, ny)
    x = (np.arange(nx) + 0.5) * h
    y = (np.arange(ny) + 0.5) * h
    X, Y = np.meshgrid(x, y, indexing="xy")
    rng = np.random.default_rng(seed)
    u = np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y)
    v = -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y)
    vel = np.stack([u, v], axis=-1)
    if noise:
        vel = vel + noise * rng.uniform(-1.0, 1.0, vel.shape)
    return np.ascontiguousarray(vel)


# ----------------------------------------------------------------------------------------------
# the reference itself (oracle/_ref/ref_harness)
# ----------------------------------------------------------------------------------------------
def have_reference():
    return os.path.exists(REF_HARNESS) and os.access(REF_HARNESS, os.X_OK)


def _level(n):
    lv = int(round(np.log2(n // 8)))
    assert 8 << lv == n, "reference harness grids are 8*2^k square"
    return lv


def have_reference_hip():
    return os.path.exists(REF_HARNESS_HIP) and os.access(REF_HARNESS_HIP, os.X_OK)


# the reference's main.cpp with its block-operator call sites replaced by this repository's C ABI (oracle/b2_patch.py):
# seam B2 compiled (needs a GPU)
REF_HARNESS_B2 = os.path.join(_HERE, "_ref", "ref_harness_b2")


def have_reference_b2():
    return os.path.exists(REF_HARNESS_B2) and os.access(REF_HARNESS_B2, os.X_OK)


def _base_grid(nx, ny):
    """(levelStart, bpdx, bpdy) of an nx x ny-cell uniform grid in the reference's terms: bpdx x bpdy base blocks refined
    levelStart times, the shorter side one base block (main.cpp:6338: h = extent / max(bpdx, bpdy) / 8 / 2^level)"""
    n = min(nx, ny)
    assert nx % n == 0 and ny % n == 0, "reference harness grids are (bpdx n) x (bpdy n), n = 8*2^k"
    return _level(n), nx // n, ny // n


def _run_ref(mode, n, d, _threads=None, _timeout=None, _hip=False, _env=None, _b2=False, **kw):
    """n: cells per side, or (nx, ny) for a rectangle of bpdx x bpdy base blocks"""
    if isinstance(n, tuple):
        lv, bx, by = _base_grid(*n)
        if (bx, by) != (1, 1):
            kw = dict(kw, bpdx=bx, bpdy=by)
    else:
        lv = _level(n)
    cmd = [REF_HARNESS_B2 if _b2 else (REF_HARNESS_HIP if _hip else REF_HARNESS), mode, str(lv), d] + ["%s=%.17g" % (k, v) if isinstance(v, float) else "%s=%s" % (k, v) for k, v in kw.items()]
    env = dict(os.environ)
    if _threads:
        env.update(OMP_NUM_THREADS=str(int(_threads)), OMP_PROC_BIND="close", OMP_PLACES="cores")
    if _env:
        env.update(_env)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=_timeout)
    if r.returncode != 0:
        raise RuntimeError("ref_harness failed (%d): %s" % (r.returncode, r.stderr.decode()[-2000:]))
    return r.stdout.decode()


def ref_functors(vel, nu, dt=None, pres=None, chi=None, udef=None, nomatrix=False):
    """Run every hot-path block functor of the reference on the given fields.
    Returns dict of arrays (see ref_harness.cpp 'functors').  nomatrix: the harness drops the Poisson triplets the
    reference's start-up assembles (the functors never read them; at 4096^2 that is most of the run time)."""
    vel = _c(vel)
    ny, n = vel.shape[:2]   # (ny, nx, 2): a square, or a rectangle of whole base blocks (keys bpdx, bpdy of the harness)
    assert vel.shape == (ny, n, 2)
    with tempfile.TemporaryDirectory() as d:
        vel.tofile(os.path.join(d, "vel.in"))
        if pres is not None:
            _c(pres).tofile(os.path.join(d, "pres.in"))
        if chi is not None:
            _c(chi).tofile(os.path.join(d, "chi.in"))
        if udef is not None:
            _c(udef).tofile(os.path.join(d, "udef.in"))
        kw = dict(nu=float(nu))
        if dt is not None:
            kw["dt"] = float(dt)
        if nomatrix:
            kw["nomatrix"] = 1
        _run_ref("functors", (n, ny), d, **kw)
        out = {}
        for name, dim in [("advdiff_rhs", 2), ("rk2_stage1", 2), ("rk2_vel", 2), ("vorticity", 1), ("pressure_rhs", 1),
                          ("poisson_b", 1), ("pgrad_tmpV", 2), ("projected_vel", 2)]:
            a = np.fromfile(os.path.join(d, name + ".out"))
            out[name] = a.reshape(ny, n, 2) if dim == 2 else a.reshape(ny, n)
        s = np.fromfile(os.path.join(d, "scalars.out"))
        out["dt_ref"], out["umax"], out["h"], out["dt"] = s
        out["block_order"] = np.fromfile(os.path.join(d, "block_order.out")).reshape(-1, 2).astype(np.int64)
    return out


def ref_run_amr(level_start, level_max, steps, rtol, ctol, nu=1e-3, cfl=0.5, max_iter=None, hip=False, env=None):
    """The reference's time loop with refinement on (ref_harness 'amr': config 5 through the reference's own
    adapt(), coarse-fine labs, flux correction and matrix assembly) from an analytic vortex pair.
    Returns dict(blocks=(nb,3) int [level, i, j], vel=(nb,64,2), pres=(nb,64), steps=[...]) sorted by
    (level, j, i).  hip=True: every linear solve goes through libcup2d_spmat.so on the GPU."""
    with tempfile.TemporaryDirectory() as d:
        kw = dict(levelmax=int(level_max), rtol=float(rtol), ctol=float(ctol), steps=int(steps), nu=float(nu), cfl=float(cfl))
        if max_iter is not None:
            kw["maxiter"] = int(max_iter)
        cmd_n = 8 << int(level_start)
        _run_ref("amr", cmd_n, d, _hip=hip, _env=env, **kw)
        a = np.fromfile(os.path.join(d, "blocks.final")).reshape(-1, 3 + 128 + 64)
        meta = open(os.path.join(d, "meta.txt")).read().strip().split("\n")
    key = a[:, 0] * 1e12 + a[:, 2] * 1e6 + a[:, 1]
    a = a[np.argsort(key)]
    out = dict(blocks=a[:, :3].astype(np.int64), vel=a[:, 3:131].reshape(-1, 64, 2), pres=a[:, 131:], steps=[])
    for line in meta:
        t = line.split()
        if t[0] == "step":
            out["steps"].append(dict(blocks=int(t[3]), lmin=int(t[5]), lmax=int(t[6]), dt=float(t[8]), update=int(t[10]),
                                     prev_iters=int(t[12])))
    return out


def ref_amr_functors(level_start, level_max, steps, rtol, ctol, nu=1e-3, max_iter=50):
    """Block functors of the reference on an ADAPTED grid (ref_harness 'amr' with reps=-1): the grid the vortex-pair
    run has after `steps` steps, analytic fields sampled at the cell centres, pressure_rhs1 with its flux correction
    (prepare0 / computeA / fillcases, main.cpp:7022-7027) and KernelVorticity.  Per block, sorted by (level, j, i):
    blocks (nb,3), pold, tmp_in, tmp_out (nb,64), vel (nb,64,2), vort (nb,64); chi, udef, prhs = pressure_rhs with its
    flux correction (main.cpp:7007-7013); pres, pcorr = pressureCorrectionKernel's tmpV (7178); Ax = the reference-assembled
    Poisson matrix (main.cpp:7034-7113) applied to pres; lab3 = the 14x14x2 ghosted tile of vel that
    KernelAdvectDiffuse sees, advdiff = its output with the flux correction (main.cpp:6611-6617); nu, dt, h0."""
    with tempfile.TemporaryDirectory() as d:
        _run_ref("amr", 8 << int(level_start), d, levelmax=int(level_max), rtol=float(rtol), ctol=float(ctol), steps=int(steps),
                 nu=float(nu), maxiter=int(max_iter), reps=-1)
        a = np.fromfile(os.path.join(d, "blocks.functors")).reshape(-1, 3 + 192 + 128 + 64 + 64 + 128 + 64 + 64 + 128 + 64 + 392 + 128 + 100 + 200)
        dt, h0 = np.fromfile(os.path.join(d, "functors_scalars"))
    a = a[np.argsort(a[:, 0] * 1e12 + a[:, 2] * 1e6 + a[:, 1])]
    o = 387
    return dict(blocks=a[:, :3].astype(np.int64), pold=a[:, 3:67], tmp_in=a[:, 67:131], tmp_out=a[:, 131:195],
                vel=a[:, 195:323].reshape(-1, 64, 2), vort=a[:, 323:387], chi=a[:, o:o + 64],
                udef=a[:, o + 64:o + 192].reshape(-1, 64, 2), prhs=a[:, o + 192:o + 256], pres=a[:, o + 256:o + 320],
                pcorr=a[:, o + 320:o + 448].reshape(-1, 64, 2), Ax=a[:, o + 448:o + 512], lab3=a[:, o + 512:o + 904].reshape(-1, 14, 14, 2),
                advdiff=a[:, o + 904:o + 1032].reshape(-1, 64, 2), lab1t_pres=a[:, o + 1032:o + 1132].reshape(-1, 10, 10),
                lab1t_vel=a[:, o + 1132:o + 1332].reshape(-1, 10, 10, 2), nu=float(nu), dt=float(dt), h0=float(h0))


def ref_amr_adapt(level_start, level_max, steps, rtol, ctol, nu=1e-3, max_iter=50):
    """The reference's own adapt() (main.cpp:4657-5440) on an adapted grid with analytic fields (ref_harness 'amr' with
    reps=-2): returns (pre, post), each dict(blocks (nb,3), chi, pres, pold (nb,64), vel, vold (nb,64,2)) sorted by
    (level, j, i)."""
    def rd(path):
        a = np.fromfile(path).reshape(-1, 3 + 64 + 128 + 128 + 64 + 64)
        a = a[np.argsort(a[:, 0] * 1e12 + a[:, 2] * 1e6 + a[:, 1])]
        return dict(blocks=a[:, :3].astype(np.int64), chi=a[:, 3:67], vel=a[:, 67:195].reshape(-1, 64, 2),
                    vold=a[:, 195:323].reshape(-1, 64, 2), pres=a[:, 323:387], pold=a[:, 387:451])
    with tempfile.TemporaryDirectory() as d:
        _run_ref("amr", 8 << int(level_start), d, levelmax=int(level_max), rtol=float(rtol), ctol=float(ctol), steps=int(steps),
                 nu=float(nu), maxiter=int(max_iter), reps=-2)
        return rd(os.path.join(d, "blocks.pre")), rd(os.path.join(d, "blocks.post"))


def ref_dump(vel, time=0.0):
    """The reference's dump() (main.cpp:3367-3466) on a velocity field: returns the bytes of the three
    files it writes, {'xyz': ..., 'attr': ..., 'xdmf2': ...}."""
    vel = _c(vel)
    n = vel.shape[0]
    with tempfile.TemporaryDirectory() as d:
        vel.tofile(os.path.join(d, "vel.in"))
        _run_ref("dump", n, d, dt=float(time))
        return {k: open(os.path.join(d, "vel." + ext), "rb").read()
                for k, ext in (("xyz", "xyz.raw"), ("attr", "attr.raw"), ("xdmf2", "xdmf2"))}


def ref_solve(b, x0=None, tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=None, hip=False, env=None):
    """BiCGSTAB (CPU port of cuda.cu) on the matrix the reference assembles; also returns A*x0.
    hip=True: the solve goes through libcup2d_spmat.so on the GPU instead (A*x0 is then None)."""
    b = _c(b)
    n = b.shape[0]
    with tempfile.TemporaryDirectory() as d:
        b.tofile(os.path.join(d, "b.in"))
        if x0 is not None:
            _c(x0).tofile(os.path.join(d, "x0.in"))
        kw = dict(tol=float(tol), reltol=float(rel_tol), restarts=int(max_restarts))
        if max_iter is not None:
            kw["maxiter"] = int(max_iter)
        _run_ref("solve", n, d, _hip=hip, _env=env, **kw)
        x = np.fromfile(os.path.join(d, "x.out")).reshape(n, n)
        ax0 = None if hip else np.fromfile(os.path.join(d, "Ax0.out")).reshape(n, n)
        s = np.fromfile(os.path.join(d, "solve_scalars.out"))
    return x, ax0, dict(iters=int(s[0]), err=s[1], err_init=s[2], restarts=int(s[3]), seconds=s[4])


def ref_run(vel0, nu, steps, cfl=0.5, tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=None, keep=None, hip=False, env=None,
            b2=False, shapes=None, threads=None):
    """The reference's own time loop (main.cpp:6576-7290) from the IC vel0 for `steps` steps.
    hip=True: with sim.mat served by libcup2d_spmat.so (GPU) instead of the CPU restatement (seam B1).
    b2=True / b2="rk2,penal,...": the loop of oracle/_ref/ref_harness_b2 -- all (or the named) block-operator call sites
    go through include/cup2d_hip.h on the GPU (seam B2), the others run the reference's own lines; b2="none" is the
    reference loop in that binary.  shapes: the reference's -shapes descriptor (a fish), e.g.
    "angle=0 L=0.4 xpos=0.5 ypos=0.5".  threads: OpenMP threads of the reference's loops (1 = fixed summation order)."""
    vel0 = _c(vel0)
    n = vel0.shape[0]
    d = keep or tempfile.mkdtemp()
    vel0.tofile(os.path.join(d, "vel.in"))
    kw = dict(nu=float(nu), cfl=float(cfl), steps=int(steps), tol=float(tol), reltol=float(rel_tol), restarts=int(max_restarts))
    if max_iter is not None:
        kw["maxiter"] = int(max_iter)
    if shapes:
        kw["shapes"] = str(shapes)
    if b2 and b2 is not True:
        env = dict(env or {}, CUP2D_B2_SITES=str(b2))
    _run_ref("run", n, d, _threads=threads, _hip=hip, _env=env, _b2=bool(b2), **kw)
    out = dict(vel=np.fromfile(os.path.join(d, "vel.final")).reshape(n, n, 2),
               pres=np.fromfile(os.path.join(d, "pres.final")).reshape(n, n), steps=[])
    meta = open(os.path.join(d, "meta.txt")).read().strip().split("\n")
    for line in meta:
        t = line.split()
        if t[0] == "step":
            k = int(t[1])
            out["steps"].append(dict(step=k, dt=float(t[3]), time=float(t[5]),
                                     vel_adv=np.fromfile(os.path.join(d, "vel_adv.%d" % k)).reshape(n, n, 2),
                                     b=np.fromfile(os.path.join(d, "b.%d" % k)).reshape(n, n),
                                     pold=np.fromfile(os.path.join(d, "pold.%d" % k)).reshape(n, n)))
        else:
            out["final"] = dict(iters=int(t[2]), err=float(t[4]), time=float(t[10]), nsteps=int(t[12]))
    if keep is None:
        import shutil
        shutil.rmtree(d)
    return out


def ref_step_time(n, steps=3, max_iter=50, nu=1e-3, vel=None, threads=None, timeout=None):
    """CPU baseline: wall time of one full pass of the reference's own time-loop body
    (main.cpp:6576-7290; Poisson solve = CPU port of cuda.cu capped at max_iter iterations, since the
    reference has no CPU solver).  Returns the harness' JSON dict (median over `steps` steps)."""
    import json
    if vel is None:
        vel = taylor_green(n)
    with tempfile.TemporaryDirectory() as d:
        _c(vel).tofile(os.path.join(d, "vel.in"))
        txt = _run_ref("run", n, d, _threads=threads, _timeout=timeout, nu=float(nu), steps=int(steps) + 1,
                       maxiter=int(max_iter), dump=0)
    return json.loads(txt.strip().split("\n")[-1])


def ref_bench(n, reps=10, vel=None, dt=1e-4, threads=None, nomatrix=False, timeout=None):
    """Time the reference functors (OpenMP) -- CPU baseline, kind='reference'.  nomatrix: the harness drops the Poisson
    triplets the reference's start-up assembles (minutes of serial host work at 4096^2 that no functor reads)."""
    import json
    if vel is None:
        vel = taylor_green(n)
    kw = dict(reps=int(reps), dt=float(dt))
    if nomatrix:
        kw["nomatrix"] = 1
    with tempfile.TemporaryDirectory() as d:
        _c(vel).tofile(os.path.join(d, "vel.in"))
        txt = _run_ref("bench", n, d, _threads=threads, _timeout=timeout, **kw)
    return json.loads(txt.strip().split("\n")[-1])
