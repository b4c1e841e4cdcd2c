"""GPU parity tests (-m gpu): every HIP kernel of the hot path, called through the C-ABI, against the
CPU oracle on the same seeded inputs and against the golden vectors generated from the reference.

Tolerances (stated per the north star; everything is FP64):
  * STRICT arithmetic policy and all 5-point kernels: bit-identical to the reference functors.
  * FAST WENO5 policy: |diff| <= 2e-13 * scale, scale = |afac|*umax^2*... measured as max|rhs|; the only
    differences are one-division weight normalisation and FMA contraction (weno.h).
  * Solver: same iteration count +-2 on the golden 32^2 system at a 1e-10 tolerance, within 25 % on larger
    random systems (BiCGSTAB's count is chaotic in dot-product round-off), |x - x_ref| <= 1e-8, and always
    the reference's own stopping criterion |b - A x|_inf <= tol checked against the oracle's operator.
"""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

from cup2d_amd import lib as L  # noqa: E402


def make_sim(n, nu=1e-3, order="hilbert", strict=True, ny=None):
    import cup2d_amd
    s = cup2d_amd.Simulation(n // 8, (ny or n) // 8, nu=nu, order=order)
    s.set_math(strict)
    return s


@pytest.mark.parametrize("case", ["functors_n32_tg.npz", "functors_n32_noise.npz"])
@pytest.mark.parametrize("order", ["hilbert", "rowmajor"])
def test_functors_vs_golden_strict_bit_exact(gpu_lib, case, order):
    G = golden(case)
    n, nu, dt = int(G["n"]), float(G["nu"]), float(G["dt"])
    with make_sim(n, nu, order, strict=True) as s:
        s.vel = G["vel"]
        assert s.max_abs_vel() == float(G["umax"])
        assert s.compute_dt() == float(G["dt_ref"])
        s.advect_diffuse_rhs(dt)
        assert np.array_equal(s.tmpV, G["advdiff_rhs"])
        s.advect_diffuse_rk2(dt)
        assert np.array_equal(s.vel, G["rk2_vel"])
        s.vorticity()
        assert np.array_equal(s.tmp, G["vorticity"])
        s.tmpV = G["udef"]
        s.chi = G["chi"]
        s.pressure_rhs(dt, use_bodies=True)
        assert np.array_equal(s.tmp, G["pressure_rhs"])
        s.pold = G["pres"]
        s.laplacian_sub()
        assert np.array_equal(s.tmp, G["poisson_b"])
        s.pres = G["pres"]
        s.pressure_correction(dt)
        assert np.array_equal(s.tmpV, G["pgrad_tmpV"])
        s.add_correction()
        assert np.array_equal(s.vel, G["projected_vel"])


@pytest.mark.parametrize("n,noise", [(64, 1e-3), (128, 0.5), (256, 1e-3)])
def test_advect_strict_and_fast_vs_oracle(gpu_lib, oracle, n, noise):
    vel = oracle.taylor_green(n, noise=noise, seed=n)
    h, nu = 1.0 / n, 1e-3
    dt = oracle.compute_dt(h, nu, 0.5, np.abs(vel).max())
    ref = oracle.advect_diffuse_rhs(vel, h, nu, dt)
    ref2, _ = oracle.rk2_advect_diffuse(vel, h, nu, dt)
    with make_sim(n, nu, strict=True) as s:
        s.vel = vel
        s.advect_diffuse_rhs(dt)
        assert np.array_equal(s.tmpV, ref)
        s.advect_diffuse_rk2(dt)
        assert np.array_equal(s.vel, ref2)
        s.set_math(False)
        s.vel = vel
        s.advect_diffuse_rhs(dt)
        fast = s.tmpV
        scale = np.abs(ref).max()
        assert np.abs(fast - ref).max() <= 2e-13 * scale
        s.advect_diffuse_rk2(dt)
        assert np.abs(s.vel - ref2).max() <= 1e-13 * np.abs(ref2).max()


def test_rectangular_grid_and_walls(gpu_lib, oracle):
    nx, ny = 128, 64
    vel = oracle.taylor_green(nx, noise=0.2, seed=5, ny=ny)
    h, nu, dt = 1.0 / nx, 1e-3, 2e-3
    with make_sim(nx, nu, ny=ny) as s:
        assert s.h == h
        s.vel = vel
        s.advect_diffuse_rhs(dt)
        assert np.array_equal(s.tmpV, oracle.advect_diffuse_rhs(vel, h, nu, dt))
        s.vorticity()
        assert np.array_equal(s.tmp, oracle.vorticity(vel, h))


@pytest.mark.parametrize("nbx,nby,order", [(8, 6, "rowmajor"), (5, 3, "rowmajor"), (7, 7, "hilbert"), (2, 2, "hilbert"), (12, 4, "hilbert")])
def test_fast_advect_on_quads_and_leftover_blocks(gpu_lib, oracle, nbx, nby, order):
    """FAST policy = the quad kernel (csrc/advect_walk.h) on every 2 x 2 patch the plan finds + the per-block kernel on
    the blocks without partners (odd block counts), any block order, walls on all sides: the oracle within the FAST
    tolerance, functor and both RK2 stages"""
    nx, ny = 8 * nbx, 8 * nby
    vel = oracle.taylor_green(nx, noise=0.3, seed=nbx * nby, ny=ny)
    h, nu = 1.0 / max(nx, ny), 1e-3
    dt = oracle.compute_dt(h, nu, 0.5, np.abs(vel).max())
    ref = oracle.advect_diffuse_rhs(vel, h, nu, dt)
    ref2, _ = oracle.rk2_advect_diffuse(vel, h, nu, dt)
    with make_sim(nx, nu, order, strict=False, ny=ny) as s:
        s.vel = vel
        s.advect_diffuse_rhs(dt)
        assert np.abs(s.tmpV - ref).max() <= 2e-13 * np.abs(ref).max()
        s.advect_diffuse_rk2(dt)
        assert np.abs(s.vel - ref2).max() <= 1e-13 * np.abs(ref2).max()


def test_block_pointer_upload_matches_slab(gpu_lib, oracle):
    n = 32
    vel = oracle.taylor_green(n, noise=0.1, seed=2)
    with make_sim(n) as s:
        slab = s.grid.to_blocks(vel)
        s.set_blocks(L.VEL, [np.ascontiguousarray(b) for b in slab])
        assert np.array_equal(s.vel, vel)
        blocks = s.get_blocks(L.VEL)
        assert np.array_equal(np.stack(blocks), slab)


def test_poisson_operator_preconditioner(gpu_lib, oracle):
    n = 64
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (n, n))
    with make_sim(n) as s:
        s.pres = x
        s.apply_A(L.TMP, L.PRES)
        assert np.array_equal(s.tmp, oracle.apply_A(x))
        P = s.P_inv()
        assert np.abs(P - oracle.P_inv()).max() < 1e-14
        s.precond(L.TMP, L.PRES)
        assert np.abs(s.tmp - oracle.precond(x, P)).max() < 1e-14


@pytest.mark.parametrize("nx,ny,order", [(64, 64, "hilbert"), (128, 128, "hilbert"), (40, 24, "hilbert"), (24, 40, "rowmajor"),
                                         (16, 8, "rowmajor"), (8, 8, "hilbert"), (136, 72, "hilbert"), (256, 256, "rowmajor")])
def test_jacobi_sweep_and_residual_bit_exact(gpu_lib, oracle, nx, ny, order):
    """the tile kernels of csrc/smoother.hip (weighted-Jacobi sweep, Poisson residual) against the oracle, bit for
    bit: full tiles, a moved last tile (block count not a multiple of 16), fewer than 16 blocks, one block, walls on
    every side, both block orders; pointer roles after an odd and an even number of sweeps"""
    import cup2d_amd
    rng = np.random.default_rng(nx * 1000 + ny)
    x0 = rng.uniform(-1, 1, (ny, nx))
    b = rng.uniform(-1, 1, (ny, nx))
    with cup2d_amd.Simulation(nx // 8, ny // 8, order=order) as s:
        s.pres, s.tmp = x0, b
        e = s.poisson_residual()
        r, eo = oracle.poisson_residual(x0, b)
        assert np.array_equal(s.pold, r) and e == eo
        assert np.array_equal(s.pres, x0) and np.array_equal(s.tmp, b)
        for nsweeps in (1, 2, 5):
            s.pres = x0
            e = s.jacobi_sweeps(nsweeps, omega=0.8)
            xo, eo = oracle.jacobi_sweeps(x0, b, 0.8, nsweeps)
            assert np.array_equal(s.pres, xo), (nsweeps, np.abs(s.pres - xo).max())
            assert e == eo
            assert np.array_equal(s.pold, oracle.jacobi_sweeps(x0, b, 0.8, nsweeps - 1)[0])  # the previous iterate
        assert s.jacobi_sweeps(0) == 0.0 and np.array_equal(s.pres, xo)


def test_jacobi_smoother_reduces_the_residual_at_2048(gpu_lib):
    """BASELINE.json configs[1] (2048^2, 50 Jacobi pressure iterations): properties that need no oracle -- constants
    are a fixed point for b = 0, the residual norm of a zero-mean problem does not grow, the norm the sweep
    reports is the norm cup2d_poisson_residual computes for the same iterate"""
    n = 2048
    rng = np.random.default_rng(11)
    with make_sim(n) as s:
        s.fill(L.TMP, 0.0)
        s.fill(L.PRES, 3.25)
        assert s.jacobi_sweeps(3) == 0.0 and np.all(s.pres == 3.25)
        b = rng.uniform(-1, 1, (n, n))
        b -= b.mean()
        s.tmp = b
        s.fill(L.PRES, 0.0)
        e0 = s.poisson_residual()
        assert e0 == np.abs(b).max()
        s.fill(L.PRES, 0.0)
        e_prev = s.jacobi_sweeps(50, omega=0.8)   # norm of the iterate before sweep 50 ...
        x49 = s.pold.copy()
        x50 = s.pres.copy()
        s.pres = x49
        assert s.poisson_residual() == e_prev     # ... is what the residual kernel gives for that iterate
        s.pres = x50
        assert s.poisson_residual() < 0.5 * e0 and e_prev < 0.5 * e0


def test_solver_vs_golden_and_oracle(gpu_lib, oracle):
    G = golden("poisson_n32.npz")
    with make_sim(32) as s:
        s.pres = G["x0"]
        s.tmp = G["b"]
        info = s.poisson_solve(tol=1e-10, rel_tol=0.0, max_restarts=100)
        x = s.pres
    assert abs(info["iters"] - int(G["iters"])) <= 2
    assert abs(info["err_init"] - float(G["err_init"])) < 1e-12
    assert info["err"] <= 1e-10
    assert np.abs(G["b"] - oracle.apply_A(x)).max() <= 1.0001e-10
    assert np.abs(x - G["x"]).max() < 1e-8


@pytest.mark.parametrize("n", [64, 256])
def test_solver_tolerances_and_zero_tolerance_mode(gpu_lib, oracle, n):
    rng = np.random.default_rng(n)
    b = rng.uniform(-1, 1, (n, n))
    b -= b.mean()
    xo, io = oracle.bicgstab(b, tol=1e-8, rel_tol=0.0, max_restarts=100)
    with make_sim(n) as s:
        s.tmp = b
        s.fill(L.PRES, 0.0)
        info = s.poisson_solve(tol=1e-8)
        x = s.pres
        # the count is chaotic in the round-off of the dot products (summation order differs from the
        # oracle's, as cuBLAS's does from both): same convergence, not the same count
        assert abs(info["iters"] - io["iters"]) <= max(5, io["iters"] // 4)
        assert np.abs(b - oracle.apply_A(x)).max() <= 1.0001e-8
        # both satisfy |b - A x|_inf <= 1e-8; the solutions then differ by at most ~|A^-1| * 2e-8, and
        # |A^-1| ~ (n/pi)^2 for the Neumann Laplacian in cell units
        assert np.abs((x - x.mean()) - (xo - xo.mean())).max() < 2e-8 * (n / np.pi) ** 2
        # relative tolerance stop (main.cpp -poissonTolRel)
        s.fill(L.PRES, 0.0)
        info = s.poisson_solve(tol=0.0, rel_tol=1e-3, max_restarts=0)
        assert info["err"] / info["err_init"] <= 1e-3
        # tolerance 0 + iteration cap: runs exactly max_iter iterations and returns the best iterate
        s.fill(L.PRES, 0.0)
        info = s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=40)
        assert info["iters"] == 40
        assert np.abs(b - oracle.apply_A(s.pres)).max() <= info["err"] * (1 + 1e-9) + 1e-14


def test_zero_rhs(gpu_lib, oracle):
    """b = 0: like the reference (error < error_opt is never true at 0, cuda.cu:535) the loop runs to
    the iteration cap and returns x_opt = x0 = 0 without producing NaNs."""
    xo, io = oracle.bicgstab(np.zeros((32, 32)), tol=1e-10, max_iter=60)
    with make_sim(32) as s:
        s.fill(L.TMP, 0.0)
        s.fill(L.PRES, 0.0)
        info = s.poisson_solve(tol=1e-10, max_iter=60)
        assert np.abs(s.pres).max() == 0.0 and info["iters"] == io["iters"] == 60
        assert np.abs(xo).max() == 0.0


def test_full_steps_vs_reference_time_loop(gpu_lib, oracle):
    G = golden("run_n32_3steps.npz")
    with make_sim(32, float(G["nu"]), strict=True) as s:
        s.vel = G["vel0"]
        for k in range(3):
            r = s.step()  # zero tolerances like main.cpp:7028-7030 for step < 10
            assert abs(r["dt"] - G["dts"][k]) < 1e-12 * r["dt"]
        assert np.abs(s.vel - G["vel"]).max() < 1e-10
        assert np.abs(s.pres - G["pres"]).max() < 1e-8


def test_step_matches_oracle_at_256_fast_math(gpu_lib, oracle):
    n = 256
    vel = oracle.taylor_green(n)
    v, p = vel.copy(), np.zeros((n, n))
    with make_sim(n, strict=False) as s:
        s.vel = vel
        for k in range(2):
            v, p, dt, info = oracle.step(v, p, 1.0 / n, 1e-3, 0.5, tol=1e-9, rel_tol=0.0, max_restarts=100)
            r = s.step(tol=1e-9, rel_tol=0.0, max_restarts=100)
            # dt of step 2 depends on max|u| after a projection solved to 1e-9
            assert abs(r["dt"] - dt) < 1e-7 * dt
        assert np.abs(s.vel - v).max() < 1e-7


def test_properties_at_scale(gpu_lib):
    """Size-independent properties at 2048^2 (a BASELINE.json config) where the CPU oracle is too slow:
    constants are the nullspace of A; a uniform interior flow has zero rhs; the projection removes
    divergence to the solver tolerance; STRICT and FAST agree to round-off."""
    import cup2d_amd
    n = 2048
    with cup2d_amd.Simulation(n // 8, nu=1e-3) as s:
        s.fill(L.PRES, 3.25)
        s.apply_A(L.TMP, L.PRES)
        assert np.abs(s.tmp).max() == 0.0
        x = (np.arange(n) + 0.5) / n
        X, Y = np.meshgrid(x, x, indexing="xy")
        vel = np.stack([np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y), -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y)], -1)
        vel += 1e-3 * np.random.default_rng(1).uniform(-1, 1, vel.shape)
        s.vel = vel
        dt = s.compute_dt()
        s.set_math(True)
        s.advect_diffuse_rhs(dt)
        a = s.tmpV
        s.set_math(False)
        s.advect_diffuse_rhs(dt)
        b = s.tmpV
        assert np.abs(a - b).max() <= 2e-13 * np.abs(a).max()
        r = s.step(tol=0.0, rel_tol=1e-3, max_restarts=100, max_iter=400)
        assert r["iters"] < 400  # converged on the relative tolerance before the cap
        assert np.isfinite(s.vel).all() and np.abs(s.vel).max() < 1.1


@pytest.mark.gpu
def test_consecutive_steps_reuse_what_the_previous_step_left_gpu(gpu_lib, oracle):
    """cup2d_step's short cuts between consecutive calls -- max|u| for dt from the maxima the previous projection kernel
    wrote, the solve told that its initial guess is zero instead of filling pres -- change no number: a run of bare
    step() calls equals a run where another call on the context sits between the steps (which switches the max|u| short
    cut off) and equals the sequence of the separate entry points."""
    import cup2d_amd
    from cup2d_amd import lib as L
    n = 128
    vel = oracle.taylor_green(n)
    runs = []
    for mode in ("bare", "interleaved", "separate"):
        with cup2d_amd.Simulation(n // 8, nu=1e-3) as s:
            s.vel = vel
            dts = []
            for k in range(4):
                if mode == "separate":
                    dt = s.compute_dt()
                    s.advect_diffuse_rk2(dt)
                    s.poisson_rhs(dt)
                    s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=60)
                    s.project(dt)
                    dts.append(dt)
                else:
                    dts.append(s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=60)["dt"])
                    if mode == "interleaved":
                        s.synchronize()
            runs.append((dts, s.vel.copy(), s.pres.copy()))
    for other in runs[1:]:
        assert other[0] == runs[0][0]
        assert np.array_equal(other[1], runs[0][1]) and np.array_equal(other[2], runs[0][2])


@pytest.mark.gpu
def test_contexts_reuse_pooled_buffers_zero_filled_gpu(gpu_lib, oracle):
    """cup2d_destroy hands a context's device buffers to the library's pool, the next cup2d_create takes them back
    zero-filled: a second simulation in recycled memory starts from zero fields and reproduces the first bit for bit;
    cup2d_trim_pool returns the idle buffers to the driver (and the next context still works)."""
    import cup2d_amd
    from cup2d_amd import lib as L
    n = 128
    vel = oracle.taylor_green(n)
    out = []
    for k in range(3):
        with cup2d_amd.Simulation(n // 8, nu=1e-3) as s:
            assert not s.pres.any() and not s.vel.any() and not s.tmp.any()  # recycled or fresh: zeros
            s.vel = vel
            r = [s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=40) for _ in range(2)]
            out.append((r, s.vel.copy(), s.pres.copy()))
            s.pres = np.full((n, n), 7.0)  # leave something behind in the buffers
        if k == 1:
            assert L.load_library().cup2d_trim_pool() == 0
    for o in out[1:]:
        assert o[0] == out[0][0] and np.array_equal(o[1], out[0][1]) and np.array_equal(o[2], out[0][2])
