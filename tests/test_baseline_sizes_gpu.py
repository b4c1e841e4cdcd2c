"""GPU parity at the sizes BASELINE.json names (-m gpu): 2048^2 (configs[1]), 4096^2 (configs[2], the bench
configuration) and one 8192^2 functor pass (configs[3]'s global grid on one GPU).

What small grids cannot catch is index arithmetic that only breaks at scale: size_t slab offsets, the persistent
grids' group_range, 262 144-block neighbour tables, the moved last tile, ticket counters.  So the comparisons here
are the same bit-level ones the 32^2..256^2 tests make, against the reference itself where it finishes in a minute or
two (oracle/_ref/ref_harness, the reference's own main.cpp, run live on the box with nomatrix=1 for the functor
passes) and against its bit-identical C restatement (oracle/liboracle.so) at 8192^2:

  * STRICT arithmetic: array_equal for the advect-diffuse RHS, the RK2 result, vorticity, pressure_rhs (with the
    chi / u_def terms), pressure_rhs1, the pressure gradient and the projected velocity, blocks in the reference's
    Hilbert order (main.cpp:5441-5503, 6607-6642, 3343-3366, 6105-6139, 6209-6230, 6021-6043, 7180-7187);
  * FAST arithmetic (what bench.py runs): |rhs - ref| <= 2e-13 max|rhs|;
  * the solver at the bench configuration (4096^2, FAST, 50 iterations at zero tolerance): fused and five-sweep
    organisation agree, and for each the residual the solver REPORTS is the residual of the iterate it RETURNS
    (cup2d_poisson_residual recomputes max|b - A x| from the fields);
  * one whole 4096^2 time step against the reference's own loop (main.cpp:6576-7187) with both solvers capped at 50
    iterations (main.cpp:7028-7030 runs the first ten steps at zero tolerance).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from cup2d_amd import lib as L  # noqa: E402


def _fields(O, n, seed=7):
    """Taylor-Green + noise and smooth-plus-noise pressure, chi in [0, 1], u_def"""
    rng = np.random.default_rng(seed + n)
    vel = O.taylor_green(n, noise=1e-3, seed=20250117)
    x = (np.arange(n) + 0.5) / n
    X, Y = np.meshgrid(x, x, indexing="xy")
    pres = np.cos(2 * np.pi * X) * np.cos(4 * np.pi * Y) + 1e-2 * rng.uniform(-1, 1, (n, n))
    chi = rng.uniform(0, 1, (n, n))
    udef = 0.1 * rng.uniform(-1, 1, (n, n, 2))
    return vel, pres, chi, udef


def _need_reference(O):
    if not O.have_reference():
        pytest.fail("oracle/_ref/ref_harness is missing: build it where /root/reference exists (make -C oracle ref); "
                    "it travels to the GPU box with the snapshot")


@pytest.mark.parametrize("n", [2048, 4096])
def test_functors_vs_live_reference_at_baseline_size(gpu_lib, oracle, n):
    import cup2d_amd
    O = oracle
    _need_reference(O)
    nu = 1e-3
    vel, pres, chi, udef = _fields(O, n)
    G = O.ref_functors(vel, nu, pres=pres, chi=chi, udef=udef, nomatrix=True)
    dt = float(G["dt"])
    with cup2d_amd.Simulation(n // 8, nu=nu) as s:  # blocks in the reference's Hilbert order
        # the device block order IS the reference's Info::id order (main.cpp:1550-1562)
        assert np.array_equal(s.grid.coords, G["block_order"])
        s.set_math(True)
        s.vel = vel
        assert s.max_abs_vel() == float(G["umax"])
        assert s.compute_dt() == float(G["dt_ref"])
        s.advect_diffuse_rhs(dt)
        assert np.array_equal(s.tmpV, G["advdiff_rhs"])
        s.advect_diffuse_rk2(dt)
        assert np.array_equal(s.vel, G["rk2_vel"])
        s.vorticity()
        assert np.array_equal(s.tmp, G["vorticity"])
        s.tmpV = udef
        s.chi = chi
        s.pressure_rhs(dt, use_bodies=True)
        assert np.array_equal(s.tmp, G["pressure_rhs"])
        s.pold = pres
        s.laplacian_sub()
        assert np.array_equal(s.tmp, G["poisson_b"])
        # main.cpp:7007-7026 as the ONE fused kernel cup2d_step launches (pold = pres; pres = 0; b = rhs - lap pold)
        s.pres = pres
        s.fill(L.TMP, 0.0)
        s.poisson_rhs(dt, use_bodies=True)
        assert np.array_equal(s.tmp, G["poisson_b"])
        assert np.array_equal(s.pold, pres) and not s.pres.any()
        s.pres = pres
        s.pressure_correction(dt)
        assert np.array_equal(s.tmpV, G["pgrad_tmpV"])
        s.add_correction()
        assert np.array_equal(s.vel, G["projected_vel"])
        # FAST arithmetic (bench.py's): round-off apart the same numbers
        s.set_math(False)
        s.vel = vel
        s.advect_diffuse_rhs(dt)
        scale = np.abs(G["advdiff_rhs"]).max()
        assert np.abs(s.tmpV - G["advdiff_rhs"]).max() <= 2e-13 * scale
        s.advect_diffuse_rk2(dt)
        assert np.abs(s.vel - G["rk2_vel"]).max() <= 1e-13 * np.abs(G["rk2_vel"]).max()
        # the fused projection of cup2d_step (gradient + update in one kernel) on x = pres, pold = 0:
        # pres <- pres - mean(pres) twice, vel += -0.5 dt h grad(pres) / h^2 (main.cpp:7120-7187)
        s.set_math(True)
        s.vel = G["rk2_vel"]
        s.pres = pres
        s.fill(L.POLD, 0.0)
        s.project(dt)
        h = 1.0 / n
        p1 = s.pres
        ref_p = pres - pres.sum() * h * h / (n * n * h * h)
        assert np.abs(p1 - ref_p).max() <= 1e-12  # two mean removals: summation order is the only freedom
        gref = O.pressure_correction(p1, h, dt)
        assert np.array_equal(s.vel, O.add_scaled(G["rk2_vel"], gref, h))


def test_functors_vs_live_reference_on_the_configs3_rectangle(gpu_lib, oracle):
    """8192 x 2048 cells (the reference's -bpdx 4 -bpdy 1 -levelStart 8): the global grid of BASELINE.json configs[3] on two ranks
    in x, whose single context is the reference of tests/test_distributed.py::test_decomposed_path_at_configs3_rank_size_gpu[2-1-..]
    -- pinned here to the reference itself (oracle/_ref/ref_harness run live), every functor STRICT bit for bit: a rectangle's
    h = extent / max(bpdx, bpdy) / 8 / 2^level (main.cpp:6338), its walls, 262 144 blocks in an order that is not the square's"""
    import cup2d_amd
    O = oracle
    _need_reference(O)
    nx, ny, nu = 8192, 2048, 1e-3
    rng = np.random.default_rng(7 + nx)
    vel = O.taylor_green(nx, noise=1e-3, seed=20250117, ny=ny)
    x, y = (np.arange(nx) + 0.5) / nx, (np.arange(ny) + 0.5) / nx
    X, Y = np.meshgrid(x, y, indexing="xy")
    pres = np.cos(2 * np.pi * X) * np.cos(4 * np.pi * Y) + 1e-2 * rng.uniform(-1, 1, (ny, nx))
    del X, Y
    chi = rng.uniform(0, 1, (ny, nx))
    udef = 0.1 * rng.uniform(-1, 1, (ny, nx, 2))
    G = O.ref_functors(vel, nu, pres=pres, chi=chi, udef=udef, nomatrix=True)
    dt = float(G["dt"])
    with cup2d_amd.Simulation(nx // 8, ny // 8, nu=nu) as s:
        assert s.h == float(G["h"]) == 1.0 / nx
        s.set_math(True)
        s.vel = vel
        assert s.max_abs_vel() == float(G["umax"]) and s.compute_dt() == float(G["dt_ref"])
        s.advect_diffuse_rhs(dt)
        assert np.array_equal(s.tmpV, G["advdiff_rhs"])
        s.advect_diffuse_rk2(dt)
        assert np.array_equal(s.vel, G["rk2_vel"])
        s.vorticity()
        assert np.array_equal(s.tmp, G["vorticity"])
        s.tmpV = udef
        s.chi = chi
        s.pressure_rhs(dt, use_bodies=True)
        assert np.array_equal(s.tmp, G["pressure_rhs"])
        s.pold = pres
        s.laplacian_sub()
        assert np.array_equal(s.tmp, G["poisson_b"])
        s.pres = pres
        s.pressure_correction(dt)
        assert np.array_equal(s.tmpV, G["pgrad_tmpV"])
        s.add_correction()
        assert np.array_equal(s.vel, G["projected_vel"])
        s.set_math(False)
        s.vel = vel
        s.advect_diffuse_rhs(dt)
        assert np.abs(s.tmpV - G["advdiff_rhs"]).max() <= 2e-13 * np.abs(G["advdiff_rhs"]).max()


def test_functor_pass_at_8192_vs_restatement(gpu_lib, oracle):
    """configs[3]'s global grid (1 048 576 blocks, 5.4 GB of fields) on one GPU: one pass of every block functor
    against oracle/liboracle.so, which tests/test_oracle_vs_reference.py pins bit for bit to the reference functors"""
    import cup2d_amd
    O = oracle
    n, nu = 8192, 1e-3
    h = 1.0 / n
    vel, pres, chi, udef = _fields(O, n)
    dt = O.compute_dt(h, nu, 0.5, np.abs(vel).max())
    with cup2d_amd.Simulation(n // 8, nu=nu) as s:
        s.set_math(True)
        s.vel = vel
        assert s.compute_dt() == dt
        s.advect_diffuse_rhs(dt)
        ref = O.advect_diffuse_rhs(vel, h, nu, dt)
        assert np.array_equal(s.tmpV, ref)
        s.set_math(False)
        s.advect_diffuse_rhs(dt)
        assert np.abs(s.tmpV - ref).max() <= 2e-13 * np.abs(ref).max()
        del ref
        s.set_math(True)
        s.advect_diffuse_rk2(dt)
        ref2, _ = O.rk2_advect_diffuse(vel, h, nu, dt)
        assert np.array_equal(s.vel, ref2)
        s.vorticity()
        assert np.array_equal(s.tmp, O.vorticity(ref2, h))
        s.tmpV = udef
        s.chi = chi
        s.pressure_rhs(dt, use_bodies=True)
        b = O.pressure_rhs(ref2, h, dt, udef=udef, chi=chi)
        assert np.array_equal(s.tmp, b)
        s.pold = pres
        s.laplacian_sub()
        b = O.laplacian_sub(pres, b)
        assert np.array_equal(s.tmp, b)
        s.pres = pres
        s.pressure_correction(dt)
        g = O.pressure_correction(pres, h, dt)
        assert np.array_equal(s.tmpV, g)
        s.add_correction()
        assert np.array_equal(s.vel, O.add_scaled(ref2, g, h))
        # the tile kernels of the path (16-block tiles, moved last tile): residual of the 5-point operator
        s.pres = pres
        s.tmp = b
        e = s.poisson_residual()
        r = b - O.apply_A(pres)
        assert np.array_equal(s.pold, r) and e == np.abs(r).max()


def _poisson_system(s, O, vel, nu):
    """the Poisson system of the first time step from `vel` (FAST arithmetic), left in TMP / PRES = 0"""
    s.set_math(False)
    s.vel = vel
    dt = s.compute_dt()
    s.advect_diffuse_rk2(dt)
    s.fill(L.PRES, 0.0)
    s.poisson_rhs(dt)
    return dt


def test_solver_at_the_bench_configuration(gpu_lib, oracle):
    """4096^2, FAST arithmetic, 50 BiCGSTAB iterations at zero tolerance: the configuration bench.py reports.
    (a) the fused solver and the five sweeps run the same recurrences (cuda.cu:403-548) and differ by round-off only
        (residuals equal to 1e-6 relative, iterates to 2e-9 of max|x|);
    (b) for both, the residual norm the solver reports for the iterate it returns (x_opt, cuda.cu:535-547) is the
        residual of that iterate: max|b - A x| recomputed from the fields by cup2d_poisson_residual.  The recurrence
        residual and the true one drift apart by round-off of size eps * |A| |x| (<= 1e-9 here), far below the
        residual itself after 50 iterations."""
    import cup2d_amd
    O = oracle
    n, nu = 4096, 1e-3
    vel = O.taylor_green(n)
    out = {}
    with cup2d_amd.Simulation(n // 8, nu=nu) as s:
        _poisson_system(s, O, vel, nu)
        b = s.tmp
        for kind in ("fused", "sweeps"):
            s.set_solver(fused=kind == "fused", finish_in_kernel=True)
            s.fill(L.PRES, 0.0)
            r = s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
            assert s.last_solver() == kind and r["iters"] == 50
            if kind == "fused":
                form = s.last_solver_form()
            assert np.array_equal(s.tmp, b)  # the solve leaves b alone
            true = s.poisson_residual()
            out[kind] = (s.pres, r, true)
            assert r["err"] < 0.2 * r["err_init"], r  # 50 iterations did reduce the residual
            assert abs(true - r["err"]) <= 1e-6 * r["err"] + 1e-9, (kind, true, r)
        placement = s.placement()
    xf, rf, _ = out["fused"]
    xs, rs, _ = out["sweeps"]
    # the first two-launch solve of this context searched for a fast placement of its vectors (krylov_fused.hip tune_placement):
    # several complete sets were timed, the kept one is the fastest seen -- and the iterates above are what the solver computes on it
    print("placement search:", placement, "form", form)
    if form[0] == "eab":   # (the search probes the two-launch organisation's kernels: CUP2D_FUSED_FORM=full / edge run without it)
        assert placement["candidates"] >= 2 and placement["kept_us"] <= placement["first_us"] <= placement["slowest_us"], placement
    assert rf["err_init"] == rs["err_init"]
    scale = np.abs(xs).max()
    print("bench-config solver: err_init %.3e  fused err %.6e  sweeps err %.6e  max|x_f - x_s| / max|x| = %.2e"
          % (rf["err_init"], rf["err"], rs["err"], np.abs(xf - xs).max() / scale))
    assert abs(rf["err"] - rs["err"]) <= 1e-6 * rs["err"]
    # measured 2.3e-10: fifty iterations of a Krylov recurrence amplify the round-off of two different summation orders
    # (P_inv on the matrix cores vs fast diagonalisation, x accumulated as x0 + P_inv y vs in place); both iterates
    # have the same residual to 7 digits, which is what the stopping rule sees
    assert np.abs(xf - xs).max() <= 2e-9 * scale


def test_solver_at_8192_two_launches_vs_five_sweeps(gpu_lib, oracle):
    """configs[3]'s global grid on one context (1 048 576 blocks, 65 536 tiles, 32 rounds per workgroup; 537 MB per Krylov
    vector): eight iterations at zero tolerance of the default two-launch organisation (k_edge MODE 3 / MODE 2, hand-over
    between sibling waves, descending rounds) against the five sweeps with the same preconditioner arithmetic -- the last
    iterates to 1e-10 of max|x| -- and, for both, the residual the recurrence carries against max|b - A x| recomputed on the
    CPU with the oracle's operator (cuda.cu:403-548; the 4096^2 tests cannot reach offsets beyond 2^31 bytes per vector
    pair or tile indices beyond 2^16)."""
    import cup2d_amd
    O = oracle
    n, nu = 8192, 1e-3
    vel = O.taylor_green(n)
    last = {}
    with cup2d_amd.Simulation(n // 8, nu=nu) as s:
        _poisson_system(s, O, vel, nu)
        del vel
        b = s.tmp
        s.set_precond(L.PRECOND_MFMA)
        for kind in ("fused", "sweeps"):
            s.set_solver(fused=kind == "fused", finish_in_kernel=True)
            s.keep_last_iterate(True)
            s.fill(L.PRES, 0.0)
            r = s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=8)
            assert s.last_solver() == kind and r["iters"] == 8
            if kind == "fused":
                assert s.last_solver_form()[:2] == ("eab", 1), s.last_solver_form()
            err_last = s.last_iterate_to(L.POLD)
            last[kind] = (s.pold, err_last, r)
            assert np.array_equal(s.tmp, b)
    xf, ef, rf = last["fused"]
    xs, es, rs = last["sweeps"]
    scale = np.abs(xs).max()
    d = np.abs(xf - xs).max() / scale
    true_f = np.abs(b - O.apply_A(xf)).max()
    true_s = np.abs(b - O.apply_A(xs)).max()
    print("8192^2, 8 iterations: max|x_two - x_five| / max|x| = %.2e; recurrence / recomputed residual: two launches %.6e / %.6e, "
          "five sweeps %.6e / %.6e (err_init %.3e)" % (d, ef, true_f, es, true_s, rf["err_init"]))
    assert rf["err_init"] == rs["err_init"] == np.abs(b).max()
    assert d <= 1e-10
    assert abs(true_f - ef) <= 1e-6 * ef + 1e-9 and abs(true_s - es) <= 1e-6 * es + 1e-9
    assert ef < rf["err_init"]


def test_whole_step_at_4096_vs_reference_loop(gpu_lib, oracle):
    """One pass of the reference's own time-loop body (main.cpp:6576-7187) at 4096^2 with the Poisson solve capped at
    50 iterations on both sides (harness key maxiter; the reference's solver is cuda.cu, restated on the CPU) against
    cup2d_step in the bench configuration.  Tolerances: dt exact; the advected velocity (the solver's input) bit-exact
    in STRICT and 1e-13 in FAST; velocity 1e-9, pressure 1e-8 after the step (the two BiCGSTABs sum their dot products
    in different orders)."""
    import cup2d_amd
    O = oracle
    _need_reference(O)
    n, nu = 4096, 1e-3
    vel0 = O.taylor_green(n)
    R = O.ref_run(vel0, nu, steps=1, cfl=0.5, tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
    st = R["steps"][0]
    with cup2d_amd.Simulation(n // 8, nu=nu) as s:
        # the pieces, STRICT: advected velocity and Poisson right-hand side are the reference's bit for bit
        s.set_math(True)
        s.vel = vel0
        dt = s.compute_dt()
        assert dt == st["dt"]
        s.advect_diffuse_rk2(dt)
        assert np.array_equal(s.vel, st["vel_adv"])
        s.fill(L.PRES, 0.0)
        s.poisson_rhs(dt)
        assert np.array_equal(s.tmp, st["b"])
        # the whole step as bench.py runs it
        for fused in (True, False):
            s.set_math(False)
            s.set_solver(fused=fused, finish_in_kernel=True)
            s.vel = vel0
            s.fill(L.PRES, 0.0)
            s.fill(L.POLD, 0.0)
            r = s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
            assert r["dt"] == st["dt"] and r["iters"] == 50
            dv = np.abs(s.vel - R["vel"]).max()
            dp = np.abs(s.pres - R["pres"]).max()
            print("4096^2 step vs reference loop (%s): max|dvel| %.2e  max|dpres| %.2e  (max|pres| %.2e, reference err %.3e, here %.3e)"
                  % ("fused" if fused else "sweeps", dv, dp, np.abs(R["pres"]).max(), R["final"]["err"], r["err"]))
            assert dv <= 1e-9 and dp <= 1e-8
