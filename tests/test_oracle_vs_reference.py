"""CPU tests: the plain-C oracle (oracle/cup2d_oracle.c) against
 (1) the committed golden vectors generated from the reference itself (tests/golden/make_golden.py),
 (2) the reference run live, when oracle/_ref/ref_harness exists (authoring container / shipped binary).
Block functors must agree BIT FOR BIT; the solver (cuBLAS/cuSPARSE order unspecified upstream) to
round-off."""
import numpy as np
import pytest

from conftest import golden


@pytest.mark.parametrize("case", ["functors_n32_tg.npz", "functors_n32_noise.npz"])
def test_functors_bit_exact_vs_golden(oracle, case):
    G = golden(case)
    n, nu, dt = int(G["n"]), float(G["nu"]), float(G["dt"])
    h = 1.0 / n
    assert float(G["h"]) == h
    assert oracle.compute_dt(h, nu, 0.5, float(G["umax"])) == float(G["dt_ref"])
    assert np.array_equal(oracle.advect_diffuse_rhs(G["vel"], h, nu, dt), G["advdiff_rhs"])
    v2, s1 = oracle.rk2_advect_diffuse(G["vel"], h, nu, dt)
    assert np.array_equal(s1, G["rk2_stage1"])
    assert np.array_equal(v2, G["rk2_vel"])
    assert np.array_equal(oracle.vorticity(v2, h), G["vorticity"])
    pr = oracle.pressure_rhs(v2, h, dt, G["udef"], G["chi"])
    assert np.array_equal(pr, G["pressure_rhs"])
    assert np.array_equal(oracle.laplacian_sub(G["pres"], pr), G["poisson_b"])
    g = oracle.pressure_correction(G["pres"], h, dt)
    assert np.array_equal(g, G["pgrad_tmpV"])
    assert np.array_equal(oracle.add_scaled(v2, g, h), G["projected_vel"])


def test_block_order_is_hilbert():
    from cup2d_amd.grid import BlockGrid
    G = golden("functors_n32_tg.npz")
    g = BlockGrid(4, 4, order="hilbert")
    assert np.array_equal(g.coords, G["block_order"])


def test_poisson_operator_and_solver_vs_golden(oracle):
    G = golden("poisson_n32.npz")
    # the reference-assembled COO matrix equals the matrix-free 5-point operator (to summation order)
    assert np.abs(oracle.apply_A(G["x0"]) - G["Ax0"]).max() < 1e-14
    x, info = oracle.bicgstab(G["b"], x0=G["x0"], tol=1e-10, max_restarts=100)
    assert info["iters"] == int(G["iters"])
    assert abs(info["err_init"] - float(G["err_init"])) < 1e-13
    assert np.abs(x - G["x"]).max() < 1e-9
    assert np.abs(G["b"] - oracle.apply_A(x)).max() <= 1e-10


def test_P_inv_is_minus_inverse(oracle):
    P = oracle.P_inv()
    A = np.zeros((64, 64))
    for i in range(64):
        for j in range(64):
            d = abs(i % 8 - j % 8) + abs(i // 8 - j // 8)
            A[i, j] = 4.0 if d == 0 else (-1.0 if d == 1 else 0.0)
    assert np.abs(P @ A + np.eye(64)).max() < 1e-13
    assert np.abs(P - P.T).max() < 1e-15


def test_full_steps_vs_reference_time_loop(oracle):
    G = golden("run_n32_3steps.npz")
    n = int(G["n"])
    v, p = G["vel0"].copy(), np.zeros((n, n))
    for k in range(3):
        v, p, dt, info = oracle.step(v, p, 1.0 / n, float(G["nu"]), float(G["cfl"]), tol=0.0, rel_tol=0.0, max_restarts=100)
        assert abs(dt - G["dts"][k]) < 1e-13 * dt + 1e-18
    # solver runs to machine precision in the first 10 steps (main.cpp:7028-7030): fields agree to
    # the conditioning of the Poisson problem
    assert np.abs(v - G["vel"]).max() < 1e-10
    assert np.abs(p - G["pres"]).max() < 1e-8


def test_analytic_properties(oracle):
    n = 32
    h = 1.0 / n
    const = np.ones((n, n, 2)) * np.array([0.3, -0.2])
    # interior of a constant field has zero rhs (walls break it at the boundary only)
    r = oracle.advect_diffuse_rhs(const, h, 1e-3, 1e-3)
    assert np.abs(r[4:-4, 4:-4]).max() == 0.0
    # A * 1 = 0 (Neumann), constants are the nullspace
    assert np.abs(oracle.apply_A(np.ones((n, n)))).max() == 0.0


def test_live_reference_if_present(oracle):
    if not oracle.have_reference():
        pytest.skip("oracle/_ref/ref_harness not built (needs /root/reference)")
    n = 64
    vel = oracle.taylor_green(n, noise=0.3, seed=99)
    rng = np.random.default_rng(5)
    pres = rng.uniform(-1, 1, (n, n))
    R = oracle.ref_functors(vel, 4e-5, pres=pres)
    h, dt = 1.0 / n, R["dt"]
    assert np.array_equal(oracle.advect_diffuse_rhs(vel, h, 4e-5, dt), R["advdiff_rhs"])
    v2, _ = oracle.rk2_advect_diffuse(vel, h, 4e-5, dt)
    assert np.array_equal(v2, R["rk2_vel"])
    pr = oracle.pressure_rhs(v2, h, dt)
    assert np.array_equal(pr, R["pressure_rhs"])
    assert np.array_equal(oracle.laplacian_sub(pres, pr), R["poisson_b"])


@pytest.mark.parametrize("nx,ny", [(128, 32), (32, 128)])
def test_live_reference_on_a_rectangle_if_present(oracle, nx, ny):
    """-bpdx 4 -bpdy 1 and -bpdx 1 -bpdy 4 (run.sh itself runs -bpdx 2 -bpdy 1): the shape of BASELINE.json configs[3]'s grid on
    two ranks (8192 x 2048 cells) and of its 2 x 4 layout's global grid -- h = extent / max(bpdx, bpdy) / 8 / 2^level
    (main.cpp:6338), walls on a non-square domain: the restatement equals the reference's functors bit for bit"""
    if not oracle.have_reference():
        pytest.skip("oracle/_ref/ref_harness not built (needs /root/reference)")
    vel = oracle.taylor_green(nx, noise=0.2, seed=7, ny=ny)
    rng = np.random.default_rng(6)
    pres, chi, udef = rng.uniform(-1, 1, (ny, nx)), rng.uniform(0, 1, (ny, nx)), 0.1 * rng.uniform(-1, 1, (ny, nx, 2))
    R = oracle.ref_functors(vel, 1e-3, pres=pres, chi=chi, udef=udef, nomatrix=True)
    h, dt = 1.0 / max(nx, ny), float(R["dt"])
    assert float(R["h"]) == h and float(R["umax"]) == np.abs(vel).max()
    assert oracle.compute_dt(h, 1e-3, 0.5, float(R["umax"])) == float(R["dt_ref"])
    assert np.array_equal(oracle.advect_diffuse_rhs(vel, h, 1e-3, dt), R["advdiff_rhs"])
    v2, _ = oracle.rk2_advect_diffuse(vel, h, 1e-3, dt)
    assert np.array_equal(v2, R["rk2_vel"])
    assert np.array_equal(oracle.vorticity(v2, h), R["vorticity"])
    pr = oracle.pressure_rhs(v2, h, dt, udef, chi)
    assert np.array_equal(pr, R["pressure_rhs"])
    assert np.array_equal(oracle.laplacian_sub(pres, pr), R["poisson_b"])
    g = oracle.pressure_correction(pres, h, dt)
    assert np.array_equal(g, R["pgrad_tmpV"])
    assert np.array_equal(oracle.add_scaled(v2, g, h), R["projected_vel"])
