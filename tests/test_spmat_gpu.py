"""GPU tests of seam B1: the reference's own main.cpp linked against libcup2d_spmat.so (this repository's
LocalSpMatDnVec on MI355X) reproduces the reference time loop; the assembled-operator path (sliced ELL)
agrees with the matrix-free stencil; custom block preconditioners; ranks sharing the GPU over MPI."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "spmat_driver")
MPIEXEC = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"


@pytest.mark.parametrize("force_matrix", [False, True])
def test_reference_time_loop_with_hip_spmat(gpu_lib, oracle, force_matrix):
    """main.cpp:6576-7290, every line the reference's own, with sim.mat -> libcup2d_spmat.so:
    3 steps from the golden initial condition must land on the golden state (which the reference
    produced with the CPU restatement of cuda.cu behind the same seam)."""
    assert oracle.have_reference_hip(), "oracle/_ref/ref_harness_hip was not shipped"
    G = golden("run_n32_3steps.npz")
    env = {"CUP2D_SPMAT_FORCE_MATRIX": "1"} if force_matrix else None
    R = oracle.ref_run(G["vel0"], float(G["nu"]), steps=3, tol=1e-11, rel_tol=0.0, max_restarts=100, hip=True, env=env)
    assert np.allclose([s["dt"] for s in R["steps"]], G["dts"], rtol=1e-12, atol=0)
    assert np.abs(R["vel"] - G["vel"]).max() < 1e-10
    assert np.abs(R["pres"] - G["pres"]).max() < 1e-8
    for k in range(3):
        assert np.abs(R["steps"][k]["b"] - G["b"][k]).max() < 1e-9


@pytest.mark.parametrize("force_matrix", [False, True])
def test_reference_assembled_system_solved_by_hip_spmat(gpu_lib, oracle, force_matrix):
    assert oracle.have_reference_hip()
    G = golden("poisson_n32.npz")
    env = {"CUP2D_SPMAT_FORCE_MATRIX": "1"} if force_matrix else None
    x, _, info = oracle.ref_solve(G["b"], x0=G["x0"], tol=1e-10, rel_tol=0.0, max_restarts=100, hip=True, env=env)
    assert abs(info["iters"] - int(G["iters"])) <= 3
    assert abs(info["err_init"] - float(G["err_init"])) < 1e-12
    assert np.abs(G["b"] - oracle.apply_A(x)).max() <= 1.0001e-10
    assert np.abs(x - G["x"]).max() < 1e-8


def test_amr_time_loop_with_hip_spmat(gpu_lib, oracle):
    """BASELINE.json configs[4] (block-AMR) through seam B1: the reference's own time loop with refinement ON --
    adapt(), coarse-fine labs, flux correction and the coarse-fine matrix rows are the reference's code -- with
    every linear solve (solveWithUpdate after each regrid: 16 -> 40 -> 76 blocks on levels 2..4) served by
    libcup2d_spmat.so's general sliced-ELL operator on the GPU, against the same run with the CPU restatement
    of cuda.cu behind the seam.  Same regrid history, same fields to the solve tolerance."""
    assert oracle.have_reference_hip(), "oracle/_ref/ref_harness_hip was not shipped"
    # one OpenMP thread: a fixed reduction order in the reference's own loops, so that the two runs differ by the solver only
    kw = dict(level_start=2, level_max=5, steps=8, rtol=2.0, ctol=0.5, nu=1e-3, max_iter=200, env={"OMP_NUM_THREADS": "1"})
    C = oracle.ref_run_amr(**kw)
    G = oracle.ref_run_amr(hip=True, **kw)
    assert [s["blocks"] for s in G["steps"]] == [s["blocks"] for s in C["steps"]]
    assert max(s["lmax"] for s in C["steps"]) - min(s["lmin"] for s in C["steps"]) >= 2  # three levels
    assert np.array_equal(G["blocks"], C["blocks"])
    assert np.allclose([s["dt"] for s in G["steps"]], [s["dt"] for s in C["steps"]], rtol=1e-9, atol=0)
    assert np.abs(G["vel"] - C["vel"]).max() < 1e-8 * max(1.0, np.abs(C["vel"]).max())
    assert np.abs(G["pres"] - C["pres"]).max() < 1e-7 * max(1.0, np.abs(C["pres"]).max())


@pytest.mark.parametrize("order,nbx,nby", [("hilbert", 8, 8), ("rowmajor", 5, 3)])
def test_assembled_operator_equals_stencil(gpu_lib, oracle, order, nbx, nby):
    import cup2d_amd
    from cup2d_amd import lib as L
    from cup2d_amd.grid import BlockGrid
    g = BlockGrid(nbx, nby, order=order)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (g.ny, g.nx))
    b = rng.uniform(-1, 1, (g.ny, g.nx))
    b -= b.mean()
    with cup2d_amd.Simulation(nbx, nby, grid=g, h=1.0 / g.nx) as s:
        s.pres = x
        s.apply_A(L.TMP, L.PRES)
        y_stencil = s.tmp
        s.tmp = b
        s.fill(L.PRES, 0.0)
        i_st = s.poisson_solve(tol=1e-9)
        x_st = s.pres
        # the same operator as an assembled matrix, entries shuffled
        r, c, v = g.poisson_coo()
        perm = rng.permutation(r.size)
        s.set_matrix_coo(r[perm], c[perm], v[perm])
        s.pres = x
        s.apply_A(L.TMP, L.PRES)
        y_mat = s.tmp
        assert np.abs(y_mat - y_stencil).max() <= 4e-15 * 8  # different summation order only
        assert np.abs(y_mat - oracle.apply_A(x)).max() <= 4e-15 * 8
        s.tmp = b
        s.fill(L.PRES, 0.0)
        i_mat = s.poisson_solve(tol=1e-9)
        x_mat = s.pres
        assert i_mat["err"] <= 1e-9 and i_st["err"] <= 1e-9
        assert abs(i_mat["iters"] - i_st["iters"]) <= max(5, i_st["iters"] // 4)
        assert np.abs(b - oracle.apply_A(x_mat)).max() <= 1.0001e-9
        assert np.abs((x_mat - x_mat.mean()) - (x_st - x_st.mean())).max() < 2e-9 * (max(g.nx, g.ny) / np.pi) ** 2
        # a matrix that is NOT the stencil: scale one block's rows -> different operator, applied as given
        v2 = v.copy()
        v2[(r // 64) == 3] *= 0.5
        s.set_matrix_coo(r, c, v2)
        s.pres = x
        s.apply_A(L.TMP, L.PRES)
        want = g.to_blocks(oracle.apply_A(x))
        want[3] *= 0.5
        assert np.abs(g.to_blocks(s.tmp) - want).max() <= 4e-15 * 8
        s.clear_matrix()
        s.apply_A(L.TMP, L.PRES)
        assert np.array_equal(s.tmp, y_stencil)


def test_preconditioner_kinds_and_custom_P(gpu_lib, oracle):
    import cup2d_amd
    from cup2d_amd import lib as L
    n = 64
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, (n, n))
    with cup2d_amd.Simulation(n // 8) as s:
        s.pres = x
        P = s.P_inv()
        want = oracle.precond(x, P)
        for kind in (L.PRECOND_FD, L.PRECOND_MFMA, L.PRECOND_LDS):
            s.set_precond(kind)
            s.precond(L.TMP, L.PRES)
            assert np.abs(s.tmp - want).max() < 1e-14, kind
        # the reference's own Cholesky-built matrix (round-off away from ours) keeps fast diagonalisation
        s.set_precond(L.PRECOND_FD)
        s.set_P_inv(oracle.P_inv())
        s.precond(L.TMP, L.PRES)
        assert np.abs(s.tmp - want).max() < 1e-14
        # a caller-supplied, non-symmetric P: applied as z_b = P p_b (cuda.cu:484-486), densely
        Q = P + 0.01 * rng.uniform(-1, 1, (64, 64))
        s.set_P_inv(Q)
        blocks = s.grid.to_blocks(x)
        want_q = s.grid.from_blocks(blocks @ Q.T, 1)
        for kind in (L.PRECOND_MFMA, L.PRECOND_LDS):
            s.set_precond(kind)
            s.precond(L.TMP, L.PRES)
            assert np.abs(s.tmp - want_q).max() < 1e-13, kind
        with pytest.raises(cup2d_amd.Cup2dError):
            s.set_precond(L.PRECOND_FD)
        # and the solver converges with it
        b = rng.uniform(-1, 1, (n, n))
        b -= b.mean()
        s.tmp = b
        s.fill(L.PRES, 0.0)
        info = s.poisson_solve(tol=1e-8)
        assert info["err"] <= 1e-8 and np.abs(b - oracle.apply_A(s.pres)).max() <= 1.0001e-8


@pytest.mark.parametrize("ranks,env", [(1, {}), (1, {"CUP2D_SPMAT_FORCE_MATRIX": "1"}), (2, {}), (3, {})])
def test_spmat_ranks_share_the_gpu(gpu_lib, ranks, env):
    """LocalSpMatDnVec driven like main.cpp:7034-7131 from 1-3 MPI ranks on this GPU: 1 rank takes the
    matrix-free path (or the assembled one when forced), several ranks the assembled operator with the
    host-staged MPI halo of cuda.cu:365-380."""
    assert os.path.exists(DRIVER) and os.path.exists(MPIEXEC), "spmat_driver / mpiexec not shipped"
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([MPIEXEC, "-n", str(ranks), DRIVER, "solve", "6", "4", "1e-9"], capture_output=True, text=True,
                         timeout=300, env=e)
    assert out.returncode == 0 and "SOLVE_OK" in out.stdout, out.stdout + out.stderr
    structured = ranks == 1 and not env
    assert ("structured %d" % int(structured)) in out.stdout


def test_b2_call_sites_compiled_into_the_reference_loop(gpu_lib, oracle):
    """Seam B2 proven by compiling it: oracle/_ref/ref_harness_b2 is the reference's main.cpp with the call sites
    INTEGRATION.md lists (main.cpp:4659, 6611-6642, 7003-7027, 7031-7119, 7120-7187) replaced by cup2d_* calls
    (oracle/b2_patch.py).  Three steps from the golden initial condition must land on the golden state, which the
    unpatched reference produced."""
    assert oracle.have_reference_b2(), "oracle/_ref/ref_harness_b2 was not shipped (make -C oracle ref_b2)"
    G = golden("run_n32_3steps.npz")
    R = oracle.ref_run(G["vel0"], float(G["nu"]), steps=3, tol=1e-11, rel_tol=0.0, max_restarts=100, b2=True)
    assert np.allclose([s["dt"] for s in R["steps"]], G["dts"], rtol=1e-12, atol=0)
    assert np.abs(R["vel"] - G["vel"]).max() < 1e-10
    assert np.abs(R["pres"] - G["pres"]).max() < 1e-8
    for k in range(3):
        assert np.abs(R["steps"][k]["b"] - G["b"][k]).max() < 1e-9
    # STRICT arithmetic behind the seam: the first step's advected velocity and right-hand side are bit-identical
    assert np.array_equal(R["steps"][0]["vel_adv"], G["vel_adv"][0])
    assert np.array_equal(R["steps"][0]["b"], G["b"][0])


def test_b2_loop_at_2048_equals_the_unpatched_reference(gpu_lib, oracle):
    """one 2048^2 step (BASELINE.json configs[1]'s grid) of the patched loop against the unpatched harness, both solvers
    capped at 50 iterations: advected velocity and Poisson right-hand side bit for bit, state after the step 1e-9 / 1e-8"""
    assert oracle.have_reference_b2() and oracle.have_reference()
    n, nu = 2048, 1e-3
    vel0 = oracle.taylor_green(n)
    kw = dict(steps=1, tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=50)
    A = oracle.ref_run(vel0, nu, **kw)
    B = oracle.ref_run(vel0, nu, b2=True, **kw)
    assert A["steps"][0]["dt"] == B["steps"][0]["dt"]
    assert np.array_equal(A["steps"][0]["vel_adv"], B["steps"][0]["vel_adv"])
    assert np.array_equal(A["steps"][0]["b"], B["steps"][0]["b"])
    assert np.abs(A["vel"] - B["vel"]).max() <= 1e-9 and np.abs(A["pres"] - B["pres"]).max() <= 1e-8


FISH = "angle=0 L=0.4 xpos=0.5 ypos=0.5"  # the reference's own shape model (main.cpp:6378-6444) supplies chi and u_def


def test_penalisation_site_is_bit_identical_to_the_reference(gpu_lib, oracle):
    """SURVEY.md 8f item 3 (main.cpp:6643-7006): the reference's loop with one fish, three steps at 256^2, once entirely
    the reference's code (all sites of ref_harness_b2 off) and once with ONLY the penalisation site served by
    cup2d_body_set / cup2d_body_momentum / cup2d_penalize.  One OpenMP thread fixes the reference's summation order; the
    moments are added up in that order, so the body velocities, the blended velocity field and everything downstream are
    equal bit for bit."""
    assert oracle.have_reference_b2()
    vel0 = 0.1 * oracle.taylor_green(256)
    kw = dict(steps=3, tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=40, shapes=FISH, threads=1)
    A = oracle.ref_run(vel0, 1e-3, b2="none", **kw)
    B = oracle.ref_run(vel0, 1e-3, b2="penal", **kw)
    assert np.abs(A["vel"] - vel0).max() > 1e-2  # the fish is there
    for k in range(3):
        assert A["steps"][k]["dt"] == B["steps"][k]["dt"]
        assert np.array_equal(A["steps"][k]["vel_adv"], B["steps"][k]["vel_adv"])  # velocity after the blend
        assert np.array_equal(A["steps"][k]["b"], B["steps"][k]["b"])              # rhs with the chi / u_def terms
    assert np.array_equal(A["vel"], B["vel"]) and np.array_equal(A["pres"], B["pres"])


def test_all_sites_with_a_body_follow_the_reference(gpu_lib, oracle):
    """every call site on the GPU (RK2, penalisation, Poisson rhs with chi / u_def, solve, projection) with the fish in the
    flow: the state after three steps is the reference's to the solve tolerance"""
    assert oracle.have_reference_b2()
    vel0 = 0.1 * oracle.taylor_green(256)
    kw = dict(steps=3, tol=1e-11, rel_tol=0.0, max_restarts=100, shapes=FISH, threads=1)
    A = oracle.ref_run(vel0, 1e-3, b2="none", **kw)
    B = oracle.ref_run(vel0, 1e-3, b2=True, **kw)
    # with a body the very first pass of the loop already solves a non-trivial system (u_def drives a pressure), so the two
    # solvers' iterates differ at the solve tolerance times the conditioning of the 256^2 operator from step one on
    dv, dp = np.abs(A["vel"] - B["vel"]).max(), np.abs(A["pres"] - B["pres"]).max()
    print("all sites with a body: max|dvel| %.2e max|dpres| %.2e, dt %s vs %s" % (dv, dp, [s["dt"] for s in A["steps"]], [s["dt"] for s in B["steps"]]))
    assert np.allclose([s["dt"] for s in A["steps"]], [s["dt"] for s in B["steps"]], rtol=1e-6, atol=0)
    assert dv < 1e-6 and dp < 5e-5
