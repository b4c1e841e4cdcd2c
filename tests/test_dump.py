"""The reference's output format (dump(), main.cpp:3367-3466; read by the reference's post.py): byte-for-byte
against files written by the reference itself (golden fixture from tests/golden/make_golden.py; live when
oracle/_ref/ref_harness is present)."""
import os

import numpy as np
import pytest

from conftest import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _files(d, stem):
    return {k: open(os.path.join(d, stem + "." + e), "rb").read() for k, e in (("xyz", "xyz.raw"), ("attr", "attr.raw"), ("xdmf2", "xdmf2"))}


def test_dump_bytes_vs_golden(tmp_path):
    from cup2d_amd import dump as D
    from cup2d_amd.grid import BlockGrid
    G = golden("dump_n32.npz")
    n = G["vel"].shape[0]
    g = BlockGrid(n // 8, n // 8)
    # the reference ran as bpdx = bpdy = 1, levelStart = 2: h0 = 1/8, level 2
    D.dump(str(tmp_path / "vel"), float(G["time"]), g, g.to_blocks(G["vel"]), 1.0 / 8, level=2)
    mine = _files(str(tmp_path), "vel")
    for k in ("xyz", "attr", "xdmf2"):
        assert mine[k] == G[k].tobytes(), k
    # read back the way post.py does
    t, xyz, attr = D.read_dump(str(tmp_path / "vel.xdmf2"))
    assert t == float(G["time"]) and xyz.shape == (n * n, 4, 2) and attr.shape == (n * n, 3)
    h = np.float32(1.0 / n)
    assert np.allclose(xyz[:, 2, 0] - xyz[:, 0, 0], h) and np.allclose(xyz[:, 1, 1] - xyz[:, 0, 1], h)
    assert np.all(attr[:, 2] == 0)
    # cell (block 0, cell 0) sits at the origin in the reference's Hilbert order
    assert xyz[0, 0, 0] == 0 and xyz[0, 0, 1] == 0


@pytest.mark.parametrize("n,order", [(16, "hilbert"), (64, "hilbert")])
def test_dump_bytes_vs_reference_live(oracle, tmp_path, n, order):
    if not oracle.have_reference():
        pytest.skip("oracle/_ref/ref_harness not present")
    from cup2d_amd import dump as D
    from cup2d_amd.grid import BlockGrid
    vel = oracle.taylor_green(n, noise=0.3, seed=n)
    ref = oracle.ref_dump(vel, time=1.5)
    g = BlockGrid(n // 8, n // 8, order=order)
    k = int(np.log2(n // 8))
    D.dump(str(tmp_path / "vel"), 1.5, g, g.to_blocks(vel), 1.0 / 8, level=k)
    assert _files(str(tmp_path), "vel") == ref


@pytest.mark.gpu
def test_simulation_dump_gpu(gpu_lib, tmp_path):
    """Simulation.dump downloads the device-resident velocity and writes the reference's files"""
    import cup2d_amd
    from cup2d_amd import dump as D
    G = golden("dump_n32.npz")
    with cup2d_amd.Simulation(4) as s:
        s.vel = G["vel"]
        s.dump(str(tmp_path / "vel"), time=float(G["time"]), level=2)
    mine = _files(str(tmp_path), "vel")
    for k in ("xyz", "attr", "xdmf2"):
        assert mine[k] == G[k].tobytes(), k
    assert D.read_dump(str(tmp_path / "vel"))[2].shape == (1024, 3)


@pytest.mark.gpu
def test_cpp_host_driver_matches_python_mirror_gpu(tmp_path):
    """the C++ host driver (csrc/cup2d_run.cpp: grid construction, upload, cup2d_step loop with the reference's
    tolerance rule, dump()) against the Python mirror on the same initial field: same dt per step to the last bit,
    byte-identical dump files (Hilbert order, float32 geometry, XDMF text)"""
    import subprocess
    import cup2d_amd
    from oracle import oracle as O
    nx, ny, steps = 64, 40, 3
    vel = O.taylor_green(nx, noise=0.02, seed=5, ny=ny)
    init = str(tmp_path / "vel.f64")
    np.ascontiguousarray(vel, dtype=np.float64).tofile(init)
    exe = os.path.join(ROOT, "cup2d_amd", "cup2d_run")
    r = subprocess.run([exe, "-n", str(nx), "-ny", str(ny), "-steps", str(steps), "-init", init, "-dump", str(tmp_path / "cpp"),
                        "-maxiter", "300"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l.split() for l in r.stdout.decode().splitlines() if l.startswith("step ")]
    assert len(lines) == steps
    with cup2d_amd.Simulation(nx // 8, ny // 8) as s:
        s.vel = vel
        for k in range(steps):
            info = s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=300)
            assert float(lines[k][5]) == info["dt"] and int(lines[k][7]) == info["iters"], (k, lines[k], info)
        s.dump(str(tmp_path / "py"))
    for ext in (".xyz.raw", ".attr.raw"):
        a = open(str(tmp_path / ("cpp.%08d" % steps)) + ext, "rb").read()
        b = open(str(tmp_path / "py") + ext, "rb").read()
        assert a == b, ext
    xa = open(str(tmp_path / ("cpp.%08d.xdmf2" % steps))).read().replace("cpp.%08d" % steps, "py")
    assert xa == open(str(tmp_path / "py.xdmf2")).read()
