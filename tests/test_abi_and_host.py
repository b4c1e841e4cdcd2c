"""CPU tests: the C-ABI library loads and exports every symbol include/cup2d_hip.h declares; host-side
grid logic; the product fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "cup2d_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cup2d_[a-z_A-Z0-9]+)\s*\(", txt)) - {"cup2d_exchange_fn", "cup2d_allreduce_fn"})


def test_library_exports_every_declared_symbol():
    import cup2d_amd
    from cup2d_amd import lib
    L = cup2d_amd.load_library()
    names = declared_symbols()
    assert len(names) >= 35
    for s in names:
        assert hasattr(L, s), "libcup2d_hip.so does not export %s" % s
    assert sorted(lib.SYMBOLS) == names
    assert b"gfx950" in L.cup2d_version()


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import cup2d_amd
    with pytest.raises(cup2d_amd.Cup2dError):
        cup2d_amd.Simulation(4)


def test_product_does_not_import_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "cup2d_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("# oracle-free", ""), "%s mentions the oracle" % f


def test_grid_roundtrip_and_neighbours():
    from cup2d_amd.grid import BlockGrid, WALL
    for order in ("hilbert", "rowmajor"):
        g = BlockGrid(8, 4, order=order)
        a = np.random.default_rng(0).uniform(size=(g.ny, g.nx, 2))
        assert np.array_equal(g.from_blocks(g.to_blocks(a), 2), a)
        s = np.random.default_rng(1).uniform(size=(g.ny, g.nx))
        assert np.array_equal(g.from_blocks(g.to_blocks(s), 1), s)
        for b in range(g.nblocks):
            x, y = g.coords[b]
            W, E, S, N = g.nbr[b]
            assert (W == WALL) == (x == 0) and (E == WALL) == (x == g.nbx - 1)
            assert (S == WALL) == (y == 0) and (N == WALL) == (y == g.nby - 1)
            if W != WALL:
                assert tuple(g.coords[W]) == (x - 1, y) and g.nbr[W][1] == b
            if N != WALL:
                assert tuple(g.coords[N]) == (x, y + 1) and g.nbr[N][2] == b
        assert g.n_inner == g.nblocks and g.nghost == 0


def test_hilbert_runs_are_compact_patches():
    from cup2d_amd.grid import BlockGrid
    g = BlockGrid(16, 16)
    for start in range(0, 256, 16):
        c = g.coords[start:start + 16]
        assert np.ptp(c[:, 0]) == 3 and np.ptp(c[:, 1]) == 3


def test_ghost_sides_order_halo_blocks_last():
    from cup2d_amd.grid import BlockGrid
    g = BlockGrid(4, 4, ghost_sides=(False, True, False, True))
    assert g.nghost == 8 and g.n_inner == 9
    touch = (g.coords[:, 0] == 3) | (g.coords[:, 1] == 3)
    assert not touch[:g.n_inner].any() and touch[g.n_inner:].all()
    assert (g.nbr[g.n_inner:] >= g.nblocks).any(axis=1).all()


def test_cpp_host_driver_fails_loudly_without_gpu():
    """cup2d_amd/cup2d_run (csrc/cup2d_run.cpp, the C++ host driver over the C ABI) has no CPU path: without a GPU it
    stops at cup2d_create with the library's error text; bad options are rejected before that"""
    import subprocess
    exe = os.path.join(ROOT, "cup2d_amd", "cup2d_run")
    assert os.path.exists(exe), "build with __graft_entry__.build()"
    r = subprocess.run([exe, "-n", "12"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 2 and b"multiples of 8" in r.stderr
    r = subprocess.run([exe, "-bogus", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 2 and b"unknown option" in r.stderr
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "-n", "16", "-steps", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 1 and b"cup2d_create" in r.stderr and b"step 1" not in r.stdout


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus N` starts its own ranks; on a node with fewer GPUs it says so and exits 2 (no hang, no
    partial launch).  Here: no GPU at all."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("a node with two GPUs launches the ranks for real")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=env, timeout=300)
    assert r.returncode == 2 and b"GPU(s) are visible" in r.stderr and not r.stdout.strip()


def test_a_gpu_test_that_aborts_costs_one_test_not_the_session(tmp_path):
    """tests/conftest.py runs the GPU tests of the heavy files one per child process and announces every test on the real
    stdout / stderr: a session in which one such test dies with SIGABRT (what a GPU memory fault is for the process that owns
    the context) reports THAT test as failed -- with the signal -- and goes on to the next one.  Round 2's driver run lost all
    104 tests to one abort."""
    import shutil
    import subprocess
    import sys
    d = tmp_path / "tests"
    d.mkdir()
    shutil.copy(os.path.join(ROOT, "tests", "conftest.py"), d / "conftest.py")
    (d / "test_amr.py").write_text(
        "import os, pytest\n"
        "@pytest.mark.gpu\n"
        "def test_dies():\n    os.abort()\n"
        "@pytest.mark.gpu\n"
        "def test_lives():\n    assert True\n"
        "def test_not_gpu():\n    assert False, 'deselected by -m gpu'\n")
    env = dict(os.environ)
    env.pop("CUP2D_TEST_CHILD", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/", "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=tmp_path, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode()
    assert r.returncode == 1, out[-3000:]
    assert "1 failed, 1 passed, 1 deselected" in out, out[-3000:]
    assert "[cup2d] start tests/test_amr.py::test_dies" in out and "[cup2d] start tests/test_amr.py::test_lives" in out
    assert "killed by signal 6" in out and "FAILED in its child process" in out, out[-3000:]
