"""CPU tests of seam B1 (cup2d_amd/libcup2d_spmat.so, the reference's cuda.h classes on MI355X): exported
C++ symbols, the reference links against it, the make() protocol across MPI ranks, and the recognition
of the same-level stencil in the assembled triplets.  Nothing here needs a GPU."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPMAT = os.path.join(ROOT, "cup2d_amd", "libcup2d_spmat.so")
DRIVER = os.path.join(ROOT, "oracle", "_ref", "spmat_driver")
HARNESS_HIP = os.path.join(ROOT, "oracle", "_ref", "ref_harness_hip")
MPIEXEC = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"

# the out-of-line members declared at cuda.h:28-39, Itanium-mangled (MPI_Comm = int in MPICH)
CUDA_H_SYMBOLS = [
    "_ZN15LocalSpMatDnVecC1EiibRKSt6vectorIdSaIdEE", "_ZN15LocalSpMatDnVecD1Ev", "_ZN15LocalSpMatDnVec7reserveEi",
    "_ZN15LocalSpMatDnVec14cooPushBackValEdxx", "_ZN15LocalSpMatDnVec14cooPushBackRowERK9SpRowInfo",
    "_ZN15LocalSpMatDnVec4makeERKSt6vectorIxSaIxEE", "_ZN15LocalSpMatDnVec15solveWithUpdateEddi",
    "_ZN15LocalSpMatDnVec13solveNoUpdateEddi",
]

needs_spmat = pytest.mark.skipif(not os.path.exists(SPMAT), reason="libcup2d_spmat.so is built only where the reference header exists")


@needs_spmat
def test_spmat_exports_the_cuda_h_members():
    out = subprocess.check_output(["nm", "-D", "--defined-only", SPMAT], text=True)
    defined = {line.split()[-1] for line in out.splitlines() if line.strip()}
    for s in CUDA_H_SYMBOLS:
        assert s in defined, "libcup2d_spmat.so does not define %s" % s


@pytest.mark.skipif(not os.path.exists(HARNESS_HIP), reason="oracle/_ref/ref_harness_hip not built")
def test_reference_main_links_against_spmat():
    """the reference's main.cpp (compiled where it lies) takes LocalSpMatDnVec from our library"""
    und = subprocess.check_output(["nm", "-D", "--undefined-only", HARNESS_HIP], text=True)
    for s in CUDA_H_SYMBOLS:
        if s.endswith("D1Ev"):
            continue  # main.cpp never deletes sim.mat
        assert s in und, "%s is not imported by the reference binary" % s
    ldd = subprocess.check_output(["ldd", HARNESS_HIP], text=True)
    assert "libcup2d_spmat.so" in ldd and "libcup2d_hip.so" in ldd and "not found" not in ldd


@needs_spmat
@pytest.mark.parametrize("order", ["hilbert", "rowmajor"])
def test_stencil_recognition(order):
    from cup2d_amd.grid import BlockGrid
    L = ctypes.CDLL(SPMAT)
    f = L.cup2d_spmat_recognise_stencil
    f.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]

    def recognise(nb, r, c, v):
        r = np.ascontiguousarray(r, dtype=np.int32)
        c = np.ascontiguousarray(c, dtype=np.int32)
        v = np.ascontiguousarray(v, dtype=np.float64)
        nbr = np.full((nb, 4), -7, dtype=np.int32)
        ok = f(nb, r.size, r.ctypes.data, c.ctypes.data, v.ctypes.data, nbr.ctypes.data)
        return bool(ok), nbr

    g = BlockGrid(5, 3, order=order)
    r, c, v = g.poisson_coo()
    assert r.size == 5 * g.nblocks * 64 - 2 * (g.nx + g.ny)  # 5-point rows minus the wall neighbours
    perm = np.random.default_rng(0).permutation(r.size)  # list order must not matter
    ok, nbr = recognise(g.nblocks, r[perm], c[perm], v[perm])
    assert ok and np.array_equal(nbr, g.nbr)
    # anything that is not exactly the same-level stencil must fall through to the general operator
    v2 = v.copy(); v2[7] = 0.5
    assert not recognise(g.nblocks, r, c, v2)[0]
    assert not recognise(g.nblocks, r[1:], c[1:], v[1:])[0]            # a missing neighbour
    assert not recognise(g.nblocks, np.append(r, 0), np.append(c, 200), np.append(v, 1.0))[0]  # a far column
    k = int(np.flatnonzero(r != c)[0])
    assert not recognise(g.nblocks, np.append(r, r[k]), np.append(c, c[k]), np.append(v, v[k]))[0]  # duplicate


@pytest.mark.skipif(not (os.path.exists(DRIVER) and os.path.exists(MPIEXEC)), reason="spmat_driver / mpiexec missing")
@pytest.mark.parametrize("ranks,nbx,nby", [(1, 4, 4), (2, 5, 4), (3, 5, 4), (4, 3, 7)])
def test_make_protocol_across_ranks(ranks, nbx, nby):
    """LocalSpMatDnVec::make (cuda.cu:611-689 protocol): halo numbering, pack lists and localised
    triplets, checked by pushing global ids through the tables (tests/spmat_driver.cpp)"""
    out = subprocess.run([MPIEXEC, "-n", str(ranks), DRIVER, "make", str(nbx), str(nby)], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0 and "MAKE_OK" in out.stdout, out.stdout + out.stderr
    assert "violations 0" in out.stdout
