#!/usr/bin/env python3
"""oracle/b2_patch.py -- TEST INFRASTRUCTURE.  Proves seam B2 by compiling it.

Reads the reference's main.cpp where it lies (never copied into the repository), replaces the block-operator CALL SITES
INTEGRATION.md lists with calls into this repository's C ABI (include/cup2d_hip.h) and writes the result to
oracle/_ref/b2/main.cpp (git-ignored, built only where /root/reference exists).  oracle/ref_harness.cpp compiled with
-DHARNESS_B2 -Ioracle/_ref/b2 then #includes the patched file instead of the original: oracle/_ref/ref_harness_b2 is the
reference's own time loop with

  main.cpp:6611-6642  prepare0 / computeA<VectorLab>(KernelAdvectDiffuse) / fillcases + axpy, twice -> cup2d_advect_diffuse_rk2
  main.cpp:7003-7027  computeB<pressure_rhs>, pold = pres, pres = 0, computeA<pressure_rhs1>         -> cup2d_poisson_rhs
  main.cpp:7031-7119  matrix assembly, getVec, LocalSpMatDnVec::solveWithUpdate / solveNoUpdate         -> cup2d_poisson_solve
  main.cpp:7120-7187  mean removal (twice), computeA<pressureCorrectionKernel>, V += tmpV / h^2           -> cup2d_project
  main.cpp:4659       computeA<VectorLab>(KernelVorticity) in adapt()                                  -> cup2d_vorticity

Every region is found by an anchor line that must occur exactly where expected (the script fails loudly otherwise).
The b2_site_* functions are defined in ref_harness.cpp (#ifdef HARNESS_B2): each puts the fields the call site reads on
the device, runs the C-ABI call and gets back what the following host code reads -- the call-site form of the binding;
keeping fields resident between sites is an optimisation of the same calls.
"""
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "b2")

src = open(os.path.join(REF, "main.cpp")).read().split("\n")


def find(text, start, exact=True):
    for i in range(start, len(src)):
        if (src[i] == text) if exact else (text in src[i]):
            return i
    raise SystemExit("b2_patch: anchor not found after line %d: %r" % (start + 1, text))


def replace(first, last_exclusive, lines):
    """replace src[first:last_exclusive] keeping the line count (later anchors keep their reference line numbers)"""
    n = last_exclusive - first
    assert len(lines) <= n
    src[first:last_exclusive] = lines + ["// (seam B2: reference lines replaced above)"] * (n - len(lines))


# vorticity inside adapt()
a = find("static void adapt() {", 0)
v = find("  computeA<VectorLab>(KernelVorticity(), var.vel, 2);", a)
assert v - a <= 3
replace(v, v + 1, ["  b2_site_vorticity();"])

loop = find("      ongrid(sim.dt);", 0)
# RK2: keep vold = vel (main.cpp:6607-6610), replace the two advect-diffuse stages
s1 = find("      if (var.tmpV->UpdateFluxCorrection) {", loop)
e1 = find("      for (const auto &shape : sim.shapes) {", s1)
assert 20 < e1 - s1 < 40, (s1, e1)
replace(s1, e1, ["      b2_site_advect_diffuse_rk2();"])
# Poisson right-hand side
sol = find("      const double max_error = sim.step < 10 ? 0.0 : sim.PoissonTol;", e1)
s2 = find("      if (var.tmp->UpdateFluxCorrection) {", e1)
assert 15 < sol - s2 < 30, (s2, sol)
replace(s2, sol, ["      b2_site_poisson_rhs();"])
# the solve
s3 = find("      if (var.pres->UpdateFluxCorrection) {", sol)
assert s3 - sol == 3
s4 = find("      std::vector<Info> &zInfo = var.pres->infos;", s3)
assert 60 < s4 - s3 < 120, (s3, s4)
replace(s3, s4, ["      b2_site_solve(max_error, max_rel_error, max_restarts);"])
# mean removal + projection
e4 = find("      computeB<KernelComputeForces, VectorLab, ScalarLab>(", s4)
assert 50 < e4 - s4 < 90, (s4, e4)
replace(s4, e4, ["      b2_site_project();"])

os.makedirs(OUT, exist_ok=True)
open(os.path.join(OUT, "main.cpp"), "w").write("\n".join(src))
print("b2_patch: wrote %s (5 call sites replaced)" % os.path.join(OUT, "main.cpp"))
