#!/usr/bin/env python3
"""oracle/b2_patch.py -- TEST INFRASTRUCTURE.  Proves seam B2 by compiling it.

Reads the reference's main.cpp where it lies (never copied into the repository) and writes oracle/_ref/b2/main.cpp
(git-ignored, built only where /root/reference exists): the same file with every block-operator CALL SITE INTEGRATION.md
lists wrapped as

      if (!b2_site_<name>(...)) {
        ... the reference's own lines, untouched ...
      }

oracle/ref_harness.cpp compiled with -DHARNESS_B2 -Ioracle/_ref/b2 #includes this copy instead of the original and
defines the b2_site_* functions: each returns false when its site is switched off (environment CUP2D_B2_SITES, a comma
list of vort,rk2,penal,rhs,solve,project; default all) -- the reference code then runs -- and otherwise serves the site
through include/cup2d_hip.h on the GPU and returns true.  So ONE binary, oracle/_ref/ref_harness_b2, is both the
reference's loop (no site on) and the loop with any subset of its call sites bound to this repository's C ABI:

  vort     main.cpp:4659       computeA<VectorLab>(KernelVorticity) in adapt()                       -> cup2d_vorticity
  rk2      main.cpp:6611-6642  prepare0 / computeA<KernelAdvectDiffuse> / fillcases + axpy, twice    -> cup2d_advect_diffuse_rk2
  penal    main.cpp:6643-7006  body momenta, velocity blend, tmpV = u_def (one body; else reference) -> cup2d_body_* / cup2d_penalize
  rhs      main.cpp:7007-7027  computeB<pressure_rhs>, pold = pres, pres = 0, computeA<pressure_rhs1> -> cup2d_poisson_rhs
  solve    main.cpp:7031-7119  matrix assembly, getVec, LocalSpMatDnVec::solve*                      -> cup2d_poisson_solve
  project  main.cpp:7120-7187  mean removal (twice), computeA<pressureCorrectionKernel>, V += tmpV/h^2 -> cup2d_project

Every region is found by anchor lines that must occur exactly where expected (the script fails loudly otherwise).
"""
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "b2")

src = open(os.path.join(REF, "main.cpp")).read().split("\n")


def find(text, start):
    for i in range(start, len(src)):
        if src[i] == text:
            return i
    raise SystemExit("b2_patch: anchor not found after line %d: %r" % (start + 1, text))


edits = []  # (first, last_exclusive, call, indent, lines inserted before the wrap)


def wrap(first, last_exclusive, call, indent, before=()):
    edits.append((first, last_exclusive, call, indent, list(before)))


# vorticity inside adapt()
a = find("static void adapt() {", 0)
v = find("  computeA<VectorLab>(KernelVorticity(), var.vel, 2);", a)
assert v - a <= 3
wrap(v, v + 1, "b2_site_vorticity()", "  ")

loop = find("      ongrid(sim.dt);", 0)
# RK2: vold = vel stays (main.cpp:6607-6610); the two advect-diffuse stages
s1 = find("      if (var.tmpV->UpdateFluxCorrection) {", loop)
e1 = find("      for (const auto &shape : sim.shapes) {", s1)
assert 20 < e1 - s1 < 40, (s1, e1)
# names the reference declares inside one site and uses in a later one: declared once ahead of the first wrap (the
# reference's own declarations then shadow them inside their block)
hoist = ["      std::vector<Info> &tmpVInfo = var.tmpV->infos;", "      std::vector<Info> &presInfo = var.pres->infos;",
         "      std::vector<Info> &poldInfo = var.pold->infos;"]
wrap(s1, e1, "b2_site_advect_diffuse_rk2()", "      ", hoist)
# penalisation: from the momentum loop to the deformation-velocity accumulation
sol = find("      const double max_error = sim.step < 10 ? 0.0 : sim.PoissonTol;", e1)
s2 = find("      if (var.tmp->UpdateFluxCorrection) {", e1)
assert 300 < s2 - e1 < 420, (e1, s2)
wrap(e1, s2, "b2_site_penalize()", "      ")
# Poisson right-hand side
assert 15 < sol - s2 < 30, (s2, sol)
wrap(s2, sol, "b2_site_poisson_rhs()", "      ")
# the solve
s3 = find("      if (var.pres->UpdateFluxCorrection) {", sol)
assert s3 - sol == 3
s4 = find("      std::vector<Info> &zInfo = var.pres->infos;", s3)
assert 60 < s4 - s3 < 120, (s3, s4)
wrap(s3, s4, "b2_site_solve(max_error, max_rel_error, max_restarts)", "      ")
# mean removal + projection
e4 = find("      computeB<KernelComputeForces, VectorLab, ScalarLab>(", s4)
assert 50 < e4 - s4 < 90, (s4, e4)
wrap(s4, e4, "b2_site_project()", "      ")

for first, last, call, indent, before in sorted(edits, reverse=True):
    body = src[first:last]
    src[first:last] = before + ["%sif (!%s) {" % (indent, call)] + body + ["%s}" % indent]

os.makedirs(OUT, exist_ok=True)
open(os.path.join(OUT, "main.cpp"), "w").write("\n".join(src))
print("b2_patch: wrote %s (%d call sites wrapped)" % (os.path.join(OUT, "main.cpp"), len(edits)))
