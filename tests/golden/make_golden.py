"""Generate tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref/ref_harness, i.e.
/root/reference/main.cpp compiled where it lies).  Run in the authoring container only:

    python tests/golden/make_golden.py

The fixtures pin the C restatement (oracle/cup2d_oracle.c) and the HIP kernels on machines where
/root/reference does not exist (the GPU box).  Inputs are seeded; everything is float64.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def functors_case(n, seed, nu, noise):
    rng = np.random.default_rng(seed)
    vel = O.taylor_green(n, noise=noise, seed=seed)
    pres = rng.uniform(-1, 1, (n, n))
    chi = rng.uniform(0, 1, (n, n))
    udef = rng.uniform(-1, 1, (n, n, 2))
    R = O.ref_functors(vel, nu, pres=pres, chi=chi, udef=udef)
    out = dict(n=n, nu=nu, vel=vel, pres=pres, chi=chi, udef=udef)
    out.update(R)
    return out


def main():
    assert O.have_reference(), "build oracle/_ref first: make -C oracle ref"
    # block functors: smooth+noise field, and a noise-dominated field (both upwind branches everywhere)
    np.savez_compressed(os.path.join(HERE, "functors_n32_tg.npz"), **functors_case(32, 20250117, 1e-3, 1e-3))
    np.savez_compressed(os.path.join(HERE, "functors_n32_noise.npz"), **functors_case(32, 7, 4e-5, 0.5))
    # Poisson: the matrix the reference assembles (main.cpp:7034-7112) applied to a vector, and a
    # tolerance solve by the CPU port of cuda.cu
    n = 32
    rng = np.random.default_rng(11)
    b = rng.uniform(-1, 1, (n, n))
    b -= b.mean()
    x0 = rng.uniform(-1, 1, (n, n))
    x, ax0, info = O.ref_solve(b, x0=x0, tol=1e-10, rel_tol=0.0, max_restarts=100)
    np.savez_compressed(os.path.join(HERE, "poisson_n32.npz"), n=n, b=b, x0=x0, x=x, Ax0=ax0,
                        iters=info["iters"], err=info["err"], err_init=info["err_init"])
    # the reference's own time loop, 3 steps from a Taylor-Green IC
    vel0 = O.taylor_green(32, noise=1e-3, seed=3)
    R = O.ref_run(vel0, 1e-3, steps=3, tol=1e-11, rel_tol=0.0, max_restarts=100)
    np.savez_compressed(os.path.join(HERE, "run_n32_3steps.npz"), n=32, nu=1e-3, cfl=0.5, vel0=vel0, vel=R["vel"],
                        pres=R["pres"], dts=np.array([s["dt"] for s in R["steps"]]),
                        vel_adv=np.stack([s["vel_adv"] for s in R["steps"]]), b=np.stack([s["b"] for s in R["steps"]]))
    # the reference's output writer dump() (main.cpp:3367-3466) on a 32^2 velocity field: raw bytes of the three files
    vd = O.taylor_green(32, noise=0.1, seed=5)
    D = O.ref_dump(vd, time=0.375)
    np.savez_compressed(os.path.join(HERE, "dump_n32.npz"), vel=vd, time=0.375, xyz=np.frombuffer(D["xyz"], dtype=np.uint8),
                        attr=np.frombuffer(D["attr"], dtype=np.uint8), xdmf2=np.frombuffer(D["xdmf2"], dtype=np.uint8))
    # block functors of the reference on an ADAPTED grid (three levels, 76 blocks): the grid its own adapt() builds around
    # an analytic vortex pair, halo-1 functors with their flux correction on analytic fields (ref_harness 'amr' reps=-1)
    np.savez_compressed(os.path.join(HERE, "amr_functors.npz"), **O.ref_amr_functors(2, 5, 4, 2.0, 0.5))
    # the reference's own adapt() on that grid with analytic fields: blocks and fields before / after
    pre, post = O.ref_amr_adapt(2, 5, 4, 2.0, 0.5)
    np.savez_compressed(os.path.join(HERE, "amr_adapt.npz"), pre_blocks=pre["blocks"], pre_vel=pre["vel"], pre_pres=pre["pres"],
                        post_blocks=post["blocks"], post_vel=post["vel"], post_pres=post["pres"], rtol=2.0, ctol=0.5, level_max=5)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
