#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: Mcell-updates/s of CUP2D's stencil hot path.

One STEP = one pass of the reference's time-loop body (main.cpp:6576-7187, body-free) over the whole
grid: dt (max|u| reduction) -> RK2 WENO5 advect-diffuse (2 fused stages) -> Poisson right-hand side ->
block-Jacobi BiCGSTAB capped at --iters iterations (zero tolerances, like the reference's first ten
steps, main.cpp:7028-7030; BASELINE.json configs[1] words this "50 pressure iters/step") -> mean removal
+ pressure-gradient projection.  value = cells * steps / seconds / 1e6, whole job, inputs resident in HBM.

N = 1 : 4096^2 uniform grid (BASELINE.json configs[2], the headline config).
N > 1 : `python bench.py --gpus N` starts its N ranks itself (torch.distributed.run, one process per GPU; under
        torch.distributed.run it is a rank).  Layout "weak" (default, the series whose N = 1 point is the headline
        config): each rank owns a 4096^2-cell patch of a px x py Cartesian decomposition.  Layout "configs3":
        BASELINE.json configs[3], 8192^2 cells GLOBAL split px x py (8 GPUs: 2 x 4, 4096 x 2048 cells per rank) --
        strong scaling.  Whichever is not --layout is measured too and reported in "second_layout".
        Face halos are packed by HIP kernels and exchanged by the library's own RCCL communicator (csrc/comm.hip:
        ncclSend/ncclRecv on a second stream, all-reduce / all-gather on the compute stream); torch.distributed
        (gloo) only hands out the rendezvous token, the barriers and the max over ranks of the wall time.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (the dominant kernel of the timed region by GPU
time, timed live with HIP events on its launch stream), "roofline_north_star" (the fused WENO5
advect-diffuse stage, the kernel BASELINE.json's target names), "roofline_all" (every kernel family),
"kernels" (per-family GPU time), "verified" (a post-run check: the residual the solver reports is the residual of
the fields it leaves), "cpu_baseline" (the reference's own loop on the host cores, bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEFAULT_SOLVER = "fused"     # the library's default; "sweeps" = the five-sweep organisation
DEFAULT_FINISH = "kernel"
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_CEILING_GBS = 6290.0  # MI355X_MICROARCH.md: the measured copy ceiling (what a streaming kernel's ACTUAL bytes can move at)
FP64_PEAK_TFLOPS = 78.6      # 256 CU x 64 FMA/clk x 2 x 2.4 GHz
CONFIGS3_N = 8192            # BASELINE.json configs[3]: 8192^2 cells, global
# Test-only: CUP2D_BENCH_SHARE_GPU=1 lets the N ranks of `--gpus N` share GPU 0 of a one-GPU box (RCCL cannot connect two
# ranks on one device, so the transport is torch.distributed's gloo, host-staged, behind cup2d_set_comm): every line of this
# file that an 8-rank run executes -- both layouts, the JSON merge, the comm block, the watchdog -- runs before the driver's
# one 8-GPU run does (tests/test_bench_world8.py).  The line it prints says "shared_gpu": true and is not a measurement.
SHARE_GPU = os.environ.get("CUP2D_BENCH_SHARE_GPU", "0") == "1"
SETUP_STEPS = 20  # whole steps at the end of the setup of a one-GPU run, in front of the --warmup steps (main())
MIN_ROOFLINE_LAUNCHES = 100  # a per-kernel roofline is reported from at least this many event-timed launches


def synthetic_velocity(nx, ny, gx0, gy0, gnx, gny, seed):
    """Taylor-Green + 1e-3 noise on the global unit square, evaluated on this rank's patch."""
    h = 1.0 / max(gnx, gny)
    x = (gx0 + np.arange(nx) + 0.5) * h
    y = (gy0 + np.arange(ny) + 0.5) * h
    X, Y = np.meshgrid(x, y, indexing="xy")
    rng = np.random.default_rng(seed)
    vel = np.empty((ny, nx, 2))
    vel[..., 0] = np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y)
    vel[..., 1] = -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y)
    vel += 1e-3 * rng.uniform(-1.0, 1.0, vel.shape)
    return vel


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=4096, help="cells per side of one rank's patch (layout weak)")
    ap.add_argument("--configs3-n", type=int, default=CONFIGS3_N, help="global cells per side of layout configs3 (tests shrink it)")
    ap.add_argument("--layout", default="weak", choices=["weak", "configs3"],
                    help="weak: --n^2 cells per rank; configs3: 8192^2 cells global over px x py ranks (BASELINE.json configs[3])")
    ap.add_argument("--no-second-layout", action="store_true", help="N > 1: do not also measure the other layout")
    ap.add_argument("--iters", type=int, default=50, help="BiCGSTAB iterations per step")
    ap.add_argument("--math", default="fast", choices=["fast", "strict"])
    ap.add_argument("--solver", default=DEFAULT_SOLVER, choices=["sweeps", "fused"],
                    help="organisation of a BiCGSTAB iteration (include/cup2d_hip.h cup2d_solver_kind)")
    ap.add_argument("--finish", default=DEFAULT_FINISH, choices=["launch", "kernel"],
                    help="reduction finish + scalar update: own launch, or by the last workgroup of the sweep")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch"],
                    help="N > 1 transport: the library's own RCCL communicator, or torch.distributed behind the callbacks")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the domain-decomposed code path (communicator, comm stream) even with one rank: a smoke "
                         "test of the N > 1 path on a single-GPU box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--cpu-n", type=int, default=1024, help="grid of the bounded CPU-baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="OpenMP threads of the CPU baseline (0 = sweep 16 / 64 / 128 and report the best)")
    ap.add_argument("--profile-tag", default="", help="profiles/<tag>_pmc_traffic.json supplies roofline.traffic (default: the newest "
                                                      "rNN tag without a suffix)")
    ap.add_argument("--cpu-functor-n", type=int, default=4096, help="grid of the reference-functor CPU baseline (the headline size)")
    ap.add_argument("--no-nrank-proxy", action="store_true", help="skip the N-rank-path leg (a self-periodic patch through RCCL on this GPU)")
    ap.add_argument("--no-north-star-floors", action="store_true", help="skip the timing-only builds of the north-star kernel")
    ap.add_argument("--no-tolerance-leg", action="store_true", help="skip the tolerance-terminated steps (run.sh:13-14)")
    ap.add_argument("--no-second-size", action="store_true", help="skip the 2048^2 leg (BASELINE.json configs[1])")
    ap.add_argument("--no-amr", action="store_true", help="skip the block-AMR leg (BASELINE.json configs[4] shape, one GPU)")
    ap.add_argument("--amr-lfine", type=int, default=9, help="finest AMR level: 2^L blocks per side (9 = 4096^2-equivalent)")
    # (ranks started by spawn_ranks get their arguments through the environment: torch.distributed.run's argparse rejects
    # "--n" in front of it as an ambiguous abbreviation of its own options, wherever on the command line it stands)
    argv = json.loads(os.environ["CUP2D_BENCH_ARGV"]) if "CUP2D_BENCH_ARGV" in os.environ and "WORLD_SIZE" in os.environ else None
    return ap.parse_args(argv)


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks (one process per GPU) and pass their
    output through.  Exits non-zero with a clear message when the node has fewer GPUs than asked."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and not (SHARE_GPU and have >= 1):
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) are visible on this node\n" % (args.gpus, have))
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    env = dict(os.environ)
    env["CUP2D_BENCH_ARGV"] = json.dumps(sys.argv[1:])
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL between processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.exit(subprocess.call(cmd, env=env))


class Watchdog:
    """A rank that makes no progress for `limit` seconds (a peer died, a collective never completes) ends itself with a
    message instead of holding the GPU until the caller's timeout: torch.distributed.run then stops the other ranks."""

    def __init__(self, limit):
        import threading
        self.limit, self.last, self.what = limit, time.monotonic(), "start"
        threading.Thread(target=self._watch, daemon=True).start()

    def beat(self, what):
        self.last, self.what = time.monotonic(), what

    def _watch(self):
        while True:
            time.sleep(2.0)
            idle = time.monotonic() - self.last
            if idle > self.limit:
                sys.stderr.write("bench.py: rank %s made no progress for %.0f s after '%s' -- giving up\n"
                                 % (os.environ.get("RANK", "0"), idle, self.what))
                sys.stderr.flush()
                os._exit(4)


WATCHDOG = None


def beat(what):
    if WATCHDOG is not None:
        WATCHDOG.beat(what)


class Runner:
    """one layout on this rank: the simulation, its timed region and its timers"""

    def __init__(self, args, nx, ny, px, py, rank, world, local_rank, dist, ctl):
        import cup2d_amd
        self.args, self.dist, self.ctl, self.world, self.rank = args, dist, ctl, world, rank
        self.nx, self.ny, self.px, self.py = nx, ny, px, py
        cx, cy = rank % px, rank // px
        self.comm_kind = "none"
        if dist is not None:
            from cup2d_amd.distributed import DistributedSimulation
            kind = args.comm
            try:
                self.sim = DistributedSimulation(nx // 8, ny // 8, px, py, nu=1e-3, cfl=0.5, device=local_rank,
                                                 comm=kind, mode="staged" if SHARE_GPU else "device", group=ctl.get("nccl"))
            except Exception as e:
                # the in-library communicator is the product path: its failure (init or self-test) is an error, not a reason
                # to measure something else under the same name.  `--comm torch` asks for the other transport explicitly.
                sys.stderr.write("bench.py: rank %d: the in-library RCCL communicator failed: %s\n"
                                 "bench.py: (`--comm torch` runs torch.distributed's nccl backend behind cup2d_set_comm instead)\n"
                                 % (rank, e))
                sys.stderr.flush()
                os._exit(5)
            self.comm_kind = kind
            self.par = "cart%dx%d" % (px, py)
        else:
            self.sim = cup2d_amd.Simulation(nx // 8, ny // 8, nu=1e-3, cfl=0.5, device=local_rank)
            self.par = "single"
        vel = synthetic_velocity(nx, ny, cx * nx, cy * ny, px * nx, py * ny, seed=20250117 + rank)
        self.sim.set_math(args.math == "strict")
        self.sim.set_solver(fused=args.solver == "fused", finish_in_kernel=args.finish == "kernel")
        self.sim.vel = vel

    def sync(self):
        import torch
        self.sim.synchronize()
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier(group=self.ctl["gloo"])
            self.sim.synchronize()
            torch.cuda.synchronize()

    def one_step(self):
        r = self.sim.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=self.args.iters)
        beat("step")
        return r

    def timed_steps(self, steps, timing_mode):
        import torch
        self.sim.set_timing(timing_mode)
        self.sync()
        t0 = time.perf_counter()
        its = 0
        for _ in range(steps):
            its += self.one_step()["iters"]
        self.sync()
        el = time.perf_counter() - t0
        if self.dist is not None:  # the slowest rank's clock
            t = torch.tensor([el], dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.ctl["gloo"])
            el = float(t.item())
        return el, its

    def verify(self):
        """Post-run check on the fields the timed steps left.  (1) Set up the next step's Poisson system, run the capped
        solve, and recompute max|b - A x| from the fields (cup2d_poisson_residual): it must be the residual the solver
        reported for the iterate it returned, everything finite, not above the initial residual.  The reference returns the
        best iterate in the max norm (cuda.cu:535-547); within the capped iterations that may still be the initial guess, and
        (1) then says nothing about the iterations.  So, on one GPU: (2) the LAST iterate of that solve -- what the timed
        iterations computed -- is taken from the solver (cup2d_solver_last_iterate), max|b - A x_last| is recomputed from
        the field and must agree with the max norm of the recurrence residual r the solver carried through all iterations
        (they drift apart only by accumulated round-off: 1e-6 relative).  (3) The same system is solved for 8 iterations by the
        timed organisation and by the five-sweep one (krylov.hip, the solver every test pins to the oracle) and the two last
        iterates must agree to 1e-10 of max|x| (zero-tolerance BiCGSTAB amplifies round-off differences from iteration to
        iteration: at 50 iterations two correct organisations differ visibly, at 8 they do not)."""
        from cup2d_amd import lib as L
        s = self.sim
        dt = s.compute_dt()
        s.advect_diffuse_rk2(dt)
        s.fill(L.PRES, 0.0)
        s.poisson_rhs(dt)
        one_gpu = self.dist is None
        if one_gpu:
            s.keep_last_iterate(True)
        r = s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=self.args.iters)
        true = s.poisson_residual()
        umax = s.max_abs_vel()
        best_is_x0 = bool(r["err"] == r["err_init"])
        ok = bool(np.isfinite([dt, r["err"], r["err_init"], true, umax]).all() and r["err"] <= r["err_init"]
                  and abs(true - r["err"]) <= 1e-6 * r["err"] + 1e-9)
        last = None
        if one_gpu:
            try:
                kind = s.last_solver()
                rec = s.last_iterate_to(L.POLD)       # recurrence max|r| after the last iteration
                lib_ = s.L
                L.check(lib_.cup2d_copy_field(s.ctx, L.PRES, L.POLD), "copy_field")  # PRES <- x_last
                res_last = s.poisson_residual()       # max|b - A x_last| from the field (overwrites POLD)
                xs = {}
                for fused in (kind == "fused", False):
                    s.set_solver(fused=fused, finish_in_kernel=self.args.finish == "kernel" if fused else True)
                    s.fill(L.PRES, 0.0)
                    r8 = s.poisson_solve(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=8)
                    s.last_iterate_to(L.POLD)
                    xs[fused] = (s.pold, r8["iters"], s.last_solver())
                s.set_solver(fused=(kind == "fused"), finish_in_kernel=self.args.finish == "kernel")
                s.keep_last_iterate(False)
                xa, xb = xs[kind == "fused"], xs[False]
                scale = float(np.abs(xb[0]).max())
                diff = float(np.abs(xa[0] - xb[0]).max())
                last = {"iterations_checked": r["iters"], "recurrence_residual_after_last_iteration": rec,
                        "residual_of_last_iterate_recomputed": res_last,
                        "relative_gap": abs(res_last - rec) / rec if rec > 0 else None,
                        "eight_iterations": {"organisations": [xa[2], xb[2]], "iters": [xa[1], xb[1]], "max_abs_x": scale,
                                             "max_abs_difference": diff, "relative": diff / scale if scale > 0 else None,
                                             "tolerance": 1e-10}}
                last["ok"] = bool(np.isfinite([rec, res_last, scale, diff]).all() and rec > 0 and scale > 0
                                  and abs(res_last - rec) <= 1e-6 * rec and diff <= 1e-10 * scale and xa[1] == xb[1] == 8)
                ok = ok and last["ok"]
            except Exception as e:
                last = {"error": str(e)[:300], "ok": False}
                ok = False
        return {"ok": ok, "residual_reported": r["err"], "residual_recomputed": true, "residual_initial": r["err_init"],
                "best_iterate_is_initial_guess": best_is_x0,
                "iters": r["iters"], "max_abs_vel": umax, "last_iterate": last,
                "check": "max|b - A x| recomputed from the fields (cup2d_poisson_residual) == the solver's reported Linf "
                         "residual within 1e-6 relative, not above the initial residual, all finite; one GPU: max|b - A x_last| "
                         "recomputed from the LAST iterate == the recurrence residual after the last iteration within 1e-6 "
                         "relative, and 8 iterations of the timed organisation == 8 iterations of the five-sweep solver to "
                         "1e-10 of max|x|",
                "note": ("best_iterate_is_initial_guess: no iterate beat x0 = 0 in the max norm within the capped iterations, so "
                         "every timed solve RETURNS x0 as the reference would (cuda.cu:535-547) -- the iterations are computed in "
                         "full, and the last_iterate block is what checks them") if best_is_x0 else None}

    def close(self):
        self.sim.close()


def north_star_floors(sim, args, nsteps=8):
    """The two floors of the fused WENO5 stage kernel, measured the way the timed region measures the kernel itself: whole steps
    of the bench's context, back to back (no call between them: the step takes its dt from the maxima the projection left, as
    in the timed region), sampled per-kernel HIP events with the launches outside the solver in every step.  Under
    cup2d_debug_walk_knockout a stage launches the knocked-out kernel first -- under the stage's timer, on the stage's inputs,
    behind the launches of the step, writing to a scratch slab -- and the product kernel behind it, untimed: the simulation
    goes on unchanged.  Legs: product, arithmetic alone (no loads / stores inside the loop), memory skeleton (no arithmetic),
    product again."""
    from cup2d_amd import lib as L
    i1, i2 = L.TIMER_NAMES.index("advect_stage"), L.TIMER_NAMES.index("advect_stage2")
    out = {}
    try:
        for tag, ko in (("product", 0), ("arithmetic_alone", 2), ("memory_skeleton", 1), ("product_again", 0)):
            sim.debug_walk_knockout(ko)
            sim.set_timing(3)
            for _ in range(nsteps):
                sim.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=args.iters)
            (m1, n1), (m2, n2) = sim.get_timing(i1), sim.get_timing(i2)
            sim.set_timing(0)
            out[tag] = {"stage1": round(1e3 * m1 / max(1, n1), 2), "stage2": round(1e3 * m2 / max(1, n2), 2), "launches": n1 + n2}
            beat("north-star floors")
    finally:
        sim.debug_walk_knockout(0)
    out["how"] = ("us per launch, HIP events on the launch stream, %d whole steps per leg back to back in the bench's context; a knocked-out "
                  "launch runs in front of the (untimed) product launch of its stage" % nsteps)
    return out


def tolerance_leg(step, sync, nsteps=5):
    """What a user of the reference waits for after step 10 (run.sh:13-14, main.cpp:7028-7030): the same step with the solve
    ended by -poissonTol 1e-3 -poissonTolRel 1e-2 -maxPoissonRestarts 0 (cap 1000 = -maxPoissonIterations) instead of by the cap
    of 50 -- a few (10-40) iterations and then the solve's fixed costs and whatever the host had enqueued behind convergence."""
    recs = []
    for _ in range(nsteps):
        sync()
        t0 = time.perf_counter()
        r = step(tol=1e-3, rel_tol=1e-2, max_restarts=0, max_iter=1000)
        sync()
        recs.append({"iters": r["iters"], "ms": round((time.perf_counter() - t0) * 1e3, 3), "err": r["err"]})
    ms = sorted(x["ms"] for x in recs)
    return {"tolerances": "-poissonTol 1e-3 -poissonTolRel 1e-2 -maxPoissonRestarts 0 -maxPoissonIterations 1000 (run.sh:13-14)",
            "steps": recs, "iters_median": sorted(x["iters"] for x in recs)[len(recs) // 2], "ms_per_step_median": ms[len(ms) // 2]}


def _proxy_velocity(nbx, nby):
    nxp, nyp = nbx * 8, nby * 8
    hh = 1.0 / max(nxp, nyp)
    X, Y = np.meshgrid((np.arange(nxp) + 0.5) * hh, (np.arange(nyp) + 0.5) * hh, indexing="xy")
    vel0 = np.stack([np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y), -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y)], -1)
    return vel0 + 1e-3 * np.random.default_rng(20250117).uniform(-1.0, 1.0, vel0.shape)


def _proxy_time(args, s, nst):
    for _ in range(2):
        s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=args.iters)
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(nst):
        r = s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=args.iters)
    s.synchronize()
    return (time.perf_counter() - t0) / nst, r


def nrank_proxy_patch(args, device, nbx, nby, axes, plain_s_per_step, with_tolerance_leg=False):
    """one self-periodic patch (axes "x": ghost blocks W and E; "xy": on all four sides, the shape of an interior rank) timed
    next to the plain context of the same size (plain_s_per_step None: timed here)"""
    import ctypes
    import cup2d_amd
    from cup2d_amd import lib as L
    from cup2d_amd.distributed import self_periodic_simulation
    nst = max(3, min(args.steps, 10))
    vel = _proxy_velocity(nbx, nby)
    if plain_s_per_step is None:
        with cup2d_amd.Simulation(nbx, nby, nu=1e-3, cfl=0.5, device=device) as p:
            p.set_math(args.math == "strict")
            p.set_solver(fused=args.solver == "fused", finish_in_kernel=args.finish == "kernel")
            p.vel = vel
            plain_s_per_step, _ = _proxy_time(args, p, nst)
    s, g = self_periodic_simulation(nbx, nby, nu=1e-3, cfl=0.5, device=device, axes=axes)
    with s:
        s.set_math(args.math == "strict")
        s.set_solver(fused=args.solver == "fused", finish_in_kernel=args.finish == "kernel")
        s.vel = vel
        del vel
        el, r = _proxy_time(args, s, nst)
        form = s.last_solver_form()
        # the cost of an iteration on the N-rank path, measured inside THIS context: the period of an iteration = (step of
        # --iters iterations - step of 10) / (--iters - 10), and what of it is not inside the two sweeps (their HIP-event averages
        # in the same context): pack launches, RCCL kernels, idle stream between them
        inside = None
        try:
            if args.iters > 20:
                a10 = argparse.Namespace(**vars(args))
                a10.iters = 10
                el10, _ = _proxy_time(a10, s, nst)
                period = (el - el10) / (args.iters - 10)
                s.set_timing(2)
                for _ in range(4):
                    s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=args.iters)
                tc, te = s.get_timing(L.TIMER_NAMES.index("sweep_C")), s.get_timing(L.TIMER_NAMES.index("sweep_EA"))
                s.set_timing(0)
                if tc[1] and te[1]:
                    sw = (tc[0] / tc[1] + te[0] / te[1]) * 1e-3
                    inside = {"period_us": round(period * 1e6, 1), "sweeps_us": round(sw * 1e6, 1), "outside_us": round((period - sw) * 1e6, 1)}
        except Exception as e:  # informative
            inside = {"error": str(e)[:120]}
        tol_leg = tolerance_leg(s.step, s.synchronize, 3) if with_tolerance_leg else None
        n, p, ex, ar, ag = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong()
        L.check(s.L.cup2d_comm_stats(s.ctx, ctypes.byref(n), ctypes.byref(p), ctypes.byref(ex), ctypes.byref(ar), ctypes.byref(ag)), "comm_stats")
        L.check(s.L.cup2d_comm_finalize(s.ctx), "comm_finalize")
    return {"blocks": "%dx%d" % (nbx, nby), "ghost_sides": {"x": "WE", "y": "SN", "xy": "WESN"}[axes], "peers": p.value,
            "ms_per_step": round(el * 1e3, 3), "value": round(nbx * nby * 64 / el / 1e6, 2), "unit": "Mcell-updates/s",
            "plain_context_ms_per_step": round(plain_s_per_step * 1e3, 3), "ratio_to_plain": round(el / plain_s_per_step, 4),
            "fixed_us_per_iteration_over_plain": round((el - plain_s_per_step) / max(1, r["iters"]) * 1e6, 1),
            "iteration_period_us": (inside or {}).get("period_us"), "sweeps_us_per_iteration": (inside or {}).get("sweeps_us"),
            "outside_the_sweeps_us_per_iteration": (inside or {}).get("outside_us"), "inside_error": (inside or {}).get("error"),
            "form": "%s merge %d" % (form[0], form[1]),
            "iters": r["iters"], "solver_form": list(form), "ghost_blocks": g.nghost, "halo_set_patch": g.halo_tile,
            "n_inner": g.n_inner, "exchanges": ex.value, "allgathers": ag.value, "solve_to_tolerance": tol_leg}


def nrank_proxy_leg(args, device, nbx, nby, plain_elapsed, plain_steps):
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    plain = plain_elapsed / plain_steps
    # the headline patch with ghost blocks on all four sides (an interior rank; the 2 x 4 layout's ranks have two or three), the
    # same with the two x sides only (round 4's figure), and BASELINE.json configs[3]'s per-rank patch (4096 x 2048 cells) on
    # four sides next to a plain context of that size
    out = nrank_proxy_patch(args, device, nbx, nby, "xy", plain, with_tolerance_leg=True)
    out["what"] = ("the N-rank code path of the step on one GPU: a patch that is its own W, E, S and N neighbour (ghost blocks on "
                   "all four sides, four send/recv pairs per ncclGroup to self with the reduction records riding in the group, the scalar "
                   "update in the consumer sweeps: k_edge MERGE 3)")
    out["timeline"] = "profiles/r06_nrank_timeline.txt"
    others = []
    plain_half = [None]
    for (bx, by, axes, pl) in ((nbx, nby, "x", plain), (nbx, max(1, nby // 2), "xy", None), (nbx, max(1, nby // 2), "y", "half")):
        try:
            if pl == "half":  # (the plain context of that size was timed for the patch before)
                pl = plain_half[0]
            others.append(nrank_proxy_patch(args, device, bx, by, axes, pl))
            if axes == "xy" and by != nby:
                plain_half[0] = others[-1]["plain_context_ms_per_step"] * 1e-3
        except Exception as e:  # informative
            others.append({"blocks": "%dx%d" % (bx, by), "error": str(e)[:200]})
    out["other_patches"] = others
    # BASELINE.json configs[3] on paper: 8 ranks of 4096 x 2048 cells against ONE GPU on the whole 8192^2 grid (timed here), before a
    # single link is crossed -- the ceiling of the strong-scaling figure the library's own per-rank costs allow
    try:
        import cup2d_amd
        n8 = 2 * nbx
        with cup2d_amd.Simulation(n8, n8, nu=1e-3, cfl=0.5, device=device) as big:
            big.set_math(args.math == "strict")
            big.set_solver(fused=args.solver == "fused", finish_in_kernel=args.finish == "kernel")
            big.vel = _proxy_velocity(n8, n8)
            el8, _ = _proxy_time(args, big, 3)
        one = n8 * n8 * 64 / el8 / 1e6
        rank4, rank2 = others[1].get("value"), others[2].get("value") if len(others) > 2 else None
        # (a rank of the 2 x 4 layout has one neighbour in x and one or two in y: 768 or 1280 ghost blocks in two or three messages; the
        # patch with ghost blocks on all four sides has 1536 in four, the one with its two long sides 1024 in two)
        out["configs3_on_paper"] = {"one_gpu_%dx%d_cells" % (n8 * 8, n8 * 8): round(one, 1),
                                    "per_rank_patch_%s_four_sides" % others[1].get("blocks"): rank4,
                                    "per_rank_patch_%s_two_long_sides" % others[1].get("blocks"): rank2,
                                    "eight_ranks_over_one_gpu": [round(8 * v / one, 2) if v else None for v in (rank4, rank2)]}
    except Exception as e:  # informative
        out["configs3_on_paper"] = {"error": str(e)[:120]}
    # the weak series on paper (4096^2 cells per rank, the bench's default layout at N > 1): 8 ranks of the headline patch against the
    # plain context of the same size = 8 / ratio, with ghost blocks on four sides and on two
    try:
        out["weak_on_paper"] = {"eight_ranks_over_one_gpu": [round(8.0 / out["ratio_to_plain"], 2), round(8.0 / others[0]["ratio_to_plain"], 2)],
                                "patches": ["%s_%s" % (out["blocks"], out["ghost_sides"]), "%s_%s" % (others[0]["blocks"], others[0]["ghost_sides"])]}
    except Exception as e:  # informative
        out["weak_on_paper"] = {"error": str(e)[:120]}
    return out


def amr_leg(args, device):
    from cup2d_amd import amr as A, lib as L
    t0 = time.perf_counter()
    g = A.circle_band_grid(args.amr_lfine)
    t_grid = time.perf_counter() - t0
    with A.AmrSimulation(g, nu=1e-3, cfl=0.5, device=device) as s:
        xc, yc = g.cell_centres()
        s.set_field(L.VEL, np.stack([np.sin(2 * np.pi * xc) * np.cos(2 * np.pi * yc), -np.cos(2 * np.pi * xc) * np.sin(2 * np.pi * yc)], -1))
        s.set_math(args.math == "strict")
        t0 = time.perf_counter()
        s.install_poisson_matrix()
        t_op = time.perf_counter() - t0
        for _ in range(2):
            s.step(max_iter=args.iters)
        nst = 5
        L.check(s.L.cup2d_synchronize(s._ctx), "synchronize")
        t0 = time.perf_counter()
        for _ in range(nst):
            r = s.step(max_iter=args.iters)
            beat("amr step")
        L.check(s.L.cup2d_synchronize(s._ctx), "synchronize")
        el = (time.perf_counter() - t0) / nst
        solver, stats = s.last_solver(), s.matrix_stats()
        # the launches of an iteration on the hybrid operator (HIP events, sampled), in the same context
        kernels_us, iteration_us = None, None
        try:
            import ctypes
            L.check(s.L.cup2d_set_timing(s._ctx, 2), "set_timing")
            for _ in range(4):
                s.step(max_iter=args.iters)
            kernels_us = {}
            for name in ("sweep_A", "sweep_C", "sweep_E", "sweep_EA", "advect_stage", "poisson_rhs", "project"):
                ms, n = ctypes.c_double(), ctypes.c_int()
                L.check(s.L.cup2d_get_timing(s._ctx, L.TIMER_NAMES.index(name), ctypes.byref(ms), ctypes.byref(n)), "get_timing")
                if n.value:
                    kernels_us[name] = round(1e3 * ms.value / n.value, 1)
            L.check(s.L.cup2d_set_timing(s._ctx, 0), "set_timing")
            iteration_us = round(sum(kernels_us.get(k, 0.0) for k in ("sweep_A", "sweep_C", "sweep_E", "sweep_EA")), 1)
        except Exception as e:  # informative
            kernels_us = {"error": str(e)[:120]}
        try:
            amr_tol = tolerance_leg(s.step, lambda: L.check(s.L.cup2d_synchronize(s._ctx), "synchronize"), 3)
        except Exception as e:  # informative
            amr_tol = {"error": str(e)[:200]}
        # one regrid the way a run does it (tags from max|vorticity| per block: here the band moves with the thresholds)
        # regrids the way a run does them (tags from max|vorticity| per block: the band moves with the thresholds, a few
        # steps in between): the FIRST one of a process pays one-time costs (kernels loaded, pools empty: first_ms), a run
        # regrids every AdaptSteps = 20 steps -- "ms" is the median of the following three
        regrids = []
        for k in range(4):
            s.vorticity()
            om = np.abs(s.get_field(L.TMP)).reshape(s.grid.nblocks, -1).max(1)
            qr, qc = ((0.97, 0.5), (0.99, 0.03), (0.985, 0.05), (0.99, 0.03))[k]
            nb_before = s.grid.nblocks
            L.check(s.L.cup2d_synchronize(s._ctx), "synchronize")
            t0 = time.perf_counter()
            changed = s.adapt(float(np.quantile(om, qr)), float(np.quantile(om, qc)), args.amr_lfine + 1)
            L.check(s.L.cup2d_synchronize(s._ctx), "synchronize")
            regrids.append({"ms": round((time.perf_counter() - t0) * 1e3, 2), "changed": bool(changed), "blocks_before": nb_before,
                            "blocks_after": s.grid.nblocks,
                            "stages_ms": {n: round(v, 2) for n, v in getattr(s, "adapt_stages_ms", {}).items()}})
            s.step(max_iter=min(args.iters, 10))
            beat("amr regrid %d" % k)
        warm = sorted(regrids[1:], key=lambda r: r["ms"])[1]
        t_adapt, changed, nb_after, stages = warm["ms"] * 1e-3, warm["changed"], warm["blocks_after"], warm["stages_ms"]
    # what a rank of configs[4]'s 8-rank run exchanges on this grid (the plan alone: host code, no second GPU here)
    from cup2d_amd.amr_dist import AmrPartition
    P = AmrPartition(g, 8, 0)
    plan8 = {"rank": 0, "owned_blocks": P.nowned, "ghost_blocks": P.nghost, "sent_blocks": P.nsend, "peers": len(P.peers),
             "cells_sent_per_refresh": {n: int(c[2].nsend) for n, c in zip(L.CELL_SET_NAMES, P.cells)},
             "cells_of_the_sent_blocks": 64 * P.nsend,
             "what": "contiguous Hilbert ranges on 8 ranks (main.cpp:6494-6504); a refresh of ghost blocks sends the cells the receiver's "
                     "kernels read (cup2d_halo_plan_cells; found by tracing the kernels' own ghost expressions), not whole blocks: "
                     "matrix = the Krylov vector, twice per BiCGSTAB iteration.  Plan only -- one GPU per box"}
    return {"workload": "three-level block-AMR grid, finest level %d^2-equivalent in a band around a circle; same step, %d BiCGSTAB "
                        "iters on the assembled coarse-fine operator" % (8 << args.amr_lfine, args.iters),
            "blocks": g.nblocks, "cells": g.nblocks * 64, "blocks_per_level": np.bincount(g.blocks[:, 0]).tolist(),
            "value": round(g.nblocks * 64 / el / 1e6, 2), "unit": "Mcell-updates/s", "ms_per_step": round(el * 1e3, 3),
            "iters": r["iters"], "solver": solver, "iteration_us": iteration_us, "kernels_us": kernels_us, "solve_to_tolerance": amr_tol, "operator": stats, "operator_install_ms": round(t_op * 1e3, 1),
            "grid_build_ms": round(t_grid * 1e3, 1), "plan_on_8_ranks": plan8,
            "regrid": {"changed": bool(changed), "blocks_before": warm["blocks_before"], "blocks_after": nb_after,
                       "ms": round(t_adapt * 1e3, 1), "first_ms": regrids[0]["ms"], "all": regrids,
                       "stages_ms": stages,
                       "what": "adapt(): tags, 2:1 balance, plan + tables of the new leaves (host, leaf lists only), new context, "
                               "prolongation / restriction / copy of five fields by k_amr_regrid between the two contexts (no "
                               "field crosses PCIe), operator"}}


def _g(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def _r(x, n=1):
    return round(x, n) if isinstance(x, (int, float)) else x


def compact_line(full, detail_path):
    """The line the driver records: the contract's keys, `roofline` (dominant kernel), `cpu_baseline`, and `summary` -- the numbers of
    every leg without their prose (that is in the detail file)."""
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data")}
    cfg = dict(full["config"])
    comm = cfg.pop("comm", None)
    line["config"] = cfg
    if comm:
        cfg["comm"] = {k: comm.get(k) for k in ("cartesian", "rccl_ranks", "shared_gpu", "exchanges", "allgathers", "allreduces")}
        cfg["comm"]["transport"] = str(comm.get("transport"))[:40]
        cfg["comm"]["selftest_us"] = [_g(comm, "selftest", "exchange_us"), _g(comm, "selftest", "reduce_us")]
        if comm.get("organisations"):
            cfg["comm"]["organisations_ms_per_step"] = {k.split(" ")[0].split(":")[0]: v.get("ms_per_step", v.get("error"))
                                                        for k, v in comm["organisations"].items()}
    rf = full.get("roofline")
    line["roofline"] = {k: rf.get(k) for k in ("kernel", "family", "bound", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_cell",
                                               "avg_launch_ms", "launches", "share_of_gpu_time")} if rf else None
    cpu = full.get("cpu_baseline")
    if cpu and "error" not in cpu:
        line["cpu_baseline"] = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind", "solver", "sample", "host_hardware_threads")}
        fh = cpu.get("reference_functors_at_headline_size") or {}
        if "advect_diffuse" in fh:
            line["cpu_baseline"]["reference_functors_at_%d2_mcells_per_s" % fh.get("n", 0)] = {
                "advect_diffuse": fh["advect_diffuse"], "pressure_rhs1": fh["pressure_rhs1"], "cores": fh.get("cores"), "kind": "reference"}
    else:
        line["cpu_baseline"] = cpu
    S = {"verified": full.get("verified_summary"), "ms_per_step_no_kernel_timers": full.get("ms_per_step_no_kernel_timers")}
    ns = full.get("roofline_north_star")
    if ns:
        tr = ns.get("traffic")
        alg = ns["bytes_per_cell"] * _g(full, "config", "global_cells", default=0) / max(1, full["n_gpus"])
        S["north_star"] = {"kernel": ns["kernel"], "frac": ns["frac"], "avg_us": _r(ns["avg_launch_ms"] * 1e3, 1), "launches": ns["launches"],
                           "bytes_per_cell": ns["bytes_per_cell"], "traffic_over_algorithmic": _r(tr / alg, 3) if tr and alg else None,
                           "fp64_frac_of_32T": ns.get("fp64_frac_of_measured_ceiling_32T")}
        for st in ("stage1", "stage2"):
            if st in ns:
                S["north_star"][st] = {k: ns[st].get(k) for k in ("avg_launch_us", "launches", "hbm_frac", "floor_memory_us", "floor_arithmetic_us",
                                                                  "bound", "frac_of_larger_floor", "floor_leg_vs_timed_region")}
        if _g(ns, "floors", "error"):
            S["north_star"]["floors_error"] = ns["floors"]["error"]
    g = full.get("gpu_ms_per_step") or {}
    S["gpu_ms_per_step"] = {k: g.get(k) for k in ("solver_sweeps", "outside_the_sweeps")}
    S["kernels"] = {}
    for fam, r in (full.get("roofline_all") or {}).items():
        alg = r["bytes_per_cell"] * _g(full, "config", "global_cells", default=0) / max(1, full["n_gpus"])
        S["kernels"][fam] = {"k": r["kernel"], "us": _r(r["avg_launch_ms"] * 1e3, 1), "frac": r["frac"], "n": r["launches"],
                             "B_per_cell": r["bytes_per_cell"], "x_alg": _r(r["traffic"] / alg, 2) if r.get("traffic") and alg else None}
    for fam in ("project", "reduce", "final_x", "scalars"):
        t = _g(full, "kernels", fam)
        if t and t.get("launches"):
            S["kernels"][fam] = {"us": _r(t["ms_avg"] * 1e3, 1), "n": t["launches"]}
    sv = full.get("solver")
    if sv:
        S["iteration"] = {"us": _r(sv["ms_per_iteration"] * 1e3, 1), "B_per_cell": sv["bytes_per_cell_iteration"], "frac": sv["frac_hbm"]}
    tl = full.get("solve_to_tolerance")
    if tl:
        S["solve_to_tolerance"] = {k: tl.get(k) for k in ("iters_median", "ms_per_step_median", "ms_not_in_iterations_or_fringe", "error") if k in tl}
    ss = full.get("second_size")
    if ss:
        S["second_size_2048"] = {k: ss.get(k) for k in ("value", "ms_per_step", "error") if k in ss}
    am = full.get("amr_configs4")
    if am:
        S["amr_configs4"] = ({"error": am["error"]} if "error" in am else
                             {"value": am["value"], "ms_per_step": am["ms_per_step"], "blocks": am["blocks"], "solver": am.get("solver"),
                              "iteration_us": am.get("iteration_us"), "kernels_us": am.get("kernels_us"),
                              "regrid_ms": _g(am, "regrid", "ms"), "regrid_stages_ms": _g(am, "regrid", "stages_ms"),
                              "tolerance": {k: _g(am, "solve_to_tolerance", k) for k in ("iters_median", "ms_per_step_median")}})
    nr = full.get("nrank_path_on_one_gpu")
    if nr:
        if "error" in nr:
            S["nrank_path_on_one_gpu"] = {"error": nr["error"]}
        else:
            def patch(p):
                if "error" in p:
                    return {"error": p["error"][:80]}
                return {k: p.get(k) for k in ("ms_per_step", "plain_context_ms_per_step", "ratio_to_plain", "fixed_us_per_iteration_over_plain",
                                              "outside_the_sweeps_us_per_iteration", "value", "form")}
            S["nrank_path_on_one_gpu"] = {"%s_%s" % (p.get("blocks"), p.get("ghost_sides")): patch(p) for p in [nr] + list(nr.get("other_patches") or [])}
            for k in ("configs3_on_paper", "weak_on_paper"):
                if nr.get(k):
                    S["nrank_path_on_one_gpu"][k] = nr[k]
    pl = full.get("placement") or {}
    S["placement"] = {k: _r(pl.get(k), 1) for k in ("candidates", "kept_us", "slowest_us", "first_us") if k in pl} or pl
    sl = full.get("second_layout")
    if sl:
        S["second_layout"] = {k: sl.get(k) for k in ("layout", "value", "ms_per_step", "cells_per_rank", "scaling", "error") if k in sl}
    line["summary"] = S
    line["detail"] = detail_path
    line["verified_ok"] = full.get("verified_ok")
    return line


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)

    # stdout carries exactly ONE line, the JSON: libraries that talk on fd 1 (RCCL prints a version banner there from C)
    # are sent to stderr for the whole run, the line is written to the saved descriptor at the very end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    global WATCHDOG
    WATCHDOG = Watchdog(float(os.environ.get("CUP2D_BENCH_WATCHDOG_S", "300")))
    import torch
    from cup2d_amd import lib as L
    beat("import")

    # one node: rendezvous sockets (gloo, RCCL's bootstrap) on loopback -- resolving the container's hostname can take
    # minutes to fail; HSA_ENABLE_IPC_MODE_LEGACY=0: dmabuf IPC between the ranks' processes
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    if SHARE_GPU:
        local_rank = 0          # every rank on GPU 0
        if world > 1 or args.force_dist:
            args.comm = "torch"  # gloo, host-staged (see SHARE_GPU above)
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU %d (%d visible)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)

    dist, ctl = None, {}
    px = py = 1
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        from cup2d_amd.distributed import cartesian_dims
        if world == 1 and "MASTER_ADDR" not in os.environ:  # --force-dist outside torch.distributed.run
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        import datetime
        # control plane only: token, barriers, max of the wall time; a rank that dies must not leave the others waiting
        dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=4))
        ctl["gloo"] = dist.group.WORLD
        if args.comm == "torch" and not SHARE_GPU:
            ctl["nccl"] = dist.new_group(backend="nccl")
        px, py = cartesian_dims(world)

    def geometry(layout):
        if layout == "configs3":
            return args.configs3_n // px, args.configs3_n // py
        return args.n, args.n

    nx, ny = geometry(args.layout)
    run = Runner(args, nx, ny, px, py, rank, world, local_rank, dist, ctl)
    beat("setup")
    sim = run.sim
    fused = args.solver == "fused"
    # Setup ends with SETUP_STEPS whole steps (reported as config.setup_steps): the first steps of a process run 0.1-0.15 ms slower
    # than the steady state whatever is timed in them (tools/gpu_calls/gpu_r06_call49.sh: with --warmup 5 the timed region took
    # 17.725 / 17.848 ms per step and the same steps repeated at once 17.558 / 17.700; with --warmup 25: 17.679 / 17.666 and 17.654 /
    # 17.614 -- clocks and caches of a fresh process settling), and the metric is the sustained rate of steps back to back.  The W
    # warm-up steps the caller asks for follow, then exactly K timed steps; nothing is left out of either.
    for _ in range(SETUP_STEPS if dist is None else 0):
        run.one_step()
    for _ in range(args.warmup):
        run.one_step()

    # Timed region: EXACTLY --steps steps.  Per-kernel HIP-event pairs are recorded on the launch stream
    # inside it in SAMPLED mode (every 32nd BiCGSTAB iteration -- every 16th until the last commits of round 6: 0.15 ms of an 17.9 ms step --; the launches outside the solver in every 4th step: an event
    # pair is a barrier packet that keeps the kernels on either side of it from overlapping, ~12 us per pair -- round 5 measured
    # 18.17 against 17.90 ms per step with twice the samples); the roofline objects are computed from those samples and, where a
    # short --steps leaves fewer than MIN_ROOFLINE_LAUNCHES of them, from more steps sampled the same way outside the timed region.  The same K steps are repeated afterwards without any events and
    # reported as ms_per_step_no_kernel_timers.
    # N ranks: the timed region runs WITHOUT per-kernel events (resolving the event pool drains the stream, and with the
    # scalar/communication launches of the N-rank organisation that costs the pipelined solve far more than the 1 % it
    # costs the one-GPU path); the per-kernel samples are then taken in extra steps outside the timed region
    timed_with_events = not args.no_kernel_timers and dist is None
    elapsed, iters = run.timed_steps(args.steps, 2 if timed_with_events else 0)
    beat("timed region")
    if not args.no_kernel_timers and not timed_with_events:
        sim.set_timing(2)
    cells_rank = nx * ny
    cells = cells_rank * world
    value = cells * args.steps / elapsed / 1e6
    extra_sampled_steps = 0
    acc = {name: list(sim.get_timing(i)) for i, name in enumerate(L.TIMER_NAMES)}   # the samples of the timed region
    if not args.no_kernel_timers:
        # too few sampled launches for a per-kernel average (every family, the north-star kernel included: one launch of each
        # RK stage per step): keep sampling OUTSIDE the timed region, with the launches outside the solver sampled in every step
        def short():
            return (max(acc["sweep_E"][1], acc["sweep_EA"][1]) < MIN_ROOFLINE_LAUNCHES
                    or acc["advect_stage"][1] + acc["advect_stage2"][1] < MIN_ROOFLINE_LAUNCHES)
        if short():
            # (back to back, the timers read once at the end: a call between two steps makes the second one recompute max|u|
            # with a pass over the velocity in front of RK stage 1 -- not what the timed region runs)
            need_adv = max(0, MIN_ROOFLINE_LAUNCHES - acc["advect_stage"][1] - acc["advect_stage2"][1] + 1) // 2
            need_sw = max(0, MIN_ROOFLINE_LAUNCHES - max(acc["sweep_E"][1], acc["sweep_EA"][1])) // max(1, (args.iters + 31) // 32) + 1
            extra_sampled_steps = min(64, max(need_adv, need_sw, 1))
            sim.set_timing(3)
            for _ in range(extra_sampled_steps):
                run.one_step()
            for i, name in enumerate(L.TIMER_NAMES):
                ms, calls = sim.get_timing(i)
                acc[name] = [acc[name][0] + ms, acc[name][1] + calls]
    timers = {}
    for name in L.TIMER_NAMES:
        ms, calls = acc[name]
        timers[name] = {"ms_total": round(ms, 4), "launches": calls, "ms_avg": round(ms / calls, 5) if calls else None}
    # the two RK stages of the north-star kernel are timed apart (stage 1: 32 B/cell, stage 2: 48); "advect_stage" below is
    # the family, one launch of each per step
    stage_t = {"stage1": dict(timers["advect_stage"]), "stage2": dict(timers["advect_stage2"])}
    if stage_t["stage2"]["launches"]:
        ms = timers["advect_stage"]["ms_total"] + timers["advect_stage2"]["ms_total"]
        calls = timers["advect_stage"]["launches"] + timers["advect_stage2"]["launches"]
        timers["advect_stage"] = {"ms_total": round(ms, 4), "launches": calls, "ms_avg": round(ms / calls, 5)}
    del timers["advect_stage2"]
    sim.set_timing(False)
    elapsed_plain = None
    if timed_with_events:
        elapsed_plain, _ = run.timed_steps(args.steps, False)
    elif not args.no_kernel_timers:
        elapsed_plain = elapsed

    # ---- outside the timed region: the Poisson smoother sweep ---------------------------------------------
    # BASELINE.json configs[1] words the pressure part "50 Jacobi pressure iters/step"; the reference has no smoother
    # (SURVEY.md F4), the step above runs its real solver.  The weighted-Jacobi sweep (csrc/smoother.hip, 24 B/cell)
    # is timed here on the Poisson system of the last step, every launch bracketed by HIP events.
    if not args.no_kernel_timers and world == 1:
        try:
            sim.jacobi_sweeps(6, omega=0.8)
            sim.set_timing(1)
            sim.jacobi_sweeps(100, omega=0.8)
            ms, calls = sim.get_timing(L.TIMER_NAMES.index("smoother"))
            timers["smoother"] = {"ms_total": round(ms, 4), "launches": calls, "ms_avg": round(ms / calls, 5) if calls else None}
            sim.set_timing(False)
        except Exception as e:  # an extra, never the reason the bench line is missing
            timers["smoother"] = {"ms_total": 0.0, "launches": 0, "ms_avg": None, "error": str(e)[:200]}

    verified = None
    if not args.no_verify:
        try:
            verified = run.verify()
        except Exception as e:
            verified = {"ok": False, "error": str(e)[:300]}

    comm_info = None
    if dist is not None:
        comm_info = {"transport": {"rccl": "in-library RCCL communicator (csrc/comm.hip)",
                                   "torch": "torch.distributed gloo, host-staged, behind cup2d_set_comm (ranks share GPU 0: test only)"
                                   if SHARE_GPU else "torch.distributed nccl behind cup2d_set_comm",
                                   "none": "none"}[run.comm_kind], "peers_of_rank0": len(sim.topo.peers), "shared_gpu": SHARE_GPU,
                     "cartesian": "%dx%d" % (px, py)}
        if run.comm_kind == "rccl":
            st = sim.comm_stats()
            rep = getattr(sim, "comm_report", {}) or {}
            comm_info.update({"rccl_ranks": st["nranks"], "exchanges": st["exchanges"], "allreduces": st["allreduces"],
                              "allgathers": st["allgathers"], "librccl": rep.get("rccl"), "peer_ranks_of_rank0": rep.get("peers"),
                              "selftest": {"ok": True, "exchange_us": rep.get("exchange_us"), "reduce_us": rep.get("reduce_us"),
                                           "what": "cup2d_comm_selftest after cup2d_comm_init: the plan's strips between all "
                                                   "peers + all-gather + all-reduce, checked values, 20 s deadline"}})
            if st["nranks"] != args.gpus:
                raise SystemExit("bench.py: RCCL communicator has %d ranks, --gpus %d" % (st["nranks"], args.gpus))

    # ---- N > 1: the three organisations of a reduction point, timed back to back on this layout (diagnostic: one run on N GPUs
    # says which one the links favour; the headline above ran the default) ------------------------------------------------
    organisations = None
    if world > 1:
        organisations = {}
        for name, (dfr, spl) in (("deferred_unsplit (default with the in-library communicator)", (1, 0)),
                                 ("allgather_and_scalar_kernel_per_reduction_point (round 4)", (0, 0)),
                                 ("split_sweeps_halo_set_first (computeA's split; round-off differs)", (0, 1)),
                                 ("overlap: split sweeps + deferred update, block transfer behind the inner launch", (1, 1))):
            try:
                sim.set_nrank_organisation(dfr, spl)
                run.one_step()
                el_o, _ = run.timed_steps(max(2, min(args.steps, 5)), False)
                form = sim.last_solver_form()
                organisations[name] = {"ms_per_step": round(el_o / max(2, min(args.steps, 5)) * 1e3, 3), "solver_form": list(form)}
            except Exception as e:  # informative
                organisations[name] = {"error": str(e)[:200]}
            beat("organisation")
        sim.set_nrank_organisation(-1, -1)
        if comm_info is not None:
            comm_info["organisations"] = organisations

    # ---- the other layout (N > 1): same command, second timed region -------------------------------------------
    second = None
    beat("first layout")
    if world > 1 and not args.no_second_layout:
        other = "configs3" if args.layout == "weak" else "weak"
        run.close()
        try:
            nx2, ny2 = geometry(other)
            run2 = Runner(args, nx2, ny2, px, py, rank, world, local_rank, dist, ctl)
            for _ in range(args.warmup):
                run2.one_step()
            el2, _ = run2.timed_steps(args.steps, False)
            second = {"layout": other, "value": round(nx2 * ny2 * world * args.steps / el2 / 1e6, 3), "unit": "Mcell-updates/s",
                      "ms_per_step": round(el2 / args.steps * 1e3, 3), "cells_per_rank": "%dx%d" % (nx2, ny2),
                      "global_cells": nx2 * ny2 * world, "scaling": "strong" if other == "configs3" else "weak",
                      "steps": args.steps, "warmup": args.warmup}
            run2.close()
        except Exception as e:
            second = {"layout": other, "error": str(e)[:300]}

    # ---- rooflines --------------------------------------------------------------------------------
    # Algorithmic (compulsory) bytes per cell and launch of every kernel family, FP64, halo re-reads
    # excluded (DESIGN.md section 4 derives each line):
    #   advect_stage  stage 1 reads vel 16 + writes mid 16 = 32; stage 2 reads mid 16 + vel 16, writes 16 = 48; mean 40
    #   poisson_rhs   reads vel 16 + pres 8, writes tmp 8 + pold 8 (pold = pres rides on this kernel in cup2d_step) = 40
    #   sweep_A       reads p, nu, r 24 + writes p, z 16 = 40      sweep_C  reads r, nu 16 + writes r, z2 16 = 32
    #   sweep_B / D   reads z (z2) 8 + rhat (r) 8, writes nu (t) 8 = 24
    #   sweep_E       reads x, z, z2, r, t, rhat 48 + writes x, r 16 = 64
    #   smoother      weighted-Jacobi sweep (not part of the step): reads x, b 16 + writes x' 8 = 24
    #   fused solver (krylov_fused.hip): sweep_A = A+B in one launch: reads p, nu, r, rhat 32 + writes p', nu' 16 = 48
    #   sweep_C = C+D: reads r, nu 16 + writes t 8 = 24 (s = r - alpha nu is not stored)
    #   sweep_E: reads y, p, r, nu, t, rhat 48 + writes y, r 16 = 64 (forms s again)
    ALGO_BYTES = {"advect_stage": 40.0, "poisson_rhs": 40.0, "sweep_A": 40.0, "sweep_B": 24.0, "sweep_C": 32.0,
                  "sweep_D": 24.0, "sweep_E": 64.0, "init_residual": 24.0, "smoother": 24.0}
    # init_residual: cup2d_step tells the solve that x0 = 0 (k_init_residual<true>: r = b, no stencil pass): reads b 8,
    # writes r, rhat 16 = 24
    mk = "true" if args.finish == "kernel" and dist is None else "false"
    # FAST policy: the quad kernel (csrc/advect_walk.h; stage 1 is <1, true>, stage 2 <1, false>) unless CUP2D_ADVECT_WALK=0
    walk = args.math == "fast" and os.environ.get("CUP2D_ADVECT_WALK", "1") != "0"
    KERNEL_OF = {"advect_stage": "k_advect_walk<1, true|false>" if walk else
                 ("k_advect_diffuse<WenoFast, 1>" if args.math == "fast" else "k_advect_diffuse<WenoStrict, 1>"),
                 "poisson_rhs": "k_pressure_rhs<false, true>", "sweep_A": "k_sweepA_fd", "sweep_B": "k_sweepBD<1, %s>" % mk,
                 "sweep_C": "k_sweepC_fd", "sweep_D": "k_sweepBD<2, %s>" % mk, "sweep_E": "k_sweepE<%s>" % mk,
                 "init_residual": "k_init_residual<true>", "smoother": "k_smoother<0, false, 1>"}
    sweeps = ("sweep_A", "sweep_B", "sweep_C", "sweep_D", "sweep_E")
    if fused:
        ALGO_BYTES.update({"sweep_A": 48.0, "sweep_C": 24.0, "sweep_E": 64.0})
        del ALGO_BYTES["sweep_B"], ALGO_BYTES["sweep_D"]
        # fused kernels: MERGE 1 = the last workgroup finishes the reduction and updates the scalars (one GPU),
        # 2 = it sums the rank's partials, all-reduce + scalar kernel follow (N GPUs), 0 = finish launch
        mf = 0 if args.finish != "kernel" else (1 if dist is None else 2)
        KERNEL_OF.update({"sweep_A": "k_fused<0, %d>" % mf, "sweep_C": "k_fused<1, %d>" % mf, "sweep_E": "k_sweepE_y<%d>" % mf})
        sweeps = ("sweep_A", "sweep_C", "sweep_E")
    # The library's default on one GPU (krylov_edge.h MODE 2 / 3, CUP2D_FUSED_FORM unset or eab): per iteration TWO launches --
    #   sweep_C  = C+D with the sums of the next beginning: reads r, nu', rhat 24 + writes t 8 = 32
    #   sweep_EA = sweep E + the next iteration's A+B: reads p', nu', r, t, y, rhat 48 + writes y', r', p'', nu'' 32 = 80
    #   sweep_A  = the A+B of iteration 0, once per solve (p = nu = 0 are not read): reads r, rhat 16 + writes p', nu' 16 = 32
    eab = fused and timers["sweep_EA"]["launches"] > 0
    if eab:
        ALGO_BYTES.update({"sweep_A": 32.0, "sweep_C": 32.0, "sweep_EA": 80.0})
        del ALGO_BYTES["sweep_E"]
        KERNEL_OF.update({"sweep_A": "k_edge<0, %d>" % mf, "sweep_C": "k_edge<3, %d>" % mf, "sweep_EA": "k_edge<2, %d>" % mf})
        sweeps = ("sweep_C", "sweep_EA")
    finish_launches = 0 if mk == "true" else 3
    # HBM bytes per launch measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH x2 on
    # gfx950), summarised by tools/prof_summary.py from the same bench command: profiles/<tag>_pmc_traffic.json
    traffic_tab, traffic_src = {}, {}
    prof_dir = os.path.join(ROOT, "profiles")
    prof_tag = args.profile_tag
    if os.path.isdir(prof_dir):
        # ONE committed summary supplies the traffic, chosen by tag: --profile-tag, else the newest round's final set (rNN
        # without a suffix; "r03a" is a first-half set and never the default)
        import re
        tags = sorted(m.group(1) for m in (re.match(r"^(r\d\d)_pmc_traffic\.json$", f) for f in os.listdir(prof_dir)) if m)
        if not prof_tag and tags:
            prof_tag = tags[-1]
        for f in ([prof_tag + "_pmc_traffic.json"] if prof_tag else []):
            if not os.path.exists(os.path.join(prof_dir, f)):
                continue
            try:
                for k, v in json.load(open(os.path.join(prof_dir, f)))["kernels"].items():
                    if k.startswith("k_fused<") and k.count(",") > 1:  # <MODE, MERGE, hybrid operator, ghost edges>: the
                        if "true" in k:                                  # uniform bench runs <MODE, MERGE, false, false>
                            continue
                        k = ",".join(k.split(",")[:2]) + ">"
                    if k.startswith("k_edge<") and k.count(",") > 1:   # <MODE, MERGE, hybrid operator> (round 6): the uniform
                        if "true" in k:                                # bench runs <MODE, MERGE, false>
                            continue
                        k = ",".join(k.split(",")[:2]) + ">"
                    traffic_tab[k] = v
                    traffic_src[k] = "profiles/" + f
            except Exception:
                pass

    def roofline_of(fam):
        t = timers[fam]
        if not t["launches"] or fam not in ALGO_BYTES:
            return None
        sec = t["ms_total"] / t["launches"] * 1e-3
        gbs = ALGO_BYTES[fam] * cells_rank / sec / 1e9
        tr = traffic_tab.get(KERNEL_OF[fam], {}).get("hbm_bytes") if (nx, ny) == (4096, 4096) and world == 1 else None
        if fam == "advect_stage" and walk and (nx, ny) == (4096, 4096) and world == 1:
            # the two RK stages are two instantiations; per launch of the family = their mean (one launch each per step)
            # (round 6: the instantiations carry the knock-out parameter: <1, true, 0> is the product)
            both = [(traffic_tab.get("k_advect_walk<1, %s, 0>" % b) or traffic_tab.get("k_advect_walk<1, %s>" % b, {})).get("hbm_bytes")
                    for b in ("true", "false")]
            if all(both):
                tr = 0.5 * (both[0] + both[1])
                traffic_src[KERNEL_OF[fam]] = traffic_src.get("k_advect_walk<1, true, 0>") or traffic_src.get("k_advect_walk<1, true>")
        return {"kernel": KERNEL_OF[fam], "family": fam, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_copy_ceiling": round(gbs / HBM_COPY_CEILING_GBS, 4),
                "traffic": tr,
                "traffic_source": traffic_src.get(KERNEL_OF[fam]) if tr else None,
                "bytes_per_cell": ALGO_BYTES[fam], "avg_launch_ms": round(sec * 1e3, 4), "launches": t["launches"],
                "share_of_gpu_time": None}

    # GPU time per step of a family = its (sampled) average launch x launches per step
    per_step = {"advect_stage": 2, "poisson_rhs": 1, "init_residual": 1, "project": 1, "reduce": 3, "final_x": 1,
                "sweep_A": args.iters, "sweep_B": args.iters, "sweep_C": args.iters, "sweep_D": args.iters,
                "sweep_E": args.iters, "scalars": finish_launches * args.iters + 1, "halo": 0}
    if fused:
        per_step["sweep_B"] = per_step["sweep_D"] = 0
    if eab:
        per_step.update({"sweep_A": 1, "sweep_E": 0, "sweep_EA": args.iters})
    step_ms = {f: (timers[f]["ms_avg"] or 0.0) * per_step.get(f, 0) for f in timers}
    gpu_ms = sum(step_ms.values()) or 1.0
    all_roof = {}
    for fam in ALGO_BYTES:
        r = roofline_of(fam)
        if r:
            r["share_of_gpu_time"] = round(step_ms[fam] / gpu_ms, 4)
            all_roof[fam] = r
    # "roofline": the dominant kernel of the timed region (largest share of GPU time)
    dominant = max(all_roof, key=lambda f: step_ms[f]) if all_roof else None
    roofline = all_roof.get(dominant)
    # the north-star kernel (fused WENO5 advect-diffuse RK stage) is FP64-issue bound, not HBM bound: next to the
    # HBM fraction report the FP64 instruction rate against the measured VALU ceiling (tools/fp64_peak.hip:
    # 29-32 T lane-instr/s sustained on this chip -- fma 29.4, mul 31.2, add 32.1; 39.3 T at the nominal 2.4 GHz) --
    # DESIGN.md section 4.1
    north = dict(all_roof["advect_stage"]) if "advect_stage" in all_roof else None
    if north:
        sec = north["avg_launch_ms"] * 1e-3
        # FP64 VALU instructions executed per cell in a block whose velocity components do not change sign
        # (counted in the gfx950 ISA of advect.hip; SQ_INSTS_VALU measures the VALU instructions of all kinds)
        # quad kernel: 645 FP64 instructions per lane and quad of 256 cells (two walks of 318 + the sign tests) = 161 per cell
        fp64_per_cell = (161.0 if walk else 241.0) if args.math == "fast" else 565.0
        rate = fp64_per_cell * cells_rank / sec / 1e12
        north.update({"mcells_per_s": round(cells_rank / sec / 1e6, 1), "fp64_instr_per_cell": fp64_per_cell,
                      "fp64_T_lane_instr_per_s": round(rate, 2), "fp64_frac_of_measured_ceiling_32T": round(rate / 32.0, 4),
                      "fp64_frac_of_nominal_39.3T": round(rate / 39.3, 4)})
        # per stage: both roofs, and the two floors of the kernel MEASURED IN THIS RUN (north_star_floors below: the same context,
        # the same data, whole steps with the same sampled timers -- knock-out instantiations of the kernel, advect.hip KO)
        for st, bpc in (("stage1", 32.0), ("stage2", 48.0)):
            t = stage_t[st]
            if t["launches"] and (nx, ny) == (4096, 4096):
                ssec = t["ms_total"] / t["launches"] * 1e-3
                gb = bpc * cells_rank / ssec / 1e9
                fr = fp64_per_cell * cells_rank / ssec / 1e12
                north[st] = {"avg_launch_us": round(ssec * 1e6, 2), "launches": t["launches"], "bytes_per_cell": bpc,
                             "hbm_frac": round(gb / HBM_PEAK_GBS, 4), "hbm_frac_of_copy_ceiling": round(gb / HBM_COPY_CEILING_GBS, 4),
                             "fp64_frac_of_measured_ceiling_32T": round(fr / 32.0, 4)}
        if not args.no_north_star_floors and world == 1 and dist is None and walk:
            try:
                fl = north_star_floors(sim, args)
                north["floors"] = fl
                for st in ("stage1", "stage2"):
                    if st in north and fl.get("product", {}).get(st):
                        mem, ari, prod = fl["memory_skeleton"][st], fl["arithmetic_alone"][st], fl["product"][st]
                        north[st].update({"floor_memory_us": mem, "floor_arithmetic_us": ari, "product_in_floor_leg_us": prod,
                                          "bound": "FP64 issue" if ari >= mem else "HBM",
                                          "frac_of_larger_floor": round(max(mem, ari) / prod, 4),
                                          "floor_leg_vs_timed_region": round(prod / north[st]["avg_launch_us"], 4)})
            except Exception as e:  # informative
                north["floors"] = {"error": str(e)[:200]}
            beat("north-star floors")
    # one BiCGSTAB iteration = sweeps A..E + 3 scalar kernels (sum of the sampled average durations)
    it_bytes = sum(ALGO_BYTES[k] for k in sweeps)
    solver = None
    if iters and all(timers[k]["launches"] for k in sweeps + ("scalars",)):
        t_it = (sum(timers[k]["ms_avg"] for k in sweeps) + finish_launches * timers["scalars"]["ms_avg"]) * 1e-3
        solver = {"iterations": iters, "ms_per_iteration": round(t_it * 1e3, 4), "bytes_per_cell_iteration": it_bytes,
                  "achieved_GBs": round(it_bytes * cells_rank / t_it / 1e9, 1),
                  "frac_hbm": round(it_bytes * cells_rank / t_it / 1e9 / HBM_PEAK_GBS, 4),
                  "mcell_iterations_per_s": round(cells_rank / t_it / 1e6, 1)}

    # GPU time of a step by part (sampled averages x launches per step): the BiCGSTAB sweeps and everything else
    sweeps_ms = sum(step_ms.get(f, 0.0) for f in ("sweep_A", "sweep_B", "sweep_C", "sweep_D", "sweep_E", "sweep_EA"))
    gpu_split = {"solver_sweeps": round(sweeps_ms, 4), "outside_the_sweeps": round(gpu_ms - sweeps_ms, 4),
                 "families": {f: round(v, 4) for f, v in step_ms.items() if v},
                 "note": "final_x = the solve's last pass x = P_inv y (under a timer since round 4)"}

    # the same workload with the solve ended by the reference's tolerances (what a run does after step 10)
    tol_leg = None
    if world == 1 and dist is None and not args.no_tolerance_leg:
        try:
            sim.set_timing(False)
            tol_leg = tolerance_leg(sim.step, sim.synchronize, 5)
            if solver:
                n_it, it_ms = tol_leg["iters_median"], solver["ms_per_iteration"]
                outside = gpu_split["outside_the_sweeps"]
                tol_leg.update({"ms_per_iteration_from_the_capped_run": it_ms, "gpu_ms_outside_the_sweeps_from_the_capped_run": outside,
                                "ms_not_in_iterations_or_fringe": round(tol_leg["ms_per_step_median"] - n_it * it_ms - outside, 3),
                                "note": "ms_not_in_iterations_or_fringe = median step - iterations x ms per iteration - the fringe of the "
                                        "capped step: what the solve's end costs (launches the host had enqueued behind convergence -- it "
                                        "looks once per group of 4 iterations and stays 2 groups ahead --, its final wait)"})
        except Exception as e:  # informative
            tol_leg = {"error": str(e)[:200]}
        beat("tolerance leg")

    # where the solver's vectors lie (csrc/krylov_fused.hip tune_placement): the search the first solve of the context ran
    placement = None
    try:
        pl = sim.placement()
        placement = dict(pl, what="the two launches of an iteration run in one of two modes that follow where the eleven vectors they "
                                  "stream lie in device memory; the first solve of a context times several complete sets (us per "
                                  "iteration on zero-filled vectors, reduction finish excluded) and keeps the fastest")
    except Exception as e:
        placement = {"error": str(e)[:200]}
    cpu = None
    beat("gpu part")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from oracle import oracle as O
            if O.have_reference():
                # thread sweep of the reference's own loop (16 / 64 / 128 of the box's hardware threads), the best is the
                # reported baseline; each point is the bounded 1024^2 sample (start-up + 3 steps, a few seconds)
                cores = os.cpu_count() or 1
                cand = [args.cpu_threads] if args.cpu_threads else sorted({min(t, cores) for t in (16, 64, 128)})
                sweep = {}
                for thr in cand:
                    try:
                        beat("cpu baseline, %d threads" % thr)
                        r = O.ref_step_time(args.cpu_n, steps=2, max_iter=args.iters, threads=thr, timeout=120)
                        sweep[thr] = (round(args.cpu_n ** 2 / r["median_step_s"] / 1e6, 4), r)
                    except Exception as e:
                        sweep[thr] = (0.0, {"error": str(e)[:100]})
                best = max(sweep, key=lambda t: sweep[t][0])
                r = sweep[best][1]
                cpu = {"value": sweep[best][0], "unit": "Mcell-updates/s", "cores": r.get("threads", best),
                       "kind": "reference", "solver": "CPU restatement of cuda.cu (the reference has no CPU solver)",
                       "thread_scaling_note": "the reference loop gets SLOWER beyond ~16 threads: its functors scale (OpenMP over blocks), "
                                              "but the Poisson solve has no CPU path in the reference (cuda.cu is its only solver) and is "
                                              "timed here as the serial CPU port of cuda.cu; with every hardware thread spinning in "
                                              "OpenMP barriers around that serial part (and the per-thread BlockLab allocations of "
                                              "computeA) the step time grows -- the best point of the sweep is the reported baseline",
                       "host_hardware_threads": cores, "thread_sweep_mcell_updates_per_s": {str(t): sweep[t][0] for t in sweep},
                       "sample": "reference main.cpp time loop (OpenMP functors; Poisson = CPU port of cuda.cu, %d iters) "
                                 "at %d^2, median of %d steps, best of the thread sweep" % (args.iters, args.cpu_n, r.get("timed_steps", 0))}
                # the reference's own functors alone (computeA over all blocks, OpenMP), median of reps
                fb = O.ref_bench(args.cpu_n, reps=5, threads=best)
                cpu["functors_mcells_per_s"] = {"advect_diffuse": round(fb["advect_diffuse_mcells"], 2),
                                                "pressure_rhs1": round(fb["pressure_rhs1_mcells"], 2),
                                                "sample": "computeA<..>(KernelAdvectDiffuse / pressure_rhs1) at %d^2, median of 5, %d threads"
                                                          % (args.cpu_n, best)}
                # ... and AT THE HEADLINE SIZE (kind "reference": nothing restated in it): computeA<VectorLab>(KernelAdvectDiffuse)
                # and computeA<ScalarLab>(pressure_rhs1) of main.cpp:3024-3061 over all 262 144 blocks; two thread counts, the
                # better one reported (the harness's start-up at this size is the bounded part of the sample: nomatrix = the
                # Poisson triplets of main.cpp:7034-7112, which no functor reads, are not assembled)
                nf = args.cpu_functor_n
                if nf and nf != args.cpu_n:
                    fsweep = {}
                    for thr in ([args.cpu_threads] if args.cpu_threads else sorted({best, min(64, cores)})):
                        try:
                            beat("cpu functors at %d^2, %d threads" % (nf, thr))
                            fsweep[thr] = O.ref_bench(nf, reps=3, threads=thr, nomatrix=True, timeout=240)
                        except Exception as e:
                            fsweep[thr] = {"error": str(e)[:100]}
                    okf = {t: v for t, v in fsweep.items() if "advect_diffuse_mcells" in v}
                    if okf:
                        tb = max(okf, key=lambda t: okf[t]["advect_diffuse_mcells"])
                        cpu["reference_functors_at_headline_size"] = {
                            "kind": "reference", "n": nf, "cores": tb, "unit": "Mcell/s",
                            "advect_diffuse": round(okf[tb]["advect_diffuse_mcells"], 2),
                            "pressure_rhs1": round(okf[tb]["pressure_rhs1_mcells"], 2),
                            "thread_sweep_advect_diffuse": {str(t): round(v["advect_diffuse_mcells"], 2) for t, v in okf.items()},
                            "sample": "computeA<VectorLab>(KernelAdvectDiffuse) / computeA<ScalarLab>(pressure_rhs1) over all blocks at "
                                      "%d^2, median of 3 passes after 2 warm-ups, best of the thread counts tried" % nf}
                    else:
                        cpu["reference_functors_at_headline_size"] = {"error": str(fsweep)[:200]}
            else:
                t1 = time.perf_counter()
                v0 = O.taylor_green(512)
                O.step(v0, np.zeros((512, 512)), 1.0 / 512, 1e-3, 0.5, tol=0.0, max_restarts=100, max_iter=args.iters)
                dt_cpu = time.perf_counter() - t1
                cpu = {"value": round(512 * 512 / dt_cpu / 1e6, 4), "unit": "Mcell-updates/s", "cores": os.cpu_count(),
                       "kind": "port", "sample": "oracle/cup2d_oracle.c oracle_step at 512^2, 1 step"}
        except Exception as e:  # the baseline is informative; never fail the bench on it
            cpu = {"error": str(e)[:200]}

    # BASELINE.json configs[4] on this GPU (not the headline; one GPU only -- the N-rank AMR path is covered by the tests):
    # a three-level block-AMR grid with the finest level 4096^2-equivalent in a band around a circle, the same step
    amr = None
    if rank == 0 and world == 1 and not args.no_amr:
        beat("amr leg")
        try:
            amr = amr_leg(args, local_rank)
        except Exception as e:  # informative; never fail the bench on it
            amr = {"error": str(e)[:200]}
        beat("amr leg done")

    # The N-rank code path of the same step on THIS one GPU (N = 1 only): a patch that is its own W and E neighbour through the
    # in-library communicator -- ghost blocks, pack kernels, whole ghost blocks of the Krylov vectors through ncclSend /
    # ncclRecv, the MERGE 2 kernels, an all-gather + one-wave kernel per reduction, one host look per iteration.  What a rank
    # of an N-rank run does except waiting for another GPU: its cost next to the plain context above is what the library
    # adds per rank (the driver's N > 1 runs add the links).
    nrank_proxy = None
    if rank == 0 and world == 1 and dist is None and not args.no_nrank_proxy:
        beat("N-rank path on one GPU")
        try:
            nrank_proxy = nrank_proxy_leg(args, local_rank, nx // 8, ny // 8, elapsed_plain or elapsed, args.steps)
        except Exception as e:  # informative; never fail the bench on it
            nrank_proxy = {"error": str(e)[:200]}
        beat("N-rank path done")

    # the one field that says the timed work was real, within the first bytes of the line (the full block is "verified")
    verified_summary = None
    if verified is not None:
        li = verified.get("last_iterate") or {}
        e8 = li.get("eight_iterations") or {}
        verified_summary = {"ok": verified.get("ok"), "reported_vs_recomputed_residual": [verified.get("residual_reported"),
                                                                                            verified.get("residual_recomputed")],
                            "last_iterate_relative_gap": li.get("relative_gap"), "eight_iterations_vs_five_sweeps": e8.get("relative"),
                            "error": verified.get("error")}
    # BASELINE.json configs[1] (2048^2 uniform, 50 iterations per step) on the same box, same step: a second, driver-visible size
    second_size = None
    if rank == 0 and world == 1 and dist is None and args.n != 2048 and not args.no_second_size:
        beat("second size")
        try:
            import cup2d_amd
            with cup2d_amd.Simulation(256, 256, nu=1e-3, cfl=0.5, device=local_rank) as s2:
                s2.set_math(args.math == "strict")
                s2.set_solver(fused=args.solver == "fused", finish_in_kernel=args.finish == "kernel")
                s2.vel = synthetic_velocity(2048, 2048, 0, 0, 2048, 2048, seed=20250117)
                el2, r2 = _proxy_time(args, s2, max(5, min(args.steps, 20)))
            second_size = {"workload": "2048x2048 uniform cells (BASELINE.json configs[1]), the same step", "value": round(2048 * 2048 / el2 / 1e6, 2),
                           "unit": "Mcell-updates/s", "ms_per_step": round(el2 * 1e3, 3), "iters": r2["iters"]}
        except Exception as e:  # informative
            second_size = {"error": str(e)[:200]}

    if rank == 0:
        out = {
            "metric": "Mcell-updates/sec (advect-diffuse+Poisson sweep) at 4096^2",
            "value": round(value, 3), "unit": "Mcell-updates/s", "n_gpus": world, "steps": args.steps,
            "verified_summary": verified_summary,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "ms_per_step_no_kernel_timers": round(elapsed_plain / args.steps * 1e3, 3) if elapsed_plain else None,
            "higher_is_better": True,
            "scaling": "strong" if args.layout == "configs3" else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%dx%d uniform cells per GPU; step = dt + RK2 WENO5 advect-diffuse + Poisson rhs + %d BiCGSTAB its + projection"
                                   % (nx, ny, args.iters),
                       "layout": args.layout, "global_cells": cells, "global_grid": "%dx%d" % (nx * px, ny * py),
                       "parallelism": run.par, "math": args.math, "bicgstab_iters_per_step": args.iters,
                       "solver": "fused" if fused else "sweeps", "finish": "kernel" if mk == "true" else "launch",
                       "setup_steps": SETUP_STEPS if dist is None else 0, "comm": comm_info},
            "verified": verified, "second_layout": second,
            "roofline": roofline, "roofline_north_star": north, "roofline_all": all_roof, "solver": solver,
            "gpu_ms_per_step": gpu_split, "solve_to_tolerance": tol_leg,
            "kernels": timers, "roofline_extra_sampled_steps_outside_timed_region": extra_sampled_steps,
            "cpu_baseline": cpu, "amr_configs4": amr, "nrank_path_on_one_gpu": nrank_proxy, "placement": placement, "second_size": second_size,
            # (the last key of the line as well: a record that keeps only the tail of stdout still says whether the timed work was real)
            "verified_ok": None if verified is None else bool(verified.get("ok")),
        }
    if second is None or world == 1:
        run.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # everything measured goes to a side file; stdout carries ONE compact line (< 8 KB: a record that keeps the tail of
        # stdout keeps all of it) with the contract's keys and a summary of every leg
        detail_path = os.environ.get("CUP2D_BENCH_DETAIL") or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
        try:
            os.makedirs(os.path.dirname(detail_path), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(out, f, indent=1)
        except Exception as e:
            detail_path = "not written: %s" % str(e)[:80]
        line = compact_line(out, os.path.relpath(detail_path, ROOT) if os.path.isabs(detail_path) else detail_path)
        os.write(json_fd, (json.dumps(line, separators=(",", ":")) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
