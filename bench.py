#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: Mcell-updates/s of CUP2D's stencil hot path.

One STEP = one pass of the reference's time-loop body (main.cpp:6576-7187, body-free) over the whole
grid: dt (max|u| reduction) -> RK2 WENO5 advect-diffuse (2 fused stages) -> Poisson right-hand side ->
block-Jacobi BiCGSTAB capped at --iters iterations (zero tolerances, like the reference's first ten
steps, main.cpp:7028-7030; BASELINE.json configs[1] words this "50 pressure iters/step") -> mean removal
+ pressure-gradient projection.  value = cells * steps / seconds / 1e6, whole job, inputs resident in HBM.

N = 1 : 4096^2 uniform grid (BASELINE.json configs[2], the headline config).
N > 1 : weak scaling, each rank owns a 4096^2-cell patch of a px x py Cartesian decomposition
        (configs[3] is the 2x4 case), face halos packed by HIP kernels and exchanged with RCCL
        send/recv (torch.distributed "nccl"), reductions by all-reduce.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (the dominant kernel of the timed region by GPU
time, timed live with HIP events on its launch stream), "roofline_north_star" (the fused WENO5
advect-diffuse stage, the kernel BASELINE.json's target names), "roofline_all" (every kernel family),
"kernels" (per-family GPU time), "cpu_baseline" (the reference's own loop on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEFAULT_SOLVER = "fused"     # the library's default; "sweeps" = the five-sweep organisation
DEFAULT_FINISH = "kernel"
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6      # 256 CU x 64 FMA/clk x 2 x 2.4 GHz


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=4096, help="cells per side of one rank's patch")
    ap.add_argument("--iters", type=int, default=50, help="BiCGSTAB iterations per step")
    ap.add_argument("--math", default="fast", choices=["fast", "strict"])
    ap.add_argument("--solver", default=DEFAULT_SOLVER, choices=["sweeps", "fused"],
                    help="organisation of a BiCGSTAB iteration (include/cup2d_hip.h cup2d_solver_kind)")
    ap.add_argument("--finish", default=DEFAULT_FINISH, choices=["launch", "kernel"],
                    help="reduction finish + scalar update: own launch, or by the last workgroup of the sweep")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the domain-decomposed code path (RCCL callbacks, comm stream) even with one rank: a smoke "
                         "test of the N > 1 path on a single-GPU box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--cpu-n", type=int, default=1024, help="grid of the bounded CPU-baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=0, help="OpenMP threads of the CPU baseline (0 = min(cores, 16): the reference's per-block loops stop scaling there)")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON: libraries that talk on fd 1 (RCCL prints a version banner there from C)
    # are sent to stderr for the whole run, the line is written to the saved descriptor at the very end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import cup2d_amd
    from cup2d_amd import lib as L

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    n = args.n

    if world > 1 or args.force_dist:
        import torch.distributed as dist
        from cup2d_amd.distributed import DistributedSimulation, cartesian_dims
        if world == 1 and "MASTER_ADDR" not in os.environ:  # --force-dist outside torch.distributed.run
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        px, py = cartesian_dims(world)
        cx, cy = rank % px, rank // px
        sim = DistributedSimulation(n // 8, n // 8, px, py, nu=1e-3, cfl=0.5, device=local_rank)
        vel = synthetic_velocity(n, n, cx * n, cy * n, px * n, py * n, seed=20250117 + rank)
        par = "cart%dx%d" % (px, py)
    else:
        dist = None
        px = py = 1
        sim = cup2d_amd.Simulation(n // 8, n // 8, nu=1e-3, cfl=0.5, device=local_rank)
        vel = synthetic_velocity(n, n, 0, 0, n, n, seed=20250117)
        par = "single"
    sim.set_math(args.math == "strict")
    fused = args.solver == "fused"  # with ghost blocks: z edges of the boundary blocks are exchanged per sweep
    sim.set_solver(fused=fused, finish_in_kernel=args.finish == "kernel")
    sim.vel = vel
    del vel

    def sync():
        sim.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def one_step():
        return sim.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=args.iters)

    for _ in range(args.warmup):
        one_step()

    def timed_steps(with_kernel_timers):
        sim.set_timing(with_kernel_timers)
        sync()
        t0 = time.perf_counter()
        its = 0
        for _ in range(args.steps):
            its += one_step()["iters"]
        sync()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, its

    # Timed region: EXACTLY --steps steps.  Per-kernel HIP-event pairs are recorded on the launch stream
    # inside it in SAMPLED mode (every launch outside the solver, every 8th BiCGSTAB iteration: a pair costs
    # ~4 us of stream time, so full instrumentation would cost 10 % of the step); the roofline objects are
    # computed from those samples.  The same K steps are repeated afterwards without any events and
    # reported as ms_per_step_no_kernel_timers.
    elapsed, iters = timed_steps(0 if args.no_kernel_timers else 2)
    cells_rank = n * n
    cells = cells_rank * world
    value = cells * args.steps / elapsed / 1e6
    timers = {}
    for i, name in enumerate(L.TIMER_NAMES):
        ms, calls = sim.get_timing(i)
        timers[name] = {"ms_total": round(ms, 4), "launches": calls, "ms_avg": round(ms / calls, 5) if calls else None}
    sim.set_timing(False)
    elapsed_plain = None
    if not args.no_kernel_timers:
        elapsed_plain, _ = timed_steps(False)

    # ---- outside the timed region: the Poisson smoother sweep ---------------------------------------------
    # BASELINE.json configs[1] words the pressure part "50 Jacobi pressure iters/step"; the reference has no smoother
    # (SURVEY.md F4), the step above runs its real solver.  The weighted-Jacobi sweep (csrc/smoother.hip, 24 B/cell)
    # is timed here on the Poisson system of the last step, every launch bracketed by HIP events.
    if not args.no_kernel_timers and world == 1:
        try:
            sim.jacobi_sweeps(5, omega=0.8)
            sim.set_timing(1)
            sim.jacobi_sweeps(50, omega=0.8)
            ms, calls = sim.get_timing(L.TIMER_NAMES.index("smoother"))
            timers["smoother"] = {"ms_total": round(ms, 4), "launches": calls, "ms_avg": round(ms / calls, 5) if calls else None}
            sim.set_timing(False)
        except Exception as e:  # an extra, never the reason the bench line is missing
            timers["smoother"] = {"ms_total": 0.0, "launches": 0, "ms_avg": None, "error": str(e)[:200]}

    # ---- rooflines --------------------------------------------------------------------------------
    # Algorithmic (compulsory) bytes per cell and launch of every kernel family, FP64, halo re-reads
    # excluded (DESIGN.md section 4 derives each line):
    #   advect_stage  stage 1 reads vel 16 + writes mid 16 = 32; stage 2 reads mid 16 + vel 16, writes 16 = 48; mean 40
    #   poisson_rhs   reads vel 16 + pold 8, writes tmp 8 = 32
    #   sweep_A       reads p, nu, r 24 + writes p, z 16 = 40      sweep_C  reads r, nu 16 + writes r, z2 16 = 32
    #   sweep_B / D   reads z (z2) 8 + rhat (r) 8, writes nu (t) 8 = 24
    #   sweep_E       reads x, z, z2, r, t, rhat 48 + writes x, r 16 = 64
    #   smoother      weighted-Jacobi sweep (not part of the step): reads x, b 16 + writes x' 8 = 24
    #   fused solver (krylov_fused.hip): sweep_A = A+B in one launch: reads p, nu, r, rhat 32 + writes p', nu' 16 = 48
    #   sweep_C = C+D: reads r, nu 16 + writes s, t 16 = 32     sweep_E: reads y, p, s, t, rhat 40 + writes y, r 16 = 56
    ALGO_BYTES = {"advect_stage": 40.0, "poisson_rhs": 32.0, "sweep_A": 40.0, "sweep_B": 24.0, "sweep_C": 32.0,
                  "sweep_D": 24.0, "sweep_E": 64.0, "init_residual": 32.0, "smoother": 24.0}
    mk = "true" if args.finish == "kernel" and world == 1 else "false"
    KERNEL_OF = {"advect_stage": "k_advect_diffuse<WenoFast, 1>" if args.math == "fast" else "k_advect_diffuse<WenoStrict, 1>",
                 "poisson_rhs": "k_pressure_rhs<false, true>", "sweep_A": "k_sweepA_fd", "sweep_B": "k_sweepBD<1, %s>" % mk,
                 "sweep_C": "k_sweepC_fd", "sweep_D": "k_sweepBD<2, %s>" % mk, "sweep_E": "k_sweepE<%s>" % mk,
                 "init_residual": "k_init_residual", "smoother": "k_smoother<0, false, 1>"}
    sweeps = ("sweep_A", "sweep_B", "sweep_C", "sweep_D", "sweep_E")
    if fused:
        ALGO_BYTES.update({"sweep_A": 48.0, "sweep_C": 32.0, "sweep_E": 56.0})
        del ALGO_BYTES["sweep_B"], ALGO_BYTES["sweep_D"]
        # fused kernels: MERGE 1 = the last workgroup finishes the reduction and updates the scalars (one GPU),
        # 2 = it sums the rank's partials, all-reduce + scalar kernel follow (N GPUs), 0 = finish launch
        mf = 0 if args.finish != "kernel" else (1 if dist is None else 2)
        KERNEL_OF.update({"sweep_A": "k_fused<0, %d>" % mf, "sweep_C": "k_fused<1, %d>" % mf, "sweep_E": "k_sweepE_y<%d>" % mf})
        sweeps = ("sweep_A", "sweep_C", "sweep_E")
    finish_launches = 0 if mk == "true" else 3
    # HBM bytes per launch measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH x2 on
    # gfx950), summarised by tools/prof_summary.py from the same bench command: profiles/<tag>_pmc_traffic.json
    traffic_tab, traffic_src = {}, {}
    prof_dir = os.path.join(ROOT, "profiles")
    if os.path.isdir(prof_dir):  # every committed summary; kernels are looked up by name (a later file wins)
        for f in sorted(f for f in os.listdir(prof_dir) if f.endswith("_pmc_traffic.json")):
            try:
                for k, v in json.load(open(os.path.join(prof_dir, f)))["kernels"].items():
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
        tr = traffic_tab.get(KERNEL_OF[fam], {}).get("hbm_bytes") if n == 4096 and world == 1 else None
        return {"kernel": KERNEL_OF[fam], "family": fam, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": tr,
                "traffic_source": traffic_src.get(KERNEL_OF[fam]) if tr else None,
                "bytes_per_cell": ALGO_BYTES[fam], "avg_launch_ms": round(sec * 1e3, 4), "launches": t["launches"],
                "share_of_gpu_time": None}

    # GPU time per step of a family = its (sampled) average launch x launches per step
    per_step = {"advect_stage": 2, "poisson_rhs": 1, "init_residual": 1, "project": 1, "reduce": 3,
                "sweep_A": args.iters, "sweep_B": args.iters, "sweep_C": args.iters, "sweep_D": args.iters,
                "sweep_E": args.iters, "scalars": finish_launches * args.iters + 1, "halo": 0}
    if fused:
        per_step["sweep_B"] = per_step["sweep_D"] = 0
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
    # 32 T lane-instr/s sustained on this chip; 39.3 T at the nominal 2.4 GHz) -- DESIGN.md section 4.1
    north = dict(all_roof["advect_stage"]) if "advect_stage" in all_roof else None
    if north:
        sec = north["avg_launch_ms"] * 1e-3
        # FP64 VALU instructions executed per cell in a block whose velocity components do not change sign
        # (counted in the gfx950 ISA of advect.hip; SQ_INSTS_VALU measures 314 VALU instructions of all kinds)
        fp64_per_cell = 241.0 if args.math == "fast" else 565.0
        rate = fp64_per_cell * cells_rank / sec / 1e12
        north.update({"mcells_per_s": round(cells_rank / sec / 1e6, 1), "fp64_instr_per_cell": fp64_per_cell,
                      "fp64_T_lane_instr_per_s": round(rate, 2), "fp64_frac_of_measured_ceiling_32T": round(rate / 32.0, 4),
                      "fp64_frac_of_nominal_39.3T": round(rate / 39.3, 4)})
    # one BiCGSTAB iteration = sweeps A..E + 3 scalar kernels (sum of the sampled average durations)
    it_bytes = sum(ALGO_BYTES[k] for k in sweeps)
    solver = None
    if iters and all(timers[k]["launches"] for k in sweeps + ("scalars",)):
        t_it = (sum(timers[k]["ms_avg"] for k in sweeps) + finish_launches * timers["scalars"]["ms_avg"]) * 1e-3
        solver = {"iterations": iters, "ms_per_iteration": round(t_it * 1e3, 4), "bytes_per_cell_iteration": it_bytes,
                  "achieved_GBs": round(it_bytes * cells_rank / t_it / 1e9, 1),
                  "frac_hbm": round(it_bytes * cells_rank / t_it / 1e9 / HBM_PEAK_GBS, 4),
                  "mcell_iterations_per_s": round(cells_rank / t_it / 1e6, 1)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from oracle import oracle as O
            if O.have_reference():
                thr = args.cpu_threads or min(os.cpu_count() or 1, 16)
                r = O.ref_step_time(args.cpu_n, steps=2, max_iter=args.iters, threads=thr, timeout=150)
                cpu = {"value": round(args.cpu_n ** 2 / r["median_step_s"] / 1e6, 4), "unit": "Mcell-updates/s",
                       "cores": r["threads"], "kind": "reference",
                       "sample": "reference main.cpp time loop (OpenMP functors; Poisson = CPU port of cuda.cu, %d iters) "
                                 "at %d^2, median of %d steps" % (args.iters, args.cpu_n, r["timed_steps"])}
                # the reference's own functors alone (computeA over all blocks, OpenMP), median of reps
                fb = O.ref_bench(args.cpu_n, reps=5, threads=thr)
                cpu["functors_mcells_per_s"] = {"advect_diffuse": round(fb["advect_diffuse_mcells"], 2),
                                                "pressure_rhs1": round(fb["pressure_rhs1_mcells"], 2),
                                                "sample": "computeA<..>(KernelAdvectDiffuse / pressure_rhs1) at %d^2, median of 5"
                                                          % args.cpu_n}
            else:
                t1 = time.perf_counter()
                v0 = O.taylor_green(512)
                O.step(v0, np.zeros((512, 512)), 1.0 / 512, 1e-3, 0.5, tol=0.0, max_restarts=100, max_iter=args.iters)
                dt_cpu = time.perf_counter() - t1
                cpu = {"value": round(512 * 512 / dt_cpu / 1e6, 4), "unit": "Mcell-updates/s", "cores": os.cpu_count(),
                       "kind": "port", "sample": "oracle/cup2d_oracle.c oracle_step at 512^2, 1 step"}
        except Exception as e:  # the baseline is informative; never fail the bench on it
            cpu = {"error": str(e)[:200]}

    if rank == 0:
        out = {
            "metric": "Mcell-updates/sec (advect-diffuse+Poisson sweep) at 4096^2",
            "value": round(value, 3), "unit": "Mcell-updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "ms_per_step_no_kernel_timers": round(elapsed_plain / args.steps * 1e3, 3) if elapsed_plain else None,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%dx%d uniform cells per GPU (%d^2 8x8 blocks), nu=1e-3, CFL 0.5; step = dt + RK2 WENO5 "
                                   "advect-diffuse + Poisson rhs + %d BiCGSTAB iters (block-Jacobi) + projection"
                                   % (n, n, n // 8, args.iters),
                       "global_cells": cells, "parallelism": par, "math": args.math, "bicgstab_iters_per_step": args.iters,
                       "solver": "fused" if fused else "sweeps", "finish": "kernel" if mk == "true" else "launch"},
            "roofline": roofline, "roofline_north_star": north, "roofline_all": all_roof, "solver": solver,
            "kernels": timers, "cpu_baseline": cpu,
        }
    sim.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
