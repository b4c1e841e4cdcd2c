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

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (fused advect-diffuse stage kernel, the
north-star kernel, timed live with HIP events on its launch stream), "kernels" (per-family GPU time),
"cpu_baseline" (the reference's own loop on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=1024, help="grid of the bounded CPU-baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=0, help="OpenMP threads of the CPU baseline (0 = min(cores, 16): the reference's per-block loops stop scaling there)")
    args = ap.parse_args()

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

    if world > 1:
        import torch.distributed as dist
        from cup2d_amd.distributed import DistributedSimulation, cartesian_dims
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
    sim.set_timing(True)
    sync()
    t0 = time.perf_counter()
    iters = 0
    for _ in range(args.steps):
        iters += one_step()["iters"]
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    cells_rank = n * n
    cells = cells_rank * world
    value = cells * args.steps / elapsed / 1e6
    timers = {}
    for i, name in enumerate(L.TIMER_NAMES):
        ms, calls = sim.get_timing(i)
        timers[name] = {"ms_total": round(ms, 4), "launches": calls, "ms_avg": round(ms / calls, 5) if calls else None}
    sim.set_timing(False)

    # roofline of the north-star kernel: fused WENO5 advect-diffuse RK stage.  Algorithmic bytes per
    # cell: stage 1 reads vel 16 B (vold == vel) + writes 16 B = 32; stage 2 reads mid 16
    # + vold 16 + writes 16 = 48; average 40 B/cell/launch.  ~574 FP64 flops/cell (32 of them divisions)
    # in the reference's formulation.
    adv = timers["advect_stage"]
    roofline = None
    if adv["launches"]:
        t_launch = adv["ms_total"] / adv["launches"] * 1e-3
        gbs = 40.0 * cells_rank / t_launch / 1e9
        roofline = {"kernel": "k_advect_diffuse (fused RK stage)", "bound": "hbm", "achieved": round(gbs, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                    "bytes_per_cell": 40.0, "avg_launch_ms": round(t_launch * 1e3, 4),
                    "mcells_per_s": round(cells_rank / t_launch / 1e6, 1),
                    "fp64_frac_at_574_flop_per_cell": round(574.0 * cells_rank / t_launch / 1e12 / FP64_PEAK_TFLOPS, 4)}
    # one BiCGSTAB iteration = sweeps A..E; algorithmic bytes/cell: A 40, B 24, C 32, D 24, E 56 = 176
    it_ms = sum(timers[k]["ms_total"] for k in ("sweep_A", "sweep_B", "sweep_C", "sweep_D", "sweep_E", "scalars"))
    solver = None
    if iters:
        t_it = it_ms / iters * 1e-3
        solver = {"iterations": iters, "ms_per_iteration": round(t_it * 1e3, 4), "bytes_per_cell_iteration": 176,
                  "achieved_GBs": round(176.0 * cells_rank / t_it / 1e9, 1),
                  "frac_hbm": round(176.0 * cells_rank / t_it / 1e9 / HBM_PEAK_GBS, 4),
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
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%dx%d uniform cells per GPU (%d^2 8x8 blocks), nu=1e-3, CFL 0.5; step = dt + RK2 WENO5 "
                                   "advect-diffuse + Poisson rhs + %d BiCGSTAB iters (block-Jacobi) + projection"
                                   % (n, n, n // 8, args.iters),
                       "global_cells": cells, "parallelism": par, "math": args.math, "bicgstab_iters_per_step": args.iters},
            "roofline": roofline, "solver": solver, "kernels": timers, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    sim.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
