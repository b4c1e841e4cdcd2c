"""Which HIP / HSA / RCCL copies does a process end up with, and does the in-library communicator come up?
usage: python tools/rccl_probe.py ours_first|torch_first|torch_cuda_first [x]   (run on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", RANK="0", WORLD_SIZE="1", GLOO_SOCKET_IFNAME="lo", NCCL_SOCKET_IFNAME="lo")


def maps(tag):
    seen = sorted({l.split()[-1] for l in open("/proc/self/maps") if any(k in l for k in ("librccl", "libamdhip64", "libhsa-runtime"))})
    print(tag, seen, flush=True)


import numpy as np
if mode != "ours_first":
    import torch
    import torch.distributed as dist
    if mode == "torch_cuda_first":
        torch.zeros(1, device="cuda")
import cup2d_amd
with cup2d_amd.Simulation(8, nu=1e-3) as s:
    s.vel = np.random.default_rng(0).uniform(-1, 1, (64, 64, 2))
    r0 = s.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=5)
maps("after plain context:")
import torch
import torch.distributed as dist
if mode == "ours_first" and len(sys.argv) > 2:
    try:
        print("torch cuda after ours:", torch.zeros(1, device="cuda").item())
    except Exception as e:
        print("torch cuda after ours FAILED:", repr(e)[:200])
dist.init_process_group("gloo", rank=0, world_size=1)
from cup2d_amd.distributed import DistributedSimulation
try:
    with DistributedSimulation(8, 8, 1, 1, nu=1e-3, comm="rccl") as d:
        d.vel = np.random.default_rng(0).uniform(-1, 1, (64, 64, 2))
        r1 = d.step(tol=0.0, rel_tol=0.0, max_restarts=100, max_iter=5)
        print("rccl path ok, same err:", r1["err"] == r0["err"])
except Exception as e:
    print("rccl path FAILED:", repr(e)[:300])
maps("at end:")
dist.destroy_process_group()
