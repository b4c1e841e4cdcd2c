#!/usr/bin/env python3
"""Host-side cost of the torch.distributed calls the comm callbacks make (one rank, nccl = RCCL): development aid."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
send = torch.zeros(1536 * 64, dtype=torch.float64, device="cuda")
recv = torch.zeros_like(send)
red = torch.zeros(8, dtype=torch.float64, device="cuda")
comm, comp = torch.cuda.Stream(), torch.cuda.Stream()
ev1, ev2 = torch.cuda.Event(), torch.cuda.Event()
ops = [dist.P2POp(dist.irecv, recv[:512 * 64], 0), dist.P2POp(dist.isend, send[:512 * 64], 0)]


def exchange():
    ev1.record(comp)
    comm.wait_event(ev1)
    with torch.cuda.stream(comm):
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        ev2.record(comm)


def wait():
    comp.wait_event(ev2)


def allreduce():
    with torch.cuda.stream(comp):
        dist.all_reduce(red[0:2])


for name, fn in (("exchange (1 peer, self)", exchange), ("wait", wait), ("allreduce", allreduce)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    n = 500
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / n
    print("%-26s host %.1f us per call, incl. GPU drain %.1f us" % (name, host * 1e6, total * 1e6), flush=True)
dist.destroy_process_group()
