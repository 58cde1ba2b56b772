#!/usr/bin/env python3
"""How long the first steps of a fresh process take: per-step GPU time (event pair around each step, averaged in groups) and host enqueue time,
from the first step on.  Explains the gap between bench.py's `value` (steps 6..25 of a process under --warmup 5 --steps 20) and `sustained`."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
grp = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist, comm = None, None
if os.environ.get("DP", "") in ("abi", "torch"):      # one-rank data-parallel exchange forced on (bench.py --force-dist)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    if os.environ["DP"] == "abi":
        dist.init_process_group("gloo", rank=0, world_size=1)
        from osvos_pytorch_amd.parallel import AbiCommunicator
        if os.environ.get("BARRIER") == "1":
            dist.barrier(); dist.barrier()
        comm = AbiCommunicator(0, 1, dev)
        if os.environ.get("TINY") == "1":
            t = torch.ones(1, device=dev)
            comm.all_reduce(t)
            torch.cuda.synchronize()
            print("tiny all-reduce:", t.item())
    else:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        dist.barrier()
wl = bench.Workload("online", os.environ.get("OSVOS_PRECISION", "fp32x3"), 480, 854, 1, 0, 0, 0, dev, 0, dist, dist is not None, comm=comm)
if os.environ.get("PROBE_FIRST") == "1":
    print("pipe probe first:", bench.pipe_sustained_tflops(dev))
for rnd in range(int(os.environ.get("ROUNDS", "1"))):
    if rnd:
        time.sleep(float(os.environ.get("IDLE_S", "0.5")))
        print("---- round %d after %.1f s idle (same process, same workload object)" % (rnd, float(os.environ.get("IDLE_S", "0.5"))))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    host = []
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(n):
        t = time.perf_counter()
        wl.step()
        host.append(time.perf_counter() - t)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    for g in range(0, n, grp):
        print("steps %3d-%3d: gpu %.3f ms/step   host enqueue %.3f ms/step" % (g, g + grp - 1, sum(ms[g:g + grp]) / grp, 1e3 * sum(host[g:g + grp]) / grp))
