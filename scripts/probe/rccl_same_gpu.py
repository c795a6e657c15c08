"""Probe: do two processes on ONE GPU get an RCCL communicator (torch.distributed nccl)?
Usage: python scripts/probe/rccl_same_gpu.py  (spawns 2 ranks on cuda:0, 90 s watchdog).
Outcome on the MI355X box (RCCL 2.26.6): no -- ncclInvalidUsage, "Duplicate GPU detected : rank 1
and rank 0 both on CUDA device 5a000".  So the N > 1 RCCL path cannot be run with real neighbours
on a one-GPU box; what stands in for it: DESIGN.md section 4."""
import os
import subprocess
import sys

if "RANK" not in os.environ:
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, __file__], env=env))
    codes = []
    for p in procs:
        try:
            codes.append(p.wait(timeout=90))
        except subprocess.TimeoutExpired:
            p.kill()
            codes.append("timeout")
    print("exit codes", codes)
    sys.exit(0)

import torch
import torch.distributed as dist
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda:0"))
    x = torch.full((4,), float(rank + 1), device="cuda:0")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print("rank", rank, "all_reduce ->", x.tolist(), flush=True)
    a = torch.arange(8, dtype=torch.float64, device="cuda:0") + 100 * rank
    b = torch.zeros(8, dtype=torch.float64, device="cuda:0")
    ops = [dist.P2POp(dist.isend, a, 1 - rank), dist.P2POp(dist.irecv, b, 1 - rank)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize()
    print("rank", rank, "sendrecv ->", b.tolist(), flush=True)
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print("rank", rank, "FAILED:", repr(e)[:400], flush=True)
    sys.exit(3)
