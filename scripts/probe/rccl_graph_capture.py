"""Probe: does this RCCL accept its operations inside a HIP stream capture?  One rank,
torch.distributed nccl: an all_reduce and a send/recv pair to the rank itself recorded in a
torch.cuda.CUDAGraph and replayed.  Usage: python scripts/probe/rccl_graph_capture.py
Outcome (RCCL 2.26.6, ROCm 7.0.2): a one-rank all_reduce is elided (torch warns that the graph is
empty), so this says nothing; the library's own attempt -- ncclSend / ncclRecv groups of a rank to
itself recorded on a forked halo stream -- captures every group and then crashes inside
hipStreamEndCapture, while the same two-stream capture with device copies in place of the groups
instantiates and replays (csrc/cmdg.hip graph_eligible)."""
import os
import sys
import faulthandler

faulthandler.enable()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29588")
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
x = torch.ones(1024, device="cuda:0", dtype=torch.float64)
dist.all_reduce(x)          # warm-up outside any capture
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
print("capturing all_reduce ...", flush=True)
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        dist.all_reduce(x)
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
print("all_reduce in a graph: ok", x[:2].tolist(), flush=True)
dist.destroy_process_group()
