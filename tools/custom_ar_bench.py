#!/usr/bin/env python3
"""Latency of the peer-access all-reduce with N ranks sharing ONE GPU (real IPC mappings, no xGMI hop:
a lower bound of the kernel + flag protocol, not a link measurement).  64 calls per HIP-graph replay.
usage: custom_ar_bench.py [world] [numel_f16]"""
import os
import socket
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, numel):
    import torch.distributed as dist
    from aphrodite_engine_amd.distributed.custom_all_reduce import CustomAllreduce
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    ca = CustomAllreduce(dist.group.WORLD, dev)
    x = torch.ones(numel, dtype=torch.float16, device=dev)
    g = torch.cuda.CUDAGraph()
    calls = 64
    with ca.capture():
        ca.custom_all_reduce(x)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
            y = x
            for _ in range(calls):
                y = ca.custom_all_reduce(x)
    for _ in range(3):
        dist.barrier()
        g.replay()
    torch.cuda.synchronize()
    reps = 20
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (reps * calls)
    ca.check()
    assert float(y[0]) == world
    if rank == 0:
        one = "one-shot" if numel * 2 <= (512 * 1024 if world <= 4 else 256 * 1024) or world == 2 else "two-shot"
        print(f"custom all-reduce, {world} ranks on one GPU, {numel * 2 // 1024} KiB f16 ({one}): {dt * 1e6:.1f} us per call")
    ca.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    numel = int(sys.argv[2]) if len(sys.argv) > 2 else 32 * 4096
    os.environ.setdefault("APHRODITE_CUSTOM_AR_TIMEOUT_MS", "5000")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(worker, args=(world, port, numel), nprocs=world, join=True)
