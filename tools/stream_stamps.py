#!/usr/bin/env python3
"""Round-5 lab: where does a W4A16 stream-kernel launch spend its time before the first weight byte is requested?
Needs tools/bin/libaphro_stamps.so (tools/build_stamp_lib.sh: the product library with -DAPHRO_STREAM_STAMPS).  Each
workgroup's wave 0 stamps (shader clock, relative to its own first instruction): 1 = kernel arguments arrived and strip
known, 2 = prologue loads issued, 5 = k loop done, 6 = reduce barrier passed, 7 = stores issued.  Weights cold (cycling over
> 640 MB), HIP-graph replay.  Scenario "same": one projection's kernel back to back; "layer": qkv, o, gate_up, down in turn
(what a decode layer does to the instruction cache).  One JSON line per (scenario, projection) -> gpurun_out/stream_stamps.jsonl"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("APHRODITE_MI355X_LIB", os.path.join(ROOT, "tools", "bin", "libaphro_stamps.so"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from aphrodite_engine_amd import _custom_ops as ops, _lib  # noqa: E402
from tools.prefetch_lab import timeit  # noqa: E402

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}
M = 32
OUT = []


def stamps(nwg):
    lib = _lib.lib()
    buf = (ctypes.c_uint32 * (1024 * 8))()
    fn = lib.aphro_lab_stream_stamps
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
    assert fn(buf, 1024 * 8) == 0
    t = torch.tensor(list(buf), dtype=torch.int64).view(1024, 8)[:nwg]
    q = lambda col, f: int(torch.quantile(t[:, col].double(), f).item())
    res = {f"s{c}": [q(c, 0.1), q(c, 0.5), q(c, 0.9)] for c in (1, 2, 5, 6, 7)}
    res["entry_spread_p10_p90"] = q(0, 0.9) - q(0, 0.1)
    return res


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    sets = {}
    for name, (K, N) in SHAPES.items():
        G = K // 128
        n = max(2, (640 << 20) // (K * N // 2))
        ws = []
        for _ in range(n):
            qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
            qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (G, N // 8), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
            sc = (torch.rand(G, N, generator=g, device="cuda") * 0.01).half()
            ws.append((ops.wna16_strip_relayout(qw, M, G), qz, sc))
            del qw
        a = torch.randn(M, K, device="cuda", dtype=torch.float16)
        sets[name] = (ws, ops.wna16_pack_a(a), K, N)

    def launch(name, i):
        ws, pk, K, N = sets[name]
        qw, qz, sc = ws[i % len(ws)]
        ops.wna16_gemm_resident(pk, M, K, qw, qz, sc, 1, mode="silu" if name == "gate_up" else "slabs", strip_layout=True)

    for name in SHAPES:
        n = len(sets[name][0])

        def same(name=name, n=n):
            for i in range(n):
                launch(name, i)
        t = timeit(same, n)
        nwg = SHAPES[name][1] // (112 if name == "gate_up" else 64)
        r = dict(scenario="same", kernel=name, us_per_launch=round(t * 1e6, 2), **stamps(min(1024, nwg)))
        OUT.append(r)
        print(json.dumps(r), flush=True)
    # a layer's order; the stamps read are the LAST launch of each kernel, so run the sequence ending with each in turn
    order = ["qkv", "o", "gate_up", "down"]
    for last in order:
        k = order.index(last)
        seq = (order[k + 1:] + order[:k + 1]) * 8

        def layer(seq=seq):
            for i, nm in enumerate(seq):
                launch(nm, i // 4)
        t = timeit(layer, len(seq))
        nwg = SHAPES[last][1] // (112 if last == "gate_up" else 64)
        r = dict(scenario="layer", kernel=last, us_per_launch_avg_of_4=round(t * 1e6, 2), **stamps(min(1024, nwg)))
        OUT.append(r)
        print(json.dumps(r), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "stream_stamps.jsonl"), "w") as f:
        for r in OUT:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
