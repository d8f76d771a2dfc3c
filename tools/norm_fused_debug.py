#!/usr/bin/env python3
"""Debug of the norm-in-consumer hand-over: where (row = producer, column strip = consumer workgroup) the fused launch differs
from the two launches, and whether the packed rows in MEMORY are right afterwards."""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aphrodite_engine_amd import _custom_ops as ops, _lib

lib = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
name = sys.argv[2] if len(sys.argv) > 2 else "qkv"
K, N = {"gate_up": (4096, 28672), "qkv": (4096, 6144)}[name]
G = K // 128
g = torch.Generator(device="cuda").manual_seed(0)
qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (G, N // 8), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
sc = (torch.rand(G, N, generator=g, device="cuda") * 0.01).half()
if name == "gate_up":
    qw, qz, sc = ops.interleave_gate_up(qw, qz, sc)
strip = ops.wna16_strip_relayout(qw, M, G)
mode = "silu" if name == "gate_up" else "slabs"
for rep in range(4):
    slabs = torch.randn(4, M, K, device="cuda", generator=g) * 0.25
    res0 = torch.randn(M, K, device="cuda", generator=g).half()
    w = (1 + 0.1 * torch.randn(K, device="cuda", generator=g)).half()
    r1 = res0.clone()
    pk_ref, _ = ops.fused_add_rms_norm_pack(None, slabs, r1, True, w, 1e-5)
    want = ops.wna16_gemm_resident(pk_ref, M, K, strip, qz, sc, 1, mode=mode, strip_layout=True)
    want = want if mode == "silu" else want[0]
    torch.cuda.synchronize()
    for prefill in (0.0, float("nan")):
        nb = lib.aphro_wna16_packed_a_bytes(M, K) // 2
        pk = torch.full((nb,), prefill, dtype=torch.float16, device="cuda")
        sync = torch.zeros(1, dtype=torch.int32, device="cuda")
        r2 = res0.clone()
        ks = lib.aphro_wna16_resident_ksplit(M, N, K, G)
        out = torch.zeros((ks, M, N), dtype=torch.float32, device="cuda") if mode == "slabs" else \
            torch.zeros(lib.aphro_wna16_packed_a_bytes(M, N // 2) // 2, dtype=torch.float16, device="cuda")
        torch.cuda.synchronize()
        rc = lib.aphro_wna16_gemm_norm_fused(slabs.data_ptr(), 4, r2.data_ptr(), w.data_ptr(), 1e-5, pk.data_ptr(), strip.data_ptr(),
                                             qz.data_ptr(), sc.data_ptr(), out.data_ptr() if mode == "slabs" else None,
                                             out.numel() * 4 if mode == "slabs" else 0, out.data_ptr() if mode == "silu" else None,
                                             M, N, K, G, 1, 0, sync.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert rc == 0, lib.aphro_last_error()
        mt = (M + 15) // 16
        def rows(p):      # [mt*16, K] view of a packed buffer
            a = p.cpu().numpy().view(np.uint16)[: (K // 128) * 4 * mt * 64 * 8].reshape(K // 128, 4, mt, 4, 16, 8)
            return a.transpose(2, 4, 0, 3, 1, 5).reshape(mt * 16, K)[:M]
        mem_ok = np.array_equal(rows(pk), rows(pk_ref))
        if not mem_ok and rep == 0:
            a_, b_ = rows(pk), rows(pk_ref)
            d = a_ != b_
            print("   packed mismatches:", int(d.sum()), "of", d.size, "rows", np.nonzero(d.any(1))[0].tolist()[:8],
                  "k (first 24)", np.nonzero(d.any(0))[0].tolist()[:24], "k count", int(d.any(0).sum()))
            r0 = int(np.nonzero(d.any(1))[0][0]); ks_ = np.nonzero(d[r0])[0][:8]
            print("   row", r0, "k", ks_.tolist(), "got", a_[r0, ks_].view(np.float16).tolist(), "want", b_[r0, ks_].view(np.float16).tolist())
        res_ok = torch.equal(r1, r2)
        if mode == "slabs":
            bad = ~((out == want) | (out.isnan() & want.isnan()))
            bad = bad.any(0).cpu().numpy()          # [M, N]
            cw = 64
        else:
            a = out.cpu().numpy().view(np.uint16); b = want.cpu().numpy().view(np.uint16)
            n2 = N // 2
            def rows2(p):
                x = p[: (n2 // 128) * 4 * mt * 64 * 8].reshape(n2 // 128, 4, mt, 4, 16, 8)
                return x.transpose(2, 4, 0, 3, 1, 5).reshape(mt * 16, n2)[:M]
            bad = rows2(a) != rows2(b)
            cw = 56
        nbad = int(bad.sum())
        by_row = bad.sum(1)
        by_strip = bad.reshape(M, -1, cw).sum((0, 2))
        print(f"rep {rep} prefill {prefill}: sync {int(sync.item())} packed-in-memory ok {mem_ok} residual ok {res_ok} bad {nbad} "
              f"rows with bad {np.nonzero(by_row)[0].tolist()[:40]} strips with bad {np.nonzero(by_strip)[0].tolist()[:64]} "
              f"nan {int(np.isnan(out.float().cpu().numpy()).sum()) if mode == 'slabs' else -1}")
