#!/usr/bin/env python3
"""Golden vectors for the GPTQ exllama path from the REFERENCE's own CUDA kernels, executed on the host.

Run in the build container only (needs /root/reference):
    make -C oracle ref && python tests/golden/make_golden_gptq.py

oracle/_ref/libaphro_ref_gptq.so is the reference's kernels/quantization/gptq/q_gemm.cu (+ qdq_4.cuh,
matrix_view.cuh) compiled for the CPU against oracle/cuda_host_shim/ (oracle/Makefile); this script
feeds it seeded inputs and stores what it returns in tests/golden/gptq_ref.npz:

  shuffle_*      ops.gptq_shuffle               shuffle_exllama_weight (q_gemm.cu:1822-1872): make_sequential + shuffle_4bit_8
  recon_exl_*    reconstruct_exllama_4bit_kernel (q_gemm.cu:856-965)   fp16 W from the shuffled weight (+ act-order perm)
  recon_gptq_*   reconstruct_gptq_kernel<q4,4>  (q_gemm.cu:1394-1434)  fp16 W = (q - (z + 1)) * s from checkpoint order + g_idx
  gemm_*         gemm_half_q_half_gptq_4bit_kernel (q_gemm.cu:190-326) the M <= 50 exllama GEMM, rows 1 / 5 / 8 / 13

These pin oracle/quant.py (tests/test_oracle_golden.py::test_gptq_*_reference_kernels); the same tests
call the library directly when it is present.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "..", "oracle", "_ref", "libaphro_ref_gptq.so")


def load():
    lib = ctypes.CDLL(LIB)
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def ref_shuffle(lib, qweight, perm):
    qw = np.ascontiguousarray(qweight).copy()
    k8, n = qw.shape
    pp = np.ascontiguousarray(perm, dtype=np.int32) if perm is not None else None
    lib.ref_gptq_shuffle(_p(qw), _p(pp), ctypes.c_int(k8 * 8), ctypes.c_int(n))
    return qw


def ref_recon_exllama(lib, qw_shuf, perm, qzeros, scales):
    k8, n = qw_shuf.shape
    g = scales.shape[0]
    out = np.zeros((k8 * 8, n), dtype=np.float16)
    pp = np.ascontiguousarray(perm, dtype=np.int32) if perm is not None else None
    lib.ref_gptq_reconstruct_exllama(_p(np.ascontiguousarray(qw_shuf)), _p(pp), _p(np.ascontiguousarray(qzeros)),
                                     _p(np.ascontiguousarray(scales)), ctypes.c_int(k8 * 8), ctypes.c_int(n),
                                     ctypes.c_int(g), _p(out))
    return out


def ref_recon_gptq(lib, qweight, qzeros, scales, g_idx):
    k8, n = qweight.shape
    g = scales.shape[0]
    out = np.zeros((k8 * 8, n), dtype=np.float16)
    lib.ref_gptq_reconstruct(_p(np.ascontiguousarray(qweight)), _p(np.ascontiguousarray(qzeros)),
                             _p(np.ascontiguousarray(scales)), _p(np.ascontiguousarray(g_idx, dtype=np.int32)),
                             ctypes.c_int(k8 * 8), ctypes.c_int(n), ctypes.c_int(g), _p(out))
    return out


def ref_recon_gptq_bits(lib, qweight, qzeros, scales, g_idx, bits):
    """reconstruct_gptq for 2 / 3 / 8-bit checkpoints (q_gemm.cu:1394-1505)."""
    rows, n = qweight.shape
    k = rows * 32 // bits
    g = scales.shape[0]
    out = np.zeros((k, n), dtype=np.float16)
    rc = lib.ref_gptq_reconstruct_bits(_p(np.ascontiguousarray(qweight)), _p(np.ascontiguousarray(qzeros)),
                                       _p(np.ascontiguousarray(scales)), _p(np.ascontiguousarray(g_idx, dtype=np.int32)),
                                       ctypes.c_int(k), ctypes.c_int(n), ctypes.c_int(g), ctypes.c_int(bits), _p(out))
    assert rc == 0
    return out


def make_case_bits(rng, k, n, group_size, act_order, bits):
    g = k // group_size
    qweight = rng.integers(0, 2**32, size=(k * bits // 32, n), dtype=np.uint32).view(np.int32)
    qzeros = rng.integers(0, 2**32, size=(g, n * bits // 32), dtype=np.uint32).view(np.int32)
    scales = (rng.uniform(0.002, 0.02, size=(g, n))).astype(np.float16)
    if act_order:
        g_idx = rng.permutation(np.arange(k) // group_size).astype(np.int32)
    else:
        g_idx = (np.arange(k) // group_size).astype(np.int32)
    return qweight, qzeros, scales, g_idx


def ref_gemm(lib, a, qw_shuf, qzeros, scales, perm):
    m, k = a.shape
    n = qw_shuf.shape[1]
    g = scales.shape[0]
    c = np.full((m, n), np.float16(np.nan), dtype=np.float16)      # the kernel zeroes its output itself
    pp = np.ascontiguousarray(perm, dtype=np.int32) if perm is not None else None
    rc = lib.ref_gptq_gemm_exllama(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(qw_shuf)),
                                   _p(np.ascontiguousarray(qzeros)), _p(np.ascontiguousarray(scales)), _p(pp), _p(c),
                                   ctypes.c_int(m), ctypes.c_int(n), ctypes.c_int(k), ctypes.c_int(g))
    assert rc == 0
    return c


def make_case(rng, k, n, group_size, act_order):
    g = k // group_size
    qweight = rng.integers(0, 2**32, size=(k // 8, n), dtype=np.uint32).view(np.int32)
    qzeros = rng.integers(0, 2**32, size=(g, n // 8), dtype=np.uint32).view(np.int32)
    scales = (rng.uniform(0.002, 0.02, size=(g, n))).astype(np.float16)
    if act_order:
        g_idx = rng.permutation(np.arange(k) // group_size).astype(np.int32)     # row -> group, shuffled
        perm = np.argsort(g_idx, kind="stable").astype(np.int32)                 # gptq.py:219-221
    else:
        g_idx = (np.arange(k) // group_size).astype(np.int32)
        perm = None
    return qweight, qzeros, scales, g_idx, perm


def main():
    lib = load()
    rng = np.random.default_rng(20240924)
    out = {}
    cases = {"a": (256, 96, 128, False), "b": (256, 96, 64, True), "c": (512, 512, 128, False), "d": (384, 256, 32, True)}
    for name, (k, n, gs, ao) in cases.items():
        qweight, qzeros, scales, g_idx, perm = make_case(rng, k, n, gs, ao)
        shuf = ref_shuffle(lib, qweight, perm)
        out[f"{name}_qweight"], out[f"{name}_qzeros"], out[f"{name}_scales"] = qweight, qzeros, scales
        out[f"{name}_g_idx"] = g_idx
        out[f"{name}_perm"] = perm if perm is not None else np.zeros((0,), np.int32)
        out[f"{name}_shuffle"] = shuf
        out[f"{name}_recon_exl"] = ref_recon_exllama(lib, shuf, perm, qzeros, scales)
        out[f"{name}_recon_gptq"] = ref_recon_gptq(lib, qweight, qzeros, scales, g_idx)
        if name in ("c", "d"):
            for m in (1, 5, 8, 13):
                a = rng.standard_normal((m, k)).astype(np.float16)
                out[f"{name}_gemm_a{m}"] = a
                out[f"{name}_gemm_c{m}"] = ref_gemm(lib, a, shuf, qzeros, scales, perm)
    # 2 / 3 / 8-bit checkpoints: the layout definition (reconstruct_gptq) -> gptq_ref_bits.npz
    outb = {}
    for bits in (2, 3, 8):
        for name, (k, n, gs, ao) in {"p": (128, 64, 32, False), "q": (256, 96, 128, True)}.items():
            qweight, qzeros, scales, g_idx = make_case_bits(rng, k, n, gs, ao, bits)
            key = f"b{bits}{name}"
            outb[f"{key}_qweight"], outb[f"{key}_qzeros"], outb[f"{key}_scales"], outb[f"{key}_g_idx"] = qweight, qzeros, scales, g_idx
            outb[f"{key}_recon_gptq"] = ref_recon_gptq_bits(lib, qweight, qzeros, scales, g_idx, bits)
    np.savez_compressed(os.path.join(HERE, "gptq_ref_bits.npz"), **outb)
    np.savez_compressed(os.path.join(HERE, "gptq_ref.npz"), **out)
    print("wrote gptq_ref.npz:", {k: v.shape for k, v in out.items() if k.endswith("shuffle") or "gemm_c" in k})


if __name__ == "__main__":
    main()
