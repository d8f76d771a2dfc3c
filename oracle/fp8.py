"""Oracle: FP8 codec (OCP e4m3fn / e5m2), scaled quantisation, scaled matmul.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

gfx950 (CDNA4) implements the OCP formats natively, so this framework follows
the reference's *NVIDIA* (OCP) semantics, not its MI300 e4m3fnuz branch
(DESIGN.md "FP8 flavour").

Reference anchors (relative to /root/reference):
  scaled_fp8_conversion      kernels/quantization/fp8/common.cu:46-64
  per-token / per-tensor     common.cu:72-256, tests/kernels/quant_utils.py:18-81
  scaled_fp8_quant wrapper   aphrodite/_custom_ops.py:632-685
  KV store  fp8(x / scale)   kernels/cache_kernels.cu:198-201,
                             fp8/nvidia/quant_utils.cuh:440-460 (__NV_SATFINITE)
  KV load   float(fp8)*scale fp8/nvidia/quant_utils.cuh:292-313
  scaled_mm                  tests/kernels/test_cutlass.py:36-47,
                             quantization/utils/w8a8_utils.py:143-183
"""
import numpy as np
import torch

E4M3_MAX = 448.0
E5M2_MAX = 57344.0


def _kind(kv_cache_dtype):
    if kv_cache_dtype in ("fp8", "fp8_e4m3", "e4m3"):
        return "e4m3"
    if kv_cache_dtype in ("fp8_e5m2", "e5m2"):
        return "e5m2"
    raise ValueError(f"Unsupported data type of kv cache: {kv_cache_dtype}")


def fp8_decode_table(kind):
    """256-entry float32 table built from the bit layout alone (independent of
    torch): e4m3fn = 1-4-3 bias 7, no inf, NaN = S.1111.111;
    e5m2 = 1-5-2 bias 15, IEEE inf/NaN."""
    kind = _kind(kind)
    t = np.zeros(256, dtype=np.float32)
    for b in range(256):
        s = -1.0 if b & 0x80 else 1.0
        if kind == "e4m3":
            e, m = (b >> 3) & 0xF, b & 0x7
            if e == 0xF and m == 0x7:
                v = np.nan
            elif e == 0:
                v = m * 2.0 ** (-9)
            else:
                v = (1 + m / 8.0) * 2.0 ** (e - 7)
        else:
            e, m = (b >> 2) & 0x1F, b & 0x3
            if e == 0x1F:
                v = np.inf if m == 0 else np.nan
            elif e == 0:
                v = m * 2.0 ** (-16)
            else:
                v = (1 + m / 4.0) * 2.0 ** (e - 15)
        t[b] = s * v
    return t


def fp8_decode(bits, kind):
    """uint8 array -> float32."""
    return fp8_decode_table(kind)[np.asarray(bits).astype(np.uint8)]


def fp8_encode(x, kind):
    """float array -> uint8 bits, round-to-nearest-even, saturating to the
    largest finite value (__NV_SATFINITE / the clamp in common.cu:56)."""
    kind = _kind(kind)
    x = torch.as_tensor(np.asarray(x, dtype=np.float32))
    if kind == "e4m3":
        q = x.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn)
    else:
        q = x.clamp(-E5M2_MAX, E5M2_MAX).to(torch.float8_e5m2)
    return q.view(torch.uint8).numpy()


def kv_quant(x, scale, kind):
    """Cache write: fp8(x / scale)   (cache_kernels.cu:198-201)."""
    x = np.asarray(x, dtype=np.float32)
    return fp8_encode(x / np.float32(scale), kind)


def kv_dequant(bits, scale, kind):
    """Cache read: float(fp8) * scale   (nvidia/quant_utils.cuh:292-313)."""
    return fp8_decode(bits, kind) * np.float32(scale)


def convert_fp8(src, scale, kind, to_fp8):
    """ops.convert_fp8 (cache_kernels.cu:334-409)."""
    return kv_quant(src, scale, kind) if to_fp8 else kv_dequant(src, scale, kind)


# -- activation quantisation --------------------------------------------------
def static_scaled_fp8_quant(x, scale):
    """common.cu:187-199: out = fp8(clamp(x * (1/scale)))."""
    x = np.asarray(x).astype(np.float32)
    inv = np.float32(1.0) / np.float32(scale)
    return fp8_encode(x * inv, "e4m3")


def dynamic_scaled_fp8_quant(x):
    """common.cu:72-140 + 187-199: scale = absmax/448, then static."""
    x = np.asarray(x).astype(np.float32)
    scale = np.float32(np.abs(x).max()) / np.float32(E4M3_MAX)
    return static_scaled_fp8_quant(x, scale), np.array([scale], np.float32)


def dynamic_per_token_scaled_fp8_quant(x, scale_ub=None):
    """common.cu:201-256: per row scale = max(min(absmax, ub)/448,
    1/(448*512)); out = fp8(clamp(x / scale))."""
    x = np.asarray(x).astype(np.float32)
    amax = np.abs(x).max(axis=-1).astype(np.float32)
    if scale_ub is not None:
        amax = np.minimum(amax, np.float32(scale_ub))
    min_sf = np.float32(1.0) / (np.float32(E4M3_MAX) * np.float32(512.0))
    scales = np.maximum(amax / np.float32(E4M3_MAX), min_sf).astype(np.float32)
    q = fp8_encode(x / scales[:, None], "e4m3")
    return q, scales[:, None]


def scaled_fp8_quant(x, scale=None, num_token_padding=None, scale_ub=None,
                     use_per_token_if_dynamic=False):
    """_custom_ops.py:632-685 (padding rows are left unspecified there; the
    oracle zero-fills them and tests compare only the first M rows)."""
    x = np.asarray(x)
    m = x.shape[0]
    if scale is None:
        if use_per_token_if_dynamic:
            q, s = dynamic_per_token_scaled_fp8_quant(x, scale_ub)
        else:
            q, s = dynamic_scaled_fp8_quant(x)
    else:
        q = static_scaled_fp8_quant(x, np.asarray(scale).reshape(-1)[0])
        s = np.asarray(scale, dtype=np.float32)
    if num_token_padding and num_token_padding > m:
        pad = np.zeros((num_token_padding - m, x.shape[1]), np.uint8)
        q = np.concatenate([q, pad], 0)
    return q, s


def scaled_mm(a_bits, b_bits, scale_a, scale_b, bias=None, kind="e4m3"):
    """test_cutlass.py:36-47: (scale_a * (scale_b * (A @ B))) + bias in fp32
    (fp64 accumulate here).  a_bits [M,K], b_bits [K,N] uint8 fp8 bit patterns;
    scale_a scalar or [M,1]; scale_b scalar or [1,N] / [N]."""
    a = fp8_decode(a_bits, kind).astype(np.float64)
    b = fp8_decode(b_bits, kind).astype(np.float64)
    sa = np.asarray(scale_a, dtype=np.float64).reshape(-1, 1)
    sb = np.asarray(scale_b, dtype=np.float64).reshape(1, -1)
    out = sa * (sb * (a @ b))
    if bias is not None:
        out = out + np.asarray(bias, dtype=np.float64).reshape(1, -1)
    return out


def fp8_w8a16_gemm(a, w_bits, w_scale, kind="e4m3"):
    """fp8_marlin_gemm role (fp8/fp8_marlin.cu:1212): C = A @ (fp8->hp(W) * s_n).
    a [M,K] float, w_bits [K,N] uint8, w_scale scalar or [N]."""
    w = fp8_decode(w_bits, kind).astype(np.float64)
    s = np.asarray(w_scale, dtype=np.float64).reshape(1, -1)
    return np.asarray(a).astype(np.float64) @ (w * s)
