"""Lab: phase stamps of flash_attn_varlen_v3 (heaviest workgroup, head 0) at T = 8192 causal."""
import ctypes, os
os.environ["APHRO_FA_DEBUG"] = os.environ.get("APHRO_FA_DEBUG", "1")
import torch
from aphrodite_engine_amd import _custom_ops as ops, _lib
T, Hq, Hkv, D = 8192, 32, 8, 128
qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device="cuda", dtype=torch.float16) * 0.5
q = qkv[:, :Hq * D].view(T, Hq, D); k = qkv[:, Hq * D:(Hq + Hkv) * D].view(T, Hkv, D); v = qkv[:, (Hq + Hkv) * D:].view(T, Hkv, D)
cu = torch.tensor([0, T], dtype=torch.int32, device="cuda")
for _ in range(3):
    ops.flash_attn_varlen(q, k, v, cu, T, D ** -0.5, causal=True)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 512)()
_lib.lib().aphro_fa_debug_dump(buf, 512)
for w in range(8):
    t = [buf[w * 64 + i] for i in range(12)]
    print(f"wave {w}: per tile (it 32..64) {(t[2] - t[1]) / 32:.0f} cycles | tile 40: top->waited {t[9] - t[8]}  barrier {t[10] - t[9]}  DMA issue {t[11] - t[10]}  "
          f"first half {t[4] - t[3]}  second half {t[6] - t[4]}  | total loop {t[7] - t[0]}")
