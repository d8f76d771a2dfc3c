"""Lab: loop duration of flash_attn_varlen_v3's heaviest workgroup in shader cycles (s_memtime) and in 100 MHz
wall-clock ticks (s_memrealtime): what clock does the chip hold under this kernel?"""
import ctypes, os
os.environ["APHRO_FA_DEBUG"] = "1"
import torch
from aphrodite_engine_amd import _custom_ops as ops, _lib
T, Hq, Hkv, D = 8192, 32, 8, 128
qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device="cuda", dtype=torch.float16) * 0.5
q = qkv[:, :Hq * D].view(T, Hq, D); k = qkv[:, Hq * D:(Hq + Hkv) * D].view(T, Hkv, D); v = qkv[:, (Hq + Hkv) * D:].view(T, Hkv, D)
cu = torch.tensor([0, T], dtype=torch.int32, device="cuda")
for _ in range(20):
    ops.flash_attn_varlen(q, k, v, cu, T, D ** -0.5, causal=True)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 512)()
_lib.lib().aphro_fa_debug_dump(buf, 512)
for w in (0, 7):
    t = [buf[w * 64 + i] for i in range(14)]
    cyc, ticks = t[7] - t[0], t[13] - t[12]
    print(f"wave {w}: loop {cyc} s_memtime ticks, {ticks} wall ticks (100 MHz) = {ticks / 100:.1f} us -> s_memtime runs at {cyc / ticks * 100:.0f} MHz; "
          f"per tile (it 32..64) {(t[2] - t[1]) / 32:.0f} ticks")
