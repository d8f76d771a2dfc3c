import numpy as np, torch, sys
sys.path.insert(0, '.')
from aphrodite_engine_amd import _custom_ops as ops
from aphrodite_engine_amd.model import _rope_cache
DEV='cuda'
rng = np.random.default_rng(33)
T_, Hq, Hkv, hd, BS, NB = 9, 8, 2, 128, 16, 6
ntot = (Hq + 2 * Hkv) * hd
dtype=torch.float16
cs = _rope_cache(hd, 512, 10000.0, dtype, DEV)
pos = torch.from_numpy(rng.integers(0, 512, size=T_).astype(np.int64)).to(DEV)
slots = torch.from_numpy(rng.permutation(NB * BS)[:T_].astype(np.int64)).to(DEV)
slabs = torch.from_numpy(rng.standard_normal((2, T_, ntot)).astype(np.float32)).to(DEV)
qkv = (slabs[0] + slabs[1]).to(dtype)
ref = qkv.clone()
q, k, v = ref.split([Hq * hd, Hkv * hd, Hkv * hd], dim=-1)
ops.rotary_embedding(pos, q, k, hd, cs, True)
kc2 = torch.zeros(NB, Hkv, hd // 8, BS, 8, dtype=dtype, device=DEV); vc2 = torch.zeros(NB, Hkv, hd, BS, dtype=dtype, device=DEV)
q2 = ops.rope_cache(None, slabs, pos, cs, True, kc2, vc2, slots, Hq, Hkv, hd, "auto", 1.0, 1.0)
d = (q2.float() - q.float()).abs()
print("mismatch", (q2 != q).sum().item(), "of", q.numel(), "max", d.max().item())
idx = (q2 != q).nonzero()[:5]
print(idx)
q3 = ops.rope_cache(qkv.clone(), None, pos, cs, True, kc2, vc2, slots, Hq, Hkv, hd, "auto", 1.0, 1.0)
print("rowmajor-in mismatch", (q3 != q).sum().item())
tok, j = 4, 696
h, r = j // 128, j % 128
x = qkv[tok, h*128 + r].float().item(); y = qkv[tok, h*128 + r + 64].float().item()
p = pos[tok].item()
c = cs[p, r].float().item(); s = cs[p, 64 + r].float().item()
import numpy as np
f = np.float32
xo = np.float32(np.float64(f(x))*np.float64(f(c)) - np.float64(f(f(y)*f(s))))
print("x,y,c,s", x, y, c, s)
print("fma-style", xo, np.float16(xo), " two-round", f(f(x)*f(c)) - f(f(y)*f(s)), "unfused", q[tok, j].item(), "fused", q2[tok, j].item())
print("exact", np.float64(x)*c - np.float64(y)*s)
