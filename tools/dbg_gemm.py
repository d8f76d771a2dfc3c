import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from aphrodite_engine_amd import _custom_ops as ops
from oracle import quant as oq
import test_ops_gpu as T
for (M,K,N,G) in [(1,512,256,128),(32,512,256,128),(16,1024,64,128)]:
    rng = np.random.default_rng(M * 131 + K + N)
    qweight, qzeros, s, _ = T.make_gptq(rng, K, N, G)
    a = rng.standard_normal((M, K)).astype(np.float32)
    a_t = T.t(a, torch.float16); s_t = T.t(s, torch.float16)
    shuf = T.t(oq.gptq_shuffle(qweight))
    got = ops.gptq_gemm(a_t, shuf, T.t(qzeros), s_t, torch.empty(0, dtype=torch.int32, device="cuda"), True, 4).float().cpu().numpy()
    ref = oq.gptq_gemm(a_t.float().cpu().numpy(), shuf.cpu().numpy(), qzeros, s_t.float().cpu().numpy(), None, True)
    print(M,K,N,G, "got", got[0,:6], "ref", ref[0,:6], "ratio", (got/ref)[0,:6])
    # zero-only / scale-only checks
    wz = oq.gptq_dequant(shuf.cpu().numpy(), qzeros, s_t.float().cpu().numpy(), None, True) if hasattr(oq,'gptq_dequant') else None
