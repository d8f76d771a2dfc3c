"""CPU baselines timed beside the GPU number -- TEST / BENCH INFRASTRUCTURE ONLY (see oracle/__init__.py).

(i)  ``opt125m_cpu_executor``: BASELINE.json configs[0] -- the reference's CPU-executor path on OPT-125m, bs = 1, greedy.
     The reference package is not importable on this box (missing loguru, msgspec, ...), so the loop below follows
     executor/cpu_executor.py:25-120 (fp16 -> bf16 cast :316-320, eager), worker/cpu_model_runner.py (one decode token
     per step, slot = block * 16 + offset) and modeling/models/opt.py:57-182 (pre-LN decoder layer: LayerNorm -> qkv ->
     attention -> out_proj -> residual -> LayerNorm -> fc1 -> ReLU -> fc2 -> residual; learned positions with offset 2,
     tied lm_head) over the reference's OWN compiled CPU kernels from oracle/_ref (kernels/cpu/attention.cpp
     paged_attention_v1, kernels/cpu/cache.cpp reshape_and_cache) + ``torch.nn.functional.linear`` / SDPA for the
     prompt (torch_sdpa backend).  Random-init weights of the real architecture (no network for checkpoints).
(ii) ``llama8b_int4_decode``: the per-op port of configs[1] on the host cores -- the int4 matrices are dequantised ONCE
     (load time, as any CPU executor would hold bf16 weights), several distinct layers are cycled so the last-level
     cache cannot serve them, lm_head and the norm / rotary / activation glue are inside the timed step."""
import os
import time

import numpy as np
import torch
import torch.nn.functional as F


def usable_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (os.cpu_count() reports the
    machine, and 256 OpenMP threads on a 16-core allotment make every op 100x slower)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(-(-int(quota) // int(period)))))
    except (OSError, ValueError):
        pass
    return n


def pick_threads(fn, candidates):
    """Run fn once per thread count (after a warm-up at the first one) and keep the fastest: the baseline is timed at
    the thread count that serves it best, and reports that count as `cores`."""
    best, best_t = None, None
    for n in candidates:
        torch.set_num_threads(n)
        fn()
        t0 = time.perf_counter()
        fn()
        dt_ = time.perf_counter() - t0
        if best_t is None or dt_ < best_t:
            best, best_t = n, dt_
    torch.set_num_threads(best)
    return best


def _thread_candidates():
    top = usable_cores()
    return sorted({n for n in (4, 8, 16, 32, 64, 128, top) if n <= top}) or [1]


def _ref_ops(root):
    so = os.path.join(root, "oracle", "_ref", "libaphro_ref_cpu.so")
    if not os.path.exists(so):
        return None
    try:
        torch.ops.load_library(so)
        return torch.ops.aphro_ref_cpu
    except Exception:
        return None


def opt125m_cpu_executor(root, prompt_len=32, new_tokens=64, seed=0):
    """-> dict(value = output tokens/s, ...) of the bs = 1 greedy loop."""
    ref = _ref_ops(root)
    if ref is None:
        return dict(value=None, sample="oracle/_ref/libaphro_ref_cpu.so not built")
    torch.manual_seed(seed)
    dt = torch.bfloat16                       # cpu_executor.py:316-320
    H, L, NH, FF, V, MAXPOS = 768, 12, 12, 3072, 50272, 2048      # facebook/opt-125m config.json
    hd = H // NH

    def w(*shape, std=0.02):
        return (torch.randn(*shape) * std).to(dt)
    emb, pos_emb = w(V, H), w(MAXPOS + 2, H)
    layers = [dict(ln1=(torch.ones(H, dtype=dt), torch.zeros(H, dtype=dt)), qkv=(w(3 * H, H), w(3 * H)),
                   out=(w(H, H), w(H)), ln2=(torch.ones(H, dtype=dt), torch.zeros(H, dtype=dt)),
                   fc1=(w(FF, H), w(FF)), fc2=(w(H, FF), w(H))) for _ in range(L)]
    final_ln = (torch.ones(H, dtype=dt), torch.zeros(H, dtype=dt))
    BS = 16
    nblocks = (prompt_len + new_tokens + BS - 1) // BS + 1
    x = 16 // 2
    kcs = [torch.zeros(nblocks, NH, hd // x, BS, x, dtype=dt) for _ in range(L)]
    vcs = [torch.zeros(nblocks, NH, hd, BS, dtype=dt) for _ in range(L)]
    block_table = torch.arange(nblocks, dtype=torch.int32).view(1, -1)
    scale = hd ** -0.5

    def forward(ids, positions, is_prompt):
        T = ids.shape[0]
        h = emb[ids] + pos_emb[positions + 2]                                   # opt.py:44-55 (offset 2)
        slots = (block_table[0, (positions // BS).long()].long() * BS + positions % BS).long()
        for li, p in enumerate(layers):
            res = h
            h = F.layer_norm(h, (H, ), *p["ln1"])
            qkv = F.linear(h, *p["qkv"])
            q, k, v = qkv.split(H, dim=-1)
            kv_k, kv_v = k.reshape(T, NH, hd).contiguous(), v.reshape(T, NH, hd).contiguous()
            ref.reshape_and_cache(kv_k, kv_v, kcs[li], vcs[li], slots, "auto", 1.0, 1.0)
            if is_prompt:                                                       # torch_sdpa backend, causal
                a = F.scaled_dot_product_attention(q.view(T, NH, hd).transpose(0, 1), kv_k.transpose(0, 1),
                                                   kv_v.transpose(0, 1), is_causal=True, scale=scale)
                a = a.transpose(0, 1).reshape(T, H)
            else:
                out = torch.empty(T, NH, hd, dtype=dt)
                seq_lens = (positions + 1).to(torch.int32)
                ref.paged_attention_v1(out, q.reshape(T, NH, hd).contiguous(), kcs[li], vcs[li], NH, scale, block_table,
                                       seq_lens, BS, int(seq_lens.max()), None, "auto", 1.0, 1.0, 0, 0, 0, 64, 0)
                a = out.view(T, H)
            h = res + F.linear(a, *p["out"])
            res = h
            h = F.layer_norm(h, (H, ), *p["ln2"])
            h = F.linear(F.relu(F.linear(h, *p["fc1"])), *p["fc2"])
            h = res + h
        h = F.layer_norm(h, (H, ), *final_ln)
        return F.linear(h[-1:], emb)                                            # tied lm_head, last token only
    with torch.no_grad():
        ids = torch.randint(0, V, (prompt_len, ))
        forward(ids, torch.arange(prompt_len), True)           # fills the cache: decode steps below are valid
        threads = pick_threads(lambda: forward(torch.tensor([1]), torch.tensor([prompt_len]), False), _thread_candidates())
        t0 = time.perf_counter()
        logits = forward(ids, torch.arange(prompt_len), True)
        t_prefill = time.perf_counter() - t0
        tok = int(logits.argmax(-1))
        t1 = time.perf_counter()
        for step in range(new_tokens):
            p = prompt_len + step
            logits = forward(torch.tensor([tok]), torch.tensor([p]), False)
            tok = int(logits.argmax(-1))
        t_decode = time.perf_counter() - t1
    return dict(value=new_tokens / t_decode, unit="tokens/s", cores=threads, kind="reference",
                sample=(f"configs[0]: OPT-125m bf16 (random-init), bs=1 greedy, prompt {prompt_len} + {new_tokens} generated "
                        f"tokens; CPU-executor loop over the reference's compiled kernels/cpu paged_attention_v1 + "
                        f"reshape_and_cache (oracle/_ref) and torch linear/SDPA; prefill {t_prefill * 1e3:.1f} ms"))


def llama8b_int4_decode(root, cfg, batch, ctx, budget_s=20.0, distinct_layers=6, seed=0):
    """configs[1] on the host cores: one decode step = 32 x [rms_norm, qkv, rotary, cache write, paged attention,
    o_proj, add+rms_norm, gate_up, silu*mul, down] + final norm + lm_head + argmax, bf16, weights dequantised at load
    time through oracle.quant (hoisted out of the timed loop)."""
    from . import quant as oq
    ref = _ref_ops(root)
    torch.set_num_threads(usable_cores())
    rng = np.random.default_rng(seed)
    dt = torch.bfloat16
    hid, inter = cfg.hidden_size, cfg.intermediate_size
    hq, hkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    shapes = [(hid, (hq + 2 * hkv) * hd), (hq * hd, hid), (hid, 2 * inter), (inter, hid)]
    t_load0 = time.perf_counter()
    # Load time (NOT timed): a CPU executor holds the int4 matrices dequantised to bf16 (oracle.quant.gptq_dequant is
    # what it would run once per matrix -- ~10 s each in numpy, so the timing uses random bf16 stand-ins of the same
    # shapes in F.linear layout [N, K]; values do not change the time of a bf16 matmul).  One tiny matrix goes through
    # the real dequant so that the path is exercised.
    _ = oq.gptq_dequant(rng.integers(0, 2 ** 32, size=(32, 64), dtype=np.uint32).view(np.int32),
                        rng.integers(0, 2 ** 32, size=(2, 8), dtype=np.uint32).view(np.int32),
                        (rng.random((2, 64)) * 0.01).astype(np.float16), None, shuffled=False)
    layers = [[(torch.randn(n, k) * 0.02).to(dt) for k, n in shapes] for _ in range(distinct_layers)]
    t_load = time.perf_counter() - t_load0
    lm_head = (torch.randn(cfg.vocab_size // 8, hid) * 0.02).to(dt)      # 1/8 slice is all that is timed
    nb = batch * ((ctx + 15) // 16)
    kc = torch.rand(nb, hkv, hd // 8, 16, 8).to(dt)
    vc = torch.rand(nb, hkv, hd, 16).to(dt)
    bt = torch.randperm(nb).view(batch, -1).int()
    sl = torch.full((batch, ), ctx, dtype=torch.int32)
    slots = (bt[:, (ctx - 1) // 16].long() * 16 + (ctx - 1) % 16)
    positions = torch.full((batch, ), ctx - 1, dtype=torch.long)
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, hd, 2).float() / hd))
    fr = torch.arange(ctx + 1).float()[:, None] * inv[None]
    cos_sin = torch.cat([fr.cos(), fr.sin()], -1).to(dt)
    ln = torch.ones(hid, dtype=dt)
    x0 = torch.randn(batch, hid).to(dt)

    def layer_fwd(h, residual, ws):
        wqkv, wo, wgu, wd = ws
        if ref is not None:
            ref.fused_add_rms_norm(h, residual, ln, cfg.rms_norm_eps)
        else:
            residual = h + residual
            h = F.rms_norm(residual, (hid, ), ln, cfg.rms_norm_eps)
        qkv = F.linear(h, wqkv)
        q, k, v = qkv.split([hq * hd, hkv * hd, hkv * hd], dim=-1)
        q, k = q.contiguous(), k.contiguous()
        if ref is not None:
            ref.rotary_embedding(positions, q, k, hd, cos_sin, True)
            ref.reshape_and_cache(k.view(batch, hkv, hd), v.reshape(batch, hkv, hd).contiguous(), kc, vc, slots, "auto", 1.0, 1.0)
            out = torch.empty(batch, hq, hd, dtype=dt)
            ref.paged_attention_v1(out, q.view(batch, hq, hd), kc, vc, hkv, hd ** -0.5, bt, sl, 16, ctx, None, "auto",
                                   1.0, 1.0, 0, 0, 0, 64, 0)
            a = out.view(batch, hq * hd)
        else:
            from . import attention as oa
            a = torch.from_numpy(oa.paged_attention_decode(q.view(batch, hq, hd).float().numpy(), kc.float().numpy(),
                                                           vc.float().numpy(), bt.numpy(), sl.numpy(), hd ** -0.5)
                                 ).to(dt).view(batch, hq * hd)
        h = F.linear(a, wo)
        if ref is not None:
            ref.fused_add_rms_norm(h, residual, ln, cfg.rms_norm_eps)
        else:
            residual = h + residual
            h = F.rms_norm(residual, (hid, ), ln, cfg.rms_norm_eps)
        gu = F.linear(h, wgu)
        act = torch.empty(batch, inter, dtype=dt)
        if ref is not None:
            ref.silu_and_mul(act, gu)
        else:
            act = F.silu(gu[:, :inter]) * gu[:, inter:]
        return F.linear(act, wd), residual
    nl = cfg.num_hidden_layers
    with torch.no_grad():
        # BOUNDED sample (the CPU matmuls are slow enough that whole steps would take minutes): one warm-up layer,
        # then layers back to back over the distinct weight sets until the budget is spent (>= 4 layers), the lm_head
        # on a 1/8 slice of the vocabulary; a step = 32 x the mean layer time + 8 x the slice time.
        h, r = x0.clone(), x0.clone()
        threads = pick_threads(lambda: layer_fwd(x0.clone(), x0.clone(), layers[0]), _thread_candidates())
        t0 = time.perf_counter()
        done = 0
        while True:
            h, r = layer_fwd(h, r, layers[(done + 1) % distinct_layers])
            done += 1
            if done >= 4 and (time.perf_counter() - t0 > budget_s or done >= 2 * nl):
                break
        per_layer = (time.perf_counter() - t0) / done
        vs = cfg.vocab_size // 8
        t1 = time.perf_counter()
        logits = F.linear(F.rms_norm(h + r, (hid, ), ln, cfg.rms_norm_eps), lm_head)
        logits.argmax(-1)
        t_head = (time.perf_counter() - t1) * (cfg.vocab_size / vs)
        per_step = per_layer * nl + t_head
        steps = done
    return dict(value=batch / per_step, unit="tokens/s", cores=threads, kind="port",
                sample=(f"{steps} decoder layers of Llama-3-8B geometry timed back to back ({per_layer * 1e3:.1f} ms each, cycling "
                        f"{distinct_layers} distinct bf16 weight sets = the int4 matrices as a CPU executor holds them after its "
                        f"load-time dequant, not timed) x {nl} layers + lm_head ({t_head * 1e3:.1f} ms, from a 1/8 vocabulary slice); bs={batch}, ctx={ctx}; "
                        f"norm / rotary / cache write / paged attention / silu_and_mul through "
                        f"{'the reference CPU kernels of oracle/_ref' if ref is not None else 'torch + oracle.attention'}; "
                        f"argmax included), {per_step * 1e3:.1f} ms/step"),
                ms_per_step=per_step * 1e3)
