#!/usr/bin/env python3
"""bench.py -- decode throughput of the MI355X quantized-inference hot path.

Metric (BASELINE.json): output tokens/s + HBM-roofline fraction, Llama-3-8B
GPTQ-int4 g128 (configs[1]: TP=1, bs=32) on 1/2/4/8 MI355X.

A "step" is one full greedy decode step of the whole model over a batch of
synthetic sequences: embedding gather -> 32 x [rms_norm, qkv GPTQ GEMM, RoPE,
KV-cache write, paged-attention decode, o_proj GEMM, fused add+rms_norm,
gate_up GEMM, silu*mul, down GEMM] -> final norm -> fp16 lm_head -> argmax ->
advance (positions, seq_lens, slot mapping, next input ids).  Nothing is
skipped or cached inside the timed region; the sampled tokens feed the next
step.  The step is captured once into a HIP graph (as the reference captures
decode, worker/model_runner.py:1360-1507) and replayed K times.

Launch:  python bench.py --gpus N --steps K --warmup W
 N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
             --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
   default parallelism "dp": N independent replicas (the 8B model fits one GPU;
   SURVEY 8e) -> weak scaling, value = total tokens/s of all replicas.
   --parallelism tp: Megatron TP over RCCL (strong scaling), for reference.

Prints ONE JSON line on rank 0 (contract in the task statement), extended with
"roofline" (dominant kernel, live HIP-event timing) and "cpu_baseline".
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides: 8 TB/s; ~6.3 TB/s achievable)
# Every capture in this script: thread-local error mode.  With a process group up (N > 1) the RCCL watchdog thread polls the
# events of earlier collectives (the barriers); under the default "global" mode such a query from ANOTHER thread while this
# thread captures is "operation not permitted when stream is capturing" and takes the process down (seen once in
# tests/test_custom_ar_gpu.py on one box, never on two others: a race with the watchdog's polling interval).
GRAPH_KW = {"capture_error_mode": "thread_local"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--sim-tp", type=int, default=0,
                    help="one-GPU shard bench: time ONE rank of a TP group of this size (real per-GPU shard shapes, every "
                         "all-reduce replaced by a launch that holds the stream for --sim-ar-us); configs[3]: --model "
                         "llama3-70b --quant awq --batch 64 --sim-tp 8, configs[4]: --model mixtral-8x7b --sim-tp 4")
    ap.add_argument("--sim-ar-us", type=float, default=0.0,
                    help="latency of the stubbed all-reduce (default: the one-GPU lower bound of the peer-access kernel for "
                         "the message size, profiles/r1_custom_ar_one_gpu.txt)")
    ap.add_argument("--sim-ar-stub", action="store_true",
                    help="with --sim-tp: every all-reduce is a launch holding the stream for --sim-ar-us (the round-3/4 stub) "
                         "instead of the real peer-access kernels on a loopback communicator")
    ap.add_argument("--model", default="llama3-8b", choices=["llama3-8b", "llama3-70b", "mixtral-8x7b"],
                    help="mixtral-8x7b: BASELINE configs[4] (int4 experts through the grouped GEMM); gptq / awq only")
    ap.add_argument("--quant", default="gptq", choices=["gptq", "fp8", "awq", "fp8ct"],
                    help="fp8: per-tensor W8A8 (Fp8Config); fp8ct: compressed-tensors W8A8, per-token x per-channel")
    ap.add_argument("--act-scheme", default="dynamic", choices=["dynamic", "static"],
                    help="fp8ct: per-token dynamic activation scales (llm-compressor FP8_DYNAMIC, configs[2]) or the "
                         "checkpoint's static per-tensor input_scale; fp8 (Fp8Config): dynamic per-tensor or static")
    ap.add_argument("--kv-cache-dtype", default="auto", choices=["auto", "fp8", "fp8_e5m2"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--ctx", type=int, default=1024, help="context length at the first timed step")
    ap.add_argument("--parallelism", default="dp", choices=["dp", "tp"])
    ap.add_argument("--sampling", default="greedy", choices=["greedy", "random"],
                    help="random: temperature 0.8, top-k 50, top-p 0.95 through the fused sampling kernel "
                         "(in-kernel noise, per-row seeds advanced on the device) instead of argmax")
    ap.add_argument("--overlap", action="store_true",
                    help="tp only: all-reduces on a side stream + a prefetch of the next projection's weights through the "
                         "Infinity Cache on the compute stream.  OFF by default since round 4: the same side-stream prefetch "
                         "made the one-GPU step 38-61 %% slower (profiles/r3_prefetch_lab.txt) and no multi-GPU box has shown "
                         "the all-reduce side of the trade; the TP section of a multi-GPU run still reports both arms")
    ap.add_argument("--no-overlap", action="store_true", help="(accepted for older command lines: overlap is off by default)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--two-copies", action="store_true",
                    help="keep the [K/8, N] originals of the int4 matrices beside the strip-major decode copies (the layout of rounds 3-5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefill-info", action="store_true", help="skip the MFMA-bound prefill kernels' info section")
    ap.add_argument("--no-prefill-e2e", action="store_true",
                    help="skip the end-to-end 8192-token prefill legs (int4 and FP8, ours and the library arm)")
    ap.add_argument("--no-ops-path", action="store_true", help="skip the second measurement on the op-by-op (drop-in) path")
    ap.add_argument("--ragged", action="store_true",
                    help="ragged context lengths randint(1, ctx) (seed 0; SURVEY 8d, the reference's "
                         "tests/benchmarks/kernels/paged_attention.py) instead of one length for every sequence")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the extra legs of the default run (FP8 configs, ragged batch, TP shards: each a child "
                         "process of this script, summarised under `legs` in the line)")
    ap.add_argument("--no-tp-section", action="store_true",
                    help="N > 1, replicas: skip the tensor-parallel section (TP = N timing, overlap A/B, all-reduce latencies)")
    ap.add_argument("--tp-child", action="store_true", help="(internal) the tensor-parallel section of an N > 1 run, in its own processes")
    ap.add_argument("--layers", type=int, default=0, help="debug only: fewer layers (marks the line invalid)")
    return ap.parse_args()


def build(args, device):
    from aphrodite_engine_amd import model as M
    from aphrodite_engine_amd.quantization.awq import AWQConfig
    from aphrodite_engine_amd.quantization.fp8 import Fp8Config
    from aphrodite_engine_amd.quantization.gptq import GPTQConfig
    import dataclasses
    cfg = {"llama3-8b": M.LLAMA3_8B, "llama3-70b": M.LLAMA3_70B, "mixtral-8x7b": M.MIXTRAL_8X7B}[args.model]
    if args.model == "mixtral-8x7b" and args.quant not in ("gptq", "awq", "fp8"):
        raise SystemExit("--model mixtral-8x7b runs int4 (--quant gptq / awq) or FP8 (--quant fp8) experts")
    if args.layers:
        cfg = dataclasses.replace(cfg, num_hidden_layers=args.layers)
    # the rotary table must cover every position the loop reaches (--ctx 8192 ends past Llama-3-8B's 8192 window:
    # the long-context variants of the same geometry ship a longer table)
    last_pos = args.ctx + args.steps + args.warmup + 8
    if last_pos > cfg.max_position_embeddings:
        cfg = dataclasses.replace(cfg, max_position_embeddings=(last_pos + 1023) // 1024 * 1024)
    if args.quant == "gptq":
        qc = GPTQConfig(4, 128, False)
    elif args.quant == "awq":
        qc = AWQConfig(4, 128, True, prepack=True)
    elif args.quant == "fp8ct":
        from aphrodite_engine_amd.quantization.fp8 import CompressedTensorsW8A8Fp8Config
        qc = CompressedTensorsW8A8Fp8Config(strategy="channel", is_static_input_scheme=args.act_scheme == "static")
    else:
        qc = Fp8Config(is_checkpoint_fp8_serialized=True, activation_scheme=args.act_scheme)
    dtype = torch.float16
    torch.cuda.synchronize()
    mem0 = torch.cuda.memory_allocated(device)
    model = M.LlamaForCausalLM(cfg, qc, dtype, args.kv_cache_dtype)
    model.init_synthetic(device, seed=0)
    if os.environ.get("APHRO_NO_FUSED_ROPE"):
        for layer in model.layers:
            layer.fuse_rope_attention = False
    if not os.environ.get("APHRO_NO_FUSED_SILU"):
        # load-time relayout: SiluAndMul + pack run in the gate_up GEMM epilogue
        for layer in model.layers:
            layer.enable_fused_silu(args.batch, keep_original=False)
        if not getattr(args, "two_copies", False):
            # ONE resident copy of each int4 matrix: the strip-major order the decode kernels stream (the prompt-sized
            # kernels address it in place) -- TP 1 dense layers; a no-op elsewhere (model.DecoderLayer.enable_one_copy).
            # (after ALL layers have their decode copies: released layer by layer, the allocator hands layer i's [K/8, N]
            #  blocks to layer i + 1's strip-major copies)
            for layer in model.layers:
                layer.enable_one_copy()
    if args.quant.startswith("fp8") and args.batch <= 32:
        for layer in model.layers:
            layer.enable_fp8_strips(args.batch)     # load-time relayout for the resident W8A8 decode GEMM
    torch.cuda.synchronize()
    torch.cuda.empty_cache()        # (also: no big cached free blocks for the step's small hot buffers to scatter into,
                                    #  profiles/r6_one_copy.txt)
    # everything the model keeps in HBM (every layout copy of every matrix, embedding, lm_head, rotary table): the footprint
    # a serving engine pays, next to the one-copy algorithmic bytes of the roofline
    model.weight_bytes_resident = int(torch.cuda.memory_allocated(device) - mem0)
    return model, cfg, dtype


class DecodeLoop:
    """Persistent device state of a greedy decode loop (what the reference's
    CUDAGraphRunner keeps in its input buffers, model_runner.py:1682-1790)."""

    def __init__(self, model, cfg, dtype, args, device, total_steps):
        from aphrodite_engine_amd import model as M
        self.model, self.cfg, self.bs = model, cfg, 16
        self.block_size = 16
        max_len = args.ctx + total_steps + 1
        ctx = args.ctx
        if getattr(args, "ragged", False):
            import random
            rnd = random.Random(0)
            ctx = [rnd.randint(1, args.ctx) for _ in range(args.batch)]
        self.ctx_sum0 = float(sum(ctx)) if not isinstance(ctx, int) else float(ctx * args.batch)
        self.meta, self.positions, nblocks = M.make_decode_metadata(
            args.batch, ctx, self.block_size, device,
            blocks_per_seq=(max_len + self.block_size - 1) // self.block_size)
        self.meta.max_decode_seq_len = max_len      # capture-time maximum
        self.kv_caches = M.make_kv_caches(cfg, nblocks, self.block_size, dtype,
                                          args.kv_cache_dtype, device)
        g = torch.Generator(device=device).manual_seed(1)
        self.input_ids = torch.randint(0, cfg.vocab_size, (args.batch, ), generator=g,
                                       device=device)
        self.next_ids = torch.zeros_like(self.input_ids)
        self.sampling = args.sampling
        if self.sampling == "random":
            self.temperature = torch.full((args.batch, ), 0.8, device=device)
            self.top_k = torch.full((args.batch, ), 50, dtype=torch.int32, device=device)
            self.top_p = torch.full((args.batch, ), 0.95, device=device)
            self.seeds = torch.arange(args.batch, dtype=torch.int64, device=device) * 1000003 + 17

    def step(self):
        m = self.meta
        hidden = self.model(self.input_ids, self.positions, self.kv_caches, m)
        if self.sampling == "random":
            from aphrodite_engine_amd import _custom_ops as ops_
            logits = self.model.compute_logits(hidden)
            ops_.sample_top_k_top_p(logits, self.temperature, self.top_k, self.top_p, seeds=self.seeds,
                                    out=self.next_ids)
            self.seeds.add_(1)           # a fresh stream every step, also under graph replay
        elif self.model.use_fused_decode:
            # greedy: LM head GEMM + argmax in one launch where served (model.greedy_tokens, csrc/lm_head.hip)
            self.model.greedy_tokens(hidden, self.next_ids)
        else:
            # the op-by-op path keeps the reference's two steps (LogitsProcessor, then Sampler._greedy_sample)
            self.model.sample_greedy(self.model.compute_logits(hidden), self.next_ids)
        # advance: the generated token becomes the next input, context grows by one
        # (advance_step_flashattn, prepare_inputs/advance_step.cu: one launch instead of six)
        from aphrodite_engine_amd import _custom_ops as ops
        ops.advance_step_flashattn(self.input_ids.shape[0], self.input_ids.shape[0], self.block_size,
                                   self.input_ids, self.next_ids, self.positions, m.seq_lens_tensor,
                                   m.slot_mapping, m.block_tables)


def gemm_bytes(lin, M_rows):
    n = 0
    for name, p in lin.named_parameters():
        if name in ("g_idx", "input_scale"):
            continue
        n += p.numel() * p.element_size()
    return n + M_rows * lin.in_features * 2 + M_rows * lin.out_features * 2


def measure_kernel(fn, launches_per_call, iters=5):
    """Average duration of one launch: the launches are captured into a HIP graph
    (no host launch overhead between them, exactly like the timed decode step)
    and the replays are bracketed by HIP events on the replaying stream."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, **GRAPH_KW):
        fn()
    g.replay()
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record(stream)
    for _ in range(iters):
        g.replay()
    end.record(stream)
    end.synchronize()
    return start.elapsed_time(end) * 1e-3 / (iters * launches_per_call)


def roofline_section(model, loop, args):
    """Live per-kernel timing of the two HBM-heavy kernels, cycling over all
    layers so the 256 MiB Infinity Cache cannot serve the weights / KV."""
    from aphrodite_engine_amd import _custom_ops as ops
    bs = args.batch
    layers = list(model.layers)
    out = {}

    fast = all(l.fused_decode_ok(bs) for l in layers) and getattr(model, "use_fused_decode", False)
    dense_names = ("gate_up_proj", "down_proj", "qkv_proj", "o_proj")
    if layers[0].is_moe:
        # the sparse MLP as the decode step runs it: router GEMM, top-k softmax, align, gather-pack, the two
        # grouped int4 GEMMs, combine.  Real routing of random activations: at bs 32, top-2 of 8 every expert
        # is active, so the algorithmic bytes are all experts' weights (+ router, + activations).
        dense_names = ("qkv_proj", "o_proj")
        xin = torch.randn(bs, model.cfg.hidden_size, device="cuda", dtype=model.dtype)

        def run_moe():
            for layer in layers:
                layer.moe_block(xin)
        ex = getattr(layers[0].experts, "experts_packed", None)
        ex_tensors = (ex.w13 + ex.w2) if ex is not None else (layers[0].experts.w13_weight, layers[0].experts.w2_weight)
        for layer in layers:
            layer.experts.record_routing = True
        run_moe()
        torch.cuda.synchronize()
        active = sum(int(torch.unique(l.experts.last_topk_ids).numel()) for l in layers) / len(layers)
        for layer in layers:
            layer.experts.record_routing = False
        eb = sum(t.numel() * t.element_size() for t in ex_tensors) * active / layers[0].experts.num_experts \
            + layers[0].moe_gate.numel() * 2 + 2 * bs * model.cfg.hidden_size * 2
        t = measure_kernel(run_moe, len(layers))
        out["moe_block"] = dict(kernel=("sparse MLP block: router + topk_softmax + moe_align + gather_pack + "
                                        "wna16 grouped GEMM x2 + combine (seconds = whole block, not one launch)") if ex is not None else
                                ("sparse MLP block, FP8 experts: router + topk_softmax + moe_align + scaled_fp8_quant x2 + "
                                 "fp8_moe_gemm x2 + silu_and_mul + sum (seconds = whole block, not one launch)"),
                                bytes=eb, seconds=t, active_experts=active)
    for name in dense_names:
        lin0 = getattr(layers[0], name)
        xin = torch.randn(bs, lin0.in_features, device="cuda", dtype=model.dtype)
        if fast:
            # exactly what the decode step launches: packed activations in, fp32 slabs (or, for
            # gate_up with the fused epilogue, packed activations) out -- one kernel per call
            packed = ops.wna16_pack_a(xin)
            silu = name == "gate_up_proj" and layers[0].gate_up_interleaved is not None

            # 33..64 rows: the MLP weights run the one-pass kernel (model.py forward_decode_fused picks it the same way)
            n_out, groups = lin0.out_features, lin0.in_features // 128
            mid = 32 < bs <= 64 and not os.environ.get("APHRO_DECODE_NO_MID") and (
                (silu and ops.wna16_gemm_mid_ksplit(bs, n_out, lin0.in_features, groups) == 1) or
                (name == "down_proj" and n_out * lin0.in_features >= 2 ** 25
                 and ops.wna16_gemm_mid_ksplit(bs, n_out, lin0.in_features, groups) > 0))

            # 33..64 rows (round 4): the stream kernel on two 32-row halves where the layer has its strip-major copies
            # (a layer with ONE resident copy keeps its MLP weights on the one-pass kernel, which reads the strip-major words in
            #  place; only qkv / o go to the halves -- model.forward_decode_fused)
            halves = layers[0]._row_halves(bs)
            strip_m = 32 if getattr(lin0, "qweight_strip_major", False) else 0
            mlp_halves = halves and (not strip_m or os.environ.get("APHRO_DECODE_ROW_HALVES") == "1")
            use_halves = mlp_halves if name in ("gate_up_proj", "down_proj") else halves
            resident = silu and layers[0].gate_up_strip is not None and (bs <= 32 or use_halves) \
                and ops.wna16_resident_ksplit(bs, n_out, lin0.in_features, groups) == 1
            if resident or (use_halves and name in getattr(layers[0], "strip", {})):
                mid = False

            def run_lin(name=name, packed=packed, silu=silu, k=lin0.in_features, mid=mid, resident=resident, strip_m=strip_m):
                for layer in layers:
                    if silu:
                        qw, qz, sc, zo = layer.gate_up_interleaved
                        if resident:      # what forward_decode_fused launches at <= 32 rows
                            ops.wna16_gemm_resident(packed, bs, k, layer.gate_up_strip, qz, sc, zo, mode="silu", strip_layout=True)
                        elif mid:
                            ops.wna16_gemm_mid_silu_pack(packed, bs, k, qw, qz, sc, zo, strip_m=strip_m)
                        else:
                            ops.wna16_gemm_silu_pack(packed, bs, k, qw, qz, sc, zo)
                    elif mid:
                        qw, qz, sc, zo = getattr(layer, name).fast_params()
                        ops.wna16_gemm_mid_packed(packed, bs, k, qw, qz, sc, zo, partials=True, strip_m=strip_m)
                    else:
                        layer._gemm_slabs(name, packed, bs, k)      # resident kernel on the strip-major copy where the layer has one
            res_slabs = (not silu) and (not mid) and bs <= 64 and name in getattr(layers[0], "strip", {})
            stream = not os.environ.get("APHRO_WNA16_STREAM") == "0"     # (the four configs[1] plans dispatch to the stream kernel)
            kname = ("wna16_gemm_mid_kernel" if mid else
                     ("wna16_gemm_stream_kernel" + (" (two 32-row halves)" if bs > 32 else "")
                      if stream and (bs > 32 or ((bs, model.cfg.hidden_size) == (bs, 4096) and args.model == "llama3-8b"))
                      else "wna16_gemm_resident_kernel") + " (strip-major weights)" if (resident or res_slabs)
                     else "wna16_gemm_kernel") + (" (+SiluAndMul epilogue)" if silu else "")
        elif args.quant == "fp8ct" and getattr(model, "use_fused_decode", False):
            # the FP8 decode fast path hands every GEMM pre-quantised activations (the quantisation is fused into the norm /
            # SiluAndMul kernels): time the GEMM launch alone, in the form the step uses (fp32 slabs for qkv / o / down,
            # the scaled epilogue for gate_up)
            qx, sx = ops.scaled_fp8_quant(xin, None, use_per_token_if_dynamic=True)
            slab_form = name != "gate_up_proj"

            res8 = bs <= 32 and name in getattr(layers[0], "fp8_strip", {}) and (
                slab_form or ops.fp8_gemm_resident_ksplit(bs, lin0.out_features, lin0.in_features) == 1)

            # round 6 (dynamic scheme, <= 32 rows): gate_up runs SiluAndMul in its epilogue on the interleaved strip copy, and
            # o / down take the producer's 16-bit activations + absmax partials and quantise on load (the step's forms)
            l0_ = layers[0]
            il = getattr(l0_, "fp8_gate_up_il", None) is not None and bs <= 32
            dyn = getattr(lin0, "input_scale", None) is None
            np_aq = {"o_proj": l0_.num_kv_heads,
                     "down_proj": ops.fp8_gemm_resident_strips(bs, l0_.gate_up_proj.out_features, l0_.gate_up_proj.in_features)}.get(name, 0)
            aq = il and dyn and name in ("o_proj", "down_proj") and res8 and l0_.tp == 1 and \
                ops.fp8_gemm_resident_aq_supported(bs, lin0.out_features, lin0.in_features, np_aq)
            absmax = xin.float().abs().view(bs, np_aq, -1).amax(dim=2).contiguous() if aq else None
            if aq:          # the producers leave the activation pair-major (every A load of the GEMM lane-linear)
                xp = torch.zeros(ops.aq_pairs_numel(bs, xin.shape[1]), dtype=xin.dtype, device=xin.device)
                xp[ops.aq_pairs_index(bs, xin.shape[1], xin.device).flatten()] = xin.flatten()
                xin = xp

            def run_lin(name=name, qx=qx, sx=sx, slab_form=slab_form, res8=res8, il=il, aq=aq, absmax=absmax, xin=xin):
                for layer in layers:
                    lin = getattr(layer, name)
                    if aq:
                        ops.fp8_gemm_resident_aq(xin, absmax, layer.fp8_strip[name], a_pairs=True)
                    elif slab_form:
                        layer._fp8_slabs(name, qx)              # the resident kernel on the strip-major copy where the layer has one
                    elif il:
                        ops.fp8_gemm_resident_silu(qx, layer.fp8_gate_up_il, sx, lin.weight_scale, model.dtype, act_pairs=dyn)
                    elif res8:
                        ops.fp8_gemm_resident(qx, layer.fp8_strip[name], sx, lin.weight_scale, out_dtype=model.dtype)
                    else:
                        ops.cutlass_scaled_mm(qx, lin.weight, out_dtype=model.dtype, scale_a=sx, scale_b=lin.weight_scale)
            kname = ("fp8_gemm_resident_kernel (strip-major weights)" if (res8 or il) else "fp8_gemm_fast_kernel") + \
                (" (16-bit A quantised on load, fp32 slabs)" if aq else " (fp32 slabs)" if slab_form
                 else " (+SiluAndMul epilogue, absmax partials)" if il else " (scaled epilogue)")
        else:
            def run_lin(name=name, xin=xin):
                for layer in layers:
                    getattr(layer, name)(xin)
            kname = ("fp8_gemm_kernel" if args.quant.startswith("fp8") else "wna16_gemm_kernel") + " (+quant/pack/splitk_reduce)"
        t = measure_kernel(run_lin, len(layers))
        out[name] = dict(kernel=kname, shape=[bs, lin0.in_features, lin0.out_features],
                         bytes=gemm_bytes(lin0, bs), seconds=t)
    # decode attention over the real caches / metadata of the loop, in the form the step launches
    l0 = layers[0]
    meta = loop.meta
    from aphrodite_engine_amd.attention.paged_attn import PagedAttention
    caches = [PagedAttention.split_kv_cache(c, l0.num_kv_heads, l0.head_dim) for c in loop.kv_caches]
    esz = 1 if args.kv_cache_dtype != "auto" else 2
    tokens = int(meta.seq_lens_tensor.sum().item())
    ab = 2 * tokens * l0.num_kv_heads * l0.head_dim * esz + 2 * bs * l0.q_size * 2
    ntot = l0.q_size + 2 * l0.kv_size
    if fast and l0.head_dim == 128 and l0.fuse_rope_attention:
        slabs = torch.randn(2, bs, ntot, device="cuda", dtype=torch.float32) * 0.1
        cs_tok = model.cos_sin.index_select(0, loop.positions if hasattr(loop, "positions")
                                            else (meta.seq_lens_tensor.long() - 1))

        def run_attn():
            for kc, vc in caches:
                ops.paged_attention_rope_packed(slabs, None, cs_tok, meta.slot_mapping, kc, vc, l0.num_heads,
                                                l0.num_kv_heads, l0.attn.scale, meta.block_tables,
                                                meta.seq_lens_tensor, 16, meta.max_decode_seq_len, None,
                                                args.kv_cache_dtype, 1.0, 1.0)
        kname = "paged_attention_kernel<ROPE> (qkv slab reduce + rotary + cache write + attention)"
        ab += slabs.numel() * 4
    elif args.quant == "fp8ct" and getattr(model, "use_fused_decode", False) and l0.head_dim == 128 and l0.fuse_rope_attention:
        # the FP8 fused step launches the SCALED-slab form (raw fp32 slabs of the W8A8 qkv GEMM dequantised on the fly, rotary,
        # cache write; round 6: + absmax partials and the pair-major output for the o_proj GEMM that quantises on load).  Until
        # round 6 this entry timed the plain kernel (24.9 us at ctx 1024 where the step's form takes 33 us in the trace).
        slabs = torch.randn(2, bs, ntot, device="cuda", dtype=torch.float32) * 0.1
        cs_tok = model.cos_sin.index_select(0, loop.positions if hasattr(loop, "positions")
                                            else (meta.seq_lens_tensor.long() - 1))
        row_sc = torch.rand(bs, 1, device="cuda") + 0.5
        col_sc = l0._channel_scale(l0.qkv_proj)
        dyn_o = getattr(l0.o_proj, "input_scale", None) is None and getattr(l0, "fp8_gate_up_il", None) is not None and bs <= 32

        def run_attn():
            for kc, vc in caches:
                ops.paged_attention_rope_scaled(slabs, row_sc, col_sc, None, cs_tok, meta.slot_mapping, kc, vc, l0.num_heads,
                                                l0.num_kv_heads, l0.attn.scale, meta.block_tables, meta.seq_lens_tensor, 16,
                                                meta.max_decode_seq_len, None, args.kv_cache_dtype, 1.0, 1.0,
                                                want_absmax=dyn_o, out_pairs=dyn_o)
        kname = "paged_attention_kernel<ROPE = scaled slabs> (slab reduce + dequant + rotary + cache write + attention" + \
            (" + absmax partials, pair-major output)" if dyn_o else ")")
        ab += slabs.numel() * 4
    else:
        q = torch.randn(bs, ntot, device="cuda", dtype=model.dtype)[:, :l0.q_size]

        def run_attn():
            for kc, vc in caches:
                ops.paged_attention_packed(q.view(bs, l0.num_heads, l0.head_dim), kc, vc, l0.num_kv_heads,
                                           l0.attn.scale, meta.block_tables, meta.seq_lens_tensor, 16,
                                           meta.max_decode_seq_len, None, args.kv_cache_dtype, 1.0, 1.0)
        kname = "paged_attention_kernel (v1 form)"
    t2 = measure_kernel(run_attn, len(caches))
    out["paged_attention"] = dict(kernel=kname, bytes=ab, seconds=t2, launches_per_call=1)
    return out


def cpu_baseline(args, cfg):
    """CPU baselines on the box's host cores, same run (harness: oracle/cpu_executor.py):
    (ii) ``value``: configs[1] per-op port -- full decode steps of the same Llama-3-8B geometry in bf16 with the int4
         matrices dequantised ONCE at load time (hoisted: no CPU executor unpacks weights per step), several distinct
         layers cycled, lm_head and glue included, attention / norms / rotary / activation through the reference's own
         compiled CPU kernels (oracle/_ref);
    (i)  ``configs0``: the reference's CPU-executor loop on OPT-125m, bs = 1 greedy (BASELINE configs[0])."""
    from oracle import cpu_executor as ce
    port = ce.llama8b_int4_decode(ROOT, cfg, args.batch, args.ctx, budget_s=12.0, distinct_layers=4)
    out = dict(value=port["value"], unit="tokens/s", cores=port["cores"], kind="port", sample=port["sample"],
               ms_per_step=port.get("ms_per_step"))
    try:
        out["configs0"] = ce.opt125m_cpu_executor(ROOT, prompt_len=32, new_tokens=48)
    except Exception as e:
        out["configs0"] = {"value": None, "sample": f"failed: {e!r}"}
    return out


def prefill_section(cfg):
    """The MFMA-bound side of the same path (configs[2]'s prefill: seq 8192), timed live with HIP events: causal prefill
    attention, prefill with cached context, and the prefill-sized W4A16 / W8A8 GEMMs on the model's gate_up shape.
    Reported as information next to the decode roofline (same JSON line, key ``prefill_kernels``); frac = TFLOP/s over the
    dense MFMA peak of the type (2.5 PFLOP/s f16, 5 PFLOP/s fp8: MI355X_MICROARCH.md)."""
    from aphrodite_engine_amd import _custom_ops as ops
    dev = "cuda"
    Hq, Hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim

    def timeit(fn, iters=5):
        # steady state: these kernels follow a light decode section, and the first ~50 ms of MFMA-heavy work run below
        # the clocks the chip then holds (tools/prefill_section_alone.py: the same section three times back to back gives
        # 882 -> 1037 -> 1066 TFLOP/s on the W4A16 GEMM, 696 -> 815 -> 859 on the attention) -- warm up for >= 60 ms first
        fn()
        torch.cuda.synchronize()
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.06:
            fn()
            torch.cuda.synchronize()
        iters = max(iters, 8)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        e.synchronize()
        return s.elapsed_time(e) * 1e-3 / iters

    out = {}
    g = torch.Generator(device=dev).manual_seed(7)
    T = 8192
    qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device=dev, dtype=torch.float16, generator=g) * 0.5
    q, k, v = qkv[:, :Hq * D].view(T, Hq, D), qkv[:, Hq * D:(Hq + Hkv) * D].view(T, Hkv, D), qkv[:, (Hq + Hkv) * D:].view(T, Hkv, D)
    cu = torch.tensor([0, T], dtype=torch.int32, device=dev)
    t = timeit(lambda: ops.flash_attn_varlen(q, k, v, cu, T, D ** -0.5, causal=True))
    fl = 4.0 * T * T * D * Hq / 2
    out["flash_attn_varlen causal T=8192"] = dict(ms=t * 1e3, TFLOPs=fl / t / 1e12, frac=fl / t / 2.5e15, bound="mfma f16")
    ctx, new, BS = 6144, 2048, 16
    nblk = (ctx + new) // BS
    kc = (torch.randn(nblk, Hkv, D // 8, BS, 8, device=dev, generator=g) * 0.5).half()
    vc = (torch.randn(nblk, Hkv, D, BS, device=dev, generator=g) * 0.5).half()
    bt = torch.randperm(nblk, device=dev, generator=g).to(torch.int32).view(1, nblk)
    o = torch.empty(new, Hq, D, device=dev, dtype=torch.float16)
    i32 = lambda *a: torch.tensor(a, dtype=torch.int32, device=dev)
    args_ = ("auto", kc, vc, bt, i32(0, new), i32(ctx + new), i32(ctx), new, 1.0, 1.0, None, None)
    t = timeit(lambda: ops.context_attention_fwd(q[:new], k[:new], v[:new], o, *args_, max_seq_len=ctx + new, total_kv_tokens=ctx + new))
    fl = 4.0 * D * Hq * (new * ctx + new * new / 2)
    out["context_attention_fwd 6144 cached + 2048 new"] = dict(ms=t * 1e3, TFLOPs=fl / t / 1e12, frac=fl / t / 2.5e15, bound="mfma f16")
    M, K, N = 8192, cfg.hidden_size, 2 * cfg.intermediate_size
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 128, N // 8), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    sc = (torch.rand(K // 128, N, generator=g, device=dev) * 0.01 + 0.005).half()
    a = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
    empty = torch.empty(0, dtype=torch.int32, device=dev)
    t = timeit(lambda: ops.gptq_gemm(a, qw, qz, sc, empty, True, 4), 3)
    fl = 2.0 * M * N * K
    out[f"gptq_gemm W4A16 {M}x{K}x{N}"] = dict(ms=t * 1e3, TFLOPs=fl / t / 1e12, frac=fl / t / 2.5e15, bound="mfma f16")
    del qw, qz, sc
    w8 = (torch.randn(N, K, device=dev, generator=g) * 0.5).to(torch.float8_e4m3fn)
    a8 = a.to(torch.float8_e4m3fn)
    sa = torch.rand(M, 1, device=dev, generator=g) * 0.1 + 0.05
    sb = torch.rand(N, device=dev, generator=g) * 0.01 + 0.005
    ob = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.cutlass_scaled_mm(a8, w8.t(), sa, sb, torch.bfloat16, out=ob), 3)
    out[f"cutlass_scaled_mm W8A8 fp8 {M}x{K}x{N}"] = dict(ms=t * 1e3, TFLOPs=fl / t / 1e12, frac=fl / t / 5.0e15, bound="mfma fp8")
    del w8, a8, sa, sb, ob
    # ---- every prompt-sized GEMM shape of the model, ours next to what the reference already calls on ROCm, same run: gptq
    # dequantise + hipBLASLt (q_gemm.cu:1529-1544's reconstruct + hipBLAS) and torch._scaled_mm (w8a8_utils.py:130,165); causal
    # attention next to torch SDPA.  A stated baseline: ratio > 1 = the hand-written kernel is faster.
    vs = {}
    import torch.nn.functional as F
    try:
        qs = q.transpose(0, 1).unsqueeze(0)
        ks, vs_ = k.transpose(0, 1).unsqueeze(0), v.transpose(0, 1).unsqueeze(0)
        t_lib = timeit(lambda: F.scaled_dot_product_attention(qs, ks, vs_, is_causal=True, scale=D ** -0.5, enable_gqa=True))
        t_our = out["flash_attn_varlen causal T=8192"]["ms"] * 1e-3
        vs["attention causal T=8192"] = dict(ours_ms=t_our * 1e3, library_ms=t_lib * 1e3, ratio=t_lib / t_our, library="torch SDPA")
    except Exception as e:
        vs["attention causal T=8192"] = {"error": repr(e)[:120]}
    h, inter = cfg.hidden_size, cfg.intermediate_size
    shapes = {"qkv": (h, (Hq + 2 * Hkv) * D), "o": (Hq * D, h), "gate_up": (h, 2 * inter), "down": (inter, h)}

    def with_env(name, fn):
        old = os.environ.get(name)
        os.environ[name] = "1"
        try:
            return timeit(fn, 3)
        finally:
            if old is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = old
    for name, (K2, N2) in shapes.items():
        try:
            qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K2 // 8, N2), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
            qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K2 // 128, N2 // 8), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
            sc = (torch.rand(K2 // 128, N2, generator=g, device=dev) * 0.01 + 0.005).half()
            a2 = torch.randn(M, K2, device=dev, dtype=torch.float16, generator=g)
            fn = lambda: ops.gptq_gemm(a2, qw, qz, sc, empty, True, 4)
            t_our, t_lib = timeit(fn, 3), with_env("APHRO_WNA16_NO_LARGE", fn)
            fl2 = 2.0 * M * N2 * K2
            vs[f"W4A16 {name} {M}x{K2}x{N2}"] = dict(ours_ms=t_our * 1e3, ours_TFLOPs=fl2 / t_our / 1e12, library_ms=t_lib * 1e3,
                                                    ratio=t_lib / t_our, library="gptq dequantise + torch.matmul (hipBLASLt)")
            del qw, qz, sc
            w8 = (torch.randn(N2, K2, device=dev, generator=g) * 0.5).to(torch.float8_e4m3fn)
            a8 = a2.to(torch.float8_e4m3fn)
            sa = torch.rand(M, 1, device=dev, generator=g) * 0.1 + 0.05
            sb = torch.rand(N2, device=dev, generator=g) * 0.01 + 0.005
            ob = torch.empty(M, N2, device=dev, dtype=torch.bfloat16)
            fn8 = lambda: ops.cutlass_scaled_mm(a8, w8.t(), sa, sb, torch.bfloat16, out=ob)
            t_our, t_lib = timeit(fn8, 3), with_env("APHRO_FP8_NO_LARGE", fn8)
            vs[f"W8A8 {name} {M}x{K2}x{N2}"] = dict(ours_ms=t_our * 1e3, ours_TFLOPs=fl2 / t_our / 1e12, library_ms=t_lib * 1e3,
                                                   ratio=t_lib / t_our, library="torch._scaled_mm (hipBLASLt)")
            del w8, a8, sa, sb, ob, a2
        except Exception as e:
            vs[f"{name} {M}x{K2}x{N2}"] = {"error": repr(e)[:120]}
    out["vs_library"] = vs
    return out



def prefill_e2e_section(T=8192, library=True, which=("int4", "fp8")):
    """End-to-end prefill (time to first token) of ONE 8192-token prompt -- BASELINE configs[2]'s sequence length -- through
    this package's Llama-3-8B, GPTQ int4 g128 (configs[1]'s model) and compressed-tensors FP8 W8A8 + FP8 KV (configs[2]):
    embedding gather, 32 x (norm, qkv GEMM, rotary, cache write, causal attention, o GEMM, norm, gate_up GEMM, SiluAndMul,
    down GEMM), final norm, lm_head of the last token, argmax -- everything on the clock, no graph.  Harness shape:
    the reference's tests/benchmarks/engine/throughput.py:353-363 (--input-len).
    Next to it, timed in the same process on the same weights, the SAME model with the hand-written MFMA kernels switched
    off in favour of what the reference already calls on ROCm: gptq dequantise + hipBLASLt (torch.matmul) for W4A16
    (q_gemm.cu:1529-1544's reconstruct + hipBLAS path), torch._scaled_mm (hipBLASLt) for W8A8
    (quantization/utils/w8a8_utils.py:130,165), torch SDPA (flash / CK) for the causal attention -- a stated baseline,
    never the target.  frac = time at the dense MFMA peaks (GEMMs at 2.5 PF f16 / 5 PF fp8, attention at 2.5 PF) / time."""
    import dataclasses
    import torch.nn.functional as F
    from aphrodite_engine_amd import _custom_ops as ops
    from aphrodite_engine_amd import model as M
    from aphrodite_engine_amd.attention.backend import MI355XAttentionMetadata
    from aphrodite_engine_amd.quantization.fp8 import CompressedTensorsW8A8Fp8Config
    from aphrodite_engine_amd.quantization.gptq import GPTQConfig
    dev = torch.device("cuda", torch.cuda.current_device())
    BS = 16
    out = {}

    def sdpa_varlen(q, k, v, cu_seqlens, max_seqlen, softmax_scale, causal=True, alibi_slopes=None, window_size=None):
        # one sequence (this section's prompt): [T, H, D] -> [1, H, T, D]; GQA inside SDPA
        assert cu_seqlens.numel() == 2 and alibi_slopes is None and (window_size is None or window_size[0] < 0)
        o = F.scaled_dot_product_attention(q.transpose(0, 1).unsqueeze(0), k.transpose(0, 1).unsqueeze(0),
                                           v.transpose(0, 1).unsqueeze(0), is_causal=causal, scale=softmax_scale,
                                           enable_gqa=True)
        return o.squeeze(0).transpose(0, 1).contiguous()

    def one(name, qc, kv, gemm_peak):
        cfg = dataclasses.replace(M.LLAMA3_8B, max_position_embeddings=max(8192, T))
        model = M.LlamaForCausalLM(cfg, qc, torch.float16, kv)
        model.init_synthetic(dev, seed=0)
        for layer in model.layers:      # the layouts the engine adapter leaves behind (reference_model._finish): ONE interleaved
            layer.enable_fused_silu(32, keep_original=False)      # gate_up copy -> SiluAndMul rides in the prefill GEMM's epilogue
        for layer in model.layers:
            layer.enable_one_copy()         # ... and only the strip-major copy of every int4 matrix resident, as in the headline
        nblk = (T + BS - 1) // BS
        caches = M.make_kv_caches(cfg, nblk, BS, torch.float16, kv, dev, fill=False)
        bt = torch.randperm(nblk, device=dev).to(torch.int32).view(1, nblk)
        pos = torch.arange(T, device=dev, dtype=torch.int64)
        slots = (bt[0, (pos // BS)].long() * BS + pos % BS)
        i32 = lambda *a: torch.tensor(a, dtype=torch.int32, device=dev)
        meta = MI355XAttentionMetadata(
            num_prefills=1, num_prefill_tokens=T, num_decode_tokens=0, slot_mapping=slots, seq_lens=[T],
            seq_lens_tensor=i32(T), max_query_len=T, max_prefill_seq_len=T, max_decode_seq_len=0,
            query_start_loc=i32(0, T), seq_start_loc=i32(0, T), context_lens_tensor=i32(0), block_tables=bt,
            use_cuda_graph=False, max_context_len=0)      # (a fresh prompt: the builder's max_context_len = 0 -> flash_attn_varlen)
        ids = torch.randint(0, cfg.vocab_size, (T, ), device=dev)

        def step():
            h = model(ids, pos, caches, meta)
            return model.sample_greedy(model.compute_logits(h[-1:]))

        def timed(n=3):
            step()
            step()                      # (the first ~50 ms of MFMA-heavy work run below the clocks the chip then holds)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                tok = step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n, tok

        h, i = cfg.hidden_size, cfg.intermediate_size
        qkv = h * (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim
        gemm = 2.0 * T * (qkv + h * h + 3 * h * i) * cfg.num_hidden_layers
        attn = 4.0 * T * T * cfg.head_dim * cfg.num_attention_heads / 2 * cfg.num_hidden_layers
        t_peak = gemm / gemm_peak + attn / 2.5e15
        dt, tok = timed()
        rec = {"ms": dt * 1e3, "tokens_per_s": T / dt, "TFLOPs": (gemm + attn) / dt / 1e12, "frac": t_peak / dt,
               "flops": {"gemm": gemm, "attention": attn}, "kv_cache": kv, "tokens": T}
        if not library:
            out[name] = rec
            del model, caches
            torch.cuda.empty_cache()
            return
        # the library arm: same weights, same process -- on the layouts the loader leaves behind (the [K/8, N] words back in the
        # parameters: its dequantise kernel reads them directly, no strip-major detour is charged to the library)
        for layer in model.layers:
            layer.restore_op_level_layouts()
        saved_env = {k: os.environ.get(k) for k in ("APHRO_WNA16_NO_LARGE", "APHRO_FP8_NO_LARGE")}
        saved_fa = ops.flash_attn_varlen
        try:
            os.environ["APHRO_WNA16_NO_LARGE"] = "1"
            os.environ["APHRO_FP8_NO_LARGE"] = "1"
            ops.flash_attn_varlen = sdpa_varlen
            try:
                dt_l, tok_l = timed()
                rec["library"] = {"ms": dt_l * 1e3, "TFLOPs": (gemm + attn) / dt_l / 1e12,
                                  "what": "dequantise + torch.matmul / torch._scaled_mm (hipBLASLt), torch SDPA attention",
                                  "same_first_token": bool(torch.equal(tok, tok_l))}
                rec["vs_library"] = dt_l / dt
            except Exception as e:
                rec["library"] = {"error": repr(e)[:200]}
        finally:
            ops.flash_attn_varlen = saved_fa
            for k, v_ in saved_env.items():
                if v_ is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v_
        out[name] = rec
        del model, caches
        torch.cuda.empty_cache()

    with torch.no_grad():
        if "int4" in which:
            one("int4_gptq_g128", GPTQConfig(4, 128, False), "auto", 2.5e15)
        if "fp8" in which:
            one("fp8_w8a8_fp8kv", CompressedTensorsW8A8Fp8Config(strategy="channel", is_static_input_scheme=False), "fp8", 5.0e15)
    return out


def all_reduce_section(args, model, device, ca):
    """TP only (every rank calls it): latency of the [bs, hidden] sum the decode layer issues twice, as the decode
    graph runs it -- 32 back-to-back all-reduces captured into one HIP graph -- for the xGMI peer-access kernel
    (with its one-/two-shot choice at this size) and for RCCL, plus the one-/two-shot crossover sizes."""
    import contextlib
    import torch.distributed as dist
    from aphrodite_engine_amd import _lib
    from aphrodite_engine_amd import distributed as D
    world = D.get_tensor_model_parallel_world_size()
    info = {}
    sizes = [args.batch * model.cfg.hidden_size * 2]
    for extra in (64 * 1024, 256 * 1024, 512 * 1024, 1024 * 1024, 4 * 1024 * 1024):
        if extra not in sizes:
            sizes.append(extra)
    lib = _lib.lib()

    def timed_eager(fn, x, n=32, iters=5):
        """32 chained calls issued eagerly between two events (no capture: a failed capture of a collective would leave
        the CUDA generator in capture state and break every later torch.randn of this process)."""
        fn(x)
        torch.cuda.synchronize()
        dist.barrier()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(iters):
            y = x
            for _ in range(n):
                y = fn(y)
        end.record()
        end.synchronize()
        return start.elapsed_time(end) * 1e3 / (iters * n)

    def timed(fn, x, n=32, iters=5):
        fn(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with (ca.capture() if ca is not None and not ca.disabled else contextlib.nullcontext()):
            with torch.cuda.stream(st), torch.cuda.graph(g, stream=st, **GRAPH_KW):
                y = x
                for _ in range(n):
                    y = fn(y)
        torch.cuda.synchronize()
        dist.barrier()
        g.replay()
        torch.cuda.synchronize()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        start.record()
        for _ in range(iters):
            g.replay()
        end.record()
        end.synchronize()
        return start.elapsed_time(end) * 1e3 / (iters * n)
    for nbytes in sizes:
        x = torch.zeros(nbytes // 2, dtype=torch.float16, device=device)
        row = {"algo": None, "custom_us": None, "rccl_us": None,
               # which implementation tensor_model_parallel_all_reduce hands a decode-step tensor of this size to
               "served_by": "RCCL"}
        if ca is not None and not ca.disabled and ca.should_custom_ar(x):
            row["algo"] = "one-shot" if lib.aphro_custom_ar_should_one_shot(world, nbytes) else "two-shot"
            row["served_by"] = "peer-access kernel (" + row["algo"] + ")"
            try:
                row["custom_us"] = timed(lambda t_: ca.custom_all_reduce(t_), x)
                ca.check()
                if nbytes == sizes[0]:
                    # the [M, hidden] sum of the decode layer as it really runs: fused with residual add + RMSNorm + pack
                    xm = x.view(args.batch, -1)
                    res = torch.zeros_like(xm)
                    wn = torch.ones(xm.shape[1], dtype=xm.dtype, device=device)
                    if ca.fused_norm_eligible(xm):
                        row["fused_norm_us"] = timed(lambda t_: (ca.fused_add_rms_norm(t_, res, True, wn, 1e-5, pack=False, want_out=True)[1]), xm)
                        ca.check()
            except Exception as e:       # report, do not fail the section
                row["custom_error"] = repr(e)[:160]

        def rccl(t_):
            dist.all_reduce(t_, group=D._TP_GROUP)
            return t_
        try:
            row["rccl_us"] = timed_eager(rccl, x)      # eager issue: includes the host launch of each collective
        except Exception as e:       # report, do not fail the bench
            row["rccl_us"] = None
            row["rccl_error"] = repr(e)[:120]
        info[str(nbytes)] = row
    return {"world": world, "decode_all_reduce_bytes": sizes[0], "per_size": info,
            "note": "us per all-reduce on rank 0's clock: the peer-access kernel as 32 chained calls per HIP-graph replay, RCCL as 32 chained eager calls (host launch included)"}


def timed_decode(loop, args, world, one_gpu, device, ca=None):
    """Capture loop.step into a HIP graph (the all-reduce inputs seen while capturing are registered with the peers
    afterwards) and time args.steps replays: barrier + synchronize on both sides, max over ranks."""
    import contextlib
    import torch.distributed as dist
    for _ in range(2):
        loop.step()
    torch.cuda.synchronize()
    run = loop.step
    if not args.no_graph:
        graph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with (ca.capture() if ca is not None and not ca.disabled else contextlib.nullcontext()):
            with torch.cuda.stream(s):
                loop.step()
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(graph, **GRAPH_KW):
                loop.step()
        run = graph.replay
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device="cpu" if one_gpu else device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if ca is not None:
        ca.check()
    return elapsed


def tp_section(args, device, world, rank, one_gpu):
    """After the replica ("dp") timing of an N > 1 run: the SAME model sharded Megatron-style over the N ranks
    (north_star: RCCL all-reduce over xGMI overlapped with the quantized GEMMs), so that ONE `bench.py --gpus N` run
    yields both points of the scaling story: tokens/s of TP = N with the all-reduce overlapped with the next weights'
    prefetch and without, which all-reduce served the [M, hidden] sums (peer-access kernel vs RCCL, one- / two-shot),
    their latencies per size, and how many ranks RCCL really spans.  Never executed on real links before the driver's
    first 8-GPU run: every failure is reported in the dict instead of raised."""
    import copy
    import torch.distributed as dist
    from aphrodite_engine_amd import distributed as D
    out = {"tp": world}
    if one_gpu and not args.no_graph:
        # test rig (gloo on one GPU): the host-side collectives of gloo cannot be captured; RCCL's can (tests/test_custom_ar_gpu.py)
        args = copy.copy(args)
        args.no_graph = True
        out["graph"] = "off (gloo test rig)"
    ones = torch.ones(1, device="cpu" if one_gpu else device)
    dist.all_reduce(ones)
    out["ranks_seen_by_collective"] = int(ones.item())
    out["backend"] = dist.get_backend()
    D.init_tensor_parallel(world)
    ca = None
    if not os.environ.get("APHRO_NO_CUSTOM_AR"):
        try:
            ca = D.enable_custom_all_reduce(device)
            out["custom_all_reduce"] = "enabled" if (ca is not None and not ca.disabled) else \
                f"disabled: {getattr(ca, 'disabled_reason', 'ineligible')}"
        except Exception as e:
            out["custom_all_reduce"] = f"failed: {e!r}"[:200]
            ca = None
    model, cfg, dtype = build(args, device)
    total = 2 * (args.warmup + args.steps + 8)
    loop = DecodeLoop(model, cfg, dtype, args, device, total)
    # first contact with the real group: verify the peer-access kernels against RCCL per message size before anything is
    # captured or timed (a size that does not verify sends every all-reduce to RCCL: a report, not a hang or a wrong sum)
    try:
        out["self_check"] = D.all_reduce_self_check(device)
    except Exception as e:
        out["self_check"] = {"error": repr(e)[:200]}
    with torch.no_grad():
        # the side-stream overlap stays on only behind a measured win on THIS group (distributed.choose_all_reduce_overlap):
        # both arms are timed, the record says which one the step then runs
        def arm():
            el = timed_decode(loop, args, world, one_gpu, device, ca)
            name = "overlap" if D.get_all_reduce_overlap() is not None else "no_overlap"
            out[name] = {"tokens_per_s": args.batch * args.steps / el, "ms_per_step": el / args.steps * 1e3}
            return el
        try:
            out["overlap_decision"] = D.choose_all_reduce_overlap(device, arm)
        except Exception as e:
            out["overlap_decision"] = {"error": repr(e)[:200]}
        D.enable_all_reduce_overlap(device, enabled=False)
        try:
            out["all_reduce_info"] = all_reduce_section(args, model, device, ca)
        except Exception as e:
            out["all_reduce_info"] = {"error": repr(e)[:200]}
    del loop, model
    torch.cuda.empty_cache()
    return out



# ---- multi-rank control flow (first contact with a multi-GPU node happens in the driver's own run: every step below is
# ---- written so that the replica ("dp") line survives whatever the tensor-parallel side does) -------------------------------
def init_process_group_safe(world, rank, device, want_nccl):
    """RCCL ("nccl") for the replicas' barriers and the max over ranks; if it cannot be brought up the run continues on gloo
    (the replicas exchange no data) and says so in the line.  Returns {"backend", "world", "note"}."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    info = {"note": None}
    if want_nccl:
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            t = torch.ones(1, device=device)
            dist.all_reduce(t)                      # first collective: communicator really up, every rank present
            torch.cuda.synchronize()
            if int(t.item()) != world:
                raise RuntimeError(f"first all-reduce saw {int(t.item())} of {world} ranks")
        except Exception as e:                      # noqa: BLE001 -- anything: the DP line must not depend on RCCL
            info["note"] = f"nccl process group failed ({e!r}"[:200] + "): replicas synchronise over gloo"
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:
                pass
            # a different rendezvous port: the failed attempt may still hold the first one
            os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 17)
            os.environ.pop("TORCHELASTIC_USE_AGENT_STORE", None)
            dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    info["backend"] = dist.get_backend()
    info["world"] = dist.get_world_size()
    return info


def timed_region(run, steps, world, sync, max_device):
    """EXACTLY `steps` calls of run() bracketed by barrier + device synchronize on both sides; max over ranks."""
    import torch.distributed as dist
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=max_device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    return elapsed


def run_tp_children(argv, world, rank):
    """The tensor-parallel section of an N > 1 run in CHILD processes (one per rank, own process group on its own
    rendezvous port): a collective that hangs, an IPC mapping that faults or an RCCL abort ends a child, not the process
    that holds the replica numbers.  Every rank spawns its child and waits (APHRO_BENCH_TP_TIMEOUT_S, default 200 s);
    rank 0 returns the child's result dict (or {"error": ...})."""
    import signal
    import subprocess
    import torch.distributed as dist
    port = [0]
    if rank == 0:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port[0] = sk.getsockname()[1]
    dist.broadcast_object_list(port, src=0)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port[0]), APHRO_BENCH_TP_CHILD="1")
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)          # the child's rank 0 hosts its own store on the new port
    cmd = [sys.executable, os.path.abspath(__file__), "--tp-child"] + [a for a in argv if a != "--tp-child"]
    budget = float(os.environ.get("APHRO_BENCH_TP_TIMEOUT_S", "200"))
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    result = None
    try:
        out, err = p.communicate(timeout=budget)
        if p.returncode != 0:
            result = {"error": f"tensor-parallel child exited with {p.returncode}: {(err or out)[-300:]}"}
        else:
            for ln in out.splitlines():
                if ln.startswith("TP_RESULT "):
                    result = json.loads(ln[len("TP_RESULT "):])
            if result is None and rank == 0:
                result = {"error": "tensor-parallel child printed no result: " + (err or out)[-200:]}
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)               # the session we started, nothing else
        except ProcessLookupError:
            pass
        p.communicate()
        result = {"error": f"tensor-parallel child timed out after {budget:.0f} s (killed): replica numbers are unaffected"}
    return result


def tp_child_main(args):
    """`bench.py --tp-child ...` (spawned by run_tp_children): own process group, the tensor-parallel section, one
    TP_RESULT line on rank 0."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dry = os.environ.get("APHRO_BENCH_DRY_CPU") == "1"
    one_gpu = os.environ.get("APHRO_BENCH_ONE_GPU") == "1"
    import torch.distributed as dist
    if dry:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        fail = os.environ.get("APHRO_BENCH_DRY_TP_FAIL", "")
        if fail == "crash" and rank == world - 1:
            os.abort()
        if fail == "hang":
            time.sleep(3600)
        if fail == "raise":
            raise RuntimeError("dry-run tensor-parallel failure")
        t = torch.ones(1)
        dist.all_reduce(t)
        out = {"tp": world, "backend": dist.get_backend(), "ranks_seen_by_collective": int(t.item()), "dry_run": True}
    else:
        local_rank = 0 if one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        info = init_process_group_safe(world, rank, device, want_nccl=not one_gpu)
        from aphrodite_engine_amd import _lib
        _lib.lib()
        out = tp_section(args, device, world, rank, one_gpu)
        out["process_group"] = info
    if rank == 0:
        print("TP_RESULT " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main_dry_cpu(args):
    """APHRO_BENCH_DRY_CPU=1 (tests/test_host_cpu.py): the multi-rank control flow of main() -- process group, barriers,
    max over ranks, the tensor-parallel children and their failure containment, the ONE line of rank 0 -- with a stub in
    place of the model, on CPU over gloo.  The numbers mean nothing."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    pg = {"backend": None, "world": 1, "note": None}
    if world > 1:
        pg = init_process_group_safe(world, rank, torch.device("cpu"), want_nccl=False)
    x = torch.zeros(64, 64)

    def step():
        x.add_(1.0)
    for _ in range(args.warmup):
        step()
    elapsed = timed_region(step, args.steps, world, lambda: None, "cpu")
    line = None
    if rank == 0:
        line = {"metric": "output tokens/sec + HBM-roofline %, Llama-3-8B int4/fp8 @ 1/2/4/8 MI355X", "value": args.batch * args.steps * world / elapsed,
                "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "dry run", "data": "none",
                "config": {"workload": "DRY RUN on CPU (control flow only)", "INVALID": "dry run"}, "process_group": pg}
    if world > 1 and not args.no_tp_section:
        tp = run_tp_children(sys.argv[1:], world, rank)
        if rank == 0:
            line["tp_section"] = tp
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.tp_child:
        return tp_child_main(args)
    if os.environ.get("APHRO_BENCH_DRY_CPU") == "1":
        return main_dry_cpu(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (torch.cuda.is_available() is False)", file=sys.stderr)
        sys.exit(2)
    # APHRO_BENCH_ONE_GPU=1 (test rig only): every rank runs on cuda:0 and the process group is gloo -- exercises the
    # multi-rank control flow (barriers, max over ranks, rank-0 line) on a one-GPU box; the numbers are meaningless.
    one_gpu = os.environ.get("APHRO_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    pg_info = {"backend": None, "world": 1, "note": None}
    if world > 1:
        pg_info = init_process_group_safe(world, rank, device, want_nccl=not one_gpu)
    from aphrodite_engine_amd import _lib
    from aphrodite_engine_amd import distributed as D
    _lib.lib()  # fail loudly if the HIP library is missing
    tp = world if (args.parallelism == "tp" and world > 1) else 1
    D.init_tensor_parallel(tp)
    sim_ar_us = None
    if args.sim_tp > 1:
        assert world == 1, "--sim-tp times one rank of the group on one GPU"
        # [M, hidden] f16 message of this workload; stub latency = the peer-access kernel + flag protocol measured with
        # the ranks on one GPU (no link time): <= 256 KiB one-shot 6.7 us, above two-shot 11.4 us (4 ranks)
        msg = args.batch * {"llama3-8b": 4096, "llama3-70b": 8192, "mixtral-8x7b": 4096}[args.model] * 2
        sim_ar_us = args.sim_ar_us if args.sim_ar_us > 0 else (6.7 if msg <= 256 * 1024 else 11.4)
        D.init_simulated_tensor_parallel(args.sim_tp, sim_ar_us)
        if not args.sim_ar_stub:
            # the REAL peer-access kernels (all-reduce, and all-reduce + residual add + RMSNorm + pack in one launch) on a
            # loopback communicator: flags, scratch and sim_tp reads per element against local memory, zero link time.
            # The stream-holding stub stays the fallback for sizes the kernels do not take (and --sim-ar-stub forces it).
            D.enable_loopback_all_reduce(device)
    ca = None
    if tp > 1 and not os.environ.get("APHRO_NO_CUSTOM_AR"):
        # xGMI peer-access all-reduce for the [M, hidden] sums (RCCL stays the fallback for ineligible sizes)
        ca = D.enable_custom_all_reduce(device)
    overlap = None
    if tp > 1 and args.overlap and not args.no_overlap:
        overlap = D.enable_all_reduce_overlap(device)

    model, cfg, dtype = build(args, device)
    ops_path = world == 1 and not args.no_ops_path and args.model == "llama3-8b" and args.sim_tp <= 1
    total = (args.warmup + args.steps + 4) * (2 if ops_path else 1)
    loop = DecodeLoop(model, cfg, dtype, args, device, total)

    with torch.no_grad():
        # eager warm-up (allocates workspaces, loads code objects)
        for _ in range(2):
            loop.step()
        torch.cuda.synchronize()
        graph = None
        if not args.no_graph:
            import contextlib
            graph = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            # inputs of the all-reduces seen while capturing are registered with the peers afterwards
            with (ca.capture() if ca is not None and not ca.disabled else contextlib.nullcontext()):
                with torch.cuda.stream(s):
                    loop.step()
                torch.cuda.current_stream().wait_stream(s)
                with torch.cuda.graph(graph, **GRAPH_KW):
                    loop.step()
        run = graph.replay if graph is not None else loop.step
        for _ in range(args.warmup):
            run()
        max_dev = "cpu" if (one_gpu or pg_info["backend"] == "gloo") else device
        elapsed = timed_region(run, args.steps, world, torch.cuda.synchronize, max_dev)
        if ca is not None:
            ca.check()      # a timed-out barrier would have produced garbage: fail loudly
        ctx_end = int(loop.meta.seq_lens_tensor[0].item())
        ctx_mean_end = float(loop.meta.seq_lens_tensor.float().mean().item())
        ctx_mean_start = loop.ctx_sum0 / args.batch
        active_frac = 1.0
        if cfg.num_local_experts:
            # experts the decode step really routes to (one more eager step with the routing recorded)
            for layer in model.layers:
                layer.experts.record_routing = True
            loop.step()
            torch.cuda.synchronize()
            active_frac = sum(int(torch.unique(l.experts.last_topk_ids).numel()) for l in model.layers) \
                / (len(model.layers) * cfg.num_local_experts)
            for layer in model.layers:
                layer.experts.record_routing = False
        ar_info = None
        if tp > 1:
            if overlap is not None:
                D.enable_all_reduce_overlap(device, enabled=False)     # measure the bare collective
            try:
                ar_info = all_reduce_section(args, model, device, ca)
            except Exception as e:
                ar_info = {"error": repr(e)[:200]}
        roof = roofline_section(model, loop, args) if rank == 0 else None
        # (the per-kernel section above runs on the allocations the timed step ran on; the op-by-op leg below releases and
        #  rebuilds every decode layout, which moves them -- round 6, profiles/r6_one_copy.txt)
        # ---- the same workload on the OP-BY-OP path: what a reference LlamaDecoderLayer (modeling/models/llama.py:234,
        # quant_method.apply -> ops.gptq_gemm, quantization/gptq.py:230-243; Attention.forward -> ops.reshape_and_cache +
        # ops.paged_attention_*) gets through the plugin without adopting forward_decode_fused: one launch per reference op
        # (17 per layer instead of 7), row-major activations between them.  Captured and timed exactly like the headline.
        elapsed_ops = None
        if ops_path:
            # (on the layouts the loader leaves behind -- no interleaved gate_up to undo per step, the op-level strip-major
            #  copy of the big matrices in place; the fused layouts are rebuilt afterwards for the per-kernel section)
            for layer in model.layers:
                layer.restore_op_level_layouts()
            model.use_fused_decode = False
            for _ in range(2):
                loop.step()
            torch.cuda.synchronize()
            run_ops = loop.step
            if not args.no_graph:
                g_ops = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_ops, **GRAPH_KW):
                    loop.step()
                run_ops = g_ops.replay
            for _ in range(args.warmup):
                run_ops()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                run_ops()
            torch.cuda.synchronize()
            elapsed_ops = time.perf_counter() - t1
            model.use_fused_decode = True
            if not os.environ.get("APHRO_NO_FUSED_SILU"):
                for layer in model.layers:
                    layer.enable_fused_silu(args.batch, keep_original=False)
                if not getattr(args, "two_copies", False):
                    for layer in model.layers:
                        layer.enable_one_copy()
                torch.cuda.synchronize()
                torch.cuda.empty_cache()    # (the [K/8, N] blocks of the op-by-op leg go back to the driver, as after build_model)

    line = None
    if rank == 0:
        replicas = world if tp == 1 else 1
        tokens = args.batch * args.steps * replicas
        ms_per_step = elapsed / args.steps * 1e3
        # step-level algorithmic bytes
        weight_bytes_resident = getattr(model, "weight_bytes_resident", None)
        w_bytes = model.weight_bytes_per_layer(active_frac) * cfg.num_hidden_layers
        lm_head = model.lm_head.numel() * 2
        esz = 1 if args.kv_cache_dtype != "auto" else 2
        # mean context over the timed steps (uniform lengths: the first sequence's; --ragged: the batch mean)
        ctx_mid = (ctx_mean_start + args.warmup + 2 + ctx_mean_end) / 2.0
        tp_eff = max(tp, args.sim_tp, 1)
        kv_bytes = args.batch * ctx_mid * 2 * max(1, cfg.num_key_value_heads // tp_eff) * cfg.head_dim * esz * cfg.num_hidden_layers
        step_bytes = w_bytes + lm_head + kv_bytes
        # dominant kernel = the kernel TEMPLATE with the largest total time per step: several
        # projections may run one template (qkv / o / down on wna16_gemm_kernel); its roofline point is their summed
        # algorithmic bytes over their summed launch time, per launch the average.
        groups = {}
        for k, v in roof.items():
            tmpl = v.get("kernel", k).split(" ")[0].split("<")[0]
            gsum = groups.setdefault(tmpl, {"bytes": 0.0, "seconds": 0.0, "members": []})
            gsum["bytes"] += v["bytes"]
            gsum["seconds"] += v["seconds"]
            gsum["members"].append(k)
        dom_name = max(groups, key=lambda k: groups[k]["seconds"])
        dom = groups[dom_name]
        achieved = dom["bytes"] / dom["seconds"] / 1e9
        n_members = len(dom["members"])
        # HBM bytes per launch: PMC counters cannot be collected from inside this process (rocprofv3 wraps it).  The
        # committed passes of the newest round (profiles/r6_pmc_traffic.json, else r5 / r4: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE,
        # separate runs of tools/prof_step_kernels.py at THESE shapes, gfx950 x2 correction on FETCH_SIZE) are attached
        # with their provenance; null when the file has no entry for the dominant template.
        traffic = None
        traffic_src = None
        try:
            pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
            pmc_file = next(f for f in ("r6_pmc_traffic.json", "r5_pmc_traffic.json", "r4_pmc_traffic.json") if os.path.exists(os.path.join(pdir, f)))
            pmc = json.load(open(os.path.join(pdir, pmc_file)))
            pref = "fp8_" if args.quant.startswith("fp8") else ""     # (the FP8 resident kernels have their own entries)
            ent = [pmc["kernels"][pref + m] for m in dom["members"] if pref + m in pmc.get("kernels", {})]
            if len(ent) == n_members and args.quant in ("gptq", "awq", "fp8ct", "fp8") and args.batch == 32 \
                    and (args.kv_cache_dtype == "auto" or pref):
                traffic = sum(e["hbm_bytes_per_launch"] for e in ent) / n_members
                traffic_src = pmc.get("source")
        except Exception:
            pass
        # The same fraction from the COMMITTED kernel trace of this command (profiles/r6_bench_one_copy_kernel_stats.csv, rocprofv3
        # --kernel-trace --stats): averages over every launch of the run -- mostly the step's own, back to back in the graph, where
        # a launch starts under the previous one's tail -- next to the live figure, which times each kernel on its own.  Only for
        # the workload that trace was taken on (the default int4 line); null otherwise.
        traced = None
        try:
            if dom_name == "wna16_gemm_stream_kernel" and args.model == "llama3-8b" and args.quant == "gptq" and args.batch == 32 \
                    and tp == 1 and not args.ragged and args.ctx == 1024:
                import csv as _csv
                tf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r6_bench_one_copy_kernel_stats.csv")
                avg = [float(r["AverageNs"]) for r in _csv.DictReader(open(tf)) if "wna16_gemm_stream_kernel" in r["Name"]]
                if len(avg) == n_members:
                    t_us = sum(avg) / 1e3
                    traced = {"total_us_per_layer": t_us, "avg_launch_us": t_us / n_members,
                              "frac": dom["bytes"] / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                              "source": "profiles/r6_bench_one_copy_kernel_stats.csv (committed rocprofv3 --kernel-trace --stats run of this "
                                        "command on the round's last tree; NOT measured in this process)"}
        except Exception:
            traced = None
        line = {
            "metric": "output tokens/sec + HBM-roofline %, Llama-3-8B int4/fp8 @ 1/2/4/8 MI355X",
            "value": tokens / elapsed,
            "unit": "tokens/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak" if tp == 1 else "strong",
            "vs_baseline": None,
            "dtype": {"gptq": "int4 weights x f16 (fp32 accumulate)", "awq": "int4 weights x f16 (fp32 accumulate)",
                      "fp8": "fp8-e4m3 x fp8-e4m3 (fp32 accumulate)",
                      "fp8ct": "fp8-e4m3 x fp8-e4m3 (fp32 accumulate)"}[args.quant],
            "data": "synthetic (random-init weights in the real GPTQ/FP8 formats, random token ids, random-permutation block tables)",
            "config": {
                "workload": f"{ {'llama3-8b': 'Llama-3-8B', 'llama3-70b': 'Llama-3-70B', 'mixtral-8x7b': 'Mixtral-8x7B (top-2 of 8 experts)'}[args.model]} "
                            f"{args.quant.upper()} {('W8A8' + (' static input scales' if args.act_scheme == 'static' else '')) if args.quant.startswith('fp8') else '4-bit g128'}, decode, "
                            f"{'greedy' if args.sampling == 'greedy' else 'random sampling (T 0.8, top-k 50, top-p 0.95)'}, "
                            f"bs={args.batch}/GPU, "
                            + (f"ragged contexts randint(1, {args.ctx}) (seed 0; mean {ctx_mean_start:.0f}->{ctx_mean_end:.0f}), "
                               if args.ragged else f"context {args.ctx}->{ctx_end}, ")
                            + f"kv_cache={args.kv_cache_dtype}, "
                            f"block_size=16, HIP-graph={'off' if args.no_graph else 'on'}",
                "global_batch": args.batch * replicas,
                "seq_len": args.ctx,
                "parallelism": f"{args.parallelism}{world}",
                "all_reduce": ("xGMI peer-access kernel" if ca is not None and not ca.disabled else "RCCL") if tp > 1 else None,
                "all_reduce_overlap": (overlap is not None) if tp > 1 else None,
                "layers": cfg.num_hidden_layers,
                "process_group": {"backend": pg_info["backend"], "ranks": pg_info["world"], "note": pg_info["note"]},
            },
            "step_hbm": {"active_expert_fraction": active_frac, "algorithmic_bytes": step_bytes,
                         "weight_bytes_one_copy": w_bytes + lm_head + model.embed_tokens.numel() * 2,
                         "weight_bytes_resident": weight_bytes_resident, "achieved_GBps": step_bytes / (elapsed / args.steps) / 1e9,
                         "frac_of_peak": step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
            "roofline": {"bound": "hbm", "kernel": dom_name, "members": dom["members"], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_launch": dom["bytes"] / n_members, "avg_launch_us": dom["seconds"] / n_members * 1e6,
                         "total_us_per_layer": dom["seconds"] * 1e6, "traced": traced,
                         # the whole step against the same peak (the headline fraction; `frac` is the dominant kernel's)
                         "step_frac": step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
            "roofline_all": {k: {"GBps": v["bytes"] / v["seconds"] / 1e9, "frac": v["bytes"] / v["seconds"] / 1e9 / HBM_PEAK_GBS,
                                 "bytes": v["bytes"], "avg_us": v["seconds"] * 1e6,
                                 **({"active_experts": v["active_experts"]} if "active_experts" in v else {})}
                             for k, v in roof.items()},
        }
        if elapsed_ops is not None:
            line["value_ops_path"] = args.batch * args.steps / elapsed_ops
            line["ops_path"] = {"ms_per_step": elapsed_ops / args.steps * 1e3,
                                "what": "same workload, one launch per reference op (LlamaDecoderLayer.forward -> quant_method.apply / "
                                        "ops.* through the plugin surface), HIP graph; `value` is the fused fast path (forward_decode_fused)"}
        if ar_info is not None:
            line["all_reduce_info"] = ar_info
        if args.layers:
            line["config"]["INVALID"] = "debug run with fewer layers"
        if args.sim_tp > 1:
            line["config"]["parallelism"] = f"ONE rank of tp{args.sim_tp}, simulated on one GPU"
            line["config"]["simulated_all_reduce_us"] = sim_ar_us if args.sim_ar_stub else None
            line["config"]["all_reduce"] = ("stub: a launch holding the stream for simulated_all_reduce_us" if args.sim_ar_stub else
                                            "the peer-access kernels on a LOOPBACK communicator (csrc/custom_all_reduce.hip: "
                                            "all-reduce + residual add + RMSNorm + pack in one launch where the layer allows, "
                                            f"fused={'off' if os.environ.get('APHRO_NO_FUSED_AR_NORM') == '1' else 'on'})")
            line["config"]["NOTE"] = ("per-GPU shard timing: real shard shapes and kernels of one rank; the all-reduces run the real "
                                      "kernel instruction stream, flag protocol and scratch traffic with every peer resolved to this "
                                      "GPU's own memory (no link time: a lower bound); tokens/s is what the TP group would deliver "
                                      "at that all-reduce latency; outputs are not meaningful (the sums are over copies of one partial)")

    if world > 1 and tp == 1 and not args.no_tp_section and args.model == "llama3-8b":
        # free the replica first; the sharded model runs in CHILD processes (run_tp_children): whatever happens there, the
        # replica numbers above are already measured and the line below is printed
        loop = model = None
        torch.cuda.empty_cache()
        try:
            tp_info = run_tp_children(sys.argv[1:], world, rank)
        except Exception as e:                      # noqa: BLE001
            tp_info = {"error": repr(e)[:300]}
        if rank == 0:
            line["tp_section"] = tp_info
    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if world == 1 and not args.no_prefill_info and args.model == "llama3-8b":
        try:
            del loop, model               # the decode state is no longer needed: give its memory back first
            torch.cuda.empty_cache()
            line["prefill_kernels"] = prefill_section(cfg)
            # counter evidence for the MFMA-bound kernels, attached the way the decode `traffic` is: the committed
            # rocprofv3 --pmc passes of tools/prof_prefill_r6.py (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES ... GRBM_GUI_ACTIVE),
            # reduced by tools/pmc_prefill.py -- HBM-side bytes per launch against the algorithmic ones, MFMA-pipe utilisation
            try:
                pm = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r6_pmc_prefill.json")))
                line["prefill_kernels"]["pmc"] = {
                    "source": pm["source"],
                    "rows": [{k: r[k] for k in ("kernel", "case", "read_bytes", "alg_read_bytes", "read_ratio", "write_bytes",
                                                "alg_write_bytes", "mfma_util", "clock_ghz_under_pmc", "wait_any_frac",
                                                "wait_inst_any_frac", "active_inst_any_frac", "active_inst_valu_frac")}
                             for r in pm["rows"]]}
            except Exception as e:
                line["prefill_kernels"]["pmc"] = {"error": repr(e)[:120]}
        except Exception as e:
            line["prefill_kernels"] = {"error": repr(e)}
        if not args.no_prefill_e2e:
            try:
                torch.cuda.empty_cache()
                line["prefill_e2e"] = prefill_e2e_section()
            except Exception as e:
                line["prefill_e2e"] = {"error": repr(e)[:300]}
    if world == 1 and not args.no_cpu_baseline and args.model == "llama3-8b":
        try:
            line["cpu_baseline"] = cpu_baseline(args, cfg)
        except Exception as e:  # never lose the GPU number to a CPU-side problem
            line["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(),
                                    "kind": "port", "sample": f"failed: {e!r}"}
    if world == 1 and default_workload(args) and not args.no_extra_legs and not os.environ.get("APHRO_BENCH_LEG"):
        line["legs"] = extra_legs(args)
        # the FP8 half of the metric next to the int4 headline
        line["fp8"] = {k: line["legs"][k] for k in ("fp8_w8a8_fp8kv_ctx8192", "fp8_w8a8_ctx1024") if k in line["legs"]}
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def default_workload(args) -> bool:
    """True for the driver's command (`bench.py --gpus 1 --steps K --warmup W`): the extra legs ride on that line only."""
    return (args.model == "llama3-8b" and args.quant == "gptq" and args.kv_cache_dtype == "auto" and args.batch == 32
            and args.ctx == 1024 and args.sim_tp <= 1 and not args.ragged and not args.layers and not args.no_graph
            and args.sampling == "greedy")


# The legs of the default run: BASELINE.json configs[2] (FP8 weights + FP8 KV at 8192 tokens), the FP8 model at the
# headline's context, a ragged batch of the headline workload, and ONE rank of configs[3] / configs[4] (real shard shapes,
# all-reduces replaced by a stream-holding stub -- no multi-GPU box has been available to this project).  Each leg is this
# script again in a child process (own HIP context, the parent's model already freed), timed exactly like the headline.
LEGS = {
    "fp8_w8a8_fp8kv_ctx8192": ["--quant", "fp8ct", "--kv-cache-dtype", "fp8", "--ctx", "8192"],
    "fp8_w8a8_ctx1024": ["--quant", "fp8ct"],
    "int4_ragged": ["--ragged"],
    "cfg3_llama70b_awq_tp8_one_rank": ["--model", "llama3-70b", "--quant", "awq", "--batch", "64", "--sim-tp", "8"],
    "cfg4_mixtral_gptq_tp4_one_rank": ["--model", "mixtral-8x7b", "--sim-tp", "4"],
}


def extra_legs(args):
    import subprocess
    out = {}
    env = dict(os.environ, APHRO_BENCH_LEG="1")
    budget_s = float(os.environ.get("APHRO_BENCH_LEG_TIMEOUT_S", "150"))
    for name, extra in LEGS.items():
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--no-cpu-baseline", "--no-prefill-info", "--no-ops-path", "--no-prefill-e2e"] + extra
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=budget_s)
            rec = None
            for ln in r.stdout.splitlines():
                if ln.startswith("{") and '"metric"' in ln:
                    rec = json.loads(ln)
            if rec is None:
                out[name] = {"error": (r.stderr or r.stdout)[-300:]}
                continue
            roof = rec.get("roofline", {})
            out[name] = {"value": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"],
                         "workload": rec["config"]["workload"], "parallelism": rec["config"].get("parallelism"),
                         "roofline": {"kernel": roof.get("kernel"), "frac": roof.get("frac"), "step_frac": roof.get("step_frac"),
                                      "avg_launch_us": roof.get("avg_launch_us")},
                         "per_kernel_us": {k: round(v["avg_us"], 2) for k, v in rec.get("roofline_all", {}).items()}}
            if rec["config"].get("all_reduce"):
                out[name]["all_reduce"] = rec["config"]["all_reduce"]
            if "simulated_all_reduce_us" in rec["config"]:
                out[name]["simulated_all_reduce_us"] = rec["config"]["simulated_all_reduce_us"]
        except Exception as e:
            out[name] = {"error": repr(e)[:300]}
    return out


if __name__ == "__main__":
    main()
