"""CPU-side tests: the C-ABI library loads and exports every declared symbol
(no compute calls without a GPU), the host-side mirror of the reference's
plugin surface behaves like the reference, and the TP collectives work over
gloo with world_size 2."""
import os
import re

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "aphrodite_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(aphro_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from aphrodite_engine_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build the HIP library first (__graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/aphrodite_mi355x.h but not exported"
    # the ctypes table binds exactly the declared ABI
    assert sorted(_lib.SIGNATURES) == syms
    assert _lib.lib().aphro_abi_version() == 1


def test_int4_gemm_plans_through_their_workspace_sizes():
    """The launch plans of the prefill-sized and the 33..64-row int4 GEMMs are host logic (no device call): their
    workspace queries say which plan a shape gets -- K slices by waves and slab bytes (wna16_gemm_large.hip), 8
    K-splitting waves without slabs when the column tiles cover the chip (wna16_gemm_mid.hip)."""
    from aphrodite_engine_amd import _lib
    L = _lib.lib()
    slab = lambda m, n: m * n * 4
    large = lambda m, n, k: L.aphro_wna16_gemm_large_workspace_bytes(m, n, k, k // 128, 0)
    assert large(128, 28672, 4096) == 2 * slab(128, 28672)        # 224 two-wave workgroups -> 2 K slices
    assert large(256, 6144, 4096) == 4 * slab(256, 6144)          # 8 slices would be 50 MB of slabs: capped
    assert large(1024, 6144, 4096) == 0                           # one slab pair already 50 MB: no split
    assert large(96, 4096, 14336) == 14 * slab(96, 4096)
    assert large(4096, 28672, 4096) == 4096 + 256 * 8 * 32 * 1024  # stream-K: flags + one image per workgroup
    # round 6, from 6144 rows: + the f16 W^T [N, K] of the two-pass form (dequantised once per call instead of once per row tile)
    assert large(8192, 28672, 4096) == 28672 * 4096 * 2 + 4096 + 256 * 8 * 32 * 1024
    assert large(256, 28672, 4096, ) % 256 == 0
    mid_ok = lambda m, n, k: L.aphro_wna16_gemm_mid_supported(m, n, k, k // 128)
    mid = lambda m, n, k: L.aphro_wna16_gemm_mid_workspace_bytes(m, n, k, k // 128)
    assert mid_ok(64, 28672, 4096) and mid_ok(1, 128, 512) and not mid_ok(65, 28672, 4096)
    assert not mid_ok(64, 28672, 4096 + 128) and not mid_ok(64, 28672 + 64, 4096)
    assert mid(64, 28672, 4096) == 64 * 4096 * 2                  # packed activations only: 8 waves split K, no slabs
    assert mid(32, 28672, 4096) == 32 * 4096 * 2
    assert mid(64, 4096, 14336) == 64 * 14336 * 2 + 7 * slab(64, 4096)
    assert mid(65, 28672, 4096) == 0


def test_new_entry_points_refuse_bad_arguments_before_touching_a_device():
    """Argument checks run on the host, before any device call: error code + message, nothing launched (no GPU here)."""
    import ctypes
    from aphrodite_engine_amd import _lib
    L = _lib.lib()
    L.aphro_last_error.restype = ctypes.c_char_p
    # SiluAndMul on slabs: missing slabs, unsupported width, unsupported dtype
    assert L.aphro_silu_and_mul_pack_slabs(None, 2, None, None, 4, 1024, _lib.F16, None) != 0
    assert b"slabs" in L.aphro_last_error()
    assert L.aphro_silu_and_mul_pack_slabs(ctypes.c_void_p(256), 2, ctypes.c_void_p(256), None, 4, 1000, _lib.F16, None) != 0
    assert L.aphro_silu_and_mul_pack_slabs(ctypes.c_void_p(256), 2, None, None, 4, 1024, _lib.F32, None) != 0
    # 33..64 rows: the plan query answers without a device, > 64 rows is refused, the row-major one-launch op stays <= 32
    assert L.aphro_wna16_resident_ksplit(64, 28672, 4096, 32) == 1 and L.aphro_wna16_resident_ksplit(33, 7168, 8192, 64) == 4
    assert L.aphro_wna16_resident_ksplit(65, 28672, 4096, 32) == 0
    assert L.aphro_wna16_gemm_rowmajor_supported(32, 28672, 4096, 32, _lib.F16) == 1
    assert L.aphro_wna16_gemm_rowmajor_supported(33, 28672, 4096, 32, _lib.F16) == 0
    # zero tokens: nothing to do, no device call
    assert L.aphro_silu_and_mul_pack_slabs(ctypes.c_void_p(256), 2, ctypes.c_void_p(256), None, 0, 1024, _lib.F16, None) == 0


def test_gate_up_interleave_round_trip():
    """interleave_gate_up (load-time column permutation for the SiluAndMul epilogue) and its inverse are pure tensor ops:
    column 2j = gate_j, 2j + 1 = up_j; zero-point nibbles follow their columns; the inverse restores every bit."""
    import torch
    from aphrodite_engine_amd import _custom_ops as ops
    g = torch.Generator().manual_seed(0)
    K, N, G = 256, 64, 2
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, dtype=torch.int64).to(torch.int32)
    qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (G, N // 8), generator=g, dtype=torch.int64).to(torch.int32)
    sc = torch.rand(G, N, generator=g).half()
    qi, zi, si = ops.interleave_gate_up(qw, qz, sc)
    assert torch.equal(qi[:, 0::2], qw[:, :N // 2]) and torch.equal(qi[:, 1::2], qw[:, N // 2:])
    assert torch.equal(si[:, 0::2], sc[:, :N // 2]) and torch.equal(si[:, 1::2], sc[:, N // 2:])
    nib = lambda z: ((z.unsqueeze(-1) >> torch.arange(0, 32, 4, dtype=torch.int32)) & 0xF).reshape(G, N)
    assert torch.equal(nib(zi)[:, 0::2], nib(qz)[:, :N // 2]) and torch.equal(nib(zi)[:, 1::2], nib(qz)[:, N // 2:])
    qb, zb, sb = ops.deinterleave_gate_up(qi, zi, si)
    assert torch.equal(qb, qw) and torch.equal(zb, qz) and torch.equal(sb, sc)


def test_no_cpu_fallback():
    from aphrodite_engine_amd import _custom_ops as ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gptq_gemm(torch.zeros(1, 64, dtype=torch.half),
                      torch.zeros(8, 16, dtype=torch.int32),
                      torch.zeros(1, 2, dtype=torch.int32),
                      torch.zeros(1, 16, dtype=torch.half), torch.empty(0), True, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.scaled_fp8_quant(torch.zeros(2, 8))
    # nothing under the product package imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "aphrodite_engine_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from aphrodite_engine_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.lib()


def test_gptq_config_and_weights():
    from aphrodite_engine_amd.quantization import get_quantization_config
    from aphrodite_engine_amd.quantization.gptq import ExllamaState, GPTQLinearMethod
    cfg = get_quantization_config("gptq").from_config(
        {"bits": 4, "group_size": 128, "desc_act": False})
    assert cfg.get_name() == "gptq" and int(cfg.pack_factor) == 8
    layer = torch.nn.Module()
    m = cfg.get_quant_method(layer, "")
    assert isinstance(m, GPTQLinearMethod)
    # the reference routes classes NAMED like its own v2-loader methods to param.load_*_weight(); ours use
    # plain Parameters + v1 metadata, so the class name must not collide (linear.py:28-44, 330-332)
    from aphrodite_engine_amd.quantization.awq import AWQLinearMethod
    from aphrodite_engine_amd.quantization.fp8 import Fp8LinearMethod
    v2_names = {"AWQLinearMethod", "AWQMarlinLinearMethod", "CompressedTensorsLinearMethod", "FBGEMMFp8LinearMethod",
                "Fp8LinearMethod", "GPTQLinearMethod", "GPTQMarlin24LinearMethod", "GPTQMarlinLinearMethod",
                "HQQMarlinMethod", "MarlinLinearMethod", "ModelOptFp8LinearMethod", "QQQLinearMethod"}
    for cls in (GPTQLinearMethod, AWQLinearMethod, Fp8LinearMethod):
        assert cls.__name__ not in v2_names
    m.create_weights(layer, 4096, [4096, 1024, 1024], 4096, 6144, torch.float16)
    assert layer.qweight.shape == (512, 6144) and layer.qweight.dtype == torch.int32
    assert layer.qzeros.shape == (32, 768) and layer.scales.shape == (32, 6144)
    assert layer.g_idx.shape == (4096, ) and layer.exllama_state == ExllamaState.UNINITIALIZED
    # row-parallel + act-order disables exllama (gptq.py:137-139)
    cfg2 = type(cfg)(4, 128, True)
    layer2 = torch.nn.Module()
    cfg2.get_quant_method(layer2, "").create_weights(layer2, 2048, [4096], 4096, 4096, torch.float16)
    assert layer2.exllama_state == ExllamaState.UNUSED
    with pytest.raises(ValueError):
        cfg.get_quant_method(layer, "").create_weights(torch.nn.Module(), 100, [64], 100, 64,
                                                      torch.float16)
    with pytest.raises(ValueError):
        get_quantization_config("nope")


def test_awq_and_fp8_configs():
    from aphrodite_engine_amd.quantization.awq import AWQConfig
    from aphrodite_engine_amd.quantization.fp8 import Fp8Config, requantize_with_max_scale
    a = AWQConfig.from_config({"w_bit": 4, "q_group_size": 128, "zero_point": True})
    layer = torch.nn.Module()
    a.get_quant_method(layer, "").create_weights(layer, 8192, [1280], 8192, 1280, torch.float16)
    assert layer.qweight.shape == (8192, 160) and layer.qzeros.shape == (64, 160)
    with pytest.raises(ValueError):
        AWQConfig(3, 128, True)
    f = Fp8Config.from_config({"quant_method": "fp8", "activation_scheme": "static",
                               "ignored_layers": ["lm_head"]})
    assert f.is_checkpoint_fp8_serialized and f.get_quant_method(layer, "lm_head") is None
    layer = torch.nn.Module()
    f.get_quant_method(layer, "x").create_weights(layer, 64, [32, 16], 64, 48, torch.bfloat16)
    assert layer.weight.dtype == torch.float8_e4m3fn and layer.weight_scale.shape == (2, )
    # requantize_with_max_scale (w8a8_utils.py:54-80) on CPU tensors
    w = (torch.randn(48, 64)).to(torch.float8_e4m3fn)
    s = torch.tensor([0.5, 2.0])
    w2 = w.clone()
    mx, w2 = requantize_with_max_scale(w2, s, [32, 16])
    assert float(mx) == 2.0
    ref0 = (w[:32].float() * 0.5 / 2.0).clamp(-448, 448).to(torch.float8_e4m3fn)
    assert torch.equal(w2[:32].view(torch.uint8), ref0.view(torch.uint8))
    assert torch.equal(w2[32:].view(torch.uint8), w[32:].view(torch.uint8))


def test_mp_linear_kernel_selection():
    from aphrodite_engine_amd.quantization.kernels import (MPLinearLayerConfig,
                                                           choose_mp_linear_kernel,
                                                           register_with_reference)
    from aphrodite_engine_amd.quantization.kernels.cdna4 import CDNA4LinearKernel
    from aphrodite_engine_amd.scalar_type import scalar_types
    c = MPLinearLayerConfig((4096, 4096), (4096, 4096), scalar_types.uint4b8, torch.float16,
                            128, False, False)
    assert choose_mp_linear_kernel(c) is CDNA4LinearKernel
    c8 = MPLinearLayerConfig((4096, 4096), (4096, 4096), scalar_types.uint8b128, torch.float16, 128, False, False)
    assert choose_mp_linear_kernel(c8) is CDNA4LinearKernel       # 8-bit symmetric: the wnx kernels behind the same seam
    for bad in (MPLinearLayerConfig((4096, 4096), (4096, 4096), scalar_types.uint4b8, torch.float16, 48, False, False),
                MPLinearLayerConfig((4096, 4096), (4096, 4096), scalar_types.uint8b128, torch.float16, 128, True, False),
                MPLinearLayerConfig((4096, 4096), (4096, 4096), scalar_types.uint4b8, torch.float32, 128, False, False)):
        with pytest.raises(ValueError):
            choose_mp_linear_kernel(bad)
    with pytest.raises(ValueError):
        choose_mp_linear_kernel(c, compute_capability=90)   # not gfx950
    lst = ["Machete", "Marlin"]
    register_with_reference(lst)
    register_with_reference(lst)                             # idempotent
    assert lst[0] is CDNA4LinearKernel and len(lst) == 3
    assert scalar_types.uint4b8.min() == -8 and scalar_types.uint4b8.max() == 7


def test_attention_host_logic():
    from aphrodite_engine_amd.attention import MI355XAttentionBackend, PagedAttention
    from aphrodite_engine_amd.model import make_decode_metadata
    shape = MI355XAttentionBackend.get_kv_cache_shape(10, 16, 8, 128)
    assert shape == (2, 10, 16 * 8 * 128)
    kv = torch.zeros(shape, dtype=torch.float16)
    k, v = PagedAttention.split_kv_cache(kv, 8, 128)
    assert k.shape == (10, 8, 16, 16, 8) and v.shape == (10, 8, 128, 16)
    kv8 = torch.zeros(shape, dtype=torch.uint8)
    k8, _ = PagedAttention.split_kv_cache(kv8, 8, 128)
    assert k8.shape == (10, 8, 8, 16, 16)
    meta, pos, nblocks = make_decode_metadata(4, [5, 16, 17, 40], 16, "cpu")
    assert meta.decode_metadata.num_decode_tokens == 4 and meta.prefill_metadata is None
    assert pos.tolist() == [4, 15, 16, 39] and nblocks == 12
    bt = meta.block_tables
    assert sorted(bt.flatten().tolist()) == list(range(12))        # random permutation
    for i, L in enumerate([5, 16, 17, 40]):
        assert int(meta.slot_mapping[i]) == int(bt[i, (L - 1) // 16]) * 16 + (L - 1) % 16
    with pytest.raises(ValueError):
        MI355XAttentionBackend.get_impl_cls()(8, 72, 1.0, 2)          # unsupported head size


def _tp_worker(rank, world, port, q):
    import torch.distributed as dist
    from aphrodite_engine_amd import distributed as d
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)
    d.init_tensor_parallel(world, backend="gloo")
    x = torch.full((3, 4), float(rank + 1))
    y = d.tensor_model_parallel_all_reduce(x.clone())
    g = d.tensor_model_parallel_all_gather(torch.full((2, 3), float(rank)), dim=-1)
    q.put((rank, d.get_tensor_model_parallel_rank(), y.tolist(), g.tolist()))
    dist.destroy_process_group()


def _tp_selfcheck_worker(rank, world, port, q):
    """the first-contact self-check and the measured overlap gate, on a real (gloo) group of two."""
    import torch.distributed as dist
    from aphrodite_engine_amd import distributed as d
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    d.init_tensor_parallel(world, backend="gloo")
    chk = d.all_reduce_self_check(torch.device("cpu"), sizes=(4096, 65536))
    # the overlap arm needs HIP streams: stand in for it, the DECISION logic is what runs here (both ranks must agree: rank 1
    # measures the overlap arm slower than rank 0 does -- the max over the group decides)
    state = {"on": False}
    d.enable_all_reduce_overlap = lambda device, enabled=True: state.__setitem__("on", bool(enabled))
    times = iter([1.0, 0.90 if rank == 0 else 0.99])
    rec = d.choose_all_reduce_overlap(torch.device("cpu"), lambda: next(times))
    times2 = iter([1.0, 0.80])
    rec2 = d.choose_all_reduce_overlap(torch.device("cpu"), lambda: next(times2))
    q.put((rank, chk, rec, rec2, state["on"]))
    dist.destroy_process_group()


def test_all_reduce_self_check_and_measured_overlap_gate_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_tp_selfcheck_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    for rank, chk, rec, rec2, on in res:
        assert chk["decision"] == "RCCL (no peer-access communicator)"                 # no GPU here: the report says who serves
        assert chk["4096"]["rccl"] == "ok" and chk["4096"]["peer"].startswith("off")
        assert rec["enabled"] is False and rec["overlap_s"] == 0.99 and rec["serial_s"] == 1.0     # max over ranks: inside the margin
        assert rec2["enabled"] is True and on is True


def test_tensor_parallel_collectives_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_tp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    for rank, tp_rank, y, g in res:
        assert rank == tp_rank
        assert y == [[3.0] * 4] * 3                               # 1 + 2
        assert g == [[0.0, 0.0, 0.0, 1.0, 1.0, 1.0]] * 2          # concatenated on dim -1


def test_torch_library_registration():
    """The reference's op schemas resolve to our implementations
    (aphrodite_engine_amd/torch_ops.py); registered under private namespaces here
    so the test cannot clash with a real aphrodite._C."""
    from aphrodite_engine_amd import torch_ops
    torch_ops._REGISTERED = False
    torch_ops.register("_aphro_t_C", "_aphro_t_cache", "_aphro_t_rocm", "_aphro_t_moe")
    for name in ("paged_attention_v1", "paged_attention_v2", "gptq_gemm", "gptq_shuffle", "awq_gemm",
                 "awq_dequantize", "static_scaled_fp8_quant", "dynamic_scaled_fp8_quant",
                 "dynamic_per_token_scaled_fp8_quant", "cutlass_scaled_mm", "rms_norm",
                 "fused_add_rms_norm", "silu_and_mul", "rotary_embedding", "gptq_marlin_gemm", "gptq_marlin_repack",
                 "awq_marlin_repack", "fp8_marlin_gemm"):
        assert hasattr(torch.ops._aphro_t_C, name), name
    assert hasattr(torch.ops._aphro_t_cache, "reshape_and_cache")
    assert hasattr(torch.ops._aphro_t_rocm, "paged_attention")
    assert torch.ops._aphro_t_C.cutlass_scaled_mm_supports_fp8(95) is True
    # CPU tensors have no kernel registered: dispatch fails loudly, no fallback
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops._aphro_t_C.gptq_gemm(torch.zeros(1, 64, dtype=torch.half),
                                       torch.zeros(8, 16, dtype=torch.int32),
                                       torch.zeros(1, 2, dtype=torch.int32),
                                       torch.zeros(1, 16, dtype=torch.half),
                                       torch.empty(0, dtype=torch.int32), True, 4)


def test_cpp_torch_library_registration():
    """INTEGRATION.md option 3 as real code: csrc_torch/torch_bindings.cpp registers ops from C++ (TORCH_LIBRARY_FRAGMENT)
    with the same schemas as the Python registration; it loads without a GPU and has no CPU kernels."""
    from aphrodite_engine_amd import torch_cpp, torch_ops
    torch_cpp.load()
    torch_ops.register("_aphro_t_C", "_aphro_t_cache", "_aphro_t_rocm", "_aphro_t_moe")         # idempotent (registered above)
    py_ns = [ns for ns in ("_aphro_t_C", "_aphro_g_C", "_C") if hasattr(torch.ops, ns) and hasattr(getattr(torch.ops, ns), "gptq_gemm")]
    strip = lambda sch: str(sch).split("::", 1)[1]
    for name in ("gptq_gemm", "paged_attention_v1", "cutlass_scaled_mm",
                 # round 4: the memory-bound ops, the partitioned attention, shuffles / dequant, FP8 quantisers, advance_step
                 "paged_attention_v2", "gptq_shuffle", "awq_dequantize", "rms_norm", "fused_add_rms_norm", "silu_and_mul",
                 "rotary_embedding", "static_scaled_fp8_quant", "dynamic_per_token_scaled_fp8_quant", "advance_step_flashattn"):
        cpp = getattr(torch.ops._C_mi355x, name).default._schema
        assert py_ns, "python registration missing"
        assert strip(cpp) == strip(getattr(getattr(torch.ops, py_ns[0]), name).default._schema), name
    assert "reshape_and_cache(Tensor key, Tensor value" in str(torch.ops._C_mi355x_cache_ops.reshape_and_cache.default._schema)
    cache_ns = [ns for ns in ("_aphro_t_cache", "_aphro_g_cache", "_C_cache_ops") if hasattr(torch.ops, ns)
                and hasattr(getattr(torch.ops, ns), "convert_fp8")]
    # round 5: block copies / swaps, the whole-tensor quantiser, the MoE routing ops
    for name in ("dynamic_scaled_fp8_quant", "moe_align_block_size"):
        assert strip(getattr(torch.ops._C_mi355x, name).default._schema) == strip(getattr(getattr(torch.ops, py_ns[0]), name).default._schema), name
    moe_ns = [ns for ns in ("_aphro_t_moe", "_aphro_g_moe", "_moe_C") if hasattr(torch.ops, ns) and hasattr(getattr(torch.ops, ns), "topk_softmax")]
    assert moe_ns and strip(torch.ops._C_mi355x_moe.topk_softmax.default._schema) == strip(getattr(torch.ops, moe_ns[0]).topk_softmax.default._schema)
    for name in ("reshape_and_cache", "reshape_and_cache_flash", "convert_fp8", "swap_blocks", "copy_blocks"):
        cpp = getattr(torch.ops._C_mi355x_cache_ops, name).default._schema
        assert cache_ns and strip(cpp) == strip(getattr(getattr(torch.ops, cache_ns[0]), name).default._schema), name
    # round 5, second batch: the AWQ GEMM, the fp8 capability query, _rocm_C::paged_attention, the _C_custom_ar ops whose
    # arguments cross the dispatcher unchanged
    for name in ("awq_gemm", "cutlass_scaled_mm_supports_fp8", "gptq_marlin_repack", "awq_marlin_repack", "gptq_marlin_gemm",
                 "fp8_marlin_gemm"):
        assert strip(getattr(torch.ops._C_mi355x, name).default._schema) == strip(getattr(getattr(torch.ops, py_ns[0]), name).default._schema), name
    assert torch.ops._C_mi355x.cutlass_scaled_mm_supports_fp8(94) is True           # no tensor arguments: runs without a GPU
    rocm_ns = [ns for ns in ("_aphro_t_rocm", "_aphro_g_rocm", "_rocm_C") if hasattr(torch.ops, ns) and hasattr(getattr(torch.ops, ns), "paged_attention")]
    assert rocm_ns and strip(torch.ops._rocm_C_mi355x.paged_attention.default._schema) == strip(getattr(torch.ops, rocm_ns[0]).paged_attention.default._schema)
    from aphrodite_engine_amd import _custom_ops as ops
    assert torch.ops._C_mi355x_custom_ar.meta_size() == ops.meta_size() > 0
    ar_ns = [ns for ns in ("_aphro_t_C_custom_ar", "_aphro_g_C_custom_ar", "_C_custom_ar") if hasattr(torch.ops, ns)
             and hasattr(getattr(torch.ops, ns), "all_reduce_reg")]
    for name in ("all_reduce_reg", "all_reduce_unreg", "meta_size", "init_custom_ar", "dispose", "register_buffer",
                 "get_graph_buffer_ipc_meta", "register_graph_buffers"):
        assert ar_ns and strip(getattr(torch.ops._C_mi355x_custom_ar, name).default._schema) == \
            strip(getattr(getattr(torch.ops, ar_ns[0]), name).default._schema), name
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops._C_mi355x.gptq_gemm(torch.zeros(1, 64, dtype=torch.half), torch.zeros(8, 16, dtype=torch.int32),
                                      torch.zeros(1, 2, dtype=torch.int32), torch.zeros(1, 16, dtype=torch.half),
                                      torch.empty(0, dtype=torch.int32), True, 4)


def test_quant_configs_dispatch_on_layer_family():
    """The reference asks every Attention / FusedMoE / embedding layer for a quant method too (attention/
    layer.py:60-75 asserts a BaseKVCacheMethod): a config must answer by layer family, not always with a
    linear method."""
    from aphrodite_engine_amd.quantization.awq import AWQConfig
    from aphrodite_engine_amd.quantization.compressed_tensors import CompressedTensorsConfig
    from aphrodite_engine_amd.quantization.fp8 import Fp8Config
    from aphrodite_engine_amd.quantization.gptq import GPTQConfig
    from aphrodite_engine_amd.quantization.kv_cache import BaseKVCacheMethod
    mk = lambda name: type(name, (torch.nn.Module, ), {})()        # noqa: E731
    attn, lin, emb, head = mk("Attention"), mk("ColumnParallelLinear"), mk("VocabParallelEmbedding"), mk("ParallelLMHead")
    gptq, awq = GPTQConfig(4, 128, False), AWQConfig(4, 128, True)
    fp8 = Fp8Config(True, "dynamic", ignored_layers=["lm_head"])
    ct = CompressedTensorsConfig.from_config({"format": "float-quantized", "ignore": ["lm_head"], "config_groups": {
        "g": {"targets": ["Linear"], "weights": {"num_bits": 8, "type": "float", "strategy": "channel"},
              "input_activations": {"num_bits": 8, "type": "float", "strategy": "token", "dynamic": True}}}})
    for cfg in (gptq, awq):
        assert cfg.get_quant_method(attn, "model.layers.0.self_attn.attn") is None
        assert cfg.get_quant_method(emb, "model.embed_tokens") is None
        assert cfg.get_quant_method(head, "lm_head") is None
        assert cfg.get_quant_method(lin, "model.layers.0.mlp.down_proj") is not None
    assert GPTQConfig(4, 128, False, lm_head_quantized=True).get_quant_method(head, "lm_head") is not None
    for cfg in (fp8, ct):
        assert isinstance(cfg.get_quant_method(attn, "model.layers.0.self_attn.attn"), BaseKVCacheMethod)
        assert cfg.get_quant_method(emb, "model.embed_tokens") is None
        assert cfg.get_quant_method(lin, "model.layers.0.mlp.down_proj") is not None
        assert cfg.get_quant_method(lin, "lm_head") is None          # ignored layer, reference not installed here
    # experts: FP8 checkpoints get Fp8MoEMethod (fp8.py:86-87); compressed-tensors experts CompressedTensorsMoEMethod
    # (compressed_tensors.py:77-78), which serves pack-quantized symmetric int4 like the reference's (and refuses the rest)
    from aphrodite_engine_amd.moe import CompressedTensorsMoEMethod, Fp8MoEMethod
    assert isinstance(fp8.get_quant_method(mk("FusedMoE"), "model.layers.0.block_sparse_moe.experts"), Fp8MoEMethod)
    with pytest.raises(ValueError, match="only pack-quantized is supported"):        # the float-quantized config above
        ct.get_quant_method(mk("FusedMoE"), "model.layers.0.block_sparse_moe.experts")

    def ct_w(bits=4, strategy="group", symmetric=True, actorder=None):
        w = {"num_bits": bits, "type": "int", "symmetric": symmetric, "strategy": strategy, "actorder": actorder}
        if strategy == "group":
            w["group_size"] = 128
        return CompressedTensorsConfig.from_config({"format": "pack-quantized", "config_groups": {
            "g": {"targets": ["Linear"], "weights": w, "input_activations": None}}})
    for kw in ({}, {"strategy": "channel"}):
        mth = ct_w(**kw).get_quant_method(mk("FusedMoE"), "model.layers.0.block_sparse_moe.experts")
        assert isinstance(mth, CompressedTensorsMoEMethod) and mth.layout == "gptq" and mth.group_size == 128
    with pytest.raises(ValueError, match="Only symmetric quantization is supported for MoE"):
        ct_w(symmetric=False).get_quant_method(mk("FusedMoE"), "x.experts")
    with pytest.raises(NotImplementedError):
        ct_w(bits=8).get_quant_method(mk("FusedMoE"), "x.experts")
    with pytest.raises(NotImplementedError):
        ct_w(actorder="group").get_quant_method(mk("FusedMoE"), "x.experts")
    # the KV-cache method registers k_scale / v_scale and resolves them after loading
    attn.kv_cache_dtype = "fp8"
    m = fp8.get_quant_method(attn, "x")
    m.create_weights(attn)
    attn.k_scale.data.fill_(0.02)
    attn.v_scale.data.fill_(0.03)
    m.process_weights_after_loading(attn)
    assert (attn._k_scale, attn._v_scale) == (pytest.approx(0.02), pytest.approx(0.03)) and not hasattr(attn, "k_scale")


# ---- AttentionState / AttentionMetadataBuilder (a13): pinned by the reference's own CommonMetadataBuilder /
# CommonAttentionState run on the same scenarios (tests/golden/make_golden_attn_builder.py) ---------------------------
def _attn_golden():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "attn_builder.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", ["decode_eager", "decode_graph_pad", "prefill_only", "chunked_mixed",
                                  "prefix_cache_hit", "sliding_window", "profile_run"])
def test_attention_metadata_builder_matches_reference(name):
    import attn_builder_cases as cases
    from aphrodite_engine_amd.attention.backend import MI355XAttentionBackend
    sc = cases.scenarios()[name]
    ib = cases.make_input_builder(sc)
    builder = MI355XAttentionBackend.make_metadata_builder(ib)
    assert isinstance(builder, MI355XAttentionBackend.get_builder_cls())
    meta = builder.build(sc["seq_lens"], sc["query_lens"], sc["pad"], sc["batch"])
    assert cases.to_plain(meta) == _attn_golden()[name]
    # dtypes of the tensors the kernels read (utils.py:229-247)
    assert meta.slot_mapping.dtype == torch.long and meta.seq_lens_tensor.dtype == torch.int32
    assert meta.block_tables.dtype == torch.int32 and meta.query_start_loc.dtype == torch.int32
    # host-side context knowledge for the sync-free prefill dispatch
    prefill_ctx = [c for g in sc["groups"] if g.is_prompt for c in g.context_lens]
    assert meta.max_context_len == max(prefill_ctx, default=0)


def test_attention_state_matches_reference():
    import attn_builder_cases as cases
    from types import SimpleNamespace
    from aphrodite_engine_amd.attention.backend import MI355XAttentionBackend
    runner = SimpleNamespace(device="cpu", graph_block_tables=np.arange(48, dtype=np.int32).reshape(8, 6),
                             max_seq_len_to_capture=96, attn_backend=MI355XAttentionBackend)
    st = MI355XAttentionBackend.get_state_cls()(runner)
    golden = _attn_golden()
    with st.graph_capture(8):
        clone = st.graph_clone(4)
        assert type(clone) is type(st) and clone.runner is runner
        m = st.graph_capture_get_metadata_for_batch(4)
        assert cases.to_plain(m) == golden["state_capture_batch4"]
        bufs = st.get_graph_input_buffers(m)
        assert sorted(bufs) == golden["state_buffer_keys"]
        # the buffers ARE the persistent capture tensors: refreshing them is what a replay reads
        assert bufs["seq_lens_tensor"].data_ptr() == m.seq_lens_tensor.data_ptr()
        live = MI355XAttentionBackend.make_metadata(
            num_prefills=0, num_prefill_tokens=0, num_decode_tokens=4, slot_mapping=torch.tensor([5, 6, 7, -1]),
            seq_lens=None, seq_lens_tensor=torch.tensor([9, 17, 33, 1], dtype=torch.int32), max_query_len=None,
            max_prefill_seq_len=0, max_decode_seq_len=33, query_start_loc=None, seq_start_loc=None,
            context_lens_tensor=None, block_tables=torch.ones(4, 6, dtype=torch.int32), use_cuda_graph=True)
        st.prepare_graph_input_buffers(bufs, live)
        assert m.seq_lens_tensor.tolist() == [9, 17, 33, 1] and int(m.block_tables.sum()) == 24
    assert not st._is_graph_capturing and not hasattr(st, "_graph_seq_lens")
    st.begin_forward(None)


def test_fused_fp8_decode_path_selection():
    """Which FP8 checkpoints take the fused decode path (model.fused_decode_fp8_ok, host logic only): compressed-tensors
    W8A8 with per-token dynamic OR static per-tensor activation scales (compressed_tensors_w8a8_fp8.py:98-148), Fp8Config
    with the static scheme (fp8.py:150-199) -- not its dynamic one (ONE scale over the whole tensor needs a cross-workgroup
    absmax), not weight-only (Marlin role), not a layer whose projections disagree on the scheme."""
    from aphrodite_engine_amd import model as M
    from aphrodite_engine_amd.quantization.fp8 import CompressedTensorsW8A8Fp8Config, Fp8Config

    def build(qc):
        m = M.LlamaForCausalLM(M.TINY, qc, torch.float16, "auto")
        for layer in m.layers:
            for lin in layer.linears():
                lin.quant_method.process_weights_after_loading(lin)
        return m

    for qc, want in ((CompressedTensorsW8A8Fp8Config("channel", False), True),
                     (CompressedTensorsW8A8Fp8Config("channel", True), True),
                     (CompressedTensorsW8A8Fp8Config("tensor", True), True),
                     (Fp8Config(True, "static"), True),
                     (Fp8Config(True, "dynamic"), False)):
        m = build(qc)
        assert [layer.fused_decode_fp8_ok(5) for layer in m.layers] == [want] * len(m.layers), type(qc).__name__
        assert not any(layer.fused_decode_fp8_ok(65) for layer in m.layers)          # decode-sized batches only
    m = build(CompressedTensorsW8A8Fp8Config("channel", True))
    assert all(lin.input_scale.shape == (1, ) and lin.input_scale.dtype == torch.float32
               for layer in m.layers for lin in layer.linears())
    m.layers[0].o_proj.input_scale = None                                             # schemes disagree inside a layer
    assert [layer.fused_decode_fp8_ok(5) for layer in m.layers] == [False] + [True] * (len(m.layers) - 1)
    # the library-side shape rules behind it are host logic too
    from aphrodite_engine_amd import _lib
    lib = _lib.lib()
    assert lib.aphro_fp8_gemm_stream_silu_supported(32, 28672, 4096) == 1
    assert lib.aphro_fp8_gemm_stream_silu_supported(33, 28672, 4096) == 0            # M <= 32
    assert lib.aphro_fp8_gemm_stream_silu_supported(32, 28672 + 16, 4096) == 0       # halves of 8-row groups
    assert lib.aphro_fp8_gemm_stream_silu_supported(32, 4096, 14336) == 0            # K beyond one workgroup's 8 x 8 segments


def test_custom_routing_function_is_called_like_the_reference(monkeypatch):
    """FusedMoE.select_experts (fused_moe/layer.py:400-430): a model's own routing function is called with
    hidden_states / gating_output / topk / renormalize keywords and its (weights, ids) go to moe_align_block_size --
    host wiring, checked with the alignment op stubbed out."""
    from aphrodite_engine_amd import moe
    seen = {}

    def routing(hidden_states, gating_output, topk, renormalize):
        seen.update(h=hidden_states.shape, g=gating_output.shape, k=topk, r=renormalize)
        w, ids = torch.topk(torch.softmax(gating_output.float(), -1), topk, dim=-1)
        return w.double(), ids                                  # (wrong dtypes on purpose: int64 ids, f64 weights)

    def fake_align(topk_ids, block_size, num_experts, want_inverse=False):
        assert topk_ids.dtype == torch.int32 and topk_ids.is_contiguous() and block_size == moe.MOE_BLOCK_M
        return ("sorted", "experts", "post") + (("inv", ) if want_inverse else ())
    monkeypatch.setattr(moe, "moe_align_block_size", fake_align)
    x, g = torch.randn(5, 64), torch.randn(5, 8)
    w, ids, s, e, p, inv = moe.route_and_align(x, g, 2, True, 8, want_inverse=True, custom_routing_function=routing)
    assert seen == {"h": x.shape, "g": g.shape, "k": 2, "r": True}
    assert w.dtype == torch.float32 and ids.dtype == torch.int32 and (s, e, p, inv) == ("sorted", "experts", "post", "inv")
    assert moe.route_and_align(x, g, 2, False, 8, custom_routing_function=routing)[5] is None


def test_route_and_align_falls_back_outside_the_one_launch_forms_limits(monkeypatch):
    """the one-workgroup route + align launcher serves <= 256 experts and (slots + 66 E + 1) int32 of LDS
    <= 64 KB; the Python guard must send every other call to fused_topk + moe_align_block_size instead of letting the
    launcher's check surface as a RuntimeError (E = 128 with top-8 stops at 991 tokens; E >= 249 is never served)."""
    from aphrodite_engine_amd import moe
    calls = []
    monkeypatch.setattr(moe.ops, "moe_route_align", lambda *a, **k: calls.append("fused") or ("w", "i", "s", "e", "p", None))
    monkeypatch.setattr(moe, "fused_topk", lambda h, g, k, r: calls.append("topk") or ("w", torch.zeros(1, dtype=torch.int32)))
    monkeypatch.setattr(moe, "moe_align_block_size", lambda ids, b, e, want_inverse=False: ("s", "e", "p"))

    def served(tokens, experts, topk):
        calls.clear()
        moe.route_and_align(torch.zeros(tokens, 8), torch.zeros(tokens, experts), topk, True, experts)
        return calls == ["fused"]
    assert served(32, 8, 2) and served(991, 128, 8) and served(1, 248, 8)
    assert not served(4, 248, 8)          # LDS: 66 x 248 + slots + 1 words
    assert not served(992, 128, 8)        # LDS
    assert not served(1024, 128, 8)
    assert not served(4, 249, 8)          # LDS at any token count
    assert not served(4, 300, 8)          # more than 256 experts
    assert not served(4, 4, 8)            # topk > experts


def test_attention_impl_and_metadata_take_the_reference_layers_calls(monkeypatch):
    """What the reference's Attention layer and multi-step runner call on a backend: ``impl.forward(..., attn_type=)``
    (attention/layer.py:99-106; non-decoder types refused as rocm_flash_attn.py:395-399) and
    ``attn_metadata.advance_step(model_input, sampled_token_ids, block_size, num_seqs, num_queries)``
    (rocm_flash_attn.py:185-229): host bookkeeping here, the device update through advance_step_flashattn (stubbed)."""
    import enum
    import inspect
    import types
    from aphrodite_engine_amd.attention import backend as B
    assert "attn_type" in inspect.signature(B.MI355XAttentionImpl.forward).parameters
    AttentionType = enum.Enum("AttentionType", ["DECODER", "ENCODER", "ENCODER_DECODER"])
    impl = B.MI355XAttentionImpl(4, 128, 0.1, 2)
    # the window as ROCmFlashAttentionImpl keeps it (rocm_flash_attn.py:321-322): a (left, right) pair for the prompt kernels
    assert impl.sliding_window == (-1, -1) and B.MI355XAttentionImpl(4, 128, 0.1, 2, sliding_window=4096).sliding_window == (4096, 4096)
    with pytest.raises(NotImplementedError):
        impl.forward(torch.zeros(1, 512), torch.zeros(1, 256), torch.zeros(1, 256), None, None, attn_type=AttentionType.ENCODER)
    calls = {}
    monkeypatch.setattr(B.ops, "advance_step_flashattn", lambda **kw: calls.update(kw))
    n_seqs, n_q = 4, 3                                        # a graph-padded batch: one padding row
    md = B.MI355XAttentionMetadata(
        num_prefills=0, num_prefill_tokens=0, num_decode_tokens=n_seqs, slot_mapping=torch.zeros(n_seqs, dtype=torch.int64),
        seq_lens=[5, 9, 2, 1], seq_lens_tensor=torch.tensor([5, 9, 2, 1], dtype=torch.int32), max_query_len=1,
        max_prefill_seq_len=0, max_decode_seq_len=9, query_start_loc=None, seq_start_loc=None, context_lens_tensor=None,
        block_tables=torch.zeros(n_seqs, 2, dtype=torch.int32), use_cuda_graph=True)
    mi = types.SimpleNamespace(input_tokens=torch.zeros(n_seqs, dtype=torch.int64),
                               input_positions=torch.zeros(n_seqs, dtype=torch.int64))
    sampled = torch.tensor([[7], [8], [9]])
    md.advance_step(mi, sampled, 16, n_seqs, n_q)
    assert md.seq_lens == [6, 10, 3, 1] and md.max_decode_seq_len == 10          # only the real queries move
    assert calls["num_seqs"] == n_seqs and calls["num_queries"] == n_q and calls["block_size"] == 16
    assert calls["sampled_token_ids"] is sampled and calls["seq_lens"] is md.seq_lens_tensor
    assert calls["input_tokens"] is mi.input_tokens and calls["slot_mapping"] is md.slot_mapping
    assert md.decode_metadata.max_decode_seq_len == 10
    md.use_cuda_graph = False
    with pytest.raises(ValueError):
        md.advance_step(mi, sampled, 16, n_seqs, n_q)                            # padding without a captured graph
    md2 = B.MI355XAttentionMetadata(**{**md.__dict__, "num_prefills": 1})
    with pytest.raises(ValueError):
        md2.advance_step(mi, sampled, 16, n_seqs, n_seqs)                        # not a decode-only batch


def test_int4_expert_methods_refuse_group_sizes_the_grouped_kernel_does_not_serve():
    from aphrodite_engine_amd.moe import Wna16MoEMethod
    for ok in (128, 256, 512):
        assert Wna16MoEMethod("gptq", ok).group_size == ok
    for bad in (32, 64, 192, 0):
        with pytest.raises(NotImplementedError):
            Wna16MoEMethod("gptq", bad)


# ---- bench.py --gpus 2: the driver's launch form, control flow only -----------------------------
def _run_bench_dry(extra_env, timeout=180):
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, APHRO_BENCH_DRY_CPU="1", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=root)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, (r.stdout[-1000:], r.stderr[-1000:])            # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_two_ranks_control_flow_over_gloo():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` exactly as the driver launches it, with a stub
    in place of the model (APHRO_BENCH_DRY_CPU): process group, barrier + max-over-ranks timing, the tensor-parallel section in
    child processes on their own rendezvous, one line from rank 0 with whole-job throughput."""
    line = _run_bench_dry({})
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["value"] > 0
    assert line["process_group"]["backend"] == "gloo" and line["process_group"]["world"] == 2
    assert line["tp_section"] == {"tp": 2, "backend": "gloo", "ranks_seen_by_collective": 2, "dry_run": True}


@pytest.mark.parametrize("failure", ["crash", "raise", "hang"])
def test_bench_replica_line_survives_a_failing_tensor_parallel_section(failure):
    """A tensor-parallel child that aborts, raises or hangs (killed at the budget) costs the tp_section, not the line."""
    line = _run_bench_dry({"APHRO_BENCH_DRY_TP_FAIL": failure, "APHRO_BENCH_TP_TIMEOUT_S": "8"})
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert "error" in line["tp_section"], line["tp_section"]
    if failure == "hang":
        assert "timed out" in line["tp_section"]["error"]


def test_per_file_compile_flags_have_one_source_of_truth():
    """The Makefile's per-file EXTRA flags (VGPR-form MFMA results, kernel-argument preload) are what the GPU suite validated;
    tools/kernel_resources.py recompiles every file to report registers / spills and must use the same ones."""
    import importlib.util
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mk = open(os.path.join(root, "aphrodite_engine_amd", "csrc", "Makefile")).read()
    var = {"VGPR_FORM": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "KERNARG_PRELOAD": ["-mllvm", "-amdgpu-kernarg-preload-count=14"],
           "PA_EXTRA": []}
    for name, flags in var.items():
        if flags:
            assert re.search(rf"^{name}\s*:=\s*{re.escape(' '.join(flags))}\s*$", mk, re.M), name
    want = {}
    for m in re.finditer(r"^\$\(BUILD\)/(\w+)\.o: EXTRA := (.*)$", mk, re.M):      # (BUILD = build, or build_lab under LAB=1)
        want[m.group(1) + ".hip"] = sum((var[v] for v in re.findall(r"\$\((\w+)\)", m.group(2))), [])
    assert set(want) >= {"paged_attention.hip", "wna16_gemm_resident.hip", "fp8_gemm_resident.hip", "fp8_gemm_stream.hip", "wna16_gemm.hip"}
    assert "kernarg-preload" not in re.search(r"^FLAGS\s*:=.*(?:\n\s+.*)*", mk, re.M).group(0)      # per file, not global
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(root, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    assert {k: v for k, v in kr.EXTRA.items()} == want


def test_deferred_all_reduce_falls_back_to_the_two_launches(monkeypatch):
    """DeferredAllReduce.finish when the fused all-reduce + norm launch stopped applying between defer and finish (communicator
    disabled after a peer timeout): the all-reduce, then ops.fused_add_rms_norm_pack -- the two launches the fused one stands
    for, same arguments -- instead of an assertion."""
    from aphrodite_engine_amd import _custom_ops as ops, distributed as d
    calls = []
    monkeypatch.setattr(d, "tensor_model_parallel_all_reduce_norm", lambda *a, **k: None)
    monkeypatch.setattr(d, "tensor_model_parallel_all_reduce", lambda x, *a, **k: calls.append(("ar", x)) or x + 1)

    def norm(x, slabs, residual, has_residual, weight, eps, pack=True, want_out=False):
        calls.append(("norm", x, slabs, residual, has_residual, weight, eps, pack, want_out))
        return "packed", "out"
    monkeypatch.setattr(ops, "fused_add_rms_norm_pack", norm)
    partial, res, w = torch.zeros(2, 8), torch.ones(2, 8), torch.ones(8)
    assert d.DeferredAllReduce(partial).finish(res, w, 1e-5, pack=False, want_out=True) == ("packed", "out")
    assert calls[0][0] == "ar" and calls[0][1] is partial
    kind, x, slabs, residual, has_residual, weight, eps, pack, want_out = calls[1]
    assert kind == "norm" and torch.equal(x, partial + 1) and slabs is None and residual is res and has_residual is True
    assert weight is w and eps == 1e-5 and pack is False and want_out is True



def test_environment_surface_is_one_table_read_once():
    """the C library's environment switches are read in ONE place (csrc/runtime.hip: read_knobs), never
    on a launch path; every other override is a lab override behind APHRO_LAB_ENV_INT (compiled in by `make LAB=1` only); no lab
    #ifdef branches in the product kernels; every switch -- C and Python side -- is documented in INTEGRATION.md."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "aphrodite_engine_amd", "csrc")
    names = set()
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        src = open(f).read()
        code = re.sub(r"//[^\n]*", "", src)
        if os.path.basename(f) == "runtime.hip":
            names |= set(re.findall(r'(?:env_i|env_l|getenv)\("([A-Z_0-9]+)"', code))
            continue
        if os.path.basename(f) == "common.h":
            code = code.replace("const char* e = getenv(name);", "")          # the one getenv of the LAB-only helper
        assert "getenv(" not in code, f"{os.path.basename(f)} reads the environment outside runtime.hip"
        for lab in ("PA_LAB", "FA_LAB", "FA4_LAB", "F8_LAB", "LG_LAB", "ABL_", "LMH_CONTIG", "LMH_KBLOCK", "FA_TRAIL"):
            assert not re.search(rf"#\s*if[a-z]*\s+.*{lab}", code), f"{os.path.basename(f)} still carries {lab} branches"
    assert 10 <= len(names) <= 15, sorted(names)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    for n in names:
        assert n in doc, f"{n} is read by the library but not documented in INTEGRATION.md"
    from aphrodite_engine_amd import switches
    with pytest.raises(KeyError):
        switches.switch("APHRO_NOT_A_SWITCH")
    pysrc = ""
    for f in glob.glob(os.path.join(root, "aphrodite_engine_amd", "**", "*.py"), recursive=True):
        if os.path.basename(f) not in ("switches.py", "_lib.py"):
            pysrc += open(f).read()
    used = set(re.findall(r'switch\("([A-Z_0-9]+)"\)', pysrc))
    assert used <= set(switches.SWITCHES), used - set(switches.SWITCHES)
    assert not re.findall(r'os\.environ\.get\("APHRO', pysrc)            # the package reads its switches through the registry only


def test_lab_patches_apply_to_the_product_sources():
    """tools/lab_patches/*.patch (the kernels' stamp / ablation branches, kept out of the product sources) still apply to the
    files they instrument -- a kernel edit that moves a hunk's context is caught here, not when a lab build is next needed."""
    import glob
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("patch") is None:
        pytest.skip("no patch(1) in this image")
    patches = sorted(glob.glob(os.path.join(root, "tools", "lab_patches", "*.patch")))
    assert patches
    for pf in patches:
        target = os.path.join(root, "aphrodite_engine_amd", "csrc", os.path.basename(pf)[:-len(".patch")])
        assert os.path.exists(target), target
        with open(pf) as f:
            r = subprocess.run(["patch", "--dry-run", "-s", "-p0", target], stdin=f, capture_output=True, text=True)
        assert r.returncode == 0, (os.path.basename(pf), r.stdout[-400:], r.stderr[-400:])
