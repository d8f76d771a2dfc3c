"""Registers the MI355X kernels as the torch.library ops the reference binds
(`kernels/torch_bindings.cpp`, `kernels/rocm/torch_bindings.cpp`): after
``import aphrodite_engine_amd.torch_ops`` (or the ``aphrodite.general_plugins``
entry point in ``plugin.py``) the reference's own ``aphrodite/_custom_ops.py``
resolves ``torch.ops._C.*`` / ``_C_cache_ops.*`` / ``_rocm_C.*`` to this library
without any change.  Schemas are the reference's, verbatim (file:line in the
table below); implementations are registered for the CUDA(HIP) dispatch key and
forward to ``aphrodite_engine_amd._custom_ops``.
"""
from typing import Optional

import torch

from . import _custom_ops as ops

_LIBS = []
_REGISTERED = False

# (namespace, schema, python impl)            reference torch_bindings.cpp line
_C_OPS = [
    ("paged_attention_v1(Tensor! out, Tensor query, Tensor key_cache, Tensor value_cache, int num_kv_heads, "
     "float scale, Tensor block_tables, Tensor seq_lens, int block_size, int max_seq_len, Tensor? alibi_slopes, "
     "str kv_cache_dtype, float k_scale, float v_scale, int tp_rank, int blocksparse_local_blocks, "
     "int blocksparse_vert_stride, int blocksparse_block_size, int blocksparse_head_sliding_step) -> ()",
     ops.paged_attention_v1),                                                              # :25-35
    ("paged_attention_v2(Tensor! out, Tensor! exp_sums, Tensor! max_logits, Tensor! tmp_out, Tensor query, "
     "Tensor key_cache, Tensor value_cache, int num_kv_heads, float scale, Tensor block_tables, Tensor seq_lens, "
     "int block_size, int max_seq_len, Tensor? alibi_slopes, str kv_cache_dtype, float k_scale, float v_scale, "
     "int tp_rank, int blocksparse_local_blocks, int blocksparse_vert_stride, int blocksparse_block_size, "
     "int blocksparse_head_sliding_step) -> ()", ops.paged_attention_v2),                  # :38-49
    ("gptq_gemm(Tensor a, Tensor b_q_weight, Tensor b_gptq_qzeros, Tensor b_gptq_scales, Tensor b_g_idx, "
     "bool use_exllama, int bit) -> Tensor", ops.gptq_gemm),                               # :357-361
    ("gptq_shuffle(Tensor! q_weight, Tensor q_perm, int bit) -> ()", ops.gptq_shuffle),    # :364-365
    ("awq_gemm(Tensor _in_feats, Tensor _kernel, Tensor _scaling_factors, Tensor _zeros, int split_k_iters) "
     "-> Tensor", ops.awq_gemm),                                                           # :142-145
    ("awq_dequantize(Tensor _kernel, Tensor _scaling_factors, Tensor _zeros, int split_k_iters, int thx, "
     "int thy) -> Tensor", ops.awq_dequantize),                                            # :148-151
    ("rms_norm(Tensor! out, Tensor input, Tensor weight, float epsilon) -> ()", ops.rms_norm),          # :101-105
    ("fused_add_rms_norm(Tensor! input, Tensor! residual, Tensor weight, float epsilon) -> ()",
     ops.fused_add_rms_norm),                                                              # :108-111
    ("silu_and_mul(Tensor! out, Tensor input) -> ()", ops.silu_and_mul),                   # :56-57
    ("rotary_embedding(Tensor positions, Tensor! query, Tensor! key, int head_size, Tensor cos_sin_cache, "
     "bool is_neox) -> ()", ops.rotary_embedding),                                         # :117-121
]


def _static_scaled_fp8_quant(out, input, scale):
    ops.scaled_fp8_quant(input, scale, out=out)                       # written in place: no staging copy


def _dynamic_scaled_fp8_quant(out, input, scale):
    ops.scaled_fp8_quant(input, out=out, scale_out=scale)


def _dynamic_per_token_scaled_fp8_quant(out, input, scale, scale_ub: Optional[torch.Tensor]):
    ops.scaled_fp8_quant(input, scale_ub=scale_ub, use_per_token_if_dynamic=True, out=out, scale_out=scale)


def _cutlass_scaled_mm(out, a, b, a_scales, b_scales, bias: Optional[torch.Tensor]):
    ops.cutlass_scaled_mm(a, b, a_scales, b_scales, out.dtype, bias, out=out)


_C_OPS += [
    ("static_scaled_fp8_quant(Tensor! out, Tensor input, Tensor scale) -> ()", _static_scaled_fp8_quant),   # :374-376
    ("dynamic_scaled_fp8_quant(Tensor! out, Tensor input, Tensor! scale) -> ()", _dynamic_scaled_fp8_quant),  # :379-382
    ("dynamic_per_token_scaled_fp8_quant(Tensor! out, Tensor input, Tensor! scale, Tensor? scale_ub) -> ()",
     _dynamic_per_token_scaled_fp8_quant),                                                 # :385-390
    ("cutlass_scaled_mm(Tensor! out, Tensor a, Tensor b, Tensor a_scales, Tensor b_scales, Tensor? bias) -> ()",
     _cutlass_scaled_mm),                                                                  # :235-239
    ("cutlass_scaled_mm_supports_fp8(int cuda_device_capability) -> bool",
     ops.cutlass_scaled_mm_supports_fp8),                                                  # :242-244
]

def _fp8_marlin_gemm(a, b_q_weight, b_scales, workspace, num_bits, size_m, size_n, size_k):
    return ops.fp8_marlin_gemm(a, b_q_weight, b_scales, workspace, num_bits, size_m, size_n, size_k)


def _gptq_marlin_gemm(a, b_q_weight, b_scales, b_zeros, g_idx, perm, workspace, b_q_type, size_m, size_n, size_k,
                      is_k_full, has_zp, use_fp32_reduce, is_zp_float):
    if isinstance(b_q_type, int):          # standalone schema: the type travels as its size in bits
        from .scalar_type import ScalarType
        b_q_type = ScalarType.uint(b_q_type, 0 if has_zp else 8)
    return ops.gptq_marlin_gemm(a, b_q_weight, b_scales, b_zeros, g_idx, perm, workspace, b_q_type, size_m, size_n,
                                size_k, is_k_full, has_zp, use_fp32_reduce, is_zp_float)


# torch_bindings.cpp:195-201.  The verbatim schema names the torchbind class ``_core_C.ScalarType``
# (kernels/core/torch_bindings.cpp:13), which exists once the reference's ``_core_C`` extension is loaded, or -- standalone
# (tests, bench) -- once this package's own registration of it is (csrc_torch/core_scalar_type.cpp,
# torch_cpp.ensure_scalar_type_class).  Only when neither library is there is the op defined with ``int b_q_type``
# (size in bits) in that position.
_MARLIN_GEMM_TAIL = ("int size_m, int size_n, int size_k, bool is_k_full, bool has_zp, bool use_fp32_reduce, "
                     "bool is_zp_float) -> Tensor")
_MARLIN_GEMM_HEAD = ("gptq_marlin_gemm(Tensor a, Tensor b_q_weight, Tensor b_scales, Tensor b_zeros, Tensor g_idx, "
                     "Tensor perm, Tensor workspace, ")
GPTQ_MARLIN_GEMM_SCHEMAS = (_MARLIN_GEMM_HEAD + "__torch__.torch.classes._core_C.ScalarType b_q_type, " + _MARLIN_GEMM_TAIL,
                            _MARLIN_GEMM_HEAD + "int b_q_type, " + _MARLIN_GEMM_TAIL)

_C_OPS += [
    ("gptq_marlin_repack(Tensor b_q_weight, Tensor perm, SymInt size_k, SymInt size_n, int num_bits) -> Tensor",
     ops.gptq_marlin_repack),                                                              # :204-208
    ("awq_marlin_repack(Tensor b_q_weight, SymInt size_k, SymInt size_n, int num_bits) -> Tensor",
     ops.awq_marlin_repack),                                                               # :211-215
    ("fp8_marlin_gemm(Tensor a, Tensor b_q_weight, Tensor b_scales, Tensor! workspace, int num_bits, "
     "int size_m, int size_n, int size_k) -> Tensor", _fp8_marlin_gemm),                   # :218-222
]

def _moe_align_block_size(topk_ids, num_experts, block_size, sorted_token_ids, experts_ids, num_tokens_post_pad):
    ops.moe_align_block_size(topk_ids, num_experts, block_size, sorted_token_ids, experts_ids, num_tokens_post_pad)


_C_OPS += [
    ("moe_align_block_size(Tensor topk_ids, int num_experts, int block_size, Tensor! sorted_token_ids, "
     "Tensor! experts_ids, Tensor! num_tokens_post_pad) -> ()", _moe_align_block_size),     # :394-399
]

_C_OPS += [
    ("advance_step_flashattn(int num_seqs, int num_queries, int block_size, Tensor! input_tokens, "
     "Tensor sampled_token_ids, Tensor! input_positions, Tensor! seq_lens, Tensor! slot_mapping, "
     "Tensor block_tables) -> ()", ops.advance_step_flashattn),                             # :77-82
]

_MOE_OPS = [
    ("topk_softmax(Tensor! topk_weights, Tensor! topk_indices, Tensor! token_expert_indices, "
     "Tensor gating_output) -> ()", ops.topk_softmax),                 # kernels/moe/torch_bindings.cpp:11-14
]

_CACHE_OPS = [
    ("reshape_and_cache(Tensor key, Tensor value, Tensor! key_cache, Tensor! value_cache, Tensor slot_mapping, "
     "str kv_cache_dtype, float k_scale, float v_scale) -> ()", ops.reshape_and_cache),    # :467-473
    ("convert_fp8(Tensor! dst_cache, Tensor src_cache, float scale, str kv_cache_dtype) -> ()",
     ops.convert_fp8),                                                                     # :487-490
    ("reshape_and_cache_flash(Tensor key, Tensor value, Tensor! key_cache, Tensor! value_cache, "
     "Tensor slot_mapping, str kv_cache_dtype, float k_scale, float v_scale) -> ()",
     ops.reshape_and_cache_flash),                                                         # :476-484
    ("copy_blocks(Tensor(a!)[] key_caches, Tensor[](b!) value_caches, Tensor block_mapping) -> ()",
     ops.copy_blocks),                                                                     # :461-464
    ("swap_blocks(Tensor src, Tensor! dst, Tensor block_mapping) -> ()", ops.swap_blocks),  # :456-458
]

_ROCM_OPS = [
    ("paged_attention(Tensor! out, Tensor exp_sums, Tensor max_logits, Tensor tmp_out, Tensor query, "
     "Tensor key_cache, Tensor value_cache, int num_kv_heads, float scale, Tensor block_tables, "
     "Tensor context_lens, int block_size, int max_context_len, Tensor? alibi_slopes, str kv_cache_dtype, "
     "float k_scale, float v_scale) -> ()", ops.paged_attention_rocm),   # kernels/rocm/torch_bindings.cpp:17-28
]


# _C_custom_ar (kernels/torch_bindings.cpp:506-536), schemas verbatim.  The tensor ops dispatch on the CUDA(HIP) key; the
# ops without tensor arguments (dispose, meta_size, get_graph_buffer_ipc_meta, register_graph_buffers -- bound from C++
# function signatures in the reference, :524-535) are CompositeExplicitAutograd.  IPC handles travel as ``str``: a
# Python-implemented op receives them UTF-8 DECODED, so the only forms that survive the dispatcher are hex (128 digits)
# and latin-1 text; callers that hold raw ``bytes`` (the reference's CustomAllreduce) call
# ``aphrodite_engine_amd._custom_ops`` directly, exactly as they call ``aphrodite._custom_ops`` today.
def _car_get_graph_buffer_ipc_meta(fa: int):
    blob, offsets = ops.get_graph_buffer_ipc_meta(fa)
    return list(blob), offsets          # std::vector<uint8_t> -> int[] (custom_all_reduce.cu:120-128)


CUSTOM_AR_SCHEMAS = [
    ("init_custom_ar(Tensor meta, Tensor rank_data, str[] handles, int[] offsets, int rank, bool full_nvlink) -> int",
     ops.init_custom_ar, "CUDA"),                                                          # :510-514
    ("all_reduce_reg(int fa, Tensor inp, Tensor! out) -> ()", ops.all_reduce_reg, "CUDA"),  # :516-517
    ("all_reduce_unreg(int fa, Tensor inp, Tensor reg_buffer, Tensor! out) -> ()", ops.all_reduce_unreg, "CUDA"),  # :519-522
    ("dispose(int fa) -> ()", ops.dispose, "CompositeExplicitAutograd"),                   # :524
    ("meta_size() -> int", ops.meta_size, "CompositeExplicitAutograd"),                    # :526
    ("register_buffer(int fa, Tensor t, str[] handles, int[] offsets) -> ()", ops.register_buffer, "CUDA"),  # :528-531
    ("get_graph_buffer_ipc_meta(int fa) -> (int[], int[])", _car_get_graph_buffer_ipc_meta,
     "CompositeExplicitAutograd"),                                                         # :533
    ("register_graph_buffers(int fa, str[] handles, int[][] offsets) -> ()", ops.register_graph_buffers,
     "CompositeExplicitAutograd"),                                                         # :535
]


def register(ns_c: str = "_C", ns_cache: str = "_C_cache_ops", ns_rocm: str = "_rocm_C",
             ns_moe: str = "_moe_C", ns_custom_ar: Optional[str] = None) -> None:
    """Idempotent.  Pass other namespaces to avoid clashing with an already
    loaded ``aphrodite._C`` (e.g. in A/B comparisons)."""
    global _REGISTERED
    if _REGISTERED:
        return
    for ns, table in ((ns_c, _C_OPS), (ns_cache, _CACHE_OPS), (ns_rocm, _ROCM_OPS), (ns_moe, _MOE_OPS)):
        lib = torch.library.Library(ns, "FRAGMENT")
        for schema, fn in table:
            name = schema.split("(", 1)[0]
            lib.define(schema)
            if name == "cutlass_scaled_mm_supports_fp8":
                lib.impl(name, fn, "CompositeExplicitAutograd")
            else:
                lib.impl(name, fn, "CUDA")
        if ns == ns_c:
            from . import torch_cpp
            torch_cpp.ensure_scalar_type_class()      # standalone: this package's own registration of _core_C.ScalarType
            try:
                lib.define(GPTQ_MARLIN_GEMM_SCHEMAS[0])
            except Exception:          # _core_C.ScalarType is not registered: standalone form
                lib.define(GPTQ_MARLIN_GEMM_SCHEMAS[1])
            lib.impl("gptq_marlin_gemm", _gptq_marlin_gemm, "CUDA")
        _LIBS.append(lib)
    lib = torch.library.Library(ns_custom_ar or ns_c + "_custom_ar", "FRAGMENT")
    for schema, fn, key in CUSTOM_AR_SCHEMAS:
        lib.define(schema)
        lib.impl(schema.split("(", 1)[0], fn, key)
    _LIBS.append(lib)
    _REGISTERED = True


def schema_arity(ns_c: str = "_C", ns_cache: str = "_C_cache_ops", ns_rocm: str = "_rocm_C", ns_moe: str = "_moe_C") -> dict:
    """{(namespace, op): number of schema arguments} of everything ``register`` defines -- what a caller that passes its
    arguments positionally (the reference's ``aphrodite/_custom_ops.py`` wrappers) must match
    (tests/test_reference_binding_cpu.py)."""
    out = {}
    for ns, table in ((ns_c, _C_OPS), (ns_cache, _CACHE_OPS), (ns_rocm, _ROCM_OPS), (ns_moe, _MOE_OPS)):
        for schema, _ in table:
            out[(ns, schema.split("(", 1)[0])] = len(torch._C.parse_schema(ns + "::" + schema).arguments)
    for schema in GPTQ_MARLIN_GEMM_SCHEMAS[1:2]:
        out[(ns_c, "gptq_marlin_gemm")] = len(torch._C.parse_schema(ns_c + "::" + schema).arguments)
    for schema, _, _ in CUSTOM_AR_SCHEMAS:
        out[(ns_c + "_custom_ar", schema.split("(", 1)[0])] = len(torch._C.parse_schema(ns_c + "_custom_ar::" + schema).arguments)
    return out
