"""ctypes binding of libaphrodite_mi355x.so (the C ABI in include/aphrodite_mi355x.h).

The product path has NO CPU fallback: if the HIP library is missing or a symbol
is absent this module raises at import/lookup time (the reference behaves the
same way -- `aphrodite/_custom_ops.py:13-43` re-raises with a hint when
`aphrodite._C` cannot be imported).
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get(
    "APHRODITE_MI355X_LIB", os.path.join(_HERE, "lib", "libaphrodite_mi355x.so"))

P, I, L, F, Z = c_void_p, c_int, c_int64, c_float, c_size_t

# name -> (restype, argtypes); mirrors include/aphrodite_mi355x.h one to one
SIGNATURES = {
    "aphro_last_error": (ctypes.c_char_p, []),
    "aphro_abi_version": (I, []),
    "aphro_gptq_shuffle": (I, [P, P, L, L, I, P, P]),
    "aphro_gptq_repack": (I, [P, P, P, L, L, I, P]),
    "aphro_gptq_gemm": (I, [P, P, P, P, P, P, P, P, Z, L, L, L, L, L, I, I, P]),
    "aphro_wna16_workspace_bytes": (Z, [L, L, L]),
    "aphro_wna16_packed_a_bytes": (Z, [L, L]),
    "aphro_wna16_pack_a": (I, [P, P, P, L, L, L, I, P]),
    "aphro_wna16_ksplit": (I, [L, L, L, L]),
    "aphro_wna16_gemm_packed": (I, [P, P, P, P, P, P, Z, L, L, L, L, I, I, P]),
    "aphro_wna16_gemm_silu_pack": (I, [P, P, P, P, P, L, L, L, L, I, I, P]),
    "aphro_paged_attention_packed": (I, [P, P, P, P, P, I, I, I, I, F, P, P, I, I, I, P,
                                         L, L, L, I, I, F, F, P]),
    "aphro_paged_attention_rope_packed": (I, [P, P, P, I, P, P, P, P, P, I, I, I, I, F, P, P, I, I, I, P,
                                              L, L, I, I, F, F, P]),
    "aphro_fp8_gemm_ksplit": (I, [L, L, L]),
    "aphro_scaled_mm_fp8_slabs": (I, [P, P, P, Z, L, L, L, P]),
    "aphro_fused_add_rms_norm_quant_fp8": (I, [P, P, I, P, P, I, I, P, I, P, F, P, P, P, L, I, I, P]),
    "aphro_silu_and_mul_quant_fp8": (I, [P, P, P, P, L, I, I, P]),
    "aphro_fused_add_rms_norm_quant_fp8_static": (I, [P, P, I, P, P, I, I, P, I, P, F, P, P, P, L, I, I, P, P]),
    "aphro_silu_and_mul_quant_fp8_static": (I, [P, P, P, P, L, I, I, P, P]),
    "aphro_paged_attention_rope_packed_scaled": (I, [P, P, P, I, P, P, P, P, P, P, P, I, I, I, I, F, P, P,
                                                     I, I, I, P, L, L, I, I, F, F, P]),
    "aphro_paged_attention_rope_scaled_q8": (I, [P, P, P, P, I, P, P, P, P, P, P, P, I, I, I, I, F, P, P,
                                                 I, I, I, P, L, L, I, I, F, F, P]),
    "aphro_paged_attention_rope_scaled_absmax": (I, [P, P, P, P, I, P, P, P, P, P, P, P, I, I, I, I, F, P, P,
                                                     I, I, I, P, L, L, I, I, F, F, P]),
    "aphro_sample_top_k_top_p": (I, [P, P, L, P, P, P, P, P, L, P, P, L, L, I, P]),
    "aphro_custom_ar_meta_size": (L, []),
    "aphro_ipc_handle_bytes": (I, []),
    "aphro_custom_ar_alloc_shared": (I, [P, Z]),
    "aphro_custom_ar_free_shared": (I, [P]),
    "aphro_ipc_get_mem_handle": (I, [P, P, P]),
    "aphro_custom_ar_init": (I, [P, P, P, P, P, Z, P, P, P, Z, I, I]),
    "aphro_custom_ar_dispose": (I, [P]),
    "aphro_custom_ar_register_buffer": (I, [P, P, P, P]),
    "aphro_custom_ar_should_one_shot": (I, [I, Z]),
    "aphro_custom_ar_all_reduce": (I, [P, P, P, L, I, P, Z, P]),
    "aphro_custom_ar_get_graph_buffer_ipc_meta": (I, [P, P, P, I, P]),
    "aphro_custom_ar_register_graph_buffers": (I, [P, P, P, I]),
    "aphro_custom_ar_error": (I, [P]),
    "aphro_custom_ar_fused_norm_one_shot": (I, [I, L, I, I]),
    "aphro_custom_ar_fused_add_rms_norm": (I, [P, P, P, I, P, F, P, P, L, I, I, P, Z, P, Z, P]),
    "aphro_custom_ar_fused_add_rms_norm_quant_fp8": (I, [P, P, P, I, P, F, P, P, P, P, L, I, I, P, Z, P]),
    "aphro_custom_ar_fused_add_rms_norm_router": (I, [P, P, P, I, P, F, P, P, P, I, L, I, I, P, Z, P]),
    "aphro_custom_ar_init_loopback": (I, [P, P, P, Z, P, Z, I]),
    "aphro_advance_step_flashattn": (I, [I, I, I, P, P, P, P, P, P, L, P]),
    "aphro_argmax_rows": (I, [P, P, L, L, L, I, P]),
    "aphro_topk_softmax": (I, [P, P, P, P, L, I, I, P]),
    "aphro_moe_align_block_size": (I, [P, I, I, P, P, P, P, L, P]),
    "aphro_moe_gather_pack": (I, [P, P, P, P, L, L, L, L, I, I, P]),
    "aphro_wna16_grouped_ksplit": (I, [L, L, L, L]),
    "aphro_wna16_gemm_grouped": (I, [P, P, P, P, P, P, P, P, Z, P, L, L, L, L, I, I, P]),
    "aphro_moe_combine": (I, [P, P, I, L, P, P, L, I, L, I, P]),
    "aphro_fused_add_rms_norm_pack": (I, [P, P, I, P, I, P, F, P, P, L, I, I, P]),
    "aphro_silu_and_mul_pack_slabs": (I, [P, I, P, P, L, I, I, P]),
    "aphro_silu_and_mul_pack": (I, [P, P, P, L, I, I, P]),
    "aphro_rope_cache": (I, [P, L, P, I, P, P, I, I, P, P, P, P, L, I, I, I, I, I, I, I, F, F, P]),
    "aphro_gptq_dequant": (I, [P, P, P, P, P, L, L, L, I, I, I, P]),
    "aphro_awq_dequantize": (I, [P, P, P, P, L, L, L, I, P]),
    "aphro_awq_gemm_workspace_bytes": (Z, [L, L, L, L]),
    "aphro_awq_gemm": (I, [P, P, P, P, P, P, Z, L, L, L, L, L, I, P]),
    "aphro_awq_repack": (I, [P, P, L, L, P]),
    "aphro_awq_repack_zeros": (I, [P, P, L, L, P]),
    "aphro_reshape_and_cache": (I, [P, P, P, P, P, L, I, I, I, I, L, L, I, I, F, F, P]),
    "aphro_reshape_and_cache_flash": (I, [P, P, P, P, P, L, I, I, I, L, L, L, I, I, F, F, P]),
    "aphro_copy_blocks": (I, [P, P, I, P, L, L, P]),
    "aphro_swap_blocks": (I, [P, P, P, L, L, I, P]),
    "aphro_prefetch": (I, [P, Z, P]),
    "aphro_spin_us": (I, [ctypes.c_double, P]),
    "aphro_wna16_gemm_large_workspace_bytes": (Z, [L, L, L, L, I]),
    "aphro_wna16_gemm_large": (I, [P, P, P, P, P, P, Z, L, L, L, L, L, I, I, P]),
    "aphro_wna16_gemm_large_silu_supported": (I, [L, L, L, L]),
    "aphro_wna16_gemm_large_silu": (I, [P, P, P, P, P, P, Z, L, L, L, L, L, I, I, P]),
    "aphro_wna16_gemm_mid_supported": (I, [L, L, L, L]),
    "aphro_wna16_gemm_mid_workspace_bytes": (Z, [L, L, L, L]),
    "aphro_wna16_gemm_mid": (I, [P, P, P, P, P, P, Z, L, L, L, L, L, I, I, P]),
    "aphro_wna16_gemm_mid_ksplit": (I, [L, L, L, L]),
    "aphro_wna16_gemm_mid_packed": (I, [P, P, P, P, P, P, Z, P, L, L, L, L, I, I, P]),
    "aphro_wna16_resident_ksplit": (I, [L, L, L, L]),
    "aphro_wna16_gemm_resident": (I, [P, P, P, P, P, P, Z, P, L, L, L, L, I, I, I, P]),
    "aphro_wna16_strip_relayout": (I, [P, P, L, L, L, L, P]),
    "aphro_wna16_strip_unrelayout": (I, [P, P, L, L, L, L, P]),
    "aphro_wna16_strip_geometry": (I, [L, L, L, L, P]),
    "aphro_wna16_gemm_mid_packed_strip": (I, [P, P, P, P, P, P, Z, P, L, L, L, L, I, I, L, P]),
    "aphro_wna16_gemm_large_strip_workspace_bytes": (Z, [L, L, L, L, I, L]),
    "aphro_wna16_gemm_large_strip": (I, [P, P, P, P, P, P, Z, L, L, L, L, L, I, I, I, L, P]),
    "aphro_wna16_gemm_rowmajor_supported": (I, [L, L, L, L, I]),
    "aphro_lm_head_argmax_supported": (I, [L, L, L, L, I]),
    "aphro_fp8_moe_gemm": (I, [P, P, P, P, P, P, P, P, P, L, L, L, L, I, I, P]),
    "aphro_moe_route_align": (I, [P, P, P, L, P, P, P, P, L, I, I, I, I, I, P]),
    "aphro_moe_route_gather_supported": (I, [L, I, I, I, L]),
    "aphro_moe_route_gather": (I, [P, P, P, L, P, P, P, P, L, I, I, I, I, P, L, P, L, L, I, P]),
    "aphro_fused_add_rms_norm_router": (I, [P, P, I, P, I, P, F, P, P, P, I, L, I, I, P]),
    "aphro_fused_add_rms_norm_pack_combine": (I, [P, I, L, P, P, I, P, I, P, F, P, P, L, I, I, P]),
    "aphro_fp8_gemm_stream_ksplit": (I, [L, L, L]),
    "aphro_fp8_gemm_stream": (I, [P, L, P, P, P, P, P, P, Z, L, L, L, I, I, I, P]),
    "aphro_fp8_gemm_stream_silu_supported": (I, [L, L, L]),
    "aphro_fp8_gemm_resident_ksplit": (I, [L, L, L]),
    "aphro_fp8_strip_relayout": (I, [P, P, L, L, L, P]),
    "aphro_fp8_gemm_resident": (I, [P, L, P, P, P, P, P, P, Z, L, L, L, I, I, I, P]),
    "aphro_reload_env": (None, []),
    "aphro_fp8_gemm_resident_strips": (I, [L, L, L]),
    "aphro_fp8_strip_relayout_interleaved": (I, [P, P, L, L, L, P]),
    "aphro_fp8_gemm_resident_aq": (I, [P, L, I, P, I, P, P, P, P, P, P, Z, L, L, L, I, I, P]),
    "aphro_fp8_gemm_resident_silu": (I, [P, L, P, P, P, P, I, P, P, P, L, L, L, I, I, I, P]),
    "aphro_fp8_quant_rows_aq": (I, [P, P, I, P, P, L, L, I, P]),
    "aphro_fp8_gemm_stream_silu_quant": (I, [P, L, P, P, P, P, P, P, L, L, L, I, I, I, P]),
    "aphro_gptq_dequant_bits": (I, [P, P, P, P, P, L, L, L, I, I, P]),
    "aphro_gptq_gemm_bits_supported": (I, [L, L, L, L, I]),
    "aphro_gptq_gemm_bits": (I, [P, L, P, P, P, P, L, L, L, L, I, I, P]),
    "aphro_gptq_make_sequential_bits": (I, [P, P, P, L, L, I, P]),
    "aphro_lm_head_argmax": (I, [P, L, P, L, P, L, P, L, L, L, I, P]),
    "aphro_wna16_gemm_rowmajor": (I, [P, L, P, P, P, P, P, Z, L, L, L, L, I, I, I, P]),
    "aphro_convert_fp8": (I, [P, P, L, F, I, I, I, P]),
    "aphro_paged_attention": (I, [P, P, P, P, P, P, P, I, I, I, I, F, P, P, I, I, I, P,
                                  L, L, L, I, I, F, F, I, P]),
    "aphro_static_scaled_fp8_quant": (I, [P, P, P, L, L, I, P]),
    "aphro_dynamic_scaled_fp8_quant": (I, [P, P, P, L, L, I, P]),
    "aphro_dynamic_scaled_fp8_quant_ws": (I, [P, P, P, P, Z, L, L, I, P]),
    "aphro_dynamic_per_token_scaled_fp8_quant": (I, [P, P, P, P, L, L, I, P]),
    "aphro_scaled_mm_fp8": (I, [P, P, P, P, P, P, P, Z, L, L, L, I, I, I, P]),
    "aphro_scaled_mm_fp8_large_workspace_bytes": (Z, [L, L, L]),
    "aphro_fp8_w8a16_gemm_large_workspace_bytes": (Z, [L, L, L, I]),
    "aphro_fp8_w8a16_gemm_large": (I, [P, P, P, P, P, P, Z, L, L, L, L, I, I, P]),
    "aphro_scaled_mm_fp8_large": (I, [P, P, P, P, P, P, P, Z, L, L, L, I, I, I, P]),
    "aphro_fp8_w8a16_gemm": (I, [P, P, P, P, P, P, Z, L, L, L, L, I, I, P]),
    "aphro_fp8_gemm_workspace_bytes": (Z, [L, L, L]),
    "aphro_rms_norm": (I, [P, P, P, F, L, I, L, I, P]),
    "aphro_fused_add_rms_norm": (I, [P, P, P, F, L, I, I, P]),
    "aphro_silu_and_mul": (I, [P, P, L, I, I, P]),
    "aphro_silu_and_mul_interleaved": (I, [P, P, L, I, I, P]),
    "aphro_rotary_embedding": (I, [P, P, P, L, I, I, I, I, P, L, L, I, I, P]),
    "aphro_flash_attn_varlen": (I, [P, P, P, P, P, I, I, I, I, I, L, L, L, F, I, P, I, P]),
    "aphro_flash_attn_varlen_window": (I, [P, P, P, P, P, I, I, I, I, I, L, L, L, F, I, P, I, I, P]),
    "aphro_context_attention": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, L, L, L, L,
                                    F, F, F, P, I, I, I, P]),
    "aphro_context_attention_workspace_bytes": (Z, [L, I, I, I]),
    # out q k v k_cache v_cache block_tables q_start_loc seq_lens ctx_lens | batch max_query_len max_seq_len | total_kv_tokens |
    # max_blocks num_heads num_kv_heads head_size block_size x | 4 strides | scale k_scale v_scale | alibi | dtype kv_dtype |
    # workspace bytes stream   (round 3: sliding_window before dtype)
    "aphro_context_attention_gathered": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, L, I, I, I, I, I, I, L, L, L, L,
                                             F, F, F, P, I, I, I, P, Z, P]),
}

OK = 0
F16, BF16, F32 = 0, 1, 2
KV_AUTO, KV_FP8_E4M3, KV_FP8_E5M2 = 0, 1, 2

_lib = None


def lib():
    """Load the shared library (once).  Raises ImportError with a build hint."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the MI355X HIP library has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or "
            "`make -C aphrodite_engine_amd/csrc`). There is no CPU fallback.")
    l = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = l
    return l


def check(rc, what=""):
    if rc != OK:
        msg = lib().aphro_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what}: {msg} (code {rc})" if what else f"{msg} (code {rc})")
