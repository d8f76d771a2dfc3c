// C++ TORCH_LIBRARY registration of the MI355X hot-path kernels -- the reference's own registration style
// (kernels/torch_bindings.cpp) for maintainers who do not want Python in the dispatch path.  Thin: every op checks its
// arguments the way the reference's host function does, allocates what the schema says it returns, and forwards
// device pointers + the current HIP stream to the C ABI (include/aphrodite_mi355x.h).  No kernels in this file.
//
// The namespace is a build-time macro (APHRO_TORCH_NS, default `_C_mi355x`): built with -DAPHRO_TORCH_NS=_C it takes the
// place of the reference's extension; the default lets both be loaded side by side (A/B runs, the tests here).
// Schemas are the reference's, verbatim (kernels/torch_bindings.cpp line numbers on each def).
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/all.h>
#include <torch/library.h>

#include "aphrodite_mi355x.h"

#ifndef APHRO_TORCH_NS
#define APHRO_TORCH_NS _C_mi355x
#endif
#define APHRO_CONCAT_(a, b) a##b
#define APHRO_CONCAT(a, b) APHRO_CONCAT_(a, b)

namespace {

void* cur_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }

int act_dtype(const torch::Tensor& t) {
  TORCH_CHECK(t.scalar_type() == torch::kHalf || t.scalar_type() == torch::kBFloat16, "expected a float16 or bfloat16 tensor");
  return t.scalar_type() == torch::kHalf ? APHRO_F16 : APHRO_BF16;
}

int kv_dtype(const std::string& s) {
  if (s == "auto") return APHRO_KV_AUTO;
  if (s == "fp8" || s == "fp8_e4m3") return APHRO_KV_FP8_E4M3;
  if (s == "fp8_e5m2") return APHRO_KV_FP8_E5M2;
  TORCH_CHECK(false, "Unsupported data type of kv cache: ", s);
}

void ok(int rc, const char* op) { TORCH_CHECK(rc == APHRO_OK, op, ": ", aphro_last_error()); }

// gptq_gemm (q_gemm.cu:1824-1859): c [M, N] in a's dtype
torch::Tensor gptq_gemm(torch::Tensor a, torch::Tensor b_q_weight, torch::Tensor b_gptq_qzeros, torch::Tensor b_gptq_scales,
                        torch::Tensor b_g_idx, bool use_exllama, int64_t bit) {
  TORCH_CHECK(use_exllama && (bit == 2 || bit == 3 || bit == 4 || bit == 8),
              "gptq_gemm on MI355X: exllama-shuffled 2 / 3 / 4 / 8-bit weights (the non-exllama form is served by the Python op)");
  TORCH_CHECK(a.is_cuda() && a.dim() == 2 && a.stride(1) == 1, "gptq_gemm: a must be a row-major device matrix");
  const int64_t m = a.size(0), k = a.size(1), n = b_q_weight.size(1);
  auto out = torch::empty({m, n}, a.options());
  const bool act_order = b_g_idx.numel() > 0 && b_g_idx.device().is_cuda();
  const int64_t groups = b_gptq_scales.size(0);
  TORCH_CHECK(groups > 0 && k % groups == 0, "gptq_gemm: scales [groups, N] with groups dividing K (groups=", groups, ", K=", k, ")");
  if (bit != 4) {   // csrc/wnx_gemm.hip: MFMA small-M kernel on the sequential layout, else reconstruct + library GEMM
    torch::Tensor ag = act_order ? a.index_select(1, b_g_idx.to(torch::kLong)) : a;
    if (ag.stride(1) != 1 || ag.stride(0) % 8 != 0) ag = ag.contiguous();
    if (m >= 1 && m <= 64 && aphro_gptq_gemm_bits_supported(m < 32 ? m : 32, n, k, groups, (int)bit)) {
      for (int64_t m0 = 0; m0 < m; m0 += 32) {          // (as the Python op: up to 64 rows in two passes of the 32-row kernel)
        const int64_t rows = m - m0 < 32 ? m - m0 : 32;
        ok(aphro_gptq_gemm_bits((const char*)ag.data_ptr() + m0 * ag.stride(0) * ag.element_size(), ag.stride(0),
                                (const uint32_t*)b_q_weight.data_ptr(), (const uint32_t*)b_gptq_qzeros.data_ptr(),
                                b_gptq_scales.data_ptr(), (char*)out.data_ptr() + m0 * n * out.element_size(), rows, n, k,
                                groups, (int)bit, act_dtype(a), cur_stream()),
           "gptq_gemm");
      }
      return out;
    }
    auto w = torch::empty({k, n}, a.options());
    ok(aphro_gptq_dequant_bits((const uint32_t*)b_q_weight.data_ptr(), (const uint32_t*)b_gptq_qzeros.data_ptr(),
                               b_gptq_scales.data_ptr(), nullptr, w.data_ptr(), k, n, groups, (int)bit, act_dtype(a), cur_stream()),
       "gptq_gemm");
    return torch::matmul(ag, w);
  }
  if (m > 64 && n % 128 == 0 && k % 64 == 0 && k % groups == 0 && (k / groups) % 64 == 0) {
    // prefill-sized M: one MFMA kernel that dequantises in registers (the reference reconstructs + calls hipBLAS,
    // q_gemm.cu:1529-1544); act-order = gather the activation columns once
    torch::Tensor ag = act_order ? a.index_select(1, b_g_idx.to(torch::kLong)) : a;
    auto wsl = torch::empty({(int64_t)aphro_wna16_gemm_large_workspace_bytes(m, n, k, groups, act_dtype(a))},
                            a.options().dtype(torch::kUInt8));
    ok(aphro_wna16_gemm_large(ag.data_ptr(), (const uint32_t*)b_q_weight.data_ptr(), (const uint32_t*)b_gptq_qzeros.data_ptr(),
                              b_gptq_scales.data_ptr(), out.data_ptr(), wsl.data_ptr(), (size_t)wsl.numel(), m, n, k, groups,
                              ag.stride(0), 1, act_dtype(a), cur_stream()),
       "gptq_gemm");
    return out;
  }
  auto ws = torch::empty({(int64_t)aphro_wna16_workspace_bytes(m < 64 ? m : 64, n, k)}, a.options().dtype(torch::kUInt8));
  torch::Tensor tmp;
  if (act_order) tmp = torch::empty({m < 64 ? m : 64, k}, a.options());
  for (int64_t m0 = 0; m0 < m; m0 += 64) {            // the decode-sized kernel takes up to 64 rows per launch
    const int64_t rows = m - m0 < 64 ? m - m0 : 64;
    ok(aphro_gptq_gemm((const char*)a.data_ptr() + m0 * a.stride(0) * a.element_size(), (const uint32_t*)b_q_weight.data_ptr(),
                       (const uint32_t*)b_gptq_qzeros.data_ptr(), b_gptq_scales.data_ptr(),
                       act_order ? (const int32_t*)b_g_idx.data_ptr() : nullptr, act_order ? tmp.data_ptr() : nullptr,
                       (char*)out.data_ptr() + m0 * n * out.element_size(), ws.data_ptr(), (size_t)ws.numel(), rows, n, k,
                       groups, a.stride(0), 1, act_dtype(a), cur_stream()),
       "gptq_gemm");
  }
  return out;
}

// paged_attention_v1 (attention_kernels.cu:672-760)
void paged_attention_v1(torch::Tensor out, torch::Tensor query, torch::Tensor key_cache, torch::Tensor value_cache,
                        int64_t num_kv_heads, double scale, torch::Tensor block_tables, torch::Tensor seq_lens, int64_t block_size,
                        int64_t max_seq_len, const c10::optional<torch::Tensor>& alibi_slopes, std::string kv_cache_dtype,
                        double k_scale, double v_scale, int64_t tp_rank, int64_t blocksparse_local_blocks,
                        int64_t blocksparse_vert_stride, int64_t blocksparse_block_size, int64_t blocksparse_head_sliding_step) {
  TORCH_CHECK(blocksparse_vert_stride <= 1, "blocksparse attention is not implemented on MI355X");
  const int64_t num_seqs = query.size(0), num_heads = query.size(1), head_size = query.size(2);
  ok(aphro_paged_attention(out.data_ptr(), nullptr, nullptr, nullptr, query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
                           (int)num_seqs, (int)num_heads, (int)num_kv_heads, (int)head_size, (float)scale,
                           block_tables.data_ptr<int>(), seq_lens.data_ptr<int>(), (int)block_tables.size(1), (int)block_size,
                           (int)max_seq_len, alibi_slopes ? alibi_slopes->data_ptr<float>() : nullptr, query.stride(0),
                           key_cache.stride(0), key_cache.stride(1), act_dtype(query), kv_dtype(kv_cache_dtype), (float)k_scale,
                           (float)v_scale, 0, cur_stream()),
     "paged_attention_v1");
}

// reshape_and_cache (cache_kernels.cu:207-262)
void reshape_and_cache(torch::Tensor key, torch::Tensor value, torch::Tensor key_cache, torch::Tensor value_cache,
                       torch::Tensor slot_mapping, std::string kv_cache_dtype, double k_scale, double v_scale) {
  ok(aphro_reshape_and_cache(key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
                             slot_mapping.data_ptr<int64_t>(), key.size(0), key.size(1), key.size(2), key_cache.size(3),
                             key_cache.size(4), key.stride(0), value.stride(0), act_dtype(key), kv_dtype(kv_cache_dtype),
                             (float)k_scale, (float)v_scale, cur_stream()),
     "reshape_and_cache");
}

// cutlass_scaled_mm (scaled_mm_entry.cu:92-137): fp8 e4m3 only on gfx950
void cutlass_scaled_mm(torch::Tensor out, torch::Tensor a, torch::Tensor b, torch::Tensor a_scales, torch::Tensor b_scales,
                       const c10::optional<torch::Tensor>& bias) {
  TORCH_CHECK(a.dim() == 2 && b.dim() == 2 && out.dim() == 2 && out.size(0) == a.size(0) && a.size(1) == b.size(0) &&
              b.size(1) == out.size(1), "cutlass_scaled_mm: shape mismatch");
  TORCH_CHECK(a.stride(1) == 1 && out.stride(1) == 1 && b.stride(0) == 1, "cutlass_scaled_mm: a / out row-major, b column-major");
  TORCH_CHECK(a.scalar_type() == at::kFloat8_e4m3fn && b.scalar_type() == at::kFloat8_e4m3fn,
              "cutlass_scaled_mm on MI355X implements fp8 (e4m3fn) only");
  TORCH_CHECK(a_scales.scalar_type() == torch::kFloat && b_scales.scalar_type() == torch::kFloat, "scales must be float32");
  const int64_t m = a.size(0), k = a.size(1), n = b.size(1);
  TORCH_CHECK(a_scales.numel() == 1 || a_scales.numel() == m, "a_scales: 1 or M elements");
  TORCH_CHECK(b_scales.numel() == 1 || b_scales.numel() == n, "b_scales: 1 or N elements");
  if (bias) TORCH_CHECK(bias->scalar_type() == out.scalar_type() && bias->numel() == n && bias->is_contiguous(),
                        "cutlass_scaled_mm: bias must be a contiguous [N] tensor of the output dtype");
  // The kernels index rows as base + row * K / N.  The reference admits row-strided a / out (stride(0) % 16 == 0,
  // scaled_mm_entry.cu:104-110): serve those through a contiguous copy instead of reading the wrong rows.
  if (a.stride(0) != k) a = a.contiguous();
  TORCH_CHECK(b.stride(1) == k || n == 1, "cutlass_scaled_mm: b must be a dense column-major [K, N] matrix");
  if (out.stride(0) != n) {
    auto tmp = torch::empty({m, n}, out.options());
    cutlass_scaled_mm(tmp, a, b, a_scales, b_scales, bias);
    out.copy_(tmp);
    return;
  }
  const int odt = act_dtype(out);
  const void* bp = bias ? bias->data_ptr() : nullptr;
  const int a_tok = a_scales.numel() > 1, b_ch = b_scales.numel() > 1;
  if (m > 64 && n % 128 == 0 && k % 128 == 0) {         // prefill-sized: the MFMA-bound stream-K kernel
    auto ws = torch::empty({(int64_t)aphro_scaled_mm_fp8_large_workspace_bytes(m, n, k)}, a.options().dtype(torch::kUInt8));
    ok(aphro_scaled_mm_fp8_large(out.data_ptr(), a.data_ptr(), b.data_ptr(), a_scales.data_ptr<float>(), b_scales.data_ptr<float>(),
                                 bp, ws.data_ptr(), (size_t)ws.numel(), m, n, k, a_tok, b_ch, odt, cur_stream()),
       "cutlass_scaled_mm");
    return;
  }
  auto ws = torch::empty({(int64_t)aphro_fp8_gemm_workspace_bytes(m < 64 ? m : 64, n, k)}, a.options().dtype(torch::kUInt8));
  for (int64_t m0 = 0; m0 < m; m0 += 64) {
    const int64_t rows = m - m0 < 64 ? m - m0 : 64;
    ok(aphro_scaled_mm_fp8((char*)out.data_ptr() + m0 * n * out.element_size(), (const char*)a.data_ptr() + m0 * k, b.data_ptr(),
                           a_scales.data_ptr<float>() + (a_tok ? m0 : 0), b_scales.data_ptr<float>(), bp, ws.data_ptr(),
                           (size_t)ws.numel(), rows, n, k, a_tok, b_ch, odt, cur_stream()),
       "cutlass_scaled_mm");
  }
}

// ---- round 4: the memory-bound ops and the remaining cache / quantisation ops (one C-ABI call each) --------------------------
int64_t rows_of(const torch::Tensor& t) { return t.numel() / t.size(-1); }

// rms_norm (layernorm_kernels.cu:17-45, 160-198)
void rms_norm(torch::Tensor out, torch::Tensor input, torch::Tensor weight, double epsilon) {
  const int64_t hidden = input.size(-1);
  auto x = input.reshape({-1, hidden});
  TORCH_CHECK(out.is_contiguous() && x.stride(1) == 1, "rms_norm: contiguous rows expected");
  ok(aphro_rms_norm(out.data_ptr(), x.data_ptr(), weight.data_ptr(), (float)epsilon, x.size(0), (int)hidden, x.stride(0),
                    act_dtype(input), cur_stream()),
     "rms_norm");
}

// fused_add_rms_norm (layernorm_kernels.cu:200-240)
void fused_add_rms_norm(torch::Tensor input, torch::Tensor residual, torch::Tensor weight, double epsilon) {
  TORCH_CHECK(input.is_contiguous() && residual.is_contiguous(), "fused_add_rms_norm: contiguous input / residual expected");
  ok(aphro_fused_add_rms_norm(input.data_ptr(), residual.data_ptr(), weight.data_ptr(), (float)epsilon, rows_of(input),
                              (int)input.size(-1), act_dtype(input), cur_stream()),
     "fused_add_rms_norm");
}

// silu_and_mul (activation_kernels.cu:55-75)
void silu_and_mul(torch::Tensor out, torch::Tensor input) {
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous() && input.size(-1) % 2 == 0, "silu_and_mul: contiguous [.., 2 d] expected");
  const int64_t d = input.size(-1) / 2;
  ok(aphro_silu_and_mul(out.data_ptr(), input.data_ptr(), input.numel() / (2 * d), (int)d, act_dtype(input), cur_stream()),
     "silu_and_mul");
}

// rotary_embedding (pos_encoding_kernels.cu:120-160), in place on query / key
void rotary_embedding(torch::Tensor positions, torch::Tensor query, torch::Tensor key, int64_t head_size,
                      torch::Tensor cos_sin_cache, bool is_neox) {
  const int64_t num_tokens = positions.numel();
  auto q2 = query.dim() == 2 ? query : query.view({num_tokens, -1});
  auto k2 = key.dim() == 2 ? key : key.view({num_tokens, -1});
  auto pos = positions.scalar_type() == torch::kLong ? positions : positions.to(torch::kLong);
  ok(aphro_rotary_embedding(pos.data_ptr<int64_t>(), q2.data_ptr(), k2.data_ptr(), num_tokens, (int)(q2.size(1) / head_size),
                            (int)(k2.size(1) / head_size), (int)head_size, (int)cos_sin_cache.size(1), cos_sin_cache.data_ptr(),
                            q2.stride(0), k2.stride(0), is_neox ? 1 : 0, act_dtype(query), cur_stream()),
     "rotary_embedding");
}

// paged_attention_v2 (attention_kernels.cu:833-998): scratch sized for 512-token partitions (paged_attn.py:13, 118-119)
void paged_attention_v2(torch::Tensor out, torch::Tensor exp_sums, torch::Tensor max_logits, torch::Tensor tmp_out,
                        torch::Tensor query, torch::Tensor key_cache, torch::Tensor value_cache, int64_t num_kv_heads, double scale,
                        torch::Tensor block_tables, torch::Tensor seq_lens, int64_t block_size, int64_t max_seq_len,
                        const c10::optional<torch::Tensor>& alibi_slopes, std::string kv_cache_dtype, double k_scale,
                        double v_scale, int64_t tp_rank, int64_t blocksparse_local_blocks, int64_t blocksparse_vert_stride,
                        int64_t blocksparse_block_size, int64_t blocksparse_head_sliding_step) {
  TORCH_CHECK(blocksparse_vert_stride <= 1, "blocksparse attention is not implemented on MI355X");
  const int64_t part = 512, need = (max_seq_len + part - 1) / part;
  TORCH_CHECK(tmp_out.is_contiguous() && tmp_out.size(2) >= need, "tmp_out has ", tmp_out.size(2), " partitions, need ", need,
              " for max_seq_len=", max_seq_len);
  const int64_t num_seqs = query.size(0), num_heads = query.size(1), head_size = query.size(2);
  ok(aphro_paged_attention(out.data_ptr(), exp_sums.data_ptr<float>(), max_logits.data_ptr<float>(), tmp_out.data_ptr(),
                           query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), (int)num_seqs, (int)num_heads,
                           (int)num_kv_heads, (int)head_size, (float)scale, block_tables.data_ptr<int>(), seq_lens.data_ptr<int>(),
                           (int)block_tables.size(1), (int)block_size, (int)max_seq_len,
                           alibi_slopes ? alibi_slopes->data_ptr<float>() : nullptr, query.stride(0), key_cache.stride(0),
                           key_cache.stride(1), act_dtype(query), kv_dtype(kv_cache_dtype), (float)k_scale, (float)v_scale, (int)part,
                           cur_stream()),
     "paged_attention_v2");
}

// gptq_shuffle (q_gemm.cu:1862-1872): 4-bit in C++; the 2 / 3 / 8-bit widths keep their sequential layout (Python op)
void gptq_shuffle(torch::Tensor q_weight, torch::Tensor q_perm, int64_t bit) {
  TORCH_CHECK(bit == 4, "gptq_shuffle (C++ registration): 4-bit weights; 2 / 3 / 8-bit go through the Python op");
  const bool act_order = q_perm.numel() > 0 && q_perm.device().is_cuda();
  torch::Tensor perm = act_order ? q_perm.to(torch::kInt) : q_perm;
  torch::Tensor tmp = act_order ? torch::empty_like(q_weight) : torch::Tensor();
  ok(aphro_gptq_shuffle((uint32_t*)q_weight.data_ptr(), act_order ? (const int32_t*)perm.data_ptr() : nullptr,
                        q_weight.size(0) * 8, q_weight.size(1), 4, act_order ? (uint32_t*)tmp.data_ptr() : nullptr, cur_stream()),
     "gptq_shuffle");
}

// awq_dequantize (awq/gemm_kernels.cu:694-781): [K, N] in the scales' dtype
torch::Tensor awq_dequantize(torch::Tensor kernel, torch::Tensor scaling_factors, torch::Tensor zeros, int64_t split_k_iters,
                             int64_t thx, int64_t thy) {
  const int64_t k = kernel.size(0), n = kernel.size(1) * 8;
  auto out = torch::empty({k, n}, scaling_factors.options());
  ok(aphro_awq_dequantize((const uint32_t*)kernel.data_ptr(), scaling_factors.data_ptr(), (const uint32_t*)zeros.data_ptr(),
                          out.data_ptr(), k, n, scaling_factors.size(0), act_dtype(scaling_factors), cur_stream()),
     "awq_dequantize");
  return out;
}

int quant_in_dtype(const torch::Tensor& t) {
  if (t.scalar_type() == torch::kFloat) return APHRO_F32;
  return act_dtype(t);
}

// static_scaled_fp8_quant (fp8/common.cu:187-199)
void static_scaled_fp8_quant(torch::Tensor out, torch::Tensor input, torch::Tensor scale) {
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous() && scale.numel() == 1 && scale.scalar_type() == torch::kFloat,
              "static_scaled_fp8_quant: contiguous tensors and one fp32 scale expected");
  ok(aphro_static_scaled_fp8_quant(out.data_ptr(), input.data_ptr(), scale.data_ptr<float>(), rows_of(input), input.size(-1),
                                   quant_in_dtype(input), cur_stream()),
     "static_scaled_fp8_quant");
}

// dynamic_per_token_scaled_fp8_quant (fp8/common.cu:201-256)
void dynamic_per_token_scaled_fp8_quant(torch::Tensor out, torch::Tensor input, torch::Tensor scale,
                                        const c10::optional<torch::Tensor>& scale_ub) {
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous() && scale.scalar_type() == torch::kFloat && scale.numel() >= rows_of(input),
              "dynamic_per_token_scaled_fp8_quant: contiguous tensors and one fp32 scale per token expected");
  ok(aphro_dynamic_per_token_scaled_fp8_quant(out.data_ptr(), input.data_ptr(), scale.data_ptr<float>(),
                                              scale_ub ? scale_ub->data_ptr<float>() : nullptr, rows_of(input), input.size(-1),
                                              quant_in_dtype(input), cur_stream()),
     "dynamic_per_token_scaled_fp8_quant");
}

// advance_step_flashattn (prepare_inputs/advance_step.cu)
void advance_step_flashattn(int64_t num_seqs, int64_t num_queries, int64_t block_size, torch::Tensor input_tokens,
                            torch::Tensor sampled_token_ids, torch::Tensor input_positions, torch::Tensor seq_lens,
                            torch::Tensor slot_mapping, torch::Tensor block_tables) {
  TORCH_CHECK(input_tokens.scalar_type() == torch::kLong && sampled_token_ids.scalar_type() == torch::kLong &&
              input_positions.scalar_type() == torch::kLong && seq_lens.scalar_type() == torch::kInt &&
              slot_mapping.scalar_type() == torch::kLong && block_tables.scalar_type() == torch::kInt,
              "advance_step_flashattn: int64 tokens / positions / slots, int32 seq_lens / block_tables expected");
  TORCH_CHECK(input_tokens.is_contiguous() && sampled_token_ids.is_contiguous() && input_positions.is_contiguous() &&
              seq_lens.is_contiguous() && slot_mapping.is_contiguous() && block_tables.stride(1) == 1,
              "advance_step_flashattn: contiguous tensors expected");
  ok(aphro_advance_step_flashattn((int)num_seqs, (int)num_queries, (int)block_size, input_tokens.data_ptr<int64_t>(),
                                  sampled_token_ids.data_ptr<int64_t>(), input_positions.data_ptr<int64_t>(), seq_lens.data_ptr<int>(),
                                  slot_mapping.data_ptr<int64_t>(), block_tables.data_ptr<int>(), block_tables.stride(0), cur_stream()),
     "advance_step_flashattn");
}

// reshape_and_cache_flash (cache_kernels.cu:265-309): caches [NB, block, H, hd]
void reshape_and_cache_flash(torch::Tensor key, torch::Tensor value, torch::Tensor key_cache, torch::Tensor value_cache,
                             torch::Tensor slot_mapping, std::string kv_cache_dtype, double k_scale, double v_scale) {
  TORCH_CHECK(slot_mapping.scalar_type() == torch::kLong, "slot_mapping must be int64");
  TORCH_CHECK(key_cache.stride(0) == value_cache.stride(0), "key_cache and value_cache must have the same block stride");
  TORCH_CHECK(key.stride(1) == key.size(2) && value.stride(1) == value.size(2), "key/value heads must be contiguous");
  ok(aphro_reshape_and_cache_flash(key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
                                   slot_mapping.data_ptr<int64_t>(), key.size(0), (int)key.size(1), (int)key.size(2),
                                   (int)key_cache.size(1), key_cache.stride(0), key.stride(0), value.stride(0), act_dtype(key),
                                   kv_dtype(kv_cache_dtype), (float)k_scale, (float)v_scale, cur_stream()),
     "reshape_and_cache_flash");
}

// convert_fp8 (cache_kernels.cu:330-380)
void convert_fp8(torch::Tensor dst_cache, torch::Tensor src_cache, double scale, std::string kv_cache_dtype) {
  // "auto" = the platform's fp8 format, as the reference's dispatch accepts it (cache_kernels.cu:373-388); ADVICE r4
  const int kvd = kv_dtype(kv_cache_dtype == "auto" ? std::string("fp8_e4m3") : kv_cache_dtype);
  TORCH_CHECK(dst_cache.is_contiguous() && src_cache.is_contiguous(), "convert_fp8 needs contiguous tensors");
  const bool to_fp8 = dst_cache.scalar_type() == torch::kByte;
  const torch::Tensor& hp = to_fp8 ? src_cache : dst_cache;
  ok(aphro_convert_fp8(dst_cache.data_ptr(), src_cache.data_ptr(), src_cache.numel(), (float)scale, quant_in_dtype(hp), kvd,
                       to_fp8 ? 1 : 0, cur_stream()),
     "convert_fp8");
}

// dynamic_scaled_fp8_quant (fp8/common.cu:133-186): ONE scale over the whole tensor.  `scale` is zeroed here, as the
// reference's Python wrapper does before the call (_custom_ops.py:676), so that the atomic-max form needs no scratch.
void dynamic_scaled_fp8_quant(torch::Tensor out, torch::Tensor input, torch::Tensor scale) {
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous() && scale.scalar_type() == torch::kFloat && scale.numel() >= 1,
              "dynamic_scaled_fp8_quant: contiguous tensors and an fp32 scale expected");
  scale.zero_();
  ok(aphro_dynamic_scaled_fp8_quant(out.data_ptr(), input.data_ptr(), scale.data_ptr<float>(), rows_of(input), input.size(-1),
                                    quant_in_dtype(input), cur_stream()),
     "dynamic_scaled_fp8_quant");
}

// moe_align_block_size (moe_align_block_size_kernels.cu)
void moe_align_block_size(torch::Tensor topk_ids, int64_t num_experts, int64_t block_size, torch::Tensor sorted_token_ids,
                          torch::Tensor experts_ids, torch::Tensor num_tokens_post_pad) {
  TORCH_CHECK(topk_ids.scalar_type() == torch::kInt && topk_ids.is_contiguous(), "moe_align_block_size: topk_ids must be contiguous int32");
  const int64_t numel = topk_ids.numel(), need = numel + num_experts * (block_size - 1);
  TORCH_CHECK(sorted_token_ids.numel() >= need && experts_ids.numel() >= (need + block_size - 1) / block_size,
              "moe_align_block_size: output tensors too small");
  ok(aphro_moe_align_block_size(topk_ids.data_ptr<int32_t>(), (int)num_experts, (int)block_size, sorted_token_ids.data_ptr<int32_t>(),
                                experts_ids.data_ptr<int32_t>(), num_tokens_post_pad.data_ptr<int32_t>(), nullptr, numel, cur_stream()),
     "moe_align_block_size");
}

// _moe_C::topk_softmax (kernels/moe/topk_softmax_kernels.cu)
void topk_softmax(torch::Tensor topk_weights, torch::Tensor topk_indices, torch::Tensor token_expert_indices, torch::Tensor gating_output) {
  TORCH_CHECK(gating_output.scalar_type() == torch::kFloat && gating_output.is_contiguous() && gating_output.dim() == 2,
              "topk_softmax: gating_output must be contiguous float32 [tokens, experts]");
  TORCH_CHECK(topk_indices.scalar_type() == torch::kInt && topk_weights.scalar_type() == torch::kFloat,
              "topk_softmax: topk_weights must be float32 and topk_ids int32");
  ok(aphro_topk_softmax(topk_weights.data_ptr<float>(), topk_indices.data_ptr<int32_t>(),
                        token_expert_indices.defined() && token_expert_indices.numel() ? token_expert_indices.data_ptr<int32_t>() : nullptr,
                        gating_output.data_ptr<float>(), gating_output.size(0), (int)gating_output.size(1), (int)topk_indices.size(1),
                        cur_stream()),
     "topk_softmax");
}

// swap_blocks (cache_kernels.cu:24-63): block_mapping is a CPU int64 [pairs, 2] tensor
void swap_blocks(torch::Tensor src, torch::Tensor dst, const torch::Tensor& block_mapping) {
  TORCH_CHECK(block_mapping.device().is_cpu(), "block_mapping must be on CPU");
  int kind;
  if (src.is_cuda() && dst.is_cuda()) {
    TORCH_CHECK(src.device().index() == dst.device().index(), "src and dst must be on the same GPU");
    kind = 0;
  } else if (src.is_cuda() && dst.device().is_cpu()) {
    kind = 1;
  } else if (src.device().is_cpu() && dst.is_cuda()) {
    kind = 2;
  } else {
    TORCH_CHECK(false, "Invalid device combination");
  }
  const torch::Tensor bm = block_mapping.to(torch::kLong).contiguous();
  if (bm.numel() == 0) return;
  const int64_t block_bytes = src[0].numel() * src.element_size();
  const c10::DeviceGuard guard(src.is_cuda() ? src.device() : dst.device());
  ok(aphro_swap_blocks(src.data_ptr(), dst.data_ptr(), bm.data_ptr<int64_t>(), bm.size(0), block_bytes, kind, cur_stream()), "swap_blocks");
}

// copy_blocks (cache_kernels.cu:103-148): block_mapping int64 [pairs, 2] on the device
void copy_blocks(std::vector<torch::Tensor> const& key_caches, std::vector<torch::Tensor> const& value_caches,
                 const torch::Tensor& block_mapping) {
  const int64_t num_layers = (int64_t)key_caches.size();
  TORCH_CHECK(num_layers == (int64_t)value_caches.size(), "copy_blocks: key_caches and value_caches differ in length");
  if (num_layers == 0 || block_mapping.numel() == 0) return;
  TORCH_CHECK(key_caches[0].is_cuda() && block_mapping.is_cuda(), "copy_blocks: device tensors expected");
  std::vector<int64_t> kp(num_layers), vp(num_layers);
  for (int64_t i = 0; i < num_layers; ++i) { kp[i] = (int64_t)key_caches[i].data_ptr(); vp[i] = (int64_t)value_caches[i].data_ptr(); }
  const auto opts = torch::TensorOptions().dtype(torch::kLong);
  const torch::Tensor kpt = torch::from_blob(kp.data(), {num_layers}, opts).to(key_caches[0].device());
  const torch::Tensor vpt = torch::from_blob(vp.data(), {num_layers}, opts).to(key_caches[0].device());
  const torch::Tensor bm = block_mapping.to(torch::kLong).contiguous();
  const int64_t block_bytes = key_caches[0][0].numel() * key_caches[0].element_size();
  ok(aphro_copy_blocks(kpt.data_ptr<int64_t>(), vpt.data_ptr<int64_t>(), (int)num_layers, bm.data_ptr<int64_t>(), bm.size(0),
                       block_bytes, cur_stream()),
     "copy_blocks");
}

// awq_gemm (awq/gemm_kernels.cu:784-858): [M, N] in the activations' dtype.  Checkpoint-layout AWQ tensors; from 256 rows
// the nibbles are transposed once per call and the prefill-sized MFMA kernel runs (the Python op's rule, _custom_ops.awq_gemm)
torch::Tensor awq_gemm(torch::Tensor in_feats, torch::Tensor kernel, torch::Tensor scaling_factors, torch::Tensor zeros,
                       int64_t split_k_iters) {
  TORCH_CHECK(in_feats.is_cuda() && kernel.is_cuda() && scaling_factors.is_cuda() && zeros.is_cuda() && in_feats.dim() == 2,
              "awq_gemm: device tensors and a 2-D activation matrix expected");
  torch::Tensor a = in_feats.stride(1) == 1 ? in_feats : in_feats.contiguous();
  const int64_t m = a.size(0), k = a.size(1), n = kernel.size(1) * 8, groups = scaling_factors.size(0);
  TORCH_CHECK(groups > 0 && k % groups == 0, "awq_gemm: scales [groups, N] with groups dividing K");
  const int dt = act_dtype(a);
  auto out = torch::empty({m, n}, a.options());
  const int64_t gs = k / groups;
  const bool large = m >= 256 && n % 128 == 0 && k % 64 == 0 && gs % 64 == 0 && (k / 8) * n * 4 < (int64_t(1) << 32) &&
                     m * k * 2 < (int64_t(1) << 32) && getenv("APHRO_WNA16_NO_LARGE") == nullptr;
  if (large) {
    auto qw = torch::empty({k / 8, n}, kernel.options());
    auto qz = torch::empty_like(zeros);
    ok(aphro_awq_repack((const uint32_t*)kernel.data_ptr(), (uint32_t*)qw.data_ptr(), k, n, cur_stream()), "awq_gemm");
    ok(aphro_awq_repack_zeros((const uint32_t*)zeros.data_ptr(), (uint32_t*)qz.data_ptr(), zeros.size(0), n, cur_stream()), "awq_gemm");
    if (a.stride(0) % 8 != 0 || ((uintptr_t)a.data_ptr() % 16) != 0) a = a.contiguous();
    const size_t nb = aphro_wna16_gemm_large_workspace_bytes(m, n, k, groups, dt);
    auto ws = torch::empty({(int64_t)nb}, a.options().dtype(torch::kUInt8));
    ok(aphro_wna16_gemm_large(a.data_ptr(), (const uint32_t*)qw.data_ptr(), (const uint32_t*)qz.data_ptr(),
                              scaling_factors.data_ptr(), out.data_ptr(), nb ? ws.data_ptr() : nullptr, nb, m, n, k, groups,
                              a.stride(0), 0, dt, cur_stream()),
       "awq_gemm");
    return out;
  }
  const size_t nb = aphro_awq_gemm_workspace_bytes(m < 64 ? m : 64, n, k, groups);
  auto ws = torch::empty({(int64_t)nb}, a.options().dtype(torch::kUInt8));
  const int64_t esz = a.element_size();
  for (int64_t m0 = 0; m0 < m; m0 += 64) {
    const int64_t rows = m - m0 < 64 ? m - m0 : 64;
    ok(aphro_awq_gemm((const char*)a.data_ptr() + m0 * a.stride(0) * esz, (const uint32_t*)kernel.data_ptr(),
                      scaling_factors.data_ptr(), (const uint32_t*)zeros.data_ptr(), (char*)out.data_ptr() + m0 * n * esz,
                      ws.data_ptr(), nb, rows, n, k, groups, a.stride(0), dt, cur_stream()),
       "awq_gemm");
  }
  return out;
}

// gptq_marlin_repack (gptq_marlin_repack.cu:257-340): the Marlin-role load-time prepack into the CDNA4 K-packed layout
// ([K/8, N], same shape as the input); perm = argsort(g_idx) or empty
torch::Tensor gptq_marlin_repack(torch::Tensor b_q_weight, torch::Tensor perm, c10::SymInt size_k, c10::SymInt size_n, int64_t num_bits) {
  TORCH_CHECK(b_q_weight.is_cuda(), "gptq_marlin_repack: device tensor expected");
  const bool has_perm = perm.defined() && perm.numel() > 0;
  torch::Tensor p32 = has_perm ? perm.to(torch::kInt) : perm;
  if (num_bits == 8) {   // uint8b128: the sequential [K/4, N] words stay; act-order rows are made sequential (gptq_shuffle's 8-bit form)
    if (!has_perm) return b_q_weight.clone();
    auto seq = torch::empty_like(b_q_weight);
    ok(aphro_gptq_make_sequential_bits((const uint32_t*)b_q_weight.data_ptr(), (uint32_t*)seq.data_ptr(), (const int32_t*)p32.data_ptr(),
                                       size_k.expect_int(), size_n.expect_int(), 8, cur_stream()),
       "gptq_marlin_repack");
    return seq;
  }
  auto out = torch::empty_like(b_q_weight);
  ok(aphro_gptq_repack((const uint32_t*)b_q_weight.data_ptr(), has_perm ? (const int32_t*)p32.data_ptr() : nullptr,
                       (uint32_t*)out.data_ptr(), size_k.expect_int(), size_n.expect_int(), (int)num_bits, cur_stream()),
     "gptq_marlin_repack");
  return out;
}

// awq_marlin_repack (awq_marlin_repack.cu:199-267): AWQ [K, N/8] -> CDNA4 K-packed [K/8, N]
torch::Tensor awq_marlin_repack(torch::Tensor b_q_weight, c10::SymInt size_k, c10::SymInt size_n, int64_t num_bits) {
  TORCH_CHECK(b_q_weight.is_cuda(), "awq_marlin_repack: device tensor expected");
  TORCH_CHECK(num_bits == 4, "only 4-bit AWQ is implemented");
  const int64_t k = size_k.expect_int(), n = size_n.expect_int();
  auto out = torch::empty({k / 8, n}, b_q_weight.options().dtype(torch::kInt));
  ok(aphro_awq_repack((const uint32_t*)b_q_weight.data_ptr(), (uint32_t*)out.data_ptr(), k, n, cur_stream()), "awq_marlin_repack");
  return out;
}

// The W4A16 dispatch of the Python op surface (_custom_ops._wna16): prompt-sized tile machine, the one-pass 33..64-row kernel
// where it wins, else the decode kernel 64 rows at a time.  K-packed exllama-order weights; perm = act-order gather or undefined.
torch::Tensor wna16_dispatch(torch::Tensor a, const torch::Tensor& qweight, const torch::Tensor& qzeros, const torch::Tensor& scales,
                             const torch::Tensor& perm, int zero_offset, const char* op) {
  const int64_t m = a.size(0), k = a.size(1), n = qweight.size(1), groups = scales.size(0);
  const int dt = act_dtype(a);
  const bool has_perm = perm.defined() && perm.numel() > 0;
  const int64_t gs = groups > 0 ? k / groups : 0;
  const bool large_ok = m > 64 && n % 128 == 0 && k % 64 == 0 && groups > 0 && k % groups == 0 && gs % 64 == 0 &&
                        (k / 8) * n * 4 < (int64_t(1) << 32) && m * k * 2 < (int64_t(1) << 32);
  const bool prefers_large = m > 128 || n * k >= (int64_t(1) << 25);
  auto out = torch::empty({m, n}, a.options());
  auto gathered = [&]() {
    torch::Tensor x = has_perm ? a.index_select(1, perm.to(torch::kLong)) : a;
    if (x.stride(1) != 1 || x.stride(0) % 8 != 0 || ((uintptr_t)x.data_ptr() % 16) != 0) x = x.contiguous();
    return x;
  };
  if (large_ok && prefers_large && getenv("APHRO_WNA16_NO_LARGE") == nullptr) {
    torch::Tensor x = gathered();
    const size_t nb = aphro_wna16_gemm_large_workspace_bytes(m, n, k, groups, dt);
    auto ws = torch::empty({(int64_t)nb}, a.options().dtype(torch::kUInt8));
    ok(aphro_wna16_gemm_large(x.data_ptr(), (const uint32_t*)qweight.data_ptr(), (const uint32_t*)qzeros.data_ptr(), scales.data_ptr(),
                              out.data_ptr(), nb ? ws.data_ptr() : nullptr, nb, m, n, k, groups, x.stride(0), zero_offset, dt, cur_stream()),
       op);
    return out;
  }
  if (m > 32 && m <= 64 && n * k >= (int64_t(1) << 25) && m * k * 2 < (int64_t(1) << 32) &&
      aphro_wna16_gemm_mid_supported(m, n, k, groups) && getenv("APHRO_WNA16_NO_MID") == nullptr) {
    torch::Tensor x = gathered();
    const size_t nb = aphro_wna16_gemm_mid_workspace_bytes(m, n, k, groups);
    auto ws = torch::empty({(int64_t)nb}, a.options().dtype(torch::kUInt8));
    ok(aphro_wna16_gemm_mid(x.data_ptr(), (const uint32_t*)qweight.data_ptr(), (const uint32_t*)qzeros.data_ptr(), scales.data_ptr(),
                            out.data_ptr(), ws.data_ptr(), nb, m, n, k, groups, x.stride(0), zero_offset, dt, cur_stream()),
       op);
    return out;
  }
  if (a.stride(1) != 1) a = a.contiguous();
  auto ws = torch::empty({(int64_t)aphro_wna16_workspace_bytes(m < 64 ? m : 64, n, k)}, a.options().dtype(torch::kUInt8));
  torch::Tensor tmp, p32;
  if (has_perm) {
    tmp = torch::empty({m < 64 ? m : 64, k}, a.options());
    p32 = perm.to(torch::kInt);
  }
  for (int64_t m0 = 0; m0 < m; m0 += 64) {
    const int64_t rows = m - m0 < 64 ? m - m0 : 64;
    ok(aphro_gptq_gemm((const char*)a.data_ptr() + m0 * a.stride(0) * a.element_size(), (const uint32_t*)qweight.data_ptr(),
                       (const uint32_t*)qzeros.data_ptr(), scales.data_ptr(), has_perm ? (const int32_t*)p32.data_ptr() : nullptr,
                       has_perm ? tmp.data_ptr() : nullptr, (char*)out.data_ptr() + m0 * n * out.element_size(), ws.data_ptr(),
                       (size_t)ws.numel(), rows, n, k, groups, a.stride(0), zero_offset, dt, cur_stream()),
       op);
  }
  return out;
}

// gptq_marlin_gemm (gptq_marlin.cu:2136-2330; schema :195-201 with the type as its size in bits -- the verbatim schema names
// the torchbind class _core_C.ScalarType, which only exists inside the reference: torch_ops.py).  b_q_weight = the
// gptq_marlin_repack / awq_marlin_repack output; b_zeros plain-order int32 [G, N/8] when has_zp, else the uint4b8 zero point 8.
torch::Tensor gptq_marlin_gemm(torch::Tensor a, torch::Tensor b_q_weight, torch::Tensor b_scales, torch::Tensor b_zeros,
                               torch::Tensor g_idx, torch::Tensor perm, torch::Tensor workspace, int64_t b_q_type, int64_t size_m,
                               int64_t size_n, int64_t size_k, bool is_k_full, bool has_zp, bool use_fp32_reduce, bool is_zp_float) {
  TORCH_CHECK(!is_zp_float, "gptq_marlin_gemm: float zero points are not supported");
  TORCH_CHECK(b_q_type == 4 || b_q_type == 8, "gptq_marlin_gemm on MI355X serves uint4 / uint4b8 and uint8b128 weights");
  torch::Tensor x = a.reshape({-1, a.size(-1)});
  if (b_q_type == 8) {   // uint8b128 (marlin_utils.py:28-45): sequential [K/4, N] words, zero point 128 = 127 per byte stored-minus-one
    TORCH_CHECK(!has_zp, "gptq_marlin_gemm: the 8-bit type served is uint8b128 (symmetric, no zero points)");
    TORCH_CHECK(x.size(0) == size_m && x.size(1) == size_k && b_q_weight.dim() == 2 && b_q_weight.size(0) == size_k / 4 &&
                b_q_weight.size(1) == size_n, "gptq_marlin_gemm: shape mismatch");
    auto zp8 = torch::full({b_scales.size(0), size_n / 4}, (int64_t)0x7f7f7f7f, b_q_weight.options().dtype(torch::kInt));
    torch::Tensor g = perm.defined() && perm.numel() > 0 ? perm : torch::empty({0}, b_q_weight.options().dtype(torch::kInt));
    return gptq_gemm(x.stride(1) == 1 ? x : x.contiguous(), b_q_weight, zp8, b_scales, g, true, 8);
  }
  TORCH_CHECK(x.size(0) == size_m && x.size(1) == size_k && b_q_weight.dim() == 2 && b_q_weight.size(0) == size_k / 8 &&
              b_q_weight.size(1) == size_n, "gptq_marlin_gemm: shape mismatch");
  torch::Tensor zp = b_zeros;
  if (!has_zp)
    zp = torch::full({b_scales.size(0), size_n / 8}, (int64_t)0x88888888 - ((int64_t)1 << 32), b_q_weight.options().dtype(torch::kInt));
  return wna16_dispatch(x, b_q_weight, zp, b_scales, perm, 0, "gptq_marlin_gemm");
}

// fp8_marlin_gemm (fp8_marlin.cu:1212; :218-222): W8A16 on the checkpoint layout, e4m3 [N, K] row-major, scales fp32 [1] or [N]
torch::Tensor fp8_marlin_gemm(torch::Tensor a, torch::Tensor b_q_weight, torch::Tensor b_scales, torch::Tensor workspace,
                              int64_t num_bits, int64_t size_m, int64_t size_n, int64_t size_k) {
  TORCH_CHECK(num_bits == 8 && a.is_cuda() && b_q_weight.is_cuda() && b_scales.is_cuda(), "fp8_marlin_gemm: 8-bit device tensors expected");
  if (a.stride(1) != 1) a = a.contiguous();
  const int dt = act_dtype(a);
  torch::Tensor sb = b_scales.reshape({-1}).to(torch::kFloat);
  const int per_channel = sb.numel() > 1 ? 1 : 0;
  if (size_m > 64 && size_n % 128 == 0 && size_k % 64 == 0 && getenv("APHRO_WNA16_NO_LARGE") == nullptr) {
    torch::Tensor x = a.narrow(0, 0, size_m);
    if (x.stride(0) % 8 != 0 || ((uintptr_t)x.data_ptr() % 16) != 0) x = x.contiguous();
    auto out = torch::empty({size_m, size_n}, a.options());
    const size_t nb = aphro_fp8_w8a16_gemm_large_workspace_bytes(size_m, size_n, size_k, dt);
    auto ws = torch::empty({(int64_t)nb}, a.options().dtype(torch::kUInt8));
    ok(aphro_fp8_w8a16_gemm_large(out.data_ptr(), x.data_ptr(), b_q_weight.data_ptr(), sb.data_ptr<float>(), nullptr,
                                  nb ? ws.data_ptr() : nullptr, nb, size_m, size_n, size_k, x.stride(0), per_channel, dt, cur_stream()),
       "fp8_marlin_gemm");
    return out;
  }
  if (size_m >= 256) {     // shapes the hand-written kernel does not tile: widen the weight once (exact) + a library GEMM
    torch::Tensor w = b_q_weight.to(a.scalar_type());
    torch::Tensor s = b_scales.reshape({-1}).to(a.scalar_type());
    w = per_channel ? w * s.reshape({-1, 1}) : w * s;
    return torch::matmul(a.narrow(0, 0, size_m), w.t());
  }
  auto out = torch::empty({size_m, size_n}, a.options());
  const size_t nb = aphro_fp8_gemm_workspace_bytes(size_m < 64 ? size_m : 64, size_n, size_k);
  auto ws = torch::empty({(int64_t)nb}, a.options().dtype(torch::kUInt8));
  for (int64_t m0 = 0; m0 < size_m; m0 += 64) {
    const int64_t rows = size_m - m0 < 64 ? size_m - m0 : 64;
    ok(aphro_fp8_w8a16_gemm((char*)out.data_ptr() + m0 * size_n * out.element_size(),
                            (const char*)a.data_ptr() + m0 * a.stride(0) * a.element_size(), b_q_weight.data_ptr(), sb.data_ptr<float>(),
                            nullptr, ws.data_ptr(), nb, rows, size_n, size_k, a.stride(0), per_channel, dt, cur_stream()),
       "fp8_marlin_gemm");
  }
  return out;
}

// cutlass_scaled_mm_supports_fp8 (scaled_mm_entry.cu:30-50): gfx950 has native OCP fp8 MFMA
bool cutlass_scaled_mm_supports_fp8(int64_t cuda_device_capability) { return true; }

// _rocm_C::paged_attention (kernels/rocm/attention.cu:903-1000; schema kernels/rocm/torch_bindings.cpp:17-28): scratch
// sized for 512-token partitions; one launch over whole sequences where that is the faster form (the rule of the Python
// op, _custom_ops.paged_attention_rocm -- the reference's own v1 / v2 rule, paged_attn.py:112-121)
void paged_attention_rocm(torch::Tensor out, torch::Tensor exp_sums, torch::Tensor max_logits, torch::Tensor tmp_out,
                          torch::Tensor query, torch::Tensor key_cache, torch::Tensor value_cache, int64_t num_kv_heads,
                          double scale, torch::Tensor block_tables, torch::Tensor context_lens, int64_t block_size,
                          int64_t max_context_len, const c10::optional<torch::Tensor>& alibi_slopes, std::string kv_cache_dtype,
                          double k_scale, double v_scale) {
  int64_t part = 512;
  const int64_t need = (max_context_len + part - 1) / part;
  TORCH_CHECK(tmp_out.is_contiguous() && tmp_out.size(2) >= need, "tmp_out has ", tmp_out.size(2), " partitions, need ", need,
              " for max_seq_len=", max_context_len);
  const int64_t num_seqs = query.size(0), num_heads = query.size(1), head_size = query.size(2);
  TORCH_CHECK(query.stride(2) == 1 && query.stride(1) == head_size && out.is_contiguous(),
              "paged_attention: contiguous query heads and out expected");
  const bool whole = max_context_len <= 8192 && (need == 1 || num_seqs * num_heads > 512) &&
                     getenv("APHRO_PA_ROCM_PARTITIONED") == nullptr;
  if (whole) part = 0;
  ok(aphro_paged_attention(out.data_ptr(), whole ? nullptr : exp_sums.data_ptr<float>(),
                           whole ? nullptr : max_logits.data_ptr<float>(), whole ? nullptr : tmp_out.data_ptr(), query.data_ptr(),
                           key_cache.data_ptr(), value_cache.data_ptr(), (int)num_seqs, (int)num_heads, (int)num_kv_heads,
                           (int)head_size, (float)scale, block_tables.data_ptr<int>(), context_lens.data_ptr<int>(),
                           (int)block_tables.stride(0), (int)block_size, (int)max_context_len,
                           alibi_slopes ? alibi_slopes->data_ptr<float>() : nullptr, query.stride(0), key_cache.stride(0),
                           key_cache.stride(1), act_dtype(query), kv_dtype(kv_cache_dtype), (float)k_scale, (float)v_scale,
                           (int)part, cur_stream()),
     "paged_attention");
}

// _C_custom_ar (custom_all_reduce.cu).  IPC handles are `str[]` in the schema: through a C++-registered op a Python `bytes`
// object arrives as the raw std::string the reference's C++ expects (the Python-registered ops only see decoded text,
// torch_ops.py) -- see ipc_raw below.
int ar_dtype(const torch::Tensor& t) {
  if (t.scalar_type() == torch::kFloat) return APHRO_F32;
  TORCH_CHECK(t.scalar_type() == torch::kHalf || t.scalar_type() == torch::kBFloat16,
              "custom allreduce only supports float32, float16 and bfloat16");
  return act_dtype(t);
}
void ar_check_io(const torch::Tensor& inp, const torch::Tensor& out) {
  TORCH_CHECK(inp.is_cuda() && out.is_cuda() && inp.scalar_type() == out.scalar_type() && inp.numel() == out.numel(),
              "all_reduce: inp and out must be device tensors of the same dtype and number of elements");
}
// custom_all_reduce.cu:84-92
void all_reduce_reg(int64_t fa, torch::Tensor inp, torch::Tensor out) {
  ar_check_io(inp, out);
  ok(aphro_custom_ar_all_reduce((void*)fa, inp.data_ptr(), out.data_ptr(), inp.numel(), ar_dtype(inp), nullptr, 0, cur_stream()),
     "all_reduce_reg");
}
// custom_all_reduce.cu:94-109
void all_reduce_unreg(int64_t fa, torch::Tensor inp, torch::Tensor reg_buffer, torch::Tensor out) {
  ar_check_io(inp, out);
  const size_t nbytes = (size_t)inp.numel() * inp.element_size(), cap = (size_t)reg_buffer.numel() * reg_buffer.element_size();
  TORCH_CHECK(nbytes <= cap, "registered buffer is too small to contain the input");
  ok(aphro_custom_ar_all_reduce((void*)fa, inp.data_ptr(), out.data_ptr(), inp.numel(), ar_dtype(inp), reg_buffer.data_ptr(), cap,
                                cur_stream()),
     "all_reduce_unreg");
}
int64_t meta_size() { return aphro_custom_ar_meta_size(); }   // custom_all_reduce.cu:111-113

// IPC handles arrive as `str` = std::string: raw bytes when the caller passes Python `bytes` (what the reference's
// CustomAllreduce holds, custom_all_reduce.py:210-214), or the 128 hex digits / latin-1 text the Python-registered ops take
std::string ipc_raw(const std::string& h) {
  const size_t n = (size_t)aphro_ipc_handle_bytes();
  if (h.size() == n) return h;
  if (h.size() == 2 * n) {
    std::string out(n, '\0');
    for (size_t i = 0; i < n; ++i) {
      auto nib = [&](char ch) -> int {
        if (ch >= '0' && ch <= '9') return ch - '0';
        if (ch >= 'a' && ch <= 'f') return ch - 'a' + 10;
        if (ch >= 'A' && ch <= 'F') return ch - 'A' + 10;
        TORCH_CHECK(false, "IPC handle: not a hex string");
      };
      out[i] = (char)((nib(h[2 * i]) << 4) | nib(h[2 * i + 1]));
    }
    return out;
  }
  // latin-1 text that the dispatcher UTF-8 encoded on the way in: code points 0x80..0xff became two bytes
  std::string out;
  for (size_t i = 0; i < h.size(); ++i) {
    const unsigned char c0 = (unsigned char)h[i];
    if (c0 < 0x80) { out.push_back((char)c0); continue; }
    TORCH_CHECK((c0 == 0xc2 || c0 == 0xc3) && i + 1 < h.size(), "IPC handle has ", h.size(), " bytes, expected ", n);
    out.push_back((char)(((c0 & 0x03) << 6) | ((unsigned char)h[++i] & 0x3f)));
  }
  TORCH_CHECK(out.size() == n, "IPC handle has ", out.size(), " bytes, expected ", n);
  return out;
}
std::string ipc_pack(const std::vector<std::string>& handles) {
  std::string raw;
  for (const auto& h : handles) raw += ipc_raw(h);
  return raw;
}

// init_custom_ar (custom_all_reduce.cu:13-41): meta = this rank's signal area followed by the two-shot scratch
int64_t init_custom_ar(torch::Tensor meta, torch::Tensor rank_data, std::vector<std::string> handles, std::vector<int64_t> offsets,
                       int64_t rank, bool full_nvlink) {
  TORCH_CHECK(meta.is_cuda() && rank_data.is_cuda(), "init_custom_ar: device tensors expected");
  const int64_t world = (int64_t)handles.size();
  TORCH_CHECK((int64_t)offsets.size() == world, "handles length should equal to offsets length");
  TORCH_CHECK(rank >= 0 && rank < world, "invalid rank passed in");
  const int64_t ms = aphro_custom_ar_meta_size(), total = meta.numel() * meta.element_size();
  TORCH_CHECK(total > ms, "meta must hold the signal area and the two-shot scratch (meta_size() + max_size bytes)");
  const size_t scratch_bytes = (size_t)((total - ms) / 16 * 16);
  const std::string raw = ipc_pack(handles);
  std::vector<int64_t> soff(offsets);
  for (auto& o : soff) o += ms;
  void* fa = nullptr;
  ok(aphro_custom_ar_init(&fa, meta.data_ptr(), raw.data(), offsets.data(), (char*)meta.data_ptr() + ms, scratch_bytes, raw.data(),
                          soff.data(), rank_data.data_ptr(), (size_t)(rank_data.numel() * rank_data.element_size()), (int)rank,
                          (int)world),
     "init_custom_ar");
  return (int64_t)fa;
}
void dispose(int64_t fa) { aphro_custom_ar_dispose((void*)fa); }   // custom_all_reduce.cu:111
// custom_all_reduce.cu:114-118
void register_buffer(int64_t fa, torch::Tensor t, std::vector<std::string> handles, std::vector<int64_t> offsets) {
  TORCH_CHECK(handles.size() == offsets.size(), "register_buffer: one offset per handle");
  const std::string raw = ipc_pack(handles);
  ok(aphro_custom_ar_register_buffer((void*)fa, t.data_ptr(), raw.data(), offsets.data()), "register_buffer");
}
// custom_all_reduce.cu:120-128: (handle bytes as int[], offsets)
std::tuple<std::vector<int64_t>, std::vector<int64_t>> get_graph_buffer_ipc_meta(int64_t fa) {
  int count = 0;
  ok(aphro_custom_ar_get_graph_buffer_ipc_meta((void*)fa, nullptr, nullptr, 0, &count), "get_graph_buffer_ipc_meta");
  const size_t hb = (size_t)aphro_ipc_handle_bytes();
  std::string raw((size_t)count * hb, '\0');
  std::vector<int64_t> offs((size_t)count);
  if (count > 0)
    ok(aphro_custom_ar_get_graph_buffer_ipc_meta((void*)fa, &raw[0], offs.data(), count, &count), "get_graph_buffer_ipc_meta");
  std::vector<int64_t> bytes(raw.size());
  for (size_t i = 0; i < raw.size(); ++i) bytes[i] = (unsigned char)raw[i];
  return {bytes, offs};
}
// custom_all_reduce.cu:130-137: handles[r] = rank r's blob (count handles back to back), offsets[r] its offsets
void register_graph_buffers(int64_t fa, std::vector<std::string> handles, std::vector<std::vector<int64_t>> offsets) {
  TORCH_CHECK(handles.size() == offsets.size(), "register_graph_buffers: handles and offsets must have one entry per rank");
  const size_t hb = (size_t)aphro_ipc_handle_bytes(), count = offsets.empty() ? 0 : offsets[0].size();
  std::string raw;
  std::vector<int64_t> flat;
  for (size_t r = 0; r < handles.size(); ++r) {
    std::string blob = handles[r];
    if (blob.size() == 2 * hb * count && count > 0) {          // hex text
      std::string dec;
      for (size_t i = 0; i < count; ++i) dec += ipc_raw(blob.substr(2 * hb * i, 2 * hb));
      blob = dec;
    }
    TORCH_CHECK(offsets[r].size() == count && blob.size() == count * hb,
                "register_graph_buffers: every rank must contribute the same number of buffers");
    raw += blob;
    flat.insert(flat.end(), offsets[r].begin(), offsets[r].end());
  }
  if (raw.empty()) raw.push_back('\0');
  if (flat.empty()) flat.push_back(0);
  ok(aphro_custom_ar_register_graph_buffers((void*)fa, raw.data(), flat.data(), (int)count), "register_graph_buffers");
}

}  // namespace

TORCH_LIBRARY_FRAGMENT(APHRO_TORCH_NS, m) {
  m.def("paged_attention_v2(Tensor! out, Tensor! exp_sums, Tensor! max_logits, Tensor! tmp_out, Tensor query, "
        "Tensor key_cache, Tensor value_cache, int num_kv_heads, float scale, Tensor block_tables, Tensor seq_lens, "
        "int block_size, int max_seq_len, Tensor? alibi_slopes, str kv_cache_dtype, float k_scale, float v_scale, "
        "int tp_rank, int blocksparse_local_blocks, int blocksparse_vert_stride, int blocksparse_block_size, "
        "int blocksparse_head_sliding_step) -> ()");                                                    // :38-49
  m.impl("paged_attention_v2", torch::kCUDA, &paged_attention_v2);
  m.def("gptq_shuffle(Tensor! q_weight, Tensor q_perm, int bit) -> ()");                                // :364-365
  m.impl("gptq_shuffle", torch::kCUDA, &gptq_shuffle);
  m.def("awq_dequantize(Tensor _kernel, Tensor _scaling_factors, Tensor _zeros, int split_k_iters, int thx, "
        "int thy) -> Tensor");                                                                          // :148-151
  m.impl("awq_dequantize", torch::kCUDA, &awq_dequantize);
  m.def("rms_norm(Tensor! out, Tensor input, Tensor weight, float epsilon) -> ()");                     // :101-105
  m.impl("rms_norm", torch::kCUDA, &rms_norm);
  m.def("fused_add_rms_norm(Tensor! input, Tensor! residual, Tensor weight, float epsilon) -> ()");     // :108-111
  m.impl("fused_add_rms_norm", torch::kCUDA, &fused_add_rms_norm);
  m.def("silu_and_mul(Tensor! out, Tensor input) -> ()");                                               // :56-57
  m.impl("silu_and_mul", torch::kCUDA, &silu_and_mul);
  m.def("rotary_embedding(Tensor positions, Tensor! query, Tensor! key, int head_size, Tensor cos_sin_cache, "
        "bool is_neox) -> ()");                                                                         // :117-121
  m.impl("rotary_embedding", torch::kCUDA, &rotary_embedding);
  m.def("static_scaled_fp8_quant(Tensor! out, Tensor input, Tensor scale) -> ()");                      // :374-376
  m.impl("static_scaled_fp8_quant", torch::kCUDA, &static_scaled_fp8_quant);
  m.def("dynamic_per_token_scaled_fp8_quant(Tensor! out, Tensor input, Tensor! scale, Tensor? scale_ub) -> ()");   // :385-390
  m.impl("dynamic_per_token_scaled_fp8_quant", torch::kCUDA, &dynamic_per_token_scaled_fp8_quant);
  m.def("advance_step_flashattn(int num_seqs, int num_queries, int block_size, Tensor! input_tokens, "
        "Tensor sampled_token_ids, Tensor! input_positions, Tensor! seq_lens, Tensor! slot_mapping, "
        "Tensor block_tables) -> ()");                                                                  // :77-82
  m.impl("advance_step_flashattn", torch::kCUDA, &advance_step_flashattn);
}

TORCH_LIBRARY_FRAGMENT(APHRO_CONCAT(APHRO_TORCH_NS, _cache_ops), m) {
  m.def("convert_fp8(Tensor! dst_cache, Tensor src_cache, float scale, str kv_cache_dtype) -> ()");     // :487-490
  m.impl("convert_fp8", torch::kCUDA, &convert_fp8);
  m.def("reshape_and_cache_flash(Tensor key, Tensor value, Tensor! key_cache, Tensor! value_cache, "
        "Tensor slot_mapping, str kv_cache_dtype, float k_scale, float v_scale) -> ()");                // :476-484
  m.impl("reshape_and_cache_flash", torch::kCUDA, &reshape_and_cache_flash);
}

TORCH_LIBRARY_FRAGMENT(APHRO_TORCH_NS, m) {
  m.def("paged_attention_v1(Tensor! out, Tensor query, Tensor key_cache, Tensor value_cache, int num_kv_heads, float scale, "
        "Tensor block_tables, Tensor seq_lens, int block_size, int max_seq_len, Tensor? alibi_slopes, str kv_cache_dtype, "
        "float k_scale, float v_scale, int tp_rank, int blocksparse_local_blocks, int blocksparse_vert_stride, "
        "int blocksparse_block_size, int blocksparse_head_sliding_step) -> ()");                        // :25-35
  m.impl("paged_attention_v1", torch::kCUDA, &paged_attention_v1);
  m.def("gptq_gemm(Tensor a, Tensor b_q_weight, Tensor b_gptq_qzeros, Tensor b_gptq_scales, Tensor b_g_idx, "
        "bool use_exllama, int bit) -> Tensor");                                                       // :357-361
  m.impl("gptq_gemm", torch::kCUDA, &gptq_gemm);
  m.def("cutlass_scaled_mm(Tensor! out, Tensor a, Tensor b, Tensor a_scales, Tensor b_scales, Tensor? bias) -> ()");  // :235-239
  m.impl("cutlass_scaled_mm", torch::kCUDA, &cutlass_scaled_mm);
}

TORCH_LIBRARY_FRAGMENT(APHRO_CONCAT(APHRO_TORCH_NS, _cache_ops), m) {
  m.def("reshape_and_cache(Tensor key, Tensor value, Tensor! key_cache, Tensor! value_cache, Tensor slot_mapping, "
        "str kv_cache_dtype, float k_scale, float v_scale) -> ()");                                     // :467-473
  m.impl("reshape_and_cache", torch::kCUDA, &reshape_and_cache);
}

// round 5: the whole-tensor quantiser, the MoE routing ops, block copies / swaps
TORCH_LIBRARY_FRAGMENT(APHRO_TORCH_NS, m) {
  m.def("dynamic_scaled_fp8_quant(Tensor! out, Tensor input, Tensor! scale) -> ()");                    // :379-382
  m.impl("dynamic_scaled_fp8_quant", torch::kCUDA, &dynamic_scaled_fp8_quant);
  m.def("moe_align_block_size(Tensor topk_ids, int num_experts, int block_size, Tensor! sorted_token_ids, "
        "Tensor! experts_ids, Tensor! num_tokens_post_pad) -> ()");                                     // :394-399
  m.impl("moe_align_block_size", torch::kCUDA, &moe_align_block_size);
}

TORCH_LIBRARY_FRAGMENT(APHRO_CONCAT(APHRO_TORCH_NS, _moe), m) {
  m.def("topk_softmax(Tensor! topk_weights, Tensor! topk_indices, Tensor! token_expert_indices, "
        "Tensor gating_output) -> ()");                                                 // kernels/moe/torch_bindings.cpp:11-14
  m.impl("topk_softmax", torch::kCUDA, &topk_softmax);
}

TORCH_LIBRARY_FRAGMENT(APHRO_CONCAT(APHRO_TORCH_NS, _cache_ops), m) {
  m.def("swap_blocks(Tensor src, Tensor! dst, Tensor block_mapping) -> ()");                            // :456-458
  m.impl("swap_blocks", torch::kCUDA, &swap_blocks);
  m.def("copy_blocks(Tensor(a!)[] key_caches, Tensor[](b!) value_caches, Tensor block_mapping) -> ()"); // :461-464
  m.impl("copy_blocks", torch::kCUDA, &copy_blocks);
}

// round 5, second batch
#ifndef APHRO_TORCH_ROCM_NS
#define APHRO_TORCH_ROCM_NS _rocm_C_mi355x      // (-DAPHRO_TORCH_ROCM_NS=_rocm_C next to -DAPHRO_TORCH_NS=_C takes the reference's place)
#endif
TORCH_LIBRARY_FRAGMENT(APHRO_TORCH_NS, m) {
  m.def("awq_gemm(Tensor _in_feats, Tensor _kernel, Tensor _scaling_factors, Tensor _zeros, int split_k_iters) -> Tensor");  // :133-135
  m.impl("awq_gemm", torch::kCUDA, &awq_gemm);
  m.def("cutlass_scaled_mm_supports_fp8(int cuda_device_capability) -> bool");                          // :250-252
  m.impl("cutlass_scaled_mm_supports_fp8", &cutlass_scaled_mm_supports_fp8);
  m.def("gptq_marlin_repack(Tensor b_q_weight, Tensor perm, SymInt size_k, SymInt size_n, int num_bits) -> Tensor");  // :204-208
  m.impl("gptq_marlin_repack", torch::kCUDA, &gptq_marlin_repack);
  m.def("awq_marlin_repack(Tensor b_q_weight, SymInt size_k, SymInt size_n, int num_bits) -> Tensor");  // :211-215
  m.impl("awq_marlin_repack", torch::kCUDA, &awq_marlin_repack);
  // (gptq_marlin_gemm: registered by aphro_torch_register_marlin below -- its schema depends on whether the process has the
  //  torchbind class _core_C.ScalarType)
  m.def("fp8_marlin_gemm(Tensor a, Tensor b_q_weight, Tensor b_scales, Tensor! workspace, int num_bits, "
        "int size_m, int size_n, int size_k) -> Tensor");                                               // :218-222
  m.impl("fp8_marlin_gemm", torch::kCUDA, &fp8_marlin_gemm);
}

TORCH_LIBRARY_FRAGMENT(APHRO_TORCH_ROCM_NS, m) {
  m.def("paged_attention(Tensor! out, Tensor exp_sums, Tensor max_logits, Tensor tmp_out, Tensor query, "
        "Tensor key_cache, Tensor value_cache, int num_kv_heads, float scale, Tensor block_tables, "
        "Tensor context_lens, int block_size, int max_context_len, Tensor? alibi_slopes, str kv_cache_dtype, "
        "float k_scale, float v_scale) -> ()");                                                         // kernels/rocm/torch_bindings.cpp:17-28
  m.impl("paged_attention", torch::kCUDA, &paged_attention_rocm);
}

TORCH_LIBRARY_FRAGMENT(APHRO_CONCAT(APHRO_TORCH_NS, _custom_ar), m) {
  m.def("all_reduce_reg(int fa, Tensor inp, Tensor! out) -> ()");                                       // :516-517
  m.impl("all_reduce_reg", torch::kCUDA, &all_reduce_reg);
  m.def("all_reduce_unreg(int fa, Tensor inp, Tensor reg_buffer, Tensor! out) -> ()");                  // :519-522
  m.impl("all_reduce_unreg", torch::kCUDA, &all_reduce_unreg);
  m.def("meta_size() -> int");                                                                          // :526
  m.impl("meta_size", &meta_size);
  m.def("init_custom_ar(Tensor meta, Tensor rank_data, str[] handles, int[] offsets, int rank, bool full_nvlink) -> int");  // :510-514
  m.impl("init_custom_ar", torch::kCUDA, &init_custom_ar);
  m.def("dispose(int fa) -> ()");                                                                       // :524
  m.impl("dispose", &dispose);
  m.def("register_buffer(int fa, Tensor t, str[] handles, int[] offsets) -> ()");                       // :528-531
  m.impl("register_buffer", torch::kCUDA, &register_buffer);
  m.def("get_graph_buffer_ipc_meta(int fa) -> (int[], int[])");                                         // :533
  m.impl("get_graph_buffer_ipc_meta", &get_graph_buffer_ipc_meta);
  m.def("register_graph_buffers(int fa, str[] handles, int[][] offsets) -> ()");                        // :535
  m.impl("register_graph_buffers", &register_graph_buffers);
}


// gptq_marlin_gemm, kernels/torch_bindings.cpp:195-201: the verbatim schema names the torchbind class _core_C.ScalarType
// (kernels/core/torch_bindings.cpp:10-13).  with_class != 0: the class resolves in this process (the reference's own _core_C, or
// csrc_torch/core_scalar_type.cpp loaded by torch_cpp.load()) -- the op is defined with the verbatim schema and a BOXED kernel
// that reads the type through its `size_bits` / `bias` properties, so it does not matter whose C++ class sits behind the name;
// with_class == 0: the type travels as its size in bits (`int b_q_type`).  Called once, after the library is loaded.
static int64_t scalar_type_property(const c10::IValue& v, const char* name) {
  auto obj = v.toObject();
  auto prop = obj->type()->getProperty(name);
  TORCH_CHECK(prop.has_value() && prop->getter != nullptr, "gptq_marlin_gemm: b_q_type has no property ", name);
  return (*prop->getter)({v}).toInt();
}

static void gptq_marlin_gemm_boxed(const c10::OperatorHandle&, c10::DispatchKeySet, torch::jit::Stack* stack) {
  constexpr size_t NARGS = 15;
  auto args = torch::jit::last(*stack, NARGS);
  const c10::IValue qt = args[7];
  const bool has_zp = args[12].toBool();
  int64_t bits;
  if (qt.isInt()) {
    bits = qt.toInt();
  } else {
    bits = scalar_type_property(qt, "size_bits");
    const int64_t bias = scalar_type_property(qt, "bias");
    // the types the Marlin role takes (quantization/utils/marlin_utils.py:28-45; awq_marlin: uint4 with zero points)
    TORCH_CHECK((bits == 4 && (bias == 8 || (bias == 0 && has_zp))) || (bits == 8 && bias == 128),
                "gptq_marlin_gemm: weight type with ", bits, " bits and bias ", bias, " is not one of uint4 (zero points), uint4b8, uint8b128");
  }
  torch::Tensor out = gptq_marlin_gemm(args[0].toTensor(), args[1].toTensor(), args[2].toTensor(), args[3].toTensor(), args[4].toTensor(),
                                       args[5].toTensor(), args[6].toTensor(), bits, args[8].toInt(), args[9].toInt(), args[10].toInt(),
                                       args[11].toBool(), has_zp, args[13].toBool(), args[14].toBool());
  torch::jit::drop(*stack, NARGS);
  torch::jit::push(*stack, std::move(out));
}

#define APHRO_STR2(x) #x
#define APHRO_STR(x) APHRO_STR2(x)
extern "C" int aphro_torch_register_marlin(int with_class) {
  static torch::Library* lib = nullptr;
  if (lib != nullptr) return 1;                      // once per process
  try {
    lib = new torch::Library(torch::Library::FRAGMENT, APHRO_STR(APHRO_TORCH_NS), c10::nullopt, __FILE__, __LINE__);
    const std::string head = "gptq_marlin_gemm(Tensor a, Tensor b_q_weight, Tensor b_scales, Tensor b_zeros, Tensor g_idx, Tensor perm, "
                             "Tensor workspace, ";
    const std::string tail = " b_q_type, int size_m, int size_n, int size_k, bool is_k_full, bool has_zp, bool use_fp32_reduce, "
                             "bool is_zp_float) -> Tensor";
    lib->def((head + (with_class ? "__torch__.torch.classes._core_C.ScalarType" : "int") + tail).c_str());
    lib->impl("gptq_marlin_gemm", torch::kCUDA, torch::CppFunction::makeFromBoxedFunction<&gptq_marlin_gemm_boxed>());
  } catch (const std::exception& e) {
    fprintf(stderr, "aphro_torch_register_marlin: %s\n", e.what());
    delete lib;
    lib = nullptr;
    return -1;
  }
  return 0;
}
