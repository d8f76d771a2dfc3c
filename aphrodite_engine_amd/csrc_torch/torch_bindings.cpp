// C++ TORCH_LIBRARY registration of the MI355X hot-path kernels -- the reference's own registration style
// (kernels/torch_bindings.cpp) for maintainers who do not want Python in the dispatch path.  Thin: every op checks its
// arguments the way the reference's host function does, allocates what the schema says it returns, and forwards
// device pointers + the current HIP stream to the C ABI (include/aphrodite_mi355x.h).  No kernels in this file.
//
// The namespace is a build-time macro (APHRO_TORCH_NS, default `_C_mi355x`): built with -DAPHRO_TORCH_NS=_C it takes the
// place of the reference's extension; the default lets both be loaded side by side (A/B runs, the tests here).
// Schemas are the reference's, verbatim (kernels/torch_bindings.cpp line numbers on each def).
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
    if (m >= 1 && m <= 32 && aphro_gptq_gemm_bits_supported(m, n, k, groups, (int)bit)) {
      ok(aphro_gptq_gemm_bits(ag.data_ptr(), ag.stride(0), (const uint32_t*)b_q_weight.data_ptr(),
                              (const uint32_t*)b_gptq_qzeros.data_ptr(), b_gptq_scales.data_ptr(), out.data_ptr(), m, n, k,
                              groups, (int)bit, act_dtype(a), cur_stream()),
         "gptq_gemm");
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

}  // namespace

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
