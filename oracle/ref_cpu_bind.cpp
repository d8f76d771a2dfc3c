// TEST INFRASTRUCTURE ONLY -- never linked into or loaded by the product path.
//
// Binding shim (our own code) that exposes the REFERENCE's CPU kernels --
// compiled where they lie under /root/reference/kernels/cpu/*.cpp by
// oracle/Makefile, never copied into this repo -- as torch ops in the
// namespace `aphro_ref_cpu`, so tests and bench.py's `cpu_baseline` leg can
// call the reference implementation itself:
//   paged_attention_v1 / v2   kernels/cpu/attention.cpp:421-441, 739-758
//   reshape_and_cache         kernels/cpu/cache.cpp:107-132
//   copy_blocks               kernels/cpu/cache.cpp:88-105
//   silu_and_mul              kernels/cpu/activation.cpp:87
//   rms_norm/fused_add_rms_norm  kernels/cpu/layernorm.cpp:90,104
//   rotary_embedding          kernels/cpu/pos_encoding.cpp:170
// The reference's own registration file (kernels/cpu/torch_bindings.cpp) also
// pulls oneDNN int8 code that cannot build here, hence this shim.
#include <torch/library.h>
#include <torch/all.h>
#include <string>

void paged_attention_v1(
    torch::Tensor& out, torch::Tensor& query, torch::Tensor& key_cache,
    torch::Tensor& value_cache, int64_t num_kv_heads, double scale,
    torch::Tensor& block_tables, torch::Tensor& seq_lens, int64_t block_size,
    int64_t max_seq_len, const c10::optional<torch::Tensor>& alibi_slopes,
    const std::string& kv_cache_dtype, double k_scale, double v_scale,
    const int64_t tp_rank, const int64_t blocksparse_local_blocks,
    const int64_t blocksparse_vert_stride, const int64_t blocksparse_block_size,
    const int64_t blocksparse_head_sliding_step);

void paged_attention_v2(
    torch::Tensor& out, torch::Tensor& exp_sums, torch::Tensor& max_logits,
    torch::Tensor& tmp_out, torch::Tensor& query, torch::Tensor& key_cache,
    torch::Tensor& value_cache, int64_t num_kv_heads, double scale,
    torch::Tensor& block_tables, torch::Tensor& seq_lens, int64_t block_size,
    int64_t max_seq_len, const c10::optional<torch::Tensor>& alibi_slopes,
    const std::string& kv_cache_dtype, double k_scale, double v_scale,
    const int64_t tp_rank, const int64_t blocksparse_local_blocks,
    const int64_t blocksparse_vert_stride, const int64_t blocksparse_block_size,
    const int64_t blocksparse_head_sliding_step);

void reshape_and_cache(torch::Tensor& key, torch::Tensor& value,
                       torch::Tensor& key_cache, torch::Tensor& value_cache,
                       torch::Tensor& slot_mapping,
                       const std::string& kv_cache_dtype, double k_scale,
                       double v_scale);

void copy_blocks(std::vector<torch::Tensor> const& key_caches, std::vector<torch::Tensor> const& value_caches,
                 const torch::Tensor& block_mapping);

void silu_and_mul(torch::Tensor& out, torch::Tensor& input);
void rms_norm(torch::Tensor& out, torch::Tensor& input, torch::Tensor& weight,
              double epsilon);
void fused_add_rms_norm(torch::Tensor& input, torch::Tensor& residual,
                        torch::Tensor& weight, double epsilon);
void rotary_embedding(torch::Tensor& positions, torch::Tensor& query,
                      torch::Tensor& key, int64_t head_size,
                      torch::Tensor& cos_sin_cache, bool is_neox);

TORCH_LIBRARY(aphro_ref_cpu, ops) {
  ops.def(
      "paged_attention_v1(Tensor! out, Tensor query, Tensor key_cache, Tensor "
      "value_cache, int num_kv_heads, float scale, Tensor block_tables, Tensor "
      "seq_lens, int block_size, int max_seq_len, Tensor? alibi_slopes, str "
      "kv_cache_dtype, float k_scale, float v_scale, int tp_rank, int "
      "blocksparse_local_blocks, int blocksparse_vert_stride, int "
      "blocksparse_block_size, int blocksparse_head_sliding_step) -> ()");
  ops.impl("paged_attention_v1", torch::kCPU, &paged_attention_v1);
  ops.def(
      "paged_attention_v2(Tensor! out, Tensor! exp_sums, Tensor! max_logits, "
      "Tensor! tmp_out, Tensor query, Tensor key_cache, Tensor value_cache, "
      "int num_kv_heads, float scale, Tensor block_tables, Tensor seq_lens, "
      "int block_size, int max_seq_len, Tensor? alibi_slopes, str "
      "kv_cache_dtype, float k_scale, float v_scale, int tp_rank, int "
      "blocksparse_local_blocks, int blocksparse_vert_stride, int "
      "blocksparse_block_size, int blocksparse_head_sliding_step) -> ()");
  ops.impl("paged_attention_v2", torch::kCPU, &paged_attention_v2);
  ops.def(
      "reshape_and_cache(Tensor key, Tensor value, Tensor! key_cache, Tensor! "
      "value_cache, Tensor slot_mapping, str kv_cache_dtype, float k_scale, "
      "float v_scale) -> ()");
  ops.impl("reshape_and_cache", torch::kCPU, &reshape_and_cache);
  ops.def("copy_blocks(Tensor(a!)[] key_caches, Tensor[](b!) value_caches, Tensor block_mapping) -> ()");
  ops.impl("copy_blocks", torch::kCPU, &copy_blocks);
  ops.def("silu_and_mul(Tensor! out, Tensor input) -> ()");
  ops.impl("silu_and_mul", torch::kCPU, &silu_and_mul);
  ops.def(
      "rms_norm(Tensor! out, Tensor input, Tensor weight, float epsilon) -> ()");
  ops.impl("rms_norm", torch::kCPU, &rms_norm);
  ops.def(
      "fused_add_rms_norm(Tensor! input, Tensor! residual, Tensor weight, "
      "float epsilon) -> ()");
  ops.impl("fused_add_rms_norm", torch::kCPU, &fused_add_rms_norm);
  ops.def(
      "rotary_embedding(Tensor positions, Tensor! query, Tensor! key, int "
      "head_size, Tensor cos_sin_cache, bool is_neox) -> ()");
  ops.impl("rotary_embedding", torch::kCPU, &rotary_embedding);
}
