/*
 * aphrodite_mi355x.h -- C ABI of libaphrodite_mi355x.so
 *
 * MI355X (gfx950 / CDNA4) native kernels for the quantized-inference hot path
 * of PygmalionAI/aphrodite-engine.  Plain pointers (device memory unless
 * noted), sizes and a hipStream_t (passed as void*) -- no torch types.  Every
 * entry point is asynchronous on `stream`, allocates nothing, never
 * synchronises (HIP-graph capturable) and returns APHRO_OK or a negative
 * error code; aphro_last_error() describes the failure (thread-local).
 *
 * Each function names the reference interface it replaces
 * (file:line relative to the reference tree, kernels/torch_bindings.cpp = the
 * op schema, kernels/...cu = the launcher).
 */
#ifndef APHRODITE_MI355X_H_
#define APHRODITE_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APHRO_OK 0
#define APHRO_ERR_INVALID (-1)  /* unsupported shape / dtype / argument     */
#define APHRO_ERR_LAUNCH (-2)   /* hipLaunch failure                        */
#define APHRO_ERR_WORKSPACE (-3) /* caller workspace too small              */

/* activation / output dtypes */
#define APHRO_F16 0
#define APHRO_BF16 1
#define APHRO_F32 2
/* KV-cache dtypes ("auto" | "fp8"=="fp8_e4m3" | "fp8_e5m2"), OCP encodings */
#define APHRO_KV_AUTO 0
#define APHRO_KV_FP8_E4M3 1
#define APHRO_KV_FP8_E5M2 2

const char* aphro_last_error(void);
int aphro_abi_version(void);
/* The library reads its environment switches ONCE (at load; the list is csrc/common.h `Knobs` = INTEGRATION.md "Switches") and
 * never on a launch path.  A process that changes one afterwards -- the tests do -- calls this to have them read again. */
void aphro_reload_env(void);

/* ------------------------------------------------------------------------
 * GPTQ 4-bit
 * ---------------------------------------------------------------------- */

/* _C::gptq_shuffle(Tensor! q_weight, Tensor q_perm, int bit)
 *   kernels/torch_bindings.cpp:364-365, quantization/gptq/q_gemm.cu:2263-2277.
 * In-place exllama repack of q_weight int32 [K/8, N]: optional row gather by
 * q_perm (int32 [K], act-order) then the per-word nibble permutation of
 * qdq_4.cuh:17-35.  `tmp` (K/8*N words) is required only when q_perm != NULL
 * (the reference cudaMallocs it; we take it from the caller).  bit must be 4. */
int aphro_gptq_shuffle(uint32_t* q_weight, const int32_t* q_perm, int64_t size_k,
                       int64_t size_n, int bit, uint32_t* tmp, void* stream);

/* Out-of-place variant of the above: the gptq_marlin_repack role
 * (kernels/torch_bindings.cpp:204-208) for our CDNA4 K-packed layout. */
int aphro_gptq_repack(const uint32_t* q_weight, const int32_t* q_perm,
                      uint32_t* out, int64_t size_k, int64_t size_n, int bit,
                      void* stream);

/* _C::gptq_gemm(Tensor a, Tensor b_q_weight, Tensor b_gptq_qzeros,
 *               Tensor b_gptq_scales, Tensor b_g_idx, bool use_exllama, int bit)
 *   kernels/torch_bindings.cpp:357-361, q_gemm.cu:2238-2261  (exllama path).
 * c[M,N] = a[M,K] . dequant(W), W = (q - (qzero+1)) * scale, group = k/(K/groups)
 * a: [M,K] row stride lda (elements), dtype f16|bf16; q_weight: exllama-shuffled
 * int32 [K/8,N]; qzeros int32 [groups, N/8]; scales dtype [groups,N];
 * perm: int32 [K] or NULL (act-order: column gather of a, q_gemm.cu:219-226),
 * needs a_perm_tmp [M,K] dtype.  workspace: fp32 split-K partials, size from
 * aphro_wna16_workspace_bytes().  zero_offset: 1 for GPTQ v1 checkpoints
 * (stored zero - 1), 0 for AWQ-style zero points in the same K-packed layout.
 * M <= APHRO_WNA16_MAX_M rows per call (larger M: caller loops or uses
 * aphro_gptq_dequant + a library GEMM, as the reference does for M > 50,
 * q_gemm.cu:1529-1544). */
#define APHRO_WNA16_MAX_M 64
int aphro_gptq_gemm(const void* a, const uint32_t* q_weight, const uint32_t* qzeros,
                    const void* scales, const int32_t* perm, void* a_perm_tmp,
                    void* c, void* workspace, size_t workspace_bytes, int64_t M,
                    int64_t N, int64_t K, int64_t groups, int64_t lda,
                    int zero_offset, int dtype, void* stream);
size_t aphro_wna16_workspace_bytes(int64_t M, int64_t N, int64_t K);

/* The fast W4A16 kernel consumes the activations FRAGMENT-MAJOR (f16): block
 * (seg = k/128, u = (k%32)/8, mtile = m/16) is 1 KiB, lane (g = (k%128)/32, m%16)
 * holds the 8 halfs A[m][k..k+7].  aphro_gptq_gemm packs internally; the decode
 * fast path packs once in the producer (fused norm / activation / attention
 * epilogue) and calls aphro_wna16_gemm_packed.  With c == NULL the fp32 split-K
 * slabs [ksplit][M][N] are left in `partials` for a fused consumer
 * (aphro_fused_add_rms_norm_pack, aphro_rope_cache) -- the gptq_marlin_gemm role
 * (kernels/torch_bindings.cpp:195-201) with use_fp32_reduce, re-designed. */
size_t aphro_wna16_packed_a_bytes(int64_t M, int64_t K);
int aphro_wna16_pack_a(const void* a, const int32_t* perm, void* packed, int64_t M,
                       int64_t K, int64_t lda, int dtype, void* stream);
int aphro_wna16_ksplit(int64_t M, int64_t N, int64_t K, int64_t groups);
int aphro_wna16_gemm_packed(const void* a_packed, const uint32_t* q_weight,
                            const uint32_t* qzeros, const void* scales, void* c,
                            float* partials, size_t partial_bytes, int64_t M, int64_t N,
                            int64_t K, int64_t groups, int zero_offset, int dtype,
                            void* stream);
/* gate_up projection with SiluAndMul (modeling/layers/activation.py SiluAndMul ->
 * kernels/activation_kernels.cu:12-75) and the activation pack fused into the GEMM
 * epilogue.  Weight columns are INTERLEAVED at load time (column 2j = gate_j,
 * 2j+1 = up_j; qzeros and scales permuted alike) so both halves of an output
 * feature meet in one lane.  act_packed: fragment-major f16 [M, N/2].  Served only
 * when aphro_wna16_ksplit(M,N,K,groups) == 1; roundings identical to gptq_gemm ->
 * silu_and_mul. */
int aphro_wna16_gemm_silu_pack(const void* a_packed, const uint32_t* q_weight,
                               const uint32_t* qzeros, const void* scales,
                               void* act_packed, int64_t M, int64_t N, int64_t K,
                               int64_t groups, int zero_offset, int dtype, void* stream);

/* Reconstruct W[K,N] (dtype f16|bf16) from a GPTQ tensor set
 *   q_gemm.cu:1394-1434 (reconstruct_gptq, shuffled=0, g_idx int32 [K] or NULL)
 *   q_gemm.cu:856-965   (reconstruct_exllama, shuffled=1, perm ignored: rows are
 *                        in the permuted order, exactly as the reference). */
int aphro_gptq_dequant(const uint32_t* q_weight, const uint32_t* qzeros,
                       const void* scales, const int32_t* g_idx, void* out,
                       int64_t K, int64_t N, int64_t groups, int shuffled,
                       int zero_offset, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * AWQ 4-bit (row a8)
 * ---------------------------------------------------------------------- */

/* _C::awq_dequantize(Tensor _kernel, Tensor _scaling_factors, Tensor _zeros,
 *                    int split_k_iters, int thx, int thy)
 *   kernels/torch_bindings.cpp:148-151, awq/gemm_kernels.cu:720-776.
 * out[K,N] f16 = (q - z) * s; qweight int32 [K,N/8] (AWQ nibble order),
 * qzeros int32 [groups,N/8], scales f16 [groups,N]. */
int aphro_awq_dequantize(const uint32_t* qweight, const void* scales,
                         const uint32_t* qzeros, void* out, int64_t K, int64_t N,
                         int64_t groups, int dtype, void* stream);

/* _C::awq_gemm(Tensor _in_feats, Tensor _kernel, Tensor _scaling_factors,
 *              Tensor _zeros, int split_k_iters)
 *   kernels/torch_bindings.cpp:142-145, awq/gemm_kernels.cu:784-841.
 * c[M,N] = a[M,K] . ((q - z) * s).  Positional order as the C++ op. */
size_t aphro_awq_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t groups);
int aphro_awq_gemm(const void* a, const uint32_t* qweight, const void* scales,
                   const uint32_t* qzeros, void* c, void* workspace,
                   size_t workspace_bytes, int64_t M, int64_t N, int64_t K,
                   int64_t groups, int64_t lda, int dtype, void* stream);

/* awq_marlin_repack role (kernels/torch_bindings.cpp:211-215): AWQ [K,N/8]
 * qweight -> CDNA4 K-packed exllama-order [K/8,N]; AWQ qzeros [G,N/8] ->
 * plain column order [G,N/8] (nibble j of word c = column 8c+j). */
int aphro_awq_repack(const uint32_t* qweight, uint32_t* out, int64_t K, int64_t N,
                     void* stream);
int aphro_awq_repack_zeros(const uint32_t* qzeros, uint32_t* out, int64_t groups,
                           int64_t N, void* stream);

/* ------------------------------------------------------------------------
 * Paged KV cache + decode attention (rows a1-a4)
 * ---------------------------------------------------------------------- */

/* _C_cache_ops::reshape_and_cache(Tensor key, Tensor value, Tensor! key_cache,
 *     Tensor! value_cache, Tensor slot_mapping, str kv_cache_dtype,
 *     float k_scale, float v_scale)
 *   kernels/torch_bindings.cpp:467-473, kernels/cache_kernels.cu:263-289.
 * key/value [T,Hkv,hd] with token strides (elements); key_cache
 * [NB,Hkv,hd/x,block,x], value_cache [NB,Hkv,hd,block]; slot_mapping int64 [T]
 * (-1 = padding). */
int aphro_reshape_and_cache(const void* key, const void* value, void* key_cache,
                            void* value_cache, const int64_t* slot_mapping,
                            int64_t num_tokens, int num_kv_heads, int head_size,
                            int block_size, int x, int64_t key_stride,
                            int64_t value_stride, int dtype, int kv_dtype,
                            float k_scale, float v_scale, void* stream);

/* _C_cache_ops::reshape_and_cache_flash(Tensor key, Tensor value, Tensor! key_cache,
 *     Tensor! value_cache, Tensor slot_mapping, str kv_cache_dtype, float k_scale,
 *     float v_scale)   kernels/torch_bindings.cpp:476-484, cache_kernels.cu:207-330.
 * Caches in the flash layout [NB, block, H, hd]; block_stride = key_cache.stride(0)
 * in elements. */
int aphro_reshape_and_cache_flash(const void* key, const void* value, void* key_cache,
                                  void* value_cache, const int64_t* slot_mapping,
                                  int64_t num_tokens, int num_heads, int head_size,
                                  int block_size, int64_t block_stride,
                                  int64_t key_stride, int64_t value_stride, int dtype,
                                  int kv_dtype, float k_scale, float v_scale,
                                  void* stream);

/* _C_cache_ops::copy_blocks(Tensor(a!)[] key_caches, Tensor[](b!) value_caches,
 *     Tensor block_mapping)   kernels/torch_bindings.cpp:461-464, cache_kernels.cu:66-148.
 * key/value_cache_ptrs: DEVICE arrays of num_layers base addresses; block_mapping:
 * device int64 [num_pairs, 2] (src, dst); block_bytes = bytes of one cache block. */
int aphro_copy_blocks(const int64_t* key_cache_ptrs, const int64_t* value_cache_ptrs,
                      int num_layers, const int64_t* block_mapping, int64_t num_pairs,
                      int64_t block_bytes, void* stream);

/* _C_cache_ops::swap_blocks(Tensor src, Tensor! dst, Tensor block_mapping)
 *   kernels/torch_bindings.cpp:456-458, cache_kernels.cu:24-63.
 * block_mapping_host: HOST int64 [num_pairs, 2]; kind 0 = device->device,
 * 1 = device->host, 2 = host->device; async copies on `stream`. */
int aphro_swap_blocks(const void* src, void* dst, const int64_t* block_mapping_host,
                      int64_t num_pairs, int64_t block_bytes, int kind, void* stream);

/* _C_cache_ops::convert_fp8(Tensor! dst_cache, Tensor src_cache, float scale,
 *                           str kv_cache_dtype)
 *   kernels/torch_bindings.cpp:487-490, cache_kernels.cu:356-409.
 * to_fp8 != 0: dst uint8 = fp8(src / scale); else dst = float(src fp8) * scale. */
int aphro_convert_fp8(void* dst, const void* src, int64_t numel, float scale,
                      int hp_dtype, int kv_dtype, int to_fp8, void* stream);

/* _C::paged_attention_v1 / _C::paged_attention_v2 / _rocm_C::paged_attention
 *   kernels/torch_bindings.cpp:25-49, kernels/rocm/torch_bindings.cpp:17-28,
 *   launchers attention/attention_kernels.cu:809-830, 974-998,
 *   rocm/attention.cu:1077-1117.
 * out [S,Hq,hd] (row stride = Hq*hd); query [S,Hq,hd] with seq stride q_stride;
 * caches as above with block stride kv_block_stride and head stride
 * kv_head_stride (elements of the cache dtype).  partition_size == 0 selects
 * the v1 form (no scratch); otherwise exp_sums/max_logits [S,Hq,P] fp32 and
 * tmp_out [S,Hq,P,hd] (dtype) with P = ceil(max_seq_len/partition_size) are
 * written exactly as the reference defines them (attention_kernels.cu:350-358)
 * and merged into out.  alibi_slopes fp32 [Hq] or NULL. */
int aphro_paged_attention(void* out, float* exp_sums, float* max_logits,
                          void* tmp_out, const void* query, const void* key_cache,
                          const void* value_cache, int num_seqs, int num_heads,
                          int num_kv_heads, int head_size, float scale,
                          const int32_t* block_tables, const int32_t* seq_lens,
                          int max_num_blocks_per_seq, int block_size,
                          int max_seq_len, const float* alibi_slopes,
                          int64_t q_stride, int64_t kv_block_stride,
                          int64_t kv_head_stride, int dtype, int kv_dtype,
                          float k_scale, float v_scale, int partition_size,
                          void* stream);

/* Same kernel in its single-launch (v1) form, additionally writing the output
 * fragment-major (f16) for the o_proj GEMM of the decode fast path; `out` may be
 * NULL. */
int aphro_paged_attention_packed(void* out, void* out_packed, const void* query,
                                 const void* key_cache, const void* value_cache,
                                 int num_seqs, int num_heads, int num_kv_heads,
                                 int head_size, float scale, const int32_t* block_tables,
                                 const int32_t* seq_lens, int max_num_blocks_per_seq,
                                 int block_size, int max_seq_len, const float* alibi_slopes,
                                 int64_t q_stride, int64_t kv_block_stride,
                                 int64_t kv_head_stride, int dtype, int kv_dtype,
                                 float k_scale, float v_scale, void* stream);

/* Decode fast path, one launch: [qkv GEMM split-K slab reduce] + rotary_embedding
 * (NeoX, rot_dim == head_size == 128; pos_encoding_kernels.cu:10-160) +
 * reshape_and_cache (cache_kernels.cu:152-204) + paged attention (v1 form) +
 * fragment-major output.  qkv_slabs fp32 [nslab][num_seqs][(Hq+2Hkv)*hd] as left by
 * aphro_wna16_gemm_packed(c = NULL); positions / slot_mapping int64 [num_seqs];
 * cos_sin_cache [max_pos, hd] in the activation dtype (positions == NULL: cos_sin_cache
 * is already gathered, row i belongs to sequence i).  The new token's K/V are
 * WRITTEN to the caches (same roundings as the separate ops) before being read. */
int aphro_paged_attention_rope_packed(void* out, void* out_packed, const float* qkv_slabs,
                                      int nslab, const int64_t* positions,
                                      const void* cos_sin_cache, const int64_t* slot_mapping,
                                      void* key_cache, void* value_cache, int num_seqs,
                                      int num_heads, int num_kv_heads, int head_size,
                                      float scale, const int32_t* block_tables,
                                      const int32_t* seq_lens, int max_num_blocks_per_seq,
                                      int block_size, int max_seq_len,
                                      const float* alibi_slopes, int64_t kv_block_stride,
                                      int64_t kv_head_stride, int dtype, int kv_dtype,
                                      float k_scale, float v_scale, void* stream);

/* ------------------------------------------------------------------------
 * FP8 activations + GEMMs (rows a9-a11)
 * ---------------------------------------------------------------------- */

/* _C::static_scaled_fp8_quant / dynamic_scaled_fp8_quant /
 * dynamic_per_token_scaled_fp8_quant
 *   kernels/torch_bindings.cpp:374-390, quantization/fp8/common.cu:258-321.
 * out uint8 (OCP e4m3fn) [M,K]; input dtype f16|bf16|f32 [M,K] contiguous. */
int aphro_static_scaled_fp8_quant(void* out, const void* input, const float* scale,
                                  int64_t M, int64_t K, int dtype, void* stream);
/* aphro_dynamic_scaled_fp8_quant with caller scratch (>= 8 KiB) instead of a zero-initialised
 * scale and atomics: identical scale and bytes (max is order independent), two launches instead
 * of fill + absmax + quant.  Used by ops.scaled_fp8_quant(scale=None). */
int aphro_dynamic_scaled_fp8_quant_ws(void* out, const void* input, float* scale,
                                      float* partials, size_t partials_bytes, int64_t M,
                                      int64_t K, int dtype, void* stream);

/* scale must be zero-initialised by the caller (as _custom_ops.py:676 does). */
int aphro_dynamic_scaled_fp8_quant(void* out, const void* input, float* scale,
                                   int64_t M, int64_t K, int dtype, void* stream);
int aphro_dynamic_per_token_scaled_fp8_quant(void* out, const void* input,
                                             float* scales, const float* scale_ub,
                                             int64_t M, int64_t K, int dtype,
                                             void* stream);

/* _C::cutlass_scaled_mm(Tensor! out, Tensor a, Tensor b, Tensor a_scales,
 *                       Tensor b_scales, Tensor? bias)
 *   kernels/torch_bindings.cpp:235-239, cutlass_w8a8/scaled_mm_entry.cu:92-137.
 * out[M,N] (f16|bf16) = a_scales * (a[M,K] . b[K,N]) * b_scales + bias.
 * a: e4m3 [M,K] row-major; b: e4m3 column-major [K,N] == row-major [N,K];
 * a_scales fp32 [1] or [M]; b_scales fp32 [1] or [N]; bias (out dtype) [N]|NULL. */
int aphro_scaled_mm_fp8(void* out, const void* a, const void* b,
                        const float* a_scales, const float* b_scales,
                        const void* bias, void* workspace, size_t workspace_bytes,
                        int64_t M, int64_t N, int64_t K, int a_scale_per_token,
                        int b_scale_per_channel, int out_dtype, void* stream);

/* The same op for prefill-sized M (meant for M > 64; any M): MFMA-bound kernel on
 * v_mfma_scale_f32_32x32x64_f8f6f4, both tiles through LDS (fp8_gemm_large.hip) -- the
 * cutlass sm89/sm90 kernels' role (cutlass_w8a8/scaled_mm_c3x.cu, scaled_mm_c2x.cu), which
 * on ROCm the reference hands to torch._scaled_mm (w8a8_utils.py:130-183).
 * N % 128 == 0, K % 128 == 0.  workspace: aphro_scaled_mm_fp8_large_workspace_bytes (split-K
 * slabs for small grids; may be 0). */
size_t aphro_scaled_mm_fp8_large_workspace_bytes(int64_t M, int64_t N, int64_t K);
int aphro_scaled_mm_fp8_large(void* out, const void* a, const void* b,
                              const float* a_scales, const float* b_scales,
                              const void* bias, void* workspace, size_t workspace_bytes,
                              int64_t M, int64_t N, int64_t K, int a_scale_per_token,
                              int b_scale_per_channel, int out_dtype, void* stream);

/* _C::fp8_marlin_gemm role (kernels/torch_bindings.cpp:218-222,
 * quantization/fp8/fp8_marlin.cu:1212): W8A16, c = a . (fp8->hp(W) * s_n).
 * a f16|bf16 [M,K]; w e4m3 row-major [N,K]; w_scales fp32 [1] or [N]. */
int aphro_fp8_w8a16_gemm(void* out, const void* a, const void* w,
                         const float* w_scales, const void* bias, void* workspace,
                         size_t workspace_bytes, int64_t M, int64_t N, int64_t K,
                         int64_t lda, int w_scale_per_channel, int dtype,
                         void* stream);

/* The same op for prefill-sized M (meant for M > 64): the tile machine of aphro_wna16_gemm_large with e4m3 weights widened
 * to f16 in registers and the scale applied in the epilogue (wna16_gemm_large.hip).  N % 128 == 0, K % 64 == 0.
 * workspace: the saturating f16 copy of bf16 activations + fp32 split-K slabs for small grids (may be 0). */
size_t aphro_fp8_w8a16_gemm_large_workspace_bytes(int64_t M, int64_t N, int64_t K, int dtype);
int aphro_fp8_w8a16_gemm_large(void* out, const void* a, const void* w, const float* w_scales, const void* bias,
                               void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K, int64_t lda,
                               int w_scale_per_channel, int dtype, void* stream);
size_t aphro_fp8_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);

/* ------------------------------------------------------------------------
 * Glue between the hot kernels
 * ---------------------------------------------------------------------- */
/* _C::rms_norm / fused_add_rms_norm  kernels/layernorm_kernels.cu:282-352 */
int aphro_rms_norm(void* out, const void* input, const void* weight, float eps,
                   int64_t num_tokens, int hidden, int64_t in_stride, int dtype,
                   void* stream);
int aphro_fused_add_rms_norm(void* input, void* residual, const void* weight,
                             float eps, int64_t num_tokens, int hidden, int dtype,
                             void* stream);
/* _C::silu_and_mul  kernels/activation_kernels.cu:55-75 : out[T,d] from in[T,2d] */
int aphro_silu_and_mul(void* out, const void* input, int64_t num_tokens, int d,
                       int dtype, void* stream);
/* SiluAndMul over a gate_up output with INTERLEAVED (gate_j, up_j) columns -- the column order a layer's weights have once
 * SiluAndMul rides in the decode GEMM's epilogue: prompt-sized batches run on the same single copy of the weights.  Same
 * bits as aphro_silu_and_mul on the de-interleaved input (kernels/activation_kernels.cu:12-75).  csrc/glue.hip. */
int aphro_silu_and_mul_interleaved(void* out, const void* input, int64_t num_tokens, int d,
                       int dtype, void* stream);
/* _C::rotary_embedding  kernels/pos_encoding_kernels.cu:120-160 (in place) */
int aphro_rotary_embedding(const int64_t* positions, void* query, void* key,
                           int64_t num_tokens, int num_heads, int num_kv_heads,
                           int head_size, int rot_dim, const void* cos_sin_cache,
                           int64_t query_stride, int64_t key_stride, int is_neox,
                           int dtype, void* stream);

/* Fused glue of the decode fast path.  Each reproduces the
 * unfused op sequence bit for bit:
 *  fused_add_rms_norm_pack: x = input (dtype) or the sum of `nslab` fp32 slabs
 *    [nslab][T][hidden] rounded to dtype; residual' = x + residual (in place;
 *    has_residual == 0: residual' = x, stored if residual != NULL);
 *    y = rms_norm(residual') * weight -> `packed` (fragment-major f16) and/or `out`.
 *    (_C::fused_add_rms_norm, kernels/layernorm_kernels.cu:200-240)
 *  silu_and_mul_pack: _C::silu_and_mul -> packed and/or row-major.
 *  rope_cache: qkv row (dtype, or fp32 slabs) -> _C::rotary_embedding on q,k then
 *    _C_cache_ops::reshape_and_cache; q_out [T, Hq*hd] (dtype). */
int aphro_fused_add_rms_norm_pack(const void* input, const float* slabs, int nslab,
                                  void* residual, int has_residual, const void* weight,
                                  float eps, void* packed, void* out, int64_t tokens,
                                  int hidden, int dtype, void* stream);
int aphro_silu_and_mul_pack(const void* input, void* packed, void* out, int64_t tokens,
                            int d, int dtype, void* stream);
/* the same on the gate_up GEMM's fp32 split-K slabs [nslab][tokens][2 d] (slab order, one rounding to dtype: the
 * splitk reduce rides in this launch) */
int aphro_silu_and_mul_pack_slabs(const float* slabs, int nslab, void* packed, void* out, int64_t tokens,
                                  int d, int dtype, void* stream);
int aphro_rope_cache(const void* qkv, int64_t qkv_stride, const float* slabs, int nslab,
                     const int64_t* positions, const void* cos_sin_cache, int rot_dim,
                     int is_neox, void* q_out, void* key_cache, void* value_cache,
                     const int64_t* slot_mapping, int64_t tokens, int num_heads,
                     int num_kv_heads, int head_size, int block_size, int x, int dtype,
                     int kv_dtype, float k_scale, float v_scale, void* stream);

/* ------------------------------------------------------------------------
 * Prefill attention (row a5): causal varlen flash attention on MFMA.
 *   replaces attention/ops/triton_flash_attn.py:700-820 / CK
 *   flash_attn_varlen_func as called at backends/rocm_flash_attn.py:455-508.
 * q [T,Hq,hd], k/v [T,Hkv,hd] with token strides (elements), out [T,Hq,hd]
 * contiguous; cu_seqlens int32 [B+1] (same for q and k: self-attention);
 * alibi_slopes fp32 [Hq] or NULL (bias = slope * (key_pos - query_pos)). */
int aphro_flash_attn_varlen(void* out, const void* q, const void* k, const void* v,
                            const int32_t* cu_seqlens, int batch, int max_seqlen,
                            int num_heads, int num_kv_heads, int head_size,
                            int64_t q_stride, int64_t k_stride, int64_t v_stride,
                            float scale, int causal, const float* alibi_slopes,
                            int dtype, void* stream);
/* The same op with a sliding window (Mistral-style checkpoints): window > 0, causal only -- query i sees keys j with
 * i - j < window.  The reference hands its window to flash_attn_varlen_func as window_size = (sliding_window,
 * sliding_window) under causal = True (backends/rocm_flash_attn.py:321-322, 497-507): keys i - sliding_window .. i, i.e.
 * window = sliding_window + 1 here.  window <= 0: aphro_flash_attn_varlen. */
int aphro_flash_attn_varlen_window(void* out, const void* q, const void* k, const void* v,
                                   const int32_t* cu_seqlens, int batch, int max_seqlen,
                                   int num_heads, int num_kv_heads, int head_size,
                                   int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                   float scale, int causal, const float* alibi_slopes,
                                   int window, int dtype, void* stream);

/* Prefill with cached context -- the context_attention_fwd role
 *   attention/ops/prefix_prefill.py:696-858 (kernel :58-255), called through
 *   PagedAttention.forward_prefix (ops/paged_attn.py:192-231) from
 *   backends/rocm_flash_attn.py:509-527.
 * q/k/v: the NEW tokens [T,H,hd] with token strides (elements); k_cache
 * [NB,Hkv,hd/x,block,x], v_cache [NB,Hkv,hd,block] (uint8 for fp8: dequantised
 * as float(fp8) * scale, rounded to the query dtype); block_tables int32
 * [B,max_blocks]; q_start_loc int32 [B+1]; seq_lens (context + new) and ctx_lens
 * int32 [B].  sliding_window 0 = off.  NOTE: the reference kernel ignores the
 * layer's scale and uses 1/sqrt(hd) (prefix_prefill.py:745); `scale` here is
 * explicit and the Python mirror passes 1/sqrt(hd). */
int aphro_context_attention(void* out, const void* q, const void* k, const void* v,
                            const void* k_cache, const void* v_cache,
                            const int32_t* block_tables, const int32_t* q_start_loc,
                            const int32_t* seq_lens, const int32_t* ctx_lens, int batch,
                            int max_query_len, int max_blocks, int num_heads,
                            int num_kv_heads, int head_size, int block_size, int x,
                            int64_t q_stride, int64_t k_stride, int64_t v_stride,
                            int64_t o_stride, float scale, float k_scale, float v_scale,
                            const float* alibi_slopes, int sliding_window, int dtype,
                            int kv_dtype, void* stream);

/* The same op for long prompts (head_size 128, no sliding window): the cached context is gathered once into contiguous
 * rows (workspace) and the third-generation prefill kernel (256-row workgroups, 32x32 MFMA, direct-to-LDS K/V) runs over
 * context + new tokens.  max_seq_len = max(seq_lens), total_kv_tokens >= sum(seq_lens).  Round 3: every head size of
 * aphro_context_attention (64 / 96 / 128 / 256), ALiBi and sliding_window (> 0: keys within the window, prefix_prefill.py:
 * 756-760) -- the second / first generation tile machines take the shapes the third does not. */
size_t aphro_context_attention_workspace_bytes(int64_t total_kv_tokens, int batch, int num_kv_heads, int head_size);
int aphro_context_attention_gathered(void* out, const void* q, const void* k, const void* v, const void* k_cache,
                                     const void* v_cache, const int32_t* block_tables, const int32_t* q_start_loc,
                                     const int32_t* seq_lens, const int32_t* ctx_lens, int batch, int max_query_len,
                                     int max_seq_len, int64_t total_kv_tokens, int max_blocks, int num_heads,
                                     int num_kv_heads, int head_size, int block_size, int x, int64_t q_stride,
                                     int64_t k_stride, int64_t v_stride, int64_t o_stride, float scale, float k_scale,
                                     float v_scale, const float* alibi_slopes, int sliding_window, int dtype,
                                     int kv_dtype, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * FP8 (W8A8, per-token dynamic activation scale) decode fast path -- configs[2],
 * compressed_tensors_w8a8_fp8.py:139-148.  Bit-identical to the op sequences they replace.
 * ---------------------------------------------------------------------- */

/* W8A8 GEMM leaving raw fp32 accumulators as split-K slabs [ksplit][M][N]
 * (ksplit = aphro_fp8_gemm_ksplit) for a fused consumer that applies
 * a_scale * (b_scale * acc) -- cutlass_scaled_mm without its epilogue. */
int aphro_fp8_gemm_ksplit(int64_t M, int64_t N, int64_t K);
int aphro_scaled_mm_fp8_slabs(const void* a, const void* b, float* partials,
                              size_t partial_bytes, int64_t M, int64_t N, int64_t K,
                              void* stream);

/* [slab reduce + dequant] + fused_add_rms_norm (layernorm_kernels.cu:200-240) +
 * dynamic_per_token_scaled_fp8_quant (fp8/common.cu:201-256).  input (T) XOR slabs
 * (+ slab scales: a per token or [1], b per channel or [1]); residual updated in place
 * (has_residual == 0: residual = x); q_out fp8 [tokens, hidden], scale_out fp32 [tokens];
 * out (optional) = the normalised activations in T. */
int aphro_fused_add_rms_norm_quant_fp8(const void* input, const float* slabs, int nslab,
                                       const float* slab_a_scales, const float* slab_b_scales,
                                       int a_scale_per_token, int b_scale_per_channel,
                                       void* residual, int has_residual, const void* weight,
                                       float eps, void* q_out, float* scale_out, void* out,
                                       int64_t tokens, int hidden, int dtype, void* stream);
/* The same with a STATIC per-tensor activation scale (static_scale: fp32 [1] on the device, the layer's input_scale;
 * NULL = the dynamic form): q = fp8(y * (1 / scale)) as static_scaled_fp8_quant (fp8/common.cu:187-199),
 * scale_out[token] = scale. */
int aphro_fused_add_rms_norm_quant_fp8_static(const void* input, const float* slabs, int nslab,
                                              const float* slab_a_scales, const float* slab_b_scales,
                                              int a_scale_per_token, int b_scale_per_channel,
                                              void* residual, int has_residual, const void* weight,
                                              float eps, void* q_out, float* scale_out, void* out,
                                              int64_t tokens, int hidden, int dtype,
                                              const float* static_scale, void* stream);

/* silu_and_mul (activation_kernels.cu:12-75) + dynamic_per_token_scaled_fp8_quant over
 * input T [tokens, 2d]; out (optional) = the activations in T. */
int aphro_silu_and_mul_quant_fp8(const void* input, void* q_out, float* scale_out, void* out,
                                 int64_t tokens, int d, int dtype, void* stream);
/* ... + static_scaled_fp8_quant (static_scale fp32 [1]; NULL = the dynamic form). */
int aphro_silu_and_mul_quant_fp8_static(const void* input, void* q_out, float* scale_out, void* out,
                                        int64_t tokens, int d, int dtype, const float* static_scale,
                                        void* stream);

/* aphro_paged_attention_rope_packed over the slabs of a QUANTISED qkv projection:
 * value = slab_row_scale[seq] * (slab_col_scale[c] * sum of slabs) before the rounding. */
int aphro_paged_attention_rope_packed_scaled(void* out, void* out_packed, const float* qkv_slabs,
                                             int nslab, const float* slab_row_scale,
                                             const float* slab_col_scale, const int64_t* positions,
                                             const void* cos_sin_cache, const int64_t* slot_mapping,
                                             void* key_cache, void* value_cache, int num_seqs,
                                             int num_heads, int num_kv_heads, int head_size,
                                             float scale, const int32_t* block_tables,
                                             const int32_t* seq_lens, int max_num_blocks_per_seq,
                                             int block_size, int max_seq_len,
                                             const float* alibi_slopes, int64_t kv_block_stride,
                                             int64_t kv_head_stride, int dtype, int kv_dtype,
                                             float k_scale, float v_scale, void* stream);
/* ... with the output ALSO (or only: out may be NULL) as e4m3 under a static per-tensor scale, for an FP8 o_proj whose
 * checkpoint carries input_scale: out_q8[seq, head, d] = fp8(T(out) * (1 / *out_q8_scale)) -- the bits of
 * static_scaled_fp8_quant (fp8/common.cu:187-199) over `out`, without its launch. */
int aphro_paged_attention_rope_scaled_q8(void* out, void* out_q8, const float* out_q8_scale,
                                         const float* qkv_slabs, int nslab, const float* slab_row_scale,
                                         const float* slab_col_scale, const int64_t* positions,
                                         const void* cos_sin_cache, const int64_t* slot_mapping,
                                         void* key_cache, void* value_cache, int num_seqs, int num_heads,
                                         int num_kv_heads, int head_size, float scale,
                                         const int32_t* block_tables, const int32_t* seq_lens,
                                         int max_num_blocks_per_seq, int block_size, int max_seq_len,
                                         const float* alibi_slopes, int64_t kv_block_stride,
                                         int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale,
                                         float v_scale, void* stream);

/* ... with, for an FP8 o_proj under DYNAMIC per-token activation scales, the absmax of every (sequence, kv-head) slice of
 * `out` in out_absmax [num_seqs][num_kv_heads] (`out` row-major [S, Hq, hd] and / or out_pairs: the same values in the
 * pair-major layout of aphro_fp8_gemm_resident_aq over [S, Hq * hd]; either may be NULL): aphro_fp8_gemm_resident_aq reduces the partials to the row scale and
 * quantises on load -- dynamic_per_token_scaled_fp8_quant (fp8/common.cu:201-256) without its launch. */
int aphro_paged_attention_rope_scaled_absmax(void* out, void* out_pairs, float* out_absmax, const float* qkv_slabs, int nslab,
                                             const float* slab_row_scale, const float* slab_col_scale,
                                             const int64_t* positions, const void* cos_sin_cache,
                                             const int64_t* slot_mapping, void* key_cache, void* value_cache,
                                             int num_seqs, int num_heads, int num_kv_heads, int head_size, float scale,
                                             const int32_t* block_tables, const int32_t* seq_lens,
                                             int max_num_blocks_per_seq, int block_size, int max_seq_len,
                                             const float* alibi_slopes, int64_t kv_block_stride,
                                             int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale,
                                             float v_scale, void* stream);

/* ------------------------------------------------------------------------
 * Random sampling inside the decode graph: temperature -> top-k -> top-p -> softmax
 * -> multinomial in one launch.  Semantics of modeling/layers/sampler.py: logits / t (:256-262, t <
 * 1e-5 -> 1), _apply_top_k_top_p (:865-891; values tied with a threshold are kept as a group),
 * _apply_min_p (:894-908: prob < min_p * max prob dropped), _multinomial = argmax(probs / q), q ~ Exp(1)
 * (:1273-1292).  logits [rows, vocab] f16 / bf16 / f32
 * with row_stride elements between rows; temperature / top_k / top_p / min_p per row or NULL (disabled;
 * top_k <= 0 or >= vocab disables); q: the caller's Exp(1) draws [rows, vocab] (q_stride), or NULL
 * to draw them in the kernel from seeds[rows].  out: int64 [rows]; logprobs_out (optional, fp32 [rows]):
 * log_softmax of the temperature-scaled row AFTER the masks, at the sampled token (sampler.py:545).
 * ---------------------------------------------------------------------- */
int aphro_sample_top_k_top_p(int64_t* out, const void* logits, int64_t row_stride,
                             const float* temperature, const int32_t* top_k, const float* top_p,
                             const float* min_p, const float* q, int64_t q_stride,
                             const int64_t* seeds, float* logprobs_out, int64_t rows, int64_t vocab,
                             int dtype, void* stream);

/* ------------------------------------------------------------------------
 * Tensor-parallel sum all-reduce through xGMI peer access -- the `_C_custom_ar::*` ops
 * (kernels/torch_bindings.cpp:506-536, kernels/custom_all_reduce.cu; compiled out of the reference's
 * ROCm build).  One process per GPU; buffers are exchanged as HIP IPC handles (64 opaque bytes each,
 * aphro_ipc_handle_bytes()) gathered over any host-side channel.  Calls on one communicator must be
 * issued in the same order with the same sizes on every rank.
 * ---------------------------------------------------------------------- */
int64_t aphro_custom_ar_meta_size(void);                       /* meta_size(): bytes of one signal area */
int aphro_ipc_handle_bytes(void);
/* zero-filled, peer-visible UNCACHED device memory for the signal area and the two-shot scratch */
int aphro_custom_ar_alloc_shared(void** ptr, size_t bytes);
int aphro_custom_ar_free_shared(void* ptr);
/* IPC handle of the allocation containing ptr + ptr's offset in it (storage._share_cuda_() role) */
int aphro_ipc_get_mem_handle(const void* ptr, char* handle_out, int64_t* offset_out);
/* init_custom_ar(meta, rank_data, handles, offsets, rank, full_nvlink) -> fa.  *_handles: world
 * handles in rank order (the entry of `rank` itself is ignored); rank_data: device memory holding
 * one 64-byte peer table per registered buffer. */
int aphro_custom_ar_init(void** fa_out, void* signal, const char* signal_handles,
                         const int64_t* signal_offsets, void* scratch, size_t scratch_bytes,
                         const char* scratch_handles, const int64_t* scratch_offsets, void* rank_data,
                         size_t rank_data_bytes, int rank, int world);
int aphro_custom_ar_dispose(void* fa);
/* register_buffer(fa, t, handles, offsets) */
int aphro_custom_ar_register_buffer(void* fa, const void* local_ptr, const char* handles,
                                    const int64_t* offsets);
/* 1: one-shot (every rank sums all inputs), 0: two-shot (reduce-scatter + all-gather) */
int aphro_custom_ar_should_one_shot(int world, size_t bytes);
/* all_reduce_reg(fa, inp, out) when reg_buffer == NULL (inp registered, or the stream is capturing
 * and inp gets registered by register_graph_buffers afterwards); all_reduce_unreg(fa, inp,
 * reg_buffer, out) otherwise.  dtype APHRO_F16 / BF16 / F32; bytes % 16 == 0.  Sum in rank order,
 * fp32 accumulate: bit-identical on every rank. */
int aphro_custom_ar_all_reduce(void* fa, const void* inp, void* out, int64_t numel, int dtype,
                               void* reg_buffer, size_t reg_buffer_bytes, void* stream);
/* get_graph_buffer_ipc_meta(fa) -> (handles, offsets); handles_out == NULL: only *count */
int aphro_custom_ar_get_graph_buffer_ipc_meta(void* fa, char* handles_out, int64_t* offsets_out,
                                              int cap, int* count);
/* register_graph_buffers(fa, handles, offsets): rank-major [world][count] */
int aphro_custom_ar_register_graph_buffers(void* fa, const char* handles, const int64_t* offsets,
                                           int count);
/* 1 if one of this rank's barriers timed out since the last query (bounded spin: a lost peer raises
 * this instead of hanging the GPU) */
int aphro_custom_ar_error(void* fa);
/* The row-parallel linear's all-reduce (modeling/layers/linear.py:1142-1143) fused with the residual add + RMSNorm that
 * follows it in every decoder layer (models/llama.py: post_attention_layernorm / the next layer's input_layernorm;
 * kernels/layernorm_kernels.cu:200-240), ONE launch, the bits of aphro_custom_ar_all_reduce -> aphro_fused_add_rms_norm_pack:
 * x = sum over the ranks of inp [tokens, hidden] (tokens <= 64), residual' = x + residual (in place), y = rms_norm(residual')
 * * weight -> `packed` (aphro_wna16_packed_a_bytes; f16) and / or row-major `out`; every rank ends with every row.  One-shot
 * sizes (aphro_custom_ar_fused_norm_one_shot == 1): one workgroup per token row reads the row from every rank.  Larger:
 * reduce-scatter by column slice of every row, the row's workgroup gathers the slices and normalises.  prefetch /
 * prefetch_bytes (optional): the packed weights of the GEMM that consumes the norm's output -- extra workgroups of the same
 * launch pull them through the Infinity Cache while the reducing workgroups wait on flags and links (the all-reduce
 * overlapped with the GEMM's weight stream, BASELINE north_star).  reg_buffer as in aphro_custom_ar_all_reduce.
 * csrc/custom_all_reduce.hip. */
int aphro_custom_ar_fused_norm_one_shot(int world, int64_t tokens, int hidden, int esz);
int aphro_custom_ar_fused_add_rms_norm(void* fa, const void* inp, void* residual, int has_residual,
                                       const void* weight, float eps, void* packed, void* out,
                                       int64_t tokens, int hidden, int dtype,
                                       const void* prefetch, size_t prefetch_bytes,
                                       void* reg_buffer, size_t reg_buffer_bytes, void* stream);
/* The same launch for an FP8 W8A8 layer under tensor parallelism: the all-reduce of the row-parallel linear
 * (modeling/layers/linear.py:1142-1143), the residual add + RMSNorm (kernels/layernorm_kernels.cu:200-240) and the
 * activation quantisation at the head of the next FP8 linear (quantization/fp8.py Fp8LinearMethod.apply ->
 * ops.scaled_fp8_quant, kernels/quantization/fp8/common.cu:187-256) -- the bits of aphro_custom_ar_all_reduce ->
 * aphro_fused_add_rms_norm_quant_fp8(input = the sum).  q_out [tokens, hidden] e4m3; scale_out [tokens] f32 (dynamic
 * per-token: max(absmax / 448, 1 / (448 * 512)); static_scale [1] given: q = fp8(y * (1 / s)), every entry = s);
 * `out` (optional): the normalised rows in the activation dtype.  tokens <= 64. */
int aphro_custom_ar_fused_add_rms_norm_quant_fp8(void* fa, const void* inp, void* residual, int has_residual,
                                                 const void* weight, float eps, void* q_out, float* scale_out,
                                                 const float* static_scale, void* out, int64_t tokens, int hidden,
                                                 int dtype, void* reg_buffer, size_t reg_buffer_bytes, void* stream);
/* ... and for a sparse-MLP (Mixtral) layer under tensor parallelism: the attention block's all-reduce (linear.py:1142-1143),
 * post_attention_layernorm and the replicated router linear of MixtralMoE (models/mixtral.py:60-110) in one launch -- the
 * bits of aphro_custom_ar_all_reduce -> aphro_fused_add_rms_norm_router(input = the sum).  out [tokens, hidden] (the
 * experts' input), router_out [tokens, num_experts] in the activation dtype, num_experts <= 16, tokens <= 64. */
int aphro_custom_ar_fused_add_rms_norm_router(void* fa, const void* inp, void* residual, int has_residual,
                                              const void* weight, float eps, void* out, const void* router_w,
                                              void* router_out, int num_experts, int64_t tokens, int hidden, int dtype,
                                              void* reg_buffer, size_t reg_buffer_bytes, void* stream);
/* Loopback communicator (timing rig for ONE rank of a TP group on a one-GPU box, bench.py --sim-tp): `world` ranks that
 * all resolve to this process's buffers; the kernels above run unchanged (flags and scratch through uncached memory,
 * `world` reads per element) with local memory in place of the xGMI links.  Results are not a sum over real ranks. */
int aphro_custom_ar_init_loopback(void** fa_out, void* signal, void* scratch, size_t scratch_bytes,
                                  void* rank_data, size_t rank_data_bytes, int world);

/* W4A16 GEMM for prefill-sized M (M > 64) -- the role of `_C::gptq_marlin_gemm` at large M
 * (kernels/torch_bindings.cpp:195-201, kernels/quantization/gptq_marlin/gptq_marlin.cu:544,2247) and of the reference's
 * `reconstruct + hipBLAS` fallback above 50 rows (kernels/quantization/gptq/q_gemm.cu:1529-1544), which this replaces
 * with ONE MFMA kernel that dequantises in registers (csrc/wna16_gemm_large.hip).  Same tensors as aphro_gptq_gemm:
 * q_weight [K/8, N] in the exllama-shuffled order, qzeros [G, N/8], scales [G, N] in `dtype`; c [M, N] in `dtype`.
 * N % 128 == 0, K % 64 == 0, group size % 64 == 0.  Act-order: pass activations already gathered (a[:, perm]).
 * workspace: aphro_wna16_gemm_large_workspace_bytes (f16 copy of bf16 activations, saturating; fp32 split-K slabs when
 * the tile grid alone would not cover the chip). */
size_t aphro_wna16_gemm_large_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t groups, int dtype);
int aphro_wna16_gemm_large(const void* a, const uint32_t* q_weight, const uint32_t* qzeros, const void* scales,
                           void* c, void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K,
                           int64_t groups, int64_t lda, int zero_offset, int dtype, void* stream);
/* The same GEMM on a gate_up matrix whose columns are interleaved (gate_j, up_j) pairs, SiluAndMul in the epilogue:
 * act [M, N / 2] = silu_and_mul(a . dequant(W)), the GEMM result rounded to the dtype first -- the bits of
 * aphro_wna16_gemm_large followed by aphro_silu_and_mul_interleaved (gptq_gemm / awq_gemm + kernels/activation_kernels.cu:12-75
 * in the reference's LlamaMLP, modeling/models/llama.py:88-93) without the [M, N] round trip.  Shapes the plan cuts into
 * K slices are not served (aphro_wna16_gemm_large_silu_supported == 0).  csrc/wna16_gemm_large.hip. */
int aphro_wna16_gemm_large_silu_supported(int64_t M, int64_t N, int64_t K, int64_t groups);
int aphro_wna16_gemm_large_silu(const void* a, const uint32_t* q_weight, const uint32_t* qzeros, const void* scales,
                                void* act, void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K,
                                int64_t groups, int64_t lda, int zero_offset, int dtype, void* stream);

/* W4A16 GEMM for decode batches of 33..64 rows (any M in 1..64 is accepted) -- the reference runs its exllama kernel
 * up to 50 rows and reconstruct + hipBLAS above (kernels/quantization/gptq/q_gemm.cu:1529-1544); Marlin covers the range
 * with one kernel (gptq_marlin.cu:2247).  csrc/wna16_gemm_mid.hip: weights global -> VGPR -> 32x32x16 MFMA (the exllama
 * dword is the fragment), activations fragment-major from L2, 4 waves split K and meet in an LDS butterfly, fp32 slabs
 * for more K slices.  Same tensors as aphro_wna16_gemm_large.  N % 128 == 0, group size % 128 == 0, K % (4 groups) == 0.
 * workspace: aphro_wna16_gemm_mid_workspace_bytes (packed activations + slabs). */
int aphro_wna16_gemm_mid_supported(int64_t M, int64_t N, int64_t K, int64_t groups);
size_t aphro_wna16_gemm_mid_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t groups);
int aphro_wna16_gemm_mid(const void* a, const uint32_t* q_weight, const uint32_t* qzeros, const void* scales,
                         void* c, void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K,
                         int64_t groups, int64_t lda, int zero_offset, int dtype, void* stream);

/* The same kernel on the decode fast path (33..64 rows): activations already in the decode kernel's fragment-major
 * format (aphro_wna16_pack_a / the fused producers), results handed to the next fused kernel -- act_packed != NULL:
 * gate_up form (interleaved gate / up columns, SiluAndMul + pack epilogue = the role of aphro_wna16_gemm_silu_pack,
 * `_C::silu_and_mul` fused, activation_kernels.cu:14-28); slabs != NULL: fp32 slabs [aphro_wna16_gemm_mid_ksplit][M][N]
 * summed by the consumer (aphro_fused_add_rms_norm_pack); else c [M, N].  aphro_wna16_gemm_mid_ksplit: K slices of the
 * plan, 0 if the shape is not served. */
int aphro_wna16_gemm_mid_ksplit(int64_t M, int64_t N, int64_t K, int64_t groups);
int aphro_wna16_gemm_mid_packed(const void* a_packed, const uint32_t* q_weight, const uint32_t* qzeros,
                                const void* scales, void* c, void* slabs, size_t slabs_bytes, void* act_packed,
                                int64_t M, int64_t N, int64_t K, int64_t groups, int zero_offset, int dtype,
                                void* stream);

/* Decode GEMM for M <= 32 with the activations resident in registers (round 3, csrc/wna16_gemm_resident.hip): same role
 * and arithmetic as aphro_wna16_gemm_packed / aphro_wna16_gemm_silu_pack (the reference's exllama small-M kernel,
 * kernels/quantization/gptq/q_gemm.cu:190-326; gptq_marlin_gemm at small M, gptq_marlin.cu:2247), one workgroup per CU:
 * column strips x K slices.  a_packed: aphro_wna16_pack_a's fragment-major f16.  Exactly one output: act_packed
 * (interleaved gate / up columns, SiluAndMul + pack epilogue, `_C::silu_and_mul` fused, activation_kernels.cu:14-28; one
 * K slice), slabs (fp32 [aphro_wna16_resident_ksplit][M][N], summed by the consumer) or c ([M, N] in `dtype`, one K
 * slice).  strip_layout != 0: q_weight was re-laid by aphro_wna16_strip_relayout for this (N, K, groups).
 * aphro_wna16_resident_ksplit: K slices of the plan, 0 = shape not served (the caller keeps aphro_wna16_gemm_packed). */
int aphro_wna16_resident_ksplit(int64_t M, int64_t N, int64_t K, int64_t groups);
int aphro_wna16_gemm_resident(const void* a_packed, const uint32_t* q_weight, const uint32_t* qzeros,
                              const void* scales, void* c, float* slabs, size_t slabs_bytes, void* act_packed,
                              int64_t M, int64_t N, int64_t K, int64_t groups, int zero_offset, int dtype,
                              int strip_layout, void* stream);
/* Load-time: [K/8, N] exllama-ordered words (aphro_gptq_shuffle's output) -> strip-major order of the resident kernel's
 * plan for this shape (every wave's 16-byte pieces in the order it reads them).  A permutation of the words; out != in. */
int aphro_wna16_strip_relayout(const uint32_t* q_weight, uint32_t* out, int64_t M, int64_t N, int64_t K,
                               int64_t groups, void* stream);
/* ONE resident copy of a decode matrix (round 6; the reference keeps one too: its exllama / Marlin kernels serve every M from
 * the same repacked tensor, quantization/gptq.py:214-228, gptq_marlin.py:296-330 -- spare HBM is KV blocks,
 * worker/cache_engine.py:66-86).  The strip-major copy stays, the [K/8, N] original goes:
 *   aphro_wna16_strip_unrelayout   the permutation backwards (strip-major -> [K/8, N]; out != strip), for kernels that do not
 *                                  read the strip-major order;
 *   aphro_wna16_strip_geometry     geom[5] = {waves, 128-k segments per wave, 64-column passes, 16-column remainder units,
 *                                  K slices} of the strip-major order for (M class, N, K, groups); 1 = served, 0 = none;
 *   aphro_wna16_gemm_large_strip   aphro_wna16_gemm_large (silu == 0) / aphro_wna16_gemm_large_silu (silu != 0) with
 *                                  q_weight_strip = aphro_wna16_strip_relayout's output for the M class strip_m (32): every
 *                                  plan addresses the 16-byte pieces in place (same loads, same bits; workspace:
 *                                  aphro_wna16_gemm_large_strip_workspace_bytes = the [K/8, N] entry's).
 * csrc/wna16_gemm_resident.hip, csrc/wna16_gemm_large.hip. */
int aphro_wna16_strip_unrelayout(const uint32_t* strip, uint32_t* out, int64_t M, int64_t N, int64_t K,
                                 int64_t groups, void* stream);
/*   aphro_wna16_gemm_mid_packed_strip   aphro_wna16_gemm_mid_packed (33..64 rows, one pass over the weights) with q_weight_strip
 *                                  = the strip-major copy for the M class strip_m: same loads at strip-major addresses
 *                                  (csrc/wna16_strip.h), same bits.  csrc/wna16_gemm_mid.hip. */
int aphro_wna16_gemm_mid_packed_strip(const void* a_packed, const uint32_t* q_weight_strip, const uint32_t* qzeros,
                                      const void* scales, void* c, void* slabs, size_t slabs_bytes, void* act_packed,
                                      int64_t M, int64_t N, int64_t K, int64_t groups, int zero_offset, int dtype,
                                      int64_t strip_m, void* stream);
int aphro_wna16_strip_geometry(int64_t M, int64_t N, int64_t K, int64_t groups, int* geom);
size_t aphro_wna16_gemm_large_strip_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t groups, int dtype, int64_t strip_m);
int aphro_wna16_gemm_large_strip(const void* a, const uint32_t* q_weight_strip, const uint32_t* qzeros, const void* scales,
                                 void* c, void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K,
                                 int64_t groups, int64_t lda, int zero_offset, int dtype, int silu, int64_t strip_m,
                                 void* stream);

/* The op-level decode GEMM (`_C::gptq_gemm` / `_C::awq_gemm` at M <= 32, torch_bindings.cpp:229-243; the role of
 * gemm_half_q_half_gptq_4bit_kernel, q_gemm.cu:190-326, which also reads row-major `a` and writes [M, N] in one launch) in
 * ONE launch of the resident kernel: the A fragments are gathered from the row-major f16 rows in place (no
 * aphro_wna16_pack_a launch) and a K-sliced shape is reduced by the last-arriving slice of each strip inside the kernel
 * (write-through partials, one ticket per strip, slices added in slice order: bits independent of arrival order) instead
 * of a reduce launch.  workspace: K-slices x M x N floats (aphro_wna16_workspace_bytes covers it).  aphro_gptq_gemm takes
 * this path by itself when `aphro_wna16_gemm_rowmajor_supported` (f16, no act-order permutation, a shape the resident
 * kernel tiles; APHRO_WNA16_OP_NO_RESIDENT=1 turns it off).  APHRO_ERR_WORKSPACE without any launch when the ticket array
 * cannot be allocated (first call under a stream capture). */
int aphro_wna16_gemm_rowmajor_supported(int64_t M, int64_t N, int64_t K, int64_t groups, int dtype);
int aphro_wna16_gemm_rowmajor(const void* a, int64_t lda, const uint32_t* q_weight, const uint32_t* qzeros,
                              const void* scales, void* c, void* workspace, size_t workspace_bytes, int64_t M,
                              int64_t N, int64_t K, int64_t groups, int zero_offset, int dtype, int strip_layout,
                              void* stream);

/* Test-rig helper: a launch that holds `stream` for `us` microseconds without touching memory -- the stand-in for an
 * all-reduce when one rank of a tensor-parallel group is timed alone (bench.py --sim-tp).  No reference counterpart. */
int aphro_spin_us(double us, void* stream);

/* Overlap helper for tensor parallelism (north_star: "all-reduce overlapped with the quantized GEMMs on HIP
 * streams"): streams `bytes` at `ptr` through the memory-side Infinity Cache on `stream` while the all-reduce of the
 * previous row-parallel projection runs on a side stream -- see aphrodite_engine_amd/distributed/overlap.py.  Reads
 * only; ptr 16-byte aligned.  No reference counterpart (the reference serialises its all-reduce with the GEMMs,
 * parallel_state.py:321-379). */
int aphro_prefetch(const void* ptr, size_t bytes, void* stream);

/* ------------------------------------------------------------------------
 * Mixture of experts: routing, dispatch and the grouped W4A16 GEMM
 * ---------------------------------------------------------------------- */

/* _moe_C::topk_softmax(Tensor! topk_weights, Tensor! topk_indices,
 *     Tensor! token_expert_indices, Tensor gating_output)
 *   kernels/moe/torch_bindings.cpp:11-14, kernels/moe/softmax.cu:17-520.
 * gating fp32 [T, E] (E <= 256); weights = softmax probabilities of the k winners
 * (not renormalised), ids int32, token_expert_indices[t][k] = k * T + t (may be NULL). */
int aphro_topk_softmax(float* topk_weights, int32_t* topk_ids, int32_t* token_expert_indices,
                       const float* gating_output, int64_t num_tokens, int num_experts,
                       int topk, void* stream);

/* _C::moe_align_block_size(Tensor topk_ids, int num_experts, int block_size,
 *     Tensor! sorted_token_ids, Tensor! experts_ids, Tensor! num_tokens_post_pad)
 *   kernels/torch_bindings.cpp:394-399, kernels/moe/align_block_size_kernel.cu:17-126.
 * sorted_token_ids int32 [numel + E*(block-1)] (padding entries = numel; filled here),
 * expert_ids int32 [ceil(that / block)] (-1 beyond the used blocks), inv_pos (optional,
 * ours) int32 [numel]: position of slot i in sorted_token_ids. */
int aphro_moe_align_block_size(const int32_t* topk_ids, int num_experts, int block_size,
                               int32_t* sorted_token_ids, int32_t* expert_ids,
                               int32_t* num_tokens_post_pad, int32_t* inv_pos,
                               int64_t numel, void* stream);

/* aphro_fused_add_rms_norm_pack for a sparse-MLP layer: [slab reduce] + fused_add_rms_norm + the router's logits
 * router_out[tokens, num_experts] = round_T(y . router_w[e]) -- the replicated `gate` linear of MixtralMoE
 * (modeling/models/mixtral.py:60-110), a [M, E] library GEMM launch of its own otherwise.  num_experts <= 16; `out` =
 * the row-major normalised activations.  csrc/fused_decode.hip. */
int aphro_fused_add_rms_norm_router(const void* input, const float* slabs, int nslab, void* residual, int has_residual,
                                    const void* weight, float eps, void* out, const void* router_w, void* router_out,
                                    int num_experts, int64_t tokens, int hidden, int dtype, void* stream);

/* The norm that follows a sparse MLP with moe_combine folded into its input stage: x[t] = sum_k round_T(w[t, k] * sum_s
 * slab[s][inv_pos[t k + kk]]) (fused_moe.py:520-542; = aphro_moe_combine), then fused_add_rms_norm (+ pack) as
 * aphro_fused_add_rms_norm_pack.  slabs fp32 [nslab][m_pad][hidden].  csrc/fused_decode.hip. */
int aphro_fused_add_rms_norm_pack_combine(const float* slabs, int nslab, int64_t m_pad, const int32_t* inv_pos,
                                          const float* topk_weights, int topk, void* residual, int has_residual,
                                          const void* weight, float eps, void* packed, void* out, int64_t tokens, int hidden,
                                          int dtype, void* stream);

/* fused_topk (fused_moe.py:369-402: gating.float() -> topk_softmax -> optional renormalise, w / sum_k w in fp32) +
 * moe_align_block_size (:174-228) in ONE launch for decode-sized batches (num_tokens * topk <= 8192, topk <= 8): same
 * outputs as the separate ops (the same routing arithmetic and counting sort: ids and lists identical, weights identical up
 * to the order of the renormalising sum for topk > 2).  gating:
 * [num_tokens, gating_stride] f16 / bf16 / f32 router logits.  csrc/moe.hip. */
int aphro_moe_route_align(float* topk_weights, int32_t* topk_ids, const void* gating, int64_t gating_stride,
                          int32_t* sorted_token_ids, int32_t* expert_ids, int32_t* num_tokens_post_pad, int32_t* inv_pos,
                          int64_t num_tokens, int num_experts, int topk, int renormalize, int block_size, int dtype,
                          void* stream);

/* aphro_moe_route_align + aphro_moe_gather_pack in ONE launch (decode-sized batches, <= 16 experts, <= 256 tokens, block 16,
 * f16 / bf16 logits and activations: aphro_moe_route_gather_supported): every workgroup of the gather redoes the routing of
 * the few dozen tokens in its own LDS, workgroup 0 publishes it.  Same outputs, bit for bit, as the two calls
 * (fused_topk + moe_align_block_size, fused_moe.py:223-268, 405-436; the sorted_ids addressing of marlin_gemm_moe).
 * m_pad: a multiple of 16 >= num_tokens * topk + num_experts * (block_size - 1). */
int aphro_moe_route_gather_supported(int64_t num_tokens, int num_experts, int topk, int block_size, int64_t K);
int aphro_moe_route_gather(float* topk_weights, int32_t* topk_ids, const void* gating, int64_t gating_stride,
                           int32_t* sorted_token_ids, int32_t* expert_ids, int32_t* num_tokens_post_pad, int32_t* inv_pos,
                           int64_t num_tokens, int num_experts, int topk, int renormalize, int block_size, const void* a,
                           int64_t lda, void* packed, int64_t m_pad, int64_t K, int dtype, void* stream);

/* Fragment-major activation pack of the rows a[sorted_token_ids[r] / topk] (zero rows for
 * padding) -- the sorted_ids / replicate_input addressing of marlin_gemm_moe
 * (kernels/moe/marlin_moe_ops.cu) done once, ahead of the GEMM. */
int aphro_moe_gather_pack(const void* a, const int32_t* sorted_token_ids,
                          const int32_t* num_tokens_post_pad, void* packed, int64_t m_pad,
                          int64_t K, int64_t lda, int64_t numel, int topk, int dtype,
                          void* stream);

/* Grouped expert GEMM, the marlin_gemm_moe role (kernels/moe/torch_bindings.cpp:17-24):
 * one launch, m-tile z uses expert expert_ids[z].  Weights [E][K/8][N] K-packed,
 * qzeros [E][G][N/8], scales [E][G][N].  Exactly one of act_packed (SiluAndMul + pack
 * epilogue over interleaved gate/up columns; needs aphro_wna16_grouped_ksplit == 1),
 * c (T [m_pad, N]) or partials (fp32 [ksplit][m_pad][N], c == act_packed == NULL). */
int aphro_wna16_grouped_ksplit(int64_t m_pad, int64_t N, int64_t K, int64_t groups);
int aphro_wna16_gemm_grouped(const void* a_packed, const uint32_t* q_weight,
                             const uint32_t* qzeros, const void* scales,
                             const int32_t* expert_ids, const int32_t* num_tokens_post_pad,
                             void* c, float* partials, size_t partial_bytes,
                             void* act_packed, int64_t m_pad, int64_t N, int64_t K,
                             int64_t groups, int zero_offset, int dtype, void* stream);

/* out[t] = sum_k round(w[t][k] * y[inv_pos[t*topk+k]]) with y = sum of the nslab fp32 slabs
 * [nslab][m_pad][N] of the second expert GEMM (fused_moe.py:520-542). */
int aphro_moe_combine(void* out, const float* slabs, int nslab, int64_t m_pad,
                      const int32_t* inv_pos, const float* topk_weights, int64_t num_tokens,
                      int topk, int64_t N, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * Per-step bookkeeping inside the HIP graph
 * ---------------------------------------------------------------------- */

/* _C::advance_step_flashattn(int num_seqs, int num_queries, int block_size,
 *     Tensor! input_tokens, Tensor sampled_token_ids, Tensor! input_positions,
 *     Tensor! seq_lens, Tensor! slot_mapping, Tensor block_tables)
 *   kernels/torch_bindings.cpp:77-82, kernels/prepare_inputs/advance_step.cu:13-51.
 * int64 tokens / positions / slots, int32 seq_lens and block_tables (row stride in elements). */
int aphro_advance_step_flashattn(int num_seqs, int num_queries, int block_size,
                                 int64_t* input_tokens, const int64_t* sampled_token_ids,
                                 int64_t* input_positions, int32_t* seq_lens,
                                 int64_t* slot_mapping, const int32_t* block_tables,
                                 int64_t block_tables_stride, void* stream);

/* GPTQ 2 / 3 / 8-bit weights (`_C::gptq_gemm` / `_C::gptq_shuffle` with bit != 4, torch_bindings.cpp:229-243;
 * gemm_half_q_half_gptq_{2,3,8}bit_kernel q_gemm.cu:329-700, reconstruct_gptq q_gemm.cu:1394-1505, make_sequential /
 * shuffle q_gemm.cu:1659-1872; csrc/wnx_gemm.hip).  Layout: the checkpoint's -- values (zero points: along N) laid end to
 * end in uint32 words, 32 values per `bits` words; after gptq_shuffle the SAME words with act-order rows made sequential
 * (the post-shuffle layout is private to the library that reads it).
 *   aphro_gptq_dequant_bits   [K, N] 16-bit = (q - (z + 1)) * s, one rounding: the reference's reconstruct kernels bit for
 *                             bit (g_idx: row -> group, or NULL = k / group_size).
 *   aphro_gptq_gemm_bits      M <= 32 rows on the sequential layout, MFMA on the integers, fp32 group sums; act-order:
 *                             pass a[:, perm].  aphro_gptq_gemm_bits_supported: 1 if the shape is served.
 *   aphro_gptq_make_sequential_bits   new row k = source row perm[k]; out != q_weight. */
int aphro_gptq_dequant_bits(const uint32_t* q_weight, const uint32_t* qzeros, const void* scales, const int32_t* g_idx,
                            void* out, int64_t K, int64_t N, int64_t groups, int bits, int dtype, void* stream);
int aphro_gptq_gemm_bits_supported(int64_t M, int64_t N, int64_t K, int64_t groups, int bits);
int aphro_gptq_gemm_bits(const void* a, int64_t lda, const uint32_t* q_weight, const uint32_t* qzeros, const void* scales,
                         void* c, int64_t M, int64_t N, int64_t K, int64_t groups, int bits, int dtype, void* stream);
int aphro_gptq_make_sequential_bits(const uint32_t* q_weight, uint32_t* out, const int32_t* perm, int64_t K, int64_t N,
                                    int bits, void* stream);

/* FP8 W8A8 decode GEMM for M <= 32, one workgroup per CU on a strip-major copy of the weight (round 4,
 * csrc/fp8_gemm_resident.hip): same role and arithmetic as aphro_scaled_mm_fp8 / aphro_scaled_mm_fp8_slabs
 * (cutlass_scaled_mm, kernels/quantization/cutlass_w8a8/scaled_mm_entry.cu:92-137; torch._scaled_mm on ROCm,
 * quantization/utils/w8a8_utils.py:83-183).  aphro_fp8_gemm_resident_ksplit: K slices of the plan for (M, N, K), 0 =
 * shape not served.  aphro_fp8_strip_relayout: load time, [N, K] row-major e4m3 -> the strip-major order of that plan
 * (a permutation of 16-byte pieces; out != w).  aphro_fp8_gemm_resident: a e4m3 [M, lda] (lda % 16 == 0), exactly one
 * of out ([M, N] in `dtype` = a_scales * (b_scales * acc) + bias, plans with one K slice) / slabs (fp32
 * [ksplit][M][N] raw accumulators for a fused consumer). */
int aphro_fp8_gemm_resident_ksplit(int64_t M, int64_t N, int64_t K);
int aphro_fp8_strip_relayout(const void* w, void* out, int64_t M, int64_t N, int64_t K, void* stream);
int aphro_fp8_gemm_resident(const void* a, int64_t lda, const void* w_strip, const float* a_scales,
                            const float* b_scales, const void* bias, void* out, float* slabs, size_t slabs_bytes,
                            int64_t M, int64_t N, int64_t K, int a_scale_per_token, int b_scale_per_channel,
                            int dtype, void* stream);

/* Round 6 -- the dynamic per-token scheme without its quantising launches (what llm-compressor checkpoints select,
 * quantization/compressed_tensors/schemes/compressed_tensors_w8a8_fp8.py:133-141; quantiser
 * kernels/quantization/fp8/common.cu:187-256, called from quantization/utils/w8a8_utils.py:83-183 through
 * _custom_ops.scaled_fp8_quant :632-685).
 * aphro_fp8_gemm_resident_aq: the resident GEMM with `a16` = the PRODUCER's 16-bit activations [M, lda] (lda in elements,
 * lda % 8 == 0; or, a_pairs / act_pairs != 0, the PAIR-MAJOR layout the fused producers write for it: element (m, k) at
 * ((((k / 64) * ceil(M / 16) + m / 16) * 2 + (k % 16) / 8) * 64 + ((k % 64) / 16) * 16 + m % 16) * 8 + k % 8, every
 * 16-byte load of the GEMM lane-linear) and `absmax` = the producer's absmax partials [M][np] (np % 4 == 0, 4 <= np <= 256): the launch reduces the
 * partials to the row scale max(absmax / 448, 1 / (448 * 512)) -> scale_out[M] (may be NULL) and quantises every A fragment
 * on load as fp8(x / scale) with the IEEE quotient -- the bits of dynamic_per_token_scaled_fp8_quant followed by
 * aphro_fp8_gemm_resident.  aphro_fp8_quant_rows_aq: that quantiser alone (same device code) for parity tests.
 * aphro_fp8_strip_relayout_interleaved / aphro_fp8_gemm_resident_silu: gate_up with SiluAndMul in the epilogue (plans with
 * one K slice): the strip-major copy pairs (gate row j, up row N / 2 + j) as adjacent columns; dynamic scheme: act_out
 * `dtype` [M, N / 2] + absmax_out [M][strips] (strips = aphro_fp8_gemm_resident_strips(M, N, K)); static scheme: q8_out e4m3
 * [M, N / 2] = fp8(act * (1 / *static_out_scale)) (static_scaled_fp8_quant, common.cu:187-199) -- the bits of
 * aphro_fp8_gemm_resident -> silu_and_mul (kernels/activation_kernels.cu:12-75) [-> the quantiser]. */
int aphro_fp8_gemm_resident_strips(int64_t M, int64_t N, int64_t K);
int aphro_fp8_strip_relayout_interleaved(const void* w, void* out, int64_t M, int64_t N, int64_t K, void* stream);
int aphro_fp8_gemm_resident_aq(const void* a16, int64_t lda, int a_pairs, const float* absmax, int np, const void* w_strip,
                               float* scale_out, const float* b_scales, const void* bias, void* out, float* slabs,
                               size_t slabs_bytes, int64_t M, int64_t N, int64_t K, int b_scale_per_channel, int dtype,
                               void* stream);
int aphro_fp8_gemm_resident_silu(const void* a, int64_t lda, const void* w_strip_il, const float* a_scales,
                                 const float* b_scales, void* act_out, int act_pairs, float* absmax_out, void* q8_out,
                                 const float* static_out_scale, int64_t M, int64_t N, int64_t K, int a_scale_per_token,
                                 int b_scale_per_channel, int dtype, void* stream);
int aphro_fp8_quant_rows_aq(const void* x, const float* absmax, int np, void* q, float* scale_out, int64_t M, int64_t K,
                            int dtype, void* stream);

/* FP8 W8A8 decode GEMM for M <= 32 on the LDS-DMA streaming structure (round 3, csrc/fp8_gemm_stream.hip): same role and
 * arithmetic as aphro_scaled_mm_fp8 / aphro_scaled_mm_fp8_slabs (cutlass_scaled_mm, scaled_mm_entry.cu:92-137), which route
 * to it by themselves.  a e4m3 [M, lda], w e4m3 [N, K] row-major; exactly one of out ([M, N] in `dtype`, shapes whose K fits
 * one workgroup: aphro_fp8_gemm_stream_ksplit == 1) / slabs (fp32 [ksplit][M][N] raw accumulators).
 * aphro_fp8_gemm_stream_ksplit: K slices, 0 = shape not served (APHRO_FP8_NO_STREAM=1 turns the kernel off). */
int aphro_fp8_gemm_stream_ksplit(int64_t M, int64_t N, int64_t K);
int aphro_fp8_gemm_stream(const void* a, int64_t lda, const void* w, const float* a_scales, const float* b_scales,
                          const void* bias, void* out, float* slabs, size_t slabs_bytes, int64_t M, int64_t N, int64_t K,
                          int a_scale_per_token, int b_scale_per_channel, int dtype, void* stream);
/* gate_up of an FP8 MLP + SiluAndMul + static fp8 quantisation in ONE launch: w = [gate rows | up rows] ([N, K]);
 * q_out e4m3 [M, N / 2] = fp8(T(silu(T(gate)) * T(up)) * (1 / *q_scale)) with T = dtype -- the bits of cutlass_scaled_mm
 * -> silu_and_mul (activation_kernels.cu:12-75) -> static_scaled_fp8_quant (fp8/common.cu:187-199).
 * aphro_fp8_gemm_stream_silu_supported: 1 when the shape is served (N % 32 == 0 and aphro_fp8_gemm_stream_ksplit == 1). */
int aphro_fp8_gemm_stream_silu_supported(int64_t M, int64_t N, int64_t K);
int aphro_fp8_gemm_stream_silu_quant(const void* a, int64_t lda, const void* w, const float* a_scales,
                                     const float* b_scales, const void* bias, void* q_out, const float* q_scale,
                                     int64_t M, int64_t N, int64_t K, int a_scale_per_token, int b_scale_per_channel,
                                     int dtype, void* stream);

/* One grouped FP8 W8A8 GEMM of a mixture-of-experts layer: the reference's Triton fused_moe_kernel with use_fp8_w8a8
 * (aphrodite/modeling/layers/fused_moe/fused_moe.py:20-170; called twice by fused_experts :566-690 for
 * Fp8MoEMethod.apply, quantization/fp8.py:468-503).  For every valid slot s of the expert-sorted list
 * (moe_align_block_size with block 16): c[s, :] = T(((a[s / top_k_div, :] . w[expert]^T) * topk_weights[s]) * a_scale *
 * b_scales[expert]).  a e4m3 [rows, K]; w e4m3 [E, N, K]; a_scale [1]; b_scales [E]; topk_weights [num_valid] or NULL;
 * c [num_valid, N] f16 / bf16; max_blocks = length of expert_ids.  K % 128 == 0, N % 16 == 0.  csrc/fp8_moe.hip. */
int aphro_fp8_moe_gemm(const void* a, const void* w, const float* a_scale, const float* b_scales, const float* topk_weights,
                       const int32_t* sorted_ids, const int32_t* expert_ids, const int32_t* num_post_pad, void* c,
                       int64_t num_valid, int64_t N, int64_t K, int64_t max_blocks, int top_k_div, int dtype, void* stream);

/* Decode-time LM head with the greedy argmax folded in: out_ids[m] = argmax_v round_T(hidden[m, :] . weight[v, :]) for
 * M <= 32 rows, one launch.  The reference computes the logits with a library GEMM (LogitsProcessor._get_logits,
 * modeling/layers/logits_processor.py:78-96: lm_head.linear_method.apply) and Sampler._greedy_sample takes torch.argmax
 * (modeling/layers/sampler.py); this entry is what a greedy-only decode batch needs of the two: the fp32 sums are rounded
 * to the activation dtype before they are compared (the values the reference's argmax sees), ties go to the lowest index,
 * NaN never wins (= aphro_argmax_rows).  logits != NULL also stores them ([M, ldl], columns < V).  weight: [V, ldw]
 * row-major 16-bit (K contiguous), V >= 16; K in {1024, 2048, 3072, 4096}.  HBM-bound (every weight byte read once).
 * APHRO_ERR_WORKSPACE without a launch when the library-owned scratch cannot be allocated (first call under a stream
 * capture).  TP: a vocabulary-parallel head needs the (value, index) pairs of all ranks -- callers with tp > 1 keep the
 * GEMM + all-gather + argmax path. */
int aphro_lm_head_argmax_supported(int64_t M, int64_t K, int64_t V, int64_t ldw, int dtype);
int aphro_lm_head_argmax(const void* hidden, int64_t lda, const void* weight, int64_t ldw, void* logits, int64_t ldl,
                         int64_t* out_ids, int64_t M, int64_t K, int64_t V, int dtype, void* stream);

/* Greedy sampling: out[r] = argmax_c x[r][c] (lowest index on ties) -- the torch.argmax of
 * modeling/layers/sampler.py:_greedy_sample as one pass over each logits row. */
int aphro_argmax_rows(int64_t* out, const void* x, int64_t rows, int64_t cols,
                      int64_t row_stride, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* APHRODITE_MI355X_H_ */
