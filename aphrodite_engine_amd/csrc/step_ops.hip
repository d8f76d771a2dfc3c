// Per-step bookkeeping kernels that keep a whole decode step inside one HIP graph:
//   advance_step_flashattn   kernels/prepare_inputs/advance_step.cu:13-51 (schema kernels/torch_bindings.cpp:77-82)
//   argmax_rows              greedy sampling over the logits (modeling/layers/sampler.py _greedy_sample:
//                            torch.argmax(logprobs, dim=-1)); torch's generic reduce takes ~50 us for
//                            [32, 128256] fp16, this one streams each row once with 16-byte loads.
#include "common.h"

namespace aphro {

__global__ void advance_step_kernel(int num_queries, int block_size, int64_t* __restrict__ input_tokens,
                                    const int64_t* __restrict__ sampled_token_ids,
                                    int64_t* __restrict__ input_positions, int32_t* __restrict__ seq_lens,
                                    int64_t* __restrict__ slot_mapping, const int32_t* __restrict__ block_tables,
                                    int64_t block_tables_stride) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= num_queries) return;
  input_tokens[q] = sampled_token_ids[q];
  const int next_seq_len = seq_lens[q] + 1;
  const int next_pos = next_seq_len - 1;
  seq_lens[q] = next_seq_len;
  input_positions[q] = next_pos;
  const int32_t* bt = block_tables + block_tables_stride * q;
  slot_mapping[q] = (int64_t)bt[next_pos / block_size] * block_size + next_pos % block_size;
}

// one workgroup per row; ties -> the lowest index; NaN never wins (like a max over ordered floats)
template <typename T>
__global__ void argmax_rows_kernel(int64_t* __restrict__ out, const typename T::storage* __restrict__ x,
                                   int64_t cols, int64_t row_stride) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  const typename T::storage* row = x + (size_t)blockIdx.x * row_stride;
  float best = -INFINITY;
  int best_i = 0x7fffffff;
  int64_t nvec = 0;
  if constexpr (sizeof(typename T::storage) == 2) {   // 16-byte loads for the 16-bit types
    nvec = ((uintptr_t)row % 16 == 0) ? cols / 8 : 0;
    for (int64_t i = threadIdx.x; i < nvec; i += blockDim.x) {
      const u16x8 v = *reinterpret_cast<const u16x8*>(row + 8 * i);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = T::to_f32(v[j]);
        if (f > best) { best = f; best_i = (int)(8 * i + j); }   // ascending index inside a thread
      }
    }
  }
  for (int64_t i = nvec * 8 + threadIdx.x; i < cols; i += blockDim.x) {
    const float f = T::to_f32(row[i]);
    if (f > best || (f == best && (int)i < best_i)) { best = f; best_i = (int)i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(best_i, o, 64);
    if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = best_i; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < best_i)) { best = bv[w]; best_i = bi[w]; }
    out[blockIdx.x] = best_i == 0x7fffffff ? 0 : best_i;
  }
}

}  // namespace aphro

using namespace aphro;

extern "C" int aphro_advance_step_flashattn(int num_seqs, int num_queries, int block_size, int64_t* input_tokens,
                                            const int64_t* sampled_token_ids, int64_t* input_positions,
                                            int32_t* seq_lens, int64_t* slot_mapping, const int32_t* block_tables,
                                            int64_t block_tables_stride, void* stream) {
  APHRO_CHECK(num_queries >= 0 && num_queries <= num_seqs && block_size > 0, "advance_step: bad arguments");
  if (num_queries == 0) return APHRO_OK;
  hipLaunchKernelGGL(advance_step_kernel, dim3((unsigned)((num_queries + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, num_queries, block_size, input_tokens, sampled_token_ids,
                     input_positions, seq_lens, slot_mapping, block_tables, block_tables_stride);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_argmax_rows(int64_t* out, const void* x, int64_t rows, int64_t cols, int64_t row_stride,
                                 int dtype, void* stream) {
  APHRO_CHECK(dtype >= APHRO_F16 && dtype <= APHRO_F32, "argmax_rows: unsupported dtype %d", dtype);
  APHRO_CHECK(cols > 0 && cols < 0x7fffffff, "argmax_rows: bad column count");
  if (rows == 0) return APHRO_OK;
  dim3 grid((unsigned)rows), block(1024);
  if (dtype == APHRO_F16)
    hipLaunchKernelGGL((argmax_rows_kernel<Half>), grid, block, 0, (hipStream_t)stream, out, (const uint16_t*)x, cols, row_stride);
  else if (dtype == APHRO_BF16)
    hipLaunchKernelGGL((argmax_rows_kernel<BFloat>), grid, block, 0, (hipStream_t)stream, out, (const uint16_t*)x, cols, row_stride);
  else
    hipLaunchKernelGGL((argmax_rows_kernel<Float>), grid, block, 0, (hipStream_t)stream, out, (const float*)x, cols, row_stride);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
