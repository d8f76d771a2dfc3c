// Shared by the prefill attention kernels (flash_attn.hip, flash_attn_v4.hip).
#pragma once
#include "common.h"

namespace aphro {

struct FAParams {
  void* out;
  const void* q;
  const void* k;
  const void* v;
  const int32_t* cu_seqlens;
  const float* alibi;
  int num_heads, num_kv_heads;
  int64_t q_stride, k_stride, v_stride;
  float scale;
  int causal;
  int debug;
  int nqt_max, xcd_remap;   // third-generation kernel: query tiles per sequence in the grid; kv-head -> XCD placement
  const int32_t* cu_seqlens_k;   // third-generation kernel: key rows per sequence when they differ from the query rows
  int64_t o_stride;              // ... and the output row stride (elements)
  int window;                    // first / second generation kernels: sliding window (keys > query position - window), 0 = off
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* fa_lds_ptr;

template <typename T>
__device__ __forceinline__ f32x16 fa_mfma32(u32x4 a, u32x4 b, f32x16 c) {
  if constexpr (__is_same(T, Half))
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// fourth-generation prefill kernel (flash_attn_v4.hip): p fully set up incl. nqt_max / xcd_remap
int fa_v4_launch(const FAParams& p, int dtype, int batch, hipStream_t st);

}  // namespace aphro
