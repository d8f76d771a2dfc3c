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

// One LDS-DMA instruction (64 lanes x 16 bytes -> 1 KiB at LDS byte address `lds`), hidden from hipcc: behind a DMA it can
// see, hipcc drains vmcnt to 0 in front of the next LDS read of the kernel (it cannot tell the slots apart) -- every piece
// was waited for, the full memory latency exposed, 4 times per phase (SQ_WAIT_ANY 48 % of the wave cycles).  The pieces are
// counted by hand instead (vmcnt before the barrier at the top of a body).  s_nop 4: SGPR operands may be fresh SALU results.
__device__ __forceinline__ void fa_dma16(u32x4 rsrc, uint32_t lds, int voff, int soff) {
  uint32_t keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// a raw buffer descriptor (base, no stride, bytes, dword format) from wave-uniform values, for fa_dma16
__device__ __forceinline__ u32x4 fa_make_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = (uint64_t)base;
  return u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a), (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu),
               (uint32_t)__builtin_amdgcn_readfirstlane(bytes), 0x00020000u};
}

// fourth-generation prefill kernel (flash_attn_v4.hip): p fully set up incl. nqt_max / xcd_remap
int fa_v4_launch(const FAParams& p, int dtype, int batch, hipStream_t st);

}  // namespace aphro
