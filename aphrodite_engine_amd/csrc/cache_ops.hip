// KV-cache management ops (SURVEY 8b schema list, 8f row 4):
//   copy_blocks              kernels/cache_kernels.cu:66-148   (copy-on-write of forked sequences)
//   swap_blocks              kernels/cache_kernels.cu:24-63    (GPU <-> CPU / GPU <-> GPU block moves)
//   reshape_and_cache_flash  kernels/cache_kernels.cu:207-330  ([NB, block, H, hd] cache layout)
// All byte movers: copy_blocks runs 16-byte lanes over (layer, pair) workgroups,
// swap_blocks is a batch of async copies on the caller's stream.
#include "common.h"

namespace aphro {

// grid (num_layers, num_pairs); a block of the cache is `block_bytes` (multiple of 16 when VEC16)
template <bool VEC16>
__global__ void copy_blocks_kernel(const int64_t* __restrict__ key_cache_ptrs,
                                   const int64_t* __restrict__ value_cache_ptrs,
                                   const int64_t* __restrict__ block_mapping, int64_t block_bytes) {
  const int layer = blockIdx.x, pair = blockIdx.y;
  const int64_t src = block_mapping[2 * pair], dst = block_mapping[2 * pair + 1];
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    uint8_t* base = reinterpret_cast<uint8_t*>(which ? value_cache_ptrs[layer] : key_cache_ptrs[layer]);
    const uint8_t* s = base + src * block_bytes;
    uint8_t* d = base + dst * block_bytes;
    if constexpr (VEC16) {
      const int64_t n = block_bytes >> 4;
      for (int64_t i = threadIdx.x; i < n; i += blockDim.x)
        reinterpret_cast<u32x4*>(d)[i] = reinterpret_cast<const u32x4*>(s)[i];
    } else {
      for (int64_t i = threadIdx.x; i < block_bytes; i += blockDim.x) d[i] = s[i];
    }
  }
}

template <typename T, int KV>
__global__ void reshape_and_cache_flash_kernel(const typename T::storage* __restrict__ key,
                                               const typename T::storage* __restrict__ value,
                                               void* __restrict__ key_cache, void* __restrict__ value_cache,
                                               const int64_t* __restrict__ slot_mapping, int64_t block_stride,
                                               int64_t key_stride, int64_t value_stride, int num_heads,
                                               int head_size, int block_size, float k_scale, float v_scale) {
  const int64_t token = blockIdx.x;
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;  // padded token (cache_kernels.cu:220-223)
  const int64_t blk = slot / block_size, off = slot % block_size;
  const int n = num_heads * head_size;
  const int64_t dst0 = blk * block_stride + off * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    if constexpr (KV == 0) {
      ((typename T::storage*)key_cache)[dst0 + i] = key[token * key_stride + i];
      ((typename T::storage*)value_cache)[dst0 + i] = value[token * value_stride + i];
    } else {
      const float kf = T::to_f32(key[token * key_stride + i]) / k_scale;
      const float vf = T::to_f32(value[token * value_stride + i]) / v_scale;
      ((uint8_t*)key_cache)[dst0 + i] = (uint8_t)f32x2_to_fp8<KV == 2>(kf, 0.f);
      ((uint8_t*)value_cache)[dst0 + i] = (uint8_t)f32x2_to_fp8<KV == 2>(vf, 0.f);
    }
  }
}

}  // namespace aphro

using namespace aphro;

extern "C" int aphro_copy_blocks(const int64_t* key_cache_ptrs, const int64_t* value_cache_ptrs, int num_layers,
                                 const int64_t* block_mapping, int64_t num_pairs, int64_t block_bytes,
                                 void* stream) {
  APHRO_CHECK(num_layers >= 0 && num_pairs >= 0 && block_bytes > 0, "copy_blocks: bad sizes");
  if (num_layers == 0 || num_pairs == 0) return APHRO_OK;
  APHRO_CHECK(num_pairs <= 65535, "copy_blocks: at most 65535 pairs per call");
  dim3 grid((unsigned)num_layers, (unsigned)num_pairs), block(256);
  if (block_bytes % 16 == 0)
    hipLaunchKernelGGL(copy_blocks_kernel<true>, grid, block, 0, (hipStream_t)stream, key_cache_ptrs,
                       value_cache_ptrs, block_mapping, block_bytes);
  else
    hipLaunchKernelGGL(copy_blocks_kernel<false>, grid, block, 0, (hipStream_t)stream, key_cache_ptrs,
                       value_cache_ptrs, block_mapping, block_bytes);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_swap_blocks(const void* src, void* dst, const int64_t* block_mapping_host, int64_t num_pairs,
                                 int64_t block_bytes, int kind, void* stream) {
  APHRO_CHECK(kind >= 0 && kind <= 2, "swap_blocks: Invalid device combination");
  APHRO_CHECK(block_bytes > 0 && num_pairs >= 0, "swap_blocks: bad sizes");
  const hipMemcpyKind k = kind == 0 ? hipMemcpyDeviceToDevice : kind == 1 ? hipMemcpyDeviceToHost
                                                                          : hipMemcpyHostToDevice;
  for (int64_t i = 0; i < num_pairs; ++i) {
    const int64_t s = block_mapping_host[2 * i], d = block_mapping_host[2 * i + 1];
    hipError_t e = hipMemcpyAsync((char*)dst + d * block_bytes, (const char*)src + s * block_bytes,
                                  (size_t)block_bytes, k, (hipStream_t)stream);
    if (e != hipSuccess) {
      set_error("swap_blocks: hipMemcpyAsync failed: %s", hipGetErrorString(e));
      return APHRO_ERR_LAUNCH;
    }
  }
  return APHRO_OK;
}

extern "C" int aphro_reshape_and_cache_flash(const void* key, const void* value, void* key_cache,
                                             void* value_cache, const int64_t* slot_mapping, int64_t num_tokens,
                                             int num_heads, int head_size, int block_size, int64_t block_stride,
                                             int64_t key_stride, int64_t value_stride, int dtype, int kv_dtype,
                                             float k_scale, float v_scale, void* stream) {
  APHRO_CHECK(dtype >= APHRO_F16 && dtype <= APHRO_F32, "reshape_and_cache_flash: unsupported dtype %d", dtype);
  APHRO_CHECK(kv_dtype >= APHRO_KV_AUTO && kv_dtype <= APHRO_KV_FP8_E5M2,
              "Unsupported data type of kv cache: %d", kv_dtype);
  if (num_tokens == 0) return APHRO_OK;
  const int n = num_heads * head_size;
  dim3 grid((unsigned)num_tokens), block((unsigned)(n < 512 ? (n + 63) / 64 * 64 : 512));
  hipStream_t st = (hipStream_t)stream;
#define APHRO_RCF(TT, KVV)                                                                              \
  hipLaunchKernelGGL((reshape_and_cache_flash_kernel<TT, KVV>), grid, block, 0, st,                      \
                     (const typename TT::storage*)key, (const typename TT::storage*)value, key_cache,    \
                     value_cache, slot_mapping, block_stride, key_stride, value_stride, num_heads,       \
                     head_size, block_size, k_scale, v_scale)
#define APHRO_RCF_T(KVV)                                  \
  if (dtype == APHRO_F16) APHRO_RCF(Half, KVV);           \
  else if (dtype == APHRO_BF16) APHRO_RCF(BFloat, KVV);   \
  else APHRO_RCF(Float, KVV);
  if (kv_dtype == APHRO_KV_AUTO) { APHRO_RCF_T(0) }
  else if (kv_dtype == APHRO_KV_FP8_E4M3) { APHRO_RCF_T(1) }
  else { APHRO_RCF_T(2) }
#undef APHRO_RCF_T
#undef APHRO_RCF
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}


// ---- weight prefetch (tensor-parallel overlap) -------------------------------------------------------------------
// While a row-parallel projection's all-reduce crosses the xGMI links (latency bound, no HBM traffic) the compute
// stream pulls the NEXT projection's packed weights through the memory-side Infinity Cache (256 MiB), so that the
// GEMM that follows the all-reduce finds them on die.  Plain 16-byte loads, nothing is written.
namespace aphro {
// 4 independent 16-byte loads per thread and trip (64 bytes in flight per lane): a small grid (one or two workgroups per CU,
// APHRO_PREFETCH_BLOCKS) streams at the HBM rate and leaves the wave slots to the kernels it runs beside.
__global__ __launch_bounds__(256) void prefetch_kernel(const u32x4* __restrict__ p, size_t n16, uint32_t* __restrict__ sink) {
  u32x4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const u32x4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
    acc ^= a ^ b ^ c ^ d;
  }
  for (; i < n16; i += stride) acc ^= p[i];
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u && sink != nullptr) *sink = 1;   // keeps the loads alive
}
}  // namespace aphro

// Streams [ptr, ptr + bytes) through the memory-side cache.  A pointer that is not 16-byte aligned is rounded up (the
// first bytes are skipped: a prefetch is a hint, it must never fail a decode step).
extern "C" int aphro_prefetch(const void* ptr, size_t bytes, void* stream) {
  const uintptr_t a = ((uintptr_t)ptr + 15) & ~(uintptr_t)15;
  const size_t skip = (size_t)(a - (uintptr_t)ptr);
  if (bytes <= skip) return APHRO_OK;
  const size_t n16 = (bytes - skip) / 16;
  if (n16 == 0) return APHRO_OK;
  const int max_blocks = APHRO_LAB_ENV_INT("APHRO_PREFETCH_BLOCKS", 512);
  unsigned blocks = (unsigned)((n16 + 256 * 4 - 1) / (256 * 4));
  if (blocks > (unsigned)max_blocks) blocks = (unsigned)max_blocks;
  hipLaunchKernelGGL(aphro::prefetch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)a, n16,
                     (uint32_t*)nullptr);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
