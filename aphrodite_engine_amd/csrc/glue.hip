// Memory-bound glue ops between the hot kernels:
//   rms_norm / fused_add_rms_norm   kernels/layernorm_kernels.cu:17-45, 200-240, 282-352
//   silu_and_mul                    kernels/activation_kernels.cu:12-75
//   rotary_embedding                kernels/pos_encoding_kernels.cu:10-160
// One workgroup per token, 16-byte vector accesses, fp32 math, wave64 reductions.
#include "common.h"

namespace aphro {

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int nw = blockDim.x >> 6;
  if (nw == 1) return v;
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < nw; ++w) t += red[w];
  __syncthreads();
  return t;
}

// ADD: residual' = input + residual (stored), input' = norm(residual') * w  (in place)
template <typename T, bool ADD>
__global__ void rms_norm_kernel(uint16_t* __restrict__ out, uint16_t* __restrict__ input,
                                uint16_t* __restrict__ residual, const uint16_t* __restrict__ weight,
                                float eps, int hidden, int64_t in_stride) {
  __shared__ float red[16];
  const int64_t tok = blockIdx.x;
  uint16_t* x = input + tok * in_stride;
  uint16_t* res = ADD ? residual + tok * (int64_t)hidden : nullptr;
  uint16_t* o = ADD ? x : out + tok * (int64_t)hidden;
  const int nv = hidden >> 3;
  // up to 2 vectors of 8 per thread are kept in registers (hidden <= 16 * blockDim)
  float v[2][8];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) {
      u16x8 a = *reinterpret_cast<const u16x8*>(x + 8 * i);
      if constexpr (ADD) {
        u16x8 b = *reinterpret_cast<const u16x8*>(res + 8 * i);
        u16x8 s;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // the sum is rounded to the storage type before the statistics
          // (layernorm_kernels.cu:214-218)
          uint16_t r = T::from_f32(T::to_f32(a[j]) + T::to_f32(b[j]));
          s[j] = r;
          v[it][j] = T::to_f32(r);
        }
        *reinterpret_cast<u16x8*>(res + 8 * i) = s;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[it][j] = T::to_f32(a[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v[it][j] * v[it][j];
    }
  }
  ss = block_sum(ss, red);
  const float inv = __frsqrt_rn(ss / (float)hidden + eps);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) {
      u16x8 w = *reinterpret_cast<const u16x8*>(weight + 8 * i);
      u16x8 r;
#pragma unroll
      for (int j = 0; j < 8; ++j)  // ((scalar_t)(x * s_variance)) * weight
        r[j] = T::from_f32(T::to_f32(T::from_f32(v[it][j] * inv)) * T::to_f32(w[j]));
      *reinterpret_cast<u16x8*>(o + 8 * i) = r;
    }
  }
}

template <typename T>
__global__ void silu_and_mul_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ in, int d) {
  const int64_t tok = blockIdx.x;
  const uint16_t* a = in + tok * 2 * (int64_t)d;
  const uint16_t* b = a + d;
  uint16_t* o = out + tok * (int64_t)d;
  for (int i = threadIdx.x; i < (d >> 3); i += blockDim.x) {
    u16x8 x = *reinterpret_cast<const u16x8*>(a + 8 * i);
    u16x8 y = *reinterpret_cast<const u16x8*>(b + 8 * i);
    u16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float xf = T::to_f32(x[j]);
      // (T)(x / (1 + exp(-x))) * y   (activation_kernels.cu:14-17, 25-28)
      float s = T::to_f32(T::from_f32(xf / (1.0f + __expf(-xf))));
      r[j] = T::from_f32(s * T::to_f32(y[j]));
    }
    *reinterpret_cast<u16x8*>(o + 8 * i) = r;
  }
}

// The same arithmetic on a gate_up GEMM output whose columns are interleaved, (gate_j, up_j) pairs -- what the layer's
// weights look like once SiluAndMul rides in the decode GEMM's epilogue (ops.interleave_gate_up): the prompt-sized GEMMs run
// on the SAME copy of the weights and this kernel pairs the columns back up, so that no second [gate | up] copy of the
// matrix has to stay in HBM.  Same bits as silu_and_mul_kernel on the de-interleaved input.
template <typename T>
__global__ void silu_and_mul_interleaved_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ in, int d) {
  const int64_t tok = blockIdx.x;
  const uint16_t* a = in + tok * 2 * (int64_t)d;
  uint16_t* o = out + tok * (int64_t)d;
  for (int i = threadIdx.x; i < (d >> 3); i += blockDim.x) {
    const u16x8 lo = *reinterpret_cast<const u16x8*>(a + 16 * i);
    const u16x8 hi = *reinterpret_cast<const u16x8*>(a + 16 * i + 8);
    u16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint16_t g = j < 4 ? lo[2 * j] : hi[2 * j - 8], u = j < 4 ? lo[2 * j + 1] : hi[2 * j - 7];
      const float xf = T::to_f32(g);
      const float s = T::to_f32(T::from_f32(xf / (1.0f + __expf(-xf))));
      r[j] = T::from_f32(s * T::to_f32(u));
    }
    *reinterpret_cast<u16x8*>(o + 8 * i) = r;
  }
}

template <typename T, bool NEOX>
__global__ void rotary_kernel(const int64_t* __restrict__ positions, uint16_t* __restrict__ query,
                              uint16_t* __restrict__ key, const uint16_t* __restrict__ cache, int rot_dim,
                              int64_t query_stride, int64_t key_stride, int num_heads, int num_kv_heads,
                              int head_size) {
  const int64_t tok = blockIdx.x;
  const int64_t pos = positions[tok];
  const uint16_t* cs = cache + pos * rot_dim;
  const int embed = rot_dim >> 1;
  const int nq = num_heads * embed;
  const int nk = num_kv_heads * embed;
  for (int i = threadIdx.x; i < nq + nk; i += blockDim.x) {
    const bool is_k = i >= nq;
    const int ii = is_k ? i - nq : i;
    const int h = ii / embed, r = ii % embed;
    uint16_t* arr = is_k ? key + tok * key_stride + (int64_t)h * head_size
                         : query + tok * query_stride + (int64_t)h * head_size;
    const int xi = NEOX ? r : 2 * r;
    const int yi = NEOX ? embed + r : 2 * r + 1;
    const float c = T::to_f32(cs[r]);
    const float s = T::to_f32(cs[embed + r]);
    const float x = T::to_f32(arr[xi]);
    const float y = T::to_f32(arr[yi]);
    float xo, yo;
    rope_pair(x, y, c, s, xo, yo);
    arr[xi] = T::from_f32(xo);
    arr[yi] = T::from_f32(yo);
  }
}

}  // namespace aphro

using namespace aphro;

static int norm_threads(int hidden) {
  // same thread -> element mapping as add_rms_norm_pack_kernel (fused_decode.hip): the
  // sum-of-squares reduction order, hence every output bit, is shared by the two paths
  int nv = hidden / 8;
  int t = nv <= 1024 ? nv : (nv + 1) / 2;
  t = (t + 63) / 64 * 64;
  return t < 64 ? 64 : (t > 1024 ? 1024 : t);
}

extern "C" int aphro_rms_norm(void* out, const void* input, const void* weight, float eps, int64_t num_tokens,
                              int hidden, int64_t in_stride, int dtype, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "rms_norm: dtype must be f16 or bf16");
  APHRO_CHECK(hidden % 8 == 0 && hidden <= 16384 && in_stride % 8 == 0, "rms_norm: hidden=%d unsupported", hidden);
  if (num_tokens == 0) return APHRO_OK;
  dim3 grid((unsigned)num_tokens), block(norm_threads(hidden));
  if (dtype == APHRO_F16)
    hipLaunchKernelGGL((rms_norm_kernel<Half, false>), grid, block, 0, (hipStream_t)stream, (uint16_t*)out,
                       (uint16_t*)input, (uint16_t*)nullptr, (const uint16_t*)weight, eps, hidden, in_stride);
  else
    hipLaunchKernelGGL((rms_norm_kernel<BFloat, false>), grid, block, 0, (hipStream_t)stream, (uint16_t*)out,
                       (uint16_t*)input, (uint16_t*)nullptr, (const uint16_t*)weight, eps, hidden, in_stride);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_fused_add_rms_norm(void* input, void* residual, const void* weight, float eps,
                                        int64_t num_tokens, int hidden, int dtype, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "fused_add_rms_norm: dtype must be f16 or bf16");
  APHRO_CHECK(hidden % 8 == 0 && hidden <= 16384, "fused_add_rms_norm: hidden=%d unsupported", hidden);
  if (num_tokens == 0) return APHRO_OK;
  dim3 grid((unsigned)num_tokens), block(norm_threads(hidden));
  if (dtype == APHRO_F16)
    hipLaunchKernelGGL((rms_norm_kernel<Half, true>), grid, block, 0, (hipStream_t)stream, (uint16_t*)nullptr,
                       (uint16_t*)input, (uint16_t*)residual, (const uint16_t*)weight, eps, hidden,
                       (int64_t)hidden);
  else
    hipLaunchKernelGGL((rms_norm_kernel<BFloat, true>), grid, block, 0, (hipStream_t)stream, (uint16_t*)nullptr,
                       (uint16_t*)input, (uint16_t*)residual, (const uint16_t*)weight, eps, hidden,
                       (int64_t)hidden);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_silu_and_mul(void* out, const void* input, int64_t num_tokens, int d, int dtype,
                                  void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "silu_and_mul: dtype must be f16 or bf16");
  APHRO_CHECK(d % 8 == 0, "silu_and_mul: d=%d must be a multiple of 8", d);
  if (num_tokens == 0) return APHRO_OK;
  int threads = d / 8 >= 1024 ? 1024 : ((d / 8 + 63) / 64 * 64);
  dim3 grid((unsigned)num_tokens), block(threads);
  if (dtype == APHRO_F16)
    hipLaunchKernelGGL((silu_and_mul_kernel<Half>), grid, block, 0, (hipStream_t)stream, (uint16_t*)out,
                       (const uint16_t*)input, d);
  else
    hipLaunchKernelGGL((silu_and_mul_kernel<BFloat>), grid, block, 0, (hipStream_t)stream, (uint16_t*)out,
                       (const uint16_t*)input, d);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

// out [tokens, d] = SiluAndMul of in [tokens, 2 d] with interleaved (gate_j, up_j) columns.
extern "C" int aphro_silu_and_mul_interleaved(void* out, const void* input, int64_t num_tokens, int d, int dtype,
                                  void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "silu_and_mul_interleaved: dtype must be f16 or bf16");
  APHRO_CHECK(d % 8 == 0, "silu_and_mul_interleaved: d=%d must be a multiple of 8", d);
  if (num_tokens == 0) return APHRO_OK;
  int threads = d / 8 >= 1024 ? 1024 : ((d / 8 + 63) / 64 * 64);
  dim3 grid((unsigned)num_tokens), block(threads);
  if (dtype == APHRO_F16)
    hipLaunchKernelGGL((silu_and_mul_interleaved_kernel<Half>), grid, block, 0, (hipStream_t)stream, (uint16_t*)out,
                       (const uint16_t*)input, d);
  else
    hipLaunchKernelGGL((silu_and_mul_interleaved_kernel<BFloat>), grid, block, 0, (hipStream_t)stream, (uint16_t*)out,
                       (const uint16_t*)input, d);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_rotary_embedding(const int64_t* positions, void* query, void* key, int64_t num_tokens,
                                      int num_heads, int num_kv_heads, int head_size, int rot_dim,
                                      const void* cos_sin_cache, int64_t query_stride, int64_t key_stride,
                                      int is_neox, int dtype, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "rotary_embedding: dtype must be f16 or bf16");
  APHRO_CHECK(rot_dim % 2 == 0 && rot_dim <= head_size, "rotary_embedding: bad rot_dim");
  if (num_tokens == 0) return APHRO_OK;
  int work = (num_heads + num_kv_heads) * rot_dim / 2;
  int threads = work >= 512 ? 512 : (work + 63) / 64 * 64;
  dim3 grid((unsigned)num_tokens), block(threads);
#define RL(TT, NX)                                                                                        \
  hipLaunchKernelGGL((rotary_kernel<TT, NX>), grid, block, 0, (hipStream_t)stream, positions,              \
                     (uint16_t*)query, (uint16_t*)key, (const uint16_t*)cos_sin_cache, rot_dim, query_stride, \
                     key_stride, num_heads, num_kv_heads, head_size)
  if (dtype == APHRO_F16) { if (is_neox) RL(Half, true); else RL(Half, false); }
  else { if (is_neox) RL(BFloat, true); else RL(BFloat, false); }
#undef RL
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
