// Activation -> FP8 (OCP e4m3fn) quantisation for gfx950.
// Reference: kernels/quantization/fp8/common.cu:72-321 (NVIDIA/OCP branch:
// FP8_E4M3_MAX = 448; the MI300 fnuz branch with max 224 does not apply to
// CDNA4 whose converters are OCP).
#include "common.h"

namespace aphro {

constexpr float FP8_MAX = 448.f;

__device__ __forceinline__ uint8_t to_e4m3(float x) {
  // fmax(-MAX, fmin(x, MAX)) then RNE convert (common.cu:56-58)
  float r = __builtin_fmaxf(-FP8_MAX, __builtin_fminf(x, FP8_MAX));
  return (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(r, 0.f, 0, false) & 0xff);
}

template <typename T>
__device__ __forceinline__ void load4(const typename T::storage* p, float (&v)[4]) {
  if constexpr (sizeof(typename T::storage) == 2) {
    u16x4 r = *reinterpret_cast<const u16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = T::to_f32(r[i]);
  } else {
    f32x4 r = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = r[i];
  }
}

__device__ __forceinline__ uint32_t pack4_e4m3(const float (&v)[4], float s, bool inverted) {
  float a[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float x = inverted ? v[i] * s : v[i] / s;
    a[i] = __builtin_fmaxf(-FP8_MAX, __builtin_fminf(x, FP8_MAX));
  }
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a[0], a[1], 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a[2], a[3], w, true);
  return (uint32_t)w;
}

// out = fp8(x * (1 / *scale))   (common.cu:187-199)
template <typename T>
__global__ void static_quant_kernel(uint8_t* __restrict__ out, const typename T::storage* __restrict__ in,
                                    const float* __restrict__ scale, int64_t n) {
  const float inv = 1.0f / (*scale);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t nvec = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    float v[4];
    load4<T>(in + 4 * i, v);
    reinterpret_cast<uint32_t*>(out)[i] = pack4_e4m3(v, inv, true);
  }
  for (int64_t i = nvec * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = to_e4m3(T::to_f32(in[i]) * inv);
}

// *scale = max(*scale, absmax / 448) via integer atomicMax on the (non-negative)
// float bits (common.cu:36-44, 72-140).  *scale must start <= 0.
template <typename T>
__global__ void absmax_scale_kernel(float* __restrict__ scale, const typename T::storage* __restrict__ in,
                                    int64_t n) {
  __shared__ float red[16];
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t nvec = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    float v[4];
    load4<T>(in + 4 * i, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) m = __builtin_fmaxf(m, __builtin_fabsf(v[j]));
  }
  for (int64_t i = nvec * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    m = __builtin_fmaxf(m, __builtin_fabsf(T::to_f32(in[i])));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = __builtin_fmaxf(m, red[w]);
    atomicMax(reinterpret_cast<int*>(scale), __builtin_bit_cast(int, m / FP8_MAX));
  }
}

// Atomic-free dynamic per-tensor scale: every workgroup leaves its |x| maximum in partials[blockIdx.x]
// (no zero-initialised scale, no atomicMax, hence no fill launch) ...
template <typename T>
__global__ void absmax_partial_kernel(float* __restrict__ partials, const typename T::storage* __restrict__ in,
                                      int64_t n) {
  __shared__ float red[16];
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t nvec = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    float v[4];
    load4<T>(in + 4 * i, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) m = __builtin_fmaxf(m, __builtin_fabsf(v[j]));
  }
  for (int64_t i = nvec * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    m = __builtin_fmaxf(m, __builtin_fabsf(T::to_f32(in[i])));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = __builtin_fmaxf(m, red[w]);
    partials[blockIdx.x] = m;
  }
}

// ... and every workgroup of the quantisation pass reduces the (<= 2048) partials itself:
// scale = max|x| / 448 exactly as absmax_scale_kernel computes it (max is order independent).
template <typename T>
__global__ void dynamic_quant_kernel(uint8_t* __restrict__ out, float* __restrict__ scale,
                                     const float* __restrict__ partials, int nparts,
                                     const typename T::storage* __restrict__ in, int64_t n) {
  __shared__ float red[16];
  float m = 0.f;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) m = __builtin_fmaxf(m, partials[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = __builtin_fmaxf(m, red[w]);
  const float s = m / FP8_MAX;
  if (blockIdx.x == 0 && threadIdx.x == 0) *scale = s;
  const float inv = 1.0f / s;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t nvec = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    float v[4];
    load4<T>(in + 4 * i, v);
    reinterpret_cast<uint32_t*>(out)[i] = pack4_e4m3(v, inv, true);
  }
  for (int64_t i = nvec * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = to_e4m3(T::to_f32(in[i]) * inv);
}

// one workgroup per token (common.cu:201-256)
template <typename T>
__global__ void per_token_quant_kernel(uint8_t* __restrict__ out, float* __restrict__ scales,
                                       const typename T::storage* __restrict__ in,
                                       const float* __restrict__ scale_ub, int hidden) {
  __shared__ float red[16];
  __shared__ float tok_scale;
  const int64_t tok = blockIdx.x;
  const typename T::storage* x = in + tok * hidden;
  uint8_t* o = out + tok * hidden;
  const bool vec = (hidden & 3) == 0;
  float m = 0.f;
  if (vec) {
    for (int i = threadIdx.x; i < (hidden >> 2); i += blockDim.x) {
      float v[4];
      load4<T>(x + 4 * i, v);
#pragma unroll
      for (int j = 0; j < 4; ++j) m = __builtin_fmaxf(m, __builtin_fabsf(v[j]));
    }
  } else {
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) m = __builtin_fmaxf(m, __builtin_fabsf(T::to_f32(x[i])));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = __builtin_fmaxf(m, red[w]);
    if (scale_ub) m = __builtin_fminf(m, *scale_ub);
    const float min_sf = 1.0f / (FP8_MAX * 512.f);
    float s = __builtin_fmaxf(m / FP8_MAX, min_sf);
    tok_scale = s;
    scales[tok] = s;
  }
  __syncthreads();
  const float s = tok_scale;
  if (vec) {
    for (int i = threadIdx.x; i < (hidden >> 2); i += blockDim.x) {
      float v[4];
      load4<T>(x + 4 * i, v);
      reinterpret_cast<uint32_t*>(o)[i] = pack4_e4m3(v, s, false);  // division: matches FBGemm (common.cu:243)
    }
  } else {
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) o[i] = to_e4m3(T::to_f32(x[i]) / s);
  }
}

// Prompt-sized form (round 6): one WAVE per token, the row held in registers between the absmax and the quantisation -- one read
// of the row, no workgroup barrier (the form above reads it twice around two __syncthreads: 1.9 TB/s at 8192 x 4096,
// profiles/r6_prefill_e2e_trace.txt).  Same loads (load4), same max (order-free), same division (pack4_e4m3): same bits.
// hidden % 4 == 0, hidden <= 256 * NQ.
template <typename T, int NQ>
__global__ __launch_bounds__(256) void per_token_quant_wave_kernel(uint8_t* __restrict__ out, float* __restrict__ scales,
                                                                   const typename T::storage* __restrict__ in,
                                                                   const float* __restrict__ scale_ub, int hidden, int64_t tokens) {
  const int lane = threadIdx.x & 63;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= tokens) return;
  const typename T::storage* x = in + tok * hidden;
  const int quads = hidden >> 2;
  float v[NQ][4];
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int q = j * 64 + lane;
    if (q < quads) {
      load4<T>(x + 4 * q, v[j]);
#pragma unroll
      for (int e = 0; e < 4; ++e) m = __builtin_fmaxf(m, __builtin_fabsf(v[j][e]));
    }
  }
  m = wave_max(m);
  if (scale_ub) m = __builtin_fminf(m, *scale_ub);
  const float min_sf = 1.0f / (FP8_MAX * 512.f);
  const float s = __builtin_fmaxf(m / FP8_MAX, min_sf);
  if (lane == 0) scales[tok] = s;
  uint32_t* o = reinterpret_cast<uint32_t*>(out + tok * hidden);
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int q = j * 64 + lane;
    if (q < quads) o[q] = pack4_e4m3(v[j], s, false);   // division: matches FBGemm (common.cu:243)
  }
}

}  // namespace aphro

using namespace aphro;

#define DISPATCH_IN(dtype, CALL)                  \
  if (dtype == APHRO_F16) { CALL(Half); }         \
  else if (dtype == APHRO_BF16) { CALL(BFloat); } \
  else { CALL(Float); }

static unsigned blocks_for(int64_t n) {
  int64_t b = (n / 4 + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

extern "C" int aphro_static_scaled_fp8_quant(void* out, const void* input, const float* scale, int64_t M,
                                             int64_t K, int dtype, void* stream) {
  APHRO_CHECK(dtype >= APHRO_F16 && dtype <= APHRO_F32, "scaled_fp8_quant: unsupported dtype %d", dtype);
  int64_t n = M * K;
  if (n == 0) return APHRO_OK;
#define CALL(TT)                                                                                        \
  hipLaunchKernelGGL((static_quant_kernel<TT>), dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, \
                     (uint8_t*)out, (const typename TT::storage*)input, scale, n)
  DISPATCH_IN(dtype, CALL)
#undef CALL
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_dynamic_scaled_fp8_quant(void* out, const void* input, float* scale, int64_t M,
                                              int64_t K, int dtype, void* stream) {
  APHRO_CHECK(dtype >= APHRO_F16 && dtype <= APHRO_F32, "scaled_fp8_quant: unsupported dtype %d", dtype);
  int64_t n = M * K;
  if (n == 0) return APHRO_OK;
#define CALL(TT)                                                                                        \
  hipLaunchKernelGGL((absmax_scale_kernel<TT>), dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, \
                     scale, (const typename TT::storage*)input, n)
  DISPATCH_IN(dtype, CALL)
#undef CALL
  APHRO_LAUNCH_CHECK();
  return aphro_static_scaled_fp8_quant(out, input, scale, M, K, dtype, stream);
}

// Same result as aphro_dynamic_scaled_fp8_quant with caller scratch (>= 2048 floats) instead of a
// zero-initialised scale + atomics: two launches instead of fill + absmax + quant.
extern "C" int aphro_dynamic_scaled_fp8_quant_ws(void* out, const void* input, float* scale, float* partials,
                                                 size_t partials_bytes, int64_t M, int64_t K, int dtype,
                                                 void* stream) {
  APHRO_CHECK(dtype >= APHRO_F16 && dtype <= APHRO_F32, "scaled_fp8_quant: unsupported dtype %d", dtype);
  int64_t n = M * K;
  if (n == 0) return APHRO_OK;
  const unsigned nb = blocks_for(n);
  if (partials == nullptr || partials_bytes < nb * sizeof(float)) {
    set_error("scaled_fp8_quant: scratch %zu < %zu bytes", partials_bytes, (size_t)nb * sizeof(float));
    return APHRO_ERR_WORKSPACE;
  }
#define CALL(TT)                                                                                          \
  hipLaunchKernelGGL((absmax_partial_kernel<TT>), dim3(nb), dim3(256), 0, (hipStream_t)stream, partials,   \
                     (const typename TT::storage*)input, n);                                              \
  hipLaunchKernelGGL((dynamic_quant_kernel<TT>), dim3(nb), dim3(256), 0, (hipStream_t)stream,              \
                     (uint8_t*)out, scale, (const float*)partials, (int)nb, (const typename TT::storage*)input, n)
  DISPATCH_IN(dtype, CALL)
#undef CALL
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_dynamic_per_token_scaled_fp8_quant(void* out, const void* input, float* scales,
                                                        const float* scale_ub, int64_t M, int64_t K,
                                                        int dtype, void* stream) {
  APHRO_CHECK(dtype >= APHRO_F16 && dtype <= APHRO_F32, "scaled_fp8_quant: unsupported dtype %d", dtype);
  if (M == 0 || K == 0) return APHRO_OK;
  // prompt-sized calls: one wave per token, the row in registers (decode-sized calls keep a whole workgroup per row: latency)
  if (M >= 256 && dtype != APHRO_F32 && K % 4 == 0 && K <= 8192 && ((uintptr_t)input % 8) == 0 && ((uintptr_t)out % 4) == 0) {
    const dim3 wgrid((unsigned)((M + 3) / 4));
#define CALLW(TT)                                                                                                  \
    if (K <= 1024) hipLaunchKernelGGL((per_token_quant_wave_kernel<TT, 4>), wgrid, dim3(256), 0, (hipStream_t)stream,  \
                                      (uint8_t*)out, scales, (const typename TT::storage*)input, scale_ub, (int)K, M); \
    else if (K <= 2048) hipLaunchKernelGGL((per_token_quant_wave_kernel<TT, 8>), wgrid, dim3(256), 0, (hipStream_t)stream, \
                                      (uint8_t*)out, scales, (const typename TT::storage*)input, scale_ub, (int)K, M); \
    else if (K <= 4096) hipLaunchKernelGGL((per_token_quant_wave_kernel<TT, 16>), wgrid, dim3(256), 0, (hipStream_t)stream, \
                                      (uint8_t*)out, scales, (const typename TT::storage*)input, scale_ub, (int)K, M); \
    else hipLaunchKernelGGL((per_token_quant_wave_kernel<TT, 32>), wgrid, dim3(256), 0, (hipStream_t)stream,           \
                            (uint8_t*)out, scales, (const typename TT::storage*)input, scale_ub, (int)K, M)
    if (dtype == APHRO_F16) { CALLW(Half); } else { CALLW(BFloat); }
#undef CALLW
    APHRO_LAUNCH_CHECK();
    return APHRO_OK;
  }
  int threads = K >= 4096 ? 1024 : (K >= 1024 ? 256 : 64);
#define CALL(TT)                                                                                          \
  hipLaunchKernelGGL((per_token_quant_kernel<TT>), dim3((unsigned)M), dim3(threads), 0, (hipStream_t)stream, \
                     (uint8_t*)out, scales, (const typename TT::storage*)input, scale_ub, (int)K)
  DISPATCH_IN(dtype, CALL)
#undef CALL
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
