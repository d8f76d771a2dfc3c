// Fused glue of the FP8 (W8A8, per-token dynamic activation scale) decode path -- BASELINE
// configs[2], the compressed-tensors scheme (compressed_tensors_w8a8_fp8.py:139-148:
// use_per_token_if_dynamic=True).  Each kernel reproduces the unfused op sequence bit for bit:
//   [split-K slab reduce + dequant] + fused_add_rms_norm + dynamic_per_token_scaled_fp8_quant
//   silu_and_mul + dynamic_per_token_scaled_fp8_quant
// Reference semantics: kernels/layernorm_kernels.cu:200-240, kernels/activation_kernels.cu:12-75,
// kernels/quantization/fp8/common.cu:201-256 (scale = max(absmax / 448, 1 / (448 * 512)), x / scale),
// cutlass_scaled_mm epilogue a_scale * (b_scale * acc) (tests/kernels/test_cutlass.py:43).
// STATIC activation scheme (input_scale in the checkpoint: compressed_tensors_w8a8_fp8.py:98-113, fp8.py static
// activation_scheme): the same kernels with `static_scale` given quantise as static_scaled_fp8_quant does -- x * (1 / scale),
// common.cu:187-199 -- skip the absmax reduction, and still fill scale_out[token] so that every consumer is unchanged.
#include "common.h"

namespace aphro {

constexpr float FQ_MAX = 448.f;

__device__ __forceinline__ float fq_block_reduce(float v, float* red, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int nw = blockDim.x >> 6;
  if (nw > 1) {
    __syncthreads();  // red[] may still be read from a previous reduction
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = red[0];
    for (int w = 1; w < nw; ++w) t = is_max ? __builtin_fmaxf(t, red[w]) : t + red[w];
    v = t;
  }
  return v;
}

__device__ __forceinline__ uint32_t fq_pack4(const float (&v)[4], float s) {
  float a[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = __builtin_fmaxf(-FQ_MAX, __builtin_fminf(v[i] / s, FQ_MAX));
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a[0], a[1], 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a[2], a[3], w, true);
  return (uint32_t)w;
}

__device__ __forceinline__ uint32_t fq_pack4_inv(const float (&v)[4], float inv) {   // static scheme: x * (1 / scale)
  float a[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = __builtin_fmaxf(-FQ_MAX, __builtin_fminf(v[i] * inv, FQ_MAX));
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a[0], a[1], 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a[2], a[3], w, true);
  return (uint32_t)w;
}

// x = input (T) or T(sa[tok] * (sb[col] * sum_k slabs[k])); residual' = T(x + residual) (or x);
// y = T(T(residual' * rstd) * w); q = fp8(y / scale), scale = max(absmax(y) / 448, 1/(448*512)).
// Same thread -> element mapping and reduction order as rms_norm_kernel / add_rms_norm_pack_kernel.
// NS > 0 (round 6, as add_rms_norm_pack_kernel got in round 4): x = the sum of exactly NS slabs with every slab piece of a
// thread requested before the first one is waited for -- the run-time loop `for (s = 1; s < nslab; ++s) a += slab[s]`
// compiles to one dependent memory round trip per K slice.  Same summation order, same bits.
template <typename T, int NS = 0>
__global__ void add_rms_norm_quant_kernel(const uint16_t* __restrict__ input, const float* __restrict__ slabs,
                                          int nslab, const float* __restrict__ sa, const float* __restrict__ sb,
                                          int sa_per_token, int sb_per_channel, uint16_t* __restrict__ residual,
                                          int has_residual, const uint16_t* __restrict__ weight, float eps,
                                          uint8_t* __restrict__ q_out, float* __restrict__ scale_out,
                                          uint16_t* __restrict__ out, int tokens, int hidden,
                                          const float* __restrict__ static_scale) {
  __shared__ float red[16];
  const int tok = blockIdx.x;
  const int nv = hidden >> 3;
  const size_t slab_stride = (size_t)tokens * hidden;
  float v[2][8];
  u16x8 wv[2];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) {
      const size_t off = (size_t)tok * hidden + 8 * i;
      wv[it] = *reinterpret_cast<const u16x8*>(weight + 8 * i);
      float x[8];
      if (slabs) {
        f32x4 a, b;
        if constexpr (NS > 0) {
          f32x4 sa4[NS], sb4[NS];
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            sa4[s] = *reinterpret_cast<const f32x4*>(slabs + s * slab_stride + off);
            sb4[s] = *reinterpret_cast<const f32x4*>(slabs + s * slab_stride + off + 4);
          }
          a = sa4[0]; b = sb4[0];
#pragma unroll
          for (int s = 1; s < NS; ++s) { a += sa4[s]; b += sb4[s]; }
        } else {
          a = *reinterpret_cast<const f32x4*>(slabs + off);
          b = *reinterpret_cast<const f32x4*>(slabs + off + 4);
          for (int s = 1; s < nslab; ++s) {
            a += *reinterpret_cast<const f32x4*>(slabs + s * slab_stride + off);
            b += *reinterpret_cast<const f32x4*>(slabs + s * slab_stride + off + 4);
          }
        }
        const float sav = sa ? sa[sa_per_token ? tok : 0] : 1.f;
        f32x4 sb0, sb1;                                   // the 8 column scales: two 16-byte loads, not 8 conditional ones
        if (sb && sb_per_channel) {
          sb0 = *reinterpret_cast<const f32x4*>(sb + 8 * i);
          sb1 = *reinterpret_cast<const f32x4*>(sb + 8 * i + 4);
        } else {
          const float s1 = sb ? sb[0] : 1.f;
          sb0 = f32x4{s1, s1, s1, s1};
          sb1 = sb0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float sbv = j < 4 ? sb0[j] : sb1[j - 4];
          const float acc = j < 4 ? a[j] : b[j - 4];
          x[j] = T::to_f32(from_f32_exact<T>(sav * (sbv * acc)));
        }
      } else {
        u16x8 a = *reinterpret_cast<const u16x8*>(input + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = T::to_f32(a[j]);
      }
      u16x8 rs;
      if (has_residual) {
        u16x8 r = *reinterpret_cast<const u16x8*>(residual + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          rs[j] = from_f32_exact<T>(x[j] + T::to_f32(r[j]));
          v[it][j] = T::to_f32(rs[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          rs[j] = from_f32_exact<T>(x[j]);
          v[it][j] = x[j];
        }
      }
      if (residual) *reinterpret_cast<u16x8*>(residual + off) = rs;
      {
        // squares rounded, then added -- pinned as in add_rms_norm_pack_kernel: the all-reduce + norm + quant launch
        // (custom_all_reduce.hip, Q8 kernels) reproduces this row bit for bit; left alone hipcc contracted two of the eight
#pragma clang fp contract(off)
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += v[it][j] * v[it][j];
      }
    }
  }
  ss = fq_block_reduce(ss, red, false);
  const float inv = __frsqrt_rn(ss / (float)hidden + eps);
  float amax = 0.f;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) {
      u16x8 y;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        y[j] = from_f32_exact<T>(T::to_f32(from_f32_exact<T>(v[it][j] * inv)) * T::to_f32(wv[it][j]));
        v[it][j] = T::to_f32(y[j]);
        amax = __builtin_fmaxf(amax, __builtin_fabsf(v[it][j]));
      }
      if (out) *reinterpret_cast<u16x8*>(out + (size_t)tok * hidden + 8 * i) = y;
    }
  }
  float scale, inv_static = 0.f;
  if (static_scale) {
    scale = *static_scale;
    inv_static = 1.0f / scale;
  } else {
    amax = fq_block_reduce(amax, red, true);
    scale = __builtin_fmaxf(amax / FQ_MAX, 1.0f / (FQ_MAX * 512.f));
  }
  if (threadIdx.x == 0) scale_out[tok] = scale;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) {
      const float lo[4] = {v[it][0], v[it][1], v[it][2], v[it][3]};
      const float hi[4] = {v[it][4], v[it][5], v[it][6], v[it][7]};
      u32x2 q = static_scale ? u32x2{fq_pack4_inv(lo, inv_static), fq_pack4_inv(hi, inv_static)}
                             : u32x2{fq_pack4(lo, scale), fq_pack4(hi, scale)};
      *reinterpret_cast<u32x2*>(q_out + (size_t)tok * hidden + 8 * i) = q;
    }
  }
}

// act = T(T(silu(gate)) * up) over x = [gate | up] (T [tokens, 2d]); q = fp8(act / scale) per token
template <typename T, int VPT>
__global__ void silu_mul_quant_kernel(const uint16_t* __restrict__ in, uint8_t* __restrict__ q_out,
                                      float* __restrict__ scale_out, uint16_t* __restrict__ out, int d,
                                      const float* __restrict__ static_scale) {
  __shared__ float red[16];
  const int tok = blockIdx.x;
  const uint16_t* a = in + (size_t)tok * 2 * d;
  const uint16_t* b = a + d;
  const int nv = d >> 3;
  float v[VPT][8];
  float amax = 0.f;
#pragma unroll
  for (int it = 0; it < VPT; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) {
      u16x8 x = *reinterpret_cast<const u16x8*>(a + 8 * i);
      u16x8 y = *reinterpret_cast<const u16x8*>(b + 8 * i);
      u16x8 r;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        r[j] = silu_mul_bits<T>(T::to_f32(x[j]), T::to_f32(y[j]));
        v[it][j] = T::to_f32(r[j]);
        amax = __builtin_fmaxf(amax, __builtin_fabsf(v[it][j]));
      }
      if (out) *reinterpret_cast<u16x8*>(out + (size_t)tok * d + 8 * i) = r;
    }
  }
  float scale, inv_static = 0.f;
  if (static_scale) {
    scale = *static_scale;
    inv_static = 1.0f / scale;
  } else {
    amax = fq_block_reduce(amax, red, true);
    scale = __builtin_fmaxf(amax / FQ_MAX, 1.0f / (FQ_MAX * 512.f));
  }
  if (threadIdx.x == 0) scale_out[tok] = scale;
#pragma unroll
  for (int it = 0; it < VPT; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) {
      const float lo[4] = {v[it][0], v[it][1], v[it][2], v[it][3]};
      const float hi[4] = {v[it][4], v[it][5], v[it][6], v[it][7]};
      u32x2 q = static_scale ? u32x2{fq_pack4_inv(lo, inv_static), fq_pack4_inv(hi, inv_static)}
                             : u32x2{fq_pack4(lo, scale), fq_pack4(hi, scale)};
      *reinterpret_cast<u32x2*>(q_out + (size_t)tok * d + 8 * i) = q;
    }
  }
}

}  // namespace aphro

using namespace aphro;

// static_scale: NULL = dynamic per-token scales; else the layer's per-tensor input scale ([1], device)
extern "C" int aphro_fused_add_rms_norm_quant_fp8_static(const void* input, const float* slabs, int nslab,
                                                         const float* slab_a_scales, const float* slab_b_scales,
                                                         int a_scale_per_token, int b_scale_per_channel, void* residual,
                                                         int has_residual, const void* weight, float eps, void* q_out,
                                                         float* scale_out, void* out, int64_t tokens, int hidden,
                                                         int dtype, const float* static_scale, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "fused_add_rms_norm_quant: dtype must be f16 or bf16");
  APHRO_CHECK((input != nullptr) != (slabs != nullptr), "fused_add_rms_norm_quant: exactly one of input / slabs");
  APHRO_CHECK(hidden % 8 == 0 && hidden <= 16384, "fused_add_rms_norm_quant: hidden=%d unsupported", hidden);
  APHRO_CHECK(!has_residual || residual != nullptr, "fused_add_rms_norm_quant: residual missing");
  APHRO_CHECK(q_out && scale_out, "fused_add_rms_norm_quant: outputs missing");
  APHRO_CHECK(!(slabs && slab_b_scales && b_scale_per_channel) || ((uintptr_t)slab_b_scales % 16) == 0,
              "fused_add_rms_norm_quant: per-channel scales must be 16-byte aligned");
  if (tokens == 0) return APHRO_OK;
  int nv = hidden / 8, t = nv <= 1024 ? nv : (nv + 1) / 2;   // same mapping as the norm kernels (glue.hip)
  t = (t + 63) / 64 * 64;
  t = t < 64 ? 64 : (t > 1024 ? 1024 : t);
#define L(TT, NSV)                                                                                               \
  hipLaunchKernelGGL((add_rms_norm_quant_kernel<TT, NSV>), dim3((unsigned)tokens), dim3(t), 0, (hipStream_t)stream, \
                     (const uint16_t*)input, slabs, nslab, slab_a_scales, slab_b_scales, a_scale_per_token,      \
                     b_scale_per_channel, (uint16_t*)residual, has_residual, (const uint16_t*)weight, eps,       \
                     (uint8_t*)q_out, scale_out, (uint16_t*)out, (int)tokens, hidden, static_scale)
#define LS(NSV) do { if (dtype == APHRO_F16) L(Half, NSV); else L(BFloat, NSV); } while (0)
  if (slabs != nullptr && nslab == 4) LS(4);
  else if (slabs != nullptr && nslab == 2) LS(2);
  else LS(0);
#undef LS
#undef L
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_fused_add_rms_norm_quant_fp8(const void* input, const float* slabs, int nslab,
                                                  const float* slab_a_scales, const float* slab_b_scales,
                                                  int a_scale_per_token, int b_scale_per_channel, void* residual,
                                                  int has_residual, const void* weight, float eps, void* q_out,
                                                  float* scale_out, void* out, int64_t tokens, int hidden,
                                                  int dtype, void* stream) {
  return aphro_fused_add_rms_norm_quant_fp8_static(input, slabs, nslab, slab_a_scales, slab_b_scales, a_scale_per_token,
                                                   b_scale_per_channel, residual, has_residual, weight, eps, q_out,
                                                   scale_out, out, tokens, hidden, dtype, nullptr, stream);
}

extern "C" int aphro_silu_and_mul_quant_fp8_static(const void* input, void* q_out, float* scale_out, void* out,
                                                   int64_t tokens, int d, int dtype, const float* static_scale,
                                                   void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "silu_and_mul_quant: dtype must be f16 or bf16");
  APHRO_CHECK(d % 8 == 0 && d <= 4 * 1024 * 8, "silu_and_mul_quant: d=%d unsupported", d);
  if (tokens == 0) return APHRO_OK;
  const int nv = d / 8;
  const int vpt = (nv + 1023) / 1024;                 // vectors per thread at 1024 threads
  int threads = vpt == 1 ? (nv + 63) / 64 * 64 : 1024;
  dim3 grid((unsigned)tokens), block(threads);
#define L(TT, V)                                                                                              \
  hipLaunchKernelGGL((silu_mul_quant_kernel<TT, V>), grid, block, 0, (hipStream_t)stream, (const uint16_t*)input, \
                     (uint8_t*)q_out, scale_out, (uint16_t*)out, d, static_scale)
#define LV(TT) { if (vpt == 1) L(TT, 1); else if (vpt == 2) L(TT, 2); else L(TT, 4); }
  if (dtype == APHRO_F16) LV(Half) else LV(BFloat)
#undef LV
#undef L
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_silu_and_mul_quant_fp8(const void* input, void* q_out, float* scale_out, void* out,
                                            int64_t tokens, int d, int dtype, void* stream) {
  return aphro_silu_and_mul_quant_fp8_static(input, q_out, scale_out, out, tokens, d, dtype, nullptr, stream);
}
