// Fused glue kernels of the decode fast path ("fusing them into
// GEMM prologues/epilogues ... they are ~all remaining HBM traffic per layer").
// On MI355X the small ops between the GEMMs are latency bound (~4.5 us each at
// batch 32: two dependent memory round trips), and the unfused layer spends as
// long in them as in the GEMMs.  These kernels merge
//   [split-K slab reduce] + fused_add_rms_norm + [activation pack]      (K1/K6)
//   [split-K slab reduce] + rotary_embedding + reshape_and_cache        (K3)
//   silu_and_mul + [activation pack]                                    (K8)
// while reproducing the unfused op sequence bit for bit (every intermediate is
// rounded to the activation dtype exactly where the separate ops would).
// Reference semantics: kernels/layernorm_kernels.cu:200-240,
// kernels/pos_encoding_kernels.cu:10-160, kernels/cache_kernels.cu:152-204,
// kernels/activation_kernels.cu:12-75.
#include "common.h"

namespace aphro {

__device__ __forceinline__ float block_sum_f(float v, float* red) {
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

// fragment-major offset (in halfs) of the 8-element chunk holding A[row][k..k+7], k % 8 == 0
__device__ __forceinline__ size_t packed_chunk(int row, int k, int mtiles) {
  const int seg = k >> 7, g = (k & 127) >> 5, u = (k & 31) >> 3;
  const int mt = row >> 4;
  return ((((size_t)seg * 4 + u) * mtiles + mt) * 64 + g * 16 + (row & 15)) * 8;
}

template <typename T>
__device__ __forceinline__ uint16_t to_f16_bits(uint16_t tbits) {
  if constexpr (__is_same(T, Half)) return tbits;
  else return bf16_bits_to_f16_bits_sat(tbits);
}

// x = input (T) or sum of `nslab` fp32 slabs (rounded to T like the GEMM's own
// reduce would); residual' = round(x + residual) (or x when residual_in is null);
// y = round(round(residual' * rstd) * w).  y goes to the packed buffer (f16) and/or
// row-major `out`.
// ROUTER (sparse-MLP layers): the router's logits of the token, round_T(y . Wg[e]) for e < num_experts <= 16 (the
// replicated `gate` linear of MixtralMoE, modeling/models/mixtral.py:60-110, a [M, E] library GEMM launch of its own
// otherwise), from the normalised values the thread already holds.  A separate instantiation: the dense model's norm kernel
// is not touched.
// COMBINE (the norm that follows a sparse MLP): x = moe_combine of the expert GEMM's fp32 slabs, sum_k round_T(w[t, k] *
// sum_s slab[s][inv_pos[t k + kk]]) rounded to T -- moe_combine_kernel's arithmetic (moe.hip), without its launch and its
// [T, hidden] round trip.
// NS > 0 (dense decode layers, round 4): x = the sum of exactly NS slabs, every load of a thread -- the NS slab pieces, the
// residual, the norm weights -- issued BEFORE the first wait.  With the runtime `nslab` loop hipcc emits load, wait, add
// per slab and the residual load behind them: five dependent L2 / Infinity-Cache round trips in a 4.9 us launch.  Same
// additions in the same order: bit-identical.
template <typename T, bool ROUTER = false, bool COMBINE = false, int NS = 0>
__global__ void add_rms_norm_pack_kernel(const uint16_t* __restrict__ input, const float* __restrict__ slabs,
                                         int nslab, uint16_t* __restrict__ residual, int has_residual,
                                         const uint16_t* __restrict__ weight, float eps,
                                         uint16_t* __restrict__ packed, uint16_t* __restrict__ out, int tokens,
                                         int hidden, const uint16_t* __restrict__ router_w = nullptr,
                                         uint16_t* __restrict__ router_out = nullptr, int num_experts = 0,
                                         const int32_t* __restrict__ inv_pos = nullptr,
                                         const float* __restrict__ topk_w = nullptr, int topk = 0,
                                         int64_t comb_stride = 0) {
  __shared__ float red[16];
  __shared__ float rred[ROUTER ? 16 * 16 : 1];
  float rpart[ROUTER ? 16 : 1];
  if constexpr (ROUTER) {
#pragma unroll
    for (int e = 0; e < 16; ++e) rpart[e] = 0.f;
  }
  const int tok = blockIdx.x;
  const int nv = hidden >> 3;
  const int mtiles = (tokens + 15) >> 4;
  const size_t slab_stride = (size_t)tokens * hidden;
  float v[2][8];
  u16x8 wv[2];  // norm weights: fetched with the inputs, not after the reduction barrier
  u16x8 g8v[ROUTER ? 2 : 1][ROUTER ? 8 : 1];   // ... and so are the first 8 router rows (a serial L2 round trip otherwise)
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) {
      const size_t off = (size_t)tok * hidden + 8 * i;
      wv[it] = *reinterpret_cast<const u16x8*>(weight + 8 * i);
      if constexpr (ROUTER) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < num_experts) g8v[it][e] = *reinterpret_cast<const u16x8*>(router_w + (size_t)e * hidden + 8 * i);
      }
      float x[8];
      if constexpr (COMBINE) {
        float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int kk = 0; kk < topk; ++kk) {
          const int pos = inv_pos[tok * topk + kk];
          const float w = topk_w[tok * topk + kk];
          const float* p0 = slabs + (size_t)pos * hidden + 8 * i;
          f32x4 a = *reinterpret_cast<const f32x4*>(p0);
          f32x4 b = *reinterpret_cast<const f32x4*>(p0 + 4);
          for (int s2 = 1; s2 < nslab; ++s2) {
            a += *reinterpret_cast<const f32x4*>(p0 + s2 * comb_stride);
            b += *reinterpret_cast<const f32x4*>(p0 + s2 * comb_stride + 4);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc8[j] += T::to_f32(from_f32_exact<T>(a[j] * w));
            acc8[4 + j] += T::to_f32(from_f32_exact<T>(b[j] * w));
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = T::to_f32(T::from_f32(acc8[j]));
      } else if constexpr (NS > 0) {
        f32x4 sa[NS], sb[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          sa[s] = *reinterpret_cast<const f32x4*>(slabs + s * slab_stride + off);
          sb[s] = *reinterpret_cast<const f32x4*>(slabs + s * slab_stride + off + 4);
        }
        // (the residual rides along: a pointer select, not a branch -- when there is none the weights are re-read, unused)
        u16x8 r_early = *reinterpret_cast<const u16x8*>(has_residual ? residual + off : weight + 8 * i);
        asm volatile("" : "+v"(r_early));    // (used HERE as far as hipcc can tell: it would sink the load into the branch below)
        f32x4 a = sa[0], b = sb[0];
#pragma unroll
        for (int s = 1; s < NS; ++s) { a += sa[s]; b += sb[s]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          x[j] = T::to_f32(T::from_f32(a[j]));
          x[4 + j] = T::to_f32(T::from_f32(b[j]));
        }
        u16x8 rs2;
        if (has_residual) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            rs2[j] = T::from_f32(x[j] + T::to_f32(r_early[j]));
            v[it][j] = T::to_f32(rs2[j]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            rs2[j] = T::from_f32(x[j]);
            v[it][j] = x[j];
          }
        }
        if (residual) *reinterpret_cast<u16x8*>(residual + off) = rs2;
        {
          // squares rounded, then added -- pinned: the norm-in-consumer GEMM launch (wna16_gemm_resident.hip, res_norm_finish)
          // reproduces this row bit for bit and must not depend on which of the two loops hipcc contracts into FMAs
#pragma clang fp contract(off)
#pragma unroll
          for (int j = 0; j < 8; ++j) ss += v[it][j] * v[it][j];
        }
        continue;
      } else if (slabs) {
        f32x4 a = *reinterpret_cast<const f32x4*>(slabs + off);
        f32x4 b = *reinterpret_cast<const f32x4*>(slabs + off + 4);
        for (int s = 1; s < nslab; ++s) {
          a += *reinterpret_cast<const f32x4*>(slabs + s * slab_stride + off);
          b += *reinterpret_cast<const f32x4*>(slabs + s * slab_stride + off + 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          x[j] = T::to_f32(T::from_f32(a[j]));
          x[4 + j] = T::to_f32(T::from_f32(b[j]));
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
          rs[j] = T::from_f32(x[j] + T::to_f32(r[j]));
          v[it][j] = T::to_f32(rs[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          rs[j] = T::from_f32(x[j]);
          v[it][j] = x[j];
        }
      }
      if (residual) *reinterpret_cast<u16x8*>(residual + off) = rs;
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v[it][j] * v[it][j];
    }
  }
  ss = block_sum_f(ss, red);
  const float inv = __frsqrt_rn(ss / (float)hidden + eps);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) {
      const u16x8 w = wv[it];
      u16x8 y, yh;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // (from_f32_exact: product rounded to fp32, THEN converted -- (scalar_t)(x * s_variance), layernorm_kernels.cu:228;
        //  pinned so that the norm-in-consumer GEMM launch can promise the same bits)
        y[j] = T::from_f32(T::to_f32(from_f32_exact<T>(v[it][j] * inv)) * T::to_f32(w[j]));
        yh[j] = to_f16_bits<T>(y[j]);
      }
      if (out) *reinterpret_cast<u16x8*>(out + (size_t)tok * hidden + 8 * i) = y;
      if (packed) *reinterpret_cast<u16x8*>(packed + packed_chunk(tok, 8 * i, mtiles)) = yh;
      if constexpr (ROUTER) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (e < num_experts) {
            u16x8 g8;
            if (e < 8) g8 = g8v[it][e < 8 ? e : 0];
            else g8 = *reinterpret_cast<const u16x8*>(router_w + (size_t)e * hidden + 8 * i);
            {
              // products rounded, then added -- pinned (what hipcc emits for the f16 instantiation: v_pk_mul_f32 + adds): the
              // all-reduce + norm + router launch (custom_all_reduce.hip, ROUTER epilogue) reproduces these logits bit for bit
#pragma clang fp contract(off)
#pragma unroll
              for (int j = 0; j < 8; ++j) rpart[e] += T::to_f32(y[j]) * T::to_f32(g8[j]);
            }
          }
      }
    }
  }
  if constexpr (ROUTER) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = (blockDim.x + 63) >> 6;
    // 16 wave sums in 17 shuffles instead of 96: a reduce-scatter butterfly -- at offset 32 a lane keeps one half of the
    // experts and hands the other half to its partner, at 16 a quarter, ... after offset 4 it holds ONE expert's partial
    // (expert = lane bits 5..2), two plain steps finish it
#define APHRO_RS_STEP(O, HALF)                                        \
    {                                                                 \
      const bool up = (lane & O) != 0;                                \
      _Pragma("unroll") for (int j = 0; j < HALF; ++j) {              \
        const float send = up ? rpart[j] : rpart[j + HALF];           \
        const float keep = up ? rpart[j + HALF] : rpart[j];           \
        rpart[j] = keep + __shfl_xor(send, O, 64);                    \
      }                                                               \
    }
    APHRO_RS_STEP(32, 8)
    APHRO_RS_STEP(16, 4)
    APHRO_RS_STEP(8, 2)
    APHRO_RS_STEP(4, 1)
#undef APHRO_RS_STEP
    float v2 = rpart[0];
    v2 += __shfl_xor(v2, 2, 64);
    v2 += __shfl_xor(v2, 1, 64);
    if ((lane & 3) == 0) rred[wave * 16 + ((lane >> 2) & 15)] = v2;
    __syncthreads();
    if ((int)threadIdx.x < num_experts) {
      float sum = 0.f;
      for (int w2 = 0; w2 < nwave; ++w2) sum += rred[w2 * 16 + threadIdx.x];
      router_out[(size_t)tok * num_experts + threadIdx.x] = T::from_f32(sum);
    }
  }
}

// act = round(round(silu(gate)) * up) -> packed f16 (and/or row-major).
// SLABS: [gate | up] = the sum of `nslab` fp32 split-K slabs [nslab][tokens][2 d] rounded to T -- splitk_reduce_kernel's
// arithmetic (wna16_gemm.hip: slab order, one rounding) without its launch and its [tokens, 2 d] round trip (round 4: the
// K-sliced gate_up of a TP shard, e.g. 8192 x 7168 at 64 rows, used to be GEMM + splitk_reduce + silu_and_mul_pack).
template <typename T, bool SLABS = false>
__global__ void silu_mul_pack_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ packed,
                                     uint16_t* __restrict__ out, int tokens, int d,
                                     const float* __restrict__ slabs = nullptr, int nslab = 0) {
  const int tok = blockIdx.x;
  const int mtiles = (tokens + 15) >> 4;
  const uint16_t* a = in + (size_t)tok * 2 * d;
  const uint16_t* b = a + d;
  const size_t slab_stride = (size_t)tokens * 2 * d;
  for (int i = threadIdx.x; i < (d >> 3); i += blockDim.x) {
    u16x8 x, y;
    if constexpr (SLABS) {
      const float* pg = slabs + (size_t)tok * 2 * d + 8 * i;
      f32x4 g0 = *reinterpret_cast<const f32x4*>(pg), g1 = *reinterpret_cast<const f32x4*>(pg + 4);
      f32x4 u0 = *reinterpret_cast<const f32x4*>(pg + d), u1 = *reinterpret_cast<const f32x4*>(pg + d + 4);
      for (int s = 1; s < nslab; ++s) {
        g0 += *reinterpret_cast<const f32x4*>(pg + s * slab_stride);
        g1 += *reinterpret_cast<const f32x4*>(pg + s * slab_stride + 4);
        u0 += *reinterpret_cast<const f32x4*>(pg + s * slab_stride + d);
        u1 += *reinterpret_cast<const f32x4*>(pg + s * slab_stride + d + 4);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[j] = T::from_f32(g0[j]); x[4 + j] = T::from_f32(g1[j]);
        y[j] = T::from_f32(u0[j]); y[4 + j] = T::from_f32(u1[j]);
      }
    } else {
      x = *reinterpret_cast<const u16x8*>(a + 8 * i);
      y = *reinterpret_cast<const u16x8*>(b + 8 * i);
    }
    u16x8 r, rh;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float xf = T::to_f32(x[j]);
      float s = T::to_f32(T::from_f32(xf / (1.0f + __expf(-xf))));
      r[j] = T::from_f32(s * T::to_f32(y[j]));
      rh[j] = to_f16_bits<T>(r[j]);
    }
    if (out) *reinterpret_cast<u16x8*>(out + (size_t)tok * d + 8 * i) = r;
    if (packed) *reinterpret_cast<u16x8*>(packed + packed_chunk(tok, 8 * i, mtiles)) = rh;
  }
}

// qkv row (T, or fp32 slabs) -> RoPE(q) to q_out, RoPE(k) and v into the paged cache.
template <typename T, int KV, bool NEOX>
__global__ void rope_cache_kernel(const uint16_t* __restrict__ qkv, int64_t qkv_stride,
                                  const float* __restrict__ slabs, int nslab, int tokens,
                                  const int64_t* __restrict__ positions, const uint16_t* __restrict__ cos_sin,
                                  int rot_dim, uint16_t* __restrict__ q_out, void* __restrict__ key_cache,
                                  void* __restrict__ value_cache, const int64_t* __restrict__ slot_mapping,
                                  int num_heads, int num_kv_heads, int head_size, int block_size, int x,
                                  float k_scale, float v_scale) {
  const int tok = blockIdx.x;
  const int nq = num_heads * head_size, nkv = num_kv_heads * head_size;
  const int ntot = nq + 2 * nkv;
  const size_t slab_stride = (size_t)tokens * ntot;
  auto val = [&](int j) -> float {  // element j of the token's qkv row, rounded to T
    if (slabs) {
      float s = slabs[(size_t)tok * ntot + j];
      for (int k = 1; k < nslab; ++k) s += slabs[k * slab_stride + (size_t)tok * ntot + j];
      return T::to_f32(T::from_f32(s));
    }
    return T::to_f32(qkv[(size_t)tok * qkv_stride + j]);
  };
  const int64_t pos = positions ? positions[tok] : (int64_t)tok;  // NULL: cos_sin rows pre-gathered per token
  const uint16_t* cs = cos_sin + pos * rot_dim;
  const int embed = rot_dim >> 1;
  const int64_t slot = slot_mapping[tok];
  const int64_t blk = slot >= 0 ? slot / block_size : 0, off = slot >= 0 ? slot % block_size : 0;
  auto kstore = [&](int h, int d, float f) {
    if (slot < 0) return;
    const int64_t dst = (((blk * num_kv_heads + h) * (head_size / x) + d / x) * block_size + off) * x + d % x;
    if constexpr (KV == 0) ((uint16_t*)key_cache)[dst] = T::from_f32(f);
    else ((uint8_t*)key_cache)[dst] = (uint8_t)f32x2_to_fp8<KV == 2>(T::to_f32(T::from_f32(f)) / k_scale, 0.f);
  };
  // rotary pairs of q and k heads
  const int npairs = (num_heads + num_kv_heads) * embed;
  for (int i = threadIdx.x; i < npairs; i += blockDim.x) {
    const int h = i / embed, r = i % embed;
    const bool is_k = h >= num_heads;
    const int base = is_k ? nq + (h - num_heads) * head_size : h * head_size;
    const int xi = NEOX ? r : 2 * r, yi = NEOX ? embed + r : 2 * r + 1;
    const float c = T::to_f32(cs[r]), s = T::to_f32(cs[embed + r]);
    const float xv = val(base + xi), yv = val(base + yi);
    float xo, yo;
    rope_pair(xv, yv, c, s, xo, yo);
    if (!is_k) {
      q_out[(size_t)tok * nq + base + xi] = T::from_f32(xo);
      q_out[(size_t)tok * nq + base + yi] = T::from_f32(yo);
    } else {
      kstore(h - num_heads, xi, xo);
      kstore(h - num_heads, yi, yo);
    }
  }
  // pass-through dims beyond rot_dim
  if (rot_dim < head_size) {
    const int rest = head_size - rot_dim;
    for (int i = threadIdx.x; i < (num_heads + num_kv_heads) * rest; i += blockDim.x) {
      const int h = i / rest, d = rot_dim + i % rest;
      if (h < num_heads) q_out[(size_t)tok * nq + h * head_size + d] = T::from_f32(val(h * head_size + d));
      else kstore(h - num_heads, d, val(nq + (h - num_heads) * head_size + d));
    }
  }
  // v
  if (slot >= 0) {
    for (int i = threadIdx.x; i < nkv; i += blockDim.x) {
      const int h = i / head_size, d = i % head_size;
      const float f = val(nq + nkv + i);
      const int64_t dst = ((blk * num_kv_heads + h) * head_size + d) * block_size + off;
      if constexpr (KV == 0) ((uint16_t*)value_cache)[dst] = T::from_f32(f);
      else ((uint8_t*)value_cache)[dst] = (uint8_t)f32x2_to_fp8<KV == 2>(f / v_scale, 0.f);
    }
  }
}

// Vectorised form of rope_cache_kernel for the common case (NeoX rotary over the
// whole head, head_size % 16 == 0): one thread = 8 consecutive dims of the first
// half of a head and their 8 partners in the second half; 16-byte loads/stores.
template <typename T, int KV>
__global__ void rope_cache_vec_kernel(const uint16_t* __restrict__ qkv, int64_t qkv_stride,
                                      const float* __restrict__ slabs, int nslab, int tokens,
                                      const int64_t* __restrict__ positions, const uint16_t* __restrict__ cos_sin,
                                      uint16_t* __restrict__ q_out, void* __restrict__ key_cache,
                                      void* __restrict__ value_cache, const int64_t* __restrict__ slot_mapping,
                                      int num_heads, int num_kv_heads, int head_size, int block_size, int x,
                                      float k_scale, float v_scale) {
  const int tok = blockIdx.x;
  const int nq = num_heads * head_size, nkv = num_kv_heads * head_size;
  const int ntot = nq + 2 * nkv;
  const size_t slab_stride = (size_t)tokens * ntot;
  auto val8 = [&](int j, float (&o)[8]) {  // 8 consecutive elements from j (j % 8 == 0), rounded to T
    if (slabs) {
      const float* p0 = slabs + (size_t)tok * ntot + j;
      f32x4 a = *reinterpret_cast<const f32x4*>(p0), b = *reinterpret_cast<const f32x4*>(p0 + 4);
      for (int k = 1; k < nslab; ++k) {
        a += *reinterpret_cast<const f32x4*>(p0 + k * slab_stride);
        b += *reinterpret_cast<const f32x4*>(p0 + k * slab_stride + 4);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[i] = T::to_f32(T::from_f32(a[i]));
        o[4 + i] = T::to_f32(T::from_f32(b[i]));
      }
    } else {
      u16x8 v = *reinterpret_cast<const u16x8*>(qkv + (size_t)tok * qkv_stride + j);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = T::to_f32(v[i]);
    }
  };
  const int64_t pos = positions ? positions[tok] : (int64_t)tok;  // NULL: cos_sin rows pre-gathered per token
  const int half = head_size >> 1;
  const uint16_t* cs = cos_sin + pos * head_size;
  const int64_t slot = slot_mapping[tok];
  const int64_t blk = slot >= 0 ? slot / block_size : 0, off = slot >= 0 ? slot % block_size : 0;
  const int cph = half >> 3;  // chunks per head (first half)
  const int nrope = (num_heads + num_kv_heads) * cph;
  const int nv = nkv >> 3;
  for (int i = threadIdx.x; i < nrope + nv; i += blockDim.x) {
    if (i < nrope) {
      const int h = i / cph, d0 = (i % cph) * 8;
      const bool is_k = h >= num_heads;
      const int base = is_k ? nq + (h - num_heads) * head_size : h * head_size;
      float xv[8], yv[8];
      val8(base + d0, xv);
      val8(base + half + d0, yv);
      u16x8 c8 = *reinterpret_cast<const u16x8*>(cs + d0);
      u16x8 s8 = *reinterpret_cast<const u16x8*>(cs + half + d0);
      u16x8 xo8, yo8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float xo, yo;
        rope_pair(xv[j], yv[j], T::to_f32(c8[j]), T::to_f32(s8[j]), xo, yo);
        xo8[j] = T::from_f32(xo);
        yo8[j] = T::from_f32(yo);
      }
      if (!is_k) {
        *reinterpret_cast<u16x8*>(q_out + (size_t)tok * nq + base + d0) = xo8;
        *reinterpret_cast<u16x8*>(q_out + (size_t)tok * nq + base + half + d0) = yo8;
      } else if (slot >= 0) {
        const int hk = h - num_heads;
#pragma unroll
        for (int part = 0; part < 2; ++part) {
          const int d = part ? half + d0 : d0;
          const u16x8& o8 = part ? yo8 : xo8;
          const int64_t dst = (((blk * num_kv_heads + hk) * (head_size / x) + d / x) * block_size + off) * x + d % x;
          if constexpr (KV == 0) {
            *reinterpret_cast<u16x8*>((uint16_t*)key_cache + dst) = o8;
          } else {
            uint32_t w0 = 0, w1 = 0;
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
              w0 |= f32x2_to_fp8<KV == 2>(T::to_f32(o8[j]) / k_scale, T::to_f32(o8[j + 1]) / k_scale) << (8 * j);
              w1 |= f32x2_to_fp8<KV == 2>(T::to_f32(o8[4 + j]) / k_scale, T::to_f32(o8[5 + j]) / k_scale) << (8 * j);
            }
            *reinterpret_cast<u32x2*>((uint8_t*)key_cache + dst) = u32x2{w0, w1};
          }
        }
      }
    } else if (slot >= 0) {
      const int e = (i - nrope) * 8;
      const int h = e / head_size, d = e % head_size;
      float v[8];
      val8(nq + nkv + e, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t dst = ((blk * num_kv_heads + h) * head_size + d + j) * block_size + off;
        if constexpr (KV == 0) ((uint16_t*)value_cache)[dst] = T::from_f32(v[j]);
        else ((uint8_t*)value_cache)[dst] = (uint8_t)f32x2_to_fp8<KV == 2>(v[j] / v_scale, 0.f);
      }
    }
  }
}

}  // namespace aphro

using namespace aphro;

extern "C" int aphro_fused_add_rms_norm_pack(const void* input, const float* slabs, int nslab, void* residual,
                                             int has_residual, const void* weight, float eps, void* packed,
                                             void* out, int64_t tokens, int hidden, int dtype, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "fused_add_rms_norm_pack: dtype must be f16 or bf16");
  APHRO_CHECK((input != nullptr) != (slabs != nullptr), "fused_add_rms_norm_pack: exactly one of input / slabs");
  APHRO_CHECK(hidden % 8 == 0 && hidden <= 16384, "fused_add_rms_norm_pack: hidden=%d unsupported", hidden);
  APHRO_CHECK(packed == nullptr || hidden % 128 == 0, "fused_add_rms_norm_pack: packing needs hidden %% 128 == 0");
  APHRO_CHECK(!has_residual || residual != nullptr, "fused_add_rms_norm_pack: residual missing");
  if (tokens == 0) return APHRO_OK;
  // latency bound (one workgroup per token): one 8-element vector per thread while that fits
  int nv = hidden / 8, t = nv <= 1024 ? nv : (nv + 1) / 2;
  t = (t + 63) / 64 * 64;
  t = t < 64 ? 64 : (t > 1024 ? 1024 : t);
  dim3 grid((unsigned)tokens), block(t);
#define L(TT, NSV)                                                                                                        \
  hipLaunchKernelGGL((add_rms_norm_pack_kernel<TT, false, false, NSV>), grid, block, 0, (hipStream_t)stream, (const uint16_t*)input, \
                     slabs, nslab, (uint16_t*)residual, has_residual, (const uint16_t*)weight, eps,                         \
                     (uint16_t*)packed, (uint16_t*)out, (int)tokens, hidden)
#define LS(NSV) do { if (dtype == APHRO_F16) L(Half, NSV); else L(BFloat, NSV); } while (0)
  if (slabs != nullptr && nslab == 4 && !APHRO_LAB_ENV_INT("APHRO_NORM_GENERIC", 0)) LS(4);
  else if (slabs != nullptr && nslab == 2 && !APHRO_LAB_ENV_INT("APHRO_NORM_GENERIC", 0)) LS(2);
  else if (slabs != nullptr && nslab == 8 && !APHRO_LAB_ENV_INT("APHRO_NORM_GENERIC", 0)) LS(8);
  else LS(0);
#undef LS
#undef L
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

// The same kernel for a sparse-MLP layer: also writes router_out[tokens, num_experts] = round_T(y . router_w[e]) (the
// replicated gate linear of MixtralMoE; num_experts <= 16).  `out` (the row-major normalised activations the expert
// gather reads) is required.
extern "C" int aphro_fused_add_rms_norm_router(const void* input, const float* slabs, int nslab, void* residual,
                                               int has_residual, const void* weight, float eps, void* out,
                                               const void* router_w, void* router_out, int num_experts, int64_t tokens,
                                               int hidden, int dtype, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "fused_add_rms_norm_router: dtype must be f16 or bf16");
  APHRO_CHECK((input != nullptr) != (slabs != nullptr), "fused_add_rms_norm_router: exactly one of input / slabs");
  APHRO_CHECK(hidden % 8 == 0 && hidden <= 16384 && out != nullptr, "fused_add_rms_norm_router: hidden=%d unsupported", hidden);
  APHRO_CHECK(num_experts >= 1 && num_experts <= 16 && router_w != nullptr && router_out != nullptr,
              "fused_add_rms_norm_router: 1..16 experts (got %d)", num_experts);
  APHRO_CHECK(!has_residual || residual != nullptr, "fused_add_rms_norm_router: residual missing");
  if (tokens == 0) return APHRO_OK;
  int nv = hidden / 8, t = nv <= 1024 ? nv : (nv + 1) / 2;
  t = (t + 63) / 64 * 64;
  t = t < 64 ? 64 : (t > 1024 ? 1024 : t);
  dim3 grid((unsigned)tokens), block(t);
#define L(TT)                                                                                                      \
  hipLaunchKernelGGL((add_rms_norm_pack_kernel<TT, true>), grid, block, 0, (hipStream_t)stream, (const uint16_t*)input, \
                     slabs, nslab, (uint16_t*)residual, has_residual, (const uint16_t*)weight, eps, (uint16_t*)nullptr, \
                     (uint16_t*)out, (int)tokens, hidden, (const uint16_t*)router_w, (uint16_t*)router_out, num_experts)
  if (dtype == APHRO_F16) L(Half); else L(BFloat);
#undef L
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

// The norm after a sparse MLP: x = moe_combine(slabs [nslab][m_pad][hidden], inv_pos [tokens * topk], topk_weights) folded
// into the kernel's input stage (same bits as aphro_moe_combine followed by aphro_fused_add_rms_norm_pack).
extern "C" int aphro_fused_add_rms_norm_pack_combine(const float* slabs, int nslab, int64_t m_pad, const int32_t* inv_pos,
                                                     const float* topk_weights, int topk, void* residual, int has_residual,
                                                     const void* weight, float eps, void* packed, void* out, int64_t tokens,
                                                     int hidden, int dtype, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "fused_add_rms_norm_pack_combine: dtype must be f16 or bf16");
  APHRO_CHECK(slabs != nullptr && inv_pos != nullptr && topk_weights != nullptr && nslab >= 1 && topk >= 1,
              "fused_add_rms_norm_pack_combine: slabs / inv_pos / topk_weights required");
  APHRO_CHECK(hidden % 8 == 0 && hidden <= 16384, "fused_add_rms_norm_pack_combine: hidden=%d unsupported", hidden);
  APHRO_CHECK(packed == nullptr || hidden % 128 == 0, "fused_add_rms_norm_pack_combine: packing needs hidden %% 128 == 0");
  APHRO_CHECK(!has_residual || residual != nullptr, "fused_add_rms_norm_pack_combine: residual missing");
  if (tokens == 0) return APHRO_OK;
  int nv = hidden / 8, t = nv <= 1024 ? nv : (nv + 1) / 2;
  t = (t + 63) / 64 * 64;
  t = t < 64 ? 64 : (t > 1024 ? 1024 : t);
  dim3 grid((unsigned)tokens), block(t);
#define L(TT)                                                                                                      \
  hipLaunchKernelGGL((add_rms_norm_pack_kernel<TT, false, true>), grid, block, 0, (hipStream_t)stream,             \
                     (const uint16_t*)nullptr, slabs, nslab, (uint16_t*)residual, has_residual, (const uint16_t*)weight, \
                     eps, (uint16_t*)packed, (uint16_t*)out, (int)tokens, hidden, (const uint16_t*)nullptr,         \
                     (uint16_t*)nullptr, 0, inv_pos, topk_weights, topk, (int64_t)m_pad * hidden)
  if (dtype == APHRO_F16) L(Half); else L(BFloat);
#undef L
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_silu_and_mul_pack(const void* input, void* packed, void* out, int64_t tokens, int d,
                                       int dtype, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "silu_and_mul_pack: dtype must be f16 or bf16");
  APHRO_CHECK(d % 8 == 0 && (packed == nullptr || d % 128 == 0), "silu_and_mul_pack: d=%d unsupported", d);
  if (tokens == 0) return APHRO_OK;
  int threads = d / 8 >= 1024 ? 1024 : ((d / 8 + 63) / 64 * 64);
  dim3 grid((unsigned)tokens), block(threads);
  if (dtype == APHRO_F16)
    hipLaunchKernelGGL((silu_mul_pack_kernel<Half>), grid, block, 0, (hipStream_t)stream, (const uint16_t*)input,
                       (uint16_t*)packed, (uint16_t*)out, (int)tokens, d);
  else
    hipLaunchKernelGGL((silu_mul_pack_kernel<BFloat>), grid, block, 0, (hipStream_t)stream, (const uint16_t*)input,
                       (uint16_t*)packed, (uint16_t*)out, (int)tokens, d);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

// silu_and_mul_pack on the split-K slabs of the gate_up GEMM (the slab reduce rides in this launch).
extern "C" int aphro_silu_and_mul_pack_slabs(const float* slabs, int nslab, void* packed, void* out, int64_t tokens, int d,
                                             int dtype, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "silu_and_mul_pack_slabs: dtype must be f16 or bf16");
  APHRO_CHECK(d % 8 == 0 && (packed == nullptr || d % 128 == 0), "silu_and_mul_pack_slabs: d=%d unsupported", d);
  APHRO_CHECK(slabs != nullptr && nslab >= 1 && ((uintptr_t)slabs % 16) == 0, "silu_and_mul_pack_slabs: slabs missing / misaligned");
  if (tokens == 0) return APHRO_OK;
  int threads = d / 8 >= 1024 ? 1024 : ((d / 8 + 63) / 64 * 64);
  dim3 grid((unsigned)tokens), block(threads);
  if (dtype == APHRO_F16)
    hipLaunchKernelGGL((silu_mul_pack_kernel<Half, true>), grid, block, 0, (hipStream_t)stream, (const uint16_t*)nullptr,
                       (uint16_t*)packed, (uint16_t*)out, (int)tokens, d, slabs, nslab);
  else
    hipLaunchKernelGGL((silu_mul_pack_kernel<BFloat, true>), grid, block, 0, (hipStream_t)stream, (const uint16_t*)nullptr,
                       (uint16_t*)packed, (uint16_t*)out, (int)tokens, d, slabs, nslab);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_rope_cache(const void* qkv, int64_t qkv_stride, const float* slabs, int nslab,
                                const int64_t* positions, const void* cos_sin_cache, int rot_dim, int is_neox,
                                void* q_out, void* key_cache, void* value_cache, const int64_t* slot_mapping,
                                int64_t tokens, int num_heads, int num_kv_heads, int head_size, int block_size,
                                int x, int dtype, int kv_dtype, float k_scale, float v_scale, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "rope_cache: dtype must be f16 or bf16");
  APHRO_CHECK(kv_dtype >= APHRO_KV_AUTO && kv_dtype <= APHRO_KV_FP8_E5M2, "Unsupported data type of kv cache: %d", kv_dtype);
  APHRO_CHECK((qkv != nullptr) != (slabs != nullptr), "rope_cache: exactly one of qkv / slabs");
  APHRO_CHECK(rot_dim % 2 == 0 && rot_dim <= head_size && head_size % x == 0, "rope_cache: bad rot_dim / x");
  if (tokens == 0) return APHRO_OK;
  dim3 grid((unsigned)tokens), block(512);
  if (is_neox && rot_dim == head_size && head_size % 16 == 0 && (x == 8 || x == 16) &&
      (qkv == nullptr || (qkv_stride % 8 == 0 && ((uintptr_t)qkv % 16) == 0))) {
    const int work = (num_heads + num_kv_heads) * (head_size / 16) + num_kv_heads * head_size / 8;
    dim3 vblock((unsigned)(work >= 512 ? 512 : (work + 63) / 64 * 64));
#define LV(TT, KVV)                                                                                             \
  hipLaunchKernelGGL((rope_cache_vec_kernel<TT, KVV>), grid, vblock, 0, (hipStream_t)stream, (const uint16_t*)qkv, \
                     qkv_stride, slabs, nslab, (int)tokens, positions, (const uint16_t*)cos_sin_cache,           \
                     (uint16_t*)q_out, key_cache, value_cache, slot_mapping, num_heads, num_kv_heads, head_size, \
                     block_size, x, k_scale, v_scale)
#define LVK(TT) { if (kv_dtype == APHRO_KV_AUTO) LV(TT, 0); else if (kv_dtype == APHRO_KV_FP8_E4M3) LV(TT, 1); else LV(TT, 2); }
    if (dtype == APHRO_F16) LVK(Half) else LVK(BFloat)
#undef LVK
#undef LV
    APHRO_LAUNCH_CHECK();
    return APHRO_OK;
  }
#define L(TT, KVV, NX)                                                                                          \
  hipLaunchKernelGGL((rope_cache_kernel<TT, KVV, NX>), grid, block, 0, (hipStream_t)stream, (const uint16_t*)qkv, \
                     qkv_stride, slabs, nslab, (int)tokens, positions, (const uint16_t*)cos_sin_cache, rot_dim,  \
                     (uint16_t*)q_out, key_cache, value_cache, slot_mapping, num_heads, num_kv_heads, head_size, \
                     block_size, x, k_scale, v_scale)
#define LN(TT, KVV) { if (is_neox) L(TT, KVV, true); else L(TT, KVV, false); }
#define LK(TT) { if (kv_dtype == APHRO_KV_AUTO) LN(TT, 0) else if (kv_dtype == APHRO_KV_FP8_E4M3) LN(TT, 1) else LN(TT, 2) }
  if (dtype == APHRO_F16) LK(Half) else LK(BFloat)
#undef LK
#undef LN
#undef L
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
