// FP8 (OCP e4m3fn) weight GEMMs for small M on gfx950.
//
//  * W8A8  (cutlass_scaled_mm role, kernels/quantization/cutlass_w8a8/
//    scaled_mm_entry.cu:92-137; on ROCm today torch._scaled_mm,
//    quantization/utils/w8a8_utils.py:83-183):
//       out = a_scales (.) (A_q . B_q) (.) b_scales + bias,
//    v_mfma_f32_16x16x32_fp8_fp8, fp32 accumulate, per-tensor or per-token /
//    per-channel scales fused in the epilogue (the reference's unfused ROCm path
//    materialises an fp32 [M,N] and multiplies twice).
//  * W8A16 (fp8_marlin_gemm role, quantization/fp8/fp8_marlin.cu:1212):
//       out = A . (fp8->hp(W) * s_n): W is widened to the activation dtype in
//    registers (exact) and contracted with the f16/bf16 MFMA.
//
// W is [N,K] row-major (K contiguous) -- the checkpoint layout, i.e. the
// column-major [K,N] the op schema asks for.  HBM-bound: a lane streams 32
// contiguous bytes of one weight row per 128-k macro step (4 lanes cover one
// 128-B line), every byte read once.  Work split and LDS reduction as in
// wna16_gemm.hip: 8 waves split K of one column tile, optional second-level
// split-K through an fp32 workspace.
#include "common.h"

namespace aphro {

constexpr int FNW = 8;

struct Fp8GemmParams {
  const void* a;        // A8: e4m3 [M,K] ; else T [M,lda]
  const uint8_t* w;     // e4m3 [N,K]
  const float* a_scales;
  const float* b_scales;
  const void* bias;     // T [N] or null
  void* c;              // T [M,N]
  float* partial;       // [ksplit,M,N]
  int M, N, K, lda;
  int msteps_per_split;  // macro steps (128 k) per blockIdx.y
  int ksplit;
  int a_per_token, b_per_channel;
  int force_partial;     // 1: leave the raw fp32 accumulators in `partial` even when ksplit == 1 (fused consumer)
};

template <typename T>
__device__ __forceinline__ u32x4 fp8x8_to_T(uint32_t w0, uint32_t w1) {
  u32x4 r;
  r[0] = fp8x2_to_T<T, false, false>(w0);
  r[1] = fp8x2_to_T<T, false, true>(w0);
  r[2] = fp8x2_to_T<T, false, false>(w1);
  r[3] = fp8x2_to_T<T, false, true>(w1);
  return r;
}

template <typename T>
__device__ __forceinline__ f32x4 mfma_hp(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (__is_same(T, Half))
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// A8: activations are fp8 (W8A8) ; NT: n-tiles per wave ; MT: m-tiles
template <typename T, bool A8, int NT, int MT>
__global__ __launch_bounds__(FNW * 64) void fp8_gemm_kernel(Fp8GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4;
  const int c = lane & 15;
  const int n0 = blockIdx.x * (16 * NT);
  const int m0 = blockIdx.z * (16 * MT);
  const int ncol = n0 + NT * c;

  const int total = p.K >> 7;
  const int wg_begin = blockIdx.y * p.msteps_per_split;
  const int wg_end = min(total, wg_begin + p.msteps_per_split);
  const int per_wave = (wg_end - wg_begin + FNW - 1) / FNW;
  const int s0 = wg_begin + wave * per_wave;
  const int s1 = min(wg_end, s0 + per_wave);

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const uint8_t* wrow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wrow[t] = p.w + (size_t)(ncol + t) * p.K + 32 * g;
  const char* arow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int r = min(m0 + 16 * i + c, p.M - 1);
    arow[i] = (const char*)p.a + ((size_t)r * p.lda + 32 * g) * (A8 ? 1 : 2);
  }

  for (int s = s0; s < s1; ++s) {
    u32x4 wq[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const u32x4* wp = reinterpret_cast<const u32x4*>(wrow[t] + (size_t)s * 128);
      wq[t][0] = __builtin_nontemporal_load(wp);
      wq[t][1] = __builtin_nontemporal_load(wp + 1);
    }
    if constexpr (A8) {
      u32x4 aq[MT][2];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const u32x4* ap = reinterpret_cast<const u32x4*>(arow[i] + (size_t)s * 128);
        aq[i][0] = ap[0];
        aq[i][1] = ap[1];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          long b = (long)(((uint64_t)wq[t][j >> 1][2 * (j & 1) + 1] << 32) | wq[t][j >> 1][2 * (j & 1)]);
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            long a = (long)(((uint64_t)aq[i][j >> 1][2 * (j & 1) + 1] << 32) | aq[i][j >> 1][2 * (j & 1)]);
            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, acc[i][t], 0, 0, 0);
          }
        }
    } else {
      u32x4 af[MT][4];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const u32x4* ap = reinterpret_cast<const u32x4*>(arow[i] + (size_t)s * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j) af[i][j] = ap[j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          u32x4 b = fp8x8_to_T<T>(wq[t][j >> 1][2 * (j & 1)], wq[t][j >> 1][2 * (j & 1) + 1]);
#pragma unroll
          for (int i = 0; i < MT; ++i) acc[i][t] = mfma_hp<T>(af[i][j], b, acc[i][t]);
        }
    }
  }

#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t)
      *reinterpret_cast<f32x4*>(&red[((wave * (MT * NT) + i * NT + t) * 64 + lane) * 4]) = acc[i][t];
  __syncthreads();
  for (int idx = wave; idx < MT * 4; idx += FNW) {
    const int i = idx >> 2, r = idx & 3;
    const int row = m0 + 16 * i + 4 * g + r;
    float v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float sum = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < FNW; ++w2) sum += red[((w2 * (MT * NT) + i * NT + t) * 64 + lane) * 4 + r];
      v[t] = sum;
    }
    if (row < p.M) {
      if (p.ksplit == 1 && !p.force_partial) {
        const float sa = p.a_scales ? p.a_scales[p.a_per_token ? row : 0] : 1.f;
        typename T::storage* cp = (typename T::storage*)p.c + (size_t)row * p.N + ncol;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float sb = p.b_scales ? p.b_scales[p.b_per_channel ? ncol + t : 0] : 1.f;
          float o = sa * (sb * v[t]);  // order of test_cutlass.py:43
          if (p.bias) o += T::to_f32(((const typename T::storage*)p.bias)[ncol + t]);
          cp[t] = T::from_f32(o);
        }
      } else {
        float* pp = p.partial + ((size_t)blockIdx.y * p.M + row) * p.N + ncol;
#pragma unroll
        for (int t = 0; t < NT; ++t) pp[t] = v[t];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// W8A8 fast path (decode shapes): the structure that the int4 kernel converged on
// (wna16_gemm.hip) -- 4 waves split K of one 64-column tile, a wave owns exactly
// NSEG macro steps of 128 k (template parameter -> straight-line code), weight
// loads run two macro steps ahead in registers, every global read is a buffer load
// whose per-lane address part is loop invariant and whose k part is an SGPR offset.
// No unpack at all: the 32 contiguous bytes a lane reads of one weight row are four
// fp8 MFMA B fragments as they are.
// ---------------------------------------------------------------------------
constexpr int F8W = 4;  // waves per workgroup, fast kernel

__device__ __forceinline__ __amdgpu_buffer_rsrc_t fp8_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

template <typename T, int NT, int MT, int NSEG>
__global__ __launch_bounds__(F8W * 64, 2) void fp8_gemm_fast_kernel(Fp8GemmParams p) {
  constexpr int DEPTH = NSEG < 2 ? NSEG : 2;
  constexpr int NBUF = DEPTH + 1;
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4;
  const int c = lane & 15;
  const int n0 = blockIdx.x * (16 * NT);
  const int m0 = blockIdx.z * (16 * MT);
  const int ncol = n0 + NT * c;
  const int seg0 = (blockIdx.y * F8W + wave) * NSEG;  // host guarantees K == ksplit * F8W * NSEG * 128

  const __amdgpu_buffer_rsrc_t rw = fp8_rsrc(p.w, (uint32_t)((size_t)p.N * p.K));
  const __amdgpu_buffer_rsrc_t ra = fp8_rsrc(p.a, (uint32_t)((size_t)p.M * p.lda));
  int voff_w[NT], voff_a[MT];
#pragma unroll
  for (int t = 0; t < NT; ++t) voff_w[t] = (ncol + t) * p.K + 16 * g;   // lane g: bytes [16g,16g+16) and [64+16g, ..) of the 128-B step
#pragma unroll
  for (int i = 0; i < MT; ++i) voff_a[i] = min(m0 + 16 * i + c, p.M - 1) * p.lda + 16 * g;  // same k mapping as the weights

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 wq[NBUF][NT][2];
  u32x4 aq[2][MT][2];
  auto load_w = [&](u32x4 (&wd)[NT][2], int s) {
    const int soff = (seg0 + s) * 128;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      wd[t][0] = __builtin_amdgcn_raw_buffer_load_b128(rw, voff_w[t], soff, 2);
      wd[t][1] = __builtin_amdgcn_raw_buffer_load_b128(rw, voff_w[t] + 64, soff, 2);
    }
  };
  auto load_a = [&](u32x4 (&ad)[MT][2], int s) {
    const int soff = (seg0 + s) * 128;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      ad[i][0] = __builtin_amdgcn_raw_buffer_load_b128(ra, voff_a[i], soff, 0);
      ad[i][1] = __builtin_amdgcn_raw_buffer_load_b128(ra, voff_a[i] + 64, soff, 0);
    }
  };

  load_a(aq[0], 0);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load_w(wq[d], d);
  __builtin_amdgcn_sched_barrier(0);

#pragma unroll
  for (int s = 0; s < NSEG; ++s) {
    if (s + 1 < NSEG) load_a(aq[(s + 1) & 1], s + 1);
    if (s + DEPTH < NSEG) load_w(wq[(s + DEPTH) % NBUF], s + DEPTH);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const u32x4 wv = wq[s % NBUF][t][j >> 1];
        const long b = (long)(((uint64_t)wv[2 * (j & 1) + 1] << 32) | wv[2 * (j & 1)]);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const u32x4 av = aq[s & 1][i][j >> 1];
          const long a = (long)(((uint64_t)av[2 * (j & 1) + 1] << 32) | av[2 * (j & 1)]);
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, acc[i][t], 0, 0, 0);
        }
      }
  }

  // ---- in-workgroup split-K reduction through LDS, dequant epilogue -----------------
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t)
      *reinterpret_cast<f32x4*>(&red[((wave * (MT * NT) + i * NT + t) * 64 + lane) * 4]) = acc[i][t];
  __syncthreads();
  for (int idx = wave; idx < MT * 4; idx += F8W) {
    const int i = idx >> 2, r = idx & 3;
    const int row = m0 + 16 * i + 4 * g + r;
    float v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float sum = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < F8W; ++w2) sum += red[((w2 * (MT * NT) + i * NT + t) * 64 + lane) * 4 + r];
      v[t] = sum;
    }
    if (row < p.M) {
      if (p.ksplit == 1 && !p.force_partial) {
        const float sa = p.a_scales ? p.a_scales[p.a_per_token ? row : 0] : 1.f;
        typename T::storage* cp = (typename T::storage*)p.c + (size_t)row * p.N + ncol;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float sb = p.b_scales ? p.b_scales[p.b_per_channel ? ncol + t : 0] : 1.f;
          float o = sa * (sb * v[t]);  // order of test_cutlass.py:43
          if (p.bias) o += T::to_f32(((const typename T::storage*)p.bias)[ncol + t]);
          cp[t] = T::from_f32(o);
        }
      } else {
        float* pp = p.partial + ((size_t)blockIdx.y * p.M + row) * p.N + ncol;
#pragma unroll
        for (int t = 0; t < NT; ++t) pp[t] = v[t];
      }
    }
  }
}

template <typename T>
__global__ void fp8_splitk_reduce_kernel(Fp8GemmParams p) {
  const int64_t mn = (int64_t)p.M * p.N;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= mn) return;
  float s = 0.f;
  for (int k = 0; k < p.ksplit; ++k) s += p.partial[(size_t)k * mn + i];
  const int row = (int)(i / p.N), col = (int)(i % p.N);
  const float sa = p.a_scales ? p.a_scales[p.a_per_token ? row : 0] : 1.f;
  const float sb = p.b_scales ? p.b_scales[p.b_per_channel ? col : 0] : 1.f;
  float o = sa * (sb * s);
  if (p.bias) o += T::to_f32(((const typename T::storage*)p.bias)[col]);
  ((typename T::storage*)p.c)[i] = T::from_f32(o);
}

struct Fp8Plan { int nt, mt, ksplit, msteps_per_split, nseg; };

static Fp8Plan make_fp8_plan(int64_t M, int64_t N, int64_t K, bool a8) {
  Fp8Plan pl;
  pl.nseg = 0;
  if (a8 && N % 64 == 0 && K % 128 == 0 && N * K < (int64_t)0xffffffff && !APHRO_LAB_ENV_INT("APHRO_FP8_GENERIC", 0)) {
    // fast W8A8 kernel: every wave owns exactly NSEG macro steps; split across workgroups only
    // while the grid leaves CUs idle
    const int total = (int)(K / 128);
    const int64_t tiles = N / 64 * ((M + 31) / 32);
    int best_ns = 0, best_split = 0;
    for (int split = 1; split <= 8; ++split) {
      if (total % (split * F8W) != 0) continue;
      const int ns = total / (split * F8W);
      if (!(ns == 1 || ns == 2 || ns == 4 || ns == 7 || ns == 8)) continue;
      if (best_ns == 0) { best_ns = ns; best_split = split; }
      else if (tiles * best_split < 192 && tiles * split <= 1100) { best_ns = ns; best_split = split; }
    }
    if (best_ns) {
      pl.nt = 4; pl.mt = M > 16 ? 2 : 1; pl.nseg = best_ns; pl.ksplit = best_split;
      pl.msteps_per_split = F8W * best_ns;
      return pl;
    }
  }
  pl.nt = (N % 64 == 0 && N / 64 >= 192) ? 4 : (N % 32 == 0 ? 2 : 1);
  { const int v = APHRO_LAB_ENV_INT("APHRO_FP8_NT", 0); if ((v == 1 || v == 2 || v == 4) && N % (16 * v) == 0) pl.nt = v; }
  pl.mt = M > 16 ? 2 : 1;
  const int64_t tiles = N / (16 * pl.nt) * ((M + 16 * pl.mt - 1) / (16 * pl.mt));
  const int total = (int)(K / 128);
  int target = (int)((256 + tiles / 2) / tiles);
  if (target < 1) target = 1;
  if (target > 8) target = 8;
  if (APHRO_LAB_ENV_INT("APHRO_FP8_KSPLIT", 0) > 0) target = APHRO_LAB_ENV_INT("APHRO_FP8_KSPLIT", 0);
  // prefer splits that keep whole macro steps per wave
  int best = 1;
  for (int ks = 1; ks <= target; ++ks)
    if (total % (ks * FNW) == 0 || ks == 1) best = ks;
  if (total / best < 1) best = 1;
  pl.ksplit = best;
  pl.msteps_per_split = (total + best - 1) / best;
  pl.ksplit = (total + pl.msteps_per_split - 1) / pl.msteps_per_split;
  return pl;
}

template <typename T, bool A8>
static int run_fp8(Fp8GemmParams p, const Fp8Plan& pl, hipStream_t st) {
  dim3 grid((unsigned)(p.N / (16 * pl.nt)), (unsigned)pl.ksplit, (unsigned)((p.M + 16 * pl.mt - 1) / (16 * pl.mt)));
  if (pl.nseg > 0) {
    if constexpr (A8) {
      size_t flds = (size_t)F8W * pl.mt * 4 * 64 * 4 * sizeof(float);
#define LF(MTV, NS) hipLaunchKernelGGL((fp8_gemm_fast_kernel<T, 4, MTV, NS>), grid, dim3(F8W * 64), flds, st, p)
#define LFM(NS) { if (pl.mt == 2) LF(2, NS); else LF(1, NS); }
      switch (pl.nseg) {
        case 8: LFM(8) break;
        case 7: LFM(7) break;
        case 4: LFM(4) break;
        case 2: LFM(2) break;
        default: LFM(1) break;
      }
#undef LFM
#undef LF
      APHRO_LAUNCH_CHECK();
      if (pl.ksplit > 1 && !p.force_partial) {
        int64_t mn = (int64_t)p.M * p.N;
        hipLaunchKernelGGL((fp8_splitk_reduce_kernel<T>), dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, st, p);
        APHRO_LAUNCH_CHECK();
      }
      return APHRO_OK;
    }
  }
  size_t lds = (size_t)FNW * pl.mt * pl.nt * 64 * 4 * sizeof(float);
#define L(NTV, MTV) hipLaunchKernelGGL((fp8_gemm_kernel<T, A8, NTV, MTV>), grid, dim3(FNW * 64), lds, st, p)
  switch (pl.nt * 10 + pl.mt) {
    case 41: L(4, 1); break;
    case 42: L(4, 2); break;
    case 21: L(2, 1); break;
    case 22: L(2, 2); break;
    case 11: L(1, 1); break;
    default: L(1, 2); break;
  }
#undef L
  APHRO_LAUNCH_CHECK();
  if (pl.ksplit > 1 && !p.force_partial) {
    int64_t mn = (int64_t)p.M * p.N;
    hipLaunchKernelGGL((fp8_splitk_reduce_kernel<T>), dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, st, p);
    APHRO_LAUNCH_CHECK();
  }
  return APHRO_OK;
}

static int fp8_gemm_common(Fp8GemmParams p, bool a8, int dtype, void* workspace, size_t workspace_bytes,
                           hipStream_t st) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "fp8 gemm: output dtype must be f16 or bf16");
  APHRO_CHECK(p.K % 128 == 0, "fp8 gemm: K=%d must be a multiple of 128", p.K);
  APHRO_CHECK(p.N % 16 == 0, "fp8 gemm: N=%d must be a multiple of 16", p.N);
  APHRO_CHECK(p.M <= 64, "fp8 gemm: M=%d exceeds 64 rows per call", p.M);
  if (p.M == 0) return APHRO_OK;
  // W8A8 at <= 32 rows: the LDS-DMA streaming kernel (fp8_gemm_stream.hip) where it tiles the shape -- directly to `c` when
  // K fits one workgroup, raw slabs for a fused consumer otherwise
  if (a8 && p.lda == p.K) {
    const int split = aphro_fp8_gemm_stream_ksplit(p.M, p.N, p.K);
    if (split > 0 && (p.force_partial || split == 1)) {
      if (p.force_partial) {
        const size_t need = (size_t)split * p.M * p.N * sizeof(float);
        if (!workspace || workspace_bytes < need) {
          set_error("fp8 gemm: workspace %zu < %zu bytes", workspace_bytes, need);
          return APHRO_ERR_WORKSPACE;
        }
      }
      return aphro_fp8_gemm_stream(p.a, p.lda, p.w, p.a_scales, p.b_scales, p.bias, p.force_partial ? nullptr : p.c,
                                   p.force_partial ? (float*)workspace : nullptr, workspace_bytes, p.M, p.N, p.K,
                                   p.a_per_token, p.b_per_channel, dtype, (void*)st);
    }
  }
  Fp8Plan pl = make_fp8_plan(p.M, p.N, p.K, a8);
  if (pl.ksplit > 1 || p.force_partial) {
    size_t need = (size_t)pl.ksplit * p.M * p.N * sizeof(float);
    if (!workspace || workspace_bytes < need) {
      set_error("fp8 gemm: workspace %zu < %zu bytes", workspace_bytes, need);
      return APHRO_ERR_WORKSPACE;
    }
  }
  p.partial = (float*)workspace;
  p.ksplit = pl.ksplit;
  p.msteps_per_split = pl.msteps_per_split;
  if (dtype == APHRO_F16) return a8 ? run_fp8<Half, true>(p, pl, st) : run_fp8<Half, false>(p, pl, st);
  return a8 ? run_fp8<BFloat, true>(p, pl, st) : run_fp8<BFloat, false>(p, pl, st);
}

}  // namespace aphro

using namespace aphro;

extern "C" size_t aphro_fp8_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  (void)K;
  int64_t m = M < 64 ? M : 64;
  return (size_t)8 * m * N * sizeof(float);
}

extern "C" int aphro_scaled_mm_fp8(void* out, const void* a, const void* b, const float* a_scales,
                                   const float* b_scales, const void* bias, void* workspace,
                                   size_t workspace_bytes, int64_t M, int64_t N, int64_t K,
                                   int a_scale_per_token, int b_scale_per_channel, int out_dtype,
                                   void* stream) {
  APHRO_CHECK(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0, "scaled_mm: operands must be 16-byte aligned");
  Fp8GemmParams p;
  p.a = a; p.w = (const uint8_t*)b; p.a_scales = a_scales; p.b_scales = b_scales; p.bias = bias; p.c = out;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = (int)K;
  p.a_per_token = a_scale_per_token; p.b_per_channel = b_scale_per_channel; p.force_partial = 0;
  return fp8_gemm_common(p, true, out_dtype, workspace, workspace_bytes, (hipStream_t)stream);
}

// W8A8 GEMM that leaves the raw fp32 accumulators as split-K slabs [ksplit][M][N] for a fused
// consumer (aphro_fused_add_rms_norm_quant_fp8, the rotary/attention kernel) which sums them and
// applies a_scale * (b_scale * acc) itself -- no reduce launch, no fp16 round trip.
extern "C" int aphro_fp8_gemm_ksplit(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || M > 64 || K % 128 != 0 || N % 16 != 0) return -1;
  const int split = aphro_fp8_gemm_stream_ksplit(M, N, K);     // (what fp8_gemm_common will run)
  if (split > 0) return split;
  return make_fp8_plan(M, N, K, true).ksplit;
}

extern "C" int aphro_scaled_mm_fp8_slabs(const void* a, const void* b, float* partials, size_t partial_bytes,
                                         int64_t M, int64_t N, int64_t K, void* stream) {
  APHRO_CHECK(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0, "scaled_mm: operands must be 16-byte aligned");
  Fp8GemmParams p;
  p.a = a; p.w = (const uint8_t*)b; p.a_scales = nullptr; p.b_scales = nullptr; p.bias = nullptr; p.c = nullptr;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = (int)K;
  p.a_per_token = 0; p.b_per_channel = 0; p.force_partial = 1;
  return fp8_gemm_common(p, true, APHRO_F16, partials, partial_bytes, (hipStream_t)stream);
}

extern "C" int aphro_fp8_w8a16_gemm(void* out, const void* a, const void* w, const float* w_scales,
                                    const void* bias, void* workspace, size_t workspace_bytes, int64_t M,
                                    int64_t N, int64_t K, int64_t lda, int w_scale_per_channel, int dtype,
                                    void* stream) {
  APHRO_CHECK(((uintptr_t)a % 16) == 0 && lda % 8 == 0, "fp8_w8a16: a must be 16-byte aligned");
  Fp8GemmParams p;
  p.a = a; p.w = (const uint8_t*)w; p.a_scales = nullptr; p.b_scales = w_scales; p.bias = bias; p.c = out;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = (int)lda;
  p.a_per_token = 0; p.b_per_channel = w_scale_per_channel; p.force_partial = 0;
  return fp8_gemm_common(p, false, dtype, workspace, workspace_bytes, (hipStream_t)stream);
}
