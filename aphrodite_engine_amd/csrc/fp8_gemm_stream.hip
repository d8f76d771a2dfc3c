// FP8 W8A8 decode GEMM (M <= 32) on the LDS-DMA streaming structure of lm_head.hip -- round 3.  Same role as
// fp8_gemm.hip's small-M kernels (cutlass_scaled_mm, kernels/quantization/cutlass_w8a8/scaled_mm_entry.cu:92-137; on ROCm
// torch._scaled_mm, quantization/utils/w8a8_utils.py:83-183): out = a_scales (.) (A_q . W_q^T) (.) b_scales + bias, or the
// raw fp32 accumulators for a fused consumer.  W is [N, K] e4m3 row-major (the checkpoint layout), A [M, lda] e4m3.
//
//   * one 8-wave workgroup per CU, tiles of 16 weight rows (output columns) dealt round-robin; the waves split K
//     (128-k segments interleaved over the waves, so that a request group covers 1 KiB of a row) and keep their
//     slice of the activations resident as fp8 MFMA A fragments;
//   * the weights go global -> LDS with `buffer_load ... lds`, 8 rows x 128 bytes per instruction (8 full cache lines),
//     chunks XOR-swizzled by the row pair so that the 8-byte B-fragment reads are bank-conflict free; a ring of 8 segments
//     (16 KiB) per wave = 128 KiB in flight per CU, no register touched on the way in (fp8_gemm_fast_kernel holds 2 macro
//     steps = 16 KiB per WAVE GROUP in registers and measures 0.43 of the roofline on the gate_up matrix);
//   * hand-counted vmcnt waits (hipcc does not order a ds_read after the LDS-DMA that fills it), B-fragment reads in inline
//     asm, an LDS-only workgroup barrier per tile (see lm_head.hip);
//   * K reduction over the waves through LDS once per tile, epilogue in the reference's order sa * (sb * acc) (+ bias).
//     Shapes whose K does not fit one workgroup's 8 x 8 segments are K-sliced over grid.y: raw slabs only.
//   * SILU form (gate_up of an FP8 MLP whose down_proj has a STATIC input scale): a 16-row tile is 8 gate rows and the 8 up
//     rows of the same features (the two 8-row staging instructions of a segment simply start N/2 rows apart), so the
//     finishing threads of a tile hold gate and up of a feature 8 lanes apart: out = T(sa (sb acc)) per half as the GEMM
//     would store it, SiluAndMul in T, fp8(x * (1 / scale)) -- the bits of cutlass_scaled_mm + silu_and_mul +
//     static_scaled_fp8_quant (activation_kernels.cu:12-75, fp8/common.cu:187-199) without the [M, N] round trip and
//     two launches.
#include <utility>

#include "common.h"

namespace aphro {

struct Fp8StreamParams {
  const uint8_t* a;       // e4m3 [M, lda]
  const uint8_t* w;       // e4m3 [N, K]
  const float* a_scales;  // [1] or [M] (a_per_token)
  const float* b_scales;  // [1] or [N] (b_per_channel)
  const void* bias;       // T [N] or NULL
  void* c;                // T [M, N]           (one K slice)
  float* slab;            // fp32 [ksplit][M][N] raw accumulators
  int M, N, K, lda;
  int a_per_token, b_per_channel;
  int tiles;              // N / 16
  uint8_t* q_out;         // SILU form: e4m3 [M, N / 2]
  const float* q_scale;   // SILU form: the static scale of q_out ([1])
};

typedef __attribute__((address_space(3))) void* f8s_lds_ptr;

template <int B, int E, typename F>
__device__ __forceinline__ void f8s_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    f8s_static_for<B + 1, E>(f);
  }
}

__device__ __forceinline__ void f8s_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// KS: 128-k segments per wave and K slice (K = gridDim.y x 8 waves x KS x 128).  MT: 16-token tiles.
template <typename T, int MT, int KS, bool SILU = false>
__global__ __launch_bounds__(512, 1) void fp8_gemm_stream_kernel(const uint8_t* w, const uint8_t* a, int M, int N, int K,
                                                                 int lda, int tiles, Fp8StreamParams p_in) {
  // (leading scalars: preloaded into SGPRs, Makefile -amdgpu-kernarg-preload-count; p_in carries the rest)
  Fp8StreamParams p = p_in;
  p.w = w; p.a = a; p.M = M; p.N = N; p.K = K; p.lda = lda; p.tiles = tiles;
  constexpr int NWV = 8;
  constexpr int SEGB = 16 * 128;                    // one staged segment: [16 rows][8 chunks of 16 k]
  constexpr int R = 8;                              // ring slots per wave
  constexpr int RP = 17;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [NWV][R][SEGB] | red [NWV][16 MT][RP]
  float* const red = reinterpret_cast<float*>(smem + NWV * R * SEGB);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  unsigned char* const ring = smem + wave * R * SEGB;
  const int G = gridDim.x, ky = blockIdx.y;
  const int kbase = ky * NWV * KS * 128;            // this K slice
  auto kseg = [&](int s) { return kbase + (s * NWV + wave) * 128; };

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.w), 0,
      (uint32_t)((size_t)p.N * p.K), 0x00020000);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.a), 0,
      (uint32_t)((size_t)(p.M - 1) * p.lda + p.K), 0x00020000);

  // ---- the activations of this wave's K slice: fp8 A fragments, lane (g, c) = token 16 i + c, k = kseg(s) + 32 u + 8 g .. + 8
  u32x2 af[KS][4][MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int vo = min(16 * i + c, p.M - 1) * p.lda + 8 * g;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int u = 0; u < 4; ++u) af[s][u][i] = __builtin_amdgcn_raw_buffer_load_b64(ra, vo, kseg(s) + 32 * u, 0);
  }
  // staging instruction `it` (2 per segment): lane -> slot row 8 it + lane / 8, 16-byte chunk (lane % 8) ^ ((row / 2) % 8)
  int voff_w[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int srow = 8 * it + (lane >> 3);
    // global row of slot row srow: tile * 16 + srow, or (SILU) tile * 8 + lane / 8 of the gate (it 0) / up (it 1) half
    const int grow = SILU ? (it == 1 ? (p.N >> 1) : 0) + (lane >> 3) : srow;
    voff_w[it] = grow * p.K + (((lane & 7) ^ ((srow >> 1) & 7)) << 4);
  }
  // B fragment of k-step u: row c, bytes 32 u + 8 g .. + 8 of the segment = chunk 2 u + g / 2, half g % 2
  uint32_t rd[4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
    rd[u] = (uint32_t)(uintptr_t)(f8s_lds_ptr)(ring + c * 128 + (((2 * u + (g >> 1)) ^ ((c >> 1) & 7)) << 4) + ((g & 1) << 3));

  // linear step q = (tile index in this workgroup) * KS + s -> ring slot q % R
  const int ntile = (p.tiles - (int)blockIdx.x + G - 1) / G;      // tiles blockIdx.x, + G, ...
  const int Q = ntile * KS;
  auto stage = [&](int q) {                         // segment q -> its slot (q < Q)
    const int tile = (int)blockIdx.x + (q / KS) * G, s = q % KS;
    const int so = tile * (SILU ? 8 : 16) * p.K + kseg(s);
    unsigned char* dst = ring + (q % R) * SEGB;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int vo = voff_w[it];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (f8s_lds_ptr)(dst + it * 1024), 16, vo, so, 0, 0);
    }
  };
  const int tok = tid >> 4, col = tid & 15;
  float q_inv = 0.f;                                // SILU form: fetched once, ahead of the ring (byte stores may alias it)
  if constexpr (SILU) {
    q_inv = 1.0f / *(const volatile float*)p.q_scale;
    asm volatile("" : "+v"(q_inv));                 // (computed here: not sunk into the first tile's epilogue)
  }

  __builtin_amdgcn_sched_barrier(0);
  for (int q = 0; q < R && q < Q; ++q) stage(q);
  __builtin_amdgcn_sched_barrier(0);

  int q = 0;
  for (int ti = 0; ti < ntile; ++ti) {
    const int tile = (int)blockIdx.x + ti * G;
    f32x4 acc4[4][MT];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i) acc4[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f8s_static_for<0, KS>([&](auto S_) {
      constexpr int s = decltype(S_)::value;
      // segment q has landed when at most the 2 (R - 1) loads of the R - 1 younger segments are outstanding; in the last
      // R - 1 steps fewer were issued: wait for everything (a drain of the tail only)
      if (Q - 1 - q >= R - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (R - 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint32_t so = (uint32_t)((q % R) * SEGB);
      u32x2 bf[4];
      asm volatile(
          "ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(bf[0]), "=&v"(bf[1]), "=&v"(bf[2]), "=&v"(bf[3])
          : "v"(rd[0] + so), "v"(rd[1] + so), "v"(rd[2] + so), "v"(rd[3] + so)
          : "memory");
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long b = (long)(((uint64_t)bf[u][1] << 32) | bf[u][0]);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const long a = (long)(((uint64_t)af[s][u][i][1] << 32) | af[s][u][i][0]);
          acc4[u][i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, acc4[u][i], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (q + R < Q) stage(q + R);                  // the slot is free: its reads have returned
      __builtin_amdgcn_sched_barrier(0);
      ++q;
    });
    // ---- K reduction over the waves: D[token 4 g + r][column c] -------------------------------------------------------
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const f32x4 a4 = (acc4[0][i] + acc4[1][i]) + (acc4[2][i] + acc4[3][i]);
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 16 * MT + 16 * i + 4 * g + r) * RP + c] = a4[r];
    }
    f8s_lds_barrier();
    if (tok < 16 * MT && tok < p.M) {
      float sum = red[tok * RP + col];
#pragma unroll
      for (int w2 = 1; w2 < NWV; ++w2) sum += red[(w2 * 16 * MT + tok) * RP + col];
      if constexpr (SILU) {
        const int n = (col < 8 ? 0 : (p.N >> 1)) + tile * 8 + (col & 7);
        const float sa = p.a_scales ? p.a_scales[p.a_per_token ? tok : 0] : 1.f;
        const float sb = p.b_scales ? p.b_scales[p.b_per_channel ? n : 0] : 1.f;
        float o = sa * (sb * sum);
        if (p.bias) o += T::to_f32(((const typename T::storage*)p.bias)[n]);
        const float mine = T::to_f32(T::from_f32(o));           // what the GEMM would have stored
        const float other = __shfl_xor(mine, 8, 64);            // col < 8: up of the same feature
        if (col < 8) {
          const float act = T::to_f32(silu_mul_bits<T>(mine, other));
          const float qv = __builtin_fmaxf(-448.f, __builtin_fminf(act * q_inv, 448.f));
          p.q_out[(size_t)tok * (p.N >> 1) + tile * 8 + col] = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(qv, qv, 0, false) & 0xff);
        }
      } else {
      const int n = tile * 16 + col;
      if (p.slab) {
        p.slab[((size_t)ky * p.M + tok) * p.N + n] = sum;
      } else {
        const float sa = p.a_scales ? p.a_scales[p.a_per_token ? tok : 0] : 1.f;
        const float sb = p.b_scales ? p.b_scales[p.b_per_channel ? n : 0] : 1.f;
        float o = sa * (sb * sum);                  // order of test_cutlass.py:43
        if (p.bias) o += T::to_f32(((const typename T::storage*)p.bias)[n]);
        ((typename T::storage*)p.c)[(size_t)tok * p.N + n] = T::from_f32(o);
      }
      }
    }
    f8s_lds_barrier();
  }
}

}  // namespace aphro

using namespace aphro;

// K slices the streaming kernel needs for (M, N, K): 0 = not served.  K = slices x 8 waves x KS x 128 with KS <= 8.
extern "C" int aphro_fp8_gemm_stream_ksplit(int64_t M, int64_t N, int64_t K) {
  if (M < 1 || M > 32 || N % 16 != 0 || N < 16 || K % 1024 != 0 || (size_t)N * K >= 0x7fffffffull) return 0;
  if (APHRO_LAB_ENV_INT("APHRO_FP8_NO_STREAM", 0)) return 0;
  // Measured in the decode step (bench.py --quant fp8ct, same box, against fp8_gemm_fast_kernel): gate_up [28672, 4096]
  // 31.8 -> 29.9 us, but down [4096, 14336] 16.4 -> 22.9 (two K slices), qkv 9.7 -> 13.4, o 6.7 -> 11.3: a workgroup needs
  // several 16-row tiles to amortise its prologue (the resident A gather) -- so only wide matrices whose K fits one
  // workgroup take this kernel (APHRO_FP8_STREAM_ALL=1 lifts the restriction for measurements).
  const int segs = (int)(K / 128) / 8;              // per wave over all slices
  for (int ks = 8; ks >= 1; --ks)
    if (segs % ks == 0) {
      const int split = segs / ks;
      if (split > 8) return 0;
      if (!knobs().fp8_stream_all && (split != 1 || N / 16 < 4 * (int64_t)device_cu_count())) return 0;
      return split;
    }
  return 0;
}

// out (one K slice only) or slabs ([ksplit][M][N] raw fp32 accumulators); exactly one of the two.
extern "C" int aphro_fp8_gemm_stream(const void* a, int64_t lda, const void* w, const float* a_scales, const float* b_scales,
                                     const void* bias, void* out, float* slabs, size_t slabs_bytes, int64_t M, int64_t N,
                                     int64_t K, int a_scale_per_token, int b_scale_per_channel, int dtype, void* stream) {
  const int split = aphro_fp8_gemm_stream_ksplit(M, N, K);
  APHRO_CHECK(split > 0, "fp8_gemm_stream: M=%ld N=%ld K=%ld is not served", (long)M, (long)N, (long)K);
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "fp8_gemm_stream: output dtype must be f16 or bf16");
  APHRO_CHECK((out != nullptr) != (slabs != nullptr), "fp8_gemm_stream: exactly one of out / slabs");
  APHRO_CHECK(out == nullptr || split == 1, "fp8_gemm_stream: this shape is K-sliced (%d): slabs only", split);
  APHRO_CHECK(slabs == nullptr || slabs_bytes >= (size_t)split * M * N * sizeof(float), "fp8_gemm_stream: slabs too small");
  APHRO_CHECK(((uintptr_t)a % 8) == 0 && ((uintptr_t)w % 16) == 0 && lda % 8 == 0 && lda >= K, "fp8_gemm_stream: alignment");
  Fp8StreamParams p;
  p.a = (const uint8_t*)a; p.w = (const uint8_t*)w; p.a_scales = a_scales; p.b_scales = b_scales; p.bias = bias;
  p.c = out; p.slab = slabs; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = (int)lda;
  p.a_per_token = a_scale_per_token; p.b_per_channel = b_scale_per_channel; p.tiles = (int)(N / 16);
  p.q_out = nullptr; p.q_scale = nullptr;
  const int ks = (int)(K / 128) / 8 / split, mt = M > 16 ? 2 : 1;
  int gx = device_cu_count() / split;
  if (gx < 1) gx = 1;
  if (gx > p.tiles) gx = p.tiles;
  const size_t lds = (size_t)8 * 8 * 2048 + (size_t)8 * 16 * mt * 17 * sizeof(float);
  dim3 grid((unsigned)gx, (unsigned)split);
#define L(TT, MTV, KSV)                                                                                            \
  {                                                                                                                \
    auto kern = fp8_gemm_stream_kernel<TT, MTV, KSV>;                                                              \
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
      set_error("fp8_gemm_stream: cannot raise the dynamic LDS limit to %zu", lds);                                \
      return APHRO_ERR_LAUNCH;                                                                                     \
    }                                                                                                              \
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, (hipStream_t)stream, p.w, p.a, p.M, p.N, p.K, p.lda, p.tiles, p); \
  }
#define LK(TT, MTV)                                   \
  switch (ks) {                                       \
    case 1: L(TT, MTV, 1) break;                      \
    case 2: L(TT, MTV, 2) break;                      \
    case 3: L(TT, MTV, 3) break;                      \
    case 4: L(TT, MTV, 4) break;                      \
    case 5: L(TT, MTV, 5) break;                      \
    case 6: L(TT, MTV, 6) break;                      \
    case 7: L(TT, MTV, 7) break;                      \
    default: L(TT, MTV, 8) break;                     \
  }
  if (dtype == APHRO_F16) { if (mt == 2) LK(Half, 2) else LK(Half, 1) }
  else { if (mt == 2) LK(BFloat, 2) else LK(BFloat, 1) }
#undef LK
#undef L
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

// gate_up + SiluAndMul + static fp8 quantisation in ONE launch: w = [gate rows | up rows] ([N, K], N even halves),
// q_out e4m3 [M, N / 2] = fp8(T(silu(T(gate)) * T(up)) * (1 / *q_scale)), T = `dtype` (the model's activation dtype).
// Served when aphro_fp8_gemm_stream_silu_supported: K fits one workgroup, N % 32 == 0, N / 16 >= 4 x CUs.
extern "C" int aphro_fp8_gemm_stream_silu_supported(int64_t M, int64_t N, int64_t K) {
  return N % 32 == 0 && aphro_fp8_gemm_stream_ksplit(M, N, K) == 1;
}

extern "C" int aphro_fp8_gemm_stream_silu_quant(const void* a, int64_t lda, const void* w, const float* a_scales,
                                                const float* b_scales, const void* bias, void* q_out,
                                                const float* q_scale, int64_t M, int64_t N, int64_t K,
                                                int a_scale_per_token, int b_scale_per_channel, int dtype,
                                                void* stream) {
  APHRO_CHECK(aphro_fp8_gemm_stream_silu_supported(M, N, K), "fp8_gemm_stream_silu_quant: M=%ld N=%ld K=%ld is not served",
              (long)M, (long)N, (long)K);
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "fp8_gemm_stream_silu_quant: dtype must be f16 or bf16");
  APHRO_CHECK(q_out != nullptr && q_scale != nullptr, "fp8_gemm_stream_silu_quant: output / scale missing");
  APHRO_CHECK(((uintptr_t)a % 8) == 0 && ((uintptr_t)w % 16) == 0 && lda % 8 == 0 && lda >= K, "fp8_gemm_stream_silu_quant: alignment");
  Fp8StreamParams p;
  p.a = (const uint8_t*)a; p.w = (const uint8_t*)w; p.a_scales = a_scales; p.b_scales = b_scales; p.bias = bias;
  p.c = nullptr; p.slab = nullptr; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = (int)lda;
  p.a_per_token = a_scale_per_token; p.b_per_channel = b_scale_per_channel; p.tiles = (int)(N / 16);
  p.q_out = (uint8_t*)q_out; p.q_scale = q_scale;
  const int ks = (int)(K / 128) / 8, mt = M > 16 ? 2 : 1;
  int gx = device_cu_count();
  if (gx > p.tiles) gx = p.tiles;
  const size_t lds = (size_t)8 * 8 * 2048 + (size_t)8 * 16 * mt * 17 * sizeof(float);
  dim3 grid((unsigned)gx, 1);
#define L(TT, MTV, KSV)                                                                                            \
  {                                                                                                                \
    auto kern = fp8_gemm_stream_kernel<TT, MTV, KSV, true>;                                                        \
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
      set_error("fp8_gemm_stream_silu_quant: cannot raise the dynamic LDS limit to %zu", lds);                     \
      return APHRO_ERR_LAUNCH;                                                                                     \
    }                                                                                                              \
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, (hipStream_t)stream, p.w, p.a, p.M, p.N, p.K, p.lda, p.tiles, p); \
  }
#define LK(TT, MTV)                                   \
  switch (ks) {                                       \
    case 1: L(TT, MTV, 1) break;                      \
    case 2: L(TT, MTV, 2) break;                      \
    case 3: L(TT, MTV, 3) break;                      \
    case 4: L(TT, MTV, 4) break;                      \
    case 5: L(TT, MTV, 5) break;                      \
    case 6: L(TT, MTV, 6) break;                      \
    case 7: L(TT, MTV, 7) break;                      \
    default: L(TT, MTV, 8) break;                     \
  }
  if (dtype == APHRO_F16) { if (mt == 2) LK(Half, 2) else LK(Half, 1) }
  else { if (mt == 2) LK(BFloat, 2) else LK(BFloat, 1) }
#undef LK
#undef L
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
