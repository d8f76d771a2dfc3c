// FP8 W8A8 decode GEMM for M <= 32 with the shape of work of wna16_gemm_resident.hip -- round 4.  Same role and arithmetic
// as aphro_scaled_mm_fp8 / aphro_scaled_mm_fp8_slabs (cutlass_scaled_mm, kernels/quantization/cutlass_w8a8/
// scaled_mm_entry.cu:92-137; on ROCm torch._scaled_mm, quantization/utils/w8a8_utils.py:83-183):
// out = a_scales (.) (A_q . W_q^T) (.) b_scales + bias, or the raw fp32 accumulators for a fused consumer.
//
// Why another FP8 kernel: fp8_gemm_stream_kernel walks 16-column tiles with all 8 waves splitting K and pays a cross-wave
// LDS reduce + two workgroup barriers per 16 columns (gate_up: 29.6 us for 117 MB, a 20 us stream); fp8_gemm_fast_kernel
// keeps two macro steps per wave group in registers (0.31-0.44 of the roofline on qkv / o / down).  What the int4 lab of
// this round established carries over unchanged: one workgroup per CU with equal work, a wave's weights as ONE long stream
// of lane-linear 1 KiB pieces, the activations crossing the vector L1 once per workgroup, one reduction at the end.
//
//   * grid = (N / CW) column strips x K slices = 256 workgroups for the Llama-3-8B projections (gate_up: 112-column strips
//     over the full K; down: 64 columns x K / 4; qkv: 48 x K / 2; o: 64 x K / 4), 4 waves (one per SIMD) split the K range.
//   * a k-PAIR (64 k) is the unit: lane (g, c) of a wave holds 16 bytes = k0 + 16 g .. + 16 of row c -- the low 8 bytes feed
//     one v_mfma_f32_16x16x32_fp8_fp8, the high 8 the next (any bijection between (lane group, byte) and k is a valid MFMA
//     k order as long as A and W use the same one).  So both operands are plain 16-byte loads; no LDS, no swizzle.
//   * A (row-major e4m3 [M, lda], as the quantising producers leave it) is used once per wave (a k-pair covers all the
//     strip's columns): it streams in the same register ring as the weights.
//   * W is read from a STRIP-MAJOR copy (aphro_fp8_strip_relayout, load time): for (K slice, strip, wave) the pieces
//     [k-pair][16-column tile] of 1 KiB in the order the wave reads them, DEPTH k-pairs ahead in a register ring.
//   * K reduction over the waves through an LDS [wave][row][column] tile; epilogue sa * (sb * acc) (+ bias) in the reference's
//     order (tests/kernels/test_cutlass.py:43), or raw fp32 slabs [ksplit][M][N].
#include <utility>

#include "common.h"

namespace aphro {

struct Fp8ResParams {
  const uint8_t* a;       // e4m3 [M, lda]
  const uint8_t* w;       // strip-major e4m3
  const float* a_scales;  // [1] or [M]
  const float* b_scales;  // [1] or [N]
  const void* bias;       // T [N] or NULL
  void* c;                // T [M, N] (one K slice)
  float* slab;            // fp32 [ksplit][M][N]
  int M, N, K, lda;
  int a_per_token, b_per_channel;
  int ksplit;
  int strips, xcd_shift, strips_per_xcd;   // workgroup placement, worked out by the host (see the launcher)
};

template <int B, int E, typename F>
__device__ __forceinline__ void f8r_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    f8r_static_for<B + 1, E>(f);
  }
}

__device__ __forceinline__ long f8r_lo(u32x4 v) { return (long)(((uint64_t)v[1] << 32) | v[0]); }
__device__ __forceinline__ long f8r_hi(u32x4 v) { return (long)(((uint64_t)v[3] << 32) | v[2]); }

// MT: 16-token tiles.  NSEG: 128-k segments per wave (2 k-pairs each).  NT: 16-column tiles per strip.  D: k-pairs in flight.
// Kernel arguments as in wna16_gemm_stream_kernel: what the first loads need comes first and as scalars (preloaded into
// SGPRs, Makefile: -amdgpu-kernarg-preload-count); p_in carries the rest.
template <typename T, int MT, int NSEG, int NT, int D>
__global__ __launch_bounds__(256, 1) void fp8_gemm_resident_kernel(const uint8_t* w, const uint8_t* a, int strips, int xcd_shift,
                                                                   int strips_per_xcd, int ksplit, int M, int N, int K, int lda,
                                                                   Fp8ResParams p_in) {
  Fp8ResParams p = p_in;
  p.w = w; p.a = a; p.strips = strips; p.xcd_shift = xcd_shift; p.strips_per_xcd = strips_per_xcd; p.ksplit = ksplit;
  p.M = M; p.N = N; p.K = K; p.lda = lda;
  constexpr int NWV = 4;
  constexpr int NKP = 2 * NSEG;
  constexpr int CW = 16 * NT;
  constexpr int CWP = CW + 4;
  constexpr int ROWS = 16 * MT;
  constexpr int DD = D < NKP ? D : NKP;
  constexpr int RING = DD + 1;
  extern __shared__ __attribute__((aligned(16))) float red[];     // [NWV][ROWS][CWP]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  // workgroup -> (strip, K slice): the placement rules of wna16_gemm_resident.hip (a K slice's strips on 8 / ksplit XCDs)
  // (the divisions by 8 / ksplit are the host's: in the kernel they were ~120 instructions in front of the first address,
  // profiles/r5_decode_experiments.txt (2))
  const int S = p.strips;
  int strip, ky;
  if (p.xcd_shift >= 0) {
    const int L = blockIdx.y * S + blockIdx.x, xcd = L & 7, idx = L >> 3;
    ky = xcd >> p.xcd_shift;
    strip = (xcd & ((1 << p.xcd_shift) - 1)) * p.strips_per_xcd + idx;
  } else {
    strip = (S & 7) == 0 ? (int)(blockIdx.x & 7) * (S >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    ky = blockIdx.y;
  }
  const int kp0 = (ky * NWV + wave) * NKP;          // first k-pair of this wave
  const int cb = strip * CW;

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.w), 0,
      (uint32_t)((size_t)p.N * p.K), 0x00020000);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.a), 0,
      (uint32_t)((size_t)(p.M - 1) * p.lda + p.K), 0x00020000);
  constexpr int WAVE_BYTES = NKP * NT * 1024;
  const int sbase = ((ky * S + strip) * NWV + wave) * WAVE_BYTES;
  const int voff_w = lane * 16;

  // ---- one stream per wave: the activations of k-pair kp (lane (g, c) = token 16 i + c, bytes k + 16 g .. + 16: used ONCE, every
  // k-pair covers all the strip's columns) ride in the same register ring as its weights, DD k-pairs ahead.  (First version:
  // all of A loaded up front -- 32 gathers of 16 x 64 bytes in front of the first weight byte of every wave.)
  int voff_a[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) voff_a[i] = min(16 * i + c, p.M - 1) * p.lda + 16 * g;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 wr[RING][NT], ar[RING][MT];
  auto load_kp = [&](auto KP_) {
    constexpr int kp = decltype(KP_)::value;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int vo = voff_a[i];
      ar[kp % RING][i] = __builtin_amdgcn_raw_buffer_load_b128(ra, vo, (kp0 + kp) * 64, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
      wr[kp % RING][t] = __builtin_amdgcn_raw_buffer_load_b128(rw, voff_w, sbase + (kp * NT + t) * 1024, 2);
  };
  f8r_static_for<0, DD>([&](auto KP_) { load_kp(KP_); });
  __builtin_amdgcn_sched_barrier(0);

  f8r_static_for<0, NKP>([&](auto KP_) {
    constexpr int kp = decltype(KP_)::value;
    if constexpr (kp + DD < NKP) load_kp(std::integral_constant<int, (kp + DD < NKP ? kp + DD : 0)>{});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const u32x4 b = wr[kp % RING][t];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(f8r_lo(ar[kp % RING][i]), f8r_lo(b), acc[i][t], 0, 0, 0);
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(f8r_hi(ar[kp % RING][i]), f8r_hi(b), acc[i][t], 0, 0, 0);
      }
    }
  });

  // ---- K reduction over the waves: D[token 16 i + 4 g + r][column 16 t + c] ------------------------------------------------
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * ROWS + 16 * i + 4 * g + r) * CWP + 16 * t + c] = acc[i][t][r];
  __syncthreads();
  constexpr int UNITS = ROWS * (CW / 4);            // 4 columns of one row
  for (int unit = threadIdx.x; unit < UNITS; unit += NWV * 64) {
    const int row = unit / (CW / 4), c4 = unit % (CW / 4);
    f32x4 sum = *reinterpret_cast<const f32x4*>(&red[row * CWP + 4 * c4]);
#pragma unroll
    for (int w2 = 1; w2 < NWV; ++w2) sum += *reinterpret_cast<const f32x4*>(&red[(w2 * ROWS + row) * CWP + 4 * c4]);
    if (row >= p.M) continue;
    const int n = cb + 4 * c4;
    if (p.slab) {
      *reinterpret_cast<f32x4*>(p.slab + ((size_t)ky * p.M + row) * p.N + n) = sum;
    } else {
      const float sa = p.a_scales ? p.a_scales[p.a_per_token ? row : 0] : 1.f;
      uint16_t o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sb = p.b_scales ? p.b_scales[p.b_per_channel ? n + e : 0] : 1.f;
        float v = sa * (sb * sum[e]);                // order of test_cutlass.py:43
        if (p.bias) v += T::to_f32(((const typename T::storage*)p.bias)[n + e]);
        o[e] = T::from_f32(v);
      }
      *reinterpret_cast<u32x2*>((typename T::storage*)p.c + (size_t)row * p.N + n) =
          u32x2{(uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16)};
    }
  }
}

// [N, K] row-major e4m3 -> strip-major: one thread per 16-byte piece.
__global__ void fp8_strip_relayout_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, int N, int K, int nseg, int nt,
                                          int ksplit) {
  const int64_t total = (int64_t)N * K / 16;
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= total) return;
  const int nkp = 2 * nseg, S = N / (16 * nt);
  const int64_t wave_pieces = (int64_t)nkp * nt * 64;
  int64_t r = d;
  const int64_t widx = r / wave_pieces; r -= widx * wave_pieces;      // ((ky * S + strip) * 4 + wave)
  const int wave = (int)(widx % 4);
  const int strip = (int)((widx / 4) % S);
  const int ky = (int)(widx / 4 / S);
  const int kp = (int)(r / (nt * 64)); r %= nt * 64;
  const int t = (int)(r / 64), lane = (int)(r % 64);
  const int g = lane >> 4, c = lane & 15;
  const int n = strip * 16 * nt + 16 * t + c;
  const int64_t k = ((int64_t)(ky * 4 + wave) * nkp + kp) * 64 + 16 * g;
  out[d] = in[((int64_t)n * K + k) / 16];
}

}  // namespace aphro

using namespace aphro;

struct Fp8ResConfig { int nseg, nt, ksplit; };

#define F8R_CONFIGS(X) \
  X(8, 7)              \
  X(7, 4)              \
  X(4, 3)              \
  X(2, 4)              \
  X(4, 4)              \
  X(8, 4)              \
  X(4, 2)              \
  X(8, 2)

// The plan for (M, N, K), nseg == 0: not served.  One workgroup per CU with equal work, as close to the CU count as the
// divisibility permits; candidates in order of preference (wide strips first: fewer activation bytes per weight byte).
static Fp8ResConfig f8r_plan(int64_t M, int64_t N, int64_t K) {
  const Fp8ResConfig none = {0, 0, 0};
  if (M < 1 || M > 32 || K % 128 != 0 || N % 16 != 0 || (size_t)N * K >= 0x7fffffffull) return none;
  if (getenv("APHRO_FP8_NO_RESIDENT")) return none;
  const int segs = (int)(K / 128);
  static const int cand[][2] = {{8, 7}, {7, 4}, {4, 3}, {2, 4}, {4, 4}, {8, 4}, {4, 2}, {8, 2}};
  for (const auto& cd : cand) {
    const int nseg = cd[0], nt = cd[1];
    if (N % (16 * nt) != 0 || segs % (4 * nseg) != 0) continue;
    const int ks = segs / (4 * nseg);
    if (ks < 1 || ks > 8) continue;
    const int64_t wgs = N / (16 * nt) * ks;
    if (wgs >= 192 && wgs <= 256) return Fp8ResConfig{nseg, nt, ks};
  }
  return none;
}

// K slices of the plan (fp32 slabs when > 1), 0: shape not served (the caller keeps aphro_scaled_mm_fp8[_slabs]).
extern "C" int aphro_fp8_gemm_resident_ksplit(int64_t M, int64_t N, int64_t K) { return f8r_plan(M, N, K).ksplit; }

// Load time: [N, K] row-major e4m3 (the checkpoint tensor) -> the strip-major order of the plan for (M, N, K).  out != w.
extern "C" int aphro_fp8_strip_relayout(const void* w, void* out, int64_t M, int64_t N, int64_t K, void* stream) {
  const Fp8ResConfig cf = f8r_plan(M, N, K);
  APHRO_CHECK(cf.nseg != 0 && w != out, "fp8_strip_relayout: M=%ld N=%ld K=%ld is not served by the resident kernel", (long)M, (long)N, (long)K);
  APHRO_CHECK(((uintptr_t)w % 16) == 0 && ((uintptr_t)out % 16) == 0, "fp8_strip_relayout: 16-byte alignment required");
  const int64_t total = N * K / 16;
  hipLaunchKernelGGL(fp8_strip_relayout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)w, (u32x4*)out, (int)N, (int)K, cf.nseg, cf.nt, cf.ksplit);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

// out ([M, N] in `dtype`, plans with one K slice only) or slabs ([ksplit][M][N] raw fp32 accumulators); exactly one of the two.
// w_strip: aphro_fp8_strip_relayout's output for the same (M <= 32 class, N, K).
extern "C" int aphro_fp8_gemm_resident(const void* a, int64_t lda, const void* w_strip, const float* a_scales,
                                       const float* b_scales, const void* bias, void* out, float* slabs, size_t slabs_bytes,
                                       int64_t M, int64_t N, int64_t K, int a_scale_per_token, int b_scale_per_channel,
                                       int dtype, void* stream) {
  const Fp8ResConfig cf = f8r_plan(M, N, K);
  APHRO_CHECK(cf.nseg != 0, "fp8_gemm_resident: M=%ld N=%ld K=%ld is not served", (long)M, (long)N, (long)K);
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "fp8_gemm_resident: output dtype must be f16 or bf16");
  APHRO_CHECK((out != nullptr) != (slabs != nullptr), "fp8_gemm_resident: exactly one of out / slabs");
  APHRO_CHECK(out == nullptr || cf.ksplit == 1, "fp8_gemm_resident: this shape is K-sliced (%d): slabs only", cf.ksplit);
  APHRO_CHECK(slabs == nullptr || slabs_bytes >= (size_t)cf.ksplit * M * N * sizeof(float), "fp8_gemm_resident: slabs too small");
  APHRO_CHECK(((uintptr_t)a % 16) == 0 && ((uintptr_t)w_strip % 16) == 0 && lda % 16 == 0 && lda >= K &&
              (out == nullptr || ((uintptr_t)out % 8) == 0) && (slabs == nullptr || ((uintptr_t)slabs % 16) == 0),
              "fp8_gemm_resident: alignment (16-byte rows of a, lda %% 16 == 0)");
  Fp8ResParams p;
  p.a = (const uint8_t*)a; p.w = (const uint8_t*)w_strip; p.a_scales = a_scales; p.b_scales = b_scales; p.bias = bias;
  p.c = out; p.slab = slabs; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = (int)lda;
  p.a_per_token = a_scale_per_token; p.b_per_channel = b_scale_per_channel; p.ksplit = cf.ksplit;
  const int mt = M > 16 ? 2 : 1;
  const dim3 grid((unsigned)(N / (16 * cf.nt)), (unsigned)cf.ksplit);
  p.strips = (int)grid.x; p.xcd_shift = -1; p.strips_per_xcd = 0;
  if (cf.ksplit > 1 && 8 % cf.ksplit == 0 && p.strips % (8 / cf.ksplit) == 0) {
    const int per = 8 / cf.ksplit;         // XCDs per K slice: 4, 2, 1
    p.xcd_shift = per == 4 ? 2 : per == 2 ? 1 : 0;
    p.strips_per_xcd = p.strips / per;
  }
  // k-pairs in flight per wave (bench.py --quant fp8ct, per-kernel, depth 4 / 6 / 8: gate_up 22.8 / 23.3 / 23.8 us, down 12.6 /
  // 12.5 / 12.4, qkv 7.3 / 7.1 / 6.8, o 5.6 / 5.6 / 5.7): 4 for the wide strips, 8 for the narrow ones
#ifndef F8R_DEPTH
#define F8R_DEPTH(nt) ((nt) >= 6 ? 4 : 8)
#endif
#define L(TT, MTV, NSEGV, NTV)                                                                                          \
  {                                                                                                                     \
    auto kern = fp8_gemm_resident_kernel<TT, MTV, NSEGV, NTV, F8R_DEPTH(NTV)>;                                          \
    const size_t lds = (size_t)4 * 16 * MTV * (16 * NTV + 4) * sizeof(float);                                           \
    if (lds > 64 * 1024 &&                                                                                              \
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {  \
      set_error("fp8_gemm_resident: cannot raise the dynamic LDS limit to %zu", lds);                                   \
      return APHRO_ERR_LAUNCH;                                                                                          \
    }                                                                                                                   \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, p.w, p.a, p.strips, p.xcd_shift, p.strips_per_xcd,  \
                       p.ksplit, p.M, p.N, p.K, p.lda, p);                                                             \
  }
#define X(a_, b_)                                                  \
  if (cf.nseg == a_ && cf.nt == b_) {                              \
    if (dtype == APHRO_F16) { if (mt == 2) L(Half, 2, a_, b_) else L(Half, 1, a_, b_) }       \
    else { if (mt == 2) L(BFloat, 2, a_, b_) else L(BFloat, 1, a_, b_) }                      \
  }
  F8R_CONFIGS(X)
#undef X
#undef L
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
