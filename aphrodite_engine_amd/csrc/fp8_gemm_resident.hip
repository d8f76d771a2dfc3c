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
//
// Round 6 -- the launch diet of the dynamic per-token scheme (compressed_tensors_w8a8_fp8.py:133-141; quantiser
// kernels/quantization/fp8/common.cu:187-256).  Two forms of the same kernel remove the two quantising launches a decoder
// layer still had (scaled_fp8_quant of the attention output, silu_and_mul + quant):
//   * AQ ("quantise A on load"): A arrives as the PRODUCER's 16-bit activations [M, lda] plus absmax partials
//     [M][np] (one per producing workgroup: max is order-free, so max over partials == the row's absmax, bit for bit).  The
//     prologue reduces them to the row scale max(absmax / 448, 1 / (448 * 512)) (common.cu:205,233-240) and every k-pair's
//     A fragment is quantised in registers in front of its MFMAs: fp8(x / scale) with the IEEE quotient.  The quotient is
//     hipcc's own fp32 division sequence (v_rcp, one Newton step on the reciprocal, q = n r, residual corrections) with
//     the per-ROW part -- reciprocal + its refinement -- hoisted and ONE residual correction (proven sufficient for every
//     (input, scale) pair that can occur, see F8R_DIV_STEPS): 3 VALU per element, packed two at a time (v_pk_mul_f32 /
//     v_pk_fma_f32), on |x| with the sign put back on the packed bytes (-0 stays -0 as v_div_fixup_f32 would leave it).
//     v_div_scale_f32 is the identity for every finite 16-bit x and scale in [1 / (448 * 512), 65504 / 448]; the clamp to
//     +-448 is a no-op when |x| <= absmax.  tests: aphro_fp8_quant_rows_aq runs the same device function over all 65 536
//     f16 / bf16 inputs x adversarial scales against x / scale.
//   * SILU epilogue (plans with one K slice: gate_up): the strip-major copy holds (gate_j, up_j) as adjacent columns
//     (aphro_fp8_strip_relayout_interleaved), the epilogue dequantises both the way cutlass_scaled_mm does, applies
//     silu_and_mul (activation_kernels.cu:12-75 roundings: silu_mul_bits) and writes the 16-bit activation [M, N / 2] plus
//     this strip's absmax partial per row -- or, static scheme, e4m3 directly (static_scaled_fp8_quant, common.cu:187-199).
#include <string.h>

#include <utility>

#include "common.h"

namespace aphro {

struct Fp8ResParams {
  const uint8_t* a;       // e4m3 [M, lda]  (AQ: 16-bit T [M, lda], lda in elements)
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
  // AQ: absmax partials of the producers [M][np] (np % 4 == 0, np <= 256) and where workgroup (0, 0) leaves the row scales
  const float* absmax_in;
  int np;
  float* scale_out;
  // AQ: A in the PAIR-MAJOR layout the fused producers write (a_pairs), see aq_pair_offset; else row-major [M, lda]
  int a_pairs;
  // SILU epilogue (ksplit == 1): act T [M, N / 2] + absmax partials [M][strips], or e4m3 [M, N / 2] with *static_out_scale
  int silu;
  int act_pairs;          // act_out in the pair-major layout (for an AQ consumer) instead of row-major [M, N / 2]
  void* act_out;
  float* absmax_out;
  uint8_t* q8_out;
  const float* static_out_scale;
};

// ---- fp8(x / scale) with the IEEE fp32 quotient, per-row part hoisted -------------------------------------------------------
// Residual corrections after q0 = n r.  hipcc's own fp32 division is r = refine(rcp(s)); q0 = n r; q1 = q0 + r (n - s q0);
// q = q1 + r (n - s q1) (+ v_div_scale / v_div_fixup for exponent extremes and specials).  ONE correction already lands
// on the same fp8 byte for EVERY input this kernel can meet: the scale of a row is fl(a / 448) (floored at 1 / (448 * 512))
// for a 16-bit magnitude a that occurs in the row, and the row holds 16-bit values |x| <= a -- a finite domain of
// 32 768 x 65 536 pairs per dtype, all of which tests/test_fp8_diet_gpu.py::test_quantise_on_load_every_input_absmax_pair
// runs against fp8(x / scale) (itself pinned to the CPU oracle).  -DF8R_DIV_STEPS=2 builds the full sequence (the same test
// passes with it; it costs 1 more VALU per element: o_proj 6.6 -> 6.9 us, down 14.7 -> 15.4, tools/fp8_aq_lab.py).
#ifndef F8R_DIV_STEPS
#define F8R_DIV_STEPS 1
#endif
struct F8Rcp { float s, r; };
__device__ __forceinline__ F8Rcp f8r_make_rcp(float s) {
  const float r0 = __builtin_amdgcn_rcpf(s);
  const float e0 = __builtin_fmaf(-s, r0, 1.0f);
  return F8Rcp{s, __builtin_fmaf(e0, r0, r0)};
}
typedef f16 f16x8_t __attribute__((ext_vector_type(8)));
// 8 consecutive 16-bit values (one 16-byte load) -> 8 e4m3 bytes (lo: elements 0..3, hi: 4..7)
template <typename T>
__device__ __forceinline__ void f8r_quant8(const u32x4 h, const F8Rcp rc, uint32_t& lo, uint32_t& hi) {
  const f32x2 nd = {-rc.s, -rc.s}, rr = {rc.r, rc.r};
  f32x2 q[4];
  // (a bit-cast of ONE element of an ext-vector may read element 0 -- DESIGN 3: cast the whole vector)
  const f16x8_t hv = __builtin_bit_cast(f16x8_t, h);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x2 n;
    if constexpr (__is_same(T, Half)) {
      n = f32x2{__builtin_fabsf((float)hv[2 * j]), __builtin_fabsf((float)hv[2 * j + 1])};
    } else {
      n = f32x2{__builtin_bit_cast(float, (h[j] << 16) & 0x7fffffffu), __builtin_bit_cast(float, h[j] & 0x7fff0000u)};
    }
    const f32x2 q0 = n * rr;
    const f32x2 e1 = __builtin_elementwise_fma(nd, q0, n);
    const f32x2 q1 = __builtin_elementwise_fma(e1, rr, q0);
#if F8R_DIV_STEPS >= 2
    const f32x2 e2 = __builtin_elementwise_fma(nd, q1, n);
    q[j] = __builtin_elementwise_fma(e2, rr, q1);
#else
    q[j] = q1;
#endif
  }
  int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[0][0], q[0][1], 0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[1][0], q[1][1], w0, true);
  int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[2][0], q[2][1], 0, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[3][0], q[3][1], w1, true);
  // signs: the high byte of every 16-bit element, gathered by one v_perm_b32 per four
  const uint32_t s01 = __builtin_amdgcn_perm(h[1], h[0], 0x07050301u);
  const uint32_t s23 = __builtin_amdgcn_perm(h[3], h[2], 0x07050301u);
  lo = (s01 & 0x80808080u) | (uint32_t)w0;
  hi = (s23 & 0x80808080u) | (uint32_t)w1;
}

// Row scales from the producers' absmax partials: thread (row = t / 8, slice = t % 8) reads its share with 16-byte buffer
// loads (out-of-range pieces read as 0, the identity of max over magnitudes), the eight slices meet through ds_swizzle-free
// shuffles, slice 0 leaves scale[row] in LDS.  Call with all 256 threads; ISSUES its loads, returns the handle to finish.
struct F8AbsmaxLoads { f32x4 v[8]; };
__device__ __forceinline__ F8AbsmaxLoads f8r_absmax_issue(const float* absmax, int M, int np) {
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(absmax), 0,
      (uint32_t)((size_t)M * np * sizeof(float)), 0x00020000);
  const int row = threadIdx.x >> 3, sl = threadIdx.x & 7;
  F8AbsmaxLoads L;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int j = 4 * sl + 32 * it;
    const int off = (row < M && j < np) ? (row * np + j) * 4 : 0x7ffffff0;
    L.v[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, off, 0, 0));
  }
  return L;
}
__device__ __forceinline__ void f8r_absmax_finish(const F8AbsmaxLoads& L, float* sscale /* LDS [32] */) {
  float am = 0.f;
#pragma unroll
  for (int it = 0; it < 8; ++it)
#pragma unroll
    for (int e = 0; e < 4; ++e) am = __builtin_fmaxf(am, L.v[it][e]);
  am = __builtin_fmaxf(am, __shfl_xor(am, 1));
  am = __builtin_fmaxf(am, __shfl_xor(am, 2));
  am = __builtin_fmaxf(am, __shfl_xor(am, 4));
  if ((threadIdx.x & 7) == 0) sscale[threadIdx.x >> 3] = __builtin_fmaxf(am / 448.f, 1.0f / (448.f * 512.f));   // common.cu:205,239
}

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
// AQ: A is 16-bit and quantised on load (see the header).
// Kernel arguments as in wna16_gemm_stream_kernel: what the first loads need comes first and as scalars (preloaded into
// SGPRs, Makefile: -amdgpu-kernarg-preload-count); p_in carries the rest.
template <typename T, int MT, int NSEG, int NT, int D, int AQ>
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
  constexpr int AB = AQ ? 2 : 1;                                  // bytes per element of A; 16-byte loads per (k-pair, tile)
  extern __shared__ __attribute__((aligned(16))) float red[];     // [NWV][ROWS][CWP]
  __shared__ float sscale[32];                                    // AQ: the row scales; SILU epilogue: the strip's row absmax
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
      (AQ && p.a_pairs) ? (uint32_t)((size_t)(p.K >> 6) * ((p.M + 15) >> 4) * 2048)
                        : (uint32_t)(((size_t)(p.M - 1) * p.lda + p.K) * AB), 0x00020000);
  constexpr int WAVE_BYTES = NKP * NT * 1024;
  const int sbase = ((ky * S + strip) * NWV + wave) * WAVE_BYTES;
  const int voff_w = lane * 16;

  // AQ: the absmax partials are the first bytes asked for (they gate the first MFMA, the rings do not wait for them)
  F8AbsmaxLoads aml;
  f32x4 amd[MT][4];         // np <= 16: the partials of this lane's own rows
  if constexpr (AQ) {
    if (p.np <= 16) {
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.absmax_in), 0,
          (uint32_t)((size_t)p.M * p.np * sizeof(float)), 0x00020000);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = min(16 * i + c, p.M - 1);
          const int off = 4 * j < p.np ? (row * p.np + 4 * j) * 4 : 0x7ffffff0;      // out of range reads as 0
          amd[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, off, 0, 0));
        }
    } else {
      aml = f8r_absmax_issue(p.absmax_in, p.M, p.np);
    }
  }

  // ---- one stream per wave: the activations of k-pair kp (lane (g, c) = token 16 i + c, elements k + 16 g .. + 16: used ONCE,
  // every k-pair covers all the strip's columns) ride in the same register ring as its weights, DD k-pairs ahead.  (First version:
  // all of A loaded up front -- 32 gathers of 16 x 64 bytes in front of the first weight byte of every wave.)
  // A addressing: row-major -- lane (g, c) of tile i reads row 16 i + c at element k0 + 16 g (+ 8 for the second half);
  // pair-major (AQ only) -- the lane's piece of block (k-pair, tile i, half)
  const bool pairs = AQ && p.a_pairs;
  const int mtiles_a = (p.M + 15) >> 4;
  const int kp_stride = pairs ? mtiles_a * 2048 : 64 * AB, h_stride = pairs ? 1024 : 16;
  int voff_a[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
    voff_a[i] = pairs ? min(i, mtiles_a - 1) * 2048 + lane * 16 : (min(16 * i + c, p.M - 1) * p.lda + 16 * g) * AB;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // AQ: the activations run AHEAD of the weights (DA = DD + 2 k-pairs in flight against DD): loads return in order, so when a
  // wave waits for the weights of k-pair kp the 16-bit activations of kp + 1 (L2 hits, issued two iterations earlier than
  // those weights) have long landed -- their quantisation (VALU) is done in the shadow of the wait for W(kp) from HBM instead
  // of between that wait and the MFMAs (measured, tools/fp8_aq_lab.py: down 15.6 -> see profiles/r6_fp8_diet_lab.txt).
  constexpr int DA = AQ ? (DD + 2 < NKP ? DD + 2 : NKP) : DD;
  constexpr int RA = DA + 1;
  u32x4 wr[RING][NT], ar[RA][MT][AB];
  auto load_a = [&](auto KP_) {
    constexpr int kp = decltype(KP_)::value;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int vo = voff_a[i];
#pragma unroll
      for (int h = 0; h < AB; ++h)
        ar[kp % RA][i][h] = __builtin_amdgcn_raw_buffer_load_b128(ra, vo, (kp0 + kp) * kp_stride + h * h_stride, 0);
    }
  };
  auto load_w = [&](auto KP_) {
    constexpr int kp = decltype(KP_)::value;
#pragma unroll
    for (int t = 0; t < NT; ++t)
      wr[kp % RING][t] = __builtin_amdgcn_raw_buffer_load_b128(rw, voff_w, sbase + (kp * NT + t) * 1024, 2);
  };
  if constexpr (AQ) {       // every A load of the prologue in front of the first weight load
    f8r_static_for<0, DA>([&](auto KP_) { load_a(KP_); });
    f8r_static_for<0, DD>([&](auto KP_) { load_w(KP_); });
  } else {
    f8r_static_for<0, DD>([&](auto KP_) { load_a(KP_); load_w(KP_); });
  }
  __builtin_amdgcn_sched_barrier(0);

  F8Rcp rc[MT];
  u32x4 af[2][MT];          // AQ: the e4m3 fragments of k-pair kp (slot kp & 1), made one iteration ahead
  auto quant_kp = [&](auto KP_) {
    constexpr int kp = decltype(KP_)::value;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      uint32_t q0, q1, q2, q3;
      f8r_quant8<T>(ar[kp % RA][i][0], rc[i], q0, q1);
      f8r_quant8<T>(ar[kp % RA][i][AB - 1], rc[i], q2, q3);
      af[kp & 1][i] = u32x4{q0, q1, q2, q3};
    }
  };
  if constexpr (AQ) {
    if (p.np <= 16) {
      // few partials per row (attention: one per kv-head): every lane reduces the rows it quantises by itself -- no LDS
      // round trip, no workgroup barrier in front of the first MFMA
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        float am = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) am = __builtin_fmaxf(am, amd[i][j][e]);
        const float sc_row = __builtin_fmaxf(am / 448.f, 1.0f / (448.f * 512.f));
        rc[i] = f8r_make_rcp(sc_row);
        if (p.scale_out && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0 && g == 0 && 16 * i + c < p.M) p.scale_out[16 * i + c] = sc_row;
        if (!p.slab && wave == 0 && g == 0) sscale[16 * i + c] = sc_row;       // the scaled epilogue reads the row scales from LDS
      }
    } else {
      f8r_absmax_finish(aml, sscale);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < MT; ++i) rc[i] = f8r_make_rcp(sscale[min(16 * i + c, p.M - 1)]);
      if (p.scale_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < p.M) p.scale_out[threadIdx.x] = sscale[threadIdx.x];
    }
    quant_kp(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
  }

  f8r_static_for<0, NKP>([&](auto KP_) {
    constexpr int kp = decltype(KP_)::value;
    if constexpr (kp + DA < NKP) load_a(std::integral_constant<int, (kp + DA < NKP ? kp + DA : 0)>{});
    if constexpr (kp + DD < NKP) load_w(std::integral_constant<int, (kp + DD < NKP ? kp + DD : 0)>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (AQ) {
      if constexpr (kp + 1 < NKP) {
        quant_kp(std::integral_constant<int, (kp + 1 < NKP ? kp + 1 : 0)>{});
        __builtin_amdgcn_sched_barrier(0);         // the quantisation is issued BEFORE the wait for this k-pair's weights
      }
    } else {
#pragma unroll
      for (int i = 0; i < MT; ++i) af[kp & 1][i] = ar[kp % RA][i][0];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const u32x4 b = wr[kp % RING][t];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(f8r_lo(af[kp & 1][i]), f8r_lo(b), acc[i][t], 0, 0, 0);
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(f8r_hi(af[kp & 1][i]), f8r_hi(b), acc[i][t], 0, 0, 0);
      }
    }
  });

  // ---- K reduction over the waves: D[token 16 i + 4 g + r][column 16 t + c] ------------------------------------------------
  if (!AQ && p.silu && threadIdx.x < 32) sscale[threadIdx.x] = 0.f;     // (uniform per launch) the strip's running row absmax
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
      float sa;
      if constexpr (AQ) sa = sscale[row];
      else sa = p.a_scales ? p.a_scales[p.a_per_token ? row : 0] : 1.f;
      if (!AQ && p.silu) {
        // columns n .. n + 3 of the interleaved copy = (gate j0, up j0, gate j0 + 1, up j0 + 1); their scales sit at the
        // CHECKPOINT's rows j and N / 2 + j.  cutlass_scaled_mm's value T(sa * (sb * acc)), then silu_and_mul's roundings
        const int I = p.N >> 1, j0 = n >> 1;
        float gu[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int src = (e & 1) * I + j0 + (e >> 1);
          const float sb = p.b_scales ? p.b_scales[p.b_per_channel ? src : 0] : 1.f;
          gu[e] = T::to_f32(from_f32_exact<T>(sa * (sb * sum[e])));
        }
        const uint16_t a0 = silu_mul_bits<T>(gu[0], gu[1]), a1 = silu_mul_bits<T>(gu[2], gu[3]);
        if (p.q8_out) {               // static scheme: x * (1 / scale), static_scaled_fp8_quant (fq_pack4_inv's bits)
          const float inv = 1.0f / *p.static_out_scale;
          const float q0 = __builtin_fmaxf(-448.f, __builtin_fminf(T::to_f32(a0) * inv, 448.f));
          const float q1 = __builtin_fmaxf(-448.f, __builtin_fminf(T::to_f32(a1) * inv, 448.f));
          *reinterpret_cast<uint16_t*>(p.q8_out + (size_t)row * I + j0) =
              (uint16_t)(__builtin_amdgcn_cvt_pk_fp8_f32(q0, q1, 0, false) & 0xffff);
        } else {
          uint16_t* dst = (uint16_t*)p.act_out + (p.act_pairs ? aq_pair_offset(row, j0, (p.M + 15) >> 4) : (size_t)row * I + j0);
          *reinterpret_cast<uint32_t*>(dst) = (uint32_t)a0 | ((uint32_t)a1 << 16);
          const float am = __builtin_fmaxf(__builtin_fabsf(T::to_f32(a0)), __builtin_fabsf(T::to_f32(a1)));
          atomicMax(reinterpret_cast<unsigned*>(&sscale[row]), __builtin_bit_cast(unsigned, am));   // magnitudes: uint order
        }
        continue;
      }
      uint16_t o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sb = p.b_scales ? p.b_scales[p.b_per_channel ? n + e : 0] : 1.f;
        float v = sa * (sb * sum[e]);                // order of test_cutlass.py:43
        if (p.bias) v += T::to_f32(((const typename T::storage*)p.bias)[n + e]);
        o[e] = from_f32_exact<T>(v);
      }
      *reinterpret_cast<u32x2*>((typename T::storage*)p.c + (size_t)row * p.N + n) =
          u32x2{(uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16)};
    }
  }
  if (!AQ && p.silu && p.absmax_out) {
    __syncthreads();
    if (threadIdx.x < p.M) p.absmax_out[(size_t)threadIdx.x * p.strips + strip] = sscale[threadIdx.x];
  }
}

// The quantiser of the AQ form on its own (tests: every 16-bit input x adversarial scales against x / scale; also the op-level
// "two producers' partials -> rows" check): q[M, K] = fp8(x[M, K] / scale(row)), scale from absmax partials [M][np].
template <typename T>
__global__ void fp8_quant_rows_aq_kernel(const u32x4* __restrict__ x, const float* __restrict__ absmax, int np, uint8_t* __restrict__ q,
                                         float* __restrict__ scale_out, int M, int K) {
  __shared__ float sscale[32];
  const int row0 = blockIdx.x * 32;
  const int rows = min(32, M - row0);
  const F8AbsmaxLoads aml = f8r_absmax_issue(absmax + (size_t)row0 * np, rows, np);
  f8r_absmax_finish(aml, sscale);
  __syncthreads();
  if (threadIdx.x < rows && scale_out) scale_out[row0 + threadIdx.x] = sscale[threadIdx.x];
  for (int r = 0; r < rows; ++r) {
    const F8Rcp rc = f8r_make_rcp(sscale[r]);
    for (int v = threadIdx.x; v < K / 8; v += blockDim.x) {
      uint32_t lo, hi;
      f8r_quant8<T>(x[((size_t)(row0 + r) * K) / 8 + v], rc, lo, hi);
      *reinterpret_cast<u32x2*>(q + (size_t)(row0 + r) * K + 8 * v) = u32x2{lo, hi};
    }
  }
}

// [N, K] row-major e4m3 -> strip-major: one thread per 16-byte piece.
// interleave: strip column n = (gate n / 2 | up n / 2) of a [gate; up] matrix -- checkpoint row (n & 1) * N / 2 + n / 2.
__global__ void fp8_strip_relayout_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, int N, int K, int nseg, int nt,
                                          int ksplit, int interleave) {
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
  int n = strip * 16 * nt + 16 * t + c;
  if (interleave) n = (n & 1) * (N >> 1) + (n >> 1);
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
static int f8r_relayout(const void* w, void* out, int64_t M, int64_t N, int64_t K, int interleave, void* stream) {
  const Fp8ResConfig cf = f8r_plan(M, N, K);
  APHRO_CHECK(cf.nseg != 0 && w != out, "fp8_strip_relayout: M=%ld N=%ld K=%ld is not served by the resident kernel", (long)M, (long)N, (long)K);
  APHRO_CHECK(((uintptr_t)w % 16) == 0 && ((uintptr_t)out % 16) == 0, "fp8_strip_relayout: 16-byte alignment required");
  APHRO_CHECK(!interleave || (cf.ksplit == 1 && N % 2 == 0), "fp8_strip_relayout_interleaved: the SiluAndMul epilogue needs a plan with one K slice");
  const int64_t total = N * K / 16;
  hipLaunchKernelGGL(fp8_strip_relayout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)w, (u32x4*)out, (int)N, (int)K, cf.nseg, cf.nt, cf.ksplit, interleave);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
extern "C" int aphro_fp8_strip_relayout(const void* w, void* out, int64_t M, int64_t N, int64_t K, void* stream) {
  return f8r_relayout(w, out, M, N, K, 0, stream);
}
// The same for a [gate; up] matrix whose SiluAndMul runs in the GEMM epilogue: column 2 j of the strip order = gate row j,
// column 2 j + 1 = up row j (N / 2 + j).
extern "C" int aphro_fp8_strip_relayout_interleaved(const void* w, void* out, int64_t M, int64_t N, int64_t K, void* stream) {
  return f8r_relayout(w, out, M, N, K, 1, stream);
}
// Column strips of the plan (= absmax partials per row the SiluAndMul epilogue writes); 0: shape not served.
extern "C" int aphro_fp8_gemm_resident_strips(int64_t M, int64_t N, int64_t K) {
  const Fp8ResConfig cf = f8r_plan(M, N, K);
  return cf.nseg == 0 ? 0 : (int)(N / (16 * cf.nt));
}

static int f8r_launch(Fp8ResParams p, const Fp8ResConfig cf, int aq, int dtype, void* stream) {
  const int64_t M = p.M, N = p.N;
  const int mt = M > 16 ? 2 : 1;
  const dim3 grid((unsigned)(N / (16 * cf.nt)), (unsigned)cf.ksplit);
  p.ksplit = cf.ksplit;
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
  // AQ: a k-pair is MT x 2 + NT sixteen-byte loads and a wave can have 63 counted loads in flight: 6 k-pairs of the narrow strips
#ifndef F8R_DEPTH_AQ
#define F8R_DEPTH_AQ(nt) ((nt) >= 6 ? 4 : 6)
#endif
#define L(TT, MTV, NSEGV, NTV, AQV)                                                                                     \
  {                                                                                                                     \
    auto kern = fp8_gemm_resident_kernel<TT, MTV, NSEGV, NTV, (AQV ? F8R_DEPTH_AQ(NTV) : F8R_DEPTH(NTV)), AQV>;         \
    const size_t lds = (size_t)4 * 16 * MTV * (16 * NTV + 4) * sizeof(float);                                           \
    if (lds > 64 * 1024 - 256 &&                                                                                        \
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {  \
      set_error("fp8_gemm_resident: cannot raise the dynamic LDS limit to %zu", lds);                                   \
      return APHRO_ERR_LAUNCH;                                                                                          \
    }                                                                                                                   \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, p.w, p.a, p.strips, p.xcd_shift, p.strips_per_xcd,  \
                       p.ksplit, p.M, p.N, p.K, p.lda, p);                                                             \
  }
#define LA(TT, MTV, a_, b_) { if (aq) L(TT, MTV, a_, b_, 1) else L(TT, MTV, a_, b_, 0) }
#define X(a_, b_)                                                  \
  if (cf.nseg == a_ && cf.nt == b_) {                              \
    if (dtype == APHRO_F16) { if (mt == 2) LA(Half, 2, a_, b_) else LA(Half, 1, a_, b_) }       \
    else { if (mt == 2) LA(BFloat, 2, a_, b_) else LA(BFloat, 1, a_, b_) }                      \
  }
  F8R_CONFIGS(X)
#undef X
#undef LA
#undef L
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

static Fp8ResParams f8r_params() {
  Fp8ResParams p;
  memset(&p, 0, sizeof(p));
  return p;
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
  Fp8ResParams p = f8r_params();
  p.a = (const uint8_t*)a; p.w = (const uint8_t*)w_strip; p.a_scales = a_scales; p.b_scales = b_scales; p.bias = bias;
  p.c = out; p.slab = slabs; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = (int)lda;
  p.a_per_token = a_scale_per_token; p.b_per_channel = b_scale_per_channel;
  return f8r_launch(p, cf, 0, dtype, stream);
}

// The AQ form: a16 = the producer's 16-bit activations -- row-major [M, lda] (lda in elements) or, a_pairs, the pair-major
// layout of aq_pair_offset ([K / 64][ceil(M / 16)][2][64 lanes][8]) --, absmax = its partials [M][np]; the row
// scales max(max_p absmax[m][p] / 448, 1 / (448 * 512)) are worked out in the launch, left in scale_out[M] (for the consumer
// of the slabs) and the A fragments quantised on load: the bits of dynamic_per_token_scaled_fp8_quant -> the plain kernel.
extern "C" int aphro_fp8_gemm_resident_aq(const void* a16, int64_t lda, int a_pairs, const float* absmax, int np, const void* w_strip,
                                          float* scale_out, const float* b_scales, const void* bias, void* out, float* slabs,
                                          size_t slabs_bytes, int64_t M, int64_t N, int64_t K, int b_scale_per_channel,
                                          int dtype, void* stream) {
  const Fp8ResConfig cf = f8r_plan(M, N, K);
  APHRO_CHECK(cf.nseg != 0, "fp8_gemm_resident_aq: M=%ld N=%ld K=%ld is not served", (long)M, (long)N, (long)K);
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "fp8_gemm_resident_aq: dtype must be f16 or bf16");
  APHRO_CHECK((out != nullptr) != (slabs != nullptr), "fp8_gemm_resident_aq: exactly one of out / slabs");
  APHRO_CHECK(out == nullptr || cf.ksplit == 1, "fp8_gemm_resident_aq: this shape is K-sliced (%d): slabs only", cf.ksplit);
  APHRO_CHECK(slabs == nullptr || slabs_bytes >= (size_t)cf.ksplit * M * N * sizeof(float), "fp8_gemm_resident_aq: slabs too small");
  APHRO_CHECK(absmax != nullptr && np >= 4 && np <= 256 && np % 4 == 0 && ((uintptr_t)absmax % 16) == 0,
              "fp8_gemm_resident_aq: absmax partials [M][np] with np %% 4 == 0, 4 <= np <= 256, 16-byte aligned (np=%d)", np);
  APHRO_CHECK(a_pairs == 0 || K % 64 == 0, "fp8_gemm_resident_aq: the pair-major layout needs K %% 64 == 0");
  APHRO_CHECK(((uintptr_t)a16 % 16) == 0 && ((uintptr_t)w_strip % 16) == 0 && (a_pairs || (lda % 8 == 0 && lda >= K)) &&
              (size_t)(M + 15) * (a_pairs ? K : lda) * 2 < 0x7fffffffull &&
              (out == nullptr || ((uintptr_t)out % 8) == 0) && (slabs == nullptr || ((uintptr_t)slabs % 16) == 0),
              "fp8_gemm_resident_aq: alignment (16-byte rows of a, lda %% 8 == 0)");
  Fp8ResParams p = f8r_params();
  p.a = (const uint8_t*)a16; p.w = (const uint8_t*)w_strip; p.b_scales = b_scales; p.bias = bias;
  p.c = out; p.slab = slabs; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = (int)lda;
  p.a_per_token = 1; p.b_per_channel = b_scale_per_channel;
  p.absmax_in = absmax; p.np = np; p.scale_out = scale_out; p.a_pairs = a_pairs;
  return f8r_launch(p, cf, 1, dtype, stream);
}

// gate_up with SiluAndMul in the epilogue (one K slice; w_strip_il: aphro_fp8_strip_relayout_interleaved): dynamic scheme --
// act_out T [M, N / 2] + absmax_out [M][strips] (strips: aphro_fp8_gemm_resident_strips); static scheme -- q8_out e4m3
// [M, N / 2] = fp8(act * (1 / *static_out_scale)).  The bits of the plain kernel -> silu_and_mul[_quant_fp8].
extern "C" int aphro_fp8_gemm_resident_silu(const void* a, int64_t lda, const void* w_strip_il, const float* a_scales,
                                            const float* b_scales, void* act_out, int act_pairs, float* absmax_out, void* q8_out,
                                            const float* static_out_scale, int64_t M, int64_t N, int64_t K,
                                            int a_scale_per_token, int b_scale_per_channel, int dtype, void* stream) {
  const Fp8ResConfig cf = f8r_plan(M, N, K);
  APHRO_CHECK(cf.nseg != 0 && cf.ksplit == 1 && N % 2 == 0, "fp8_gemm_resident_silu: M=%ld N=%ld K=%ld is not served with one K slice", (long)M, (long)N, (long)K);
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "fp8_gemm_resident_silu: dtype must be f16 or bf16");
  APHRO_CHECK((q8_out != nullptr) != (act_out != nullptr && absmax_out != nullptr),
              "fp8_gemm_resident_silu: either (act_out, absmax_out) or q8_out");
  APHRO_CHECK(q8_out == nullptr || static_out_scale != nullptr, "fp8_gemm_resident_silu: q8_out needs static_out_scale");
  APHRO_CHECK(((uintptr_t)a % 16) == 0 && ((uintptr_t)w_strip_il % 16) == 0 && lda % 16 == 0 && lda >= K &&
              (act_out == nullptr || ((uintptr_t)act_out % 4) == 0) && (q8_out == nullptr || ((uintptr_t)q8_out % 2) == 0),
              "fp8_gemm_resident_silu: alignment");
  Fp8ResParams p = f8r_params();
  p.a = (const uint8_t*)a; p.w = (const uint8_t*)w_strip_il; p.a_scales = a_scales; p.b_scales = b_scales;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = (int)lda;
  p.a_per_token = a_scale_per_token; p.b_per_channel = b_scale_per_channel;
  p.silu = 1; p.act_out = act_out; p.act_pairs = act_pairs; p.absmax_out = absmax_out; p.q8_out = (uint8_t*)q8_out; p.static_out_scale = static_out_scale;
  return f8r_launch(p, cf, 0, dtype, stream);
}

// q[M, K] = fp8(x[M, K] / scale(row)), scale(row) = max(max_p absmax[row][p] / 448, 1 / (448 * 512)) -> scale_out[M]: the AQ
// form's quantiser alone (same device functions), for the exhaustive parity tests.
extern "C" int aphro_fp8_quant_rows_aq(const void* x, const float* absmax, int np, void* q, float* scale_out, int64_t M,
                                       int64_t K, int dtype, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "fp8_quant_rows_aq: dtype must be f16 or bf16");
  APHRO_CHECK(K % 8 == 0 && np >= 4 && np <= 256 && np % 4 == 0 && ((uintptr_t)absmax % 16) == 0 && ((uintptr_t)x % 16) == 0 &&
              ((uintptr_t)q % 8) == 0, "fp8_quant_rows_aq: K %% 8 == 0, np %% 4 == 0 (<= 256), 16-byte aligned inputs");
  if (M == 0) return APHRO_OK;
  const dim3 grid((unsigned)((M + 31) / 32));
  if (dtype == APHRO_F16)
    hipLaunchKernelGGL(fp8_quant_rows_aq_kernel<Half>, grid, dim3(256), 0, (hipStream_t)stream, (const u32x4*)x, absmax, np,
                       (uint8_t*)q, scale_out, (int)M, (int)K);
  else
    hipLaunchKernelGGL(fp8_quant_rows_aq_kernel<BFloat>, grid, dim3(256), 0, (hipStream_t)stream, (const u32x4*)x, absmax, np,
                       (uint8_t*)q, scale_out, (int)M, (int)K);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
