// GPTQ 2 / 3 / 8-bit weights on gfx950 (the reference's gemm_half_q_half_gptq_{2,3,8}bit_kernel and
// reconstruct_{gptq,exllama}_{2,3,8}bit_kernel, kernels/quantization/gptq/q_gemm.cu:329-700, 759-1182, 1394-1505;
// qdq_{2,3,8}.cuh).  The 4-bit path is the hot one (wna16_gemm*.hip); these widths are served for completeness of the
// `_C::gptq_gemm` / `gptq_shuffle` contract with two kernels:
//
//   * gptq_dequant_bits_kernel: fp16 / bf16 [K, N] = (q - (z + 1)) * s, one rounding of the exact fp32 product -- bit for
//     bit what reconstruct_gptq_kernel returns (pinned through the oracle by the reference's own kernels run on the host,
//     tests/golden/gptq_ref_bits.npz).  Serves the non-exllama path (g_idx = row -> group) and M above the small-M kernel.
//   * wnx_gemm_kernel: M <= 32 rows, MFMA 16x16x32 on the INTEGER (q - z - 1) as exact f16 / bf16 values, one fp32
//     accumulator per quantisation group, c += s * acc at the group's end -- exact products, fp32 sums (tighter than the
//     reference's fp16 dot products).  Workgroup = 32 columns x 4 K slices (8 waves), K reduced through LDS: no workspace.
//
// Layout: the checkpoint's.  Values (and zero points, along N) are laid end to end, little-endian, in uint32 words; 32
// values occupy `bits` words; 3-bit values 10 and 21 straddle a word boundary.  The "exllama" state of these widths is the
// same words with act-order rows made sequential (aphro_gptq_make_sequential_bits): the reference's shuffle_{2,3,8}bit_kernel
// reorders the fields inside a word for ITS dequant routines, and the layout after gptq_shuffle is private to the
// kernels that consume it.
#include "common.h"

namespace aphro {

// the 8 * BITS-bit field of the 8 values k0 + 8 j .. + 7 (j = 0 .. 3 inside a 32-value unit) of one column
template <int BITS>
__device__ __forceinline__ uint64_t wnx_field(const uint32_t* __restrict__ col, int64_t row_stride, int unit, int j,
                                              int last_row) {
  const int off = 8 * j * BITS;                     // bit offset inside the unit
  const int lo = unit * BITS + (off >> 5);
  const int hi = min(lo + 1, last_row);             // (only read past the field's own words when they are not needed)
  const uint64_t f = (uint64_t)col[(int64_t)lo * row_stride] | ((uint64_t)col[(int64_t)hi * row_stride] << 32);
  return f >> (off & 31);
}

// zero point (stored value, without the + 1) of (group, column): bitstring along N
template <int BITS>
__device__ __forceinline__ int wnx_zero(const uint32_t* __restrict__ zrow, int n) {
  const int off = n * BITS;
  const int lo = off >> 5, sh = off & 31;
  uint32_t v = zrow[lo] >> sh;
  if (sh + BITS > 32) v |= zrow[lo + 1] << (32 - sh);
  return (int)(v & ((1u << BITS) - 1u));
}

template <typename T, int BITS>
__global__ void gptq_dequant_bits_kernel(const uint32_t* __restrict__ qw, const uint32_t* __restrict__ qz,
                                         const uint16_t* __restrict__ sc, const int32_t* __restrict__ g_idx,
                                         uint16_t* __restrict__ out, int K, int N, int group_size) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int unit = blockIdx.y;                      // 32 consecutive k
  if (n >= N) return;
  const int zwords = N * BITS / 32;
  const int last_row = K * BITS / 32 - 1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint64_t f = wnx_field<BITS>(qw + n, N, unit, j, last_row);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = unit * 32 + 8 * j + e;
      const int grp = g_idx ? g_idx[k] : k / group_size;
      const int z = wnx_zero<BITS>(qz + (int64_t)grp * zwords, n) + 1;
      const int q = (int)((f >> (e * BITS)) & ((1u << BITS) - 1u));
      const float w = (float)(q - z) * T::to_f32(sc[(int64_t)grp * N + n]);
      out[(int64_t)k * N + n] = from_f32_exact<T>(w);
    }
  }
}

template <typename T>
__device__ __forceinline__ f32x4 wnx_mfma(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (__is_same(T, Half))
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// grid.x = N / 32.  8 waves: wave = 2 * kslice + half; half picks 16 of the workgroup's 32 columns.
template <typename T, int BITS, int MT>
__global__ __launch_bounds__(512) void wnx_gemm_kernel(const uint16_t* __restrict__ a, int lda,
                                                       const uint32_t* __restrict__ qw, const uint32_t* __restrict__ qz,
                                                       const uint16_t* __restrict__ sc, uint16_t* __restrict__ c, int M, int N,
                                                       int K, int group_size) {
  __shared__ float red[8][16 * MT][17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, cc = lane & 15;
  const int half = wave & 1, ks = wave >> 1;
  const int n = blockIdx.x * 32 + 16 * half + cc;   // this lane's column (B operand: column cc of the tile)
  const int units = K / 32;
  const int gu = group_size / 32;                   // units per group
  // K slices on group boundaries
  const int groups = units / gu;
  const int g0 = groups * ks / 4, g1 = groups * (ks + 1) / 4;
  const int zwords = N * BITS / 32;
  const int last_row = K * BITS / 32 - 1;
  const uint32_t* col = qw + n;
  f32x4 cacc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) cacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint16_t* arow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) arow[i] = a + (int64_t)min(16 * i + cc, M - 1) * lda + 8 * g;
  for (int grp = g0; grp < g1; ++grp) {
    const int z = wnx_zero<BITS>(qz + (int64_t)grp * zwords, n) + 1;
    const float s = T::to_f32(sc[(int64_t)grp * N + n]);
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int u = grp * gu; u < (grp + 1) * gu; ++u) {
      const uint64_t f = wnx_field<BITS>(col, N, u, g, last_row);
      uint16_t b16[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) b16[e] = T::from_f32((float)((int)((f >> (e * BITS)) & ((1u << BITS) - 1u)) - z));   // exact
      const u32x4 b = {(uint32_t)b16[0] | ((uint32_t)b16[1] << 16), (uint32_t)b16[2] | ((uint32_t)b16[3] << 16),
                       (uint32_t)b16[4] | ((uint32_t)b16[5] << 16), (uint32_t)b16[6] | ((uint32_t)b16[7] << 16)};
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const u32x4 av = *reinterpret_cast<const u32x4*>(arow[i] + 32 * u);
        acc[i] = wnx_mfma<T>(av, b, acc[i]);
      }
    }
    // D[row 4 g + r][column cc]: the lane's own column -> its scale
#pragma unroll
    for (int i = 0; i < MT; ++i) cacc[i] += acc[i] * s;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][16 * i + 4 * g + r][cc] = cacc[i][r];
  __syncthreads();
  for (int idx = threadIdx.x; idx < 16 * MT * 32; idx += 512) {
    const int row = idx / 32, col32 = idx % 32;
    const int h = col32 >> 4, c16 = col32 & 15;
    float sum = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) sum += red[2 * k4 + h][row][c16];
    if (row < M) c[(int64_t)row * N + blockIdx.x * 32 + col32] = T::from_f32(sum);
  }
}

// act-order rows made sequential: new row k takes the values of source row perm[k] (make_sequential_{2,3,8}bit_kernel's
// role, q_gemm.cu:1659-1820).  One thread per (32-value unit, column) of the OUTPUT.
template <int BITS>
__global__ void wnx_make_sequential_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                           const int32_t* __restrict__ perm, int K, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int unit = blockIdx.y;
  if (n >= N) return;
  uint32_t w[BITS];
#pragma unroll
  for (int i = 0; i < BITS; ++i) w[i] = 0;
  for (int e = 0; e < 32; ++e) {
    const int src = perm[unit * 32 + e];
    const int soff = (src & 31) * BITS;
    const int srow = (src >> 5) * BITS + (soff >> 5), ssh = soff & 31;
    uint32_t v = in[(int64_t)srow * N + n] >> ssh;
    if (ssh + BITS > 32) v |= in[(int64_t)(srow + 1) * N + n] << (32 - ssh);
    v &= (1u << BITS) - 1u;
    const int doff = e * BITS;
    const int dw = doff >> 5, dsh = doff & 31;
    // (static indexing: BITS <= 8 words)
#pragma unroll
    for (int i = 0; i < BITS; ++i) {
      if (i == dw) w[i] |= v << dsh;
      if (i == dw + 1 && dsh + BITS > 32) w[i] |= v >> (32 - dsh);
    }
  }
#pragma unroll
  for (int i = 0; i < BITS; ++i) out[(int64_t)(unit * BITS + i) * N + n] = w[i];
}

}  // namespace aphro

using namespace aphro;

static bool wnx_bits_ok(int bits) { return bits == 2 || bits == 3 || bits == 8; }

// fp16 / bf16 [K, N] from a 2 / 3 / 8-bit GPTQ matrix: g_idx (row -> group) or NULL (k / group_size).
extern "C" int aphro_gptq_dequant_bits(const uint32_t* q_weight, const uint32_t* qzeros, const void* scales,
                                       const int32_t* g_idx, void* out, int64_t K, int64_t N, int64_t groups, int bits,
                                       int dtype, void* stream) {
  APHRO_CHECK(wnx_bits_ok(bits), "gptq_dequant_bits: bits must be 2, 3 or 8 (got %d)", bits);
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "gptq_dequant_bits: dtype must be f16 or bf16");
  APHRO_CHECK(K % 32 == 0 && N % 32 == 0 && groups > 0 && K % groups == 0, "gptq_dequant_bits: K and N must be multiples of 32 (K=%ld N=%ld groups=%ld)",
              (long)K, (long)N, (long)groups);
  if (K == 0 || N == 0) return APHRO_OK;
  dim3 grid((unsigned)((N + 127) / 128), (unsigned)(K / 32));
#define L(TT, BB)                                                                                                  \
  hipLaunchKernelGGL((gptq_dequant_bits_kernel<TT, BB>), grid, dim3(128), 0, (hipStream_t)stream, q_weight, qzeros, \
                     (const uint16_t*)scales, g_idx, (uint16_t*)out, (int)K, (int)N, (int)(K / groups))
#define LB(TT) { if (bits == 2) L(TT, 2); else if (bits == 3) L(TT, 3); else L(TT, 8); }
  if (dtype == APHRO_F16) LB(Half) else LB(BFloat)
#undef LB
#undef L
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

// 1: aphro_gptq_gemm_bits serves the call (M <= 32, groups of a multiple of 32 rows, at least 4 groups).
extern "C" int aphro_gptq_gemm_bits_supported(int64_t M, int64_t N, int64_t K, int64_t groups, int bits) {
  if (!wnx_bits_ok(bits) || M < 1 || M > 32 || groups < 4 || K % groups != 0) return 0;
  const int64_t gs = K / groups;
  return gs % 32 == 0 && N % 32 == 0 && K % 32 == 0;
}

// c[M, N] = a[M, K] x W for the sequential (post-gptq_shuffle) 2 / 3 / 8-bit layout, M <= 32.  Act-order: the caller
// passes a[:, perm] (q_gemm.cu:219-226 gathers the same way).
extern "C" int aphro_gptq_gemm_bits(const void* a, int64_t lda, const uint32_t* q_weight, const uint32_t* qzeros,
                                    const void* scales, void* c, int64_t M, int64_t N, int64_t K, int64_t groups, int bits,
                                    int dtype, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "gptq_gemm_bits: dtype must be f16 or bf16");
  APHRO_CHECK(aphro_gptq_gemm_bits_supported(M, N, K, groups, bits), "gptq_gemm_bits: M=%ld N=%ld K=%ld groups=%ld bits=%d is not served",
              (long)M, (long)N, (long)K, (long)groups, bits);
  APHRO_CHECK(((uintptr_t)a % 16) == 0 && lda % 8 == 0 && lda >= K, "gptq_gemm_bits: a must be 16-byte aligned with lda %% 8 == 0");
  const int mt = M > 16 ? 2 : 1;
  dim3 grid((unsigned)(N / 32));
#define L(TT, BB, MTV)                                                                                             \
  hipLaunchKernelGGL((wnx_gemm_kernel<TT, BB, MTV>), grid, dim3(512), 0, (hipStream_t)stream, (const uint16_t*)a,   \
                     (int)lda, q_weight, qzeros, (const uint16_t*)scales, (uint16_t*)c, (int)M, (int)N, (int)K,    \
                     (int)(K / groups))
#define LM(TT, BB) { if (mt == 2) L(TT, BB, 2); else L(TT, BB, 1); }
#define LB(TT) { if (bits == 2) LM(TT, 2) else if (bits == 3) LM(TT, 3) else LM(TT, 8) }
  if (dtype == APHRO_F16) LB(Half) else LB(BFloat)
#undef LB
#undef LM
#undef L
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

// ops.gptq_shuffle for 2 / 3 / 8 bits: rows made sequential through perm (argsort of g_idx); out != q_weight.
extern "C" int aphro_gptq_make_sequential_bits(const uint32_t* q_weight, uint32_t* out, const int32_t* perm, int64_t K,
                                               int64_t N, int bits, void* stream) {
  APHRO_CHECK(wnx_bits_ok(bits) && K % 32 == 0 && perm != nullptr && q_weight != out, "gptq_make_sequential_bits: bad arguments");
  if (K == 0 || N == 0) return APHRO_OK;
  dim3 grid((unsigned)((N + 127) / 128), (unsigned)(K / 32));
#define L(BB) hipLaunchKernelGGL((wnx_make_sequential_kernel<BB>), grid, dim3(128), 0, (hipStream_t)stream, q_weight, out, perm, (int)K, (int)N)
  if (bits == 2) L(2); else if (bits == 3) L(3); else L(8);
#undef L
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
