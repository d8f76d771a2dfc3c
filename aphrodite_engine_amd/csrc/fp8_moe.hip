// Grouped FP8 W8A8 GEMM for mixture-of-experts layers on gfx950: the role of the reference's Triton fused_moe_kernel with
// use_fp8_w8a8 (aphrodite/modeling/layers/fused_moe/fused_moe.py:20-170, invoked twice by fused_experts :566-690 for
// Fp8MoEMethod.apply, quantization/fp8.py:468-503): per 16-row block of the expert-sorted slot list,
//     C[slot, :] = round_T( ((A_q[slot / top_k, :] . W_q[expert]^T) * w_routed[slot]) * a_scale * b_scale[expert] )
// with per-tensor activation scale, one weight scale per expert, the routed weight only in the second GEMM -- the same
// order of the fp32 multiplications as the reference kernel (:150-163).
//
// HBM-bound on the active experts' weights: a workgroup owns 16 NT weight rows (output columns) of ONE expert for one
// block of slots; its 8 waves split K and every lane streams 32 contiguous bytes of a weight row per 128-k step (the
// fp8 MFMA B fragments as they lie), exactly as fp8_gemm.hip's generic kernel; the A rows are gathered through
// sorted_token_ids, the output rows scattered to the slots.  Blocks past num_tokens_post_padded return at once
// (the grid is sized for the worst case, fused_moe.py:174-228).
#include "common.h"

namespace aphro {

struct Fp8MoeParams {
  const uint8_t* a;            // e4m3 [rows, K]: hidden states (top_k_div = top_k) or the intermediate (top_k_div = 1)
  const uint8_t* w;            // e4m3 [E, N, K]
  const float* a_scale;        // [1]
  const float* b_scales;       // [E]
  const float* topk_weights;   // [num_valid] or NULL (MUL_ROUTED_WEIGHT)
  const int32_t* sorted_ids;   // [max padded]: slot index, >= num_valid = padding
  const int32_t* expert_ids;   // [max blocks]
  const int32_t* num_post_pad; // [1]
  void* c;                     // T [num_valid, N]
  int N, K, num_valid, top_k_div;
};

// MOE8_NW waves split K.  4, not 8: a wave's two-step pipeline needs a few steps to reach its steady state (K = 4096 over 8
// waves is 4 steps each) and the LDS reduce is half as long -- Mixtral-8x7B shapes, 8 active experts, tools/fp8_moe_bench.py:
// w13 196 -> 191 us, w2 112 -> 98 us (4.2 -> 4.8 TB/s)
template <typename T, int NT, int MOE8_NW>
__global__ __launch_bounds__(MOE8_NW * 64) void fp8_moe_gemm_kernel(Fp8MoeParams p) {
  __shared__ __attribute__((aligned(16))) float red[MOE8_NW * NT * 64 * 4];
  const int blk = blockIdx.z;
  if (blk * 16 >= *p.num_post_pad) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4;
  const int c = lane & 15;
  const int expert = p.expert_ids[blk];
  const int ncol = blockIdx.x * (16 * NT) + NT * c;

  const int total = p.K >> 7;
  const int per_wave = (total + MOE8_NW - 1) / MOE8_NW;
  const int s0 = wave * per_wave;
  const int s1 = min(total, s0 + per_wave);

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const uint8_t* wbase = p.w + (size_t)expert * p.N * p.K;
  const uint8_t* wrow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wrow[t] = wbase + (size_t)(ncol + t) * p.K + 32 * g;
  // A operand: lane (g, c) = row c of the block (a padding slot reads row 0: its result is never stored)
  const int sid = p.sorted_ids[blk * 16 + c];
  const uint8_t* arow = p.a + (size_t)(sid < p.num_valid ? sid / p.top_k_div : 0) * p.K + 32 * g;

  // two 128-k steps of weights + activations in flight per wave (the loads of step s + 1 are issued before the MFMAs of
  // step s): with 8 waves x up to 4 workgroups per CU that is what keeps the weight stream going
  u32x4 wq[2][NT][2], aq[2][2];
  auto load_step = [&](int s, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const u32x4* wp = reinterpret_cast<const u32x4*>(wrow[t] + (size_t)s * 128);
      wq[buf][t][0] = __builtin_nontemporal_load(wp);
      wq[buf][t][1] = __builtin_nontemporal_load(wp + 1);
    }
    const u32x4* ap = reinterpret_cast<const u32x4*>(arow + (size_t)s * 128);
    aq[buf][0] = ap[0];
    aq[buf][1] = ap[1];
  };
  auto compute_step = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4& av = aq[buf][j >> 1];
      const long a = (long)(((uint64_t)av[2 * (j & 1) + 1] << 32) | av[2 * (j & 1)]);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const long b = (long)(((uint64_t)wq[buf][t][j >> 1][2 * (j & 1) + 1] << 32) | wq[buf][t][j >> 1][2 * (j & 1)]);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, acc[t], 0, 0, 0);
      }
    }
  };
  if (s0 < s1) load_step(s0, 0);
  int s = s0;
  for (; s + 2 <= s1 - 1; s += 2) {       // (unrolled by two so that the buffer index is a constant)
    load_step(s + 1, 1);
    compute_step(0);
    load_step(s + 2, 0);
    compute_step(1);
  }
  if (s + 1 < s1) {
    load_step(s + 1, 1);
    compute_step(0);
    compute_step(1);
  } else if (s < s1) {
    compute_step(0);
  }

#pragma unroll
  for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(&red[((wave * NT + t) * 64 + lane) * 4]) = acc[t];
  __syncthreads();
  // D[row 4 g + r][column c of n-tile t]: wave r (< 4) finishes row 4 g + r of every lane group
  if (wave < 4) {
    const int r = wave;
    const int slot = p.sorted_ids[blk * 16 + 4 * g + r];
    if (slot < p.num_valid) {
      const float sa = p.a_scale[0], sb = p.b_scales[expert];
      const float wr = p.topk_weights ? p.topk_weights[slot] : 1.f;
      typename T::storage* cp = (typename T::storage*)p.c + (size_t)slot * p.N + ncol;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float sum = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < MOE8_NW; ++w2) sum += red[((w2 * NT + t) * 64 + lane) * 4 + r];
        if (p.topk_weights) sum = sum * wr;             // fused_moe.py:150-154, before the scales
        float o = sum * sa;
        asm("" : "+v"(o));                               // ((acc * w) * a_scale) * b_scale, each product rounded to fp32
        o = o * sb;
        cp[t] = from_f32_exact<T>(o);
      }
    }
  }
}

}  // namespace aphro

using namespace aphro;

// One of the two grouped GEMMs of an FP8 MoE layer.  a: e4m3 [rows, K]; w: e4m3 [E, N, K]; c: [num_valid, N] in
// `dtype`; sorted_ids / expert_ids / num_post_pad: moe_align_block_size(block 16) outputs; max_blocks: expert_ids' length.
// top_k_div: A row of slot s is s / top_k_div (top_k for the first GEMM, 1 for the second); topk_weights NULL = not applied.
extern "C" int aphro_fp8_moe_gemm(const void* a, const void* w, const float* a_scale, const float* b_scales,
                                  const float* topk_weights, const int32_t* sorted_ids, const int32_t* expert_ids,
                                  const int32_t* num_post_pad, void* c, int64_t num_valid, int64_t N, int64_t K,
                                  int64_t max_blocks, int top_k_div, int dtype, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "fp8_moe_gemm: dtype must be f16 or bf16");
  APHRO_CHECK(K % 128 == 0 && N % 16 == 0 && top_k_div >= 1, "fp8_moe_gemm: K %% 128 == 0 and N %% 16 == 0 required (K=%ld N=%ld)", (long)K, (long)N);
  APHRO_CHECK(((uintptr_t)a % 16) == 0 && ((uintptr_t)w % 16) == 0, "fp8_moe_gemm: operands must be 16-byte aligned");
  if (num_valid == 0 || max_blocks == 0) return APHRO_OK;
  int nt = (N % 64 == 0) ? 4 : (N % 32 == 0) ? 2 : 1;
  int nw = 4;
  { const int v = APHRO_LAB_ENV_INT("APHRO_FP8_MOE_NT", 0); if ((v == 1 || v == 2 || v == 4) && N % (16 * v) == 0) nt = v; }
  { const int v = APHRO_LAB_ENV_INT("APHRO_FP8_MOE_NW", 0); if (v == 4 || v == 8) nw = v; }
  Fp8MoeParams p;
  p.a = (const uint8_t*)a; p.w = (const uint8_t*)w; p.a_scale = a_scale; p.b_scales = b_scales; p.topk_weights = topk_weights;
  p.sorted_ids = sorted_ids; p.expert_ids = expert_ids; p.num_post_pad = num_post_pad; p.c = c;
  p.N = (int)N; p.K = (int)K; p.num_valid = (int)num_valid; p.top_k_div = top_k_div;
  dim3 grid((unsigned)(N / (16 * nt)), 1, (unsigned)max_blocks);
#define L2(TT, NTV, NWV) hipLaunchKernelGGL((fp8_moe_gemm_kernel<TT, NTV, NWV>), grid, dim3(NWV * 64), 0, (hipStream_t)stream, p)
#define L(TT, NTV) { if (nw == 8) L2(TT, NTV, 8); else L2(TT, NTV, 4); }
#define LN(TT) { if (nt == 4) L(TT, 4) else if (nt == 2) L(TT, 2) else L(TT, 1) }
  if (dtype == APHRO_F16) LN(Half) else LN(BFloat)
#undef LN
#undef L
#undef L2
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
