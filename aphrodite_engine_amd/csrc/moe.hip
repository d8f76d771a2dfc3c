// Mixture-of-experts routing + dispatch for the W4A16 expert GEMMs:
//   topk_softmax           kernels/moe/softmax.cu:17-520 (schema kernels/moe/torch_bindings.cpp:11-14)
//   moe_align_block_size   kernels/moe/align_block_size_kernel.cu:17-126 (kernels/torch_bindings.cpp:394-399)
//   moe_gather_pack        the "replicate_input / sorted_ids" addressing of marlin_gemm_moe
//                          (kernels/moe/marlin_moe_ops.cu) as a fragment-major activation pack
//   moe_combine            routed-weight multiply + sum over top-k (fused_moe.py:520-542)
// The expert GEMMs themselves are wna16_gemm_kernel with a per-m-tile expert index
// (aphro_wna16_gemm_grouped, wna16_gemm.hip): one 16-row m-tile = one block of moe_align.
#include "common.h"

namespace aphro {

// one wave per token: softmax over the experts (fp32), then k rounds of arg-max
// (ties -> lowest expert index, like cub::ArgMax).  Weights are NOT renormalised.
__global__ void topk_softmax_kernel(float* __restrict__ topk_weights, int32_t* __restrict__ topk_ids,
                                    int32_t* __restrict__ token_expert_indices, const float* __restrict__ gating,
                                    int num_tokens, int num_experts, int k) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (tok >= num_tokens) return;
  const float* row = gating + (size_t)tok * num_experts;
  // softmax (softmax.cu:60-103): max, sum of exp, exp(x - max) * (1 / sum)
  float mx = -INFINITY;
  for (int e = lane; e < num_experts; e += 64) mx = __builtin_fmaxf(mx, row[e]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int e = lane; e < num_experts; e += 64) sum += expf(row[e] - mx);
  sum = wave_sum(sum);
  const float norm = 1.f / sum;
  // lane-resident probabilities for up to 256 experts
  float prob[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = lane + 64 * j;
    prob[j] = e < num_experts ? expf(row[e] - mx) * norm : -1.f;
  }
  for (int kk = 0; kk < k; ++kk) {
    float best = -1.f;
    int best_e = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = lane + 64 * j;
      if (prob[j] > best) { best = prob[j]; best_e = e; }   // ascending e within a lane: ties keep the lower
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oe = __shfl_xor(best_e, o, 64);
      if (ob > best || (ob == best && oe < best_e)) { best = ob; best_e = oe; }
    }
    if (lane == 0) {
      topk_weights[(size_t)tok * k + kk] = best;
      topk_ids[(size_t)tok * k + kk] = best_e;
      if (token_expert_indices) token_expert_indices[(size_t)tok * k + kk] = kk * num_tokens + tok;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (lane + 64 * j == best_e) prob[j] = -1.f;  // winner leaves the pool
  }
}

// Stable counting sort of the flattened (token, k) slots by expert, every expert's segment padded
// to a multiple of block_size (padding entries hold `numel`).  One wave; thread t owns a contiguous
// shard like the reference, so the order inside an expert is ascending slot index.
// inv_pos (optional): position of slot i in sorted_token_ids.
__global__ void moe_align_kernel(const int32_t* __restrict__ topk_ids, int32_t* __restrict__ sorted_token_ids,
                                 int32_t* __restrict__ expert_ids, int32_t* __restrict__ num_tokens_post_pad,
                                 int32_t* __restrict__ inv_pos, int num_experts, int block_size, int numel,
                                 int max_padded, int max_blocks) {
  extern __shared__ int32_t sm[];
  int32_t* cnt = sm;                                  // [65][E]
  int32_t* cumsum = sm + 65 * num_experts;            // [E + 1]
  const int t = threadIdx.x;                          // 0..63
  const int per = (numel + 63) / 64;
  const int lo = t * per, hi = min(numel, lo + per);
  for (int e = 0; e < num_experts; ++e) cnt[(t + 1) * num_experts + e] = 0;
  for (int i = lo; i < hi; ++i) ++cnt[(t + 1) * num_experts + topk_ids[i]];
  for (int i = t; i < max_padded; i += 64) sorted_token_ids[i] = numel;
  for (int i = t; i < max_blocks; i += 64) expert_ids[i] = -1;
  __syncthreads();
  for (int e = t; e < num_experts; e += 64) {         // prefix over the shards, per expert
    cnt[e] = 0;
    for (int s = 1; s <= 64; ++s) cnt[s * num_experts + e] += cnt[(s - 1) * num_experts + e];
  }
  __syncthreads();
  if (t == 0) {
    cumsum[0] = 0;
    for (int e = 1; e <= num_experts; ++e)
      cumsum[e] = cumsum[e - 1] + (cnt[64 * num_experts + e - 1] + block_size - 1) / block_size * block_size;
    *num_tokens_post_pad = cumsum[num_experts];
  }
  __syncthreads();
  for (int e = t; e < num_experts; e += 64)
    for (int i = cumsum[e]; i < cumsum[e + 1]; i += block_size) expert_ids[i / block_size] = e;
  for (int i = lo; i < hi; ++i) {
    const int e = topk_ids[i];
    const int pos = cnt[t * num_experts + e] + cumsum[e];
    sorted_token_ids[pos] = i;
    if (inv_pos) inv_pos[i] = pos;
    ++cnt[t * num_experts + e];
  }
}

// One token's routing with <= 16 experts, the whole row in ONE thread (no cross-lane step).  Bit-identical to the wave form of
// topk_softmax_kernel: max is exact in any order; the sum of exps replays wave_sum's butterfly (offsets 8, 4, 2, 1 over 16
// slots, the idle lanes' zeros left out: x + 0 = x); arg-max scans ascending, so ties keep the lower expert.
template <typename T>
__device__ __forceinline__ void moe_route_row16(const typename T::storage* __restrict__ row, int num_experts, int k,
                                                float (&wk)[8], int (&we)[8], float& wsum) {
  float ex[16];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    ex[j] = j < num_experts ? T::to_f32(row[j]) : 0.f;
    if (j < num_experts) mx = __builtin_fmaxf(mx, ex[j]);
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) ex[j] = j < num_experts ? expf(ex[j] - mx) : 0.f;
  float a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = ex[j];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    float nx[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) nx[j] = a[j] + a[j ^ o];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = nx[j];
  }
  const float norm = 1.f / a[0];
  float prob[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) prob[j] = j < num_experts ? ex[j] * norm : -1.f;
  wsum = 0.f;
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    wk[kk] = 0.f;
    we[kk] = 0;
    if (kk < k) {
      float best = -1.f;
      int best_e = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (prob[j] > best) { best = prob[j]; best_e = j; }
      wk[kk] = best;
      we[kk] = best_e;
      wsum += best;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (j == best_e) prob[j] = -1.f;
    }
  }
}

// moe_align_kernel's result by wave-wide ranking (a few dozen slots, <= 64 experts; ONE wave, lane e owns expert e).  Within an
// expert the slots stay in ascending order, as the counting sort leaves them (its shards are ascending slot ranges).  The
// output pointers may be global memory or LDS.
__device__ __forceinline__ void moe_align_rank64(const int32_t* ids, int numel, int num_experts, int block_size, int max_padded,
                                                 int max_blocks, int t, int32_t* sorted_token_ids, int32_t* expert_ids,
                                                 int32_t* num_tokens_post_pad, int32_t* inv_pos) {
  int my_cnt = 0;
  for (int c0 = 0; c0 < numel; c0 += 64) {
    const int my_e = c0 + t < numel ? ids[c0 + t] : -1;
    for (int e = 0; e < num_experts; ++e) {
      const unsigned long long mask = __ballot(my_e == e);
      if (t == e) my_cnt += __popcll(mask);
    }
  }
  const int padded = t < num_experts ? (my_cnt + block_size - 1) / block_size * block_size : 0;
  int incl = padded;                                // inclusive scan over the lanes
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o, 64);
    if (t >= o) incl += up;
  }
  const int start = incl - padded;                  // cumsum[e]
  const int total = __shfl(incl, 63, 64);
  if (t == 0) *num_tokens_post_pad = total;
  for (int i = t; i < max_padded; i += 64) sorted_token_ids[i] = numel;
  for (int i = t; i < max_blocks; i += 64) expert_ids[i] = -1;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // the fills above come from other lanes than the real entries below
  if (t < num_experts)
    for (int i = start; i < start + padded; i += block_size) expert_ids[i / block_size] = t;
  int base = start;                                 // lane e: next free position of expert e
  for (int c0 = 0; c0 < numel; c0 += 64) {
    const int i = c0 + t;
    const int my_e = i < numel ? ids[i] : -1;
    int pos = -1;
    for (int e = 0; e < num_experts; ++e) {
      const unsigned long long mask = __ballot(my_e == e);
      const int b = __shfl(base, e, 64);
      if (my_e == e) pos = b + __popcll(mask & ((1ull << t) - 1ull));
      if (t == e) base += __popcll(mask);
    }
    if (pos >= 0) {
      sorted_token_ids[pos] = i;
      if (inv_pos) inv_pos[i] = pos;
    }
  }
}

// fused_topk (fused_moe.py:369-402: gating.float() -> topk_softmax -> optional renormalise) + moe_align_block_size
// (:174-228) in ONE launch for decode-sized batches: the two ops, the fp32 cast and the renormalisation (sum + divide) are
// five launches of a few microseconds each in front of every sparse MLP -- at ~5 us of fixed cost per launch that is a
// tenth of a Mixtral decode step.  One workgroup of 16 waves: wave w routes tokens w, w + 16, ... with exactly the
// arithmetic of topk_softmax_kernel (bit-identical weights and ids), the ids stay in LDS, and after a barrier wave 0 runs
// moe_align_kernel's counting sort on them.  Renormalisation: w / (w_0 + w_1 + ...) in fp32, in slot order.
template <typename T>
__global__ __launch_bounds__(1024) void moe_route_align_kernel(
    float* __restrict__ topk_weights, int32_t* __restrict__ topk_ids_out, const typename T::storage* __restrict__ gating,
    int64_t gating_stride, int32_t* __restrict__ sorted_token_ids, int32_t* __restrict__ expert_ids,
    int32_t* __restrict__ num_tokens_post_pad, int32_t* __restrict__ inv_pos, int num_tokens, int num_experts, int k,
    int renormalize, int block_size, int max_padded, int max_blocks) {
  extern __shared__ int32_t sm[];
  const int numel = num_tokens * k;
  int32_t* ids = sm;                                  // [numel]
  int32_t* cnt = sm + numel;                          // [65][E]
  int32_t* cumsum = cnt + 65 * num_experts;           // [E + 1]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nwave = blockDim.x >> 6;
  if (num_experts <= 16) {
    // <= 16 experts (Mixtral: 8): a token's whole row lives in ONE thread -- no cross-lane step at all.  Bit-identical to
    // the wave form below: max is exact in any order; the sum of exps replays wave_sum's butterfly (offsets 8, 4, 2, 1 over
    // 16 slots, the idle lanes' zeros left out: x + 0 = x); arg-max scans ascending, so ties keep the lower expert.
    for (int tok = threadIdx.x; tok < num_tokens; tok += blockDim.x) {
      float wk[8];
      int we[8];
      float wsum;
      moe_route_row16<T>(gating + (size_t)tok * gating_stride, num_experts, k, wk, we, wsum);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        if (kk < k) {
          ids[tok * k + kk] = we[kk];
          topk_ids_out[(size_t)tok * k + kk] = we[kk];
          topk_weights[(size_t)tok * k + kk] = renormalize ? wk[kk] / wsum : wk[kk];
        }
      }
    }
  } else
  for (int tok = wave; tok < num_tokens; tok += nwave) {
    const typename T::storage* row = gating + (size_t)tok * gating_stride;
    float gv[4];                                      // the row once (num_experts <= 256): one global round trip, not three
#pragma unroll
    for (int j = 0; j < 4; ++j) gv[j] = lane + 64 * j < num_experts ? T::to_f32(row[lane + 64 * j]) : 0.f;
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (lane + 64 * j < num_experts) mx = __builtin_fmaxf(mx, gv[j]);
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (lane + 64 * j < num_experts) sum += expf(gv[j] - mx);
    sum = wave_sum(sum);
    const float norm = 1.f / sum;
    float prob[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = lane + 64 * j;
      prob[j] = e < num_experts ? expf(gv[j] - mx) * norm : -1.f;
    }
    float wsum = 0.f;
    float wk[8];
    for (int kk = 0; kk < k; ++kk) {
      float best = -1.f;
      int best_e = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = lane + 64 * j;
        if (prob[j] > best) { best = prob[j]; best_e = e; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oe = __shfl_xor(best_e, o, 64);
        if (ob > best || (ob == best && oe < best_e)) { best = ob; best_e = oe; }
      }
      if (kk < 8) wk[kk] = best;
      wsum += best;
      if (lane == 0) {
        ids[tok * k + kk] = best_e;
        topk_ids_out[(size_t)tok * k + kk] = best_e;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (lane + 64 * j == best_e) prob[j] = -1.f;
    }
    if (lane == 0)
      for (int kk = 0; kk < k; ++kk) topk_weights[(size_t)tok * k + kk] = renormalize ? wk[kk] / wsum : wk[kk];
  }
  __syncthreads();
  if (wave != 0) return;
  if (num_experts <= 64) {
    // ---- the decode case: a few dozen slots, <= 64 experts -- by wave-wide ranking (the 65-deep serial prefix of the counting
    // sort below was 2/3 of this launch's 12 us)
    moe_align_rank64(ids, numel, num_experts, block_size, max_padded, max_blocks, lane, sorted_token_ids, expert_ids,
                     num_tokens_post_pad, inv_pos);
    return;
  }
  // ---- moe_align_kernel on the ids in LDS (same shards, same order) -------------------------------------------------
  const int t = lane;
  const int per = (numel + 63) / 64;
  const int lo = t * per, hi = min(numel, lo + per);
  for (int e = 0; e < num_experts; ++e) cnt[(t + 1) * num_experts + e] = 0;
  for (int i = lo; i < hi; ++i) ++cnt[(t + 1) * num_experts + ids[i]];
  for (int i = t; i < max_padded; i += 64) sorted_token_ids[i] = numel;
  for (int i = t; i < max_blocks; i += 64) expert_ids[i] = -1;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int e = t; e < num_experts; e += 64) {
    cnt[e] = 0;
    for (int s2 = 1; s2 <= 64; ++s2) cnt[s2 * num_experts + e] += cnt[(s2 - 1) * num_experts + e];
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (t == 0) {
    cumsum[0] = 0;
    for (int e = 1; e <= num_experts; ++e)
      cumsum[e] = cumsum[e - 1] + (cnt[64 * num_experts + e - 1] + block_size - 1) / block_size * block_size;
    *num_tokens_post_pad = cumsum[num_experts];
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // (the global stores of expert_ids = -1 above and of the real ids below come from the same wave in program order, but
  //  from DIFFERENT lanes: make the fill visible first)
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  for (int e = t; e < num_experts; e += 64)
    for (int i = cumsum[e]; i < cumsum[e + 1]; i += block_size) expert_ids[i / block_size] = e;
  for (int i = lo; i < hi; ++i) {
    const int e = ids[i];
    const int pos = cnt[t * num_experts + e] + cumsum[e];
    sorted_token_ids[pos] = i;
    if (inv_pos) inv_pos[i] = pos;
    ++cnt[t * num_experts + e];
  }
}

// Fragment-major pack (see pack_a_kernel, wna16_gemm.hip) of the rows selected by sorted_token_ids:
// packed row r <- a[sorted[r] / topk] (a zero row for padding entries and for rows beyond
// *num_tokens_post_pad).
template <typename T>
__global__ void moe_gather_pack_kernel(const uint16_t* __restrict__ a, const int32_t* __restrict__ sorted_token_ids,
                                       const int32_t* __restrict__ num_tokens_post_pad,
                                       uint16_t* __restrict__ out, int m_pad, int K, int lda, int numel,
                                       int topk) {
  const int mtiles = m_pad >> 4;
  const int64_t total = (int64_t)(K >> 7) * 4 * mtiles * 64;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  int64_t blk = idx >> 6;
  const int mt = (int)(blk % mtiles); blk /= mtiles;
  const int u = (int)(blk & 3);
  const int seg = (int)(blk >> 2);
  const int row = 16 * mt + (lane & 15);
  const int k0 = 128 * seg + 32 * (lane >> 4) + 8 * u;
  u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (row < *num_tokens_post_pad) {
    const int slot = sorted_token_ids[row];
    if (slot < numel) {
      v = *reinterpret_cast<const u16x8*>(a + (size_t)(slot / topk) * lda + k0);
      if constexpr (!__is_same(T, Half)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bf16_bits_to_f16_bits_sat(v[j]);
      }
    }
  }
  *reinterpret_cast<u16x8*>(out + idx * 8) = v;
}

// moe_route_align_kernel + moe_gather_pack_kernel in ONE launch for decode-sized batches with <= 16 experts (round 6): the
// routing of a few dozen tokens is a few hundred instructions, so EVERY workgroup of the gather redoes it in its own LDS
// (thread per token, then wave 0 ranks the slots) instead of waiting for a one-workgroup launch to publish it -- a launch
// boundary and a dependent round trip less in front of every sparse MLP (6.7 + 4.9 us -> one launch).  Workgroup 0 also writes
// the routing to global memory for the grouped GEMMs and the combine.  Same arithmetic as the two kernels: bit-identical
// weights, ids, slot order and packed rows.
template <typename T>
__global__ __launch_bounds__(256) void moe_route_gather_kernel(
    float* __restrict__ topk_weights, int32_t* __restrict__ topk_ids_out, const typename T::storage* __restrict__ gating,
    int64_t gating_stride, int32_t* __restrict__ sorted_token_ids, int32_t* __restrict__ expert_ids,
    int32_t* __restrict__ num_tokens_post_pad, int32_t* __restrict__ inv_pos, int num_tokens, int num_experts, int k,
    int renormalize, int block_size, int max_padded, int max_blocks, const uint16_t* __restrict__ a,
    uint16_t* __restrict__ out, int m_pad, int K, int lda) {
  extern __shared__ int32_t sm[];
  const int numel = num_tokens * k;
  int32_t* ids = sm;                                  // [numel]
  int32_t* sorted_l = ids + numel;                    // [max_padded]
  int32_t* expert_l = sorted_l + max_padded;          // [max_blocks]
  int32_t* inv_l = expert_l + max_blocks;             // [numel]
  int32_t* post_l = inv_l + numel;                    // [1]
  const bool publish = blockIdx.x == 0;
  for (int tok = threadIdx.x; tok < num_tokens; tok += blockDim.x) {
    float wk[8];
    int we[8];
    float wsum;
    moe_route_row16<T>(gating + (size_t)tok * gating_stride, num_experts, k, wk, we, wsum);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      if (kk < k) {
        ids[tok * k + kk] = we[kk];
        if (publish) {
          topk_ids_out[(size_t)tok * k + kk] = we[kk];
          topk_weights[(size_t)tok * k + kk] = renormalize ? wk[kk] / wsum : wk[kk];
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 64)
    moe_align_rank64(ids, numel, num_experts, block_size, max_padded, max_blocks, (int)threadIdx.x, sorted_l, expert_l, post_l,
                     inv_l);
  __syncthreads();
  if (publish) {
    for (int i = threadIdx.x; i < max_padded; i += blockDim.x) sorted_token_ids[i] = sorted_l[i];
    for (int i = threadIdx.x; i < max_blocks; i += blockDim.x) expert_ids[i] = expert_l[i];
    if (inv_pos)
      for (int i = threadIdx.x; i < numel; i += blockDim.x) inv_pos[i] = inv_l[i];
    if (threadIdx.x == 0) *num_tokens_post_pad = *post_l;
  }
  // ---- the gather (moe_gather_pack_kernel on the routing in LDS) ---------------------------------------------------------
  const int mtiles = m_pad >> 4;
  const int64_t total = (int64_t)(K >> 7) * 4 * mtiles * 64;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  int64_t blk = idx >> 6;
  const int mt = (int)(blk % mtiles); blk /= mtiles;
  const int u = (int)(blk & 3);
  const int seg = (int)(blk >> 2);
  const int row = 16 * mt + (lane & 15);
  const int k0 = 128 * seg + 32 * (lane >> 4) + 8 * u;
  u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (row < *post_l) {
    const int slot = row < max_padded ? sorted_l[row] : numel;
    if (slot < numel) {
      v = *reinterpret_cast<const u16x8*>(a + (size_t)(slot / k) * lda + k0);
      if constexpr (!__is_same(T, Half)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bf16_bits_to_f16_bits_sat(v[j]);
      }
    }
  }
  *reinterpret_cast<u16x8*>(out + idx * 8) = v;
}

// out[t][:] = sum_k round_T(w[t][k] * y[pos(t, k)][:])   (fused_moe.py:520-542: routed weight applied to
// the second GEMM's fp32 result, cast, then summed over top-k).  y = sum of `nslab` fp32 slabs
// [nslab][m_pad][N] (split-K partials of the expert GEMM) -- rounded once, after the weight.
template <typename T>
__global__ void moe_combine_kernel(typename T::storage* __restrict__ out, const float* __restrict__ slabs,
                                   int nslab, int64_t slab_stride, const int32_t* __restrict__ inv_pos,
                                   const float* __restrict__ topk_weights, int topk, int N) {
  const int tok = blockIdx.x;
  for (int c0 = threadIdx.x * 4; c0 < N; c0 += blockDim.x * 4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kk = 0; kk < topk; ++kk) {
      const int pos = inv_pos[tok * topk + kk];
      const float w = topk_weights[tok * topk + kk];
      const float* p0 = slabs + (size_t)pos * N + c0;
      f32x4 y = *reinterpret_cast<const f32x4*>(p0);
      for (int s = 1; s < nslab; ++s) y += *reinterpret_cast<const f32x4*>(p0 + s * slab_stride);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += T::to_f32(from_f32_exact<T>(y[j] * w));
    }
    typename T::storage* o = out + (size_t)tok * N + c0;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = T::from_f32(acc[j]);
  }
}

}  // namespace aphro

using namespace aphro;

extern "C" int aphro_topk_softmax(float* topk_weights, int32_t* topk_ids, int32_t* token_expert_indices,
                                  const float* gating_output, int64_t num_tokens, int num_experts, int topk,
                                  void* stream) {
  APHRO_CHECK(num_experts >= 1 && num_experts <= 256, "topk_softmax: num_experts=%d (1..256 supported)", num_experts);
  APHRO_CHECK(topk >= 1 && topk <= num_experts, "topk_softmax: bad topk=%d", topk);
  if (num_tokens == 0) return APHRO_OK;
  const int wpb = 4;
  hipLaunchKernelGGL(topk_softmax_kernel, dim3((unsigned)((num_tokens + wpb - 1) / wpb)), dim3(wpb * 64), 0,
                     (hipStream_t)stream, topk_weights, topk_ids, token_expert_indices, gating_output,
                     (int)num_tokens, num_experts, topk);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_moe_align_block_size(const int32_t* topk_ids, int num_experts, int block_size,
                                          int32_t* sorted_token_ids, int32_t* expert_ids,
                                          int32_t* num_tokens_post_pad, int32_t* inv_pos, int64_t numel,
                                          void* stream) {
  APHRO_CHECK(num_experts >= 1 && num_experts <= 256 && block_size >= 1, "moe_align_block_size: bad arguments");
  APHRO_CHECK(numel >= 0 && numel < (1 << 24), "moe_align_block_size: numel out of range");
  const int max_padded = (int)numel + num_experts * (block_size - 1);
  const int max_blocks = (max_padded + block_size - 1) / block_size;
  const size_t lds = (size_t)(65 * num_experts + num_experts + 1) * sizeof(int32_t);
  hipLaunchKernelGGL(moe_align_kernel, dim3(1), dim3(64), lds, (hipStream_t)stream, topk_ids, sorted_token_ids,
                     expert_ids, num_tokens_post_pad, inv_pos, num_experts, block_size, (int)numel, max_padded,
                     max_blocks);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

// fused_topk + moe_align_block_size in one launch (decode-sized batches: num_tokens * topk <= 8192, topk <= 8).
// gating: [num_tokens, gating_stride] f16 / bf16 / f32 router logits.  Outputs as aphro_topk_softmax (+ renormalised
// weights when `renormalize`) and aphro_moe_align_block_size.
extern "C" int aphro_moe_route_align(float* topk_weights, int32_t* topk_ids, const void* gating, int64_t gating_stride,
                                     int32_t* sorted_token_ids, int32_t* expert_ids, int32_t* num_tokens_post_pad,
                                     int32_t* inv_pos, int64_t num_tokens, int num_experts, int topk, int renormalize,
                                     int block_size, int dtype, void* stream) {
  APHRO_CHECK(num_experts >= 1 && num_experts <= 256 && block_size >= 1, "moe_route_align: bad arguments");
  APHRO_CHECK(topk >= 1 && topk <= 8 && topk <= num_experts, "moe_route_align: topk=%d (1..8 supported)", topk);
  APHRO_CHECK(num_tokens >= 0 && num_tokens * topk <= 8192, "moe_route_align: %ld slots exceed the one-workgroup form", (long)(num_tokens * topk));
  APHRO_CHECK(dtype >= APHRO_F16 && dtype <= APHRO_F32, "moe_route_align: unsupported gating dtype %d", dtype);
  const int numel = (int)num_tokens * topk;
  const int max_padded = numel + num_experts * (block_size - 1);
  const int max_blocks = (max_padded + block_size - 1) / block_size;
  const size_t lds = (size_t)(numel + 65 * num_experts + num_experts + 1) * sizeof(int32_t);
  APHRO_CHECK(lds <= 64 * 1024, "moe_route_align: %zu bytes of LDS", lds);
#define L(TT)                                                                                                       \
  hipLaunchKernelGGL((moe_route_align_kernel<TT>), dim3(1), dim3(1024), lds, (hipStream_t)stream, topk_weights, topk_ids, \
                     (const typename TT::storage*)gating, gating_stride, sorted_token_ids, expert_ids, num_tokens_post_pad, \
                     inv_pos, (int)num_tokens, num_experts, topk, renormalize, block_size, max_padded, max_blocks)
  if (dtype == APHRO_F16) L(Half); else if (dtype == APHRO_BF16) L(BFloat); else L(Float);
#undef L
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

// 1 when aphro_moe_route_gather serves the call: <= 16 experts, <= 256 tokens, top-k <= 8, f16 / bf16 logits AND activations.
extern "C" int aphro_moe_route_gather_supported(int64_t num_tokens, int num_experts, int topk, int block_size, int64_t K) {
  if (num_tokens < 1 || num_tokens > 256 || num_experts < 1 || num_experts > 16 || topk < 1 || topk > 8 || topk > num_experts)
    return 0;
  if (block_size != 16 || K % 128 != 0) return 0;
  return 1;
}

// aphro_moe_route_align followed by aphro_moe_gather_pack (m_pad = the padded bound rounded up to 16 rows) in one launch.
// gating [num_tokens, gating_stride] and a [num_tokens, lda] in `dtype` (f16 / bf16); packed as aphro_moe_gather_pack.
extern "C" int aphro_moe_route_gather(float* topk_weights, int32_t* topk_ids, const void* gating, int64_t gating_stride,
                                      int32_t* sorted_token_ids, int32_t* expert_ids, int32_t* num_tokens_post_pad,
                                      int32_t* inv_pos, int64_t num_tokens, int num_experts, int topk, int renormalize,
                                      int block_size, const void* a, int64_t lda, void* packed, int64_t m_pad, int64_t K,
                                      int dtype, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "moe_route_gather: dtype must be f16 or bf16");
  APHRO_CHECK(aphro_moe_route_gather_supported(num_tokens, num_experts, topk, block_size, K),
              "moe_route_gather: %ld tokens, %d experts, top-%d, block %d, K=%ld is not served", (long)num_tokens, num_experts,
              topk, block_size, (long)K);
  const int numel = (int)num_tokens * topk;
  const int max_padded = numel + num_experts * (block_size - 1);
  const int max_blocks = (max_padded + block_size - 1) / block_size;
  APHRO_CHECK(m_pad % 16 == 0 && m_pad >= max_padded && lda % 8 == 0 && lda >= K, "moe_route_gather: bad shape");
  const size_t lds = (size_t)(2 * numel + max_padded + max_blocks + 1) * sizeof(int32_t);
  const int64_t total = (K / 128) * 4 * (m_pad / 16) * 64;
  dim3 grid((unsigned)((total + 255) / 256));
#define L(TT)                                                                                                              \
  hipLaunchKernelGGL((moe_route_gather_kernel<TT>), grid, dim3(256), lds, (hipStream_t)stream, topk_weights, topk_ids,     \
                     (const typename TT::storage*)gating, gating_stride, sorted_token_ids, expert_ids, num_tokens_post_pad, \
                     inv_pos, (int)num_tokens, num_experts, topk, renormalize, block_size, max_padded, max_blocks,         \
                     (const uint16_t*)a, (uint16_t*)packed, (int)m_pad, (int)K, (int)lda)
  if (dtype == APHRO_F16) L(Half); else L(BFloat);
#undef L
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_moe_gather_pack(const void* a, const int32_t* sorted_token_ids,
                                     const int32_t* num_tokens_post_pad, void* packed, int64_t m_pad, int64_t K,
                                     int64_t lda, int64_t numel, int topk, int dtype, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "moe_gather_pack: dtype must be f16 or bf16");
  APHRO_CHECK(m_pad % 16 == 0 && K % 128 == 0 && lda % 8 == 0 && topk >= 1, "moe_gather_pack: bad shape");
  if (m_pad == 0) return APHRO_OK;
  const int64_t total = (K / 128) * 4 * (m_pad / 16) * 64;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == APHRO_F16)
    hipLaunchKernelGGL((moe_gather_pack_kernel<Half>), grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a,
                       sorted_token_ids, num_tokens_post_pad, (uint16_t*)packed, (int)m_pad, (int)K, (int)lda,
                       (int)numel, topk);
  else
    hipLaunchKernelGGL((moe_gather_pack_kernel<BFloat>), grid, dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)a, sorted_token_ids, num_tokens_post_pad, (uint16_t*)packed, (int)m_pad,
                       (int)K, (int)lda, (int)numel, topk);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_moe_combine(void* out, const float* slabs, int nslab, int64_t m_pad, const int32_t* inv_pos,
                                 const float* topk_weights, int64_t num_tokens, int topk, int64_t N, int dtype,
                                 void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "moe_combine: dtype must be f16 or bf16");
  APHRO_CHECK(N % 4 == 0 && nslab >= 1, "moe_combine: bad shape");
  if (num_tokens == 0) return APHRO_OK;
  const int threads = N / 4 >= 256 ? 256 : (int)((N / 4 + 63) / 64 * 64);
  if (dtype == APHRO_F16)
    hipLaunchKernelGGL((moe_combine_kernel<Half>), dim3((unsigned)num_tokens), dim3(threads), 0,
                       (hipStream_t)stream, (uint16_t*)out, slabs, nslab, m_pad * N, inv_pos, topk_weights, topk,
                       (int)N);
  else
    hipLaunchKernelGGL((moe_combine_kernel<BFloat>), dim3((unsigned)num_tokens), dim3(threads), 0,
                       (hipStream_t)stream, (uint16_t*)out, slabs, nslab, m_pad * N, inv_pos, topk_weights, topk,
                       (int)N);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
