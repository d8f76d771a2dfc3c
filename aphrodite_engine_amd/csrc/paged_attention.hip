// Paged-attention decode for gfx950: one kernel serves _C::paged_attention_v1,
// _C::paged_attention_v2 and _rocm_C::paged_attention.
//
// Reference semantics: kernels/attention/attention_kernels.cu:87-669 (math),
// kernels/rocm/attention.cu:206-1042 (the MI300 kernel whose *role* this
// fills; nothing of its structure is reused).
//
// Design (DESIGN.md "paged_attention"):  HBM-bound, KV read exactly once.
//  * one workgroup = one (sequence, kv-head, partition); all GQA query heads of
//    the kv-head share the K/V stream (heads sit in the MFMA N dimension).
//  * S^T[token, head] = K[token, :] . Q^T  with v_mfma_f32_16x16x32: the K cache
//    layout [hd/x][block][x] (x = 16 B) makes the A fragment of a 16-token tile a
//    lane-linear 1 KiB load (lane = 16*chunk + token) -- perfectly coalesced,
//    no LDS staging, no transposes.
//  * online softmax per head in registers; the probabilities leave the QK MFMA
//    in exactly the lane layout the PV MFMA wants as its B operand if the
//    "k index" of that MFMA is defined as (tile, 4g+r) -- so P never moves
//    between lanes.  O^T[d, head] += V^T[d, token] . P^T[token, head].
//  * fp8 (e4m3 / e5m2) KV is widened to the query dtype in registers (exact),
//    k_scale folds into the softmax scale and v_scale into the final
//    normalisation: dequant costs no extra memory pass.
//  * waves of a workgroup interleave 32-token tile pairs and merge (m, l, O)
//    through LDS; partitions merge by the reference's exp_sums/max_logits/
//    tmp_out contract.
#include <mutex>

#include "common.h"

namespace aphro {

struct PAParams {
  void* out;
  float* exp_sums;
  float* max_logits;
  void* tmp_out;
  const void* q;
  const void* kc;
  const void* vc;
  const int32_t* block_tables;
  const int32_t* seq_lens;
  const float* alibi;
  int num_heads, num_kv_heads;
  int max_blocks_per_seq;
  int partition_size;  // tokens per grid.z slice (multiple of 32) ; 0 -> whole sequence
  int max_parts;       // P of the scratch tensors (0 in v1 form)
  int nh_lds;          // query heads per kv head held in the LDS merge buffer (<= 16)
  int write_direct;    // 1: write `out` (single partition) ; 0: write scratch
  void* out_packed;    // optional fragment-major f16 copy of out for the o_proj GEMM (see wna16_gemm.hip)
  int pack_mtiles;     // ceil(num_seqs / 16)
  // optional e4m3 copy of out for an FP8 o_proj with a STATIC input scale: fp8(T(out) * (1 / *out_q8_scale)), the bits
  // static_scaled_fp8_quant (fp8/common.cu:187-199) produces from `out`
  uint8_t* out_q8;
  const float* out_q8_scale;
  // optional, DYNAMIC per-token scheme: absmax over this workgroup's slice of `out` ((sequence, kv-head): gqa x hd values)
  // -> out_absmax[seq][kv_head].  The row's absmax is the max over its num_kv_heads partials (max is order-free), so the
  // o_proj GEMM that reads `out` works out dynamic_per_token_scaled_fp8_quant's scale itself (fp8_gemm_resident.hip, AQ).
  float* out_absmax;
  void* out_pairs;     // ... and `out` in the pair-major layout that GEMM reads lane-linearly (common.h aq_pair_offset; out may be NULL)
  float scale;         // softmax scale * k_scale
  float v_scale;
  int64_t q_stride, kv_block_stride, kv_head_stride;
  // ---- fused QKV-slab reduce + rotary embedding + cache write (decode fast path) ----
  // When qkv_slabs != NULL the kernel takes q/k/v of the NEW token of every sequence from the
  // fp32 split-K slabs of the qkv GEMM ([nslab][num_seqs][(Hq+2Hkv)*hd]), applies NeoX rotary
  // embedding (rot_dim == hd) and writes K/V to the cache slot before attending -- the work of
  // rotary_embedding + reshape_and_cache (pos_encoding_kernels.cu:10-160, cache_kernels.cu:
  // 152-204) with identical roundings.  v1 form, hd == 128 only.
  const float* qkv_slabs;
  int nslab;
  int64_t slab_stride;           // num_seqs * (Hq + 2 Hkv) * hd
  const int64_t* positions;      // [num_seqs]
  const uint16_t* cos_sin;       // T [max_pos, hd]: cos | sin
  const int64_t* slot_mapping;   // [num_seqs]
  float k_scale_raw, v_scale_raw;
  // optional dequantisation of the slabs (FP8 W8A8 qkv GEMM): value = row[seq] * (col[c] * sum)
  const float* slab_row_scale;   // [num_seqs] or NULL
  const float* slab_col_scale;   // [(Hq + 2 Hkv) * hd] or NULL
  // ---- split-KV inside ONE launch (v1 / packed / fused forms; GQA <= 16) -----------------------------------------
  // grid.z = nsplit workgroups share one (sequence, kv-head): each attends over an equal run of 32-token pairs, leaves
  // its unnormalised (O, m, l) in split_scratch with write-through stores and takes a ticket from split_counter; the
  // LAST arriver merges all runs with the reference's partition-merge math (attention_kernels.cu:637-668) and writes the
  // output -- no second launch, nobody waits.  Fills the chip when num_seqs x num_kv_heads << CUs (a TP8 shard of
  // Llama-3-70B has ONE kv head per GPU: 64 workgroups at bs 64).
  int nsplit;                    // 0 / 1: off
  float* split_scratch;          // [num_seqs * Hkv][nsplit][16 * HD + 32]
  unsigned* split_counter;       // [num_seqs * Hkv], zero between launches (the merging workgroup resets it)
};


// write-through (system-scope) store / coherent loads: partial results cross XCDs (separate L2s) inside one launch.  No
// release / acquire FENCES: at agent scope those write back / invalidate the whole L2 (+16 us per launch).
__device__ __forceinline__ void st_wt_f32x4(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_wt_f32x2(float* p, f32x2 v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
// (m, l) of four runs and one O element of each, all twelve loads in flight together; the wait sits INSIDE the statement
// (with a separate s_waitcnt the compiler may move a destination register before its data has arrived -- seen in
// custom_all_reduce.hip)
__device__ __forceinline__ void ld_coh_runs4(float (&m)[4], float (&l)[4], float (&a)[4], const float* const (&pm)[4],
                                             const float* const (&pa)[4]) {
  asm volatile(
      "global_load_dword %0, %12, off sc0 sc1\n\tglobal_load_dword %4, %12, off offset:4 sc0 sc1\n\tglobal_load_dword %8, %16, off sc0 sc1\n\t"
      "global_load_dword %1, %13, off sc0 sc1\n\tglobal_load_dword %5, %13, off offset:4 sc0 sc1\n\tglobal_load_dword %9, %17, off sc0 sc1\n\t"
      "global_load_dword %2, %14, off sc0 sc1\n\tglobal_load_dword %6, %14, off offset:4 sc0 sc1\n\tglobal_load_dword %10, %18, off sc0 sc1\n\t"
      "global_load_dword %3, %15, off sc0 sc1\n\tglobal_load_dword %7, %15, off offset:4 sc0 sc1\n\tglobal_load_dword %11, %19, off sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2]), "=&v"(m[3]), "=&v"(l[0]), "=&v"(l[1]), "=&v"(l[2]), "=&v"(l[3]),
        "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3])
      : "v"(pm[0]), "v"(pm[1]), "v"(pm[2]), "v"(pm[3]), "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3])
      : "memory");
}

template <typename T>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mfma16<Half>(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mfma16<BFloat>(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// two floats (exactly representable in T) -> packed pair of T
template <typename T>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  if constexpr (__is_same(T, Half)) {
    f16x2 h = {(f16)a, (f16)b};
    return __builtin_bit_cast(uint32_t, h);
  } else {
    return (uint32_t)f32_to_bf16_bits(a) | ((uint32_t)f32_to_bf16_bits(b) << 16);
  }
}
// 4 packed fp8 -> 4 values of T (two dwords)
template <typename T, bool E5M2>
__device__ __forceinline__ u32x2 fp8x4_to_T(uint32_t w) {
  u32x2 r;
  r[0] = fp8x2_to_T<T, E5M2, false>(w);
  r[1] = fp8x2_to_T<T, E5M2, true>(w);
  return r;
}

// KV: 0 = cache holds T, 1 = e4m3, 2 = e5m2.   HD: head size.  BS: block size.
// ROPE: the fused rotary + cache-write form (a separate instantiation: the plain kernel must not
// pay for the extra live state -- measured +1.4 us per launch when it was a runtime branch)
// Kernel arguments: the head of the dependent chain (sequence length -> block table -> K/V addresses) and the sizes every
// address needs come first and as scalars, so that they are preloaded into SGPRs (Makefile: -amdgpu-kernarg-preload-count);
// p_in carries the rest, its copies of the leading fields are not read.
template <typename T, int KV, int HD, int BS, int NW, int ROPE, bool SPLIT = false>
__global__ __launch_bounds__(NW * 64) void paged_attention_kernel(const int32_t* seq_lens, const int32_t* block_tables,
                                                                  const void* kc_, const void* vc_, const float* qkv_slabs,
                                                                  int num_heads, int num_kv_heads, int max_blocks_per_seq,
                                                                  int partition_size, PAParams p_in) {
  PAParams p = p_in;
  p.seq_lens = seq_lens; p.block_tables = block_tables; p.kc = kc_; p.vc = vc_; p.qkv_slabs = qkv_slabs;
  p.num_heads = num_heads; p.num_kv_heads = num_kv_heads; p.max_blocks_per_seq = max_blocks_per_seq;
  p.partition_size = partition_size;
  constexpr bool FP8 = KV != 0;
  constexpr bool E5M2 = KV == 2;
  constexpr int XB = 16;                        // bytes per K chunk
  constexpr int XE = FP8 ? 16 : 8;              // elements per K chunk
  constexpr int NCH = HD / XE;                  // chunks per token
  constexpr int NLD = (NCH + 3) / 4;            // 16-B K loads per lane per tile
  constexpr int NKS = FP8 ? 2 * NLD : (HD + 31) / 32;  // QK MFMAs per tile
  constexpr int NDT = (HD + 15) / 16;           // PV d-tiles
  constexpr int ESZ = FP8 ? 1 : 2;              // cache element bytes
  // Token <-> MFMA slot of a 32-token tile pair.  16-bit KV: row 4g + r of score tile jj is token 16jj + 4g + r (a tile =
  // 16 consecutive tokens: whole 128-byte lines per K load).  fp8 KV (TOK8, round 6): token 8g + 4jj + r, so that the eight
  // k-slots a lane feeds to the PV MFMA -- (jj, r) -- are EIGHT CONSECUTIVE tokens of one cache block and the lane's V bytes
  // of a d-row come with ONE 8-byte load instead of two 4-byte ones (the V side of a pair: 8 vector-memory instructions
  // instead of 16; configs[2] attention 94.6 -> 90.6 us).  With 16-bit KV the same map measured +2.4 us per launch at ctx
  // 1024 and +-0 at 8192 (profiles/r6_decode_experiments.txt (5)): not used there.
  constexpr bool TOK8 = FP8;
  using VRaw = u32x2;                           // TOK8: a lane's 8 fp8 tokens of one V row; else 4 tokens (fp8: in [0])
  constexpr int NVJ = TOK8 ? 1 : 2;             // V loads per d-tile
  auto tok_of_k_row = [](int jj, int c) { return TOK8 ? 8 * (c >> 2) + 4 * jj + (c & 3) : 16 * jj + c; };
  auto tok_of_s_row = [](int jj, int g, int r) { return TOK8 ? 8 * g + 4 * jj + r : 16 * jj + 4 * g + r; };
  static_assert(HD % XE == 0, "head size must be a multiple of the K chunk");

  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int kv_flag;  // fused form: "the new token's K/V are in the cache" (see below)
  __shared__ __attribute__((aligned(16))) uint16_t q_lds[ROPE ? 16 * HD : 8];  // fused form: rotated q, [head][d]

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4;
  const int c = lane & 15;
  const int kvh = blockIdx.x;
  const int seq = blockIdx.y;
  const int part = blockIdx.z;
  const int gqa = p.num_heads / p.num_kv_heads;
  constexpr bool split = SPLIT;
  const int seq_len = p.seq_lens[seq];       // (a template parameter: the unsplit instantiations keep their register budget)
  // fp8 copy of the output (scaled-slab form only): the scale is fetched HERE, a whole kernel ahead of its use -- at the
  // store it would be a serial global load (and a reload after every byte store, which may alias it)
  float q8_inv = 0.f;
  if constexpr (ROPE == 2) {
    if (p.out_q8) q8_inv = 1.0f / *(const volatile float*)p.out_q8_scale;
    asm volatile("" : "+v"(q8_inv));
  }
  // split form: equal runs of 32-token pairs per workgroup of the (sequence, kv-head) group
  const int split_pp = split ? (((seq_len + 31) >> 5) + p.nsplit - 1) / p.nsplit : 0;
  const int psz = split ? max(split_pp, 1) * 32 : (p.partition_size > 0 ? p.partition_size : 0x7fffffe0);
  const int pstart = part * psz;
  const bool empty_run = pstart >= seq_len;
  if (empty_run && !split) return;       // (split form: an empty run still takes its ticket -- it may be the last arriver)
  const int pend = empty_run ? pstart : min(seq_len, pstart + psz);
  const int32_t* bt = p.block_tables + (size_t)seq * p.max_blocks_per_seq;
  const int alloc_tokens = ((seq_len + BS - 1) / BS) * BS;  // addressable tokens

  const char* kc = (const char*)p.kc + (size_t)kvh * p.kv_head_stride * ESZ;
  const char* vc = (const char*)p.vc + (size_t)kvh * p.kv_head_stride * ESZ;

  // ---- fused rope + cache write of the new token (see PAParams) ---------------------------
  constexpr bool fused_rope = ROPE != 0 && HD == 128;
  constexpr bool scaled_slabs = ROPE == 2;  // slabs of a quantised projection: dequantised on the fly
  if constexpr (fused_rope) {
    if (threadIdx.x == 0) kv_flag = 0;
    __syncthreads();
  }
  const int ntot = (p.num_heads + 2 * p.num_kv_heads) * HD;
  const uint16_t* cs_row = nullptr;
  auto slab8 = [&](int col, float (&o8)[8]) {  // 8 consecutive qkv columns of this sequence, rounded to T
    const float* p0 = p.qkv_slabs + (size_t)seq * ntot + col;
    f32x4 a = *reinterpret_cast<const f32x4*>(p0), b = *reinterpret_cast<const f32x4*>(p0 + 4);
    for (int k = 1; k < p.nslab; ++k) {
      a += *reinterpret_cast<const f32x4*>(p0 + k * p.slab_stride);
      b += *reinterpret_cast<const f32x4*>(p0 + k * p.slab_stride + 4);
    }
    if constexpr (scaled_slabs) {   // a_scale * (b_scale * acc), the scaled-mm epilogue order
      const float rsc = p.slab_row_scale ? p.slab_row_scale[seq] : 1.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // fp32 product first, rounding to T second (kept apart: see from_f32_exact, common.h)
        float ta = rsc * (p.slab_col_scale[col + i] * a[i]);
        float tb = rsc * (p.slab_col_scale[col + 4 + i] * b[i]);
        asm("" : "+v"(ta), "+v"(tb));
        a[i] = ta;
        b[i] = tb;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o8[i] = T::to_f32(T::from_f32(a[i]));
      o8[4 + i] = T::to_f32(T::from_f32(b[i]));
    }
  };
  auto rope8 = [&](int col, int d0, u16x8& xo8, u16x8& yo8) {  // dims d0..d0+7 and their partners d0+HD/2
    float xv[8], yv[8];
    slab8(col + d0, xv);
    slab8(col + HD / 2 + d0, yv);
    const u16x8 c8 = *reinterpret_cast<const u16x8*>(cs_row + d0);
    const u16x8 s8 = *reinterpret_cast<const u16x8*>(cs_row + HD / 2 + d0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float xo, yo;
      rope_pair(xv[j], yv[j], T::to_f32(c8[j]), T::to_f32(s8[j]), xo, yo);
      xo8[j] = T::from_f32(xo);
      yo8[j] = T::from_f32(yo);
    }
  };
  // positions == NULL: cos_sin already holds one gathered row per sequence (no dependent load)
  if (fused_rope) cs_row = p.cos_sin + (p.positions ? p.positions[seq] : (int64_t)seq) * HD;
  // The wave that owns the LAST tile pair writes the new token's K/V before it loads that pair:
  // the cache lines of one 16-token block are only ever read by the wave that owns its tile, so
  // a wave-local release/acquire fence is enough (no workgroup barrier).
  auto write_new_kv = [&]() __attribute__((always_inline)) {
    {
      const int64_t slot = p.slot_mapping[seq];
      if (slot >= 0) {
        const int64_t blk = slot / BS;
        const int off = (int)(slot % BS);
        const int nq = p.num_heads * HD;
        if (lane < HD / 16) {  // K: rotary, then [hd/x][block][x] chunks
          const int d0 = 8 * lane;
          u16x8 xo8, yo8;
          rope8(nq + kvh * HD, d0, xo8, yo8);
#pragma unroll
          for (int part2 = 0; part2 < 2; ++part2) {
            const int d = part2 ? HD / 2 + d0 : d0;
            const u16x8& o8 = part2 ? yo8 : xo8;
            char* dst = const_cast<char*>(kc) + ((size_t)blk * p.kv_block_stride + (size_t)(d / XE) * BS * XE +
                                                 (size_t)off * XE + d % XE) * ESZ;
            if constexpr (!FP8) {
              *reinterpret_cast<u16x8*>(dst) = o8;
            } else {
              uint32_t w0 = 0, w1 = 0;
#pragma unroll
              for (int j = 0; j < 4; j += 2) {
                w0 |= f32x2_to_fp8<E5M2>(T::to_f32(o8[j]) / p.k_scale_raw, T::to_f32(o8[j + 1]) / p.k_scale_raw) << (8 * j);
                w1 |= f32x2_to_fp8<E5M2>(T::to_f32(o8[4 + j]) / p.k_scale_raw, T::to_f32(o8[5 + j]) / p.k_scale_raw) << (8 * j);
              }
              *reinterpret_cast<u32x2*>(dst) = u32x2{w0, w1};
            }
          }
        } else if (lane < HD / 16 + HD / 8) {  // V: [hd][block]
          const int d0 = 8 * (lane - HD / 16);
          float vv[8];
          slab8(nq + p.num_kv_heads * HD + kvh * HD + d0, vv);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            char* dst = const_cast<char*>(vc) + ((size_t)blk * p.kv_block_stride + (size_t)(d0 + j) * BS + off) * ESZ;
            if constexpr (!FP8) *reinterpret_cast<uint16_t*>(dst) = T::from_f32(vv[j]);
            else *reinterpret_cast<uint8_t*>(dst) = (uint8_t)f32x2_to_fp8<E5M2>(vv[j] / p.v_scale_raw, 0.f);
          }
        }
      }
      // wave-local visibility only: the stores must have been performed (write-through L1) before
      // this wave's loads of the same lines.  NOT __threadfence(): an agent-scope fence writes back
      // and invalidates the whole L2 (measured +16 us per launch).
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
  };

  // (the fused forms serve GQA <= 16 only -- checked by the launcher -- : ONE pass, said at compile time, so that hipcc does not
  //  hoist the pass-invariant address arithmetic of the cache write and the epilogue out of a loop that never repeats: that
  //  hoisting cost the scaled-slab instantiation 60 bytes of scratch per lane and with it ~4 us per launch -- round 6,
  //  tools/attn_forms_bench.py: 34.0 us against 28.1 for the plain fused form, whose allocation fits)
  const int hb_end = fused_rope ? 1 : gqa;
  for (int hb = 0; hb < hb_end; hb += 16) {  // >16 query heads per kv head: extra passes
    const int nh = min(16, gqa - hb);
    const int head = kvh * gqa + hb + c;  // this lane's query head (valid if c < nh)
    const float slope = (p.alibi != nullptr && c < nh) ? p.alibi[head] : 0.f;
    f32x4 o[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -1e30f;
    float l_run = 0.f;  // this lane's share (its 8 tokens of every pair); summed over g at the end
    u32x4 qf[NKS];
    const int pair0 = pstart >> 5;
    const int pair_end = (pend + 31) >> 5;
    // K/V registers of one 32-token tile pair: kf = K fragments, vraw = 4 tokens x 16 d-rows per
    // lane (16-bit KV: 8 B, fp8: 8 tokens in one 8-byte load, see TOK8).  ONE set: a pair is loaded, then computed; the other 7 waves hide the
    // latency.  (A second set -- the next pair's loads issued before this pair's compute -- was measured in round 2:
    // 256 VGPRs with 74 spills in the fused-rope form, 44 us instead of 29.7; fp8 KV 26.6 instead of 20 us.  An L2
    // warm-up of the next pair instead -- one dword per 128-byte line, issued when this pair's compute starts -- also
    // lost: 36.1 / 24.0 us.)
    // Block-table entries of a pair ([jj] K block, [2 + jj] V block): fetched ONE PAIR AHEAD so that
    // the K/V loads never wait behind a dependent table lookup.
    auto load_ids = [&](int pr, int (&ids)[4]) __attribute__((always_inline)) {
      const int tb = pr << 5;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        ids[jj] = bt[min(tb + tok_of_k_row(jj, c), seq_len - 1) / BS];
        if constexpr (TOK8) ids[2 + jj] = bt[min(tb + 8 * g, alloc_tokens - 8) / BS];
        else ids[2 + jj] = bt[min(tb + 16 * jj + 4 * g, alloc_tokens - 4) / BS];
      }
    };
    auto load_pair = [&](int pr, const int (&ids)[4], u32x4 (&kf)[2][NLD], VRaw (&vraw)[NDT][NVJ])
        __attribute__((always_inline)) {
      const int tb = pr << 5;
      // ---- addresses -----------------------------------------------------------
      const char* kptr[2];
      const char* vptr[NVJ];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        int tk = min(tb + tok_of_k_row(jj, c), seq_len - 1);
        kptr[jj] = kc + ((size_t)ids[jj] * p.kv_block_stride + (size_t)(tk % BS) * XE) * ESZ;
        if (jj < NVJ) {
          int tv = TOK8 ? min(tb + 8 * g, alloc_tokens - 8) : min(tb + 16 * jj + 4 * g, alloc_tokens - 4);
          vptr[jj] = vc + ((size_t)ids[2 + jj] * p.kv_block_stride + (size_t)c * BS + (tv % BS)) * ESZ;
        }
      }
      // ---- issue all K and V loads of the pair ----------------------------------
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ld = 0; ld < NLD; ++ld) {
          int ch = 4 * ld + g;
          if (ch < NCH)
            kf[jj][ld] = __builtin_nontemporal_load(
                reinterpret_cast<const u32x4*>(kptr[jj] + (size_t)ch * BS * XB));
          else
            kf[jj][ld] = u32x4{0, 0, 0, 0};
        }
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int jj = 0; jj < NVJ; ++jj) {
          const char* vp = vptr[jj] + (size_t)dt * 16 * BS * ESZ;
          if (16 * dt + c < HD) {
            if constexpr (FP8 && !TOK8) {
              vraw[dt][jj][0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(vp));
              vraw[dt][jj][1] = 0;
            } else {
              vraw[dt][jj] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(vp));
            }
          } else {
            vraw[dt][jj] = u32x2{0, 0};
          }
        }
    };
    auto compute_pair = [&](int pr, u32x4 (&kf)[2][NLD], VRaw (&vraw)[NDT][NVJ]) __attribute__((always_inline)) {
      const int tb = pr << 5;
      // ---- S^T = K . Q^T -----------------------------------------------------------
      f32x4 s[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        s[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (FP8) {
#pragma unroll
          for (int ld = 0; ld < NLD; ++ld) {
            u32x2 c0 = fp8x4_to_T<T, E5M2>(kf[jj][ld][0]);
            u32x2 c1 = fp8x4_to_T<T, E5M2>(kf[jj][ld][1]);
            u32x2 c2 = fp8x4_to_T<T, E5M2>(kf[jj][ld][2]);
            u32x2 c3 = fp8x4_to_T<T, E5M2>(kf[jj][ld][3]);
            u32x4 lo = {c0[0], c0[1], c1[0], c1[1]};
            u32x4 hi = {c2[0], c2[1], c3[0], c3[1]};
            s[jj] = mfma16<T>(lo, qf[2 * ld], s[jj]);
            s[jj] = mfma16<T>(hi, qf[2 * ld + 1], s[jj]);
          }
        } else {
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) s[jj] = mfma16<T>(kf[jj][ks], qf[ks], s[jj]);
        }
      }
      // ---- online softmax (lane = head column c, tokens tb+16jj+4g+r) ------------------
      float pv[2][4];
      float mx = -1e30f;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int tok = tb + tok_of_s_row(jj, g, r);
          float x = s[jj][r] * p.scale + slope * (float)(tok - seq_len + 1);
          bool ok = tok >= pstart && tok < pend;
          x = ok ? x : -1e30f;
          pv[jj][r] = x;
          mx = __builtin_fmaxf(mx, x);
        }
      mx = __builtin_fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = __builtin_fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = __builtin_fmaxf(m_run, mx);
      const float alpha = __expf(m_run - m_new);
      m_run = m_new;
      float lsum = 0.f;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float e = pv[jj][r] > -1e29f ? __expf(pv[jj][r] - m_new) : 0.f;
          pv[jj][r] = e;
          lsum += e;
        }
      l_run = l_run * alpha + lsum;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) o[dt] *= alpha;
      u32x4 pf;  // B operand of the PV MFMA: k = (jj, 4g + r)
      pf[0] = pack2<T>(pv[0][0], pv[0][1]);
      pf[1] = pack2<T>(pv[0][2], pv[0][3]);
      pf[2] = pack2<T>(pv[1][0], pv[1][1]);
      pf[3] = pack2<T>(pv[1][2], pv[1][3]);
      // ---- O^T += V^T . P^T --------------------------------------------------------
      const bool ragged = (tb + 32 > pend);  // wave-uniform: last pair of the range
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        u32x4 vf;
        if constexpr (TOK8) {
          u32x2 c0 = fp8x4_to_T<T, E5M2>(vraw[dt][0][0]);
          u32x2 c1 = fp8x4_to_T<T, E5M2>(vraw[dt][0][1]);
          vf = u32x4{c0[0], c0[1], c1[0], c1[1]};
        } else if constexpr (FP8) {
          u32x2 c0 = fp8x4_to_T<T, E5M2>(vraw[dt][0][0]);
          u32x2 c1 = fp8x4_to_T<T, E5M2>(vraw[dt][NVJ - 1][0]);
          vf = u32x4{c0[0], c0[1], c1[0], c1[1]};
        } else {
          vf[0] = vraw[dt][0][0]; vf[1] = vraw[dt][0][1];
          vf[2] = vraw[dt][NVJ - 1][0]; vf[3] = vraw[dt][NVJ - 1][1];
        }
        if (ragged) {  // zero V of tokens outside [pstart, pend): 0 * NaN must not poison O
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
              int tok = tb + tok_of_s_row(jj, g, 2 * h2);
              uint32_t msk = (tok < pend ? 0x0000ffffu : 0u) | (tok + 1 < pend ? 0xffff0000u : 0u);
              vf[2 * jj + h2] &= msk;
            }
        }
        o[dt] = mfma16<T>(vf, pf, o[dt]);
      }
    };
    u32x4 kfa[2][NLD];
    VRaw vra[NDT][NVJ];
    int pr = pair0 + wave;
    // Fused form: the wave AFTER the owner of the last tile pair (the one with the fewest pairs in
    // the round-robin) writes the new token's K/V and raises an LDS flag; the owner polls the flag
    // right before it loads the last pair -- normally several tiles later, so it never waits.
    const bool has_last = !empty_run && pend == seq_len;       // this run holds the new token's pair
    const int owner_wave = has_last ? (pair_end - 1 - pair0) % NW : 0;
    const bool kv_owner = fused_rope && hb == 0 && has_last && wave == owner_wave;
    const bool kv_writer = fused_rope && hb == 0 && has_last && wave == (owner_wave + 1) % NW;
    // q of the fused form: issue the slab / cos-sin loads now (one rotary pair per thread)
    constexpr int QIT = (16 * (HD / 2) + NW * 64 - 1) / (NW * 64);   // pairs per thread for up to 16 heads
    float q_x[QIT], q_y[QIT];
    uint16_t q_c[QIT], q_s[QIT];
    if constexpr (fused_rope) {
      if (hb == 0) {
#pragma unroll
        for (int it = 0; it < QIT; ++it) {
          const int i = (int)threadIdx.x + it * NW * 64;
          q_x[it] = q_y[it] = 0.f;
          q_c[it] = q_s[it] = 0;
          if (i < nh * (HD / 2)) {
            const int h = i / (HD / 2), d = i % (HD / 2);
            const float* p0 = p.qkv_slabs + (size_t)seq * ntot + (kvh * gqa + h) * HD + d;
            float x = p0[0], y = p0[HD / 2];
            for (int k = 1; k < p.nslab; ++k) {
              x += p0[k * p.slab_stride];
              y += p0[k * p.slab_stride + HD / 2];
            }
            if constexpr (scaled_slabs) {
              const float rsc = p.slab_row_scale ? p.slab_row_scale[seq] : 1.f;
              const int col = (kvh * gqa + h) * HD + d;
              x = rsc * (p.slab_col_scale[col] * x);
              y = rsc * (p.slab_col_scale[col + HD / 2] * y);
              asm("" : "+v"(x), "+v"(y));
            }
            q_x[it] = x; q_y[it] = y;
            q_c[it] = cs_row[d];
            q_s[it] = cs_row[HD / 2 + d];
          }
        }
      }
    }
    // ---- Q fragments (B operand: lane (g, head) holds 8 consecutive d) --------
    // fused form: rotated from the qkv slabs INSIDE the first loop iteration, after that
    // iteration's K/V loads have been issued (their HBM latency covers the slab round trip)
    if constexpr (!fused_rope) {
      const uint16_t* qp = (const uint16_t*)p.q + (size_t)seq * p.q_stride + (size_t)head * HD;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        int d0;
        if constexpr (FP8) d0 = 64 * (ks >> 1) + 16 * g + 8 * (ks & 1);
        else d0 = 32 * ks + 8 * g;
        if (c < nh && d0 < HD) qf[ks] = *reinterpret_cast<const u32x4*>(qp + d0);
        else qf[ks] = u32x4{0, 0, 0, 0};
      }
    }

    // Rotated loop: the first pair's K/V loads are issued BEFORE the q phase of the fused form, so
    // their HBM latency covers it; afterwards load -> compute per pair as in the plain form.
    int ids[4] = {0, 0, 0, 0}, ids_next[4] = {0, 0, 0, 0};
    if (pr < pair_end) load_ids(pr, ids);
    if constexpr (!fused_rope) {
      for (; pr < pair_end; pr += NW) {
        load_pair(pr, ids, kfa, vra);
        if (pr + NW < pair_end) load_ids(pr + NW, ids_next);
        compute_pair(pr, kfa, vra);
#pragma unroll
        for (int e = 0; e < 4; ++e) ids[e] = ids_next[e];
      }
    }
    bool have = fused_rope && pr < pair_end;
    // (the owner of the last pair must not read it before the writer has stored the new token)
    const bool defer_first = kv_owner && have && pr + NW >= pair_end;
    if (have && !defer_first) load_pair(pr, ids, kfa, vra);
    if (have && pr + NW < pair_end) load_ids(pr + NW, ids_next);
    if constexpr (fused_rope) {
      // ---- q of the new token: slab reduce + rotary, ONCE per workgroup, one (d, d + hd/2) pair per
      // thread, through LDS.  The slab / cos-sin loads were issued above (before the K/V loads: the
      // vector L1 returns in order) -- see q_x / q_y.
      if (hb == 0) {
#pragma unroll
        for (int it = 0; it < QIT; ++it) {
          const int i = (int)threadIdx.x + it * NW * 64;
          if (i < nh * (HD / 2)) {
            const int h = i / (HD / 2), d = i % (HD / 2);
            float xo, yo;
            rope_pair(T::to_f32(T::from_f32(q_x[it])), T::to_f32(T::from_f32(q_y[it])), T::to_f32(q_c[it]),
                      T::to_f32(q_s[it]), xo, yo);
            q_lds[h * HD + d] = T::from_f32(xo);
            q_lds[h * HD + HD / 2 + d] = T::from_f32(yo);
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        int d0;
        if constexpr (FP8) d0 = 64 * (ks >> 1) + 16 * g + 8 * (ks & 1);
        else d0 = 32 * ks + 8 * g;
        if (c < nh && d0 < HD) qf[ks] = *reinterpret_cast<const u32x4*>(&q_lds[c * HD + d0]);
        else qf[ks] = u32x4{0, 0, 0, 0};
      }
      if (kv_writer) {  // after the barrier: nobody waits for this wave's extra round trip
        write_new_kv();
        __hip_atomic_store(&kv_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (defer_first) {
        while (__hip_atomic_load(&kv_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0)
          __builtin_amdgcn_s_sleep(1);
        load_pair(pr, ids, kfa, vra);
      }
    }
    while (have) {
      compute_pair(pr, kfa, vra);
      pr += NW;
      have = pr < pair_end;
      if (have) {
        if constexpr (fused_rope) {
          if (kv_owner && pr + NW >= pair_end) {  // the last pair holds the new token: wait for its writer
            while (__hip_atomic_load(&kv_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0)
              __builtin_amdgcn_s_sleep(1);
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) ids[e] = ids_next[e];
        load_pair(pr, ids, kfa, vra);
        if (pr + NW < pair_end) load_ids(pr + NW, ids_next);
      }
    }

    // ---- merge the NW waves through LDS ---------------------------------------------
    // layout: ml[NW][16][2] then ov[NW][16 heads][HD]
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    float* ml = lds;
    float* ov = lds + NW * 16 * 2;
    __syncthreads();  // previous head-block pass finished reading
    if (g == 0) {
      ml[(wave * 16 + c) * 2 + 0] = m_run;
      ml[(wave * 16 + c) * 2 + 1] = l_run;
    }
    if (c < nh) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        int d = 16 * dt + 4 * g;
        if (d < HD) *reinterpret_cast<f32x4*>(&ov[((size_t)wave * p.nh_lds + c) * HD + d]) = o[dt];
      }
    }
    __syncthreads();
    constexpr int SLOT = 16 * HD + 32;      // floats per (group, run) slot of split_scratch: O [16][HD], then (m, l) [16]
    float* const grp_scratch = split ? p.split_scratch + ((size_t)seq * p.num_kv_heads + kvh) * p.nsplit * SLOT : nullptr;
    if (split) {
      // ---- this run's unnormalised (O, m, l) -> scratch (write-through), ticket, and only the last arriver goes on -----
      if (!empty_run) {
        float* slot = grp_scratch + (size_t)part * SLOT;
        for (int idx = threadIdx.x; idx < nh * (HD / 4); idx += NW * 64) {      // 16 bytes per store
          const int h = idx / (HD / 4), d = 4 * (idx - h * (HD / 4));
          float M = -1e30f;
#pragma unroll
          for (int w = 0; w < NW; ++w) M = __builtin_fmaxf(M, ml[(w * 16 + h) * 2]);
          float L = 0.f;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int w = 0; w < NW; ++w) {
            float wt = __expf(ml[(w * 16 + h) * 2] - M);
            L += ml[(w * 16 + h) * 2 + 1] * wt;
            acc += *reinterpret_cast<const f32x4*>(&ov[((size_t)w * p.nh_lds + h) * HD + d]) * wt;
          }
          st_wt_f32x4(slot + h * HD + d, acc);
          if (d == 0) st_wt_f32x2(slot + 16 * HD + 2 * h, f32x2{M, L});
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my stores have been performed before the ticket is taken
      __syncthreads();
      if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(p.split_counter + (size_t)seq * p.num_kv_heads + kvh, 1u, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
        kv_flag = (old == (unsigned)p.nsplit - 1u) ? 2 : 3;
      }
      __syncthreads();
      if (kv_flag != 2) return;
      if (threadIdx.x == 0)   // everybody has arrived: ready for the next launch
        __hip_atomic_store(p.split_counter + (size_t)seq * p.num_kv_heads + kvh, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float out_amax = 0.f;
    for (int idx = threadIdx.x; idx < nh * HD; idx += NW * 64) {
      const int h = idx / HD, d = idx - h * HD;
      float M = -1e30f, L = 0.f, acc = 0.f;
      if (split) {
        // merge the runs (all loads of a thread in flight together, one wait): out = sum_z O_z e^(m_z - M) / (sum_z l_z e^(m_z - M) + 1e-6)
        float mz[8], lz[8], az[8];
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) {
          float m4[4] = {-1e30f, -1e30f, -1e30f, -1e30f}, l4[4] = {0.f, 0.f, 0.f, 0.f}, a4[4] = {0.f, 0.f, 0.f, 0.f};
          if (4 * q4 < p.nsplit) {            // (uniform) runs 4 q4 .. 4 q4 + 3; a run that does not exist reads run 0's slot
            const float* pm[4];
            const float* pa[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int z = 4 * q4 + j;
              const bool ok = z < p.nsplit && z * psz < seq_len;
              const float* slot = grp_scratch + (size_t)(ok ? z : 0) * SLOT;
              pm[j] = slot + 16 * HD + 2 * h;
              pa[j] = slot + h * HD + d;
            }
            ld_coh_runs4(m4, l4, a4, pm, pa);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int z = 4 * q4 + j;
              if (!(z < p.nsplit && z * psz < seq_len)) { m4[j] = -1e30f; l4[j] = 0.f; a4[j] = 0.f; }
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) { mz[4 * q4 + j] = m4[j]; lz[4 * q4 + j] = l4[j]; az[4 * q4 + j] = a4[j]; }
        }
#pragma unroll
        for (int z = 0; z < 8; ++z) M = __builtin_fmaxf(M, mz[z]);
#pragma unroll
        for (int z = 0; z < 8; ++z) {
          const float wt = __expf(mz[z] - M);
          L += lz[z] * wt;
          acc += az[z] * wt;
        }
      } else {
#pragma unroll
        for (int w = 0; w < NW; ++w) M = __builtin_fmaxf(M, ml[(w * 16 + h) * 2]);
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          float wt = __expf(ml[(w * 16 + h) * 2] - M);
          L += ml[(w * 16 + h) * 2 + 1] * wt;
          acc += ov[((size_t)w * p.nh_lds + h) * HD + d] * wt;
        }
      }
      const float res = acc * (1.f / (L + 1e-6f)) * p.v_scale;
      const int qh = kvh * gqa + hb + h;
      if (p.write_direct) {
        const typename T::storage r16 = T::from_f32(res);
        if (p.out) ((typename T::storage*)p.out)[((size_t)seq * p.num_heads + qh) * HD + d] = r16;
        if (p.out_packed) {
          const int k = qh * HD + d;  // column of the [num_seqs, Hq*hd] activation matrix
          const size_t chunk = ((((size_t)(k >> 7) * 4 + ((k & 31) >> 3)) * p.pack_mtiles + (seq >> 4)) * 64 +
                                ((k & 127) >> 5) * 16 + (seq & 15)) * 8;
          uint16_t h16;
          if constexpr (__is_same(T, Half)) h16 = r16;
          else h16 = bf16_bits_to_f16_bits_sat(r16);
          ((uint16_t*)p.out_packed)[chunk + (k & 7)] = h16;
        }
        if constexpr (ROPE == 2) {
          out_amax = __builtin_fmaxf(out_amax, __builtin_fabsf(T::to_f32(r16)));
          if (p.out_pairs) ((typename T::storage*)p.out_pairs)[aq_pair_offset(seq, qh * HD + d, p.pack_mtiles)] = r16;
          if (p.out_q8) {
            const float qv = __builtin_fmaxf(-448.f, __builtin_fminf(T::to_f32(r16) * q8_inv, 448.f));
            p.out_q8[((size_t)seq * p.num_heads + qh) * HD + d] = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(qv, qv, 0, false) & 0xff);
          }
        }
      } else {
        const size_t pi = ((size_t)seq * p.num_heads + qh) * p.max_parts + part;
        ((typename T::storage*)p.tmp_out)[pi * HD + d] = T::from_f32(res);
        if (d == 0) {
          p.exp_sums[pi] = L;
          p.max_logits[pi] = M;
        }
      }
    }
    if constexpr (ROPE == 2) {
      if (p.out_absmax) {                      // (uniform) one partial per (sequence, kv-head)
        __shared__ float amax_red[NW];
        out_amax = wave_max(out_amax);
        if (lane == 0) amax_red[wave] = out_amax;
        __syncthreads();
        if (threadIdx.x == 0) {
          float m = amax_red[0];
#pragma unroll
          for (int w = 1; w < NW; ++w) m = __builtin_fmaxf(m, amax_red[w]);
          p.out_absmax[(size_t)seq * p.num_kv_heads + kvh] = m;
        }
      }
    }
  }
}

// attention_kernels.cu:564-669: merge partitions.  grid (heads, seqs), block HD.
template <typename T, int HD>
__global__ void paged_attention_reduce_kernel(typename T::storage* __restrict__ out,
                                              const float* __restrict__ exp_sums,
                                              const float* __restrict__ max_logits,
                                              const typename T::storage* __restrict__ tmp_out,
                                              const int32_t* __restrict__ seq_lens, int num_heads,
                                              int max_parts, int partition_size) {
  const int head = blockIdx.x, seq = blockIdx.y;
  const int seq_len = seq_lens[seq];
  const int parts = (seq_len + partition_size - 1) / partition_size;
  const size_t base = ((size_t)seq * num_heads + head) * max_parts;
  float M = -1e30f;
  for (int j = 0; j < parts; ++j) M = __builtin_fmaxf(M, max_logits[base + j]);
  float L = 0.f;
  for (int j = 0; j < parts; ++j) L += exp_sums[base + j] * __expf(max_logits[base + j] - M);
  const float inv = 1.f / (L + 1e-6f);
  for (int d = threadIdx.x; d < HD; d += blockDim.x) {
    float acc = 0.f;
    for (int j = 0; j < parts; ++j)
      acc += T::to_f32(tmp_out[(base + j) * HD + d]) * exp_sums[base + j] * __expf(max_logits[base + j] - M);
    out[((size_t)seq * num_heads + head) * HD + d] = T::from_f32(parts > 0 ? acc * inv : 0.f);
  }
}

// ---------------------------------------------------------------------------
// cache write / fp8 convert
// ---------------------------------------------------------------------------
template <typename T, int KV>
__global__ void reshape_and_cache_kernel(const typename T::storage* __restrict__ key,
                                         const typename T::storage* __restrict__ value,
                                         void* __restrict__ key_cache, void* __restrict__ value_cache,
                                         const int64_t* __restrict__ slot_mapping, int num_kv_heads,
                                         int head_size, int block_size, int x, int64_t key_stride,
                                         int64_t value_stride, float k_scale, float v_scale) {
  const int64_t token = blockIdx.x;
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;  // padding (cache_kernels.cu:163-166)
  const int64_t blk = slot / block_size, off = slot % block_size;
  const int n = num_kv_heads * head_size;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int h = i / head_size, d = i % head_size;
    const int64_t kdst = (((blk * num_kv_heads + h) * (head_size / x) + d / x) * block_size + off) * x + d % x;
    const int64_t vdst = ((blk * num_kv_heads + h) * head_size + d) * block_size + off;
    if constexpr (KV == 0) {
      ((typename T::storage*)key_cache)[kdst] = key[token * key_stride + i];
      ((typename T::storage*)value_cache)[vdst] = value[token * value_stride + i];
    } else {
      float kf = T::to_f32(key[token * key_stride + i]) / k_scale;  // cache_kernels.cu:198-201
      float vf = T::to_f32(value[token * value_stride + i]) / v_scale;
      ((uint8_t*)key_cache)[kdst] = (uint8_t)f32x2_to_fp8<KV == 2>(kf, 0.f);
      ((uint8_t*)value_cache)[vdst] = (uint8_t)f32x2_to_fp8<KV == 2>(vf, 0.f);
    }
  }
}

// Prompt-sized form (round 6): one workgroup per (16 consecutive tokens, KV head).  The per-token kernel above stores every
// VALUE element on its own -- the V layout [blocks, Hkv, hd, block] puts a token's 128 values of a head into 128 different
// 32-byte rows: 8.4 M two-byte stores for an 8192-token Llama-3-8B prompt, 55 us per layer (profiles/r6_prefill_e2e_trace.txt).
// Here the 16 tokens' K and V rows of one head are staged through LDS (16-byte loads, 256 B per token and head) and, when the
// 16 slots are 16 consecutive places of ONE block -- what a prompt's slot mapping is, block after block -- leave as whole
// 16-byte pieces: K [hd/x][block][x]: piece (d/x, token) -> 256 contiguous bytes per d/x over the 16 tokens; V [hd][block]:
// a row's 16 tokens are 32 (fp8: 16) contiguous bytes.  Any other window (a sequence boundary, padding, a shuffled mapping)
// takes the element-wise stores of the kernel above for its tokens.  Same conversions, same bits.  2-byte inputs only.
template <typename T, int KV>
__global__ __launch_bounds__(256) void reshape_and_cache_window_kernel(
    const typename T::storage* __restrict__ key, const typename T::storage* __restrict__ value, void* __restrict__ key_cache,
    void* __restrict__ value_cache, const int64_t* __restrict__ slot_mapping, int64_t num_tokens, int num_kv_heads, int head_size,
    int block_size, int64_t key_stride, int64_t value_stride, float k_scale, float v_scale) {
  static_assert(sizeof(typename T::storage) == 2, "2-byte activations");
  using S = typename T::storage;
  constexpr int W = 16;                               // tokens per window
  constexpr int X = KV == 0 ? 8 : 16;                 // elements per 16-byte cache piece
  extern __shared__ __attribute__((aligned(16))) unsigned char rc_smem[];
  const int D = head_size, P = D + 8;                 // LDS row pitch in elements (+ 16 B: the column reads spread over the banks)
  S* sk = reinterpret_cast<S*>(rc_smem);
  S* sv = sk + W * P;
  __shared__ int64_t s_slot[W];
  __shared__ int s_fast;
  const int64_t t0 = (int64_t)blockIdx.x * W;
  const int h = blockIdx.y;
  const int nt = (int)(num_tokens - t0 < W ? num_tokens - t0 : W);
  if (threadIdx.x < W) s_slot[threadIdx.x] = threadIdx.x < nt ? slot_mapping[t0 + threadIdx.x] : -1;
  __syncthreads();
  if (threadIdx.x == 0) {
    bool f = nt == W && s_slot[0] >= 0 && (s_slot[0] % block_size) + W <= block_size;
    for (int j = 1; j < W && f; ++j) f = s_slot[j] == s_slot[0] + j;
    s_fast = f ? 1 : 0;
  }
  // stage: 16-byte pieces of the tokens' K / V rows of this head
  const int pieces = D / 8;
  for (int i = threadIdx.x; i < nt * pieces; i += blockDim.x) {
    const int j = i / pieces, pc = i % pieces;
    *reinterpret_cast<u32x4*>(sk + j * P + pc * 8) = *reinterpret_cast<const u32x4*>(key + (t0 + j) * key_stride + h * D + pc * 8);
    *reinterpret_cast<u32x4*>(sv + j * P + pc * 8) = *reinterpret_cast<const u32x4*>(value + (t0 + j) * value_stride + h * D + pc * 8);
  }
  __syncthreads();
  auto cvt = [](S e, float scale) -> uint8_t {        // cache_kernels.cu:198-201 (the division, not a reciprocal)
    return (uint8_t)f32x2_to_fp8<KV == 2>(T::to_f32(e) / scale, 0.f);
  };
  if (s_fast) {
    const int64_t blk = s_slot[0] / block_size;
    const int off0 = (int)(s_slot[0] % block_size);
    // K: piece (px = d / X, token j) -> ((blk * H + h) * (D / X) + px) * block + off0 + j, X elements each
    for (int i = threadIdx.x; i < (D / X) * W; i += blockDim.x) {
      const int px = i / W, j = i % W;
      const int64_t dst = (((blk * num_kv_heads + h) * (D / X) + px) * block_size + off0 + j) * X;
      if constexpr (KV == 0) {
        *reinterpret_cast<u32x4*>((S*)key_cache + dst) = *reinterpret_cast<const u32x4*>(sk + j * P + px * 8);
      } else {
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t v = 0;
#pragma unroll
          for (int b = 0; b < 4; ++b) v |= (uint32_t)cvt(sk[j * P + px * 16 + 4 * q + b], k_scale) << (8 * b);
          w[q] = v;
        }
        *reinterpret_cast<u32x4*>((uint8_t*)key_cache + dst) = u32x4{w[0], w[1], w[2], w[3]};
      }
    }
    // V: row d -> ((blk * H + h) * D + d) * block + off0 .. + 15: 16 tokens = 32 B (two 16-byte pieces) / fp8: 16 B (one)
    constexpr int VP = KV == 0 ? 2 : 1;               // 16-byte pieces per row
    for (int i = threadIdx.x; i < D * VP; i += blockDim.x) {
      const int d = i / VP, half = i % VP;
      const int64_t dst = ((blk * num_kv_heads + h) * D + d) * block_size + off0;
      if constexpr (KV == 0) {
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          w[q] = (uint32_t)sv[(half * 8 + 2 * q) * P + d] | ((uint32_t)sv[(half * 8 + 2 * q + 1) * P + d] << 16);
        *reinterpret_cast<u32x4*>((S*)value_cache + dst + half * 8) = u32x4{w[0], w[1], w[2], w[3]};
      } else {
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t v = 0;
#pragma unroll
          for (int b = 0; b < 4; ++b) v |= (uint32_t)cvt(sv[(4 * q + b) * P + d], v_scale) << (8 * b);
          w[q] = v;
        }
        *reinterpret_cast<u32x4*>((uint8_t*)value_cache + dst) = u32x4{w[0], w[1], w[2], w[3]};
      }
    }
    return;
  }
  // any other window: the per-token kernel's stores for this head
  for (int i = threadIdx.x; i < nt * D; i += blockDim.x) {
    const int j = i / D, d = i % D;
    const int64_t slot = s_slot[j];
    if (slot < 0) continue;                           // padding (cache_kernels.cu:163-166)
    const int64_t blk = slot / block_size, off = slot % block_size;
    const int64_t kdst = (((blk * num_kv_heads + h) * (D / X) + d / X) * block_size + off) * X + d % X;
    const int64_t vdst = ((blk * num_kv_heads + h) * D + d) * block_size + off;
    if constexpr (KV == 0) {
      ((S*)key_cache)[kdst] = sk[j * P + d];
      ((S*)value_cache)[vdst] = sv[j * P + d];
    } else {
      ((uint8_t*)key_cache)[kdst] = cvt(sk[j * P + d], k_scale);
      ((uint8_t*)value_cache)[vdst] = cvt(sv[j * P + d], v_scale);
    }
  }
}

template <typename T, int KV, bool TO_FP8>
__global__ void convert_fp8_kernel(void* __restrict__ dst, const void* __restrict__ src, int64_t n,
                                   float scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if constexpr (TO_FP8) {
      float f = T::to_f32(((const typename T::storage*)src)[i]) / scale;
      ((uint8_t*)dst)[i] = (uint8_t)f32x2_to_fp8<KV == 2>(f, 0.f);
    } else {
      float f = fp8_to_f32<KV == 2>(((const uint8_t*)src)[i]) * scale;
      ((typename T::storage*)dst)[i] = T::from_f32(f);
    }
  }
}

// ---------------------------------------------------------------------------
template <typename T, int KV, int HD, int BS, int ROPE = 0>
static int launch_pa(const PAParams& p, int num_seqs, int parts, int nw, hipStream_t st) {
  dim3 grid((unsigned)p.num_kv_heads, (unsigned)num_seqs, (unsigned)parts);
  size_t lds = ((size_t)nw * 16 * 2 + (size_t)nw * p.nh_lds * HD) * sizeof(float);
#define APHRO_PA_LAUNCH(NWV, SPL)                                                                   \
  {                                                                                                 \
    auto kern = paged_attention_kernel<T, KV, HD, BS, NWV, ROPE, SPL>;                              \
    if (lds > 64 * 1024)                                                                            \
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kern, grid, dim3(NWV * 64), lds, st, p.seq_lens, p.block_tables, p.kc, p.vc, p.qkv_slabs,        \
                       p.num_heads, p.num_kv_heads, p.max_blocks_per_seq, p.partition_size, p);                  \
  }
  if constexpr (HD == 128 && (BS == 16 || BS == 32)) {   // the split form is instantiated for the serving geometry only
    if (p.nsplit > 1) {
      if (nw == 4) APHRO_PA_LAUNCH(4, true)
      else APHRO_PA_LAUNCH(8, true)
      APHRO_LAUNCH_CHECK();
      return APHRO_OK;
    }
  }
  if (nw == 4) APHRO_PA_LAUNCH(4, false)
  else APHRO_PA_LAUNCH(8, false)
#undef APHRO_PA_LAUNCH
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

template <typename T, int KV>
static int dispatch_pa(const PAParams& p, int num_seqs, int parts, int nw, int head_size, int block_size,
                       hipStream_t st) {
  if (p.qkv_slabs != nullptr) {  // fused rotary + cache write: head_size 128 only (checked by the caller)
    if (p.slab_col_scale != nullptr) {
      if (block_size == 16) return launch_pa<T, KV, 128, 16, 2>(p, num_seqs, parts, nw, st);
      if (block_size == 32) return launch_pa<T, KV, 128, 32, 2>(p, num_seqs, parts, nw, st);
    } else {
      if (block_size == 16) return launch_pa<T, KV, 128, 16, 1>(p, num_seqs, parts, nw, st);
      if (block_size == 32) return launch_pa<T, KV, 128, 32, 1>(p, num_seqs, parts, nw, st);
    }
    set_error("paged_attention: the fused rotary form supports block_size 16 / 32");
    return APHRO_ERR_INVALID;
  }
#define APHRO_PA_CASE(HDV, BSV) \
  if (head_size == HDV && block_size == BSV) return launch_pa<T, KV, HDV, BSV>(p, num_seqs, parts, nw, st);
  APHRO_PA_CASE(128, 16)
  APHRO_PA_CASE(128, 32)
  APHRO_PA_CASE(64, 16)
  APHRO_PA_CASE(64, 32)
  APHRO_PA_CASE(96, 16)
  APHRO_PA_CASE(80, 16)
  APHRO_PA_CASE(112, 16)
  APHRO_PA_CASE(192, 16)
  APHRO_PA_CASE(256, 16)
  APHRO_PA_CASE(128, 8)
  APHRO_PA_CASE(64, 8)
#undef APHRO_PA_CASE
  set_error("paged_attention: unsupported head_size=%d / block_size=%d", head_size, block_size);
  return APHRO_ERR_INVALID;
}

}  // namespace aphro

using namespace aphro;

// Per-device workspace of the split form: tickets (zero between launches) + the runs' partial results.  Owned by the
// library (hipMalloc on first use / growth, never during a stream capture: a capturing call that finds it too small runs
// unsplit).  The decode attention launches of a process are stream-ordered (one compute stream per worker, as in the
// reference), so one workspace per device suffices.
struct SplitWs { unsigned* counter = nullptr; size_t groups = 0; float* scratch = nullptr; size_t floats = 0; };
static SplitWs g_split_ws[APHRO_MAX_DEVICES];

// Buffers are NEVER freed or moved once handed out: a HIP graph captured earlier has their raw pointers baked into its
// kernel nodes, and the reference's model runner captures many batch sizes and keeps running eager decode beside them
// (growth after capture used to hipFree what a captured graph still writes to).  The first use allocates for
// the largest plan the launcher makes by itself (groups x want < 3 CUs, want <= 8); a forced APHRO_PA_SPLITS plan that
// needs more gets NEW, larger buffers and the old ones stay alive for the life of the process.
static bool split_workspace(size_t groups, size_t floats, hipStream_t st, SplitWs** ws_out) {
  static std::mutex mu;                        // host threads racing on first use / growth
  std::lock_guard<std::mutex> lock(mu);
  SplitWs& ws = g_split_ws[device_slot()];
  if (ws.groups < groups || ws.floats < floats) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
    if (ws.groups < groups) {
      const size_t g2 = groups < 4096 ? 4096 : groups * 2;
      unsigned* fresh = nullptr;
      if (hipMalloc((void**)&fresh, g2 * sizeof(unsigned)) != hipSuccess) return false;
      if (hipMemset(fresh, 0, g2 * sizeof(unsigned)) != hipSuccess) { (void)hipFree(fresh); return false; }
      ws.counter = fresh;                      // (the previous array, if any, is retired, not freed)
      ws.groups = g2;
    }
    if (ws.floats < floats) {
      const size_t bound = (size_t)3 * device_cu_count() * (16 * 128 + 32);
      const size_t f2 = floats < bound ? bound : floats * 2;
      float* fresh = nullptr;
      if (hipMalloc((void**)&fresh, f2 * sizeof(float)) != hipSuccess) return false;
      ws.scratch = fresh;
      ws.floats = f2;
    }
  }
  *ws_out = &ws;
  return true;
}

static int paged_attention_impl(void* out, void* out_packed, float* exp_sums, float* max_logits, void* tmp_out,
                                     const void* query, const void* key_cache, const void* value_cache,
                                     int num_seqs, int num_heads, int num_kv_heads, int head_size,
                                     float scale, const int32_t* block_tables, const int32_t* seq_lens,
                                     int max_num_blocks_per_seq, int block_size, int max_seq_len,
                                     const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                                     int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale,
                                     float v_scale, int partition_size, void* stream,
                                     const float* qkv_slabs = nullptr, int nslab = 0,
                                     const int64_t* positions = nullptr, const void* cos_sin = nullptr,
                                     const int64_t* slot_mapping = nullptr, const float* slab_row_scale = nullptr,
                                     const float* slab_col_scale = nullptr, void* out_q8 = nullptr,
                                     const float* out_q8_scale = nullptr, float* out_absmax = nullptr,
                                     void* out_pairs = nullptr) {
  hipStream_t st = (hipStream_t)stream;
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "paged_attention: query dtype must be f16 or bf16");
  APHRO_CHECK(kv_dtype >= APHRO_KV_AUTO && kv_dtype <= APHRO_KV_FP8_E5M2, "Unsupported data type of kv cache: %d", kv_dtype);
  APHRO_CHECK(num_kv_heads > 0 && num_heads % num_kv_heads == 0, "paged_attention: num_heads %% num_kv_heads != 0");
  APHRO_CHECK(partition_size >= 0 && partition_size % 32 == 0, "paged_attention: partition size must be a multiple of 32");
  APHRO_CHECK(qkv_slabs != nullptr || (q_stride % 8 == 0 && ((uintptr_t)query % 16) == 0),
              "paged_attention: query must be 16-byte aligned");
  if (num_seqs == 0) return APHRO_OK;
  PAParams p;
  p.qkv_slabs = qkv_slabs; p.nslab = nslab; p.positions = positions; p.cos_sin = (const uint16_t*)cos_sin;
  p.slot_mapping = slot_mapping;
  p.slab_stride = (int64_t)num_seqs * (num_heads + 2 * num_kv_heads) * head_size;
  p.k_scale_raw = k_scale; p.v_scale_raw = v_scale;
  p.slab_row_scale = slab_row_scale; p.slab_col_scale = slab_col_scale;
  if (qkv_slabs != nullptr) {
    APHRO_CHECK(partition_size == 0 && head_size == 128 && nslab >= 1 && cos_sin && slot_mapping &&
                    num_heads / num_kv_heads <= 16,
                "paged_attention: the fused rotary + cache-write form needs the v1 form, head_size 128 and GQA <= 16");
    APHRO_CHECK(kv_head_stride == (int64_t)head_size * block_size, "paged_attention: fused form needs contiguous caches");
  }
  p.out_packed = out_packed; p.pack_mtiles = (num_seqs + 15) / 16;
  APHRO_CHECK(out_packed == nullptr || (partition_size == 0 && ((int64_t)num_heads * head_size) % 128 == 0),
              "paged_attention: packed output needs the single-kernel (v1) form and Hq*hd %% 128 == 0");
  p.out_q8 = (uint8_t*)out_q8; p.out_q8_scale = out_q8_scale;
  p.out_absmax = out_absmax; p.out_pairs = out_pairs;
  APHRO_CHECK(out_absmax == nullptr || (partition_size == 0 && qkv_slabs != nullptr && slab_col_scale != nullptr &&
                                        (out != nullptr || out_pairs != nullptr) && num_heads / num_kv_heads <= 16),
              "paged_attention: the absmax partials need the fused scaled-slab form, a 16-bit output and GQA <= 16");
  APHRO_CHECK(out_pairs == nullptr || (out_absmax != nullptr && ((int64_t)num_heads * head_size) % 64 == 0),
              "paged_attention: the pair-major output goes with the absmax partials (Hq * hd %% 64 == 0)");
  APHRO_CHECK(out_q8 == nullptr || (partition_size == 0 && out_q8_scale != nullptr && qkv_slabs != nullptr && slab_col_scale != nullptr),
              "paged_attention: the fp8 output needs the fused scaled-slab form and its scale");
  p.out = out; p.exp_sums = exp_sums; p.max_logits = max_logits; p.tmp_out = tmp_out;
  p.q = query; p.kc = key_cache; p.vc = value_cache;
  p.block_tables = block_tables; p.seq_lens = seq_lens; p.alibi = alibi_slopes;
  p.num_heads = num_heads; p.num_kv_heads = num_kv_heads;
  p.max_blocks_per_seq = max_num_blocks_per_seq;
  p.scale = scale * (kv_dtype == APHRO_KV_AUTO ? 1.f : k_scale);
  p.v_scale = kv_dtype == APHRO_KV_AUTO ? 1.f : v_scale;
  p.q_stride = q_stride; p.kv_block_stride = kv_block_stride; p.kv_head_stride = kv_head_stride;
  const int gqa = num_heads / num_kv_heads;
  p.nh_lds = gqa < 16 ? gqa : 16;
  int parts = 1, nw;
  p.nsplit = 0; p.split_scratch = nullptr; p.split_counter = nullptr;
  if (partition_size == 0) {
    p.partition_size = 0; p.max_parts = 0; p.write_direct = 1;
    // v1 form: one workgroup per (seq, kv head) walks the whole sequence
    nw = max_seq_len > 256 ? 8 : 4;
    while (nw > 4 && (size_t)nw * (32 + p.nh_lds * head_size) * 4 > 96 * 1024) nw >>= 1;
    // ... or nsplit workgroups share it (split-KV inside the launch) when (seq, kv head) groups alone leave CUs idle:
    // aim at ~2 workgroups per CU, at least 8 pairs (256 tokens) per run.  APHRO_PA_SPLITS=n forces n (1 = off).
    if (gqa <= 16 && head_size == 128 && (block_size == 16 || block_size == 32)) {
      const int64_t groups = (int64_t)num_seqs * num_kv_heads;
      const int cus = device_cu_count();
      int want = groups >= cus ? 1 : (int)((2 * cus + groups - 1) / groups);
      const int max_pairs = (max_seq_len + 31) / 32;
      if (want > max_pairs / 8) want = max_pairs / 8;
      if (knobs().pa_splits > 0) want = knobs().pa_splits;
      want = want > 8 ? 8 : want;
      if (want > 1) {
        SplitWs* ws = nullptr;
        if (split_workspace((size_t)groups, (size_t)groups * want * (16 * head_size + 32), st, &ws)) {
          p.nsplit = want; p.split_scratch = ws->scratch; p.split_counter = ws->counter;
          parts = want;
        }
      }
    }
  } else {
    APHRO_CHECK(exp_sums && max_logits && tmp_out, "paged_attention: partitioned form needs scratch tensors");
    parts = (max_seq_len + partition_size - 1) / partition_size;
    if (parts < 1) parts = 1;
    p.partition_size = partition_size; p.max_parts = parts;
    p.write_direct = parts == 1;
    nw = 4;
  }
  int rc;
#define APHRO_PA_KV(TT)                                                                                 \
  (kv_dtype == APHRO_KV_AUTO       ? dispatch_pa<TT, 0>(p, num_seqs, parts, nw, head_size, block_size, st) \
   : kv_dtype == APHRO_KV_FP8_E4M3 ? dispatch_pa<TT, 1>(p, num_seqs, parts, nw, head_size, block_size, st) \
                                   : dispatch_pa<TT, 2>(p, num_seqs, parts, nw, head_size, block_size, st))
  rc = dtype == APHRO_F16 ? APHRO_PA_KV(Half) : APHRO_PA_KV(BFloat);
#undef APHRO_PA_KV
  if (rc != APHRO_OK) return rc;
  if (!p.write_direct) {
    dim3 grid((unsigned)num_heads, (unsigned)num_seqs);
    int threads = head_size <= 64 ? 64 : 128;
#define APHRO_RED(TT, HDV)                                                                      \
  hipLaunchKernelGGL((paged_attention_reduce_kernel<TT, HDV>), grid, dim3(threads), 0, st,       \
                     (typename TT::storage*)out, exp_sums, max_logits,                          \
                     (const typename TT::storage*)tmp_out, seq_lens, num_heads, parts, partition_size)
#define APHRO_RED_HD(TT)                                     \
  switch (head_size) {                                       \
    case 64: APHRO_RED(TT, 64); break;                       \
    case 80: APHRO_RED(TT, 80); break;                       \
    case 96: APHRO_RED(TT, 96); break;                       \
    case 112: APHRO_RED(TT, 112); break;                     \
    case 128: APHRO_RED(TT, 128); break;                     \
    case 192: APHRO_RED(TT, 192); break;                     \
    default: APHRO_RED(TT, 256); break;                      \
  }
    if (dtype == APHRO_F16) { APHRO_RED_HD(Half) } else { APHRO_RED_HD(BFloat) }
#undef APHRO_RED_HD
#undef APHRO_RED
    APHRO_LAUNCH_CHECK();
  }
  return APHRO_OK;
}

extern "C" int aphro_paged_attention(void* out, float* exp_sums, float* max_logits, void* tmp_out,
                                     const void* query, const void* key_cache, const void* value_cache,
                                     int num_seqs, int num_heads, int num_kv_heads, int head_size,
                                     float scale, const int32_t* block_tables, const int32_t* seq_lens,
                                     int max_num_blocks_per_seq, int block_size, int max_seq_len,
                                     const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                                     int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale,
                                     float v_scale, int partition_size, void* stream) {
  return paged_attention_impl(out, nullptr, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs,
                              num_heads, num_kv_heads, head_size, scale, block_tables, seq_lens,
                              max_num_blocks_per_seq, block_size, max_seq_len, alibi_slopes, q_stride,
                              kv_block_stride, kv_head_stride, dtype, kv_dtype, k_scale, v_scale, partition_size,
                              stream);
}

// v1-form decode attention whose output is (also) written fragment-major for the
// o_proj GEMM of the decode fast path; `out` may be NULL.
extern "C" int aphro_paged_attention_packed(void* out, void* out_packed, const void* query, const void* key_cache,
                                            const void* value_cache, int num_seqs, int num_heads,
                                            int num_kv_heads, int head_size, float scale,
                                            const int32_t* block_tables, const int32_t* seq_lens,
                                            int max_num_blocks_per_seq, int block_size, int max_seq_len,
                                            const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                                            int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale,
                                            float v_scale, void* stream) {
  return paged_attention_impl(out, out_packed, nullptr, nullptr, nullptr, query, key_cache, value_cache, num_seqs,
                              num_heads, num_kv_heads, head_size, scale, block_tables, seq_lens,
                              max_num_blocks_per_seq, block_size, max_seq_len, alibi_slopes, q_stride,
                              kv_block_stride, kv_head_stride, dtype, kv_dtype, k_scale, v_scale, 0, stream);
}

// Decode fast path: [qkv split-K slab reduce] + rotary_embedding (NeoX, rot_dim == head_size) +
// reshape_and_cache + paged attention (v1 form) + fragment-major output, one kernel.
// key_cache / value_cache are WRITTEN (slot_mapping) before they are read.
extern "C" int aphro_paged_attention_rope_packed(void* out, void* out_packed, const float* qkv_slabs, int nslab,
                                                 const int64_t* positions, const void* cos_sin_cache,
                                                 const int64_t* slot_mapping, void* key_cache, void* value_cache,
                                                 int num_seqs, int num_heads, int num_kv_heads, int head_size,
                                                 float scale, const int32_t* block_tables,
                                                 const int32_t* seq_lens, int max_num_blocks_per_seq,
                                                 int block_size, int max_seq_len, const float* alibi_slopes,
                                                 int64_t kv_block_stride, int64_t kv_head_stride, int dtype,
                                                 int kv_dtype, float k_scale, float v_scale, void* stream) {
  APHRO_CHECK(qkv_slabs != nullptr, "paged_attention_rope_packed: qkv_slabs is NULL");
  return paged_attention_impl(out, out_packed, nullptr, nullptr, nullptr, nullptr, key_cache, value_cache,
                              num_seqs, num_heads, num_kv_heads, head_size, scale, block_tables, seq_lens,
                              max_num_blocks_per_seq, block_size, max_seq_len, alibi_slopes, 0, kv_block_stride,
                              kv_head_stride, dtype, kv_dtype, k_scale, v_scale, 0, stream, qkv_slabs, nslab,
                              positions, cos_sin_cache, slot_mapping);
}

// Same, for a quantised (FP8 W8A8) qkv projection: the slabs hold raw accumulators and are
// dequantised on the fly, value = row_scale[seq] * (col_scale[c] * sum) (cutlass_scaled_mm epilogue
// order), before the rounding to the activation dtype.
extern "C" int aphro_paged_attention_rope_packed_scaled(void* out, void* out_packed, const float* qkv_slabs,
                                                        int nslab, const float* slab_row_scale,
                                                        const float* slab_col_scale, const int64_t* positions,
                                                        const void* cos_sin_cache, const int64_t* slot_mapping,
                                                        void* key_cache, void* value_cache, int num_seqs,
                                                        int num_heads, int num_kv_heads, int head_size, float scale,
                                                        const int32_t* block_tables, const int32_t* seq_lens,
                                                        int max_num_blocks_per_seq, int block_size,
                                                        int max_seq_len, const float* alibi_slopes,
                                                        int64_t kv_block_stride, int64_t kv_head_stride, int dtype,
                                                        int kv_dtype, float k_scale, float v_scale, void* stream) {
  APHRO_CHECK(qkv_slabs != nullptr && slab_col_scale != nullptr, "paged_attention_rope_packed_scaled: NULL slabs / scales");
  return paged_attention_impl(out, out_packed, nullptr, nullptr, nullptr, nullptr, key_cache, value_cache,
                              num_seqs, num_heads, num_kv_heads, head_size, scale, block_tables, seq_lens,
                              max_num_blocks_per_seq, block_size, max_seq_len, alibi_slopes, 0, kv_block_stride,
                              kv_head_stride, dtype, kv_dtype, k_scale, v_scale, 0, stream, qkv_slabs, nslab,
                              positions, cos_sin_cache, slot_mapping, slab_row_scale, slab_col_scale);
}

// ... and the output ALSO (or only: out may be NULL) as e4m3 with a static per-tensor scale for an FP8 o_proj whose
// checkpoint carries input_scale: out_q8[seq, head, d] = fp8(T(out) * (1 / *out_q8_scale)) -- static_scaled_fp8_quant's
// bits, without its launch.
extern "C" int aphro_paged_attention_rope_scaled_q8(void* out, void* out_q8, const float* out_q8_scale,
                                                    const float* qkv_slabs, int nslab, const float* slab_row_scale,
                                                    const float* slab_col_scale, const int64_t* positions,
                                                    const void* cos_sin_cache, const int64_t* slot_mapping,
                                                    void* key_cache, void* value_cache, int num_seqs, int num_heads,
                                                    int num_kv_heads, int head_size, float scale,
                                                    const int32_t* block_tables, const int32_t* seq_lens,
                                                    int max_num_blocks_per_seq, int block_size, int max_seq_len,
                                                    const float* alibi_slopes, int64_t kv_block_stride,
                                                    int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale,
                                                    float v_scale, void* stream) {
  APHRO_CHECK(qkv_slabs != nullptr && slab_col_scale != nullptr && out_q8 != nullptr && out_q8_scale != nullptr,
              "paged_attention_rope_scaled_q8: NULL slabs / scales / output");
  return paged_attention_impl(out, nullptr, nullptr, nullptr, nullptr, nullptr, key_cache, value_cache, num_seqs,
                              num_heads, num_kv_heads, head_size, scale, block_tables, seq_lens,
                              max_num_blocks_per_seq, block_size, max_seq_len, alibi_slopes, 0, kv_block_stride,
                              kv_head_stride, dtype, kv_dtype, k_scale, v_scale, 0, stream, qkv_slabs, nslab, positions,
                              cos_sin_cache, slot_mapping, slab_row_scale, slab_col_scale, out_q8, out_q8_scale);
}

// ... and, for an FP8 o_proj with DYNAMIC per-token activation scales, the absmax of every (sequence, kv-head) slice of `out`
// (out_absmax [num_seqs][num_kv_heads]; `out` row-major and / or out_pairs in the pair-major layout of common.h
// aq_pair_offset over [num_seqs, Hq * hd]): the GEMM that reads it reduces the partials to the row scale and quantises on load
// (aphro_fp8_gemm_resident_aq) -- dynamic_per_token_scaled_fp8_quant (fp8/common.cu:201-256) without its launch.
extern "C" int aphro_paged_attention_rope_scaled_absmax(void* out, void* out_pairs, float* out_absmax, const float* qkv_slabs, int nslab,
                                                        const float* slab_row_scale, const float* slab_col_scale,
                                                        const int64_t* positions, const void* cos_sin_cache,
                                                        const int64_t* slot_mapping, void* key_cache, void* value_cache,
                                                        int num_seqs, int num_heads, int num_kv_heads, int head_size,
                                                        float scale, const int32_t* block_tables, const int32_t* seq_lens,
                                                        int max_num_blocks_per_seq, int block_size, int max_seq_len,
                                                        const float* alibi_slopes, int64_t kv_block_stride,
                                                        int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale,
                                                        float v_scale, void* stream) {
  APHRO_CHECK(qkv_slabs != nullptr && slab_col_scale != nullptr && (out != nullptr || out_pairs != nullptr) && out_absmax != nullptr,
              "paged_attention_rope_scaled_absmax: NULL slabs / scales / outputs");
  return paged_attention_impl(out, nullptr, nullptr, nullptr, nullptr, nullptr, key_cache, value_cache, num_seqs,
                              num_heads, num_kv_heads, head_size, scale, block_tables, seq_lens,
                              max_num_blocks_per_seq, block_size, max_seq_len, alibi_slopes, 0, kv_block_stride,
                              kv_head_stride, dtype, kv_dtype, k_scale, v_scale, 0, stream, qkv_slabs, nslab, positions,
                              cos_sin_cache, slot_mapping, slab_row_scale, slab_col_scale, nullptr, nullptr, out_absmax, out_pairs);
}

extern "C" int aphro_reshape_and_cache(const void* key, const void* value, void* key_cache,
                                       void* value_cache, const int64_t* slot_mapping, int64_t num_tokens,
                                       int num_kv_heads, int head_size, int block_size, int x,
                                       int64_t key_stride, int64_t value_stride, int dtype, int kv_dtype,
                                       float k_scale, float v_scale, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APHRO_CHECK(kv_dtype >= APHRO_KV_AUTO && kv_dtype <= APHRO_KV_FP8_E5M2, "Unsupported data type of kv cache: %d", kv_dtype);
  APHRO_CHECK(dtype >= APHRO_F16 && dtype <= APHRO_F32, "Unsupported input type of kv cache: %d", dtype);
  APHRO_CHECK(x > 0 && head_size % x == 0, "reshape_and_cache: head_size %% x != 0");
  if (num_tokens == 0) return APHRO_OK;
  int n = num_kv_heads * head_size;
  // prompt-sized calls on 2-byte activations: the 16-token window form (whole 16-byte stores where the slots run through a block)
  const int xe = kv_dtype == APHRO_KV_AUTO ? 8 : 16;
  if (num_tokens >= 64 && (dtype == APHRO_F16 || dtype == APHRO_BF16) && x == xe && head_size % 16 == 0 && head_size <= 512 &&
      block_size % 16 == 0 && key_stride % 8 == 0 && value_stride % 8 == 0 && ((uintptr_t)key % 16) == 0 && ((uintptr_t)value % 16) == 0 &&
      ((uintptr_t)key_cache % 16) == 0 && ((uintptr_t)value_cache % 16) == 0) {
    const dim3 wgrid((unsigned)((num_tokens + 15) / 16), (unsigned)num_kv_heads);
    const size_t lds = (size_t)2 * 16 * (head_size + 8) * 2;
#define APHRO_RCW(TT, KVV)                                                                             \
    hipLaunchKernelGGL((reshape_and_cache_window_kernel<TT, KVV>), wgrid, dim3(256), lds, st,          \
                       (const typename TT::storage*)key, (const typename TT::storage*)value, key_cache, \
                       value_cache, slot_mapping, num_tokens, num_kv_heads, head_size, block_size, key_stride, value_stride, k_scale, v_scale)
#define APHRO_RCW_T(KVV) if (dtype == APHRO_F16) APHRO_RCW(Half, KVV); else APHRO_RCW(BFloat, KVV);
    if (kv_dtype == APHRO_KV_AUTO) { APHRO_RCW_T(0) }
    else if (kv_dtype == APHRO_KV_FP8_E4M3) { APHRO_RCW_T(1) }
    else { APHRO_RCW_T(2) }
#undef APHRO_RCW_T
#undef APHRO_RCW
    APHRO_LAUNCH_CHECK();
    return APHRO_OK;
  }
  dim3 grid((unsigned)num_tokens), block((unsigned)(n < 512 ? (n + 63) / 64 * 64 : 512));
#define APHRO_RC(TT, KVV)                                                                           \
  hipLaunchKernelGGL((reshape_and_cache_kernel<TT, KVV>), grid, block, 0, st,                        \
                     (const typename TT::storage*)key, (const typename TT::storage*)value, key_cache, \
                     value_cache, slot_mapping, num_kv_heads, head_size, block_size, x, key_stride,   \
                     value_stride, k_scale, v_scale)
#define APHRO_RC_T(KVV)                                  \
  if (dtype == APHRO_F16) APHRO_RC(Half, KVV);           \
  else if (dtype == APHRO_BF16) APHRO_RC(BFloat, KVV);   \
  else APHRO_RC(Float, KVV);
  if (kv_dtype == APHRO_KV_AUTO) { APHRO_RC_T(0) }
  else if (kv_dtype == APHRO_KV_FP8_E4M3) { APHRO_RC_T(1) }
  else { APHRO_RC_T(2) }
#undef APHRO_RC_T
#undef APHRO_RC
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_convert_fp8(void* dst, const void* src, int64_t numel, float scale, int hp_dtype,
                                 int kv_dtype, int to_fp8, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APHRO_CHECK(kv_dtype == APHRO_KV_FP8_E4M3 || kv_dtype == APHRO_KV_FP8_E5M2, "Unsupported data type of kv cache: %d", kv_dtype);
  APHRO_CHECK(hp_dtype >= APHRO_F16 && hp_dtype <= APHRO_F32, "convert_fp8: unsupported dtype %d", hp_dtype);
  if (numel == 0) return APHRO_OK;
  unsigned blocks = (unsigned)((numel + 255) / 256 < 4096 ? (numel + 255) / 256 : 4096);
#define APHRO_CV(TT, KVV, DIR) \
  hipLaunchKernelGGL((convert_fp8_kernel<TT, KVV, DIR>), dim3(blocks), dim3(256), 0, st, dst, src, numel, scale)
#define APHRO_CV_T(KVV, DIR)                                  \
  if (hp_dtype == APHRO_F16) APHRO_CV(Half, KVV, DIR);        \
  else if (hp_dtype == APHRO_BF16) APHRO_CV(BFloat, KVV, DIR); \
  else APHRO_CV(Float, KVV, DIR);
  if (kv_dtype == APHRO_KV_FP8_E4M3) {
    if (to_fp8) { APHRO_CV_T(1, true) } else { APHRO_CV_T(1, false) }
  } else {
    if (to_fp8) { APHRO_CV_T(2, true) } else { APHRO_CV_T(2, false) }
  }
#undef APHRO_CV_T
#undef APHRO_CV
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
