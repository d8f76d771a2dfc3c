// Fourth-generation prefill attention kernel for gfx950 (hd 128, sequences >= 1024; SURVEY 8a row a5, VERDICT r4
// next-round 1b): ONE wave per SIMD.  4 waves x 64 query rows = 256 rows per workgroup, 64-key tiles, 32x32x16 MFMA,
// the whole 512-register file per wave.
//
// Why (measured on the third-generation kernel, 8 waves x 32 rows, 0.71 ms at T = 8192 = 5.4 k cycles per tile against
// 2 k cycles of MFMA per SIMD): every wave there needs QK^T 1280 + softmax 1224 + PV 1032 issue cycles per tile, two such
// waves share a SIMD's issue port and matrix pipe, and each re-reads its Q fragments from LDS.  Here a wave owns TWO
// 32-row query blocks, half a tile apart in the schedule, so that every group of 32 MFMAs carries the softmax of exactly
// one block as filler (~5 VALU per MFMA gap: what one wave can issue in an MFMA's 32-cycle shadow, MI355X_MICROARCH):
//       X(t): softmax of block 0 of tile t  ||  S1 = K(t) . Q1^T (16 MFMAs)      +  O1 += V(t - 1)^T . P1(t - 1)^T (16 MFMAs)
//       Y(t): softmax of block 1 of tile t  ||  O0 += V(t)^T . P0(t)^T (16 MFMAs)  +  S0 = K(t + 1) . Q0^T (16 MFMAs)
//   * no score or probability tile is double-buffered: 64 accumulator registers of scores, 128 of output, Q (64 VGPRs) in
//     registers for the whole kernel;
//   * the running maximum moves only when a row's new maximum exceeds it by more than 2^8 (defer-max): the rescale of a
//     block's 64 output registers -- which live in the accumulator file and must travel through VGPRs -- happens on the
//     first tile and almost never again; P <= 2^8 keeps its significant bits, l accumulates the same P in fp32;
//   * one barrier per tile; K / V tiles arrive by LDS-DMA into 4-deep rings, two bodies ahead of their first reader, a
//     piece every fourth step (an LDS-DMA instruction holds the issuing wave for 100-200 cycles);
//   * the LDS fragments of a step are requested 2 same-kind steps before their MFMAs.
// LDS images, swapped QK^T (S^T = K . Q^T so that a lane holds the scores of ONE query row), the probability -> PV operand
// redistribution (permlane32_swap) and the O transpose through LDS are those of the third-generation kernel
// (flash_attn.hip).  Masked tiles (diagonal, sequence end, ALiBi), the first and the last tile of a wave take a
// straight-line path of the same arithmetic.
#include "flash_attn_common.h"

namespace aphro {

typedef short fa4_s16x4 __attribute__((ext_vector_type(4)));

// (the timing-only ablations of tools/fa_lab.hip -- no softmax slices / MFMAs / staging / rescales / fragment reads, the Q
// register class, fragment lead, sched_barriers -- live in tools/lab_patches/flash_attn_v4.hip.patch, not here)
#define FA4_QC "a"          // Q fragments are read from the accumulator file
#ifndef FA4_THR
#define FA4_THR 8.0f        // defer-max threshold, log2 units
#endif

// S += K . Q^T with the register classes chosen by hand: the scores accumulate in VGPRs, the Q fragment is read straight
// from the accumulator file.  hipcc picks ONE form per kernel -- with 512 registers every MFMA result goes to the
// accumulator file, so the 64 scores of a tile came back through 64 v_accvgpr_read (measured: 22 cycles each beside
// running MFMAs, 0.19 of 0.62 ms at T = 8192), and Q, parked in the accumulator file by the register allocator, through 64
// more.  The output accumulators (touched by MFMAs only) stay in the accumulator file through the builtin.
// Inline asm: no compiler-inserted wait states.
// Safe here because (i) the K fragment comes from a compiler-visible ds_read (its lgkmcnt wait precedes the statement),
// (ii) Q was written before the tile loop, (iii) the scores are read by VALU no earlier than the next phase -- the last
// step of a phase holds no QK^T MFMA and sched_barriers pin the steps -- more than the 12 states an 8-pass MFMA result needs
// (the prologue's QK-only phase ends with an explicit s_nop).
template <typename T, bool FIRST>
__device__ __forceinline__ void fa4_qk_mfma(f32x16& acc, u32x4 kfrag, u32x4 qfrag) {
  if constexpr (__is_same(T, Half)) {
    if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(kfrag), FA4_QC(qfrag));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(kfrag), FA4_QC(qfrag));
  } else {
    if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(kfrag), FA4_QC(qfrag));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(kfrag), FA4_QC(qfrag));
  }
}

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void flash_attn_varlen_v4_kernel(FAParams p) {
  constexpr int HD = 128, BM = 256, BN = 64;
  constexpr bool BF = __is_same(T, BFloat);
  constexpr int KT = BN * HD * 2;           // bytes of one K (or V) tile: 16 KiB
  // K ring [0, 4 KT), V ring [4 KT, 8 KT); the epilogue's 4 x 16 KiB of O transposes reuse the first 64 KiB
  extern __shared__ __attribute__((aligned(16))) unsigned char fa_smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kh = lane >> 5, l31 = lane & 31;
  // ---- which (sequence, head, 256-row query tile): all query heads of one kv head on one XCD (see flash_attn.hip) ----------
  int head, seq, qt_rev;
  {
    const int G = p.num_heads / p.num_kv_heads;
    const int per_group = G * p.nqt_max;
    const int b = blockIdx.x;
    int group, r;
    if (p.xcd_remap) { group = (b % 8) + 8 * ((b / 8) / per_group); r = (b / 8) % per_group; }
    else { group = b / per_group; r = b % per_group; }
    seq = group / p.num_kv_heads;
    head = (group % p.num_kv_heads) * G + r % G;
    qt_rev = r / G;
  }
  const int kvh = head / (p.num_heads / p.num_kv_heads);
  const int s0q = p.cu_seqlens[seq];
  const int qlen = p.cu_seqlens[seq + 1] - s0q;
  const int k_row0 = p.cu_seqlens_k ? p.cu_seqlens_k[seq] : s0q;
  const int len = p.cu_seqlens_k ? p.cu_seqlens_k[seq + 1] - k_row0 : qlen;
  const int off = len - qlen;
  const int nqt = (qlen + BM - 1) / BM;
  const int qt = nqt - 1 - qt_rev;            // heavy (late) tiles first
  if (qt < 0) return;
  const int q0 = qt * BM;
  const int wq0 = q0 + 64 * wave;

  // ---- Q fragments (B operand of S^T = K . Q^T): lane (q = l31, kh) holds d = 16 ks + 8 kh .. + 7 of its row, in registers
  u32x4 qf[2][8];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = wq0 + 32 * qb + l31;
    const uint16_t* qp = (const uint16_t*)p.q + (size_t)(s0q + min(qrow, qlen - 1)) * p.q_stride + (size_t)head * HD + 8 * kh;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[qb][ks] = *reinterpret_cast<const u32x4*>(qp + 16 * ks);
  }
  const float slope2 = (p.alibi ? p.alibi[head] : 0.f) * 1.44269504088896f;
  const float c2 = p.scale * 1.44269504088896f;

  // ---- K / V staging: 16 pieces of 4 keys (1 KiB) per tile; wave w issues pieces 4 w .. 4 w + 3 of K and of V -------------
  const uint16_t* kbase = (const uint16_t*)p.k + (size_t)k_row0 * p.k_stride + (size_t)kvh * HD;
  const uint16_t* vbase = (const uint16_t*)p.v + (size_t)k_row0 * p.v_stride + (size_t)kvh * HD;
  // buffer descriptors over this sequence's rows of this kv head (reads past the end return 0): base, no stride, bytes, raw dword format
  const u32x4 rk = fa_make_rsrc(kbase, (uint32_t)(((size_t)(len - 1) * p.k_stride + HD) * 2));
  const u32x4 rv = fa_make_rsrc(vbase, (uint32_t)(((size_t)(len - 1) * p.v_stride + HD) * 2));
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)fa_smem;
  // K piece: lane L -> key 4 c + (L >> 4), LDS slot L & 15 holds d-chunk slot ^ (key & 15)
  // V piece: lane L -> key 4 c + (L & 3), d-chunk L >> 2  (image [d-chunk][key] inside the piece)
  int k_voff[4], v_voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * wave + j;
    const int kk = 4 * c + (lane >> 4);
    k_voff[j] = (int)(kk * p.k_stride * 2) + (((lane & 15) ^ (kk & 15)) << 4);
    const int vk = 4 * c + (lane & 3);
    v_voff[j] = (int)(vk * p.v_stride * 2) + ((lane >> 2) << 4);
  }
  // one 1-KiB piece (j of this wave's 4) of the K / V tile t into ring slot t & 3
  auto stage_k1 = [&](int t, int j) __attribute__((always_inline)) {
    fa_dma16(rk, lds0 + (t & 3) * KT + (4 * wave + j) * 1024, k_voff[j], __builtin_amdgcn_readfirstlane((int)(t * BN * p.k_stride * 2)));
  };
  auto stage_v1 = [&](int t, int j) __attribute__((always_inline)) {
    fa_dma16(rv, lds0 + 4 * KT + (t & 3) * KT + (4 * wave + j) * 1024, v_voff[j], __builtin_amdgcn_readfirstlane((int)(t * BN * p.v_stride * 2)));
  };
  auto stage_k = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) stage_k1(t, j);
  };
  auto stage_v = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) stage_v1(t, j);
  };
  //  K fragment: key = 32 b + l31, d-chunk 2 ks + kh -> l31 * 256 + (((2 ks + kh) ^ (l31 & 15)) << 4) = kaddr ^ (ks << 5)
  const int kaddr = l31 * 256 + ((kh ^ (l31 & 15)) << 4);
  //  V (transposing read; 16-lane group g, lane i of it): piece 4 ks + 2 kh + h, key i >> 2 of it, d = 32 db + 16 (g & 1) + 4 (i & 3)
  const int vaddr = 4 * KT + (2 * kh) * 1024 + ((((2 * ((lane >> 4) & 1) + ((lane & 3) >> 1)) * 4 + ((lane & 15) >> 2)) << 4)) + ((lane & 1) << 3);

  f32x16 o[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][db][r] = 0.f;
  float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};

  const int kv_end = p.causal ? min(len, q0 + BM + off) : len;
  const int ntile = (kv_end + BN - 1) / BN;
  // tiles this wave computes (causal: up to the diagonal of its last row), and how many need no masking for any of its 64 rows
  const int L = p.causal ? min(ntile, (wq0 + off + 63) / BN + 1) : ntile;
  const int F = slope2 != 0.f ? 0 : min(L, p.causal ? min((wq0 + off + 1) / BN, len / BN) : len / BN);

  // ---- pieces ----------------------------------------------------------------------------------------------------------------
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto other_half = [&](float x) __attribute__((always_inline)) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    return kh ? __builtin_bit_cast(float, (unsigned)sw[0]) : __builtin_bit_cast(float, (unsigned)sw[1]);
  };
  // the running maximum moves only for rows whose new maximum is more than 2^THR above it; returns the output scale
  auto new_max = [&](int qb, float mx) __attribute__((always_inline)) -> float {
    const float m_old = m_run[qb];
    const float m_new = mx > m_old + FA4_THR ? mx : m_old;
    m_run[qb] = m_new;
    return __builtin_amdgcn_exp2f(m_old - m_new);
  };
  // softmax of one query block in 16 slices, so that the caller can drop one between MFMA groups:
  // 0-3 row maximum (+ bias and mask in place on masked tiles), 4 running maximum, 5-12 exponentials (4 of the 32 values
  // each, packed to 16 bits at once; every second slice completes one PV operand fragment), 13 row sum.
  struct SmState { float mx, alpha, lsum, m_new; uint32_t lo, hi, cur; };
  auto max3 = [](float a, float b, float c) __attribute__((always_inline)) {
    float r;      // (one instruction: hipcc puts a canonicalising v_max in front of every fmaxf on an MFMA output)
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
  };
  // MASK (diagonal / sequence-end tiles, ALiBi): v holds the scaled, biased, masked score (-1e30 where key > limit)
  // Every slice comes in two halves (h = 0 / 1): a step is [MFMA] [half 0] [MFMA] [half 1], so that the second MFMA of a
  // step finds the matrix pipe free when its turn comes (an in-order wave that issues two MFMAs back to back idles ~30 cycles).
  auto softmax_slice = [&](bool MASK, int it, int qb, f32x16 (&s)[2], u32x4 (&pf)[4], SmState& st, int sl, int h) __attribute__((always_inline)) {
    if (sl < 4) {
      const int b = sl >> 1, r0 = (sl & 1) * 8 + 4 * h;
      if (MASK) {
        const int qpos = wq0 + 32 * qb + l31 + off;
        const int lim = p.causal ? min(qpos, len - 1) : len - 1;
#pragma unroll
        for (int r = r0; r < r0 + 4; ++r) {
          const int key = it * BN + 32 * b + 8 * (r >> 2) + 4 * kh + (r & 3);
          const float x = s[b][r] * c2 + slope2 * (float)(key - qpos);
          s[b][r] = key <= lim ? x : -1e30f;          // in place: the scores are VGPRs
        }
      }
      float mx = (sl == 0 && h == 0) ? s[0][0] : st.mx;
#pragma unroll
      for (int r = r0; r < r0 + 4; r += 2) mx = max3(mx, s[b][r], s[b][r + 1]);
      st.mx = mx;
    } else if (sl == 4) {
      if (h == 0) {
        float mx = MASK ? st.mx : st.mx * c2;          // scale > 0: the maximum commutes with it
        st.mx = __builtin_fmaxf(mx, other_half(mx));
      } else {
        st.alpha = new_max(qb, st.mx);
        st.m_new = m_run[qb];
        st.lsum = 0.f;
      }
    } else if (sl < 13) {
      const int e = sl - 5;                         // 8 slices x 4 values: block e >> 2, accumulator quad e & 3
      const int b = e >> 2, q4 = e & 3, r0 = 4 * q4 + 2 * h;
      float x[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (MASK) {
          const float ex = __builtin_amdgcn_exp2f(s[b][r0 + i] - st.m_new);
          x[i] = s[b][r0 + i] > -1e29f ? ex : 0.f;
        } else {
          x[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[b][r0 + i], c2, -st.m_new));
        }
        st.lsum += x[i];
      }
      const uint32_t pk = pack2_16<BF>(x[0], x[1]);
      if (h == 0) {
        st.cur = pk;
      } else if ((q4 & 1) == 0) {
        st.lo = st.cur; st.hi = pk;
      } else {                                      // quads 2 j, 2 j + 1 -> fragment 2 b + j (k-slot = tile keys 32 b + 16 j + 8 kh .. + 7)
        const auto s_lo = __builtin_amdgcn_permlane32_swap(st.lo, st.cur, false, false);
        const auto s_hi = __builtin_amdgcn_permlane32_swap(st.hi, pk, false, false);
        pf[2 * b + (q4 >> 1)] = u32x4{(uint32_t)s_lo[0], (uint32_t)s_hi[0], (uint32_t)s_lo[1], (uint32_t)s_hi[1]};
      }
    } else if (sl == 13 && h == 0) {
      l_run[qb] = l_run[qb] * st.alpha + st.lsum;
    }
  };
  // One phase: 16 MFMAs of S = K(tq) . Q^T for one block (8 k-steps x 2 key blocks; QK), 16 MFMAs of O += V(tv)^T . P^T for
  // one block (4 k-slots x 4 d blocks; PV) and the 16 softmax slices of one block (SM), interleaved step by step:
  // step g: [QK k-step g / 2 | PV steps] + one slice.  Flags are literals at every call site.
  auto phase = [&](bool QK, int tq, f32x16 (&sq)[2], const u32x4 (&qq)[8],
                   bool PV, int tv, f32x16 (&ov)[4], const u32x4 (&pv)[4],
                   bool SM, bool MASK, int tsm, int qb_sm, f32x16 (&ssm)[2], u32x4 (&pfsm)[4], SmState& st,
                   int dma, int tdma) __attribute__((always_inline)) {
    // dma: 0 none, 1 this wave's 4 pieces of K(tdma), 2 of V(tdma) -- one piece every fourth step (an LDS-DMA costs the issuing
    // wave 100-200 cycles: 8 of them in a burst at the top of a body were 0.24 of 0.80 ms, the same as in a loop with nothing else)
    const unsigned char* sk = fa_smem + (tq & 3) * KT;
    int ka = kaddr, va = vaddr + ((tv + 4) & 3) * KT;
    asm volatile("" : "+v"(ka), "+v"(va));      // (opaque per phase: hipcc would hoist the XOR-ed addresses x 4 ring slots and spill them)
    // the LDS fragments of a step are requested one same-kind step (two steps) ahead of their MFMAs: one wave per
    // SIMD has nobody to hide an LDS round trip behind (left alone hipcc reads them right in front of the MFMA)
    constexpr int D = 1;
    u32x4 kf[D + 1][2], vf[D + 1][2];
    auto rdk = [&](int ks) __attribute__((always_inline)) {
      kf[ks % (D + 1)][0] = *reinterpret_cast<const u32x4*>(sk + (ka ^ (ks << 5)));
      kf[ks % (D + 1)][1] = *reinterpret_cast<const u32x4*>(sk + 8192 + (ka ^ (ks << 5)));
    };
    auto rdv = [&](int e) __attribute__((always_inline)) {
      const int ks = e >> 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int db = 2 * (e & 1) + h;
        const unsigned char* vb = fa_smem + va + (4 * ks) * 1024 + db * 256;
        const fa4_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fa4_s16x4 __attribute__((address_space(3)))*)vb);
        const fa4_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fa4_s16x4 __attribute__((address_space(3)))*)(vb + 1024));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        vf[e % (D + 1)][h] = u32x4{l2[0], l2[1], h2[0], h2[1]};
      }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (QK) rdk(d);
      if (PV) rdv(d);
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int i = g >> 1;                      // k-step of S (even g) / (k-slot, d-block pair) of O (odd g)
      const bool qk_step = (g & 1) == 0 ? QK : false, pv_step = (g & 1) == 1 ? PV : false;
      if (qk_step && i + D < 8) rdk(i + D);
      if (pv_step && i + D < 8) rdv(i + D);
      // first MFMA
      if (qk_step) {
        if (i == 0) fa4_qk_mfma<T, true>(sq[0], kf[0][0], qq[0]);
        else fa4_qk_mfma<T, false>(sq[0], kf[i % (D + 1)][0], qq[i]);
      }
      if (pv_step) {
        ov[2 * (i & 1)] = fa_mfma32<T>(vf[i % (D + 1)][0], pv[i >> 1], ov[2 * (i & 1)]);
      }
      if ((g & 3) == 2) {
        if (dma == 1) stage_k1(tdma, g >> 2);
        if (dma == 2) stage_v1(tdma, g >> 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (SM) softmax_slice(MASK, tsm, qb_sm, ssm, pfsm, st, g, 0);
      __builtin_amdgcn_sched_barrier(0);
      // second MFMA
      if (qk_step) {
        if (i == 0) fa4_qk_mfma<T, true>(sq[1], kf[0][1], qq[0]);
        else fa4_qk_mfma<T, false>(sq[1], kf[i % (D + 1)][1], qq[i]);
      }
      if (pv_step) ov[2 * (i & 1) + 1] = fa_mfma32<T>(vf[i % (D + 1)][1], pv[i >> 1], ov[2 * (i & 1) + 1]);
      __builtin_amdgcn_sched_barrier(0);
      if (SM) softmax_slice(MASK, tsm, qb_sm, ssm, pfsm, st, g, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto rescale = [&](f32x16 (&ov)[4], float a) __attribute__((always_inline)) {
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(a != 1.0f) != 0, 0)) {      // (cold: spill weights)
      // (a quarter of a block at a time: done in one batch hipcc reserves 64 VGPRs for it, which evicts Q to the accumulator
      //  file for the whole loop -- 64 v_accvgpr_read per tile in the hot path)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ov[db][r] *= a;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- the tile loop --------------------------------------------------------------------------------------------------------
  // body(t) = X(t), Y(t), one barrier at its top.  Every wave is then past body t - 1, whose phases read K(t - 1), V(t - 2),
  // V(t - 1), K(t): the slots of K(t - 1) and V(t - 2) are dead and receive K(t + 3) (this wave's pieces go out during X) and
  // V(t + 2) (during Y) -- two bodies ahead of their first readers, so the wait at the top of a body leaves the youngest
  // body's pieces in flight (counted vmcnt).  4-deep rings.
  f32x16 sc0[2], sc1[2];
  u32x4 pf0[4], pf1[4];
  SmState st0, st1;
#pragma unroll
  for (int i = 0; i < 4; ++i) pf1[i] = u32x4{0u, 0u, 0u, 0u};          // "tile -1": body 0 multiplies these zeros into V slot 3 ...
  {                                                                     // ... which is zero-filled (stale LDS may hold NaN patterns)
    u32x4* vz = reinterpret_cast<u32x4*>(fa_smem + 4 * KT + 3 * KT);
#pragma unroll
    for (int i = 0; i < 4; ++i) vz[threadIdx.x + 256 * i] = u32x4{0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) stage_k(i);
#pragma unroll
  for (int i = 0; i < 2; ++i) stage_v(i);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  // (the Q loads are "used" here as far as hipcc can tell: otherwise it waits for them at their first uses INSIDE the loop --
  //  vmcnt(7) ... vmcnt(0) in every iteration, which drains the hand-counted LDS-DMA pieces one by one)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[qb][ks]));
  __builtin_amdgcn_s_barrier();
  phase(true, 0, sc0, qf[0], false, 0, o[0], pf0, false, false, 0, 0, sc0, pf0, st0, 0, 0);          // S0 of tile 0
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  auto top = [&](int t) __attribute__((always_inline)) {
    // everything but the 8 pieces of body t - 1 has landed.  (Every body issues its 8 pieces, also past the last tile: the
    // buffer descriptors end with the sequence, rows beyond it arrive as zeros in slots nobody reads any more.)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  // Every body runs all 64 MFMAs: at t = 0 the PV of "tile -1" multiplies zeros, at this wave's last tile the S0 of tile
  // t + 1 is computed and never used (a real tile of a later wave, or a stale finite one) -- two loops, one straight-line
  // body each (a loop mixing bodies, or bodies with optional pieces, makes hipcc shuffle and spill the 192 accumulator
  // registers at every join: measured 2 x slower than the third-generation kernel).
  const int t_fast = min(F, L);
  int t = 0;
  for (; t < t_fast; ++t) {
    top(t);
    phase(true, t, sc1, qf[1], true, t - 1, o[1], pf1, true, false, t, 0, sc0, pf0, st0, 1, t + 3);
    rescale(o[0], st0.alpha);
    phase(true, t + 1, sc0, qf[0], true, t, o[0], pf0, true, false, t, 1, sc1, pf1, st1, 2, t + 2);
    rescale(o[1], st1.alpha);
  }
  for (; t < L; ++t) {
    top(t);
    phase(true, t, sc1, qf[1], true, t - 1, o[1], pf1, true, true, t, 0, sc0, pf0, st0, 1, t + 3);
    rescale(o[0], st0.alpha);
    phase(true, t + 1, sc0, qf[0], true, t, o[0], pf0, true, true, t, 1, sc1, pf1, st1, 2, t + 2);
    rescale(o[1], st1.alpha);
  }
  for (; t <= ntile; ++t) {
    top(t);
    stage_k(t + 3);
    stage_v(t + 2);
    if (t == L)                            // this wave's last PV (block 1 of tile L - 1)
      phase(false, 0, sc1, qf[1], true, t - 1, o[1], pf1, false, false, 0, 0, sc0, pf0, st0, 0, 0);
  }
  // The tail bodies above still issued K(t + 3) / V(t + 2) pieces by LDS-DMA from inline asm hipcc cannot count, and
  // __syncthreads() does not wait on vmcnt: a piece of a causal non-final q tile (real rows below `len`) could land in the
  // K ring AFTER a wave has written its O rows there.  Drain this wave's pieces before the barrier that frees the rings.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                         // every wave is done with the K / V buffers

  // ---- normalise, transpose through LDS (wave-private 16 KiB: 64 rows x 256 B), store whole rows -------------------------------
  unsigned char* region = fa_smem + wave * 16384;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float l = l_run[qb];
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, l), __builtin_bit_cast(unsigned, l), false, false);
      l = __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
    }
    const float inv = l > 0.f ? 1.f / l : 0.f;
    unsigned char* reg = region + qb * 8192;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int c8 = 8 * db + 2 * g4 + kh;
        const u32x2 v = {pack2_16<BF>(o[qb][db][4 * g4] * inv, o[qb][db][4 * g4 + 1] * inv),
                         pack2_16<BF>(o[qb][db][4 * g4 + 2] * inv, o[qb][db][4 * g4 + 3] * inv)};
        *reinterpret_cast<u32x2*>(reg + l31 * 256 + ((c8 ^ ((l31 & 15) << 1)) << 3)) = v;
      }
  }
  uint16_t* obase = (uint16_t*)p.out + (size_t)(s0q + wq0) * p.o_stride + (size_t)head * HD;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = i * 4 + (lane >> 4), c16 = lane & 15;
    const u32x4 v = *reinterpret_cast<const u32x4*>(region + row * 256 + ((c16 ^ (row & 15)) << 4));
    if (wq0 + row < qlen) *reinterpret_cast<u32x4*>(obase + (size_t)row * p.o_stride + c16 * 8) = v;
  }
}

int fa_v4_launch(const FAParams& p, int dtype, int batch, hipStream_t st) {
  constexpr int LDS = 8 * 16384;
  dim3 grid((unsigned)(p.nqt_max * p.num_heads * batch));
  static bool attr_set_dev[APHRO_MAX_DEVICES] = {};
  bool& attr_set = attr_set_dev[device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)flash_attn_varlen_v4_kernel<Half>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)flash_attn_varlen_v4_kernel<BFloat>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      set_error("flash attention: cannot raise the dynamic LDS limit");
      return APHRO_ERR_LAUNCH;
    }
    attr_set = true;
  }
  if (dtype == APHRO_F16) hipLaunchKernelGGL((flash_attn_varlen_v4_kernel<Half>), grid, dim3(256), LDS, st, p);
  else hipLaunchKernelGGL((flash_attn_varlen_v4_kernel<BFloat>), grid, dim3(256), LDS, st, p);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

}  // namespace aphro
