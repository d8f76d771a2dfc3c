// Prefill attention for gfx950: causal var-len flash attention on MFMA
// (SURVEY 8a row a5).  Fills the role of the Triton / CK kernels the reference
// calls at aphrodite/attention/backends/rocm_flash_attn.py:455-508
// (attention/ops/triton_flash_attn.py:700-820): q [T,Hq,hd], k/v [T,Hkv,hd]
// packed sequences delimited by cu_seqlens, GQA by head / (Hq/Hkv).
//
// Design: one workgroup = 4 waves = 64 query rows of one (sequence, head); each
// wave owns 16 rows.  K/V tiles of 32 tokens are staged through LDS once per
// workgroup (K row-major with a 16-byte XOR swizzle so the ds_read_b128 fragment
// reads are conflict-free, V transposed on the way in so the PV operand is a
// contiguous 8-byte read).  As in the decode kernel the score tile is computed
// transposed, S^T[token, q] = K . Q^T, which leaves the probabilities in
// exactly the lane layout the PV MFMA consumes as its B operand:
// O^T[d, q] += V^T[d, token] . P^T[token, q].  Online softmax in registers,
// fp32 accumulation.  MFMA-bound regime; this first version is single-buffered
// (two barriers per K/V tile) -- see DESIGN.md for the planned pipeline.
#include "common.h"
#include "flash_attn_common.h"

namespace aphro {


template <typename T>
__device__ __forceinline__ f32x4 fa_mfma(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (__is_same(T, Half))
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <typename T>
__device__ __forceinline__ uint32_t fa_pack2(float a, float b) {
  if constexpr (__is_same(T, Half)) {
    f16x2 h = {(f16)a, (f16)b};
    return __builtin_bit_cast(uint32_t, h);
  } else {
    return (uint32_t)f32_to_bf16_bits(a) | ((uint32_t)f32_to_bf16_bits(b) << 16);
  }
}

constexpr int FA_BM = 64;   // query rows per workgroup (context kernel; QT * 64 in the varlen kernel)
constexpr int FA_BN = 32;   // kv tokens per tile
constexpr int FA_VT_STRIDE = 36;  // halfs per V^T row (32 + 4 pad)

// QT = 16-row query tiles per wave: every K / V^T fragment read from LDS (and every byte staged
// into it) is used by QT MFMAs -- the staging path, not the MFMA pipe, bounds this kernel.
template <typename T, int HD, int QT>
__global__ __launch_bounds__(256) void flash_attn_varlen_kernel(FAParams p) {
  constexpr int NCH = HD / 8;    // 16-byte chunks per row
  constexpr int SWZ = (NCH & -NCH) - 1;  // XOR mask: largest power of two dividing NCH, minus 1
  constexpr int NKS = HD / 32;
  constexpr int NDT = HD / 16;
  constexpr int BM = 64 * QT;
  __shared__ __attribute__((aligned(16))) uint16_t k_lds[FA_BN * HD];
  __shared__ __attribute__((aligned(16))) uint16_t vt_lds[HD * FA_VT_STRIDE];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4;
  const int c = lane & 15;
  const int head = blockIdx.y;
  const int seq = blockIdx.z;
  const int kvh = head / (p.num_heads / p.num_kv_heads);
  // query rows [s0, s0 + qlen) of q / out; key rows [k0, k0 + len) of k / v.  With cu_seqlens_k (prefill over a cached
  // context, gathered into contiguous rows) the keys are context + new tokens and query row i sits at key position i + off.
  const int s0 = p.cu_seqlens[seq];
  const int qlen = p.cu_seqlens[seq + 1] - s0;
  const int k0 = p.cu_seqlens_k ? p.cu_seqlens_k[seq] : s0;
  const int len = p.cu_seqlens_k ? p.cu_seqlens_k[seq + 1] - k0 : qlen;
  const int off = len - qlen;
  const int win = p.window;
  // heavy (late) tiles first: better tail balance under the causal triangle
  const int ntiles = (qlen + BM - 1) / BM;
  const int tile = ntiles - 1 - (int)blockIdx.x;
  if (tile < 0) return;
  const int q0 = tile * BM;
  const int wq0 = q0 + 16 * QT * wave;           // first query row of this wave
  int qrow[QT];                                  // this lane's query rows (B-operand columns)
#pragma unroll
  for (int t = 0; t < QT; ++t) qrow[t] = wq0 + 16 * t + c;

  // ---- Q fragments ---------------------------------------------------------------
  u32x4 qf[QT][NKS];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const uint16_t* qp = (const uint16_t*)p.q + (size_t)(s0 + min(qrow[t], qlen - 1)) * p.q_stride + (size_t)head * HD;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[t][ks] = *reinterpret_cast<const u32x4*>(qp + 32 * ks + 8 * g);
  }
  const float slope = p.alibi ? p.alibi[head] : 0.f;
  const float sc2 = p.scale * 1.44269504088896f;     // softmax in the log2 domain
  const float slope2 = slope * 1.44269504088896f;

  f32x4 o[QT][NDT];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
#pragma unroll
    for (int i = 0; i < NDT; ++i) o[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    m_run[t] = -1e30f;
    l_run[t] = 0.f;
  }

  const int kv_end = p.causal ? min(len, q0 + BM + off) : len;
  // sliding window: the first key any row of this workgroup can see
  const int kv_begin = win > 0 ? max(0, q0 + off - win + 1) / FA_BN * FA_BN : 0;
  const uint16_t* kbase = (const uint16_t*)p.k + (size_t)k0 * p.k_stride + (size_t)kvh * HD;
  const uint16_t* vbase = (const uint16_t*)p.v + (size_t)k0 * p.v_stride + (size_t)kvh * HD;

  for (int t0 = kv_begin; t0 < kv_end; t0 += FA_BN) {
    // ---- stage K (swizzled) and V^T into LDS -------------------------------------
    // (prefetching the next tile into registers across the compute phase was measured SLOWER:
    //  247 -> 181 TFLOP/s at T = 8192 -- the extra live registers cost more than the latency)
    __syncthreads();  // previous tile's readers are done
    for (int i = threadIdx.x; i < FA_BN * NCH; i += 256) {
      const int tok = i / NCH, ch = i % NCH;
      const int ta = min(t0 + tok, len - 1);
      u32x4 kv4 = *reinterpret_cast<const u32x4*>(kbase + (size_t)ta * p.k_stride + 8 * ch);
      *reinterpret_cast<u32x4*>(&k_lds[tok * HD + 8 * (ch ^ (tok & SWZ))]) = kv4;
      u16x8 vv = *reinterpret_cast<const u16x8*>(vbase + (size_t)ta * p.v_stride + 8 * ch);
      if (t0 + tok >= len) vv = u16x8{0, 0, 0, 0, 0, 0, 0, 0};  // 0 * garbage must stay 0
#pragma unroll
      for (int j = 0; j < 8; ++j) vt_lds[(8 * ch + j) * FA_VT_STRIDE + tok] = vv[j];
    }
    __syncthreads();
    // a wave whose rows all precede this tile (causal) has nothing to add
    const bool wave_active = !p.causal || (t0 <= wq0 + 16 * QT - 1 + off);
    if (wave_active) {
      // ---- S^T = K . Q^T : one K fragment read feeds the QT query tiles --------------
      f32x4 s[QT][2];
#pragma unroll
      for (int t = 0; t < QT; ++t) { s[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; s[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int tok = 16 * h + c;  // A-operand row of this lane
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const int ch = (4 * ks + g) ^ (tok & SWZ);
          u32x4 kf = *reinterpret_cast<const u32x4*>(&k_lds[tok * HD + 8 * ch]);
#pragma unroll
          for (int t = 0; t < QT; ++t) s[t][h] = fa_mfma<T>(kf, qf[t][ks], s[t][h]);
        }
      }
      // ---- online softmax (lane column = query row), log2 domain ----------------------
      // x2 = s * scale * log2(e) (+ alibi); p = exp2(x2 - m).  Interior tiles (every key visible to
      // every row of the wave, no ALiBi) skip the mask and bias arithmetic; the O rescale is skipped
      // when no lane's running maximum moved (alpha == 1 everywhere).
      u32x4 pf[QT];
      const bool edge = (t0 + FA_BN > len) || (p.causal && t0 + FA_BN - 1 > wq0 + off) || slope != 0.f || win > 0;
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        float pv[2][4];
        float mx = -1e30f;
        if (edge) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int tok = t0 + 16 * h + 4 * g + r;
              float x = s[t][h][r] * sc2 + slope2 * (float)(tok - qrow[t] - off);
              const bool ok = tok < len && (!p.causal || tok <= qrow[t] + off) && (win <= 0 || tok > qrow[t] + off - win);
              x = ok ? x : -1e30f;
              pv[h][r] = x;
              mx = __builtin_fmaxf(mx, x);
            }
        } else {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              pv[h][r] = s[t][h][r] * sc2;
              mx = __builtin_fmaxf(mx, pv[h][r]);
            }
        }
        mx = __builtin_fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = __builtin_fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = __builtin_fmaxf(m_run[t], mx);
        const bool moved = m_new != m_run[t];
        const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);
        m_run[t] = m_new;
        float lsum = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // masked entries sit at -1e30: exp2 underflows to exactly 0 unless the whole row is still
            // masked (m_new == -1e30), which the select handles
            float e = __builtin_amdgcn_exp2f(pv[h][r] - m_new);
            if (edge) e = pv[h][r] > -1e29f ? e : 0.f;
            pv[h][r] = e;
            lsum += e;
          }
        l_run[t] = l_run[t] * alpha + lsum;
        if (__builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) o[t][dt] *= alpha;
        }
        pf[t][0] = fa_pack2<T>(pv[0][0], pv[0][1]);
        pf[t][1] = fa_pack2<T>(pv[0][2], pv[0][3]);
        pf[t][2] = fa_pack2<T>(pv[1][0], pv[1][1]);
        pf[t][3] = fa_pack2<T>(pv[1][2], pv[1][3]);
      }
      // ---- O^T += V^T . P^T : one V^T fragment read feeds the QT query tiles ----------
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const uint16_t* vr = &vt_lds[(16 * dt + c) * FA_VT_STRIDE + 4 * g];
        u32x2 lo = *reinterpret_cast<const u32x2*>(vr);
        u32x2 hi = *reinterpret_cast<const u32x2*>(vr + 16);
        u32x4 vf = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
        for (int t = 0; t < QT; ++t) o[t][dt] = fa_mfma<T>(vf, pf[t], o[t][dt]);
      }
    }
  }

#pragma unroll
  for (int t = 0; t < QT; ++t) {
    float l = l_run[t];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (qrow[t] < qlen) {
      const float inv = l > 0.f ? 1.f / l : 0.f;
      typename T::storage* op = (typename T::storage*)p.out + (size_t)(s0 + qrow[t]) * p.o_stride + (size_t)head * HD;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        u16x4 r = {T::from_f32(o[t][dt][0] * inv), T::from_f32(o[t][dt][1] * inv), T::from_f32(o[t][dt][2] * inv),
                   T::from_f32(o[t][dt][3] * inv)};
        *reinterpret_cast<u16x4*>(op + 16 * dt + 4 * g) = r;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Second-generation prefill kernel (hd 64 / 128): 64-key tiles, V staged ROW-major with 16-byte
// LDS writes and consumed through ds_read_b64_tr_b16 -- the gfx950 transposing LDS read (probe:
// tools/tr_probe.hip: within a 16-lane group, lane i supplies &blk[i/4][4*(i%4)] of a [4 keys][16
// cols] block and lane c receives column c = {blk[0..3][c]}), which is exactly the 4-token slice of
// the PV A-operand.  Removes the 2-byte V^T scatter (16-way bank conflicts, 239 M conflict cycles
// per launch at T = 8192) and halves the barriers per key.
// ---------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int FA2_BN = 64;

template <typename T, int HD, int QT, int NWV>
__global__ __launch_bounds__(NWV * 64) void flash_attn_varlen_v2_kernel(FAParams p) {
  constexpr int NCH = HD / 8;
  constexpr int SWZ = (NCH & -NCH) - 1;
  constexpr int NKS = HD / 32;
  constexpr int NDT = HD / 16;
  constexpr int BM = 16 * QT * NWV;
  constexpr int NTHR = NWV * 64;
  constexpr int VS = HD + 16;   // halfs per V row: 8 dwords of skew per row keeps the tr reads conflict free
  __shared__ __attribute__((aligned(16))) uint16_t k_lds[FA2_BN * HD];
  __shared__ __attribute__((aligned(16))) uint16_t v_lds[FA2_BN * VS];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4;
  const int c = lane & 15;
  const int head = blockIdx.y;
  const int seq = blockIdx.z;
  const int kvh = head / (p.num_heads / p.num_kv_heads);
  // query rows [s0, s0 + qlen) of q / out; key rows [k0, k0 + len) of k / v.  With cu_seqlens_k (prefill over a cached
  // context, gathered into contiguous rows) the keys are context + new tokens and query row i sits at key position i + off.
  const int s0 = p.cu_seqlens[seq];
  const int qlen = p.cu_seqlens[seq + 1] - s0;
  const int k0 = p.cu_seqlens_k ? p.cu_seqlens_k[seq] : s0;
  const int len = p.cu_seqlens_k ? p.cu_seqlens_k[seq + 1] - k0 : qlen;
  const int off = len - qlen;
  const int win = p.window;
  const int ntiles = (qlen + BM - 1) / BM;
  const int tile = ntiles - 1 - (int)blockIdx.x;   // heavy (late) tiles first
  if (tile < 0) return;
  const int q0 = tile * BM;
  const int wq0 = q0 + 16 * QT * wave;
  int qrow[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) qrow[t] = wq0 + 16 * t + c;

  u32x4 qf[QT][NKS];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const uint16_t* qp = (const uint16_t*)p.q + (size_t)(s0 + min(qrow[t], qlen - 1)) * p.q_stride + (size_t)head * HD;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[t][ks] = *reinterpret_cast<const u32x4*>(qp + 32 * ks + 8 * g);
  }
  const float slope = p.alibi ? p.alibi[head] : 0.f;
  const float sc2 = p.scale * 1.44269504088896f;
  const float slope2 = slope * 1.44269504088896f;

  f32x4 o[QT][NDT];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
#pragma unroll
    for (int i = 0; i < NDT; ++i) o[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    m_run[t] = -1e30f;
    l_run[t] = 0.f;
  }

  const int kv_end = p.causal ? min(len, q0 + BM + off) : len;
  // sliding window: the first key any row of this workgroup can see
  const int kv_begin = win > 0 ? max(0, q0 + off - win + 1) / FA2_BN * FA2_BN : 0;
  const uint16_t* kbase = (const uint16_t*)p.k + (size_t)k0 * p.k_stride + (size_t)kvh * HD;
  const uint16_t* vbase = (const uint16_t*)p.v + (size_t)k0 * p.v_stride + (size_t)kvh * HD;
  // per-lane part of the transposing V read: row (i / 4), 4 columns at 4 * (i % 4)
  const int vtr_off = (4 * g + (c >> 2)) * VS + 4 * (c & 3);

  // K/V staging is software pipelined through registers: the global loads of tile i+1 are issued
  // right after tile i became visible in LDS and land while tile i is computed (329 -> 420 TFLOP/s
  // at T = 8192; a second LDS buffer that would save one of the two barriers per tile measured
  // slower: it halves the workgroups per CU).
  constexpr int CPT = FA2_BN * NCH / NTHR;   // 16-byte chunks per thread per tile (K and V each)
  static_assert(FA2_BN * NCH % NTHR == 0, "hd 64 / 128 only");
  u32x4 kreg[CPT], vreg[CPT];
  auto fetch = [&](int t0) {
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
      const int i = threadIdx.x + NTHR * q;
      const int tok = i / NCH, ch = i % NCH;
      const int ta = min(t0 + tok, len - 1);
      kreg[q] = *reinterpret_cast<const u32x4*>(kbase + (size_t)ta * p.k_stride + 8 * ch);
      vreg[q] = *reinterpret_cast<const u32x4*>(vbase + (size_t)ta * p.v_stride + 8 * ch);
    }
  };
  fetch(kv_begin);
  for (int t0 = kv_begin; t0 < kv_end; t0 += FA2_BN) {
    __syncthreads();  // previous tile's readers are done
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
      const int i = threadIdx.x + NTHR * q;
      const int tok = i / NCH, ch = i % NCH;
      u32x4 vv4 = vreg[q];
      if (t0 + tok >= len) vv4 = u32x4{0, 0, 0, 0};  // 0 * garbage must stay 0
      *reinterpret_cast<u32x4*>(&k_lds[tok * HD + 8 * (ch ^ (tok & SWZ))]) = kreg[q];
      *reinterpret_cast<u32x4*>(&v_lds[tok * VS + 8 * ch]) = vv4;
    }
    __syncthreads();
    if (t0 + FA2_BN < kv_end) fetch(t0 + FA2_BN);
    const bool wave_active = !p.causal || (t0 <= wq0 + 16 * QT - 1 + off);
    if (wave_active) {
      const bool edge = (t0 + FA2_BN > len) || (p.causal && t0 + FA2_BN - 1 > wq0 + off) || slope != 0.f || win > 0;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {          // two 32-key halves of the tile
        const int tb = t0 + 32 * pr;
        if (p.causal && tb > wq0 + 16 * QT - 1 + off) break;   // wave-uniform: this half is entirely masked
        // ---- S^T = K . Q^T ------------------------------------------------------------
        f32x4 s[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) { s[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; s[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int tok = 32 * pr + 16 * h + c;
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) {
            const int ch = (4 * ks + g) ^ (tok & SWZ);
            u32x4 kf = *reinterpret_cast<const u32x4*>(&k_lds[tok * HD + 8 * ch]);
#pragma unroll
            for (int t = 0; t < QT; ++t) s[t][h] = fa_mfma<T>(kf, qf[t][ks], s[t][h]);
          }
        }
        // ---- online softmax, log2 domain (see the first-generation kernel) ---------------
        u32x4 pf[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          float pv[2][4];
          float mx = -1e30f;
          if (edge) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int tok = tb + 16 * h + 4 * g + r;
                float x = s[t][h][r] * sc2 + slope2 * (float)(tok - qrow[t] - off);
                const bool ok = tok < len && (!p.causal || tok <= qrow[t] + off) && (win <= 0 || tok > qrow[t] + off - win);
                x = ok ? x : -1e30f;
                pv[h][r] = x;
                mx = __builtin_fmaxf(mx, x);
              }
          } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                pv[h][r] = s[t][h][r] * sc2;
                mx = __builtin_fmaxf(mx, pv[h][r]);
              }
          }
          mx = __builtin_fmaxf(mx, __shfl_xor(mx, 16, 64));
          mx = __builtin_fmaxf(mx, __shfl_xor(mx, 32, 64));
          const float m_new = __builtin_fmaxf(m_run[t], mx);
          const bool moved = m_new != m_run[t];
          const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);
          m_run[t] = m_new;
          float lsum = 0.f;
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float e = __builtin_amdgcn_exp2f(pv[h][r] - m_new);
              if (edge) e = pv[h][r] > -1e29f ? e : 0.f;
              pv[h][r] = e;
              lsum += e;
            }
          l_run[t] = l_run[t] * alpha + lsum;
          if (__builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) o[t][dt] *= alpha;
          }
          pf[t][0] = fa_pack2<T>(pv[0][0], pv[0][1]);
          pf[t][1] = fa_pack2<T>(pv[0][2], pv[0][3]);
          pf[t][2] = fa_pack2<T>(pv[1][0], pv[1][1]);
          pf[t][3] = fa_pack2<T>(pv[1][2], pv[1][3]);
        }
        // ---- O^T += V^T . P^T : A fragment = two transposing reads (tokens 16h + 4g .. +3) -----
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const uint16_t* vb = &v_lds[(32 * pr) * VS + 16 * dt + vtr_off];
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)vb);
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vb + 16 * VS));
          u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          u32x4 vf = {l2[0], l2[1], h2[0], h2[1]};
#pragma unroll
          for (int t = 0; t < QT; ++t) o[t][dt] = fa_mfma<T>(vf, pf[t], o[t][dt]);
        }
      }
    }
  }

#pragma unroll
  for (int t = 0; t < QT; ++t) {
    float l = l_run[t];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (qrow[t] < qlen) {
      const float inv = l > 0.f ? 1.f / l : 0.f;
      typename T::storage* op = (typename T::storage*)p.out + (size_t)(s0 + qrow[t]) * p.o_stride + (size_t)head * HD;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        u16x4 r = {T::from_f32(o[t][dt][0] * inv), T::from_f32(o[t][dt][1] * inv), T::from_f32(o[t][dt][2] * inv),
                   T::from_f32(o[t][dt][3] * inv)};
        *reinterpret_cast<u16x4*>(op + 16 * dt + 4 * g) = r;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Third-generation prefill kernel (hd 128, sequences >= 1024): 8 waves x 32 query rows = 256 rows per workgroup,
// 64-key tiles, 32x32x16 MFMA.  VERDICT r1 #6: the second-generation kernel is VALU-issue bound at 7.4 VALU per
// 16x16x32 MFMA; this one spends ~150 VALU per 32 MFMAs of 32 cycles:
//   * swapped QK^T on 32x32 tiles (S^T = K . Q^T): a lane holds 32 of the 64 scores of ONE query row, so the row
//     maximum / sum are 31 lane-local ops + one half-wave exchange (v_permlane32_swap), no LDS;
//   * softmax in the log2 domain straight off the accumulators: p = exp2(fma(s, scale*log2e, -m)), 2 VALU per score;
//   * P -> 16-bit with packed hardware converts, redistributed into PV B-operand fragments by 8 permlane32_swap per
//     32 keys (the MFMA C layout interleaves the two half-waves' keys in groups of 4);
//   * K and V tiles go global -> LDS with direct-to-LDS loads (no VGPR round trip, no address VALU in the loop),
//     a 3-deep ring, one barrier per tile; the next tile's QK^T shares a straight-line block with this tile's softmax.  The LDS image is chosen on the SOURCE side (lane L of a 1 KiB piece
//     fetches whatever must live at byte 16 L): K rows XOR-swizzled by key & 15 for conflict-free ds_read_b128
//     fragments; V in [16-byte d-chunk][4 keys] order inside each 4-key piece, which makes every
//     ds_read_b64_tr_b16 (transposing read: 4 keys of one d per lane) of a PV A-operand hit 64 distinct banks;
//   * O leaves through a wave-private LDS transpose as full 256-byte rows.
// Keys beyond the sequence are fetched as zeros by the buffer descriptor's bounds check.
// ---------------------------------------------------------------------------


template <typename T>
__global__ __launch_bounds__(512) void flash_attn_varlen_v3_kernel(FAParams p) {
  constexpr int HD = 128, BM = 256, BN = 64;
  constexpr bool BF = __is_same(T, BFloat);
  constexpr int KT = BN * HD * 2;           // bytes of one K (or V) tile: 16 KiB
  constexpr int STAGE = 2 * KT;
  extern __shared__ __attribute__((aligned(16))) unsigned char fa_smem[];   // ring of 3 x [K tile | V tile], then 8 x 8 KiB of Q fragments

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kh = lane >> 5, l31 = lane & 31;
  // ---- which (sequence, head, 256-row query tile) is this workgroup -------------------------------------------------------
  // Workgroup b runs on XCD b % 8 (round-robin dispatch).  All query heads of ONE kv head, and all their query tiles, go
  // to the same XCD and walk the K/V tiles from key 0 upwards roughly in step: a K/V tile is pulled into that XCD's
  // 4 MiB L2 once and the other workgroups hit it (K + V of one kv head at 8192 tokens = 4 MiB).  With the plain
  // (tile, head, sequence) grid every XCD streamed every kv head: the staging alone ran at the 6 TB/s MALL/HBM rate and
  // cost 0.37 of 1.06 ms at T = 8192 (tools/fa_lab.hip).  Within an XCD: heavy (late) query tiles first.
  int head, seq, qt_rev;
  {
    const int G = p.num_heads / p.num_kv_heads;
    const int per_group = G * p.nqt_max;                  // workgroups of one (sequence, kv head)
    const int b = blockIdx.x;
    int group, r;
    if (p.xcd_remap) { group = (b % 8) + 8 * ((b / 8) / per_group); r = (b / 8) % per_group; }
    else { group = b / per_group; r = b % per_group; }
    seq = group / p.num_kv_heads;
    head = (group % p.num_kv_heads) * G + r % G;
    qt_rev = r / G;
  }
  const int kvh = head / (p.num_heads / p.num_kv_heads);
  // queries [s0q, s0q + qlen) and keys [s0, s0 + len): the same rows for plain prefill; with a cached context in front
  // (cu_seqlens_k given) the keys are context + new tokens and query row i sits at key position i + off, off = len - qlen
  const int s0q = p.cu_seqlens[seq];
  const int qlen = p.cu_seqlens[seq + 1] - s0q;
  const int s0 = p.cu_seqlens_k ? p.cu_seqlens_k[seq] : s0q;
  const int len = p.cu_seqlens_k ? p.cu_seqlens_k[seq + 1] - s0 : qlen;
  const int off = len - qlen;
  const int nqt = (qlen + BM - 1) / BM;
  const int qt = nqt - 1 - qt_rev;            // heavy (late) tiles first
  if (qt < 0) return;
  const int q0 = qt * BM;
  const int wq0 = q0 + 32 * wave;
  const int qrow = wq0 + l31;

  // ---- Q fragments (B operand of S^T = K . Q^T): lane (q = l31, kh) holds d = 16 ks + 8 kh .. + 7.  They live in LDS, not
  // in 32 VGPRs: the lane that wrote a fragment is the one that reads it back (8 ds_read_b128 per tile), which is what
  // lets two score tiles + the output accumulators + the P fragments fit in 256 registers without spills.
  unsigned char* qlds = fa_smem + 3 * STAGE + wave * 8192;
  const int qaddr = l31 * 256 + ((kh ^ (l31 & 15)) << 4);          // chunk 2 ks + kh of row l31, XOR-swizzled like K
  {
    const uint16_t* qp = (const uint16_t*)p.q + (size_t)(s0q + min(qrow, qlen - 1)) * p.q_stride + (size_t)head * HD + 8 * kh;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      *reinterpret_cast<u32x4*>(qlds + (qaddr ^ (ks << 5))) = *reinterpret_cast<const u32x4*>(qp + 16 * ks);
  }
  const float slope2 = (p.alibi ? p.alibi[head] : 0.f) * 1.44269504088896f;
  const float c2 = p.scale * 1.44269504088896f;

  // ---- K / V staging: buffer descriptors over this sequence's rows of this kv head (reads past the end return 0) -------
  const uint16_t* kbase = (const uint16_t*)p.k + (size_t)s0 * p.k_stride + (size_t)kvh * HD;
  const uint16_t* vbase = (const uint16_t*)p.v + (size_t)s0 * p.v_stride + (size_t)kvh * HD;
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(kbase), 0,
      (uint32_t)(((size_t)(len - 1) * p.k_stride + HD) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(vbase), 0,
      (uint32_t)(((size_t)(len - 1) * p.v_stride + HD) * 2), 0x00020000);
  // a wave stages pieces 2 wave, 2 wave + 1 (4 keys = 1 KiB each) of both tiles.
  // K piece: lane L -> key 4 c + (L >> 4), LDS slot L & 15 holds d-chunk slot ^ (key & 15)
  // V piece: lane L -> key 4 c + (L & 3), d-chunk L >> 2  (image [d-chunk][key] inside the piece)
  int k_voff[2], v_voff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = 2 * wave + j;
    const int kk = 4 * c + (lane >> 4);
    k_voff[j] = (int)(kk * p.k_stride * 2) + (((lane & 15) ^ (kk & 15)) << 4);
    const int vk = 4 * c + (lane & 3);
    v_voff[j] = (int)(vk * p.v_stride * 2) + ((lane >> 2) << 4);
  }
  auto stage = [&](int buf, int t0) {
    unsigned char* sk = fa_smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kv = k_voff[j], vv = v_voff[j];     // (local copies: see wna16_gemm_large.hip on the hipcc host-stub bug)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (fa_lds_ptr)(sk + (2 * wave + j) * 1024), 16, kv, (int)(t0 * p.k_stride * 2), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (fa_lds_ptr)(sk + KT + (2 * wave + j) * 1024), 16, vv, (int)(t0 * p.v_stride * 2), 0, 0);
    }
  };
  // fragment read addresses (lane-invariant parts)
  //  K: key = 32 b + l31, d-chunk 2 ks + kh -> l31 * 256 + (((2 ks + kh) ^ (l31 & 15)) << 4) = kaddr ^ (ks << 5)
  const int kaddr = l31 * 256 + ((kh ^ (l31 & 15)) << 4);
  //  V (transposing read; 16-lane group g, lane i of it): piece 4 ks + 2 kh + h, key i >> 2 of it,
  //  d = 32 db + 16 (g & 1) + 4 (i & 3)
  const int vaddr = KT + (2 * kh) * 1024 + ((((2 * ((lane >> 4) & 1) + ((lane & 3) >> 1)) * 4 + ((lane & 15) >> 2)) << 4)) + ((lane & 1) << 3);

  f32x16 o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;          // log2-domain running maximum; this lane's share of the row sum

  const int kv_end = p.causal ? min(len, q0 + BM + off) : len;
  const int ntile = (kv_end + BN - 1) / BN;
  // tiles this wave computes (causal: up to its diagonal), and how many of them need no masking for its 32 rows
  const int L = p.causal ? min(ntile, (wq0 + off + 31) / BN + 1) : ntile;
  const int F = slope2 != 0.f ? 0 : min(L, p.causal ? min((wq0 + off + 1) / BN, len / BN) : len / BN);

  // ---- pieces of a tile ------------------------------------------------------------------------------------------------------
  // S^T = K . Q^T of tile `it` into dst: lane (q = l31) gets keys 32 b + 8 (r >> 2) + 4 kh + (r & 3).  The two 32-key
  // blocks are two independent accumulation chains, interleaved so that no MFMA waits for its predecessor.
  auto do_qk = [&](int it, f32x16 (&dst)[2]) __attribute__((always_inline)) {
    const unsigned char* sk = fa_smem + (it % 3) * STAGE;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // (the addresses are made opaque per tile: otherwise hipcc hoists the 16 XOR-ed fragment addresses x 3 ring slots
    //  out of the tile loop and SPILLS them -- one v_xor per read is cheaper than a scratch reload)
    int ka = kaddr, qa = qaddr;
    asm volatile("" : "+v"(ka), "+v"(qa));
    // fragments of k-step ks + 1 are read while the two MFMAs of step ks run (left alone hipcc issues each read right in
    // front of its MFMA and waits lgkmcnt(0): ~100 exposed cycles per pair)
    u32x4 fr[2][3];
    auto rd = [&](int ks, u32x4 (&f)[3]) __attribute__((always_inline)) {
      f[0] = *reinterpret_cast<const u32x4*>(qlds + (qa ^ (ks << 5)));
      f[1] = *reinterpret_cast<const u32x4*>(sk + (ka ^ (ks << 5)));
      f[2] = *reinterpret_cast<const u32x4*>(sk + 8192 + (ka ^ (ks << 5)));
    };
    rd(0, fr[0]);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks < 7) rd(ks + 1, fr[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      dst[0] = fa_mfma32<T>(fr[ks & 1][1], fr[ks & 1][0], ks == 0 ? zero : dst[0]);
      dst[1] = fa_mfma32<T>(fr[ks & 1][2], fr[ks & 1][0], ks == 0 ? zero : dst[1]);
    }
  };
  // exchange a value with the lane holding the other 32 keys of the same row
  auto other_half = [&](float x) __attribute__((always_inline)) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    return kh ? __builtin_bit_cast(float, (unsigned)sw[0]) : __builtin_bit_cast(float, (unsigned)sw[1]);
  };
  // probabilities (in src) -> PV B-operand fragments: k-slot 2 b + j holds tile keys 32 b + 16 j + 8 kh .. + 7 of row l31
  auto make_pf = [&](const f32x16 (&src)[2], u32x4 (&pf)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int g0 = 8 * j, g1 = 8 * j + 4;       // accumulator quads 2 j and 2 j + 1
        const uint32_t lo0 = pack2_16<BF>(src[b][g0], src[b][g0 + 1]), hi0 = pack2_16<BF>(src[b][g0 + 2], src[b][g0 + 3]);
        const uint32_t lo1 = pack2_16<BF>(src[b][g1], src[b][g1 + 1]), hi1 = pack2_16<BF>(src[b][g1 + 2], src[b][g1 + 3]);
        const auto s_lo = __builtin_amdgcn_permlane32_swap(lo0, lo1, false, false);
        const auto s_hi = __builtin_amdgcn_permlane32_swap(hi0, hi1, false, false);
        pf[2 * b + j] = u32x4{(uint32_t)s_lo[0], (uint32_t)s_hi[0], (uint32_t)s_lo[1], (uint32_t)s_hi[1]};
      }
  };
  // online softmax of an UNMASKED tile, straight-line (no branches: it is scheduled together with the next tile's QK^T).
  // Returns the factor the running output must be scaled with.
  auto softmax_fast = [&](f32x16 (&src)[2], u32x4 (&pf)[4]) __attribute__((always_inline)) -> float {
    float mx = __builtin_fmaxf(src[0][0], src[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = __builtin_fmaxf(mx, __builtin_fmaxf(src[0][r], src[1][r]));   // -> v_max3_f32
    mx *= c2;                                      // scale > 0: the maximum commutes with it
    mx = __builtin_fmaxf(mx, other_half(mx));
    const float m_new = __builtin_fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float lsum0 = 0.f, lsum1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      src[0][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(src[0][r], c2, -m_new));
      src[1][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(src[1][r], c2, -m_new));
      lsum0 += src[0][r];
      lsum1 += src[1][r];
    }
    l_run = l_run * alpha + (lsum0 + lsum1);
    make_pf(src, pf);
    return alpha;
  };
  // QK^T of tile it + 1 (into nxt) and the softmax of an unmasked tile (cur -> pf) as ONE hand-interleaved stream: per
  // k-step two MFMAs, then a slice of the softmax VALU that runs in their shadow; sched_barrier keeps hipcc from
  // regrouping them into an MFMA block and a VALU block (which is what every compiler-driven attempt produced).
  auto qk_softmax_fast = [&](int it_next, f32x16 (&nxt)[2], f32x16 (&cur)[2], u32x4 (&pf)[4]) __attribute__((always_inline)) -> float {
    const unsigned char* sk = fa_smem + (it_next % 3) * STAGE;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int ka = kaddr, qa = qaddr;
    asm volatile("" : "+v"(ka), "+v"(qa));
    u32x4 fr[2][3];
    auto rd = [&](int ks, u32x4 (&f)[3]) __attribute__((always_inline)) {
      f[0] = *reinterpret_cast<const u32x4*>(qlds + (qa ^ (ks << 5)));
      f[1] = *reinterpret_cast<const u32x4*>(sk + (ka ^ (ks << 5)));
      f[2] = *reinterpret_cast<const u32x4*>(sk + 8192 + (ka ^ (ks << 5)));
    };
    float mx = 0.f, m_new = 0.f, alpha = 1.f, lsum = 0.f;
    rd(0, fr[0]);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks < 7) rd(ks + 1, fr[(ks + 1) & 1]);
      nxt[0] = fa_mfma32<T>(fr[ks & 1][1], fr[ks & 1][0], ks == 0 ? zero : nxt[0]);
      nxt[1] = fa_mfma32<T>(fr[ks & 1][2], fr[ks & 1][0], ks == 0 ? zero : nxt[1]);
      if (ks == 0) {
        mx = __builtin_fmaxf(cur[0][0], cur[1][0]);
#pragma unroll
        for (int r = 1; r < 8; ++r) mx = __builtin_fmaxf(mx, __builtin_fmaxf(cur[0][r], cur[1][r]));
      } else if (ks == 1) {
#pragma unroll
        for (int r = 8; r < 16; ++r) mx = __builtin_fmaxf(mx, __builtin_fmaxf(cur[0][r], cur[1][r]));
        mx *= c2;
        mx = __builtin_fmaxf(mx, other_half(mx));
        m_new = __builtin_fmaxf(m_run, mx);
        alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
      } else if (ks < 6) {
        const int b = (ks - 2) >> 1, r0 = ((ks - 2) & 1) * 8;
#pragma unroll
        for (int r = r0; r < r0 + 8; ++r) {
          cur[b][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(cur[b][r], c2, -m_new));
          lsum += cur[b][r];
        }
      } else {
        const int b = ks - 6;
        if (b == 0) l_run = l_run * alpha + lsum;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int g0 = 8 * j, g1 = 8 * j + 4;
          const uint32_t lo0 = pack2_16<BF>(cur[b][g0], cur[b][g0 + 1]), hi0 = pack2_16<BF>(cur[b][g0 + 2], cur[b][g0 + 3]);
          const uint32_t lo1 = pack2_16<BF>(cur[b][g1], cur[b][g1 + 1]), hi1 = pack2_16<BF>(cur[b][g1 + 2], cur[b][g1 + 3]);
          const auto s_lo = __builtin_amdgcn_permlane32_swap(lo0, lo1, false, false);
          const auto s_hi = __builtin_amdgcn_permlane32_swap(hi0, hi1, false, false);
          pf[2 * b + j] = u32x4{(uint32_t)s_lo[0], (uint32_t)s_hi[0], (uint32_t)s_lo[1], (uint32_t)s_hi[1]};
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return alpha;
  };
  // the general form: key >= len, key > q (causal), ALiBi
  auto softmax_edge = [&](int it, f32x16 (&src)[2], u32x4 (&pf)[4]) __attribute__((always_inline)) -> float {
    const int t0 = it * BN;
    float mx = -1e30f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = t0 + 32 * b + 8 * (r >> 2) + 4 * kh + (r & 3);
        float x = src[b][r] * c2 + slope2 * (float)(key - (qrow + off));
        const int lim = p.causal ? min(qrow + off, len - 1) : len - 1;
        x = key <= lim ? x : -1e30f;
        src[b][r] = x;
        mx = __builtin_fmaxf(mx, x);
      }
    mx = __builtin_fmaxf(mx, other_half(mx));
    const float m_new = __builtin_fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float lsum = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float e = __builtin_amdgcn_exp2f(src[b][r] - m_new);
        e = src[b][r] > -1e29f ? e : 0.f;
        src[b][r] = e;
        lsum += e;
      }
    l_run = l_run * alpha + lsum;
    make_pf(src, pf);
    return alpha;
  };
  // O^T += V^T . P^T : A fragment of (d block db, k-slot ks) = two transposing reads
  auto do_pv = [&](int it, const u32x4 (&pf)[4]) __attribute__((always_inline)) {
    int va = vaddr + (it % 3) * STAGE;
    asm volatile("" : "+v"(va));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const unsigned char* vb = fa_smem + va + (4 * ks) * 1024 + db * 256;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)vb);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vb + 1024));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        o[db] = fa_mfma32<T>(u32x4{l2[0], l2[1], h2[0], h2[1]}, pf[ks], o[db]);
      }
  };

  // ---- the tile loop ---------------------------------------------------------------------------------------------------------
  // Iteration `it` (one barrier): QK^T of tile it + 1 || softmax of tile it, then PV of tile it.  The next tile's MFMAs are
  // independent of this tile's softmax VALU, so one straight-line block holds both (measured on the phase-by-phase form:
  // QK^T 1280 + softmax 1224 + PV 1032 cycles per tile with NOTHING overlapping, whatever the pairing of waves).
  // Ring of 3 tile buffers: at the top of iteration it every wave has finished tile it - 1 (the barrier), tiles it and
  // it + 1 are being read, tile it + 2 is written.
  // (Rotating the three phases of a tile between the two waves of a SIMD -- one wave's softmax VALU beside the other's MFMAs --
  // measured 0.852 / 0.837 ms against 0.794 unrotated at T = 8192 and is not in the tree: tools/lab_patches/flash_attn.hip.patch.)
  f32x16 s_a[2], s_b[2];
  stage(0, 0);
  if (ntile > 1) stage(1, BN);
  if (ntile > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (L > 0) do_qk(0, s_a);
  auto step = [&](int it, f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2]) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of tile it + 1 (issued one iteration ago)
    __builtin_amdgcn_s_barrier();
    if (it + 2 < ntile) stage((it + 2) % 3, (it + 2) * BN);
    if (it >= L) return;                                   // causal: this wave's rows are done (it still stages its pieces)
    u32x4 pf[4];
    float alpha;
    // QK^T of the NEXT tile is issued ahead of this tile's softmax: its MFMAs are independent of the softmax VALU
    // (unconditional: past this wave's last tile it multiplies stale ring data into a buffer nobody reads -- keeping the
    //  branch out puts these MFMAs and the softmax VALU in ONE scheduling region)
    if (it < F) {
      alpha = qk_softmax_fast(it + 1, s_nxt, s_cur, pf);
    } else {
      do_qk(it + 1, s_nxt);
      alpha = softmax_edge(it, s_cur, pf);
    }
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
    do_pv(it, pf);
  };
  for (int it = 0; it < ntile; it += 2) {
    step(it, s_a, s_b);
    if (it + 1 < ntile) step(it + 1, s_b, s_a);
  }
  __syncthreads();                                         // every wave is done with the K / V buffers

  // ---- normalise, transpose through LDS (wave-private 8 KiB: 32 rows x 256 B), store whole rows -----------------------------
  float l = l_run;
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, l), __builtin_bit_cast(unsigned, l), false, false);
    l = __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
  }
  const float inv = l > 0.f ? 1.f / l : 0.f;
  unsigned char* region = fa_smem + wave * 8192;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      // d = 32 db + 8 g4 + 4 kh .. + 3  ->  8-byte chunk c8 = 8 db + 2 g4 + kh of row l31; 16-byte chunk XOR (row & 15)
      const int c8 = 8 * db + 2 * g4 + kh;
      const u32x2 v = {pack2_16<BF>(o[db][4 * g4] * inv, o[db][4 * g4 + 1] * inv),
                       pack2_16<BF>(o[db][4 * g4 + 2] * inv, o[db][4 * g4 + 3] * inv)};
      *reinterpret_cast<u32x2*>(region + l31 * 256 + ((c8 ^ ((l31 & 15) << 1)) << 3)) = v;
    }
  uint16_t* obase = (uint16_t*)p.out + (size_t)(s0q + wq0) * p.o_stride + (size_t)head * HD;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = i * 4 + (lane >> 4), c16 = lane & 15;
    const u32x4 v = *reinterpret_cast<const u32x4*>(region + row * 256 + ((c16 ^ (row & 15)) << 4));
    if (wq0 + row < qlen) *reinterpret_cast<u32x4*>(obase + (size_t)row * p.o_stride + c16 * 8) = v;
  }
}

// ---------------------------------------------------------------------------
// Prefill WITH cached context (the context_attention_fwd role,
// aphrodite/attention/ops/prefix_prefill.py:696-858, kernel :58-255): every new
// token attends to the sequence's cached context, read from the PAGED KV cache
// (K [NB,Hkv,hd/x,block,x], V [NB,Hkv,hd,block]; fp8 caches are dequantised as
// float(fp8) * scale and rounded to the query dtype, :131-134,178-181), plus the
// causal part of the new tokens.  Same tile machine as the kernel above; only the
// K/V staging differs between the two phases.
// ---------------------------------------------------------------------------
struct CAParams {
  void* out;
  const void* q;
  const void* k;
  const void* v;
  const void* k_cache;
  const void* v_cache;
  const int32_t* block_tables;   // [B, max_blocks]
  const int32_t* q_start_loc;    // [B+1] (only [b] is read: start row of sequence b)
  const int32_t* seq_lens;       // [B] context + new
  const int32_t* ctx_lens;       // [B]
  const float* alibi;
  int num_heads, num_kv_heads, max_blocks, block_size, x;
  int64_t q_stride, k_stride, v_stride, o_stride;
  float scale, k_scale, v_scale;
  int window;                    // sliding window (0 = off): keys with qpos - kpos >= window are masked
};

template <typename T, int KV>
__device__ __forceinline__ uint16_t cache_elem_to_t(const void* base, int64_t idx, float scale) {
  if constexpr (KV == 0) return ((const uint16_t*)base)[idx];
  else return T::from_f32(fp8_to_f32<KV == 2>(((const uint8_t*)base)[idx]) * scale);
}

template <typename T, int KV, int HD>
__global__ __launch_bounds__(256) void context_attn_kernel(CAParams p) {
  constexpr int NCH = HD / 8;
  constexpr int SWZ = (NCH & -NCH) - 1;
  constexpr int NKS = HD / 32;
  constexpr int NDT = HD / 16;
  __shared__ __attribute__((aligned(16))) uint16_t k_lds[FA_BN * HD];
  __shared__ __attribute__((aligned(16))) uint16_t vt_lds[HD * FA_VT_STRIDE];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4;
  const int c = lane & 15;
  const int head = blockIdx.y;
  const int seq = blockIdx.z;
  const int kvh = head / (p.num_heads / p.num_kv_heads);
  const int s0 = p.q_start_loc[seq];
  const int ctx = p.ctx_lens[seq];
  const int len = p.seq_lens[seq] - ctx;        // new tokens
  if (len <= 0) return;
  const int ntiles = (len + FA_BM - 1) / FA_BM;
  const int tile = ntiles - 1 - (int)blockIdx.x;
  if (tile < 0) return;
  const int q0 = tile * FA_BM;
  const int qrow = q0 + 16 * wave + c;
  const bool qvalid = qrow < len;
  const int qpos = ctx + qrow;                  // absolute position of this lane's query

  u32x4 qf[NKS];
  {
    const uint16_t* qp = (const uint16_t*)p.q + (size_t)(s0 + min(qrow, len - 1)) * p.q_stride + (size_t)head * HD;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const u32x4*>(qp + 32 * ks + 8 * g);
  }
  const float slope = p.alibi ? p.alibi[head] : 0.f;

  f32x4 o[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;

  const uint16_t* kbase = (const uint16_t*)p.k + (size_t)s0 * p.k_stride + (size_t)kvh * HD;
  const uint16_t* vbase = (const uint16_t*)p.v + (size_t)s0 * p.v_stride + (size_t)kvh * HD;
  const int32_t* bt = p.block_tables + (size_t)seq * p.max_blocks;
  const int new_end = min(len, q0 + FA_BM);     // causal bound within the new tokens
  const int nctx_tiles = (ctx + FA_BN - 1) / FA_BN;
  const int nnew_tiles = (new_end + FA_BN - 1) / FA_BN;

  for (int it = 0; it < nctx_tiles + nnew_tiles; ++it) {
    const bool paged = it < nctx_tiles;
    const int t0 = paged ? it * FA_BN : (it - nctx_tiles) * FA_BN;   // first key of the tile within its phase
    const int phase_len = paged ? ctx : len;
    __syncthreads();
    for (int i = threadIdx.x; i < FA_BN * NCH; i += 256) {
      const int tok = i / NCH, ch = i % NCH;
      const int ta = min(t0 + tok, phase_len - 1);
      const bool live = t0 + tok < phase_len;
      u16x8 kk, vv;
      if (paged) {
        const int64_t blk = bt[ta / p.block_size];
        const int off = ta % p.block_size;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int d = 8 * ch + j;
          const int64_t kidx = (((blk * p.num_kv_heads + kvh) * (HD / p.x) + d / p.x) * p.block_size + off) * p.x + d % p.x;
          const int64_t vidx = ((blk * p.num_kv_heads + kvh) * HD + d) * p.block_size + off;
          kk[j] = cache_elem_to_t<T, KV>(p.k_cache, kidx, p.k_scale);
          vv[j] = cache_elem_to_t<T, KV>(p.v_cache, vidx, p.v_scale);
        }
      } else {
        kk = *reinterpret_cast<const u16x8*>(kbase + (size_t)ta * p.k_stride + 8 * ch);
        vv = *reinterpret_cast<const u16x8*>(vbase + (size_t)ta * p.v_stride + 8 * ch);
      }
      if (!live) vv = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
      *reinterpret_cast<u16x8*>(&k_lds[tok * HD + 8 * (ch ^ (tok & SWZ))]) = kk;
#pragma unroll
      for (int j = 0; j < 8; ++j) vt_lds[(8 * ch + j) * FA_VT_STRIDE + tok] = vv[j];
    }
    __syncthreads();
    const bool wave_active = paged || (t0 <= q0 + 16 * wave + 15);
    if (wave_active) {
      f32x4 s[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        s[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int tok = 16 * h + c;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const int ch = (4 * ks + g) ^ (tok & SWZ);
          u32x4 kf = *reinterpret_cast<const u32x4*>(&k_lds[tok * HD + 8 * ch]);
          s[h] = fa_mfma<T>(kf, qf[ks], s[h]);
        }
      }
      float pv[2][4];
      float mx = -1e30f;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int tok = t0 + 16 * h + 4 * g + r;          // key index within the phase
          const int kpos = paged ? tok : ctx + tok;         // absolute key position
          float xv = s[h][r] * p.scale + slope * (float)(kpos - qpos);
          bool ok = tok < phase_len && kpos <= qpos;
          if (p.window > 0 && qpos - kpos >= p.window) ok = false;
          xv = ok ? xv : -1e30f;
          pv[h][r] = xv;
          mx = __builtin_fmaxf(mx, xv);
        }
      mx = __builtin_fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = __builtin_fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = __builtin_fmaxf(m_run, mx);
      const float alpha = __expf(m_run - m_new);
      m_run = m_new;
      float lsum = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float e = pv[h][r] > -1e29f ? __expf(pv[h][r] - m_new) : 0.f;
          pv[h][r] = e;
          lsum += e;
        }
      l_run = l_run * alpha + lsum;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) o[dt] *= alpha;
      u32x4 pf;
      pf[0] = fa_pack2<T>(pv[0][0], pv[0][1]);
      pf[1] = fa_pack2<T>(pv[0][2], pv[0][3]);
      pf[2] = fa_pack2<T>(pv[1][0], pv[1][1]);
      pf[3] = fa_pack2<T>(pv[1][2], pv[1][3]);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const uint16_t* vr = &vt_lds[(16 * dt + c) * FA_VT_STRIDE + 4 * g];
        u32x2 lo = *reinterpret_cast<const u32x2*>(vr);
        u32x2 hi = *reinterpret_cast<const u32x2*>(vr + 16);
        u32x4 vf = {lo[0], lo[1], hi[0], hi[1]};
        o[dt] = fa_mfma<T>(vf, pf, o[dt]);
      }
    }
  }

  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  if (qvalid) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    typename T::storage* op = (typename T::storage*)p.out + (size_t)(s0 + qrow) * p.o_stride + (size_t)head * HD;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      u16x4 r = {T::from_f32(o[dt][0] * inv), T::from_f32(o[dt][1] * inv), T::from_f32(o[dt][2] * inv),
                 T::from_f32(o[dt][3] * inv)};
      *reinterpret_cast<u16x4*>(op + 16 * dt + 4 * g) = r;
    }
  }
}

}  // namespace aphro

using namespace aphro;

// ---------------------------------------------------------------------------
// Prefill with cached context on the third-generation tile machine: the context of every sequence is
// gathered ONCE from the paged cache into contiguous [token][kv head][hd] rows (dequantised like
// prefix_prefill.py:131-134,178-181: float(fp8) * scale rounded to the query type), the new tokens' k / v are appended,
// and flash_attn_varlen_v3_kernel runs over keys = context + new with the query rows offset by the context length.
// The gather is a streaming copy (2 x the context bytes); the attention then reads K/V tiles with direct-to-LDS loads
// instead of per-element cache gathers.  hd 128, no sliding window; everything else stays on context_attn_kernel.
// ---------------------------------------------------------------------------
__global__ void ca_prefix_kernel(const int32_t* __restrict__ seq_lens, int32_t* __restrict__ cu_k, int batch) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int acc = 0;
    for (int b = 0; b < batch; ++b) { cu_k[b] = acc; acc += seq_lens[b]; }
    cu_k[batch] = acc;
  }
}

// grid (32-token tiles, kv heads, sequences), 512 threads: thread -> token (fast index), 8-wide d chunk
template <typename T, int KV>
__global__ __launch_bounds__(512) void ca_gather_kernel(CAParams p, const int32_t* __restrict__ cu_k,
                                                        uint16_t* __restrict__ kc_out, uint16_t* __restrict__ vc_out, int hd) {
  const int seq = blockIdx.z, h = blockIdx.y;
  const int tok = blockIdx.x * 32 + (threadIdx.x & 31);
  const int total = p.seq_lens[seq];
  if (tok >= total) return;
  const int ctx = p.ctx_lens[seq];
  for (int chunk = threadIdx.x >> 5; chunk * 8 < hd; chunk += 16) {      // d = 8 chunk .. + 7 (head 256: two trips)
    const size_t orow = ((size_t)(cu_k[seq] + tok) * p.num_kv_heads + h) * hd + chunk * 8;
    u16x8 kk, vv;
    if (tok < ctx) {
      const int64_t blk = p.block_tables[(size_t)seq * p.max_blocks + tok / p.block_size];
      const int boff = tok % p.block_size;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int d = chunk * 8 + j;
        const int64_t ki = (((blk * p.num_kv_heads + h) * (hd / p.x) + d / p.x) * p.block_size + boff) * p.x + d % p.x;
        const int64_t vi = ((blk * p.num_kv_heads + h) * hd + d) * p.block_size + boff;
        kk[j] = cache_elem_to_t<T, KV>(p.k_cache, ki, p.k_scale);
        vv[j] = cache_elem_to_t<T, KV>(p.v_cache, vi, p.v_scale);
      }
    } else {
      const size_t row = (size_t)(p.q_start_loc[seq] + tok - ctx);
      kk = *reinterpret_cast<const u16x8*>((const uint16_t*)p.k + row * p.k_stride + (size_t)h * hd + chunk * 8);
      vv = *reinterpret_cast<const u16x8*>((const uint16_t*)p.v + row * p.v_stride + (size_t)h * hd + chunk * 8);
    }
    *reinterpret_cast<u16x8*>(kc_out + orow) = kk;
    *reinterpret_cast<u16x8*>(vc_out + orow) = vv;
  }
}

extern "C" int aphro_context_attention(void* out, const void* q, const void* k, const void* v, const void* k_cache,
                                       const void* v_cache, const int32_t* block_tables,
                                       const int32_t* q_start_loc, const int32_t* seq_lens,
                                       const int32_t* ctx_lens, int batch, int max_query_len, int max_blocks,
                                       int num_heads, int num_kv_heads, int head_size, int block_size, int x,
                                       int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride,
                                       float scale, float k_scale, float v_scale, const float* alibi_slopes,
                                       int sliding_window, int dtype, int kv_dtype, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "context_attention: dtype must be f16 or bf16");
  APHRO_CHECK(kv_dtype >= APHRO_KV_AUTO && kv_dtype <= APHRO_KV_FP8_E5M2, "Unsupported data type of kv cache: %d", kv_dtype);
  APHRO_CHECK(num_kv_heads > 0 && num_heads % num_kv_heads == 0, "context_attention: bad head counts");
  APHRO_CHECK(q_stride % 8 == 0 && k_stride % 8 == 0 && v_stride % 8 == 0 && o_stride % 4 == 0,
              "context_attention: strides must be multiples of 8");
  APHRO_CHECK(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0, "context_attention: 16-byte alignment");
  APHRO_CHECK(block_size > 0 && x > 0 && head_size % x == 0, "context_attention: bad cache geometry");
  if (batch == 0 || max_query_len == 0) return APHRO_OK;
  CAParams p;
  p.out = out; p.q = q; p.k = k; p.v = v; p.k_cache = k_cache; p.v_cache = v_cache;
  p.block_tables = block_tables; p.q_start_loc = q_start_loc; p.seq_lens = seq_lens; p.ctx_lens = ctx_lens;
  p.alibi = alibi_slopes; p.num_heads = num_heads; p.num_kv_heads = num_kv_heads; p.max_blocks = max_blocks;
  p.block_size = block_size; p.x = x; p.q_stride = q_stride; p.k_stride = k_stride; p.v_stride = v_stride;
  p.o_stride = o_stride; p.scale = scale; p.k_scale = k_scale; p.v_scale = v_scale; p.window = sliding_window;
  dim3 grid((unsigned)((max_query_len + FA_BM - 1) / FA_BM), (unsigned)num_heads, (unsigned)batch);
#define CA_L(TT, KVV, HDV) hipLaunchKernelGGL((context_attn_kernel<TT, KVV, HDV>), grid, dim3(256), 0, (hipStream_t)stream, p)
#define CA_K(TT, HDV) { if (kv_dtype == APHRO_KV_AUTO) CA_L(TT, 0, HDV); else if (kv_dtype == APHRO_KV_FP8_E4M3) CA_L(TT, 1, HDV); else CA_L(TT, 2, HDV); }
#define CA_T(HDV) { if (dtype == APHRO_F16) CA_K(Half, HDV) else CA_K(BFloat, HDV) }
  switch (head_size) {
    case 64: CA_T(64) break;
    case 96: CA_T(96) break;
    case 128: CA_T(128) break;
    case 256: CA_T(256) break;
    default:
      set_error("context_attention: unsupported head_size=%d", head_size);
      return APHRO_ERR_INVALID;
  }
#undef CA_T
#undef CA_K
#undef CA_L
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}


// One dispatcher for the three prefill kernel generations (p fully set up except nqt_max / xcd_remap): third generation
// for head 128, sequences >= 1024, no sliding window; second for head 64 / 128 (always when the keys carry a context
// offset: its 64-key tiles are what the gathered-context path wants); first generation for the rest.
static int fa_dispatch(FAParams p, int head_size, int dtype, int batch, int max_query_len, int max_key_len, hipStream_t st) {
  if (head_size == 128 && max_key_len >= 1024 && p.window <= 0 && !APHRO_LAB_ENV_INT("APHRO_FA_NO_V3", 0)) {
    p.nqt_max = (max_query_len + 255) / 256;
    p.xcd_remap = (batch * p.num_kv_heads) % 8 == 0 && !knobs().fa_no_xcd;
    // fourth generation (one wave per SIMD, two 32-row blocks per wave: flash_attn_v4.hip) from 4096 keys on (measured on
    // T = 8192 / 4096 / 2048 / 1111 causal, Hq 32 / Hkv 8: 0.583 / 0.174 / 0.081 / 0.046 ms against 0.62 / 0.182 / 0.078 /
    // 0.044 ms of the third generation); APHRO_FA_V4_MIN_KEYS=<n>: threshold (a huge n: third generation always)
    if (max_key_len >= knobs().fa_v4_min_keys) return fa_v4_launch(p, dtype, batch, st);
    dim3 grid3((unsigned)(p.nqt_max * p.num_heads * batch));
    static bool attr_set_dev[APHRO_MAX_DEVICES] = {}; bool& attr_set = attr_set_dev[device_slot()];
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)flash_attn_varlen_v3_kernel<Half>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840) != hipSuccess ||
          hipFuncSetAttribute((const void*)flash_attn_varlen_v3_kernel<BFloat>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840) != hipSuccess) {
        set_error("flash attention: cannot raise the dynamic LDS limit");
        return APHRO_ERR_LAUNCH;
      }
      attr_set = true;
    }
    if (dtype == APHRO_F16) hipLaunchKernelGGL((flash_attn_varlen_v3_kernel<Half>), grid3, dim3(512), 163840, st, p);
    else hipLaunchKernelGGL((flash_attn_varlen_v3_kernel<BFloat>), grid3, dim3(512), 163840, st, p);
    APHRO_LAUNCH_CHECK();
    return APHRO_OK;
  }
  p.nqt_max = 0; p.xcd_remap = 0;
  // two 16-row query tiles per wave (128-row workgroups) once the sequences are long enough to
  // fill the chip with them; head 256 keeps one tile (registers)
  const bool offset_keys = p.cu_seqlens_k != nullptr;
  const int qt = ((max_query_len >= 512 || (offset_keys && max_query_len >= 128)) && head_size <= 128 && !APHRO_LAB_ENV_INT("APHRO_FA_QT1", 0)) ? 2 : 1;
  dim3 grid((unsigned)((max_query_len + 64 * qt - 1) / (64 * qt)), (unsigned)p.num_heads, (unsigned)batch);
  const bool v2 = qt == 2 && (head_size == 64 || head_size == 128) && !APHRO_LAB_ENV_INT("APHRO_FA_V1", 0);
  // 256-row (8-wave) workgroups measured slower at T = 8192 (376 vs 413 TFLOP/s): opt-in only
  const bool v2w8 = v2 && head_size == 128 && APHRO_LAB_ENV_INT("APHRO_FA_W8", 0) != 0;
  if (v2w8) grid.x = (unsigned)((max_query_len + 255) / 256);
#define FA_L(TT, HDV) { if (v2w8) hipLaunchKernelGGL((flash_attn_varlen_v2_kernel<TT, 128, 2, 8>), grid, dim3(512), 0, st, p); \
                        else if (v2) hipLaunchKernelGGL((flash_attn_varlen_v2_kernel<TT, (HDV == 64 ? 64 : 128), 2, 4>), grid, dim3(256), 0, st, p); \
                        else if (qt == 2) hipLaunchKernelGGL((flash_attn_varlen_kernel<TT, HDV, (HDV <= 128 ? 2 : 1)>), grid, dim3(256), 0, st, p); \
                        else hipLaunchKernelGGL((flash_attn_varlen_kernel<TT, HDV, 1>), grid, dim3(256), 0, st, p); }
#define FA_T(HDV) if (dtype == APHRO_F16) FA_L(Half, HDV) else FA_L(BFloat, HDV)
  switch (head_size) {
    case 64: FA_T(64) break;
    case 96: FA_T(96) break;
    case 128: FA_T(128) break;
    case 256: FA_T(256) break;
    default:
      set_error("flash attention: unsupported head_size=%d", head_size);
      return APHRO_ERR_INVALID;
  }
#undef FA_T
#undef FA_L
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

// Scratch of aphro_context_attention_gathered: the gathered K and V rows of every sequence (context + new tokens) in
// the query type, plus the key-row offsets.
extern "C" size_t aphro_context_attention_workspace_bytes(int64_t total_kv_tokens, int batch, int num_kv_heads, int head_size) {
  return 2 * (size_t)total_kv_tokens * num_kv_heads * head_size * 2 + ((size_t)(batch + 1) * 4 + 255) / 256 * 256;
}

// Same arguments as aphro_context_attention plus: max_seq_len = max(seq_lens) (context + new), total_kv_tokens >=
// sum(seq_lens), workspace.  Head sizes 64 / 96 / 128 / 256, ALiBi, sliding window: the context is gathered once and the
// prefill tile machines run over context + new tokens (third generation for head 128 / >= 1024 keys / no window, second
// for head 64 / 128, first for the rest) -- round 3: the scalar-gather aphro_context_attention kernel is no longer on any
// path the Python op takes.
extern "C" int aphro_context_attention_gathered(void* out, const void* q, const void* k, const void* v, const void* k_cache,
                                                const void* v_cache, const int32_t* block_tables, const int32_t* q_start_loc,
                                                const int32_t* seq_lens, const int32_t* ctx_lens, int batch, int max_query_len,
                                                int max_seq_len, int64_t total_kv_tokens, int max_blocks, int num_heads,
                                                int num_kv_heads, int head_size, int block_size, int x, int64_t q_stride,
                                                int64_t k_stride, int64_t v_stride, int64_t o_stride, float scale, float k_scale,
                                                float v_scale, const float* alibi_slopes, int sliding_window, int dtype,
                                                int kv_dtype, void* workspace, size_t workspace_bytes, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "context_attention: dtype must be f16 or bf16");
  APHRO_CHECK(kv_dtype >= APHRO_KV_AUTO && kv_dtype <= APHRO_KV_FP8_E5M2, "Unsupported data type of kv cache: %d", kv_dtype);
  APHRO_CHECK(head_size == 64 || head_size == 96 || head_size == 128 || head_size == 256,
              "context_attention_gathered: head_size %d (64 / 96 / 128 / 256)", head_size);
  APHRO_CHECK(num_kv_heads > 0 && num_heads % num_kv_heads == 0, "context_attention: bad head counts");
  APHRO_CHECK(q_stride % 8 == 0 && k_stride % 8 == 0 && v_stride % 8 == 0 && o_stride % 8 == 0, "context_attention: strides must be multiples of 8");
  APHRO_CHECK(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)out % 16) == 0,
              "context_attention: 16-byte alignment");
  APHRO_CHECK(block_size > 0 && x > 0 && head_size % x == 0, "context_attention: bad cache geometry");
  if (batch == 0 || max_query_len == 0) return APHRO_OK;
  const size_t need = aphro_context_attention_workspace_bytes(total_kv_tokens, batch, num_kv_heads, head_size);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("context_attention_gathered: workspace %zu < %zu bytes", workspace_bytes, need);
    return APHRO_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t kv_bytes = (size_t)total_kv_tokens * num_kv_heads * head_size * 2;
  uint16_t* kc = (uint16_t*)workspace;
  uint16_t* vc = (uint16_t*)((char*)workspace + kv_bytes);
  int32_t* cu_k = (int32_t*)((char*)workspace + 2 * kv_bytes);
  CAParams g;
  g.out = nullptr; g.q = nullptr; g.k = k; g.v = v; g.k_cache = k_cache; g.v_cache = v_cache;
  g.block_tables = block_tables; g.q_start_loc = q_start_loc; g.seq_lens = seq_lens; g.ctx_lens = ctx_lens;
  g.alibi = nullptr; g.num_heads = num_heads; g.num_kv_heads = num_kv_heads; g.max_blocks = max_blocks;
  g.block_size = block_size; g.x = x; g.q_stride = q_stride; g.k_stride = k_stride; g.v_stride = v_stride;
  g.o_stride = o_stride; g.scale = scale; g.k_scale = k_scale; g.v_scale = v_scale; g.window = 0;
  hipLaunchKernelGGL(ca_prefix_kernel, dim3(1), dim3(64), 0, st, seq_lens, cu_k, batch);
  dim3 ggrid((unsigned)((max_seq_len + 31) / 32), (unsigned)num_kv_heads, (unsigned)batch);
#define CG_L(TT, KVV) hipLaunchKernelGGL((ca_gather_kernel<TT, KVV>), ggrid, dim3(512), 0, st, g, cu_k, kc, vc, head_size)
#define CG_K(TT) { if (kv_dtype == APHRO_KV_AUTO) CG_L(TT, 0); else if (kv_dtype == APHRO_KV_FP8_E4M3) CG_L(TT, 1); else CG_L(TT, 2); }
  if (dtype == APHRO_F16) CG_K(Half) else CG_K(BFloat)
#undef CG_K
#undef CG_L
  APHRO_LAUNCH_CHECK();
  FAParams p;
  p.out = out; p.q = q; p.k = kc; p.v = vc; p.cu_seqlens = q_start_loc; p.cu_seqlens_k = cu_k; p.alibi = alibi_slopes;
  p.num_heads = num_heads; p.num_kv_heads = num_kv_heads;
  p.q_stride = q_stride; p.k_stride = (int64_t)num_kv_heads * head_size; p.v_stride = p.k_stride; p.o_stride = o_stride;
  p.scale = scale; p.causal = 1; p.window = sliding_window > 0 ? sliding_window : 0;
  return fa_dispatch(p, head_size, dtype, batch, max_query_len, max_seq_len, st);
}


// window > 0 (causal only): query i sees keys j with i - j < window -- flash_attn_varlen_func's window_size = (left, *) under
// causal = True is window = left + 1 (rocm_flash_attn.py:497-507).  The windowed form runs on the first / second generation
// tile machines (their key loops start at the first visible key tile).
static int fa_varlen_entry(void* out, const void* q, const void* k, const void* v,
                           const int32_t* cu_seqlens, int batch, int max_seqlen, int num_heads,
                           int num_kv_heads, int head_size, int64_t q_stride, int64_t k_stride,
                           int64_t v_stride, float scale, int causal, const float* alibi_slopes, int window,
                           int dtype, void* stream) {
  APHRO_CHECK(window <= 0 || causal, "flash_attn_varlen: a sliding window needs causal attention");
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "flash_attn_varlen: dtype must be f16 or bf16");
  APHRO_CHECK(num_kv_heads > 0 && num_heads % num_kv_heads == 0, "flash_attn_varlen: bad head counts");
  APHRO_CHECK(q_stride % 8 == 0 && k_stride % 8 == 0 && v_stride % 8 == 0, "flash_attn_varlen: strides must be multiples of 8");
  APHRO_CHECK(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0, "flash_attn_varlen: 16-byte alignment");
  if (batch == 0 || max_seqlen == 0) return APHRO_OK;
  FAParams p;
  p.out = out; p.q = q; p.k = k; p.v = v; p.cu_seqlens = cu_seqlens; p.alibi = alibi_slopes;
  p.num_heads = num_heads; p.num_kv_heads = num_kv_heads;
  p.q_stride = q_stride; p.k_stride = k_stride; p.v_stride = v_stride;
  p.scale = scale; p.causal = causal;
  p.cu_seqlens_k = nullptr; p.o_stride = (int64_t)num_heads * head_size; p.nqt_max = 0; p.xcd_remap = 0;
  p.window = window > 0 ? window : 0;
  return fa_dispatch(p, head_size, dtype, batch, max_seqlen, max_seqlen, (hipStream_t)stream);
}

extern "C" int aphro_flash_attn_varlen(void* out, const void* q, const void* k, const void* v,
                                       const int32_t* cu_seqlens, int batch, int max_seqlen, int num_heads,
                                       int num_kv_heads, int head_size, int64_t q_stride, int64_t k_stride,
                                       int64_t v_stride, float scale, int causal, const float* alibi_slopes,
                                       int dtype, void* stream) {
  return fa_varlen_entry(out, q, k, v, cu_seqlens, batch, max_seqlen, num_heads, num_kv_heads, head_size, q_stride, k_stride,
                         v_stride, scale, causal, alibi_slopes, 0, dtype, stream);
}

extern "C" int aphro_flash_attn_varlen_window(void* out, const void* q, const void* k, const void* v,
                                              const int32_t* cu_seqlens, int batch, int max_seqlen, int num_heads,
                                              int num_kv_heads, int head_size, int64_t q_stride, int64_t k_stride,
                                              int64_t v_stride, float scale, int causal, const float* alibi_slopes,
                                              int window, int dtype, void* stream) {
  return fa_varlen_entry(out, q, k, v, cu_seqlens, batch, max_seqlen, num_heads, num_kv_heads, head_size, q_stride, k_stride,
                         v_stride, scale, causal, alibi_slopes, window, dtype, stream);
}

