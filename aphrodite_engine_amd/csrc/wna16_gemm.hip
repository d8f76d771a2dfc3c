// W4A16 (GPTQ / K-packed int4) small-M GEMM for gfx950 -- the decode hot kernel.
//
// Replaces the reference's exllama kernel gemm_half_q_half_gptq_4bit_kernel
// (kernels/quantization/gptq/q_gemm.cu:190-326), its shuffle / reconstruct
// kernels (:856-965, :1394-1434, :1569-1657) and fills the gptq_marlin_gemm
// role for M <= 64.  Design (DESIGN.md "wna16_gemm"):
//   * HBM-bound: every packed int4 word is read exactly once per call with
//     16 B/lane coalesced loads (256 B contiguous per 16 lanes).
//   * The exllama word layout (8 consecutive k of ONE column per dword, nibble
//     pairs (k, k+1) in the two halves) is already the B fragment of
//     v_mfma_f32_16x16x32_f16: lane (g = lane>>4, c = lane&15) holds
//     k = 8g..8g+7 of one column.  A dwordx4 load gives the lane 4 adjacent
//     columns -> 4 MFMA n-tiles whose column index is n0 + 4c + t.
//   * int4 -> f16 in registers: (w & 0x000f000f) | 0x6400 is half2(1024 + q);
//     subtracting half2(1024 + z) is exact.  The group scale is applied to
//     the fp32 partial sum of each 128-k group (exact integer weights in the
//     MFMA, fp32 everywhere after) -- strictly more accurate than the
//     reference's fp16 dot + fp16 atomics.
//   * A workgroup = 8 waves that split the K range of ONE column tile and
//     reduce through LDS (deterministic, no atomics); an optional second
//     level of split-K across workgroups goes through an fp32 workspace and
//     a tiny reduce kernel.
#include "common.h"

namespace aphro {

constexpr int NW = 8;   // waves per workgroup, generic kernel
constexpr int FNW = 4;  // waves per workgroup, fast kernel (3 workgroups resident per CU)

// ---------------------------------------------------------------------------
// int4 -> f16x8 for one exllama-ordered word.  zh = half2(1024 + z),
// zh16 = half2(-64 - z)  (z = effective zero point)
// ---------------------------------------------------------------------------
__device__ __forceinline__ f16x8 dq8_exl(uint32_t w, f16x2 zh, f16x2 zh16) {
  const f16x2 inv16 = {(f16)0.0625f, (f16)0.0625f};
  uint32_t q0 = (w & 0x000f000fu) | 0x64006400u;  // 1024 + (e0, e1)
  uint32_t q1 = (w & 0x00f000f0u) | 0x64006400u;  // 1024 + 16*(e2, e3)
  w >>= 8;
  uint32_t q2 = (w & 0x000f000fu) | 0x64006400u;  // e4, e5
  uint32_t q3 = (w & 0x00f000f0u) | 0x64006400u;  // e6, e7
  f16x2 d0 = __builtin_bit_cast(f16x2, q0) - zh;
  f16x2 d1 = __builtin_bit_cast(f16x2, q1) * inv16 + zh16;
  f16x2 d2 = __builtin_bit_cast(f16x2, q2) - zh;
  f16x2 d3 = __builtin_bit_cast(f16x2, q3) * inv16 + zh16;
  f16x8 r;
  r[0] = d0[0]; r[1] = d0[1]; r[2] = d1[0]; r[3] = d1[1];
  r[4] = d2[0]; r[5] = d2[1]; r[6] = d3[0]; r[7] = d3[1];
  return r;
}

template <typename T>
__device__ __forceinline__ f16x8 load_a_frag(const uint16_t* p);
template <>
__device__ __forceinline__ f16x8 load_a_frag<Half>(const uint16_t* p) {
  return *reinterpret_cast<const f16x8*>(p);
}
template <>
__device__ __forceinline__ f16x8 load_a_frag<BFloat>(const uint16_t* p) {
  // bf16 activations are converted to f16 for the MFMA (the
  // reference's GPTQ/AWQ kernels are fp16-only, gptq.py:54-55).
  u16x8 v = *reinterpret_cast<const u16x8*>(p);
  f16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = __builtin_bit_cast(f16, bf16_bits_to_f16_bits_sat(v[i]));
  return r;
}

struct Wna16Params {
  const uint16_t* a;      // [M, lda]
  const uint16_t* apk;    // fragment-major f16 copy of a (fast path), see pack_a_kernel
  const uint32_t* qw;     // [K/8, N] exllama order
  const uint32_t* qz;     // [G, N/8]
  const uint16_t* sc;     // [G, N]
  uint16_t* c;            // [M, N]  (when ksplit == 1)
  float* partial;         // [ksplit, M, N] (when ksplit > 1)
  int M, N, K;
  int lda;
  int group_size;         // K / groups, multiple of 32
  int ksteps_per_split;   // k-steps (32 k) per blockIdx.y
  int ksplit;
  int zero_offset;
  int gshift;             // log2(group_size / 128) (fast path)
  int force_partial;      // 1: write the fp32 slab even when ksplit == 1 (fused consumer)
  uint16_t* act_packed;   // != NULL (ksplit == 1 only): columns are (gate_j, up_j) pairs; the epilogue
                          // writes silu(gate) * up as fragment-major f16 [M, N/2] for the next GEMM
  // ---- grouped (mixture-of-experts) form: every 16-row m-tile uses the weights of ONE expert ----
  int xcd_remap;          // fast kernel, ksplit in {2, 4, 8}: K slice y runs on 8 / ksplit XCDs (see the kernel); 0 = off, else
  int xcd_tiles;          // 1 + log2(8 / ksplit), and grid.x / (8 / ksplit)
  const int32_t* expert_ids;     // [M / 16] expert of each m-tile (moe_align_block_size), < 0: skip; NULL: dense
  const int32_t* num_post_pad;   // device scalar: rows >= *num_post_pad are not computed
  int64_t w_estride, z_estride, s_estride;  // per-expert strides of qw / qz (words) and sc (elements)
};

// ---- in-workgroup split-K reduction through LDS + store ---------------------
template <typename T, int VEC, int MT, int NWV>
__device__ __forceinline__ void wna16_epilogue(const Wna16Params& p, float* red, f32x4 (&acc)[MT][VEC],
                                               int lane, int wave, int g, int m0, int ncol, int ky) {
  // ---- in-workgroup split-K reduction through LDS -------------------------
  // red[wave][q = i*VEC + t][lane] as float4 (lane-contiguous: conflict-free)
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < VEC; ++t)
      *reinterpret_cast<f32x4*>(&red[((wave * (MT * VEC) + i * VEC + t) * 64 + lane) * 4]) = acc[i][t];
  __syncthreads();
  // thread -> (lane l, row slot (i, r)): sums VEC adjacent columns of one row
  for (int idx = wave; idx < MT * 4; idx += NWV) {
    const int i = idx >> 2, r = idx & 3;
    const int row = m0 + 16 * i + 4 * g + r;
    float v[VEC];
#pragma unroll
    for (int t = 0; t < VEC; ++t) {
      float sum = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NWV; ++w2)
        sum += red[((w2 * (MT * VEC) + i * VEC + t) * 64 + lane) * 4 + r];
      v[t] = sum;
    }
    if (row < p.M) {
      if (p.act_packed) {
        // fused SiLU-and-mul + activation pack (same roundings as the separate ops: the GEMM
        // result is rounded to T first).  Output feature j = column / 2.
        if constexpr (VEC >= 2) {
          const int mtiles = (p.M + 15) >> 4;
          const int j = ncol >> 1;
          const int seg = j >> 7, gg = (j & 127) >> 5, u = (j & 31) >> 3;
          uint16_t* dst = p.act_packed +
                          ((((size_t)seg * 4 + u) * mtiles + (row >> 4)) * 64 + gg * 16 + (row & 15)) * 8 + (j & 7);
          uint16_t o[VEC / 2];
#pragma unroll
          for (int q = 0; q < VEC / 2; ++q) {
            const float gate = T::to_f32(T::from_f32(v[2 * q]));
            const float up = T::to_f32(T::from_f32(v[2 * q + 1]));
            uint16_t r = silu_mul_bits<T>(gate, up);
            if constexpr (!__is_same(T, Half)) r = bf16_bits_to_f16_bits_sat(r);
            o[q] = r;
          }
          if constexpr (VEC == 4) *reinterpret_cast<uint32_t*>(dst) = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
          else dst[0] = o[0];
        }
      } else if (p.ksplit == 1 && !p.force_partial) {
        uint16_t* cp = p.c + (size_t)row * p.N + ncol;
        if constexpr (VEC == 4) {
          u16x4 o = {T::from_f32(v[0]), T::from_f32(v[1]), T::from_f32(v[2]), T::from_f32(v[3])};
          *reinterpret_cast<u16x4*>(cp) = o;
        } else {
#pragma unroll
          for (int t = 0; t < VEC; ++t) cp[t] = T::from_f32(v[t]);
        }
      } else {
        float* pp = p.partial + ((size_t)ky * p.M + row) * p.N + ncol;
        if constexpr (VEC == 4) {
          *reinterpret_cast<f32x4*>(pp) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int t = 0; t < VEC; ++t) pp[t] = v[t];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Fast path (group size a multiple of 128).  A "segment" is 128 consecutive k
// (16 packed rows, one group scale).
//  * activations arrive FRAGMENT-MAJOR (pack_a_kernel or a fused producer):
//    block (seg, u, mtile) = 1 KiB, lane (g, m) holds A[16*mtile+m][128*seg +
//    32*g + 8*u .. +7] as f16.  A row-major A makes every MFMA A-fragment a
//    16-row gather (64 sectors per wave instruction, measured 2.4x slower).
//  * the weight row for (u, g) is 16*seg + 4g + u (same k as the A fragment).
//  * NSEG segments per wave is a compile-time constant: straight-line code
//    (hipcc's waitcnt insertion degrades to vmcnt(0) around runtime-conditional
//    loads); W loads run DEPTH segments ahead of their use and the issue points
//    are pinned with sched_barrier (the scheduler otherwise sinks loads next to
//    their first use, serialising HBM latency and compute).
//  * the group scale is folded into the f16 B fragment ((q - z) exact, one
//    rounding in the multiply -- the numerics of the reference's own
//    reconstruct kernels, q_gemm.cu:1427-1431) so no second accumulator set is
//    needed: ~165 VGPRs, three 4-wave workgroups resident per CU.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t v_and_or(uint32_t a, uint32_t mask, uint32_t orv) {
  uint32_t r;  // gfx9 VOP3 takes one SGPR: the mask goes in an SGPR, the magic in a VGPR
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask), "v"(orv));
  return r;
}

// int4 word -> 8 scaled f16: ((1024 + q) - (1024 + z)) * s, exact difference
__device__ __forceinline__ f16x8 dq8_exl_scaled(uint32_t w, f16x2 zh, f16x2 zh16, f16x2 sc, uint32_t magic) {
  const f16x2 inv16 = {(f16)0.0625f, (f16)0.0625f};
  uint32_t q0 = v_and_or(w, 0x000f000fu, magic);
  uint32_t q1 = v_and_or(w, 0x00f000f0u, magic);
  uint32_t w8 = w >> 8;
  uint32_t q2 = v_and_or(w8, 0x000f000fu, magic);
  uint32_t q3 = v_and_or(w8, 0x00f000f0u, magic);
  f16x2 d0 = (__builtin_bit_cast(f16x2, q0) - zh) * sc;
  f16x2 d1 = (__builtin_bit_cast(f16x2, q1) * inv16 + zh16) * sc;
  f16x2 d2 = (__builtin_bit_cast(f16x2, q2) - zh) * sc;
  f16x2 d3 = (__builtin_bit_cast(f16x2, q3) * inv16 + zh16) * sc;
  u32x4 r = {__builtin_bit_cast(uint32_t, d0), __builtin_bit_cast(uint32_t, d1),
             __builtin_bit_cast(uint32_t, d2), __builtin_bit_cast(uint32_t, d3)};
  return __builtin_bit_cast(f16x8, r);
}

template <int VEC>
struct SegMeta {          // RAW group scale / zero words of one segment: no ALU op may touch
  uint32_t sc[(VEC + 1) / 2];  // them before the segment is consumed, or the compiler has to
  uint32_t zw;                 // wait for the load -- and with it for every older weight load
};

// gfx950 buffer resource: base, no stride, byte count, DATA_FORMAT=32 (raw dword access)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

template <typename T, int VEC, int MT, int NSEG>
__global__ __launch_bounds__(FNW * 64, (VEC * MT >= 8) ? 2 : 3) void wna16_gemm_kernel(Wna16Params p) {
  constexpr int DEPTH = NSEG < 2 ? NSEG : 2;  // weight segments in flight ahead of the consumer
  constexpr int NBUF = DEPTH + 1;
  constexpr int ADEPTH = 1;
  constexpr int NA = ADEPTH + 1;
  constexpr int AUX_NT = 2;  // nontemporal: weights are read exactly once
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4;
  const int c = lane & 15;
  // K slices across workgroups: all column tiles of slice y read the SAME activation rows.  Workgroups are dealt
  // round-robin to the 8 XCDs (each with its own L2), so with the plain (x, y) order every XCD fetches every slice of the
  // activations: 8 x M x K x 2 bytes through the fabric (PMC: 38.8 MB read for the 31.4 MB down projection, profiles/
  // r3_pmc_traffic.txt).  xcd_remap gives slice y to 8 / ksplit XCDs only.
  int bx = blockIdx.x, by = blockIdx.y;
  if (p.xcd_remap) {                                // 1 + log2(XCDs per K slice); xcd_tiles = grid.x / that count (host:
    const int L = blockIdx.y * gridDim.x + blockIdx.x, xcd = L & 7, idx = L >> 3, sh = p.xcd_remap - 1;   // no division here)
    by = xcd >> sh;
    bx = (xcd & ((1 << sh) - 1)) * p.xcd_tiles + idx;
  }
  const int n0 = bx * (16 * VEC);
  const int m0 = blockIdx.z * (16 * MT);
  const int ncol = n0 + VEC * c;
  const int seg0 = (by * FNW + wave) * NSEG;  // host guarantees K == ksplit*FNW*NSEG*128

  // All global reads are buffer loads: the per-lane part of every address is a loop-invariant
  // VGPR offset, the (segment, k-step) part an SGPR offset -- no vector address arithmetic in
  // the loop (the kernel is instruction-issue bound on the CUs that host two workgroups).
  const int mtiles = (p.M + 15) >> 4;
  const uint32_t* qw_base = p.qw;
  const uint32_t* qz_base = p.qz;
  const uint16_t* sc_base = p.sc;
  if (p.expert_ids != nullptr) {  // grouped form (MT == 1): this m-tile's expert; uniform for the workgroup
    if (m0 >= *p.num_post_pad) return;
    const int e = p.expert_ids[m0 >> 4];
    if (e < 0) return;
    qw_base += (size_t)e * p.w_estride;
    qz_base += (size_t)e * p.z_estride;
    sc_base += (size_t)e * p.s_estride;
  }
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(qw_base, (uint32_t)((size_t)(p.K >> 3) * p.N * 4));
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.apk, (uint32_t)((size_t)(p.K >> 7) * 4 * mtiles * 1024));
  const int ngroups = p.K / p.group_size;
  const __amdgpu_buffer_rsrc_t rs_ = make_rsrc(sc_base, (uint32_t)((size_t)ngroups * p.N * 2));
  const __amdgpu_buffer_rsrc_t rz = make_rsrc(qz_base, (uint32_t)((size_t)ngroups * (p.N >> 3) * 4));
  const int roww = p.N * 4;                       // bytes per packed weight row
  const int voff_w = (4 * g * p.N + ncol) * 4;    // row 4g (+u via the SGPR offset), this lane's columns
  int voff_a[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)  // surplus m-tiles of the last z-block read tile mtiles-1 again (never stored)
    voff_a[i] = (min((m0 >> 4) + i, mtiles - 1) * 64 + lane) * 16;
  const int abytes = mtiles * 1024;               // bytes per (segment, u) block row of the packed A
  const int voff_s = ncol * 2;
  const int voff_z = (ncol >> 3) * 4;
  const int zshift = (ncol & 7) * 4;
  const float zoff = (float)p.zero_offset;
  // weights of the 8 fragment positions after the A pre-scale (see below)
  const f16x8 ones = {(f16)1.f, (f16)1.f, (f16)16.f, (f16)16.f, (f16)1.f, (f16)1.f, (f16)16.f, (f16)16.f};
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  f32x4 cacc[MT][VEC];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < VEC; ++t) cacc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  SegMeta<VEC> meta[2];
  uint32_t w[NBUF][4][VEC];
  u32x4 af[NA][4][MT];

  auto load_meta = [&](SegMeta<VEC>& m, int s) {
    const int grp = (seg0 + s) >> p.gshift;  // group_size / 128 is a power of two on this path
    m.zw = __builtin_amdgcn_raw_buffer_load_b32(rz, voff_z, grp * (p.N >> 3) * 4, 0);
    if constexpr (VEC == 4) {
      u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_, voff_s, grp * p.N * 2, 0);
      m.sc[0] = v[0]; m.sc[1] = v[1];
    } else {
      m.sc[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_, voff_s, grp * p.N * 2, 0);
    }
  };
  auto load_w = [&](uint32_t (&wd)[4][VEC], int s) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int soff = ((seg0 + s) * 16 + u) * roww;
      if constexpr (VEC == 4) {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, voff_w, soff, AUX_NT);
        wd[u][0] = v[0]; wd[u][1] = v[1]; wd[u][2] = v[2]; wd[u][3] = v[3];
      } else {
        u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rw, voff_w, soff, AUX_NT);
        wd[u][0] = v[0]; wd[u][1] = v[1];
      }
    }
  };
  auto load_a = [&](u32x4 (&ad)[4][MT], int s) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i)
        ad[u][i] = __builtin_amdgcn_raw_buffer_load_b128(ra, voff_a[i], ((seg0 + s) * 4 + u) * abytes, 0);
  };

  // ---- prologue: meta(0), A(0), W(0..DEPTH-1) ---------------------------------------
  load_meta(meta[0], 0);
#pragma unroll
  for (int d = 0; d < ADEPTH; ++d) load_a(af[d], d);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load_w(w[d], d);
  __builtin_amdgcn_sched_barrier(0);

#pragma unroll
  for (int s = 0; s < NSEG; ++s) {
    // straight-line code (NSEG is a template constant): hipcc's waitcnt insertion degrades to
    // vmcnt(0) around runtime-conditional loads, and the issue points are pinned with
    // sched_barrier (the scheduler otherwise sinks loads next to their first use).
    if (s + ADEPTH < NSEG) load_a(af[(s + ADEPTH) % NA], s + ADEPTH);
    if (s + DEPTH < NSEG) load_w(w[(s + DEPTH) % NBUF], s + DEPTH);
    if (s + 1 < NSEG) load_meta(meta[(s + 1) & 1], s + 1);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[MT][VEC];  // sum_k a'[m][k] * q[k][n] * 2^-24 over this 128-k group
    f32x4 rs[MT];        // sum_k a[m][k] (all 16 columns of the tile hold the same row sums)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f16x8 a[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        // nibbles 2,3,6,7 of a word are extracted in place (bits 4-7 of each half):
        // they weigh 16x, so those four k of the A fragment are scaled by 1/16
        // (inline asm on the integer lanes: hipcc miscompiles a bitcast of one vector element)
        u32x4 av = af[s % NA][u][i];
        asm("v_pk_mul_f16 %0, %1, %2" : "=v"(av[1]) : "v"(av[1]), "s"(0x2c002c00u));
        asm("v_pk_mul_f16 %0, %1, %2" : "=v"(av[3]) : "v"(av[3]), "s"(0x2c002c00u));
        a[i] = __builtin_bit_cast(f16x8, av);
        rs[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], ones, u == 0 ? zero4 : rs[i], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < VEC; ++t) {
        // a nibble in the low mantissa bits of an f16 IS the subnormal q * 2^-24 -- the
        // MFMA consumes subnormals exactly, so the unpack is four ANDs and one shift
        const uint32_t wv = w[s % NBUF][u][t];
        const uint32_t w8 = wv >> 8;
        u32x4 bq = {wv & 0x000f000fu, wv & 0x00f000f0u, w8 & 0x000f000fu, w8 & 0x00f000f0u};
        const f16x8 b = __builtin_bit_cast(f16x8, bq);
#pragma unroll
        for (int i = 0; i < MT; ++i)
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b, u == 0 ? zero4 : acc[i][t], 0, 0, 0);
      }
    }
    // group epilogue (fp32): c += s * (2^24 * acc - z * rowsum)
    const SegMeta<VEC>& m = meta[s & 1];
#pragma unroll
    for (int t = 0; t < VEC; ++t) {
      const float z = (float)((m.zw >> (zshift + 4 * t)) & 0xf) + zoff;
      const float sf = T::to_f32((uint16_t)(m.sc[t >> 1] >> (16 * (t & 1))));
      const float s24 = sf * 16777216.f;
      const float nzs = -z * sf;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        cacc[i][t] = __builtin_elementwise_fma(acc[i][t], f32x4{s24, s24, s24, s24}, cacc[i][t]);
        cacc[i][t] = __builtin_elementwise_fma(rs[i], f32x4{nzs, nzs, nzs, nzs}, cacc[i][t]);
      }
    }
  }
  wna16_epilogue<T, VEC, MT, FNW>(p, red, cacc, lane, wave, g, m0, ncol, by);
}

// Generic path (any group size that is a multiple of 32): per-segment loads.
// VEC = dwords per lane per k-step (4 -> 64-column tile, 2 -> 32, 1 -> 16)
// MT  = 16-row m-tiles (1 or 2)
template <typename T, int VEC, int MT>
__global__ __launch_bounds__(NW * 64) void wna16_gemm_generic_kernel(Wna16Params p) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [NW][MT*VEC][64][4]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4;   // k sub-block (8 k each)
  const int c = lane & 15;   // column slot / A row
  const int n0 = blockIdx.x * (16 * VEC);
  const int m0 = blockIdx.z * (16 * MT);
  const int ncol = n0 + VEC * c;  // first of this lane's VEC columns

  // K range of this workgroup, then of this wave (contiguous chunk of k-steps)
  const int total_steps = p.K >> 5;
  const int wg_begin = blockIdx.y * p.ksteps_per_split;
  const int wg_end = min(total_steps, wg_begin + p.ksteps_per_split);
  const int per_wave = (wg_end - wg_begin + NW - 1) / NW;
  int s = wg_begin + wave * per_wave;
  const int s_end = min(wg_end, s + per_wave);

  f32x4 acc[MT][VEC];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < VEC; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A row pointers (rows >= M are clamped; their results are never stored)
  const uint16_t* arow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int r = min(m0 + 16 * i + c, p.M - 1);
    arow[i] = p.a + (size_t)r * p.lda + 8 * g;
  }
  const uint32_t* wbase = p.qw + (size_t)g * p.N + ncol;
  const int steps_per_group = p.group_size >> 5;

  while (s < s_end) {
    const int grp = s / steps_per_group;
    const int seg_end = min(s_end, (grp + 1) * steps_per_group);
    // group scale / zero for this lane's VEC columns
    float scl[VEC];
    f16x2 zh[VEC], zh16[VEC];
    {
      const uint16_t* sp = p.sc + (size_t)grp * p.N + ncol;
      uint32_t zw = p.qz[(size_t)grp * (p.N >> 3) + (ncol >> 3)] >> ((ncol & 7) * 4);
#pragma unroll
      for (int t = 0; t < VEC; ++t) {
        scl[t] = T::to_f32(sp[t]);
        int z = (int)((zw >> (4 * t)) & 0xf) + p.zero_offset;
        f16 a = __builtin_bit_cast(f16, (uint16_t)(0x6400 | z));  // 1024 + z
        f16 b = (f16)(float)(-64 - z);
        zh[t] = f16x2{a, a};
        zh16[t] = f16x2{b, b};
      }
    }
    f32x4 part[MT][VEC];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int t = 0; t < VEC; ++t) part[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // 4 k-steps per trip: all loads first, then dequant + MFMA
    for (; s + 4 <= seg_end; s += 4) {
      uint32_t w[4][VEC];
      f16x8 af[4][MT];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t* wp = wbase + (size_t)(s + u) * 4 * p.N;
        if constexpr (VEC == 4) {
          u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp));
          w[u][0] = v[0]; w[u][1] = v[1]; w[u][2] = v[2]; w[u][3] = v[3];
        } else if constexpr (VEC == 2) {
          u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(wp));
          w[u][0] = v[0]; w[u][1] = v[1];
        } else {
          w[u][0] = __builtin_nontemporal_load(wp);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < MT; ++i) af[u][i] = load_a_frag<T>(arow[i] + (size_t)(s + u) * 32);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < VEC; ++t) {
          f16x8 b = dq8_exl(w[u][t], zh[t], zh16[t]);
#pragma unroll
          for (int i = 0; i < MT; ++i)
            part[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u][i], b, part[i][t], 0, 0, 0);
        }
    }
    for (; s < seg_end; ++s) {  // tail k-steps of a short segment
      const uint32_t* wp = wbase + (size_t)s * 4 * p.N;
      uint32_t w[VEC];
#pragma unroll
      for (int t = 0; t < VEC; ++t) w[t] = wp[t];
#pragma unroll
      for (int t = 0; t < VEC; ++t) {
        f16x8 b = dq8_exl(w[t], zh[t], zh16[t]);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          f16x8 af = load_a_frag<T>(arow[i] + (size_t)s * 32);
          part[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, b, part[i][t], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int t = 0; t < VEC; ++t) acc[i][t] += part[i][t] * scl[t];
  }

  wna16_epilogue<T, VEC, MT, NW>(p, red, acc, lane, wave, g, m0, ncol, (int)blockIdx.y);
}

// partial [S][M*N] fp32 -> c [M*N] (+ bias[N])
template <typename T>
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, uint16_t* __restrict__ c,
                                     const uint16_t* __restrict__ bias, int64_t mn, int N, int S) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= mn) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(partial + i);
  for (int k = 1; k < S; ++k) s += *reinterpret_cast<const f32x4*>(partial + (size_t)k * mn + i);
  if (bias) {
    int n = (int)(i % N);
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] += T::to_f32(bias[n + j]);
  }
  u16x4 o = {T::from_f32(s[0]), T::from_f32(s[1]), T::from_f32(s[2]), T::from_f32(s[3])};
  *reinterpret_cast<u16x4*>(c + i) = o;
}

// Fragment-major copy of the activations for the fast kernel: block (seg, u, mt)
// holds, for lane (g, m), the 8 halfs A[16*mt + m][128*seg + 32*g + 8*u .. +7]
// (rows >= M are zero; bf16 is widened to f16 here; act-order: column gather by
// perm, q_gemm.cu:219-226).  Row-major A makes every MFMA A-fragment a 16-row
// gather (64 sectors per wave instruction) that each of the N/64 column tiles
// would repeat; packing once turns them into 1 KiB coalesced loads.
template <typename T>
__global__ void pack_a_kernel(const uint16_t* __restrict__ a, const int32_t* __restrict__ perm,
                              uint16_t* __restrict__ out, int M, int K, int lda) {
  const int mtiles = (M + 15) >> 4;
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
  if (row < M) {
    if (perm) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = a[(size_t)row * lda + perm[k0 + j]];
    } else {
      v = *reinterpret_cast<const u16x8*>(a + (size_t)row * lda + k0);
    }
    if constexpr (!__is_same(T, Half)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = bf16_bits_to_f16_bits_sat(v[j]);
    }
  }
  *reinterpret_cast<u16x8*>(out + idx * 8) = v;
}

// a_perm[m][k] = a[m][perm[k]]   (act-order column gather, q_gemm.cu:219-226)
__global__ void permute_cols_kernel(const uint16_t* __restrict__ a, const int32_t* __restrict__ perm,
                                    uint16_t* __restrict__ out, int M, int K, int lda) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  int m = blockIdx.y;
  if (k < K) out[(size_t)m * K + k] = a[(size_t)m * lda + perm[k]];
}

// ---- load-time repack kernels ---------------------------------------------------
__device__ __forceinline__ uint32_t shuffle_word(uint32_t q) {
  // elements 0,2,4,6 -> bits[15:0]; 1,3,5,7 -> bits[31:16]  (qdq_4.cuh:17-35)
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r |= ((q >> (8 * i)) & 0xfu) << (4 * i);
    r |= ((q >> (8 * i + 4)) & 0xfu) << (4 * i + 16);
  }
  return r;
}
__device__ __forceinline__ uint32_t unshuffle_word(uint32_t q) {
  uint32_t r = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) r |= ((q >> ((j >> 1) * 4 + ((j & 1) ? 16 : 0))) & 0xfu) << (4 * j);
  return r;
}

// out[r][n] = shuffle(gather rows perm[8r..8r+7] of in)  (perm may be null)
__global__ void gptq_repack_kernel(const uint32_t* __restrict__ in, const int32_t* __restrict__ perm,
                                   uint32_t* __restrict__ out, int rows, int N) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  int r = blockIdx.y;
  if (n >= N) return;
  uint32_t q;
  if (perm) {
    q = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int src = perm[8 * r + i];
      uint32_t w = in[(size_t)(src >> 3) * N + n];
      q |= ((w >> ((src & 7) * 4)) & 0xfu) << (4 * i);
    }
  } else {
    q = in[(size_t)r * N + n];
  }
  out[(size_t)r * N + n] = shuffle_word(q);
}

// AWQ [K, N/8] (nibble p of word c = column 8c + {0,2,4,6,1,3,5,7}[p]) ->
// exllama-ordered K-packed [K/8, N]
__global__ void awq_repack_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int K,
                                  int N) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  int r = blockIdx.y;  // output row: k = 8r..8r+7
  if (n >= N) return;
  const int j = n & 7;
  const int shift = 4 * (((j & 1) << 2) | (j >> 1));  // column j sits at nibble [0,4,1,5,2,6,3,7][j]
  uint32_t q = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint32_t w = in[(size_t)(8 * r + i) * (N >> 3) + (n >> 3)];
    q |= ((w >> shift) & 0xfu) << (4 * i);
  }
  out[(size_t)r * N + n] = shuffle_word(q);
}
__global__ void awq_repack_zeros_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                        int64_t words) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= words) return;
  uint32_t w = in[i], r = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int shift = 4 * (((j & 1) << 2) | (j >> 1));
    r |= ((w >> shift) & 0xfu) << (4 * j);
  }
  out[i] = r;
}

// W[k][n] = (q - (z + zero_offset)) * s, one thread per packed word (8 k x 1 col)
template <typename T>
__global__ void gptq_dequant_kernel(const uint32_t* __restrict__ qw, const uint32_t* __restrict__ qz,
                                    const uint16_t* __restrict__ sc, const int32_t* __restrict__ g_idx,
                                    uint16_t* __restrict__ out, int K, int N, int group_size,
                                    int shuffled, int zero_offset) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  int r = blockIdx.y;
  if (n >= N) return;
  uint32_t w = qw[(size_t)r * N + n];
  if (shuffled) w = unshuffle_word(w);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int k = 8 * r + i;
    int grp = g_idx ? g_idx[k] : k / group_size;
    int z = (int)((qz[(size_t)grp * (N >> 3) + (n >> 3)] >> ((n & 7) * 4)) & 0xf) + zero_offset;
    float s = T::to_f32(sc[(size_t)grp * N + n]);
    int q = (int)((w >> (4 * i)) & 0xf);
    // (q - z) is exact; one rounding to the 16-bit type, as __hmul does
    // (q_gemm.cu:1427-1431)
    out[(size_t)k * N + n] = T::from_f32((float)(q - z) * s);
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct Wna16Plan {
  int vec, mt, ksplit, ksteps_per_split, nseg;
  bool fast;
};

static Wna16Plan make_plan(int64_t M, int64_t N, int64_t K, int64_t gs, int64_t ztiles = 0) {
  Wna16Plan pl;
  pl.nseg = 0;
  pl.vec = (N % 64 == 0) ? 4 : (N % 32 == 0) ? 2 : 1;
  int fv = APHRO_LAB_ENV_INT("APHRO_WNA16_VEC", 0);
  if (fv == 1 || fv == 2 || fv == 4) {
    if (N % (16 * fv) == 0) pl.vec = fv;
  }
  // (64-column tiles even for narrow N: a 32-column tile doubles the A-fragment traffic per
  //  weight byte, which costs more than the extra workgroups gain -- measured 10.3 vs 12.8 us
  //  on the 14336x4096 down projection)
  pl.mt = (M > 16) ? 2 : 1;
  const int64_t tiles = N / (16 * pl.vec) * (ztiles > 0 ? ztiles : (M + 16 * pl.mt - 1) / (16 * pl.mt));
  const int total_segs = (int)((K + 127) / 128);
  const int64_t gq = gs >> 7;
  pl.fast = (K % 128 == 0 && gs % 128 == 0 && (gq & (gq - 1)) == 0 && pl.vec >= 2 &&
             (K / 8) * N * 4 < (int64_t)0xffffffff) && !APHRO_LAB_ENV_INT("APHRO_WNA16_GENERIC", 0);
  if (pl.fast) {
    // every wave owns exactly NSEG segments; ksplit = total_segs / (FNW * NSEG) fp32 slabs.
    const int fk = APHRO_LAB_ENV_INT("APHRO_WNA16_KSPLIT", 0);
    int best_ns = 0, best_split = 0;
    for (int split = 1; split <= 8; ++split) {
      if (total_segs % (split * FNW) != 0) continue;
      const int ns = total_segs / (split * FNW);
      if (!(ns == 1 || ns == 2 || ns == 4 || ns == 7 || ns == 8)) continue;
      if (fk > 0) { if (split == fk) { best_ns = ns; best_split = split; } continue; }
      if (best_ns == 0) { best_ns = ns; best_split = split; }
      // a deeper split only while the grid leaves CUs idle (< ~0.75 workgroups per CU)
      else if (tiles * best_split < 192 && tiles * split <= 1100) { best_ns = ns; best_split = split; }
    }
    if (best_ns) {
      pl.nseg = best_ns;
      pl.ksplit = best_split;
      pl.ksteps_per_split = FNW * best_ns * 4;
      return pl;
    }
    pl.fast = false;
  }
  int target = (int)((256 + tiles / 2) / tiles);
  target = target < 1 ? 1 : (target > 8 ? 8 : target);
  int fk = APHRO_LAB_ENV_INT("APHRO_WNA16_KSPLIT", 0);
  if (fk > 0) target = fk;
  while (target > 1 && total_segs / target < 1) --target;
  int segs = (total_segs + target - 1) / target;
  pl.ksteps_per_split = segs * 4;
  pl.ksplit = (total_segs + segs - 1) / segs;
  return pl;
}

template <typename T, int VEC, int MT>
static void launch_wna16(const Wna16Params& p_in, const Wna16Plan& pl, hipStream_t st) {
  const Wna16Params& p = p_in;
  dim3 grid((unsigned)(p.N / (16 * VEC)), (unsigned)pl.ksplit, (unsigned)((p.M + 16 * MT - 1) / (16 * MT)));
  if (pl.fast) {
    if constexpr (VEC >= 2) {
      Wna16Params p = p_in;
      const unsigned per = pl.ksplit > 1 && 8 % pl.ksplit == 0 ? 8u / (unsigned)pl.ksplit : 0u;
      const bool remap = per > 0 && grid.z == 1 && p.expert_ids == nullptr && grid.x % per == 0 && !APHRO_LAB_ENV_INT("APHRO_WNA16_NO_XCD_REMAP", 0);
      p.xcd_remap = remap ? (per == 4 ? 3 : per == 2 ? 2 : 1) : 0;
      p.xcd_tiles = remap ? (int)(grid.x / per) : 0;
      size_t lds = (size_t)FNW * MT * VEC * 64 * 4 * sizeof(float);
#define APHRO_FAST(NS) hipLaunchKernelGGL((wna16_gemm_kernel<T, VEC, MT, NS>), grid, dim3(FNW * 64), lds, st, p)
      switch (pl.nseg) {
        case 8: APHRO_FAST(8); break;
        case 7: APHRO_FAST(7); break;
        case 4: APHRO_FAST(4); break;
        case 2: APHRO_FAST(2); break;
        default: APHRO_FAST(1); break;
      }
#undef APHRO_FAST
    }
  } else {
    size_t lds = (size_t)NW * MT * VEC * 64 * 4 * sizeof(float);
    hipLaunchKernelGGL((wna16_gemm_generic_kernel<T, VEC, MT>), grid, dim3(NW * 64), lds, st, p);
  }
}

template <typename T>
static int run_wna16(Wna16Params p, const Wna16Plan& pl, hipStream_t st) {
  switch (pl.vec * 10 + pl.mt) {
    case 41: launch_wna16<T, 4, 1>(p, pl, st); break;
    case 42: launch_wna16<T, 4, 2>(p, pl, st); break;
    case 21: launch_wna16<T, 2, 1>(p, pl, st); break;
    case 22: launch_wna16<T, 2, 2>(p, pl, st); break;
    case 11: launch_wna16<T, 1, 1>(p, pl, st); break;
    case 12: launch_wna16<T, 1, 2>(p, pl, st); break;
    default: set_error("bad wna16 plan"); return APHRO_ERR_INVALID;
  }
  APHRO_LAUNCH_CHECK();
  if (pl.ksplit > 1 && p.c != nullptr) {
    int64_t mn = (int64_t)p.M * p.N;
    unsigned blocks = (unsigned)((mn / 4 + 255) / 256);
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, st, p.partial, p.c,
                       (const uint16_t*)nullptr, mn, p.N, pl.ksplit);
    APHRO_LAUNCH_CHECK();
  }
  return APHRO_OK;
}

}  // namespace aphro

using namespace aphro;

static size_t packed_a_bytes(int64_t M, int64_t K) {  // fragment-major f16 copy of A
  size_t b = (size_t)((M + 15) / 16) * 16 * (size_t)K * 2;
  return (b + 255) / 256 * 256;
}

extern "C" size_t aphro_wna16_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  int64_t m = M < APHRO_WNA16_MAX_M ? M : APHRO_WNA16_MAX_M;
  // [fragment-major copy of A][fp32 split-K partial slabs]
  return packed_a_bytes(m, K) + (size_t)8 * (size_t)m * (size_t)N * sizeof(float);
}

extern "C" int aphro_gptq_gemm(const void* a, const uint32_t* q_weight, const uint32_t* qzeros,
                               const void* scales, const int32_t* perm, void* a_perm_tmp, void* c,
                               void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K,
                               int64_t groups, int64_t lda, int zero_offset, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "gptq_gemm: dtype must be f16 or bf16");
  APHRO_CHECK(M >= 0 && M <= APHRO_WNA16_MAX_M, "gptq_gemm: M=%ld exceeds APHRO_WNA16_MAX_M", (long)M);
  APHRO_CHECK(groups > 0 && K % groups == 0, "gptq_gemm: K=%ld not divisible by groups=%ld", (long)K, (long)groups);
  const int64_t gs = K / groups;
  APHRO_CHECK(K % 32 == 0 && gs % 32 == 0, "gptq_gemm: K and group size must be multiples of 32 (K=%ld, g=%ld)", (long)K, (long)gs);
  APHRO_CHECK(N % 16 == 0, "gptq_gemm: N=%ld must be a multiple of 16", (long)N);
  APHRO_CHECK(lda % 8 == 0 && ((uintptr_t)a % 16) == 0, "gptq_gemm: a must be 16-byte aligned with lda %% 8 == 0");
  if (M == 0) return APHRO_OK;
  // one launch instead of pack + GEMM (+ reduce) where the resident kernel serves the shape (wna16_gemm_resident.hip)
  if (perm == nullptr && lda >= K && ((uintptr_t)c % 16) == 0 && aphro_wna16_gemm_rowmajor_supported(M, N, K, groups, dtype)) {
    const int rc = aphro_wna16_gemm_rowmajor(a, lda, q_weight, qzeros, scales, c, workspace, workspace_bytes, M, N, K,
                                             groups, zero_offset, dtype, 0, stream);
    if (rc != APHRO_ERR_WORKSPACE) return rc;
  }
  Wna16Plan pl = make_plan(M, N, K, gs);
  const size_t apk_bytes = pl.fast ? packed_a_bytes(M, K) : 0;
  {
    size_t need = apk_bytes + (pl.ksplit > 1 ? (size_t)pl.ksplit * M * N * sizeof(float) : 0);
    if (need > 0 && (workspace == nullptr || workspace_bytes < need)) {
      set_error("gptq_gemm: workspace %zu < %zu bytes", workspace_bytes, need);
      return APHRO_ERR_WORKSPACE;
    }
  }
  const uint16_t* ap = (const uint16_t*)a;
  int64_t ld = lda;
  Wna16Params p;
  p.apk = nullptr;
  if (pl.fast) {
    const int64_t total = (K / 128) * 4 * ((M + 15) / 16) * 64;
    dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == APHRO_F16)
      hipLaunchKernelGGL((pack_a_kernel<Half>), grid, dim3(256), 0, st, ap, perm, (uint16_t*)workspace, (int)M,
                         (int)K, (int)lda);
    else
      hipLaunchKernelGGL((pack_a_kernel<BFloat>), grid, dim3(256), 0, st, ap, perm, (uint16_t*)workspace, (int)M,
                         (int)K, (int)lda);
    APHRO_LAUNCH_CHECK();
    p.apk = (const uint16_t*)workspace;
  } else if (perm) {
    APHRO_CHECK(a_perm_tmp != nullptr, "gptq_gemm: act-order needs a_perm_tmp");
    hipLaunchKernelGGL(permute_cols_kernel, dim3((unsigned)((K + 255) / 256), (unsigned)M), dim3(256), 0, st,
                       ap, perm, (uint16_t*)a_perm_tmp, (int)M, (int)K, (int)lda);
    APHRO_LAUNCH_CHECK();
    ap = (const uint16_t*)a_perm_tmp;
    ld = K;
  }
  p.a = ap; p.qw = q_weight; p.qz = qzeros; p.sc = (const uint16_t*)scales;
  p.c = (uint16_t*)c; p.partial = (float*)((char*)workspace + apk_bytes);
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = (int)ld;
  p.group_size = (int)gs; p.ksteps_per_split = pl.ksteps_per_split; p.ksplit = pl.ksplit;
  p.zero_offset = zero_offset;
  p.gshift = 0;
  for (int64_t q = gs >> 7; q > 1; q >>= 1) ++p.gshift;
  p.force_partial = 0;
  p.act_packed = nullptr;
  p.expert_ids = nullptr; p.num_post_pad = nullptr; p.w_estride = p.z_estride = p.s_estride = 0;
  return dtype == APHRO_F16 ? run_wna16<Half>(p, pl, st) : run_wna16<BFloat>(p, pl, st);
}

// Fragment-major activation packing, exposed so that producers / callers can pack
// once and feed several GEMMs (qkv and gate_up share their input).
extern "C" size_t aphro_wna16_packed_a_bytes(int64_t M, int64_t K) { return packed_a_bytes(M, K); }

extern "C" int aphro_wna16_pack_a(const void* a, const int32_t* perm, void* packed, int64_t M, int64_t K,
                                  int64_t lda, int dtype, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "pack_a: dtype must be f16 or bf16");
  APHRO_CHECK(K % 128 == 0 && lda % 8 == 0 && ((uintptr_t)a % 16) == 0, "pack_a: K %% 128 and 16-byte alignment required");
  if (M == 0) return APHRO_OK;
  const int64_t total = (K / 128) * 4 * ((M + 15) / 16) * 64;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == APHRO_F16)
    hipLaunchKernelGGL((pack_a_kernel<Half>), grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a, perm,
                       (uint16_t*)packed, (int)M, (int)K, (int)lda);
  else
    hipLaunchKernelGGL((pack_a_kernel<BFloat>), grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a, perm,
                       (uint16_t*)packed, (int)M, (int)K, (int)lda);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

// Number of fp32 K-split slabs the fast kernel will produce for this shape (1 = none),
// or 0 if the shape is not served by the fast (packed-A) kernel.
extern "C" int aphro_wna16_ksplit(int64_t M, int64_t N, int64_t K, int64_t groups) {
  if (groups <= 0 || K % groups != 0 || M <= 0 || M > APHRO_WNA16_MAX_M || N % 16 != 0) return 0;
  Wna16Plan pl = make_plan(M, N, K, K / groups);
  return pl.fast ? pl.ksplit : 0;
}

// GEMM on pre-packed activations.  c != NULL: result [M,N] in `dtype` (a reduce kernel
// runs when ksplit > 1).  c == NULL: the fp32 slabs [ksplit][M][N] are left in `partials`
// for a fused consumer (aphro_fused_add_rms_norm_pack / aphro_rope_cache) to sum.
extern "C" int aphro_wna16_gemm_packed(const void* a_packed, const uint32_t* q_weight, const uint32_t* qzeros,
                                       const void* scales, void* c, float* partials, size_t partial_bytes,
                                       int64_t M, int64_t N, int64_t K, int64_t groups, int zero_offset,
                                       int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "wna16_gemm_packed: dtype must be f16 or bf16");
  APHRO_CHECK(M > 0 && M <= APHRO_WNA16_MAX_M, "wna16_gemm_packed: M=%ld out of range", (long)M);
  APHRO_CHECK(groups > 0 && K % groups == 0, "wna16_gemm_packed: bad groups");
  Wna16Plan pl = make_plan(M, N, K, K / groups);
  APHRO_CHECK(pl.fast, "wna16_gemm_packed: shape K=%ld N=%ld g=%ld is not served by the fast kernel", (long)K,
              (long)N, (long)(K / groups));
  if (pl.ksplit > 1 || c == nullptr) {
    size_t need = (size_t)pl.ksplit * M * N * sizeof(float);
    if (partials == nullptr || partial_bytes < need) {
      set_error("wna16_gemm_packed: partial buffer %zu < %zu bytes", partial_bytes, need);
      return APHRO_ERR_WORKSPACE;
    }
  }
  Wna16Params p;
  p.a = nullptr; p.apk = (const uint16_t*)a_packed; p.qw = q_weight; p.qz = qzeros;
  p.sc = (const uint16_t*)scales; p.c = (uint16_t*)c; p.partial = partials;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = 0;
  p.group_size = (int)(K / groups); p.ksteps_per_split = pl.ksteps_per_split;
  p.ksplit = (c == nullptr && pl.ksplit == 1) ? 1 : pl.ksplit;
  p.zero_offset = zero_offset;
  p.gshift = 0;
  for (int64_t q = (K / groups) >> 7; q > 1; q >>= 1) ++p.gshift;
  p.force_partial = (c == nullptr) ? 1 : 0;
  p.act_packed = nullptr;
  p.expert_ids = nullptr; p.num_post_pad = nullptr; p.w_estride = p.z_estride = p.s_estride = 0;
  return dtype == APHRO_F16 ? run_wna16<Half>(p, pl, st) : run_wna16<BFloat>(p, pl, st);
}

// gate_up GEMM with the SiLU-and-mul + activation pack fused into its epilogue.  The weight
// columns must be INTERLEAVED (column 2j = gate_j, 2j+1 = up_j; qzeros / scales likewise) --
// a load-time permutation of the merged [gate | up] tensor.  Output: fragment-major f16
// [M, N/2] ready for aphro_wna16_gemm_packed.  Only shapes whose plan keeps the whole K inside
// one workgroup (aphro_wna16_ksplit == 1) are served.
extern "C" int aphro_wna16_gemm_silu_pack(const void* a_packed, const uint32_t* q_weight, const uint32_t* qzeros,
                                          const void* scales, void* act_packed, int64_t M, int64_t N, int64_t K,
                                          int64_t groups, int zero_offset, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "wna16_gemm_silu_pack: dtype must be f16 or bf16");
  APHRO_CHECK(M > 0 && M <= APHRO_WNA16_MAX_M, "wna16_gemm_silu_pack: M=%ld out of range", (long)M);
  APHRO_CHECK(groups > 0 && K % groups == 0, "wna16_gemm_silu_pack: bad groups");
  APHRO_CHECK(N % 256 == 0, "wna16_gemm_silu_pack: N/2 must be a multiple of 128 (packed K of the next GEMM)");
  Wna16Plan pl = make_plan(M, N, K, K / groups);
  APHRO_CHECK(pl.fast && pl.ksplit == 1 && pl.vec >= 2,
              "wna16_gemm_silu_pack: shape K=%ld N=%ld is split across workgroups; use the unfused ops", (long)K,
              (long)N);
  Wna16Params p;
  p.a = nullptr; p.apk = (const uint16_t*)a_packed; p.qw = q_weight; p.qz = qzeros;
  p.sc = (const uint16_t*)scales; p.c = nullptr; p.partial = nullptr;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = 0;
  p.group_size = (int)(K / groups); p.ksteps_per_split = pl.ksteps_per_split;
  p.ksplit = 1;
  p.zero_offset = zero_offset;
  p.gshift = 0;
  for (int64_t q = (K / groups) >> 7; q > 1; q >>= 1) ++p.gshift;
  p.force_partial = 0;
  p.act_packed = (uint16_t*)act_packed;
  p.expert_ids = nullptr; p.num_post_pad = nullptr; p.w_estride = p.z_estride = p.s_estride = 0;
  return dtype == APHRO_F16 ? run_wna16<Half>(p, pl, st) : run_wna16<BFloat>(p, pl, st);
}

// Expert GEMMs of a mixture-of-experts layer in ONE launch (the marlin_gemm_moe role,
// kernels/moe/marlin_moe_ops.cu, re-designed): activations arrive packed and sorted by expert
// (aphro_moe_gather_pack over moe_align_block_size's order, block_size 16), so every 16-row m-tile
// belongs to one expert and the dense kernel only needs that expert's weight base.  Weights:
// [E][K/8][N] K-packed (exllama order per expert), qzeros [E][G][N/8], scales [E][G][N].
// Exactly one output: act_packed (w1|w3 GEMM with interleaved gate/up columns: SiluAndMul + pack
// epilogue, needs aphro_wna16_grouped_ksplit == 1), c (T [m_pad, N]) or partials (fp32 slabs).
extern "C" int aphro_wna16_grouped_ksplit(int64_t m_pad, int64_t N, int64_t K, int64_t groups) {
  if (groups <= 0 || K % groups != 0 || m_pad <= 0 || m_pad % 16 != 0) return -1;
  Wna16Plan pl = make_plan(16, N, K, K / groups, m_pad / 16);
  return pl.fast ? pl.ksplit : -1;
}

extern "C" int aphro_wna16_gemm_grouped(const void* a_packed, const uint32_t* q_weight, const uint32_t* qzeros,
                                        const void* scales, const int32_t* expert_ids,
                                        const int32_t* num_tokens_post_pad, void* c, float* partials,
                                        size_t partial_bytes, void* act_packed, int64_t m_pad, int64_t N,
                                        int64_t K, int64_t groups, int zero_offset, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "wna16_gemm_grouped: dtype must be f16 or bf16");
  APHRO_CHECK(m_pad > 0 && m_pad % 16 == 0, "wna16_gemm_grouped: m_pad=%ld must be a positive multiple of 16", (long)m_pad);
  APHRO_CHECK(groups > 0 && K % groups == 0 && expert_ids && num_tokens_post_pad, "wna16_gemm_grouped: bad arguments");
  Wna16Plan pl = make_plan(16, N, K, K / groups, m_pad / 16);
  APHRO_CHECK(pl.fast, "wna16_gemm_grouped: shape K=%ld N=%ld g=%ld is not served by the fast kernel", (long)K,
              (long)N, (long)(K / groups));
  pl.mt = 1;
  APHRO_CHECK(act_packed == nullptr || (pl.ksplit == 1 && N % 256 == 0),
              "wna16_gemm_grouped: the SiluAndMul epilogue needs ksplit == 1 and N %% 256 == 0");
  if (act_packed == nullptr && (pl.ksplit > 1 || c == nullptr)) {
    size_t need = (size_t)pl.ksplit * m_pad * N * sizeof(float);
    if (partials == nullptr || partial_bytes < need) {
      set_error("wna16_gemm_grouped: partial buffer %zu < %zu bytes", partial_bytes, need);
      return APHRO_ERR_WORKSPACE;
    }
  }
  Wna16Params p;
  p.a = nullptr; p.apk = (const uint16_t*)a_packed; p.qw = q_weight; p.qz = qzeros;
  p.sc = (const uint16_t*)scales; p.c = (uint16_t*)c; p.partial = partials;
  p.M = (int)m_pad; p.N = (int)N; p.K = (int)K; p.lda = 0;
  p.group_size = (int)(K / groups); p.ksteps_per_split = pl.ksteps_per_split;
  p.ksplit = pl.ksplit;
  p.zero_offset = zero_offset;
  p.gshift = 0;
  for (int64_t q = (K / groups) >> 7; q > 1; q >>= 1) ++p.gshift;
  p.force_partial = (c == nullptr && act_packed == nullptr) ? 1 : 0;
  p.act_packed = (uint16_t*)act_packed;
  p.expert_ids = expert_ids; p.num_post_pad = num_tokens_post_pad;
  p.w_estride = (K / 8) * N; p.z_estride = groups * (N / 8); p.s_estride = groups * N;
  return dtype == APHRO_F16 ? run_wna16<Half>(p, pl, st) : run_wna16<BFloat>(p, pl, st);
}

extern "C" int aphro_gptq_repack(const uint32_t* q_weight, const int32_t* q_perm, uint32_t* out,
                                 int64_t size_k, int64_t size_n, int bit, void* stream) {
  APHRO_CHECK(bit == 4, "gptq_shuffle/repack: only 4-bit is implemented (bit=%d)", bit);
  APHRO_CHECK(size_k % 8 == 0, "gptq_shuffle/repack: K must be a multiple of 8");
  APHRO_CHECK(q_weight != out || q_perm == nullptr, "gptq_repack: in-place needs no perm");
  if (size_k == 0 || size_n == 0) return APHRO_OK;
  dim3 grid((unsigned)((size_n + 255) / 256), (unsigned)(size_k / 8));
  hipLaunchKernelGGL(gptq_repack_kernel, grid, dim3(256), 0, (hipStream_t)stream, q_weight, q_perm, out,
                     (int)(size_k / 8), (int)size_n);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_gptq_shuffle(uint32_t* q_weight, const int32_t* q_perm, int64_t size_k,
                                  int64_t size_n, int bit, uint32_t* tmp, void* stream) {
  if (q_perm == nullptr) return aphro_gptq_repack(q_weight, nullptr, q_weight, size_k, size_n, bit, stream);
  APHRO_CHECK(tmp != nullptr, "gptq_shuffle: act-order needs a tmp buffer of K/8*N words");
  int rc = aphro_gptq_repack(q_weight, q_perm, tmp, size_k, size_n, bit, stream);
  if (rc != APHRO_OK) return rc;
  hipError_t e = hipMemcpyAsync(q_weight, tmp, (size_t)(size_k / 8) * size_n * 4, hipMemcpyDeviceToDevice,
                                (hipStream_t)stream);
  if (e != hipSuccess) {
    set_error("gptq_shuffle: copy back failed: %s", hipGetErrorString(e));
    return APHRO_ERR_LAUNCH;
  }
  return APHRO_OK;
}

extern "C" int aphro_gptq_dequant(const uint32_t* q_weight, const uint32_t* qzeros, const void* scales,
                                  const int32_t* g_idx, void* out, int64_t K, int64_t N, int64_t groups,
                                  int shuffled, int zero_offset, int dtype, void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "gptq_dequant: dtype must be f16 or bf16");
  APHRO_CHECK(K % 8 == 0 && N % 8 == 0 && groups > 0, "gptq_dequant: bad shape");
  if (K == 0 || N == 0) return APHRO_OK;
  dim3 grid((unsigned)((N + 255) / 256), (unsigned)(K / 8));
  int gs = (int)(K / groups);
  if (dtype == APHRO_F16)
    hipLaunchKernelGGL((gptq_dequant_kernel<Half>), grid, dim3(256), 0, (hipStream_t)stream, q_weight, qzeros,
                       (const uint16_t*)scales, g_idx, (uint16_t*)out, (int)K, (int)N, gs, shuffled, zero_offset);
  else
    hipLaunchKernelGGL((gptq_dequant_kernel<BFloat>), grid, dim3(256), 0, (hipStream_t)stream, q_weight, qzeros,
                       (const uint16_t*)scales, g_idx, (uint16_t*)out, (int)K, (int)N, gs, shuffled, zero_offset);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_awq_repack(const uint32_t* qweight, uint32_t* out, int64_t K, int64_t N, void* stream) {
  APHRO_CHECK(K % 8 == 0 && N % 8 == 0, "awq_repack: K and N must be multiples of 8");
  if (K == 0 || N == 0) return APHRO_OK;
  dim3 grid((unsigned)((N + 255) / 256), (unsigned)(K / 8));
  hipLaunchKernelGGL(awq_repack_kernel, grid, dim3(256), 0, (hipStream_t)stream, qweight, out, (int)K, (int)N);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_awq_repack_zeros(const uint32_t* qzeros, uint32_t* out, int64_t groups, int64_t N,
                                      void* stream) {
  APHRO_CHECK(N % 8 == 0, "awq_repack_zeros: N must be a multiple of 8");
  int64_t words = groups * (N / 8);
  if (words == 0) return APHRO_OK;
  hipLaunchKernelGGL(awq_repack_zeros_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, qzeros, out, words);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
