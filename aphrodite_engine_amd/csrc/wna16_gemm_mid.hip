// W4A16 (GPTQ / AWQ-repacked int4) GEMM for DECODE batches of 33..64 rows on gfx950 -- between the HBM-bound decode
// kernel (wna16_gemm.hip: 32 rows per pass, a second pass re-reads and re-unpacks every weight) and the prefill tile
// machine (wna16_gemm_large.hip: 128-row tiles, one barrier + 4..6 direct-to-LDS loads per 64-k tile, latency bound at
// this size).  Same role as the reference's exllama kernel below 50 rows / the reconstruct + library GEMM above it
// (kernels/quantization/gptq/q_gemm.cu:1529-1544) and as gptq_marlin_gemm at small M (gptq_marlin.cu:2247).
//
//   * D^T[n][m] = W^T[n][k] . A^T[k][m] with v_mfma_f32_32x32x16_f16: the exllama dword (8 consecutive k of ONE column)
//     is one lane's weight fragment, so the weights go global -> VGPR -> MFMA with no LDS and no barrier in the K loop.
//     A 16-byte load gives a lane the dwords of 4 adjacent columns = 4 n-blocks whose column index is n0 + 4 i + nb
//     (i = MFMA row): one 1 KiB wave load feeds 4 MB MFMAs.
//   * activations arrive FRAGMENT-MAJOR (mid_pack_a_kernel): block (k/16, m/32) = 1 KiB, lane (m % 32, (k % 16) / 8)
//     holds A[m][8 k8 .. + 7]: every activation fragment is one lane-linear 1 KiB load that hits in L2.
//   * weights run DW steps ahead of their use in a register ring, activations DA steps; loads past a wave's K range are
//     issued anyway (buffer loads: in range of the next slice or zero) so the loop is straight-line code.
//   * the 4 (or 8) waves of a workgroup split K; their accumulators meet in a butterfly through LDS (64 / 128 KiB) after
//     which wave w owns accumulator quad q = w; the 64 x 128 tile is put together in LDS and leaves as whole rows.  More K slices
//     (grid.y) write fp32 slabs summed by mid_splitk_reduce_kernel in fixed order -- no atomics, deterministic.
//   * int4 -> f16 in registers with the reference's numerics (q_gemm.cu:1394-1434): (q - z) exact through the 1024 + q
//     trick, one rounding in the multiply by the group scale.
#include "common.h"
#include "wna16_strip.h"

#ifndef MID_ABL      // timing experiments only (tools/mid_ablate.sh): 1 no A loads, 2 no W loads, 4 no MFMA, 8 no dequant,
                     // 16 no K reduction / store (one dword per lane instead), 32 no K loop at all, 64 no prologue loads
#define MID_ABL 0
#endif

namespace aphro {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Wna16MidParams {
  const uint32_t* apk;    // packed activations [K/16][mbt][64 lanes][4 dwords]
  const uint32_t* qw;     // [K/8, N] exllama order
  const uint32_t* qz;     // [G, N/8]
  const uint16_t* sc;     // [G, N]
  uint16_t* c;            // [M, N]
  float* partial;         // [ksplit][M][N] when ksplit > 1
  int M, N, K;
  int group_size;         // multiple of 128
  int zero_offset;
  int out_bf16, scale_bf16;
  int ksplit;             // grid.y
  // ---- decode fast path (template ADEC): activations in the DECODE kernel's fragment-major format (wna16_gemm.hip:
  // block (k/128, (k%32)/8, m/16) = 1 KiB, lane ((k%128)/32, m%16)), written by the fused producers --------------------
  int mtiles;             // 16-row m-tiles of that buffer = ceil(M / 16)
  int force_partial;      // write the fp32 slab(s) even with one K slice (the consumer kernel sums them)
  uint16_t* act_packed;   // != NULL: SiluAndMul + pack epilogue (interleaved gate / up columns) into the same format
  Wna16StripGeom strip;   // .on: qw is the strip-major copy of the <= 32-row kernels, the only resident one (wna16_strip.h)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mid_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

__device__ __forceinline__ uint32_t mid_and_or(uint32_t a, uint32_t mask, uint32_t orv) {
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask), "v"(orv));
  return r;
}

// one exllama dword -> 8 scaled f16: ((1024 + q) - (1024 + z)) * s  (exact difference, one rounding)
__device__ __forceinline__ f16x8 mid_dq8(uint32_t w, f16x2 zh, f16x2 zh16, f16x2 sc) {
  const f16x2 inv16 = {(f16)0.0625f, (f16)0.0625f};
  const uint32_t magic = 0x64006400u;
  const uint32_t q0 = mid_and_or(w, 0x000f000fu, magic);
  const uint32_t q1 = mid_and_or(w, 0x00f000f0u, magic);
  const uint32_t w8 = w >> 8;
  const uint32_t q2 = mid_and_or(w8, 0x000f000fu, magic);
  const uint32_t q3 = mid_and_or(w8, 0x00f000f0u, magic);
  const f16x2 d0 = (__builtin_bit_cast(f16x2, q0) - zh) * sc;
  const f16x2 d1 = (__builtin_bit_cast(f16x2, q1) * inv16 + zh16) * sc;
  const f16x2 d2 = (__builtin_bit_cast(f16x2, q2) - zh) * sc;
  const f16x2 d3 = (__builtin_bit_cast(f16x2, q3) * inv16 + zh16) * sc;
  u32x4 r = {__builtin_bit_cast(uint32_t, d0), __builtin_bit_cast(uint32_t, d1), __builtin_bit_cast(uint32_t, d2),
             __builtin_bit_cast(uint32_t, d3)};
  return __builtin_bit_cast(f16x8, r);
}

// ---- butterfly pieces: static register indices only (a runtime quad index would put the accumulators in scratch) -----
// send / receive quads [Q0, Q0 + NQ) of n-blocks [NB0, NB0 + NNB)
template <int MB, int NQ, int Q0, int NB0 = 0, int NNB = 4>
__device__ __forceinline__ void mid_send(const f32x16 (&acc)[4][MB], float* red, int wave, int lane) {
#pragma unroll
  for (int nbi = 0; nbi < NNB; ++nbi)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        const int q = Q0 + qi, nb = NB0 + nbi;
        const f32x4 v = {acc[nb][mb][4 * q], acc[nb][mb][4 * q + 1], acc[nb][mb][4 * q + 2], acc[nb][mb][4 * q + 3]};
        *reinterpret_cast<f32x4*>(&red[(((wave * (NNB * MB * NQ)) + (nbi * MB + mb) * NQ + qi) * 64 + lane) * 4]) = v;
      }
}
template <int MB, int NQ, int Q0, int NB0 = 0, int NNB = 4>
__device__ __forceinline__ void mid_recv(f32x16 (&acc)[4][MB], const float* red, int partner, int lane) {
#pragma unroll
  for (int nbi = 0; nbi < NNB; ++nbi)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        const int q = Q0 + qi, nb = NB0 + nbi;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&red[(((partner * (NNB * MB * NQ)) + (nbi * MB + mb) * NQ + qi) * 64 + lane) * 4]);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[nb][mb][4 * q + r] += v[r];
      }
}
// wave owns quad Q (of n-blocks NB0 .. NB0 + NNB - 1): row m = mb * 32 + l31, columns 32 Q + 16 kh + (4 r + nb) of the
// workgroup's 128.  Stored straight from the registers every instruction would scatter 16-byte pieces over 32 rows
// (measured: 10 us of a 28 us kernel); the tile goes through LDS ([32 MB][128 + 4] fp32, conflict-free both ways) and
// leaves as whole 512-byte rows.
constexpr int MID_LDT = 132;
template <int MB, int Q, int NB0 = 0, int NNB = 4>
__device__ __forceinline__ void mid_put(const f32x16 (&acc)[4][MB], float* tile, int l31, int kh) {
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* dst = &tile[(mb * 32 + l31) * MID_LDT + 32 * Q + 16 * kh + 4 * r + NB0];
      if constexpr (NNB == 4)
        *reinterpret_cast<f32x4*>(dst) = f32x4{acc[0][mb][4 * Q + r], acc[1][mb][4 * Q + r], acc[2][mb][4 * Q + r], acc[3][mb][4 * Q + r]};
      else
        *reinterpret_cast<f32x2*>(dst) = f32x2{acc[NB0][mb][4 * Q + r], acc[NB0 + 1][mb][4 * Q + r]};
    }
}
template <int MB, int NWK>
__device__ __forceinline__ void mid_flush(const float* tile, const Wna16MidParams& p, int n0, int wave, int lane) {
  constexpr int ROWS = 32 * MB / NWK;           // rows per wave
  if (p.act_packed != nullptr) {
    // SiluAndMul + activation pack (the epilogue of wna16_gemm.hip's gate_up form, same roundings: the GEMM result is
    // rounded to T first, silu_mul_bits is shared with the separate op).  Columns 2j / 2j+1 = gate_j / up_j: a lane's 8
    // columns are output features j0 .. j0 + 3 = one 8-byte piece of the consumer's fragment.
#pragma unroll
    for (int i = 0; i < ROWS / 4; ++i) {
      const int m = wave * ROWS + 4 * i + (lane >> 4);
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(&tile[m * MID_LDT + 8 * (lane & 15)]);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(&tile[m * MID_LDT + 8 * (lane & 15) + 4]);
      const float g4[4] = {v0[0], v0[2], v1[0], v1[2]}, u4[4] = {v0[1], v0[3], v1[1], v1[3]};
      uint16_t o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (p.out_bf16) {
          o[q] = bf16_bits_to_f16_bits_sat(silu_mul_bits<BFloat>(BFloat::to_f32(BFloat::from_f32(g4[q])), BFloat::to_f32(BFloat::from_f32(u4[q]))));
        } else {
          o[q] = silu_mul_bits<Half>(Half::to_f32(Half::from_f32(g4[q])), Half::to_f32(Half::from_f32(u4[q])));
        }
      }
      const int j0 = (n0 >> 1) + 4 * (lane & 15);
      uint16_t* dst = p.act_packed + ((((size_t)(j0 >> 7) * 4 + ((j0 & 31) >> 3)) * p.mtiles + (m >> 4)) * 64 + ((j0 & 127) >> 5) * 16 + (m & 15)) * 8 + (j0 & 7);
      if (m < p.M) *reinterpret_cast<u32x2*>(dst) = u32x2{(uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16)};
    }
  } else if (p.ksplit > 1 || p.force_partial) {   // fp32 slab: 2 rows x 512 B per instruction
#pragma unroll
    for (int i = 0; i < ROWS / 2; ++i) {
      const int m = wave * ROWS + 2 * i + (lane >> 5);
      const f32x4 v = *reinterpret_cast<const f32x4*>(&tile[m * MID_LDT + 4 * (lane & 31)]);
      if (m < p.M) *reinterpret_cast<f32x4*>(p.partial + ((size_t)blockIdx.y * p.M + m) * p.N + n0 + 4 * (lane & 31)) = v;
    }
  } else {                                      // f16 / bf16 rows: 4 rows x 256 B per instruction
#pragma unroll
    for (int i = 0; i < ROWS / 4; ++i) {
      const int m = wave * ROWS + 4 * i + (lane >> 4);
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(&tile[m * MID_LDT + 8 * (lane & 15)]);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(&tile[m * MID_LDT + 8 * (lane & 15) + 4]);
      const u32x4 o = p.out_bf16 ? u32x4{pack2_16<true>(v0[0], v0[1]), pack2_16<true>(v0[2], v0[3]), pack2_16<true>(v1[0], v1[1]), pack2_16<true>(v1[2], v1[3])}
                                 : u32x4{pack2_16<false>(v0[0], v0[1]), pack2_16<false>(v0[2], v0[3]), pack2_16<false>(v1[0], v1[1]), pack2_16<false>(v1[2], v1[3])};
      if (m < p.M) *reinterpret_cast<u32x4*>(p.c + (size_t)m * p.N + n0 + 8 * (lane & 15)) = o;
    }
  }
}

// NWK waves split K: 4 (two workgroups per CU, more K slices go to slabs) or 8 (one workgroup per CU, 128 KiB butterfly)
// STRIP: p.qw is the strip-major copy (p.strip: wna16_strip.h) -- a template parameter, not a run-time test: a branch per weight
// load cuts the software-pipelined K loop into basic blocks the scheduler cannot interleave
template <int MB, int NWK, bool ADEC = false, bool STRIP = false>
__global__ __launch_bounds__(64 * NWK, 2) void wna16_gemm_mid_kernel(Wna16MidParams p) {
  constexpr int NB = 4;
  constexpr int DW = 8;   // weight steps in flight (one 16-byte load each)
  constexpr int DA = 4;   // activation steps in flight (MB loads each)
  extern __shared__ __attribute__((aligned(16))) float red[];   // round 1: 32 MB KiB; round 2: 16 MB KiB + the output tile behind it
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kh = lane >> 5, l31 = lane & 31;
  const int n0 = blockIdx.x * (32 * NB);
  const int steps = p.K / 16 / (NWK * p.ksplit);        // 16-k steps of this wave; host: a multiple of group_size / 16
  const int s0 = (blockIdx.y * NWK + wave) * steps;
  const int gsteps = p.group_size >> 4;

  const __amdgpu_buffer_rsrc_t rw = mid_rsrc(p.qw, (uint32_t)((size_t)(p.K >> 3) * p.N * 4));
  const __amdgpu_buffer_rsrc_t ra = mid_rsrc(p.apk, ADEC ? (uint32_t)((size_t)(p.K >> 7) * 4 * p.mtiles * 1024) : (uint32_t)((size_t)(p.K >> 4) * MB * 1024));
  const int ngroups = p.K / p.group_size;
  const __amdgpu_buffer_rsrc_t rs = mid_rsrc(p.sc, (uint32_t)((size_t)ngroups * p.N * 2));
  const __amdgpu_buffer_rsrc_t rz = mid_rsrc(p.qz, (uint32_t)((size_t)ngroups * (p.N >> 3) * 4));
  int voff_w = (kh * p.N + n0 + 4 * l31) * 4;           // packed row 2 s + kh, this lane's 4 columns
  const int wstep = 2 * p.N * 4;
  uint32_t st_mult4 = 0;                                // strip-major weights: the lane's column part + the u bit of its row (kh)
  if constexpr (STRIP) {
    uint32_t mult;
    const uint32_t cb = wna16_strip_col(p.strip, n0 + 4 * l31, mult);
    voff_w = (int)((cb + mult * 64u * (uint32_t)kh) * 4u);
    st_mult4 = mult * 4u;
  }
  // ADEC: chunk (m, k8 = 2 s + kh) sits in block ((s / 8) * 4 + 2 (s % 2) + kh, m / 16), lane ((s % 8) / 2, m % 16)
  const int voff_a = ADEC ? (kh * p.mtiles + (l31 >> 4)) * 1024 + (l31 & 15) * 16 : lane * 16;
  constexpr int astep = MB * 1024;
  const int voff_s = (n0 + 4 * l31) * 2;
  const int voff_z = ((n0 + 4 * l31) >> 3) * 4;
  const int zshift = (l31 & 1) * 16;

  f32x16 acc[NB][MB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][mb][r] = 0.f;

  u32x4 wr[DW];
  u32x4 ar[DA][MB];
  u32x2 sraw;
  uint32_t zraw;
  auto load_meta = [&](int s) {
    const int grp = min(s / gsteps, ngroups - 1);
    sraw = __builtin_amdgcn_raw_buffer_load_b64(rs, voff_s, grp * p.N * 2, 0);
    zraw = __builtin_amdgcn_raw_buffer_load_b32(rz, voff_z, grp * (p.N >> 3) * 4, 0);
  };
  auto load_w = [&](int s) {
    if constexpr (STRIP) { // packed row 2 s (+ kh in the lane part): chunk and row term are wave-uniform, one v_mad per load
      uint32_t chunk, R;
      wna16_strip_row(p.strip, 2u * (uint32_t)s, chunk, R);
      return __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(st_mult4 * R) + voff_w, (int)(chunk * 4u), 2);
    }
    return __builtin_amdgcn_raw_buffer_load_b128(rw, voff_w, s * wstep, 2);
  };
  auto load_a = [&](int s, int mb) {
    if constexpr (ADEC)
      return __builtin_amdgcn_raw_buffer_load_b128(ra, voff_a + mb * 2048, (((s >> 3) * 4 + 2 * (s & 1)) * p.mtiles) * 1024 + ((s & 7) >> 1) * 256, 0);
    else
      return __builtin_amdgcn_raw_buffer_load_b128(ra, voff_a + mb * 1024, s * astep, 0);
  };

  if constexpr (MID_ABL & 64) {
    sraw = u32x2{0x3c003c00u, 0x3c003c00u}; zraw = 0x88888888u;
#pragma unroll
    for (int j = 0; j < DA; ++j)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) ar[j][mb] = u32x4{(uint32_t)lane, 1u, 2u, 3u};
#pragma unroll
    for (int j = 0; j < DW; ++j) wr[j] = u32x4{(uint32_t)lane, 5u, 6u, 7u};
  } else {
  load_meta(s0);
#pragma unroll
  for (int j = 0; j < DA; ++j)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) ar[j][mb] = load_a(s0 + j, mb);
#pragma unroll
  for (int j = 0; j < DW; ++j) wr[j] = load_w(s0 + j);
  }
  __builtin_amdgcn_sched_barrier(0);

  f16x2 zh[NB], zh16[NB], scv[NB];
  for (int sg = 0; sg < ((MID_ABL & 32) ? 0 : steps); sg += 8) {
    // group metadata of this block of 8 steps (a block never straddles a group: group_size % 128 == 0), loaded one block
    // ahead; converted unconditionally -- a runtime "new group?" branch around loads costs hipcc's vmcnt(0) fallback
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const uint16_t sb = (uint16_t)(sraw[nb >> 1] >> (16 * (nb & 1)));
      const float sf = p.scale_bf16 ? bf16_bits_to_f32(sb) : f16_bits_to_f32(sb);
      const int z = (int)((zraw >> (zshift + 4 * nb)) & 0xf) + p.zero_offset;
      const f16 s16 = (f16)sf;
      const f16 a16 = __builtin_bit_cast(f16, (uint16_t)(0x6400 | z));   // 1024 + z
      const f16 b16 = (f16)(float)(-64 - z);
      scv[nb] = f16x2{s16, s16};
      zh[nb] = f16x2{a16, a16};
      zh16[nb] = f16x2{b16, b16};
    }
    if constexpr (!(MID_ABL & 64)) load_meta(s0 + sg + 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int s = s0 + sg + j;
      f16x8 wf[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if constexpr (MID_ABL & 8) { const uint32_t w1 = wr[j % DW][nb]; wf[nb] = __builtin_bit_cast(f16x8, u32x4{w1, w1, w1, w1}); }
        else wf[nb] = mid_dq8(wr[j % DW][nb], zh[nb], zh16[nb], scv[nb]);
      }
      if constexpr (!(MID_ABL & 2)) wr[j % DW] = load_w(s + DW);
      f16x8 af[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) af[mb] = __builtin_bit_cast(f16x8, ar[j % DA][mb]);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) { if constexpr (!(MID_ABL & 1)) ar[j % DA][mb] = load_a(s + DA, mb); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          if constexpr (MID_ABL & 4) asm volatile("" :: "v"(wf[nb]), "v"(af[mb]));
          else acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[nb], af[mb], acc[nb][mb], 0, 0, 0);
        }
    }
  }

  if constexpr (MID_ABL & 16) {
    float t = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[nb][mb][r];
    t += __builtin_bit_cast(float, wr[0][0] ^ wr[1][1] ^ wr[2][2] ^ wr[3][3] ^ wr[4][0] ^ wr[5][1] ^ wr[6][2] ^ wr[7][3] ^ ar[0][0][0] ^ ar[1][0][1] ^ ar[2][0][2] ^ ar[3][0][3]);
    p.partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (64 * NWK) + threadIdx.x] = t;
    return;
  }
  // ---- K reduction: butterfly over the wave index.  Quad rounds on bits SH+1 and SH (two quads each way, then one);
  // with 8 waves a last round on bit 0 splits the n-blocks.  Fixed pairing -> fixed summation order. ---------------------
  constexpr int SH = NWK == 8 ? 1 : 0;
  const int wq = wave >> SH;                     // the quad this wave ends up owning
  if (wq & 2) mid_send<MB, 2, 0>(acc, red, wave, lane); else mid_send<MB, 2, 2>(acc, red, wave, lane);
  __syncthreads();
  if (wq & 2) mid_recv<MB, 2, 2>(acc, red, wave ^ (2 << SH), lane); else mid_recv<MB, 2, 0>(acc, red, wave ^ (2 << SH), lane);
  __syncthreads();
  switch (wq) {
    case 0: mid_send<MB, 1, 1>(acc, red, wave, lane); break;
    case 1: mid_send<MB, 1, 0>(acc, red, wave, lane); break;
    case 2: mid_send<MB, 1, 3>(acc, red, wave, lane); break;
    default: mid_send<MB, 1, 2>(acc, red, wave, lane); break;
  }
  __syncthreads();
  float* tile = red + 4096 * MB;                 // behind the last round's buffers (16 MB KiB): no barrier between recv and put
  if constexpr (NWK == 4) {
    switch (wq) {
      case 0: mid_recv<MB, 1, 0>(acc, red, wave ^ 1, lane); mid_put<MB, 0>(acc, tile, l31, kh); break;
      case 1: mid_recv<MB, 1, 1>(acc, red, wave ^ 1, lane); mid_put<MB, 1>(acc, tile, l31, kh); break;
      case 2: mid_recv<MB, 1, 2>(acc, red, wave ^ 1, lane); mid_put<MB, 2>(acc, tile, l31, kh); break;
      default: mid_recv<MB, 1, 3>(acc, red, wave ^ 1, lane); mid_put<MB, 3>(acc, tile, l31, kh); break;
    }
  } else {
    switch (wq) {
      case 0: mid_recv<MB, 1, 0>(acc, red, wave ^ 2, lane); break;
      case 1: mid_recv<MB, 1, 1>(acc, red, wave ^ 2, lane); break;
      case 2: mid_recv<MB, 1, 2>(acc, red, wave ^ 2, lane); break;
      default: mid_recv<MB, 1, 3>(acc, red, wave ^ 2, lane); break;
    }
    __syncthreads();
    switch (wave) {       // quad wq: the even wave keeps n-blocks 0, 1, the odd one 2, 3
      case 0: mid_send<MB, 1, 0, 2, 2>(acc, red, wave, lane); break;
      case 1: mid_send<MB, 1, 0, 0, 2>(acc, red, wave, lane); break;
      case 2: mid_send<MB, 1, 1, 2, 2>(acc, red, wave, lane); break;
      case 3: mid_send<MB, 1, 1, 0, 2>(acc, red, wave, lane); break;
      case 4: mid_send<MB, 1, 2, 2, 2>(acc, red, wave, lane); break;
      case 5: mid_send<MB, 1, 2, 0, 2>(acc, red, wave, lane); break;
      case 6: mid_send<MB, 1, 3, 2, 2>(acc, red, wave, lane); break;
      default: mid_send<MB, 1, 3, 0, 2>(acc, red, wave, lane); break;
    }
    __syncthreads();
    switch (wave) {
      case 0: mid_recv<MB, 1, 0, 0, 2>(acc, red, 1, lane); mid_put<MB, 0, 0, 2>(acc, tile, l31, kh); break;
      case 1: mid_recv<MB, 1, 0, 2, 2>(acc, red, 0, lane); mid_put<MB, 0, 2, 2>(acc, tile, l31, kh); break;
      case 2: mid_recv<MB, 1, 1, 0, 2>(acc, red, 3, lane); mid_put<MB, 1, 0, 2>(acc, tile, l31, kh); break;
      case 3: mid_recv<MB, 1, 1, 2, 2>(acc, red, 2, lane); mid_put<MB, 1, 2, 2>(acc, tile, l31, kh); break;
      case 4: mid_recv<MB, 1, 2, 0, 2>(acc, red, 5, lane); mid_put<MB, 2, 0, 2>(acc, tile, l31, kh); break;
      case 5: mid_recv<MB, 1, 2, 2, 2>(acc, red, 4, lane); mid_put<MB, 2, 2, 2>(acc, tile, l31, kh); break;
      case 6: mid_recv<MB, 1, 3, 0, 2>(acc, red, 7, lane); mid_put<MB, 3, 0, 2>(acc, tile, l31, kh); break;
      default: mid_recv<MB, 1, 3, 2, 2>(acc, red, 6, lane); mid_put<MB, 3, 2, 2>(acc, tile, l31, kh); break;
    }
  }
  __syncthreads();
  mid_flush<MB, NWK>(tile, p, n0, wave, lane);
}

// partial [S][M*N] fp32 -> c [M*N] f16 / bf16, fixed summation order
__global__ void mid_splitk_reduce_kernel(const float* __restrict__ partial, uint16_t* __restrict__ c, int64_t mn, int S, int out_bf16) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= mn) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(partial + i);
  for (int k = 1; k < S; ++k) s += *reinterpret_cast<const f32x4*>(partial + (size_t)k * mn + i);
  u16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = out_bf16 ? f32_to_bf16_bits(s[j]) : f32_to_f16_bits(s[j]);
  *reinterpret_cast<u16x4*>(c + i) = o;
}

// a [M, K] (row stride lda, f16 or bf16) -> fragment-major f16: one thread per 16-byte chunk (8 consecutive k of a row);
// rows M .. 32 mbt - 1 are zero.  bf16 is widened with saturation (see bf16_bits_to_f16_bits_sat).
__global__ void mid_pack_a_kernel(const uint16_t* __restrict__ a, int lda, int M, int K, int mbt, int is_bf16,
                                  uint32_t* __restrict__ out) {
  const int kc8 = K >> 3;
  const int64_t ci = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ci >= (int64_t)mbt * 32 * kc8) return;
  const int m = (int)(ci / kc8), kc = (int)(ci % kc8);
  u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (m < M) {
    v = *reinterpret_cast<const u16x8*>(a + (size_t)m * lda + kc * 8);
    if (is_bf16) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = bf16_bits_to_f16_bits_sat(v[j]);
    }
  }
  const int s = kc >> 1, kh = kc & 1, mb = m >> 5, l31 = m & 31;
  *reinterpret_cast<u16x8*>(out + (((size_t)s * mbt + mb) * 64 + kh * 32 + l31) * 4) = v;
}

}  // namespace aphro

using namespace aphro;

struct MidPlan { int mb, nwk, ksplit; };

// Waves per workgroup (all of them split K) and K slices (grid.y); every wave's K range holds whole groups.  8 waves when
// the column tiles alone (almost) cover the chip: one workgroup per CU, no slabs, no reduce launch.  Otherwise 4 waves
// (two workgroups fit a CU) and as many K slices as it takes to have ~1 workgroup per CU, >= 8 steps per wave: the
// fp32 slabs and their reduce pass cost more than a second resident workgroup buys.
static MidPlan mid_plan(int64_t M, int64_t N, int64_t K, int64_t gs) {
  MidPlan pl;
  pl.mb = M <= 32 ? 1 : 2;
  pl.ksplit = 1;
  const int64_t tiles = N / 128;
  pl.nwk = (tiles >= 200 && K % (8 * gs) == 0) ? 8 : 4;
  { const int v = knobs().wna16_mid_waves; if (v == 4 || (v == 8 && K % (8 * gs) == 0)) pl.nwk = v; }
  if (pl.nwk == 4)
    for (int s = 2; s <= 32; ++s) {
      if (tiles * pl.ksplit >= 200) break;      // (down_proj at M = 64: 7 slices 23.7 us, 14 slices 28.9 us)
      if (K % (4 * s * gs) == 0 && K / (4 * s) >= 128) pl.ksplit = s;
    }
  { const int s = APHRO_LAB_ENV_INT("APHRO_WNA16_MID_KSPLIT", 0); if (s >= 1 && K % (pl.nwk * s * gs) == 0) pl.ksplit = s; }
  return pl;
}

static size_t mid_lds_bytes(int mb, int nwk) {   // round 1 of the butterfly, or the last round + the output tile behind it
  const size_t r1 = (size_t)nwk * mb * 8192, r2 = (size_t)mb * 16384 + (size_t)mb * 32 * MID_LDT * 4;
  return r1 > r2 ? r1 : r2;
}

template <int MB, int NWK, bool ADEC = false, bool STRIP = false>
static int mid_launch(const Wna16MidParams& p, dim3 grid, hipStream_t st) {
  const size_t lds = mid_lds_bytes(MB, NWK);
  static bool attr_set_dev[APHRO_MAX_DEVICES] = {}; bool& attr_set = attr_set_dev[device_slot()];                  // up to 128 KiB of dynamic LDS: above the default 64 KiB limit
  if (!attr_set && lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)wna16_gemm_mid_kernel<MB, NWK, ADEC, STRIP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      set_error("wna16_gemm_mid: cannot raise the dynamic LDS limit");
      return APHRO_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((wna16_gemm_mid_kernel<MB, NWK, ADEC, STRIP>), grid, dim3(64 * NWK), lds, st, p);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_wna16_gemm_mid_supported(int64_t M, int64_t N, int64_t K, int64_t groups) {
  if (groups <= 0 || K % groups != 0) return 0;
  const int64_t gs = K / groups;
  return M >= 1 && M <= 64 && N % 128 == 0 && gs % 128 == 0 && K % (4 * gs) == 0 && (K / 8) * N * 4 < 0xffffffffll;
}

// Bytes of scratch aphro_wna16_gemm_mid needs: the fragment-major activations + the fp32 split-K slabs.
extern "C" size_t aphro_wna16_gemm_mid_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t groups) {
  if (!aphro_wna16_gemm_mid_supported(M, N, K, groups)) return 0;
  const MidPlan pl = mid_plan(M, N, K, K / groups);
  const size_t apk = (size_t)pl.mb * 32 * K * 2;
  return apk + (pl.ksplit > 1 ? (size_t)pl.ksplit * M * N * sizeof(float) : 0);
}

// c[M, N] = a[M, K] . dequant(q_weight[K/8, N] exllama order, qzeros[G, N/8], scales[G, N]) for 1 <= M <= 64.
// N % 128 == 0, group size a multiple of 128, K a multiple of 4 groups.  dtype f16 / bf16 (bf16 activations are widened to
// f16 with saturation, scales and output stay bf16).  Act-order: pass the activations already gathered (a[:, perm]).
extern "C" int aphro_wna16_gemm_mid(const void* a, const uint32_t* q_weight, const uint32_t* qzeros, const void* scales,
                                    void* c, void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K,
                                    int64_t groups, int64_t lda, int zero_offset, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "wna16_gemm_mid: dtype must be f16 or bf16");
  APHRO_CHECK(aphro_wna16_gemm_mid_supported(M, N, K, groups), "wna16_gemm_mid: unsupported shape M=%ld N=%ld K=%ld groups=%ld",
              (long)M, (long)N, (long)K, (long)groups);
  APHRO_CHECK(lda % 8 == 0 && ((uintptr_t)a % 16) == 0, "wna16_gemm_mid: a must be 16-byte aligned with lda %% 8 == 0");
  const int64_t gs = K / groups;
  const MidPlan pl = mid_plan(M, N, K, gs);
  const size_t need = aphro_wna16_gemm_mid_workspace_bytes(M, N, K, groups);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("wna16_gemm_mid: workspace %zu < %zu bytes", workspace_bytes, need);
    return APHRO_ERR_WORKSPACE;
  }
  const size_t apk_bytes = (size_t)pl.mb * 32 * K * 2;
  {
    const int64_t chunks = (int64_t)pl.mb * 32 * (K / 8);
    hipLaunchKernelGGL(mid_pack_a_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, (const uint16_t*)a, (int)lda,
                       (int)M, (int)K, pl.mb, dtype == APHRO_BF16 ? 1 : 0, (uint32_t*)workspace);
    APHRO_LAUNCH_CHECK();
  }
  Wna16MidParams p;
  p.apk = (const uint32_t*)workspace; p.qw = q_weight; p.qz = qzeros; p.sc = (const uint16_t*)scales; p.c = (uint16_t*)c;
  p.partial = (float*)((char*)workspace + apk_bytes);
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.group_size = (int)gs; p.zero_offset = zero_offset;
  p.out_bf16 = dtype == APHRO_BF16; p.scale_bf16 = dtype == APHRO_BF16; p.ksplit = pl.ksplit;
  p.mtiles = 0; p.force_partial = 0; p.act_packed = nullptr; p.strip = Wna16StripGeom{};
  const dim3 grid((unsigned)(N / 128), (unsigned)pl.ksplit);
  int rc;
  if (pl.mb == 1) rc = pl.nwk == 8 ? mid_launch<1, 8>(p, grid, st) : mid_launch<1, 4>(p, grid, st);
  else rc = pl.nwk == 8 ? mid_launch<2, 8>(p, grid, st) : mid_launch<2, 4>(p, grid, st);
  if (rc != APHRO_OK) return rc;
  if (pl.ksplit > 1) {
    const int64_t mn = M * N;
    hipLaunchKernelGGL(mid_splitk_reduce_kernel, dim3((unsigned)((mn / 4 + 255) / 256)), dim3(256), 0, st, p.partial, (uint16_t*)c,
                       mn, pl.ksplit, p.out_bf16);
    APHRO_LAUNCH_CHECK();
  }
  return APHRO_OK;
}

// ---- decode fast path: 33..64 rows, activations already in the decode kernel's fragment-major format ------------------
// (aphro_wna16_pack_a / the fused producers of the decode step), results handed to the next fused kernel:
//   act_packed != NULL : gate_up form -- columns interleaved (2j = gate_j, 2j+1 = up_j, ops.interleave_gate_up), SiluAndMul
//                        + pack in the epilogue, output = the fragment-major f16 activations [M, N/2] of the down GEMM
//                        (the role of aphro_wna16_gemm_silu_pack); needs a plan without K slices (N >= 25600)
//   slabs != NULL      : fp32 slabs [aphro_wna16_gemm_mid_ksplit][M][N] for a consumer that sums them (the norm kernel)
//   else               : c [M, N] in `dtype` (one K slice only)
extern "C" int aphro_wna16_gemm_mid_ksplit(int64_t M, int64_t N, int64_t K, int64_t groups) {
  if (M <= 32 || !aphro_wna16_gemm_mid_supported(M, N, K, groups)) return 0;
  return mid_plan(M, N, K, K / groups).ksplit;
}

static int mid_packed_impl(const void* a_packed, const uint32_t* q_weight, const uint32_t* qzeros,
                           const void* scales, void* c, void* slabs, size_t slabs_bytes, void* act_packed,
                           int64_t M, int64_t N, int64_t K, int64_t groups, int zero_offset, int dtype, int64_t strip_m,
                           void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "wna16_gemm_mid_packed: dtype must be f16 or bf16");
  APHRO_CHECK(M > 32 && aphro_wna16_gemm_mid_supported(M, N, K, groups), "wna16_gemm_mid_packed: unsupported shape M=%ld N=%ld K=%ld groups=%ld",
              (long)M, (long)N, (long)K, (long)groups);
  APHRO_CHECK(((uintptr_t)a_packed % 16) == 0, "wna16_gemm_mid_packed: packed activations must be 16-byte aligned");
  const int64_t gs = K / groups;
  const MidPlan pl = mid_plan(M, N, K, gs);
  Wna16MidParams p;
  p.apk = (const uint32_t*)a_packed; p.qw = q_weight; p.qz = qzeros; p.sc = (const uint16_t*)scales; p.c = (uint16_t*)c;
  p.partial = (float*)slabs;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.group_size = (int)gs; p.zero_offset = zero_offset;
  p.out_bf16 = dtype == APHRO_BF16; p.scale_bf16 = dtype == APHRO_BF16; p.ksplit = pl.ksplit;
  p.mtiles = (int)((M + 15) / 16); p.force_partial = 0; p.act_packed = nullptr;
  p.strip = Wna16StripGeom{};
  APHRO_CHECK(strip_m == 0 || wna16_strip_fill(p.strip, strip_m, N, K, groups),
              "wna16_gemm_mid_packed_strip: no strip-major form for M class %ld, N=%ld, K=%ld, groups=%ld", (long)strip_m, (long)N, (long)K, (long)groups);
  if (act_packed != nullptr) {
    APHRO_CHECK(pl.ksplit == 1 && N % 256 == 0, "wna16_gemm_mid_packed: the SiluAndMul form needs one K slice and N/2 %% 128 == 0 (N=%ld)", (long)N);
    p.act_packed = (uint16_t*)act_packed;
  } else if (slabs != nullptr) {
    APHRO_CHECK(slabs_bytes >= (size_t)pl.ksplit * M * N * sizeof(float), "wna16_gemm_mid_packed: slabs too small");
    p.force_partial = 1;
  } else {
    APHRO_CHECK(c != nullptr && pl.ksplit == 1, "wna16_gemm_mid_packed: this shape needs the slab form (%d K slices)", pl.ksplit);
  }
  const dim3 grid((unsigned)(N / 128), (unsigned)pl.ksplit);
  if (p.strip.on) return pl.nwk == 8 ? mid_launch<2, 8, true, true>(p, grid, st) : mid_launch<2, 4, true, true>(p, grid, st);
  return pl.nwk == 8 ? mid_launch<2, 8, true>(p, grid, st) : mid_launch<2, 4, true>(p, grid, st);
}
extern "C" int aphro_wna16_gemm_mid_packed(const void* a_packed, const uint32_t* q_weight, const uint32_t* qzeros,
                                           const void* scales, void* c, void* slabs, size_t slabs_bytes, void* act_packed,
                                           int64_t M, int64_t N, int64_t K, int64_t groups, int zero_offset, int dtype,
                                           void* stream) {
  return mid_packed_impl(a_packed, q_weight, qzeros, scales, c, slabs, slabs_bytes, act_packed, M, N, K, groups, zero_offset, dtype, 0, stream);
}
// The same launch on the STRIP-MAJOR copy of the weights (aphro_wna16_strip_relayout for the M class strip_m, normally 32):
// every 16-byte weight load takes its strip-major address (wna16_strip.h) -- same loads, same bits.  For a layer that keeps
// one resident copy of each matrix (model.DecoderLayer.enable_one_copy).
extern "C" int aphro_wna16_gemm_mid_packed_strip(const void* a_packed, const uint32_t* q_weight_strip, const uint32_t* qzeros,
                                                 const void* scales, void* c, void* slabs, size_t slabs_bytes, void* act_packed,
                                                 int64_t M, int64_t N, int64_t K, int64_t groups, int zero_offset, int dtype,
                                                 int64_t strip_m, void* stream) {
  APHRO_CHECK(strip_m >= 1 && strip_m <= 64, "wna16_gemm_mid_packed_strip: strip_m=%ld", (long)strip_m);
  return mid_packed_impl(a_packed, q_weight_strip, qzeros, scales, c, slabs, slabs_bytes, act_packed, M, N, K, groups, zero_offset, dtype,
                         strip_m, stream);
}
