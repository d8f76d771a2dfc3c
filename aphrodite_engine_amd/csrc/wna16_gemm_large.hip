// W4A16 (GPTQ / AWQ-repacked int4) GEMM for PREFILL-sized M on gfx950 -- the Marlin role at large M.
//
// Replaces the reference's "reconstruct to fp16 [K, N] + cuBLAS/hipBLAS" fallback above 50 rows
// (kernels/quantization/gptq/q_gemm.cu:1529-1544: temp_dq written + read = 4x the int4 bytes) and fills the
// gptq_marlin_gemm slot (kernels/quantization/gptq_marlin/gptq_marlin.cu:544, 2247) for M > 64.  MFMA bound:
//   * D^T[n][m] = W^T[n][k] . A^T[k][m] with v_mfma_f32_32x32x16_f16: the packed exllama dword (8 consecutive k of
//     ONE column) is exactly one lane's A-operand fragment (lane: row n = lane & 31, k = 8 (lane >> 5) .. +7), so the
//     weights never need a transposing LDS layout; the activations are the B operand (lane: column m = lane & 31).
//     One dequantised weight fragment feeds 4 MFMAs (4 m-blocks of 32 rows), one activation fragment 2.
//   * int4 -> f16 in registers, the reference's own numerics (q_gemm.cu:1394-1434, qdq_4.cuh:38-63): (q - z) exactly
//     through the 1024 + q trick, one rounding in the multiply by the group scale.  13 VALU per dword against 128
//     MFMA cycles: the dequant hides in the MFMA shadow.
//   * activations: global -> LDS with direct-to-LDS loads (no VGPR round trip), 128-byte rows XOR-swizzled on the
//     SOURCE address (the LDS image of such a load is lane-linear), read back with conflict-free ds_read_b128;
//     weights: the tile's 8 packed rows go through LDS too (4-8 KiB) so the K loop has only LDS-DMA in flight;
//     group scales / zeros of the workgroup's columns are staged once for the whole K range.
//   * 2 LDS stages: the next K tile's loads are issued before the current tile's MFMAs, one barrier per tile.
//   * XCD-aware tile order: consecutive workgroups of one XCD share the A row panel (2 MiB, L2 resident) and walk
//     the column tiles.
// Workgroup = WN x WM waves (WM in the M direction), wave tile = 128 rows x 64 columns, K tile 64.
#include <type_traits>

#include "common.h"

namespace aphro {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;

struct Wna16LargeParams {
  const uint16_t* a;      // [M, lda] f16
  const uint32_t* qw;     // [K/8, N] exllama order
  const uint32_t* qz;     // [G, N/8]
  const uint16_t* sc;     // [G, N] f16 (or bf16: see scale_is_bf16)
  uint16_t* c;            // [M, N]
  int M, N, K, lda;
  int group_size;         // multiple of 64
  int zero_offset;
  int out_bf16;           // round the result to bf16 instead of f16
  int scale_bf16;         // scales are stored as bf16
  int tiles_m, tiles_n;
  int ksplit;             // > 1: blockIdx.y owns K / ksplit consecutive k and writes an fp32 slab of `partial`
  float* partial;         // [ksplit][M][N]
  // stream-K form (see fp8_gemm_large.hip): one persistent workgroup per CU, equal (tile, K tile) unit ranges; `partial`
  // then holds the accumulator images [grid][NWAVE * 32 KiB] and `flags` one "image published" word per workgroup
  int streamk, grid;      // grid: persistent workgroups (= CUs)
  unsigned* flags;
  // W8A16 form (template WFP8): e4m3 weights [N, K] (K-contiguous), per-tensor / per-channel fp32 scales applied in the
  // epilogue, optional bias in the output type.  qw / qz / sc / group_size / zero_offset are unused then.
  const uint8_t* w8;
  const float* w_scales;
  int w_per_channel;
  const uint16_t* bias;
  // SiluAndMul epilogue (round 5): the columns of the weights are interleaved (gate_j, up_j) pairs (ops.interleave_gate_up)
  // and `c` is the activation [M, N / 2]: act_j = round(round(silu(round(gate_j))) * round(up_j)) -- the bits of the GEMM
  // followed by silu_and_mul on its rounded output, without the [M, N] round trip through HBM (0.9 GB per layer of an
  // 8192-token Llama-3-8B prompt)
  int silu;
  // two-pass form (round 6, template WDMA of the eight-phase kernel): the weights dequantised ONCE per call into wt = f16
  // W^T [N, K] (k contiguous; wna16_dequant_t_kernel) and staged by LDS-DMA like the activations.  At M = 8192 every weight is
  // otherwise dequantised by 32 row-tile workgroups: the in-loop dequantisation costs 15 % of the K loop
  // (profiles/r6_w4_two_pass.txt).
  const uint16_t* wt;
  // strip-major weights (round 6, template STRIP of the eight-phase kernel / of the dequantise-transpose pass): qw is the
  // strip-major copy the <= 32-row decode kernels stream (aphro_wna16_strip_relayout, wna16_gemm_resident.hip) -- the ONLY
  // resident copy of the matrix.  Every 16-byte piece (one packed row, four 4-aligned columns) of the [K/8, N] order is 16
  // contiguous bytes there too; its dword offset is
  //   ((ky * S + strip) * nwv + wv) * wave_dw + colbase(col) + mult(col) * (256 s + 64 u + 16 g)
  // with row = 16 * ((ky * nwv + wv) * nseg + s) + 4 g + u and (colbase, mult) = (pass * nseg * 1024 + col % 64, 4) in the
  // 64-column passes, (np4 * nseg * 1024 + col - 64 np4, rem) in the remainder pass (columns inside the strip).
  int strip;
  int st_nwv, st_nseg, st_np4, st_rem, st_S;
  uint32_t st_wave_dw;
  uint32_t st_inv_nseg, st_inv_nwv;       // ceil(2^16 / d): (x * inv) >> 16 == x / d for every x the kernel divides (host-checked)
};

// The strip-major dword offset of the piece at packed row `row`, column `col` (4-aligned) -- see Wna16LargeParams::strip.
__device__ __forceinline__ uint32_t lg_strip_dw(const Wna16LargeParams& p, int row, int col) {
  const int cw = 64 * p.st_np4 + 16 * p.st_rem;
  const int strip = col / cw, cin = col - strip * cw;
  const int seg = row >> 4, g = (row >> 2) & 3, u = row & 3;
  const int kw = seg / p.st_nseg, s = seg - kw * p.st_nseg;
  const int ky = kw / p.st_nwv, wv = kw - ky * p.st_nwv;
  const uint32_t chunk = (uint32_t)((ky * p.st_S + strip) * p.st_nwv + wv) * p.st_wave_dw;
  const uint32_t R = 256u * s + 64u * u + 16u * g;
  if (cin < 64 * p.st_np4) return chunk + (uint32_t)(cin >> 6) * p.st_nseg * 1024u + (cin & 63) + 4u * R;
  return chunk + (uint32_t)p.st_np4 * p.st_nseg * 1024u + (cin - 64 * p.st_np4) + (uint32_t)p.st_rem * R;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t lg_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

__device__ __forceinline__ uint32_t vlg_and_or(uint32_t a, uint32_t mask, uint32_t orv) {
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask), "v"(orv));
  return r;
}

// one exllama dword -> 8 scaled f16: ((1024 + q) - (1024 + z)) * s  (exact difference, one rounding)
__device__ __forceinline__ f16x8 dq8_scaled(uint32_t w, f16x2 zh, f16x2 zh16, f16x2 sc) {
  const f16x2 inv16 = {(f16)0.0625f, (f16)0.0625f};
  const uint32_t magic = 0x64006400u;
  const uint32_t q0 = vlg_and_or(w, 0x000f000fu, magic);
  const uint32_t q1 = vlg_and_or(w, 0x00f000f0u, magic);
  const uint32_t w8 = w >> 8;
  const uint32_t q2 = vlg_and_or(w8, 0x000f000fu, magic);
  const uint32_t q3 = vlg_and_or(w8, 0x00f000f0u, magic);
  const f16x2 d0 = (__builtin_bit_cast(f16x2, q0) - zh) * sc;
  const f16x2 d1 = (__builtin_bit_cast(f16x2, q1) * inv16 + zh16) * sc;
  const f16x2 d2 = (__builtin_bit_cast(f16x2, q2) - zh) * sc;
  const f16x2 d3 = (__builtin_bit_cast(f16x2, q3) * inv16 + zh16) * sc;
  u32x4 r = {__builtin_bit_cast(uint32_t, d0), __builtin_bit_cast(uint32_t, d1), __builtin_bit_cast(uint32_t, d2),
             __builtin_bit_cast(uint32_t, d3)};
  return __builtin_bit_cast(f16x8, r);
}

// What happens to a wave's accumulators once the K range of a segment is done: published (stream-K, not the owner), merged
// + converted + stored (owner), or written as an fp32 slab (split-K form).  Shared by both kernels.
template <int NWAVE, bool WFP8>
__device__ __forceinline__ void wna16_large_finish(const Wna16LargeParams& p, f32x16 (&acc)[2][4], unsigned char* smem,
                                                   const __amdgpu_buffer_rsrc_t rp, bool head, bool tail, int w, int GW, int64_t U,
                                                   int tile, int ktiles_total, int m0, int n0, int wave, int wm, int wn, int lane) {
  const int kh = lane >> 5, l31 = lane & 31;
  // ---- what happens to the accumulators: lane holds, for row m = mb*32 + l31, columns nb*32 + 8 q + 4 kh + (0..3) ----------
  if (p.streamk && !head) {
    // not the owner of this tile (always a workgroup's FIRST segment): publish the accumulators as they sit in the
    // registers -- image [wave][quad = (nb*4 + mb)*4 + q][lane] f32x4, 1 KiB per store instruction, write-through
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[nb][mb][4 * q], acc[nb][mb][4 * q + 1], acc[nb][mb][4 * q + 2], acc[nb][mb][4 * q + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rp,
                                                 ((w * NWAVE + wave) * 32 + (nb * 4 + mb) * 4 + q) * 1024 + lane * 16, 0, 17);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains ...
    __syncthreads();                                     // ... before ONE lane raises the flag
    if (threadIdx.x == 0) __hip_atomic_store(p.flags + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (p.streamk && !tail) {
    // owner of a tile whose K range continues in the following workgroups (always the LAST segment): add their images in
    // workgroup order (deterministic).  Bounded wait, see fp8_gemm_large.hip.
    const int64_t tile_end = (int64_t)(tile + 1) * ktiles_total;
    for (int j = w + 1; j < GW && j * U / GW < tile_end; ++j) {
      if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(p.flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          __builtin_amdgcn_s_sleep(4);
          if (wall_clock64() - t0 > 500000000ull) __builtin_trap();
        }
      }
      __syncthreads();
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(
                rp, ((j * NWAVE + wave) * 32 + (nb * 4 + mb) * 4 + q) * 1024 + lane * 16, 0, 17);
            const f32x4 v = __builtin_bit_cast(f32x4, raw);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[nb][mb][4 * q + r] += v[r];
          }
    }
  }
  if (!p.streamk && p.ksplit > 1) {          // fp32 slab of this K range; summed by splitk_reduce_large_kernel
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int row = m0 + wm * 128 + mb * 32 + l31;
      if (row >= p.M) continue;
      float* prow = p.partial + ((size_t)blockIdx.y * p.M + row) * p.N + n0 + wn * 64;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(prow + nb * 32 + 8 * q + 4 * kh) =
              f32x4{acc[nb][mb][4 * q], acc[nb][mb][4 * q + 1], acc[nb][mb][4 * q + 2], acc[nb][mb][4 * q + 3]};
    }
    return;
  }
  unsigned char* region = smem + wave * 16384;
  f32x4 wsv[2][4], bsv[2][4];     // W8A16: per-channel scale and bias of this lane's 4-column groups, fetched up front
  if constexpr (WFP8) {
    const float s0 = p.w_per_channel ? 1.f : p.w_scales[0];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + wn * 64 + nb * 32 + 8 * q + 4 * kh;
        wsv[nb][q] = p.w_per_channel ? *reinterpret_cast<const f32x4*>(p.w_scales + col) : f32x4{s0, s0, s0, s0};
        bsv[nb][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
          const u16x4 b4 = *reinterpret_cast<const u16x4*>(p.bias + col);
#pragma unroll
          for (int r = 0; r < 4; ++r) bsv[nb][q][r] = p.out_bf16 ? bf16_bits_to_f32(b4[r]) : f16_bits_to_f32(b4[r]);
        }
      }
  }
  if (p.silu) {
    // a lane's 4 consecutive columns of an accumulator quad are two (gate, up) pairs -> two activations = one dword.  The wave
    // tile becomes 128 rows x 32 activations: a [128][64 B] image in the wave's LDS region, the 16-byte chunk XOR-ed with
    // (row >> 2) & 3 (writes 2-way conflicted, reads clean), flushed as whole 64-byte row segments.
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = mb * 32 + l31, c4 = nb * 8 + 2 * q + kh;           // 4-byte chunk 0..15 of the row
          uint32_t w2;
          if (p.out_bf16) {
            const uint16_t a0 = silu_mul_bits<BFloat>(BFloat::to_f32(BFloat::from_f32(acc[nb][mb][4 * q])), BFloat::to_f32(BFloat::from_f32(acc[nb][mb][4 * q + 1])));
            const uint16_t a1 = silu_mul_bits<BFloat>(BFloat::to_f32(BFloat::from_f32(acc[nb][mb][4 * q + 2])), BFloat::to_f32(BFloat::from_f32(acc[nb][mb][4 * q + 3])));
            w2 = (uint32_t)a0 | ((uint32_t)a1 << 16);
          } else {
            const uint16_t a0 = silu_mul_bits<Half>(Half::to_f32(Half::from_f32(acc[nb][mb][4 * q])), Half::to_f32(Half::from_f32(acc[nb][mb][4 * q + 1])));
            const uint16_t a1 = silu_mul_bits<Half>(Half::to_f32(Half::from_f32(acc[nb][mb][4 * q + 2])), Half::to_f32(Half::from_f32(acc[nb][mb][4 * q + 3])));
            w2 = (uint32_t)a0 | ((uint32_t)a1 << 16);
          }
          *reinterpret_cast<uint32_t*>(region + row * 64 + ((((c4 >> 2) ^ ((row >> 2) & 3)) << 4) | ((c4 & 3) << 2))) = w2;
        }
    const int64_t ldc = p.N / 2;
    const int64_t origin = (int64_t)(m0 + wm * 128) * ldc + (n0 + wn * 64) / 2;      // (elements of the activation)
    const int64_t bytes_left = ((int64_t)p.M * ldc - origin) * 2;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        p.c + origin, 0, (uint32_t)(bytes_left > 0xffffffffll ? 0xffffffffll : (bytes_left < 0 ? 0 : bytes_left)), 0x00020000);
    const int r16 = lane >> 2, c16 = lane & 3;
    u32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = i * 16 + r16;
      v[i] = *reinterpret_cast<const u32x4*>(region + row * 64 + ((c16 ^ ((row >> 2) & 3)) << 4));
    }
    const uint32_t voff = (uint32_t)r16 * (uint32_t)ldc * 2u + (uint32_t)c16 * 16u;
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_amdgcn_raw_buffer_store_b128(v[i], rc, voff, (uint32_t)i * 16u * (uint32_t)ldc * 2u, 0);
    return;
  }
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v0 = acc[nb][mb][4 * q], v1 = acc[nb][mb][4 * q + 1], v2 = acc[nb][mb][4 * q + 2], v3 = acc[nb][mb][4 * q + 3];
        if constexpr (WFP8) {
          v0 = v0 * wsv[nb][q][0] + bsv[nb][q][0]; v1 = v1 * wsv[nb][q][1] + bsv[nb][q][1];
          v2 = v2 * wsv[nb][q][2] + bsv[nb][q][2]; v3 = v3 * wsv[nb][q][3] + bsv[nb][q][3];
        }
        epi_put(region, mb * 32 + l31, nb * 8 + 2 * q + kh,
                p.out_bf16 ? u32x2{pack2_16<true>(v0, v1), pack2_16<true>(v2, v3)} : u32x2{pack2_16<false>(v0, v1), pack2_16<false>(v2, v3)});
      }
  {
    const int64_t origin = (int64_t)(m0 + wm * 128) * p.N + n0 + wn * 64;      // (elements)
    epi_flush_buf(region, p.c + origin, p.N, ((int64_t)p.M * p.N - origin) * 2, lane);
  }
}

template <int WM, int WN, int STAGES, bool WFP8>
__global__ __launch_bounds__(WM * WN * 64) void wna16_gemm_large_kernel(Wna16LargeParams p) {
  constexpr int NWAVE = WM * WN;
  constexpr int BM = 128 * WM, BN = 64 * WN, BK = 64;
  constexpr int A_STAGE = BM * BK * 2;              // bytes
  constexpr int B_STAGE = WFP8 ? BN * BK : (BK / 8) * BN * 4;   // W8A16: BN rows of 64 e4m3
  constexpr int STAGE = A_STAGE + B_STAGE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [STAGES][A tile | B tile] [scales: G x BN f16] [zeros: G x BN/8 words]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int kh = lane >> 5, l31 = lane & 31;
  constexpr int IMAGE = NWAVE * 32 * 1024;          // bytes of one workgroup's accumulator image (stream-K)
  // ---- this workgroup's units [u, u_end): unit = tile * (K / 64) + K tile.  XCD x gets a contiguous range of logical
  // ids; within it the column tiles are the fast index (neighbours share the A row panel) --------------------------------
  const int ktiles_total = p.K / BK;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int GW = gridDim.x;
  const int64_t U = (int64_t)ntiles * ktiles_total;
  auto xcd_contiguous = [](int bid, int n) {
    const int q = n / 8, r = n % 8, xcd = bid % 8, k = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;    // bijective for any n
  };
  int u, u_end, w = 0;
  if (p.streamk) {
    w = xcd_contiguous(blockIdx.x, GW);
    u = (int)(w * U / GW);
    u_end = (int)((w + 1) * U / GW);
  } else {
    const int per = ktiles_total / p.ksplit;
    u = xcd_contiguous(blockIdx.x, ntiles) * ktiles_total + blockIdx.y * per;
    u_end = u + per;
  }
  const __amdgpu_buffer_rsrc_t ra = lg_rsrc(p.a, (uint32_t)((size_t)p.M * p.lda * 2));
  const __amdgpu_buffer_rsrc_t rb = WFP8 ? lg_rsrc(p.w8, (uint32_t)((size_t)p.N * p.K))
                                         : lg_rsrc(p.qw, (uint32_t)((size_t)(p.K >> 3) * p.N * 4));
  const __amdgpu_buffer_rsrc_t rp = lg_rsrc(p.partial, p.streamk ? (uint32_t)((size_t)GW * IMAGE) : 0u);
  bool first_segment = true;
  while (u < u_end) {
  const int tile = u / ktiles_total;
  const int k0t = u - tile * ktiles_total;
  const int k1t = min(ktiles_total, k0t + (u_end - u));
  u += k1t - k0t;
  const bool head = k0t == 0, tail = k1t == ktiles_total;
  const int k_begin = k0t * BK, klen = (k1t - k0t) * BK;   // this segment's K range: [k_begin, k_begin + klen)
  const int g_begin = WFP8 ? 0 : k_begin / p.group_size;
  const int G = WFP8 ? 0 : (k_begin + klen - 1) / p.group_size - g_begin + 1;   // quantisation groups it touches
  uint16_t* meta_sc = reinterpret_cast<uint16_t*>(smem + STAGES * STAGE);
  uint32_t* meta_z = reinterpret_cast<uint32_t*>(smem + STAGES * STAGE + G * BN * 2);
  const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  if (!first_segment) __syncthreads();                // the previous segment's epilogue is done with the LDS
  first_segment = false;

  // ---- prologue: group scales / zeros of this workgroup's BN columns for the whole K range -> LDS ------------------
  if constexpr (!WFP8)
  for (int i = threadIdx.x; i < G * (BN / 2); i += NWAVE * 64) {
    const int g = i / (BN / 2), j = i % (BN / 2);
    reinterpret_cast<uint32_t*>(meta_sc)[i] = reinterpret_cast<const uint32_t*>(p.sc)[((size_t)(g_begin + g) * p.N + n0) / 2 + j];
  }
  if constexpr (!WFP8)
  for (int i = threadIdx.x; i < G * (BN / 8); i += NWAVE * 64) {
    const int g = i / (BN / 8), j = i % (BN / 8);
    meta_z[i] = p.qz[(size_t)(g_begin + g) * (p.N >> 3) + (n0 >> 3) + j];
  }

  // ---- staging (direct-to-LDS) ---------------------------------------------------------------------------------------
  // A tile: BM rows x 128 B; one DMA instruction = 8 rows (lane -> row l/8, 16-byte slot l%8).  LDS slot s' of row r
  // holds global slot s' ^ f(r), f(r) = (r >> 1) & 7  (conflict-free ds_read_b128 below).
  constexpr int A_INSTR = BM / 8;                   // DMA instructions per A tile
  constexpr int A_PER_WAVE = A_INSTR / NWAVE;
  int a_voff[A_PER_WAVE];
#pragma unroll
  for (int i = 0; i < A_PER_WAVE; ++i) {
    const int row = (wave * A_PER_WAVE + i) * 8 + (lane >> 3);
    const int slot = (lane & 7) ^ ((row >> 1) & 7);
    a_voff[i] = (min(m0 + row, p.M - 1) * p.lda + slot * 8) * 2;
  }
  // B tile: 8 packed rows x BN dwords = BN * 32 bytes; one DMA instruction = 1 KiB = 256 dwords
  constexpr int B_INSTR = B_STAGE / 1024;
  const int b_row_bytes = p.N * 4;
  // strip-major weights (Wna16LargeParams::strip): the lane's column is the same in every instruction of the tile (256 % BN
  // == 0) -- its column part of the address and its row multiplier once per segment, the row part per instruction
  uint32_t st_colbase = 0, st_mult = 0;
  if (!WFP8 && p.strip) {
    const int cw = 64 * p.st_np4 + 16 * p.st_rem;
    const int col = n0 + (lane * 4) % BN, strip = col / cw, cin = col - strip * cw;
    const bool p4 = cin < 64 * p.st_np4;
    st_colbase = (uint32_t)(strip * p.st_nwv) * p.st_wave_dw +
                 (p4 ? (uint32_t)(cin >> 6) * p.st_nseg * 1024u + (cin & 63) : (uint32_t)p.st_np4 * p.st_nseg * 1024u + (cin - 64 * p.st_np4));
    st_mult = p4 ? 4u : (uint32_t)p.st_rem;
  }
  auto stage = [&](int st, int kt) {
    unsigned char* sa = smem + st * STAGE;
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
      // (a local copy: with the captured array element `a_voff[i]` spelled in the builtin call itself, hipcc 7.2
      //  silently drops the HOST-side launch stub of every instantiation of this kernel template -- the library then
      //  fails to load with an undefined kernel symbol)
      const int voff = a_voff[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_ptr)(sa + (wave * A_PER_WAVE + i) * 1024), 16, voff,
                                               (k_begin + kt * BK) * 2, 0, 0);
    }
    for (int i = wave; i < B_INSTR; i += NWAVE) {
      if constexpr (WFP8) {
        // 16 weight rows x 64 B per instruction (lane -> row i * 16 + l / 4, 16-byte slot l % 4); LDS slot s' of row r
        // holds global slot s' ^ ((r >> 2) & 3): the 8-byte fragment reads below are then at most 2-way conflicted
        const int row = i * 16 + (lane >> 2);
        const int slot = (lane & 3) ^ ((row >> 2) & 3);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_ptr)(sa + A_STAGE + i * 1024), 16,
                                                 (n0 + row) * p.K + slot * 16, k_begin + kt * BK, 0, 0);
      } else {
        // dword index d = i * 256 + 4 * lane .. +3 of the [8][BN] tile
        const int d = i * 256 + lane * 4;
        const int row = d / BN, col = d % BN;
        if (p.strip) {
          const uint32_t ra_ = (uint32_t)(k_begin / 8 + kt * 8 + row), seg = ra_ >> 4;
          const uint32_t kw = (seg * p.st_inv_nseg) >> 16, sg = seg - kw * (uint32_t)p.st_nseg;
          const uint32_t ky = (kw * p.st_inv_nwv) >> 16, wv = kw - ky * (uint32_t)p.st_nwv;
          const uint32_t off = (ky * (uint32_t)(p.st_S * p.st_nwv) + wv) * p.st_wave_dw + st_colbase +
                               st_mult * (256u * sg + 64u * (ra_ & 3u) + 16u * ((ra_ >> 2) & 3u));
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_ptr)(sa + A_STAGE + i * 1024), 16, (int)(off * 4u), 0, 0, 0);
        } else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_ptr)(sa + A_STAGE + i * 1024), 16,
                                                 (n0 + col) * 4 + row * b_row_bytes, (k_begin / 8 + kt * 8) * b_row_bytes, 0, 0);
      }
    }
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][mb][r] = 0.f;

  const int ktiles = klen / BK;
  // DMA instructions one wave issues per K tile (uniform over the waves: the counted waits below rely on it)
  constexpr int DMA_PER_TILE = A_PER_WAVE + B_INSTR / NWAVE;
  static_assert(B_INSTR % NWAVE == 0, "every wave must issue the same number of B-tile loads");
  stage(0, 0);
  if constexpr (STAGES == 3) { if (ktiles > 1) stage(1, 1); }
  __syncthreads();                                     // (also publishes the group scales / zeros staged above)

  f16x2 zh[2], zh16[2], scv[2];
  int cur_group = -1;
  for (int kt = 0; kt < ktiles; ++kt) {
    const int st = kt % STAGES;
    if constexpr (STAGES == 3) {
      // tile kt has landed once at most the loads of tile kt+1 are outstanding; a RAW barrier (no vmcnt(0) drain:
      // __syncthreads() would wait for the tile in flight) publishes it to every wave and tells everyone that
      // stage (kt + 2) % 3 -- read during tile kt - 1 -- is free, so tile kt + 2 can be issued under this tile's MFMAs
      if (kt + 1 < ktiles) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DMA_PER_TILE) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 2 < ktiles) stage((kt + 2) % 3, kt + 2);
    } else {
      if (kt + 1 < ktiles) stage(st ^ 1, kt + 1);     // next tile's DMA in flight under this tile's MFMAs
    }
    const int grp = WFP8 ? 0 : (k_begin + kt * BK) / p.group_size - g_begin;
    if (!WFP8 && grp != cur_group) {                  // wave-uniform: new quantisation group
      cur_group = grp;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int col = wn * 64 + nb * 32 + l31;      // column inside the workgroup tile
        const uint16_t sb = meta_sc[grp * BN + col];
        const float sf = p.scale_bf16 ? bf16_bits_to_f32(sb) : f16_bits_to_f32(sb);
        const int z = (int)((meta_z[grp * (BN / 8) + (col >> 3)] >> ((col & 7) * 4)) & 0xf) + p.zero_offset;
        const f16 s16 = (f16)sf;
        const f16 a16 = __builtin_bit_cast(f16, (uint16_t)(0x6400 | z));   // 1024 + z
        const f16 b16 = (f16)(float)(-64 - z);
        scv[nb] = f16x2{s16, s16};
        zh[nb] = f16x2{a16, a16};
        zh16[nb] = f16x2{b16, b16};
      }
    }
    const unsigned char* sa = smem + st * STAGE;
    const uint32_t* sb = reinterpret_cast<const uint32_t*>(sa + A_STAGE);
    // Fragments of k-step j + 1 are read from LDS while the MFMAs of step j run (two register sets; hipcc left to
    // itself reuses ONE fragment register and waits lgkmcnt(0) in front of every MFMA pair: 0.40 -> see DESIGN.md).
    u32x4 af[2][4];
    uint32_t wraw[2][2], wrawh[2][2];   // (wrawh: second dword of the 8 e4m3 of the W8A16 form)
    auto read_frags = [&](int j, u32x4 (&a4)[4], uint32_t (&w2)[2], uint32_t (&w2h)[2]) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        if constexpr (WFP8) {   // 8 e4m3 = k 16 j + 8 kh .. + 7 of weight row n: 8-byte piece q = 2 j + kh of the row
          const int row = wn * 64 + nb * 32 + l31;
          const int q = 2 * j + kh;
          const u32x2 w8v = *reinterpret_cast<const u32x2*>(sa + A_STAGE + row * 64 + ((((q >> 1) ^ ((row >> 2) & 3)) << 4) | ((q & 1) << 3)));
          w2[nb] = w8v[0];
          w2h[nb] = w8v[1];
        } else {
          w2[nb] = sb[(2 * j + kh) * BN + wn * 64 + nb * 32 + l31];
        }
      }
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const int row = wm * 128 + mb * 32 + l31;
        const int slot = (2 * j + kh) ^ ((row >> 1) & 7);
        a4[mb] = *reinterpret_cast<const u32x4*>(sa + row * 128 + slot * 16);
      }
    };
    read_frags(0, af[0], wraw[0], wrawh[0]);
#pragma unroll
    for (int j = 0; j < BK / 16; ++j) {
      if (j + 1 < BK / 16) read_frags(j + 1, af[(j + 1) & 1], wraw[(j + 1) & 1], wrawh[(j + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);            // keep the next step's LDS reads ABOVE this step's MFMAs
      f16x8 wf[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        if constexpr (WFP8) {   // e4m3 -> f16 is exact: 4 packed hardware converts, the scale waits for the epilogue
          const uint32_t lo = wraw[j & 1][nb], hi = wrawh[j & 1][nb];
          const u32x4 r = {fp8x2_to_T<Half, false, false>(lo), fp8x2_to_T<Half, false, true>(lo),
                           fp8x2_to_T<Half, false, false>(hi), fp8x2_to_T<Half, false, true>(hi)};
          wf[nb] = __builtin_bit_cast(f16x8, r);
        } else {
          wf[nb] = dq8_scaled(wraw[j & 1][nb], zh[nb], zh16[nb], scv[nb]);
        }
      }
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[nb], __builtin_bit_cast(f16x8, af[j & 1][mb]), acc[nb][mb], 0, 0, 0);
    }
    if constexpr (STAGES == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile landed (this wave's share)
      __syncthreads();                                   // ... everyone's; and everyone is done reading `st`
    }
  }

  if constexpr (STAGES == 3) __syncthreads();      // every wave is done with the stage buffers
  wna16_large_finish<NWAVE, WFP8>(p, acc, smem, rp, head, tail, w, GW, U, tile, ktiles_total, m0, n0, wave, wm, wn, lane);
  }   // segments
}

// ---- the eight-phase kernel (round 5): 256 x 256 x 64 tile, 8 waves = 2 groups (wm = wave >> 2; waves w and w + 4 share a
// SIMD) x 4 column strips (wn = wave & 3), the schedule of fp8_gemm_large8_kernel (fp8_gemm_large.hip; read its header for
// the phase table and the RAW / WAR argument) ------------------------------------------------------------------------------
// What differs from the FP8 form: the WEIGHT tile is not copied, it is built.  Every thread owns four neighbouring columns
// (n = 4 lane .. + 3 of the tile) at one packed row (r = wave: k = 8 r .. + 7 of the K tile), so a K tile's int4 weights are
// ONE 16-byte load per thread (a wave reads 1 KiB of one q_weight row), its group scales one 8-byte and its zero points one
// 4-byte load.  They are requested in the shadow of phase 4's MFMAs, one whole K tile before phase 4 of the next tile turns
// them into f16 -- (q - z) * s, the reference's own numerics (q_gemm.cu:1394-1434, qdq_4.cuh:38-63) -- and writes them as
// [n][64 k] rows (128 B, the activations' XOR swizzle) into the LDS buffer that the tile after reads with ds_read_b128
// like the activations.  Dequantised ONCE per workgroup (the kernel above converts every dword in both waves that share
// its columns), 52 VALU + 4 ds_write_b128 per thread per K tile, in the one phase that has no fragment reads.
// All loads of the K loop are inline asm or LDS-DMA: nothing hipcc would wait vmcnt(0) for (cdna_hip_programming.md,
// "Three .s-level traps" (b)); counts by hand:
//   per wave and K tile t:  P1 shadow: 2 LDS-DMA Ah1(t+1)   P3 shadow: 2 LDS-DMA Ah0(t+2)   P4 shadow: W / scale / zero loads of t+2
//   P2 read block: vmcnt(2) -> the packed weights of tile t+1 (requested in P4 of tile t-1) sit in registers; dword 0 is
//   dequantised + written there, dword 1 in P3's read block, dwords 2, 3 in P4's (which has no fragment reads).
//   (profiles/r5_w4_large_lab.txt: all four in P4's read block = 100 VALU between two barriers while the partner's MFMA cluster
//   is 256 cycles long: 3260 cycles per K tile against 2048 of MFMA time; interleaved with the wave's OWN MFMAs: 3520.)
//   P4 read block: vmcnt(2) -> the activations of tile t+1 are whole; lgkmcnt(0): this wave's weight rows are written;
//   barrier; first reader: P1 of tile t+1.
template <int OFF>
__device__ __forceinline__ void lg_lds_read128(u32x4& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lg_lds_write128(uint32_t addr, const u32x4& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lg_touch(u32x4& x) { asm volatile("" : "+v"(x)); }

template <bool BF16OUT, bool WDMA = false, bool STRIP = false>
__global__ __launch_bounds__(512) void wna16_gemm_large8_kernel(Wna16LargeParams p) {
  static_assert(!(WDMA && STRIP), "the two-pass form reads f16 W^T: the strip-major order is pass 1's business");
  constexpr int NWAVE = 8, BM = 256, BN = 256, BK = 64;
  constexpr int A_REGION = BM * BK * 2;             // 32 KiB
  constexpr int BUF = 2 * A_REGION;                 // 64 KiB per K tile: [activations | f16 weights]
  constexpr int IMAGE = NWAVE * 32 * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int kh = lane >> 5, l31 = lane & 31;
  const int ktiles_total = p.K / BK;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int GW = gridDim.x;
  const int64_t U = (int64_t)ntiles * ktiles_total;
  auto xcd_contiguous = [](int bid, int n) {
    const int q = n / 8, r = n % 8, xcd = bid % 8, k = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  };
  const int w = xcd_contiguous(blockIdx.x, GW);
  int u = (int)(w * U / GW);
  const int u_end = (int)((w + 1) * U / GW);

  const __amdgpu_buffer_rsrc_t ra = lg_rsrc(p.a, (uint32_t)((size_t)p.M * p.lda * 2));
  const __amdgpu_buffer_rsrc_t rb = WDMA ? lg_rsrc(p.wt, (uint32_t)((size_t)p.N * p.K * 2))
                                         : lg_rsrc(p.qw, (uint32_t)((size_t)(p.K >> 3) * p.N * 4));
  const __amdgpu_buffer_rsrc_t rs = lg_rsrc(p.sc, (uint32_t)((size_t)(p.K / p.group_size) * p.N * 2));
  const __amdgpu_buffer_rsrc_t rz = lg_rsrc(p.qz, (uint32_t)((size_t)(p.K / p.group_size) * (p.N >> 3) * 4));
  const __amdgpu_buffer_rsrc_t rp = lg_rsrc(p.partial, (uint32_t)((size_t)GW * IMAGE));

  // fragment read addresses (LDS byte offsets): k step j of the K tile = 16-byte slot 2 j + kh, XOR f(row), f = (l31 >> 1) & 7
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int fsw = (l31 >> 1) & 7;
  uint32_t a_addr[4], w_addr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int slot = ((2 * j + kh) ^ fsw) << 4;
    a_addr[j] = lds0 + (wm * 128 + l31) * 128 + slot;
    w_addr[j] = lds0 + A_REGION + (wn * 64 + l31) * 128 + slot;
  }
  // where this thread writes its dequantised weights: rows n = 4 lane + q (q = 0..3), slot r = wave; f(n) = (2 lane + (q >> 1)) & 7
  uint32_t ww_addr[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) ww_addr[h] = lds0 + A_REGION + (4 * lane) * 128 + ((wave ^ ((2 * lane + h) & 7)) << 4);
  const int zshift = (lane & 1) * 16;               // this thread's four zero-point nibbles inside its qzeros dword

  bool first_segment = true;
  while (u < u_end) {
    const int tile = u / ktiles_total;
    const int k0 = u - tile * ktiles_total;
    const int k1 = min(ktiles_total, k0 + (u_end - u));
    u += k1 - k0;
    const bool head = k0 == 0, tail = k1 == ktiles_total;
    const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    if (!first_segment) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    first_segment = false;

    // ---- activations: this wave's two LDS-DMA instructions (8 rows x 128 B each) of the half-tiles Ah(h): rows
    // (i >> 3) * 128 + h * 64 + (i & 7) * 8 .. + 8 for instruction i = 2 wave + t ----------------------------------------
    int a_voff[2][2], a_row0[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int i = 2 * wave + t;
        a_row0[h][t] = (i >> 3) * 128 + h * 64 + (i & 7) * 8;
        const int ar = a_row0[h][t] + (lane >> 3);
        a_voff[h][t] = min(m0 + ar, p.M - 1) * p.lda * 2 + (((lane & 7) ^ ((ar >> 1) & 7)) << 4);
      }
    auto stage_a1 = [&](int h, int t, int buf, int kt) {
      const int voff = a_voff[h][t];                // (local copy: see the kernel above on the hipcc host-stub bug)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_ptr)(smem + buf * BUF + a_row0[h][t] * 128), 16, voff, kt * (BK * 2), 0, 0);
    };
    // WDMA: the f16 weight rows n0 + (the same row pattern) of W^T [N, K], same swizzle: the weight region of a buffer is filled
    // by 4 LDS-DMA instructions per wave exactly like the activation region
    int w_dvoff[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int wr_ = a_row0[h][t] + (lane >> 3);
        w_dvoff[h][t] = min(n0 + wr_, p.N - 1) * p.K * 2 + (((lane & 7) ^ ((wr_ >> 1) & 7)) << 4);
      }
    auto stage_w1 = [&](int h, int t, int buf, int kt) {
      const int voff = w_dvoff[h][t];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_ptr)(smem + buf * BUF + A_REGION + a_row0[h][t] * 128), 16, voff, kt * (BK * 2), 0, 0);
    };
    // ---- weights: one packed row piece, its scales and zero points per thread and K tile (inline asm: counted by hand) ---
    // STRIP: the lane's column part of the strip-major address (fixed per segment) and its row multiplier -- the row part is
    // wave-uniform and comes per K tile as one scalar offset + one v_mad (load_w)
    uint32_t w_voff = (uint32_t)((wave * p.N + n0 + 4 * lane) * 4);
    uint32_t w_mult4 = 0;
    if constexpr (STRIP) {
      const int cw = 64 * p.st_np4 + 16 * p.st_rem;
      const int col = n0 + 4 * lane, strip = col / cw, cin = col - strip * cw;
      const bool p4 = cin < 64 * p.st_np4;
      const uint32_t colbase = p4 ? (uint32_t)(cin >> 6) * p.st_nseg * 1024u + (cin & 63)
                                  : (uint32_t)p.st_np4 * p.st_nseg * 1024u + (cin - 64 * p.st_np4);
      w_voff = ((uint32_t)(strip * p.st_nwv) * p.st_wave_dw + colbase) * 4u;
      w_mult4 = p4 ? 16u : 4u * (uint32_t)p.st_rem;
    }
    const uint32_t s_voff = (uint32_t)((n0 + 4 * lane) * 2);
    const uint32_t z_voff = (uint32_t)(((n0 + 4 * lane) >> 3) * 4);
    u32x4 wq = {0, 0, 0, 0};                         // (read-write asm operands: one register set across the loop's back edge)
    u32x2 sq = {0, 0};
    uint32_t zq = 0;
    auto load_w = [&](int kt) {
      uint32_t so_w = (uint32_t)kt * 8u * (uint32_t)p.N * 4u;
      uint32_t voff_w = w_voff;
      if constexpr (STRIP) {
        // packed row kt * 8 + wave = 16 seg + 4 g + u (all wave-uniform: scalar arithmetic, divisions by multiply-shift)
        const uint32_t row = (uint32_t)kt * 8u + (uint32_t)wave, seg = row >> 4;
        const uint32_t kw = (seg * p.st_inv_nseg) >> 16, sg = seg - kw * (uint32_t)p.st_nseg;
        const uint32_t ky = (kw * p.st_inv_nwv) >> 16, wv = kw - ky * (uint32_t)p.st_nwv;
        so_w = (ky * (uint32_t)(p.st_S * p.st_nwv) + wv) * p.st_wave_dw * 4u;
        const uint32_t R = 256u * sg + 64u * (row & 3u) + 16u * ((row >> 2) & 3u);
        voff_w = __builtin_amdgcn_readfirstlane(R) * w_mult4 + w_voff;
      }
      const uint32_t g = (uint32_t)(kt * BK) / (uint32_t)p.group_size;
      const uint32_t so_s = g * (uint32_t)p.N * 2u, so_z = g * (uint32_t)(p.N >> 3) * 4u;
      asm volatile("buffer_load_dwordx4 %0, %3, %4, %7 offen\n\t"
                   "buffer_load_dwordx2 %1, %5, %6, %8 offen\n\t"
                   "buffer_load_dword %2, %9, %10, %11 offen"
                   : "+v"(wq), "+v"(sq), "+v"(zq)
                   : "v"(voff_w), "s"(rb), "v"(s_voff), "s"(rs), "s"(so_w), "s"(so_s), "v"(z_voff), "s"(rz), "s"(so_z)
                   : "memory");
    };
    // (q - z) * s of dword q (0..3) of the thread's packed row piece -> f16 -> row 4 lane + q of the weight region of buffer
    // `buf` (slot = wave)
    auto dequant_write1 = [&](auto Q, int buf) {
      constexpr int q = decltype(Q)::value;
      const uint16_t sb = (uint16_t)(sq[q >> 1] >> ((q & 1) * 16));
      const float sf = p.scale_bf16 ? bf16_bits_to_f32(sb) : f16_bits_to_f32(sb);
      const int z = (int)((zq >> (zshift + 4 * q)) & 0xf) + p.zero_offset;
      const f16 s16 = (f16)sf;
      const f16 a16 = __builtin_bit_cast(f16, (uint16_t)(0x6400 | z));   // 1024 + z
      const f16 b16 = (f16)(float)(-64 - z);
      const u32x4 dv = __builtin_bit_cast(u32x4, dq8_scaled(wq[q], f16x2{a16, a16}, f16x2{b16, b16}, f16x2{s16, s16}));
      lg_lds_write128<(q & 1) * 128 + (q >> 1) * 256>(ww_addr[q >> 1] + buf * BUF, dv);
    };
    using Q0 = std::integral_constant<int, 0>; using Q1 = std::integral_constant<int, 1>;
    using Q2 = std::integral_constant<int, 2>; using Q3 = std::integral_constant<int, 3>;
    auto landed_wq = [&]() { asm volatile("" : "+v"(wq), "+v"(sq), "+v"(zq)); };

    f32x16 acc[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][mb][r] = 0.f;

    const int nk = k1 - k0;
    const int klast = k0 + nk - 1;
    // ---- prologue: K tile 0 complete in buffer 0 (activations by DMA, weights dequantised), Ah0(1) and the packed weights of
    // tile 1 in flight.  Past the segment's last K tile every slot keeps loading a clamped tile nobody reads (one basic
    // block, one vmcnt count: see fp8_gemm_large8_kernel) ---------------------------------------------------------------------
    if constexpr (WDMA) {
      // tile 0 whole in buffer 0 (4 + 4 DMAs), then -- as the steady state leaves them -- Ah0 and the weights of tile 1 in flight
      stage_a1(0, 0, 0, k0); stage_a1(0, 1, 0, k0); stage_a1(1, 0, 0, k0); stage_a1(1, 1, 0, k0);
      stage_w1(0, 0, 0, k0); stage_w1(0, 1, 0, k0); stage_w1(1, 0, 0, k0); stage_w1(1, 1, 0, k0);
      { const int kt1 = min(k0 + 1, klast); stage_a1(0, 0, 1, kt1); stage_a1(0, 1, 1, kt1);
        stage_w1(0, 0, 1, kt1); stage_w1(0, 1, 1, kt1); stage_w1(1, 0, 1, kt1); stage_w1(1, 1, 1, kt1); }
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
    load_w(k0);
    stage_a1(0, 0, 0, k0); stage_a1(0, 1, 0, k0); stage_a1(1, 0, 0, k0); stage_a1(1, 1, 0, k0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    landed_wq();
    dequant_write1(Q0{}, 0); dequant_write1(Q1{}, 0); dequant_write1(Q2{}, 0); dequant_write1(Q3{}, 0);
    { const int kt1 = min(k0 + 1, klast); load_w(kt1); stage_a1(0, 0, 1, kt1); stage_a1(0, 1, 1, kt1); }
    asm volatile("s_waitcnt vmcnt(5)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();       // group 1 runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);

    u32x4 wf[2][4], af[2][4];                        // [n block][k step] / [m block of the current half][k step]
    auto read_w = [&](auto NB) {
      constexpr int nb = decltype(NB)::value;
#pragma unroll
      for (int j = 0; j < 4; ++j) lg_lds_read128<nb * 4096>(wf[nb][j], w_addr[j]);
    };
    auto read_a = [&](auto MH) {
      constexpr int mh = decltype(MH)::value;
#pragma unroll
      for (int j = 0; j < 4; ++j) lg_lds_read128<mh * 8192>(af[0][j], a_addr[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) lg_lds_read128<mh * 8192 + 4096>(af[1][j], a_addr[j]);
    };
    // the phase's 8 MFMAs; `piece(i)` issues the phase's i-th vector-memory instruction(s) behind the (i + 1)-th MFMA pair
    auto mma = [&](int nb, int mh, auto piece) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
          acc[nb][mh * 2 + mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[nb][j]), __builtin_bit_cast(f16x8, af[mb][j]),
                                                                        acc[nb][mh * 2 + mb], 0, 0, 0);
        if (j < 3) {
          __builtin_amdgcn_sched_barrier(0);
          piece(j);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
    };
    auto landed_w = [&](int nb) {
#pragma unroll
      for (int j = 0; j < 4; ++j) lg_touch(wf[nb][j]);
    };
    auto landed_a = [&]() {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int j = 0; j < 4; ++j) lg_touch(af[mb][j]);
    };
#define LG_BAR()  do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define LG_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

    for (int t = 0; t < nk; ++t) {
      const int cur = t & 1;
      const int kt1 = min(k0 + t + 1, klast), kt2 = min(k0 + t + 2, klast);
      // ---- P1 -------------------------------------------------------------------------------------------------------
      read_w(std::integral_constant<int, 0>{});
      read_a(std::integral_constant<int, 0>{});
      LG_BAR();
      LG_LGKM0(); landed_w(0); landed_a();
      mma(0, 0, [&](int i) { if (i < 2) stage_a1(1, i, cur ^ 1, kt1); });
      LG_BAR();
      // ---- P2: the packed weights of tile t + 1 (requested behind P3 of the previous tile) have landed; its weight tile is
      // built in the read blocks of P2 (1 dword), P3 (1) and P4 (2, no fragment reads there) while the SIMD's other wave is
      // in its MFMA cluster ------------------------------------------------------------------------------------------------
      read_w(std::integral_constant<int, 1>{});
      if constexpr (!WDMA) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        landed_wq();
        dequant_write1(Q0{}, cur ^ 1);
      }
      LG_BAR();
      LG_LGKM0(); landed_w(1);
      mma(1, 0, [&](int) {});
      LG_BAR();
      // ---- P3 -------------------------------------------------------------------------------------------------------
      // (WDMA: from here on every wave -- group 1 runs one barrier behind -- has finished its LDS reads of this tile's weight
      //  region (P1: nb 0, P2: nb 1): the weights of tile t + 2 go into the same buffer, 1 piece here, 3 behind P4's MFMAs)
      read_a(std::integral_constant<int, 1>{});
      if constexpr (!WDMA) dequant_write1(Q1{}, cur ^ 1);
      LG_BAR();
      LG_LGKM0(); landed_a();
      mma(1, 1, [&](int i) { if (i < 2) stage_a1(0, i, cur, kt2); else if (WDMA) stage_w1(0, 0, cur, kt2); });
      LG_BAR();
      // ---- P4 -------------------------------------------------------------------------------------------------------
      // everything older than P3's two DMAs has landed (the activations of tile t + 1 are whole); this wave's weight rows of
      // tile t + 1 are in the LDS before the barrier: the first reader is P1 of tile t + 1.  The request for tile t + 2's
      // packed weights follows in the MFMA shadow (wq is free again).
      if constexpr (WDMA) {
        // all but P3's three DMAs (Ah0 x 2, one weight piece of tile t + 2) have landed: tile t + 1 is whole
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        LG_BAR();
        mma(0, 1, [&](int i) { if (i == 0) stage_w1(0, 1, cur, kt2); else if (i == 1) stage_w1(1, 0, cur, kt2); else stage_w1(1, 1, cur, kt2); });
      } else {
      dequant_write1(Q2{}, cur ^ 1);
      dequant_write1(Q3{}, cur ^ 1);
      asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      LG_BAR();
      mma(0, 1, [&](int i) { if (i == 0) load_w(kt2); });
      }
      LG_BAR();
#pragma unroll
      for (int j = 0; j < 4; ++j) { a_addr[j] += cur ? -BUF : BUF; w_addr[j] += cur ? -BUF : BUF; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the over-run loads have landed before anything else uses the LDS
    asm volatile("" : "+v"(wq), "+v"(sq), "+v"(zq));
    if (wm == 0) __builtin_amdgcn_s_barrier();       // catch up with group 1: every wave is done with the stage buffers
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (nk & 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { a_addr[j] -= BUF; w_addr[j] -= BUF; }
    }
#undef LG_BAR
#undef LG_LGKM0
    wna16_large_finish<NWAVE, false>(p, acc, smem, rp, head, tail, w, GW, U, tile, ktiles_total, m0, n0, wave, wm, wn, lane);
  }
}

// Two-pass form, pass 1: q_weight [K/8, N] (exllama order) -> f16 W^T [N, K] with the numerics of the in-loop dequantisation
// (dq8_scaled: (q - z) * s in f16, q_gemm.cu:1394-1434) -- the reference's own large-M path reconstructs the matrix once per
// call too (q_gemm.cu:1529-1544).  One workgroup = 256 columns x 64 k, the load pattern of the eight-phase kernel: thread
// (lane: columns 4 lane .. + 3, wave: packed row) reads ONE 16-byte piece, dequantises its 4 x 8 weights into an LDS tile
// [column][64 k], and the tile leaves as whole 128-byte lines of W^T (8 lanes per line).  (First version: every thread wrote
// its own column's 16-byte pieces straight to W^T -- 64 partial lines per store instruction: 2.1 TB/s, 139 us on gate_up.)
// STRIP: qw is the strip-major copy (Wna16LargeParams::strip) -- the same 16-byte piece from its strip-major address.
template <bool STRIP>
__global__ __launch_bounds__(512) void wna16_dequant_t_kernel(const uint32_t* __restrict__ qw, const uint32_t* __restrict__ qz,
                                                             const uint16_t* __restrict__ sc, uint16_t* __restrict__ wt, int N, int K,
                                                             int group_size, int zero_offset, int scale_bf16, Wna16LargeParams sp) {
  constexpr int PITCH = 128 + 16;                       // bytes per LDS row (64 k of one column), padded
  __shared__ __attribute__((aligned(16))) unsigned char tile[256 * PITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 256, kt = blockIdx.y;
  const int nq = n0 + 4 * lane;
  const int g = (kt * 64) / group_size;
  const u32x4 wq = *reinterpret_cast<const u32x4*>(qw + (STRIP ? (size_t)lg_strip_dw(sp, kt * 8 + wave, nq) : (size_t)(kt * 8 + wave) * N + nq));
  const u32x2 sq = *reinterpret_cast<const u32x2*>(sc + (size_t)g * N + nq);
  const uint32_t zq = qz[(size_t)g * (N >> 3) + (nq >> 3)] >> ((lane & 1) * 16);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint16_t sb = (uint16_t)(sq[q >> 1] >> ((q & 1) * 16));
    const float sf = scale_bf16 ? bf16_bits_to_f32(sb) : f16_bits_to_f32(sb);
    const int z = (int)((zq >> (4 * q)) & 0xf) + zero_offset;
    const f16 s16 = (f16)sf;
    const f16 a16 = __builtin_bit_cast(f16, (uint16_t)(0x6400 | z));   // 1024 + z
    const f16 b16 = (f16)(float)(-64 - z);
    *reinterpret_cast<u32x4*>(tile + (4 * lane + q) * PITCH + wave * 16) =
        __builtin_bit_cast(u32x4, dq8_scaled(wq[q], f16x2{a16, a16}, f16x2{b16, b16}, f16x2{s16, s16}));
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = it * 512 + threadIdx.x, row = idx >> 3, ch = idx & 7;
    *reinterpret_cast<u32x4*>(wt + (size_t)(n0 + row) * K + kt * 64 + ch * 8) = *reinterpret_cast<const u32x4*>(tile + row * PITCH + ch * 16);
  }
}

// partial [S][M*N] fp32 -> c [M*N] f16 / bf16 (fixed summation order: deterministic)
__global__ void splitk_reduce_large_kernel(const float* __restrict__ partial, uint16_t* __restrict__ c, int64_t mn, int S,
                                           int out_bf16, const float* __restrict__ w_scales = nullptr, int w_per_channel = 0,
                                           const uint16_t* __restrict__ bias = nullptr, int N = 1) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= mn) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(partial + i);
  for (int k = 1; k < S; ++k) s += *reinterpret_cast<const f32x4*>(partial + (size_t)k * mn + i);
  if (w_scales) {            // W8A16 form: the epilogue the single-slice kernel applies itself
    const int col = (int)(i % N);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] *= w_scales[w_per_channel ? col + j : 0];
      if (bias) s[j] += out_bf16 ? bf16_bits_to_f32(bias[col + j]) : f16_bits_to_f32(bias[col + j]);
    }
  }
  u16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = out_bf16 ? f32_to_bf16_bits(s[j]) : f32_to_f16_bits(s[j]);
  *reinterpret_cast<u16x4*>(c + i) = o;
}

// bf16 [M, K] (row stride lda) -> f16 [M, K] contiguous, saturating (see bf16_bits_to_f16_bits_sat)
__global__ void bf16_to_f16_rows_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int M, int K, int lda) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i >= (int64_t)M * K) return;
  const int row = (int)(i / K), col = (int)(i % K);
  u16x8 v = *reinterpret_cast<const u16x8*>(in + (size_t)row * lda + col);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = bf16_bits_to_f16_bits_sat(v[j]);
  *reinterpret_cast<u16x8*>(out + i) = v;
}

template <int WM, int WN, int STAGES, bool WFP8>
static int launch_large_s(const Wna16LargeParams& p, hipStream_t st) {
  constexpr int BM = 128 * WM, BN = 64 * WN;
  Wna16LargeParams q = p;
  q.tiles_m = (p.M + BM - 1) / BM;
  q.tiles_n = p.N / BN;
  const int G = WFP8 ? 0 : (p.streamk ? p.K / p.group_size : p.K / p.ksplit / p.group_size);   // groups a segment can touch
  size_t lds = STAGES * ((size_t)BM * 64 * 2 + (WFP8 ? (size_t)BN * 64 : (size_t)8 * BN * 4)) + (size_t)G * BN * 2 + (size_t)G * (BN / 8) * 4;
  if (lds < (size_t)WM * WN * 16384) lds = (size_t)WM * WN * 16384;   // the epilogue's wave-private transpose regions
  if (lds > 160 * 1024) {
    set_error("wna16_gemm_large: %zu bytes of LDS needed (K=%d, group %d)", lds, p.K, p.group_size);
    return APHRO_ERR_INVALID;
  }
  static bool attr_set_dev[APHRO_MAX_DEVICES] = {}; bool& attr_set = attr_set_dev[device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)wna16_gemm_large_kernel<WM, WN, STAGES, WFP8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess) {
      set_error("wna16_gemm_large: cannot raise the dynamic LDS limit");
      return APHRO_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((wna16_gemm_large_kernel<WM, WN, STAGES, WFP8>), p.streamk ? dim3(p.grid) : dim3(q.tiles_m * q.tiles_n, q.ksplit), dim3(WM * WN * 64), lds, st, q);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

template <bool WDMA, bool STRIP = false>
static int launch_large8_t(const Wna16LargeParams& p, hipStream_t st) {
  Wna16LargeParams q = p;
  q.tiles_m = (p.M + 255) / 256;
  q.tiles_n = p.N / 256;
  constexpr size_t lds = 128 * 1024;
  static bool attr_set_dev[APHRO_MAX_DEVICES][2] = {};      // (one array per template instance)
  bool& attr_set = attr_set_dev[device_slot()][p.out_bf16 ? 1 : 0];
  const void* fn = p.out_bf16 ? (const void*)wna16_gemm_large8_kernel<true, WDMA, STRIP> : (const void*)wna16_gemm_large8_kernel<false, WDMA, STRIP>;
  if (!attr_set) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      set_error("wna16_gemm_large: cannot raise the dynamic LDS limit");
      return APHRO_ERR_LAUNCH;
    }
    attr_set = true;
  }
  if (p.out_bf16) hipLaunchKernelGGL((wna16_gemm_large8_kernel<true, WDMA, STRIP>), dim3(p.grid), dim3(512), lds, st, q);
  else hipLaunchKernelGGL((wna16_gemm_large8_kernel<false, WDMA, STRIP>), dim3(p.grid), dim3(512), lds, st, q);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
static int launch_large8(const Wna16LargeParams& p, hipStream_t st) {
  if (p.wt != nullptr) return launch_large8_t<true>(p, st);
  return p.strip ? launch_large8_t<false, true>(p, st) : launch_large8_t<false>(p, st);
}

// three LDS stages (loads two K tiles ahead, one raw barrier per tile) when the group metadata leaves room for them
template <int WM, int WN, bool WFP8 = false>
static int launch_large(const Wna16LargeParams& p, hipStream_t st) {
  constexpr int BM = 128 * WM, BN = 64 * WN;
  const int G = WFP8 ? 0 : (p.streamk ? p.K / p.group_size : p.K / p.ksplit / p.group_size);
  const size_t lds3 = 3 * ((size_t)BM * 64 * 2 + (WFP8 ? (size_t)BN * 64 : (size_t)8 * BN * 4)) + (size_t)G * BN * 2 + (size_t)G * (BN / 8) * 4;
  const int force = APHRO_LAB_ENV_INT("APHRO_WNA16_LARGE_STAGES", 0);
  if ((lds3 <= 160 * 1024 && force != 2) || force == 3) return launch_large_s<WM, WN, 3, WFP8>(p, st);
  return launch_large_s<WM, WN, 2, WFP8>(p, st);
}

}  // namespace aphro

using namespace aphro;

struct LargePlan { int wm, wn, ksplit, streamk, grid; };

static int large_cu_count() { return device_cu_count(); }

constexpr size_t LARGE_FLAG_BYTES = 4096;

// Stream-K (one persistent workgroup per CU, see fp8_gemm_large.hip) when the biggest tile the shape allows still gives
// >= 128 tiles.  Below that: one workgroup per tile, narrower tiles and a split of K into up to 8 fp32 slabs (summed in
// fixed order by splitk_reduce_large_kernel) so that ~256+ workgroups exist.
static LargePlan large_plan(int64_t M, int64_t N, int64_t K, int64_t gs) {
  LargePlan pl;
  pl.wm = M > 128 ? 2 : 1;
  const int64_t rows = (M + 128 * pl.wm - 1) / (128 * pl.wm);
  const int mode = APHRO_LAB_ENV_INT("APHRO_WNA16_LARGE_STREAMK", -1);
  const int64_t big_tiles = N % 256 == 0 ? rows * (N / 256) : rows * (N / 128);
  pl.streamk = (mode >= 0 ? mode : (big_tiles >= 128)) && device_coresident_cu_count() > 0;
  pl.ksplit = 1;
  pl.grid = 0;
  if (pl.streamk) {
    pl.wn = N % 256 == 0 ? 4 : 2;
    pl.grid = large_cu_count();
    return pl;
  }
  pl.wn = (N % 256 == 0 && rows * (N / 256) >= 200) ? 4 : 2;
  { const int v = APHRO_LAB_ENV_INT("APHRO_WNA16_LARGE_WN", 0); if (v == 2 || (v == 4 && N % 256 == 0)) pl.wn = v; }
  const int64_t tiles = rows * (N / (64 * pl.wn));
  const int64_t unit = gs > 64 ? gs : 64;           // a K range holds whole groups and whole K tiles
  // K slices until ~800 waves exist (a 2-wave workgroup fills half a CU's SIMDs: 400 of those), as long as the fp32
  // slabs stay small beside the weights (measured, tools/mid_gemm_sweep.py: past ~36 MB the reduce pass costs more than
  // the extra workgroups buy)
  const int64_t want = pl.wm * pl.wn <= 2 ? 400 : 200;
  const int64_t slab = M * N * 4;
  for (int s = 2; s <= 16; ++s) {
    if (tiles * pl.ksplit >= want) break;
    if (K % (s * unit) == 0 && K / s >= 512 && s * slab <= (36ll << 20)) pl.ksplit = s;
  }
  { const int s = APHRO_LAB_ENV_INT("APHRO_WNA16_LARGE_KSPLIT", 0); if (s >= 1 && K % (s * unit) == 0) pl.ksplit = s; }
  return pl;
}

// Two-pass form (dequantise-transpose once per call, then the eight-phase schedule with the weights by LDS-DMA): where every
// weight would otherwise be dequantised by >= TWO_PASS_ROW_TILES row-tile workgroups and the shape runs the eight-phase
// stream-K plan.  Costs K x N f16 of workspace per call.  APHRO_WNA16_LARGE_TWO_PASS=0/1 forces (tests compare the two forms' bits).
constexpr int64_t TWO_PASS_MIN_M = 6144;      // (M = 4096: the fused form is 2-4 % faster on all four Llama-3-8B shapes; 8192: 4-14 % slower)
static bool large_two_pass(const LargePlan& pl, int64_t M, int64_t N, int64_t K, int64_t gs) {
  const int force = knobs().wna16_large_two_pass;
  const int eight = knobs().wna16_large_8phase >= 0 ? knobs().wna16_large_8phase : (K >= 2048 ? 1 : 0);
  const bool can = pl.wm == 2 && pl.wn == 4 && pl.streamk && eight && K % 64 == 0 && N % 256 == 0 && gs % 64 == 0 &&
                   (size_t)N * K * 2 < 0xffffffffull;
  if (force >= 0) return can && force != 0;
  return can && M >= TWO_PASS_MIN_M;
}

static size_t large_scratch_bytes(const LargePlan& pl, int64_t M, int64_t N) {
  if (pl.streamk) return LARGE_FLAG_BYTES + (size_t)pl.grid * pl.wm * pl.wn * 32 * 1024;
  return pl.ksplit > 1 ? (size_t)pl.ksplit * M * N * sizeof(float) : 0;
}

// stream-K: clear the flag words (a memset node in front of the kernel: replays in HIP graphs), point the kernel at them
static int large_bind_scratch(Wna16LargeParams& p, const LargePlan& pl, char* ws, hipStream_t st) {
  p.streamk = pl.streamk; p.grid = pl.grid; p.ksplit = pl.ksplit; p.flags = nullptr; p.partial = (float*)ws;
  if (pl.streamk) {
    p.flags = (unsigned*)ws;
    p.partial = (float*)(ws + LARGE_FLAG_BYTES);
    if (hipMemsetAsync(p.flags, 0, LARGE_FLAG_BYTES, st) != hipSuccess) {
      set_error("wna16_gemm_large: cannot clear the stream-K flags");
      return APHRO_ERR_LAUNCH;
    }
  }
  return APHRO_OK;
}

extern "C" int aphro_wna16_strip_geometry(int64_t M, int64_t N, int64_t K, int64_t groups, int* geom);      // wna16_gemm_resident.hip

// (the eight-phase schedule on the 256 x 256 stream-K tile from 32 K tiles per output tile up; APHRO_WNA16_LARGE_8PHASE=0/1 forces)
static bool large_eight(const LargePlan& pl, int64_t K) {
  const int eight = knobs().wna16_large_8phase >= 0 ? knobs().wna16_large_8phase : (K >= 2048 ? 1 : 0);
  return pl.wm == 2 && pl.wn == 4 && pl.streamk && eight;
}

static size_t large_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t groups, int dtype, int strip_m) {
  if (groups <= 0 || K % groups != 0) return 0;
  const LargePlan pl = large_plan(M, N, K, K / groups);
  size_t b = dtype == APHRO_BF16 ? ((size_t)M * K * 2 + 255) / 256 * 256 : 0;
  if (large_two_pass(pl, M, N, K, K / groups)) b += ((size_t)N * K * 2 + 255) / 256 * 256;      // f16 W^T of the two-pass form
  (void)strip_m;              // (every plan addresses the strip-major copy in place: nothing is rebuilt)
  return b + large_scratch_bytes(pl, M, N);
}

// Bytes of scratch aphro_wna16_gemm_large needs: the f16 copy of bf16 activations + the fp32 split-K slabs.
extern "C" size_t aphro_wna16_gemm_large_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t groups, int dtype) {
  return large_workspace_bytes(M, N, K, groups, dtype, 0);
}
// ... aphro_wna16_gemm_large_strip needs (strip_m: the M class the strip-major copy was laid out for, as given to
// aphro_wna16_strip_relayout).
extern "C" size_t aphro_wna16_gemm_large_strip_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t groups, int dtype, int64_t strip_m) {
  return large_workspace_bytes(M, N, K, groups, dtype, (int)strip_m);
}

// c[M, N] = a[M, K] . dequant(q_weight[K/8, N] exllama order, qzeros[G, N/8], scales[G, N]); any M, meant for M > 64.
// N % 128 == 0, K % 64 == 0, group size a multiple of 64.  dtype f16 / bf16 (bf16 activations are widened to f16
// with saturation, scales and output stay bf16).  Act-order: pass the activations already gathered (a[:, perm]).
static int wna16_gemm_large_impl(const void* a, const uint32_t* q_weight, const uint32_t* qzeros, const void* scales,
                                 void* c, void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K,
                                 int64_t groups, int64_t lda, int zero_offset, int dtype, int silu, void* stream, int strip_m = 0) {
  hipStream_t st = (hipStream_t)stream;
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "wna16_gemm_large: dtype must be f16 or bf16");
  APHRO_CHECK(groups > 0 && K % groups == 0, "wna16_gemm_large: K=%ld not divisible by groups=%ld", (long)K, (long)groups);
  const int64_t gs = K / groups;
  APHRO_CHECK(K % 64 == 0 && gs % 64 == 0, "wna16_gemm_large: K and the group size must be multiples of 64 (K=%ld, g=%ld)", (long)K, (long)gs);
  APHRO_CHECK(N % 128 == 0, "wna16_gemm_large: N=%ld must be a multiple of 128", (long)N);
  APHRO_CHECK(lda % 8 == 0 && ((uintptr_t)a % 16) == 0, "wna16_gemm_large: a must be 16-byte aligned with lda %% 8 == 0");
  APHRO_CHECK((size_t)M * lda * 2 < 0xffffffffull && (size_t)(K / 8) * N * 4 < 0xffffffffull, "wna16_gemm_large: operand exceeds 4 GiB");
  if (M == 0) return APHRO_OK;
  const LargePlan pl = large_plan(M, N, K, gs);
  APHRO_CHECK(!silu || pl.streamk || pl.ksplit == 1, "wna16_gemm_large_silu: shape M=%ld N=%ld K=%ld is K-sliced (no SiluAndMul epilogue)",
              (long)M, (long)N, (long)K);
  const size_t need = large_workspace_bytes(M, N, K, groups, dtype, strip_m);
  if (need > 0 && (workspace == nullptr || workspace_bytes < need)) {
    set_error("wna16_gemm_large: workspace %zu < %zu bytes", workspace_bytes, need);
    return APHRO_ERR_WORKSPACE;
  }
  Wna16LargeParams p;
  p.silu = silu;
  p.a = (const uint16_t*)a; p.lda = (int)lda;
  char* ws = (char*)workspace;
  if (dtype == APHRO_BF16) {
    const int64_t n8 = M * K / 8;
    hipLaunchKernelGGL(bf16_to_f16_rows_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, (const uint16_t*)a,
                       (uint16_t*)ws, (int)M, (int)K, (int)lda);
    APHRO_LAUNCH_CHECK();
    p.a = (const uint16_t*)ws; p.lda = (int)K;
    ws += ((size_t)M * K * 2 + 255) / 256 * 256;
  }
  p.qw = q_weight; p.qz = qzeros; p.sc = (const uint16_t*)scales; p.c = (uint16_t*)c;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.group_size = (int)gs; p.zero_offset = zero_offset;
  p.out_bf16 = dtype == APHRO_BF16; p.scale_bf16 = dtype == APHRO_BF16;
  p.tiles_m = p.tiles_n = 0;
  p.w8 = nullptr; p.w_scales = nullptr; p.w_per_channel = 0; p.bias = nullptr;
  p.wt = nullptr;
  p.strip = 0;
  p.st_nwv = p.st_nseg = p.st_np4 = p.st_rem = p.st_S = 0; p.st_wave_dw = p.st_inv_nseg = p.st_inv_nwv = 0;
  if (strip_m > 0) {
    // q_weight is the strip-major copy of the <= 32-row decode kernels (the only resident one)
    int geom[5];
    APHRO_CHECK(aphro_wna16_strip_geometry(strip_m, N, K, groups, geom) == 1,
                "wna16_gemm_large_strip: no strip-major form for M class %d, N=%ld, K=%ld, groups=%ld", strip_m, (long)N, (long)K, (long)groups);
    p.strip = 1;
    p.st_nwv = geom[0]; p.st_nseg = geom[1]; p.st_np4 = geom[2]; p.st_rem = geom[3];
    const int cw = 64 * p.st_np4 + 16 * p.st_rem;
    p.st_S = (int)(N / cw);
    p.st_wave_dw = (uint32_t)(p.st_nseg * 256 * (4 * p.st_np4 + p.st_rem));
    p.st_inv_nseg = (65536u + p.st_nseg - 1) / p.st_nseg;
    p.st_inv_nwv = (65536u + p.st_nwv - 1) / p.st_nwv;
    const uint32_t segs = (uint32_t)(K / 128) + 1;       // (+ 1: a clamped tile past the range is never formed, the bound is slack)
    for (uint32_t x = 0; x < segs; ++x)
      APHRO_CHECK(((x * p.st_inv_nseg) >> 16) == x / p.st_nseg && ((x * p.st_inv_nwv) >> 16) == x / p.st_nwv,
                  "wna16_gemm_large_strip: multiply-shift division fails at %u (nseg %d, nwv %d)", x, p.st_nseg, p.st_nwv);
  }
  if (large_two_pass(pl, M, N, K, gs)) {
    // pass 1: the weights dequantised once into f16 W^T [N, K] (same numerics as the in-loop dequantisation: same bits out)
    const dim3 dgrid((unsigned)(N / 256), (unsigned)(K / 64));
    if (p.strip)
      hipLaunchKernelGGL(wna16_dequant_t_kernel<true>, dgrid, dim3(512), 0, st, q_weight, qzeros,
                         (const uint16_t*)scales, (uint16_t*)ws, (int)N, (int)K, (int)gs, zero_offset, p.scale_bf16, p);
    else
      hipLaunchKernelGGL(wna16_dequant_t_kernel<false>, dgrid, dim3(512), 0, st, q_weight, qzeros,
                         (const uint16_t*)scales, (uint16_t*)ws, (int)N, (int)K, (int)gs, zero_offset, p.scale_bf16, p);
    APHRO_LAUNCH_CHECK();
    p.strip = 0;                                    // (pass 2 reads W^T)
    p.wt = (const uint16_t*)ws;
    ws += ((size_t)N * K * 2 + 255) / 256 * 256;
  }
  if (int rcb = large_bind_scratch(p, pl, ws, st)) return rcb;
  int rc;
  // eight-phase schedule on the 256 x 256 stream-K tile from 32 K tiles per output tile up (APHRO_WNA16_LARGE_8PHASE=0/1 forces)
  if (large_eight(pl, K)) rc = launch_large8(p, st);
  else if (pl.wm == 2) rc = pl.wn == 4 ? launch_large<2, 4>(p, st) : launch_large<2, 2>(p, st);
  else rc = pl.wn == 4 ? launch_large<1, 4>(p, st) : launch_large<1, 2>(p, st);
  if (rc != APHRO_OK) return rc;
  if (!pl.streamk && pl.ksplit > 1) {
    const int64_t mn = M * N;
    hipLaunchKernelGGL(splitk_reduce_large_kernel, dim3((unsigned)((mn / 4 + 255) / 256)), dim3(256), 0, st, p.partial,
                       (uint16_t*)c, mn, pl.ksplit, p.out_bf16);
    APHRO_LAUNCH_CHECK();
  }
  return APHRO_OK;
}

extern "C" int aphro_wna16_gemm_large(const void* a, const uint32_t* q_weight, const uint32_t* qzeros, const void* scales,
                                      void* c, void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K,
                                      int64_t groups, int64_t lda, int zero_offset, int dtype, void* stream) {
  return wna16_gemm_large_impl(a, q_weight, qzeros, scales, c, workspace, workspace_bytes, M, N, K, groups, lda, zero_offset, dtype, 0, stream);
}

// The same GEMM on a gate_up matrix with interleaved (gate_j, up_j) columns, SiluAndMul in the epilogue: act [M, N / 2] =
// silu_and_mul(a . dequant(W)) with the GEMM result rounded to the dtype first (the bits of aphro_wna16_gemm_large followed by
// aphro_silu_and_mul_interleaved).  1 if the shape is served (not K-sliced), else 0: aphro_wna16_gemm_large_silu_supported.
// aphro_wna16_gemm_large / aphro_wna16_gemm_large_silu (silu != 0) on the STRIP-MAJOR copy of the weights
// (aphro_wna16_strip_relayout for the M class strip_m, normally 32): every plan reads it in place -- same loads, other
// addresses, same bits.  For a model that keeps one
// copy of each matrix resident (the one its decode kernels stream).
extern "C" int aphro_wna16_gemm_large_strip(const void* a, const uint32_t* q_weight_strip, const uint32_t* qzeros, const void* scales,
                                            void* c, void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K,
                                            int64_t groups, int64_t lda, int zero_offset, int dtype, int silu, int64_t strip_m,
                                            void* stream) {
  APHRO_CHECK(strip_m >= 1 && strip_m <= 64, "wna16_gemm_large_strip: strip_m=%ld", (long)strip_m);
  return wna16_gemm_large_impl(a, q_weight_strip, qzeros, scales, c, workspace, workspace_bytes, M, N, K, groups, lda, zero_offset, dtype,
                               silu ? 1 : 0, stream, (int)strip_m);
}

extern "C" int aphro_wna16_gemm_large_silu_supported(int64_t M, int64_t N, int64_t K, int64_t groups) {
  if (groups <= 0 || K % groups != 0 || K % 64 != 0 || (K / groups) % 64 != 0 || N % 128 != 0 || M <= 0) return 0;
  const LargePlan pl = large_plan(M, N, K, K / groups);
  return (pl.streamk || pl.ksplit == 1) ? 1 : 0;
}
extern "C" int aphro_wna16_gemm_large_silu(const void* a, const uint32_t* q_weight, const uint32_t* qzeros, const void* scales,
                                           void* act, void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K,
                                           int64_t groups, int64_t lda, int zero_offset, int dtype, void* stream) {
  return wna16_gemm_large_impl(a, q_weight, qzeros, scales, act, workspace, workspace_bytes, M, N, K, groups, lda, zero_offset, dtype, 1, stream);
}

// W8A16 for prefill-sized M -- the role of `_C::fp8_marlin_gemm` (kernels/torch_bindings.cpp:218-222,
// quantization/fp8/fp8_marlin.cu:1212) above 64 rows: c[M, N] = a[M, K] . (f16(w[N, K]) * w_scales[n]) + bias.  The same tile
// machine as the int4 kernel: e4m3 weights [N, K] go direct-to-LDS, are widened to f16 in registers (exact) and the scale
// is applied once per output in the epilogue.  N % 128 == 0, K % 64 == 0.  workspace: the f16 copy of bf16 activations
// + the fp32 split-K slabs of small grids.
extern "C" size_t aphro_fp8_w8a16_gemm_large_workspace_bytes(int64_t M, int64_t N, int64_t K, int dtype) {
  const LargePlan pl = large_plan(M, N, K, 64);
  size_t b = dtype == APHRO_BF16 ? ((size_t)M * K * 2 + 255) / 256 * 256 : 0;
  return b + large_scratch_bytes(pl, M, N);
}

extern "C" int aphro_fp8_w8a16_gemm_large(void* out, const void* a, const void* w, const float* w_scales, const void* bias,
                                          void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K, int64_t lda,
                                          int w_scale_per_channel, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "fp8_w8a16_gemm_large: dtype must be f16 or bf16");
  APHRO_CHECK(N % 128 == 0 && K % 64 == 0, "fp8_w8a16_gemm_large: N=%ld must be a multiple of 128, K=%ld of 64", (long)N, (long)K);
  APHRO_CHECK(lda % 8 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)w % 16) == 0, "fp8_w8a16_gemm_large: 16-byte alignment");
  APHRO_CHECK((size_t)M * lda * 2 < 0xffffffffull && (size_t)N * K < 0xffffffffull, "fp8_w8a16_gemm_large: operand exceeds 4 GiB");
  if (M == 0) return APHRO_OK;
  const size_t need = aphro_fp8_w8a16_gemm_large_workspace_bytes(M, N, K, dtype);
  if (need > 0 && (workspace == nullptr || workspace_bytes < need)) {
    set_error("fp8_w8a16_gemm_large: workspace %zu < %zu bytes", workspace_bytes, need);
    return APHRO_ERR_WORKSPACE;
  }
  Wna16LargeParams p;
  p.a = (const uint16_t*)a; p.lda = (int)lda;
  if (dtype == APHRO_BF16) {
    const int64_t n8 = M * K / 8;
    hipLaunchKernelGGL(bf16_to_f16_rows_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, (const uint16_t*)a,
                       (uint16_t*)workspace, (int)M, (int)K, (int)lda);
    APHRO_LAUNCH_CHECK();
    p.a = (const uint16_t*)workspace; p.lda = (int)K;
  }
  p.silu = 0; p.wt = nullptr; p.strip = 0;
  p.qw = nullptr; p.qz = nullptr; p.sc = nullptr; p.c = (uint16_t*)out;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.group_size = 64; p.zero_offset = 0;
  p.out_bf16 = dtype == APHRO_BF16; p.scale_bf16 = 0;
  const LargePlan pl = large_plan(M, N, K, 64);
  p.tiles_m = p.tiles_n = 0;
  if (int rcb = large_bind_scratch(p, pl, (char*)workspace + (dtype == APHRO_BF16 ? ((size_t)M * K * 2 + 255) / 256 * 256 : 0), st)) return rcb;
  p.w8 = (const uint8_t*)w; p.w_scales = w_scales; p.w_per_channel = w_scale_per_channel; p.bias = (const uint16_t*)bias;
  int rc;
  if (pl.wm == 2) rc = pl.wn == 4 ? launch_large<2, 4, true>(p, st) : launch_large<2, 2, true>(p, st);
  else rc = pl.wn == 4 ? launch_large<1, 4, true>(p, st) : launch_large<1, 2, true>(p, st);
  if (rc != APHRO_OK) return rc;
  if (!pl.streamk && pl.ksplit > 1) {
    const int64_t mn = M * N;
    hipLaunchKernelGGL(splitk_reduce_large_kernel, dim3((unsigned)((mn / 4 + 255) / 256)), dim3(256), 0, st, p.partial,
                       (uint16_t*)out, mn, pl.ksplit, p.out_bf16, w_scales, w_scale_per_channel, (const uint16_t*)bias, (int)N);
    APHRO_LAUNCH_CHECK();
  }
  return APHRO_OK;
}
