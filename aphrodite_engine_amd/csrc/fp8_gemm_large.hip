// W8A8 FP8 (OCP e4m3 x e4m3) scaled GEMM for PREFILL-sized M on gfx950 -- the cutlass_scaled_mm role above 64 rows
// (kernels/quantization/cutlass_w8a8/scaled_mm_entry.cu:32-81; on ROCm the reference falls back to
// torch._scaled_mm, modeling/layers/quantization/utils/w8a8_utils.py:130-183).  MFMA bound:
//   * out[m][n] = a_scale[m] * (b_scale[n] * sum_k A[m][k] W[n][k]) + bias[n], both operands K-contiguous in HBM
//     ([M, K] activations, [N, K] = the column-major [K, N] weight the op receives), so neither needs a transpose:
//     v_mfma_scale_f32_32x32x64_f8f6f4 (unit E8M0 scales: the double-rate fp8 path of CDNA4) takes 32 consecutive
//     k bytes of one row per lane for BOTH operands.  D^T[n][m] = W[n][k] . A^T[k][m]: the weights are the A operand,
//     the activations the B operand, like wna16_gemm_large.hip (a lane then owns 4 consecutive n of one output row).
//   * both tiles go global -> LDS with direct-to-LDS loads (no VGPR round trip), 128-byte rows XOR-swizzled on the
//     SOURCE address, read back with conflict-free ds_read_b128 (two per operand fragment).
//   * 2 LDS stages of 64 KiB (256 x 256 x 128 tile) or 3 stages of 48 KiB (narrower tiles): the next K tile's loads
//     are in flight under the current tile's 16 MFMAs per wave.
//   * STREAM-K: one persistent workgroup per CU; the (tile, K tile) units are dealt out in equal contiguous ranges, so
//     (a) there is no last partial round of tiles, and (b) the workgroups reach their tile boundaries at different
//     times -- measured on the one-workgroup-per-tile form, all 256 CUs storing their 128 KiB of C at the same moment
//     ran at 1.7 TB/s with every MFMA idle: a quarter of the kernel at K = 4096.  A tile whose K range is cut between
//     workgroups is finished by the one that owns its k = 0 end: the others publish their fp32 accumulators
//     (write-through stores + one flag word each, no fences) and the owner adds them in workgroup order
//     (deterministic).  Small grids (< 128 tiles) keep the one-workgroup-per-(tile, K slice) form with fp32 slabs.
//     Co-residency: an owner only waits for its successor, which publishes as soon as it has started -- progress is
//     guaranteed while more than half of the CUs are available to this launch; the wait is bounded (5 s, then trap).
//   * results leave through a wave-private LDS transpose as full 128-byte row segments (epi_put / epi_flush, common.h).
// Workgroup = WN x WM waves, wave tile = 128 rows (m) x 64 columns (n), K tile = 128.
//
// Round 5: the 256 x 256 tile runs the EIGHT-PHASE PING-PONG schedule (fp8_gemm_large8_kernel below; cdna_hip_programming.md
// "The 256^2 8-phase template"): the K tile is cut into four half-tiles (two of the activations, two of the weights), each
// compute phase = one C quadrant of a wave (64 m x 32 n x 128 k = 4 MFMAs, 256 matrix cycles), the two waves of every SIMD
// run one barrier apart so that one is in its MFMA cluster while the other issues its fragment reads and LDS-DMA, and the
// direct-to-LDS loads stay in flight across the barriers behind COUNTED vmcnt waits (three half-tiles ahead).  The older
// two-stage kernel (one vmcnt(0) + barrier per K tile) keeps the narrower tiles.
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace aphro {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;

struct Fp8LargeParams {
  const uint8_t* a;        // [M, K] e4m3
  const uint8_t* w;        // [N, K] e4m3
  const float* a_scales;   // [M] or [1] or null
  const float* b_scales;   // [N] or [1] or null
  const uint16_t* bias;    // [N] (output type) or null
  uint16_t* c;             // [M, N]
  int M, N, K;
  int a_per_token, b_per_channel, out_bf16;
  int tiles_m, tiles_n;
  int streamk;             // 1: persistent grid, equal unit ranges; 0: one workgroup per (tile, K slice)
  int ksplit;              // streamk == 0: K slices (blockIdx.y), fp32 slabs + reduce kernel when > 1
  float* partial;          // streamk == 0: [ksplit][M][N];  streamk == 1: [grid] accumulator images (see publish)
  unsigned* flags;         // streamk == 1: [grid] "image published", zeroed by the host before the launch
  int debug;               // lab only (APHRO_FP8_LARGE_DEBUG): 1 = no output stores, 2 = one K tile per segment
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t f8_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// XCD x (workgroup ids x, x + 8, ...) gets a contiguous range of logical ids; bijective on [0, n) for any n
__device__ __forceinline__ int xcd_contiguous(int bid, int n) {
  const int q = n / 8, r = n % 8, xcd = bid % 8, k = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

constexpr int SC_COHERENT = 17;   // buffer aux bits sc0 | sc1: write-through stores / loads that do not trust this XCD's L2

// [begin(j), begin(j + 1)): the (tile, K tile) units of logical workgroup j of a stream-K grid (equal contiguous ranges).
// (Round 5 lab, profiles/r5_fp8_large_lab.txt: starting the ranges of XCD x -- or of groups of 8 CUs inside an XCD -- late
// by a fraction of a tile so that the C store bursts do not coincide measured 2-13 % SLOWER at every shape but one: the
// cut tiles' accumulator images cost more than the bursts, which the descriptor-based flush below shrank to 5k cycles.)
struct F8Units {
  int64_t U;
  int G;
  __device__ __forceinline__ int64_t begin(int j) const { return j >= G ? U : j * U / G; }
};

// What happens to a wave's accumulators once the K range of a segment is done: published (stream-K, not the owner),
// merged + scaled-mm epilogue (owner), or written as an fp32 slab (split-K form).  Shared by both kernels.
// lane holds, for row m = mb*32 + l31, columns nb*32 + 8 q + 4 kh + (0..3), q = reg >> 2
template <int NWAVE, bool BF16OUT>
__device__ __forceinline__ void fp8_large_finish(const Fp8LargeParams& p, f32x16 (&acc)[2][4], unsigned char* smem,
                                                 const F8Units& un, const __amdgpu_buffer_rsrc_t rp, bool head, bool tail,
                                                 int w, int tile, int ktiles_total, int m0, int n0, int wave, int wm,
                                                 int wn, int lane, unsigned long long* stamp = nullptr) {
  const int G = un.G;
  const int kh = lane >> 5, l31 = lane & 31;
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
                                                 ((w * NWAVE + wave) * 32 + (nb * 4 + mb) * 4 + q) * 1024 + lane * 16, 0, SC_COHERENT);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains ...
    __syncthreads();                                     // ... before ONE lane raises the flag
    if (threadIdx.x == 0) __hip_atomic_store(p.flags + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (p.streamk && !tail) {
    // owner of a tile whose K range continues in the following workgroups (always the LAST segment): add their images
    // in workgroup order.  They were published long ago unless the whole tile is being computed right now.
    const int64_t tile_end = (int64_t)(tile + 1) * ktiles_total;
    for (int j = w + 1; j < G && un.begin(j) < tile_end; ++j) {
      if (threadIdx.x == 0) {
        // Bounded: workgroup j publishes its first segment right after it starts, so this only waits long if j is not
        // resident yet.  Progress needs two consecutive logical workgroups resident at some point, i.e. more than half
        // of the CUs available to this kernel; if something else pins the chip for 5 s, abort instead of hanging.
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(p.flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          __builtin_amdgcn_s_sleep(4);
          if (wall_clock64() - t0 > 500000000ull) __builtin_trap();      // 100 MHz ticks
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
                rp, ((j * NWAVE + wave) * 32 + (nb * 4 + mb) * 4 + q) * 1024 + lane * 16, 0, SC_COHERENT);
            const f32x4 v = __builtin_bit_cast(f32x4, raw);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[nb][mb][4 * q + r] += v[r];
          }
    }
  }
  if (!p.streamk && p.ksplit > 1) {      // fp32 slab of this K slice; summed by fp8_splitk_reduce_large_kernel
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
  // ---- scaled-mm epilogue -> wave-private LDS transpose -> full-row stores ----------------------------------------------
  unsigned char* region = smem + wave * 16384;
  const int colbase = n0 + wn * 64;
  // every scale / bias value this lane needs, fetched up front (one wait): issued one by one next to their use the 32
  // dependent L2 round trips cost 28k cycles per tile -- a fifth of the kernel at K = 4096
  f32x4 sbv[2][4];
  u16x4 bbv[2][4];
  float sav[4];
  {
    const float s0 = (p.b_scales && !p.b_per_channel) ? p.b_scales[0] : 1.f;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = colbase + nb * 32 + 8 * q + 4 * kh;
        sbv[nb][q] = (p.b_scales && p.b_per_channel) ? *reinterpret_cast<const f32x4*>(p.b_scales + col) : f32x4{s0, s0, s0, s0};
        bbv[nb][q] = p.bias ? *reinterpret_cast<const u16x4*>(p.bias + col) : u16x4{0, 0, 0, 0};
      }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
      sav[mb] = p.a_scales ? p.a_scales[p.a_per_token ? min(m0 + wm * 128 + mb * 32 + l31, p.M - 1) : 0] : 1.f;
  }
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const float sa_ = sav[mb];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 sb = sbv[nb][q];
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // order of test_cutlass.py:43; the bias vector is zeros when there is none (one code path: with a branch
          // per output type / bias hipcc spills 200 registers here)
          v[r] = sa_ * (sb[r] * acc[nb][mb][4 * q + r]) + (BF16OUT ? bf16_bits_to_f32(bbv[nb][q][r]) : f16_bits_to_f32(bbv[nb][q][r]));
          asm("" : "+v"(v[r]));                            // fp32 result first, ONE rounding to 16 bits second
        }
        epi_put(region, mb * 32 + l31, nb * 8 + 2 * q + kh, u32x2{pack2_16<BF16OUT>(v[0], v[1]), pack2_16<BF16OUT>(v[2], v[3])});
      }
  }
  if (stamp) stamp[3] = __builtin_amdgcn_s_memtime();
  if (p.debug != 1)
    {
      const int64_t origin = (int64_t)(m0 + wm * 128) * p.N + colbase;      // (elements)
      epi_flush_buf(region, p.c + origin, p.N, ((int64_t)p.M * p.N - origin) * 2, lane);
    }
  if (stamp) {
    stamp[4] = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp[5] = __builtin_amdgcn_s_memtime();
  }
}

template <int WM, int WN, int STAGES, bool BF16OUT>
__global__ __launch_bounds__(WM * WN * 64) void fp8_gemm_large_kernel(Fp8LargeParams p) {
  constexpr int NWAVE = WM * WN;
  constexpr int BM = 128 * WM, BN = 64 * WN, BK = 128;
  constexpr int A_STAGE = BM * BK;                  // bytes
  constexpr int B_STAGE = BN * BK;
  constexpr int STAGE = A_STAGE + B_STAGE;
  constexpr int IMAGE = NWAVE * 32 * 1024;          // bytes of one workgroup's accumulator image
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int kh = lane >> 5, l31 = lane & 31;
  const int ktiles_total = p.K / BK;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int G = gridDim.x;
  const int64_t U = (int64_t)ntiles * ktiles_total;

  // ---- this workgroup's units [u, u_end): unit = tile * ktiles_total + K tile; neighbouring ids share the A row panel
  F8Units un;
  un.U = U; un.G = G;
  int u, u_end, w = 0;
  if (p.streamk) {
    w = xcd_contiguous(blockIdx.x, G);
    u = (int)un.begin(w);
    u_end = (int)un.begin(w + 1);
  } else {
    const int per = ktiles_total / p.ksplit;
    u = xcd_contiguous(blockIdx.x, ntiles) * ktiles_total + blockIdx.y * per;
    u_end = u + per;
  }

  const __amdgpu_buffer_rsrc_t ra = f8_rsrc(p.a, (uint32_t)((size_t)p.M * p.K));
  const __amdgpu_buffer_rsrc_t rb = f8_rsrc(p.w, (uint32_t)((size_t)p.N * p.K));
  const __amdgpu_buffer_rsrc_t rp = f8_rsrc(p.partial, p.streamk ? (uint32_t)((size_t)G * IMAGE) : 0u);
  constexpr int A_PER_WAVE = BM / 8 / NWAVE, B_PER_WAVE = BN / 8 / NWAVE;
  static_assert((BM / 8) % NWAVE == 0 && (BN / 8) % NWAVE == 0, "every wave issues the same number of loads");
  constexpr int DMA_PER_TILE = A_PER_WAVE + B_PER_WAVE;

  // fragment addresses inside a stage: 32 consecutive k bytes = 16-byte slots 4 j + 2 kh and + 1.  Physical slot P of
  // row r holds logical slot P ^ f(r), f(r) = (r >> 1) & 7: the logical pair (2x, 2x + 1) sits at (2x ^ f, 2x ^ f ^ 1).
  // f(row) depends on l31 only (rows of one lane differ by multiples of 32), j toggles bit 2 of the slot: one base
  // register per operand, the rest are immediates / one XOR.
  const int a_base = (wm * 128 + l31) * 128 + (((2 * kh) ^ ((l31 >> 1) & 7)) << 4);
  const int b_base = A_STAGE + (wn * 64 + l31) * 128 + (((2 * kh) ^ ((l31 >> 1) & 7)) << 4);
  auto read_frag = [&](const unsigned char* sa, int off) {
    const u32x4 lo = *reinterpret_cast<const u32x4*>(sa + off);
    const u32x4 hi = *reinterpret_cast<const u32x4*>(sa + (off ^ 16));
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };

  bool first_segment = true;
  while (u < u_end) {
    const int tile = u / ktiles_total;
    const int k0 = u - tile * ktiles_total;
    int k1 = min(ktiles_total, k0 + (u_end - u));
    u += k1 - k0;
    const bool head = k0 == 0, tail = k1 == ktiles_total;
    if (p.debug == 2) k1 = k0 + 1;
    const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    if (!first_segment) __syncthreads();            // the previous segment's epilogue is done with the LDS
    first_segment = false;

    // ---- staging (direct-to-LDS): one DMA instruction = 8 rows x 128 B (lane -> row l / 8, 16-byte slot l % 8) ------
    int a_voff[A_PER_WAVE], b_voff[B_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
      const int row = (wave * A_PER_WAVE + i) * 8 + (lane >> 3);
      const int slot = (lane & 7) ^ ((row >> 1) & 7);
      a_voff[i] = min(m0 + row, p.M - 1) * p.K + slot * 16;
    }
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i) {
      const int row = (wave * B_PER_WAVE + i) * 8 + (lane >> 3);
      const int slot = (lane & 7) ^ ((row >> 1) & 7);
      b_voff[i] = (n0 + row) * p.K + slot * 16;
    }
    auto stage = [&](int st, int kt) {
      unsigned char* sa = smem + st * STAGE;
      const int soff = kt * BK;
#pragma unroll
      for (int i = 0; i < A_PER_WAVE; ++i) {
        const int voff = a_voff[i];   // (local copy: see wna16_gemm_large.hip on the hipcc host-stub bug)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_ptr)(sa + (wave * A_PER_WAVE + i) * 1024), 16, voff, soff, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < B_PER_WAVE; ++i) {
        const int voff = b_voff[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_ptr)(sa + A_STAGE + (wave * B_PER_WAVE + i) * 1024), 16, voff, soff, 0, 0);
      }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][mb][r] = 0.f;

    const int nk = k1 - k0;
    stage(0, k0);
    if constexpr (STAGES == 3) { if (nk > 1) stage(1, k0 + 1); }
    else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }

    for (int i = 0; i < nk; ++i) {
      const int st = i % STAGES;
      if constexpr (STAGES == 3) {
        // tile i has landed once at most the loads of tile i + 1 are outstanding; the raw barrier (no vmcnt(0) drain)
        // publishes it and frees stage (i + 2) % 3, read during tile i - 1
        if (i + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DMA_PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (i + 2 < nk) stage((i + 2) % 3, k0 + i + 2);
      } else {
        // the two waves of a SIMD (w, w + NWAVE/2) issue their loads at different points of the tile: while one is
        // busy issuing 8 direct-to-LDS loads the other one's MFMAs keep the matrix pipe fed
        if (i + 1 < nk && wave < NWAVE / 2) stage(st ^ 1, k0 + i + 1);
      }
      const unsigned char* sa = smem + st * STAGE;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (STAGES == 2) {
          if (j == 1) {
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < nk && wave >= NWAVE / 2) stage(st ^ 1, k0 + i + 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        i32x8 wf[2], af[4];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) wf[nb] = read_frag(sa + nb * 4096, b_base ^ (j << 6));
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) af[mb] = read_frag(sa + mb * 4096, a_base ^ (j << 6));
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[nb][mb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[nb], af[mb], acc[nb][mb], 0, 0, 0, 127, 0, 127);
      }
      if constexpr (STAGES == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
    if constexpr (STAGES == 3) __syncthreads();      // every wave is done with the stage buffers

    fp8_large_finish<NWAVE, BF16OUT>(p, acc, smem, un, rp, head, tail, w, tile, ktiles_total, m0, n0, wave, wm, wn, lane);
  }
}

// ---- the eight-phase kernel: 256 x 256 x 128 tile, 8 waves = 2 groups (wm = wave >> 2: the M half; waves w and w + 4 share
// a SIMD) x 4 column strips (wn = wave & 3) -----------------------------------------------------------------------------
// LDS: two K-tile buffers of 64 KiB = [activations 256 rows x 128 B | weights 256 rows x 128 B], rows XOR-swizzled as in
// the kernel above.  Half-tiles (16 KiB = 16 direct-to-LDS instructions, two per wave):
//   Ah(h): activation rows g*128 + h*64 + 0..63 of BOTH groups g   -- read by every wave in phase 1 (h = 0) / 3 (h = 1)
//   Wh(h): weight rows wn*64 + h*32 + 0..31 of ALL four strips     -- read by every wave in phase 1 (h = 0) / 2 (h = 1)
// so that a half-tile is dead for the whole workgroup one phase of reads after it was opened and can be refilled early.
// Per K tile t (buffer cur = t & 1), a wave runs
//   P1: read Wh0 (4 x b128), Ah0 (8); stage Ah1(t+1) -> other buffer; lgkmcnt(8); barrier; lgkmcnt(0); 4 MFMA (m-half 0, n-block 0); barrier
//   P2: read Wh1 (4);                 stage Wh0(t+2) -> cur;                           barrier; lgkmcnt(0); 4 MFMA (0, 1);               barrier
//   P3: read Ah1 (8);                 stage Ah0(t+2) -> cur;                           barrier; lgkmcnt(0); 4 MFMA (1, 1);               barrier
//   P4:                               stage Wh1(t+2) -> cur; vmcnt(6);                 barrier;             4 MFMA (1, 0);               barrier
// Group 1 runs one barrier behind group 0, so between two barriers one wave of each SIMD is in its MFMA cluster and the
// other in its read / stage block.  Hazards (cdna_hip_programming.md, "Read a staged buffer one phase AFTER the wait"):
//   RAW  a half-tile is read at the earliest in the phase after the vmcnt that retires it (P4 -> P1): every wave's wait sits
//        in front of its phase's first barrier, and the first reader passes one more barrier before it reads;
//   WAR  Wh0 is refilled ONE phase after its reads -- lgkmcnt(8) in front of P1's first barrier has retired them (they are
//        issued first); everything else two phases after its last read.
// Fragment reads are inline asm: a ds_read the compiler can see is ordered behind every LDS-DMA in flight (vmcnt(0)).
template <int OFF>
__device__ __forceinline__ void f8_lds_read128(u32x4& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void f8_touch(u32x4& x) { asm volatile("" : "+v"(x)); }

// V (lab variants; the host picks one): 1 = stage calls in the read block (the guide's template); 2 = each phase's two
// LDS-DMA instructions ride in the shadow of the wave's OWN MFMAs (one after the 1st, one after the 2nd of the cluster:
// a 32x32x64 fp8 MFMA keeps the pipe for 64 cycles, an LDS-DMA costs its wave 60-185 cycles of issue -- in the read
// block that issue time made the read block longer than the partner's MFMA cluster); 3 = 2 without the priority flips.
// The segment start is a raw barrier (no vmcnt(0): the previous tile's C stores drain under the next tile's loads).
// Lab-only ablations of V = 2 (WRONG results, timing only): 4 = no LDS-DMA in the K loop, 5 = no fragment reads in the K
// loop, 6 = no barriers in the K loop, 7 = MFMAs only.
template <bool BF16OUT, int V>
__global__ __launch_bounds__(512) void fp8_gemm_large8_kernel(Fp8LargeParams p) {
  constexpr int VS = V >= 4 ? 2 : V;                // schedule variant
  constexpr bool RAWSYNC = true;
  constexpr bool NO_DMA = V == 4 || V == 7, NO_READ = V == 5 || V == 7, NO_BAR = V == 6 || V == 7;
  constexpr int NWAVE = 8, BM = 256, BN = 256, BK = 128;
  constexpr int A_REGION = BM * BK;                 // 32 KiB
  constexpr int BUF = 2 * A_REGION;                 // 64 KiB per K tile
  constexpr int IMAGE = NWAVE * 32 * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int kh = lane >> 5, l31 = lane & 31;
  const int ktiles_total = p.K / BK;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int G = gridDim.x;
  const int64_t U = (int64_t)ntiles * ktiles_total;
  F8Units un;
  un.U = U; un.G = G;
  const int w = xcd_contiguous(blockIdx.x, G);
  int u = (int)un.begin(w);
  const int u_end = (int)un.begin(w + 1);

  const __amdgpu_buffer_rsrc_t ra = f8_rsrc(p.a, (uint32_t)((size_t)p.M * p.K));
  const __amdgpu_buffer_rsrc_t rb = f8_rsrc(p.w, (uint32_t)((size_t)p.N * p.K));
  const __amdgpu_buffer_rsrc_t rp = f8_rsrc(p.partial, (uint32_t)((size_t)G * IMAGE));

  // fragment read addresses (LDS byte offsets): [k half j][16-byte half e] of the lane's 32 consecutive k bytes; the row
  // blocks / halves of a phase are immediate offsets.  f(row) = (row >> 1) & 7 depends on l31 only.
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int fsw = (l31 >> 1) & 7;
  uint32_t a_addr[2][2], w_addr[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int slot = ((4 * j + 2 * kh + e) ^ fsw) << 4;
      a_addr[j][e] = lds0 + (wm * 128 + l31) * 128 + slot;
      w_addr[j][e] = lds0 + A_REGION + (wn * 64 + l31) * 128 + slot;
    }

  bool first_segment = true;
  int seg_no = -1;
  while (u < u_end) {
    const int tile = u / ktiles_total;
    const int k0 = u - tile * ktiles_total;
    int k1 = min(ktiles_total, k0 + (u_end - u));
    u += k1 - k0;
    const bool head = k0 == 0, tail = k1 == ktiles_total;
    if (p.debug == 2) k1 = k0 + 1;
    const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    unsigned long long* stamp = nullptr;
#define F8_STAMP(i)
    if (!first_segment) {                           // the previous segment's epilogue is done with the LDS
      if constexpr (RAWSYNC) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
      else __syncthreads();
    }
    first_segment = false;
    F8_STAMP(0)

    // ---- staging: this wave's two instructions (8 rows x 128 B each) of every half-tile kind ---------------------------
    // Ah(h): instruction i (0..15) covers rows (i >> 3) * 128 + h * 64 + (i & 7) * 8 .. + 8; Wh(h): rows (i >> 2) * 64 + h * 32 + (i & 3) * 8
    int a_voff[2][2], w_voff[2][2], a_row0[2][2], w_row0[2][2];
    bool in_loop = false;                           // (lab ablations only)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int i = 2 * wave + t;
        a_row0[h][t] = (i >> 3) * 128 + h * 64 + (i & 7) * 8;
        w_row0[h][t] = (i >> 2) * 64 + h * 32 + (i & 3) * 8;
        const int ar = a_row0[h][t] + (lane >> 3), wr = w_row0[h][t] + (lane >> 3);
        a_voff[h][t] = min(m0 + ar, p.M - 1) * p.K + (((lane & 7) ^ ((ar >> 1) & 7)) << 4);
        w_voff[h][t] = (n0 + wr) * p.K + (((lane & 7) ^ ((wr >> 1) & 7)) << 4);
      }
    auto stage_a1 = [&](int h, int t, int buf, int kt) {
      if (NO_DMA && in_loop) return;
      const int voff = a_voff[h][t];                // (local copy: see wna16_gemm_large.hip on the hipcc host-stub bug)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_ptr)(smem + buf * BUF + a_row0[h][t] * 128), 16, voff, kt * BK, 0, 0);
    };
    auto stage_w1 = [&](int h, int t, int buf, int kt) {
      if (NO_DMA && in_loop) return;
      const int voff = w_voff[h][t];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_ptr)(smem + buf * BUF + A_REGION + w_row0[h][t] * 128), 16, voff, kt * BK, 0, 0);
    };
    auto stage_a = [&](int h, int buf, int kt) { stage_a1(h, 0, buf, kt); stage_a1(h, 1, buf, kt); };
    auto stage_w = [&](int h, int buf, int kt) { stage_w1(h, 0, buf, kt); stage_w1(h, 1, buf, kt); };

    f32x16 acc[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][mb][r] = 0.f;

    const int nk = k1 - k0;
    // ---- prologue: K tile 0 complete, three half-tiles of K tile 1 in flight --------------------------------------------
    // The K loop is ONE basic block: past the segment's last K tile the stage slots keep loading (a clamped K tile, into
    // half-tiles nobody reads any more) so that neither a branch nor a second vmcnt count is needed -- with wave-uniform
    // branches around the stage calls hipcc sank the MFMAs of three phases to the end of the loop body.
    const int klast = k0 + nk - 1;
    stage_w(0, 0, k0); stage_a(0, 0, k0); stage_w(1, 0, k0); stage_a(1, 0, k0);
    { const int kt1 = min(k0 + 1, klast); stage_w(0, 1, kt1); stage_a(0, 1, kt1); stage_w(1, 1, kt1); }
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();       // group 1 runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);
    F8_STAMP(1)

    u32x4 wlo[2][2], whi[2][2], alo[2][2], ahi[2][2];   // [n block][j] / [m block of the current half][j]
    if constexpr (NO_READ) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { wlo[i][j] = u32x4{0x38383838u + lane, 0, 0, 0}; whi[i][j] = wlo[i][j]; alo[i][j] = wlo[i][j]; ahi[i][j] = wlo[i][j]; }
    }
    auto read_w = [&](auto NB) {
      constexpr int nb = decltype(NB)::value;
      if (NO_READ) return;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f8_lds_read128<nb * 4096>(wlo[nb][j], w_addr[j][0]);
        f8_lds_read128<nb * 4096>(whi[nb][j], w_addr[j][1]);
      }
    };
    auto read_a = [&](auto MH) {
      constexpr int mh = decltype(MH)::value;
      if (NO_READ) return;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (mb == 0) { f8_lds_read128<mh * 8192>(alo[0][j], a_addr[j][0]); f8_lds_read128<mh * 8192>(ahi[0][j], a_addr[j][1]); }
          else { f8_lds_read128<mh * 8192 + 4096>(alo[1][j], a_addr[j][0]); f8_lds_read128<mh * 8192 + 4096>(ahi[1][j], a_addr[j][1]); }
        }
    };
    // the phase's 4 MFMAs; `piece(t)` (VS >= 2) issues the phase's t-th LDS-DMA instruction behind the (t + 1)-th MFMA
    auto mma = [&](int nb, int mh, auto piece) {
      if constexpr (VS != 3) __builtin_amdgcn_s_setprio(1);
      int issued = 0;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const i32x8 wf = {(int)wlo[nb][j][0], (int)wlo[nb][j][1], (int)wlo[nb][j][2], (int)wlo[nb][j][3],
                            (int)whi[nb][j][0], (int)whi[nb][j][1], (int)whi[nb][j][2], (int)whi[nb][j][3]};
          const i32x8 af = {(int)alo[mb][j][0], (int)alo[mb][j][1], (int)alo[mb][j][2], (int)alo[mb][j][3],
                            (int)ahi[mb][j][0], (int)ahi[mb][j][1], (int)ahi[mb][j][2], (int)ahi[mb][j][3]};
          acc[nb][mh * 2 + mb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf, af, acc[nb][mh * 2 + mb], 0, 0, 0, 127, 0, 127);
          if constexpr (VS >= 2) {
            if (issued < 2) {
              __builtin_amdgcn_sched_barrier(0);
              piece(issued);
              __builtin_amdgcn_sched_barrier(0);
            }
            ++issued;
          }
        }
      if constexpr (VS != 3) __builtin_amdgcn_s_setprio(0);
    };
    auto landed_w = [&](int nb) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { f8_touch(wlo[nb][j]); f8_touch(whi[nb][j]); }
    };
    auto landed_a = [&]() {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int j = 0; j < 2; ++j) { f8_touch(alo[mb][j]); f8_touch(ahi[mb][j]); }
    };
#define F8_BAR()  do { __builtin_amdgcn_sched_barrier(0); if constexpr (!NO_BAR) __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define F8_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

    in_loop = true;
    for (int t = 0; t < nk; ++t) {
      const int cur = t & 1;
      const int kt1 = min(k0 + t + 1, klast), kt2 = min(k0 + t + 2, klast);
      // ---- P1 -------------------------------------------------------------------------------------------------------
      read_w(std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
      read_a(std::integral_constant<int, 0>{});
      if constexpr (VS == 1) {
        stage_a(1, cur ^ 1, kt1);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");    // the four weight reads (issued first) have returned
      }
      F8_BAR();
      F8_LGKM0(); landed_w(0); landed_a();
      mma(0, 0, [&](int i) { stage_a1(1, i, cur ^ 1, kt1); });
      F8_BAR();
      // ---- P2 -------------------------------------------------------------------------------------------------------
      read_w(std::integral_constant<int, 1>{});
      if constexpr (VS == 1) stage_w(0, cur, kt2);
      F8_BAR();
      F8_LGKM0(); landed_w(1);
      mma(1, 0, [&](int i) { stage_w1(0, i, cur, kt2); });
      F8_BAR();
      // ---- P3 -------------------------------------------------------------------------------------------------------
      read_a(std::integral_constant<int, 1>{});
      if constexpr (VS == 1) stage_a(0, cur, kt2);
      F8_BAR();
      F8_LGKM0(); landed_a();
      mma(1, 1, [&](int i) { stage_a1(0, i, cur, kt2); });
      F8_BAR();
      // ---- P4 -------------------------------------------------------------------------------------------------------
      if constexpr (VS == 1) {
        stage_w(1, cur, kt2);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // everything older than P2..P4's six loads: buffer cur^1 is whole
      } else {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // P2's and P3's four loads may be in flight: buffer cur^1 is whole
      }
      F8_BAR();
      mma(0, 1, [&](int i) { stage_w1(1, i, cur, kt2); });
      F8_BAR();
      // the next K tile lives in the other buffer
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) { a_addr[j][e] += cur ? -BUF : BUF; w_addr[j][e] += cur ? -BUF : BUF; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the over-run loads have landed before anything else uses the LDS
    if (wm == 0) __builtin_amdgcn_s_barrier();       // catch up with group 1: every wave is done with the stage buffers
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (nk & 1) {                                    // leave the read addresses on buffer 0 for the next segment
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) { a_addr[j][e] -= BUF; w_addr[j][e] -= BUF; }
    }
#undef F8_BAR
#undef F8_LGKM0
    F8_STAMP(2)
    fp8_large_finish<NWAVE, BF16OUT>(p, acc, smem, un, rp, head, tail, w, tile, ktiles_total, m0, n0, wave, wm, wn, lane, stamp);
    if (stamp) stamp[7] = __builtin_amdgcn_s_memrealtime();
#undef F8_STAMP
  }
}

// partial [S][M*N] fp32 -> c: sum in fixed order, then the scaled-mm epilogue
__global__ void fp8_splitk_reduce_large_kernel(const float* __restrict__ partial, Fp8LargeParams p, int S) {
  const int64_t mn = (int64_t)p.M * p.N;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= mn) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(partial + i);
  for (int k = 1; k < S; ++k) s += *reinterpret_cast<const f32x4*>(partial + (size_t)k * mn + i);
  const int row = (int)(i / p.N), col = (int)(i % p.N);
  const float sa_ = p.a_scales ? p.a_scales[p.a_per_token ? row : 0] : 1.f;
  u16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float sb = p.b_scales ? p.b_scales[p.b_per_channel ? col + j : 0] : 1.f;
    float v = sa_ * (sb * s[j]);
    if (p.bias) v += p.out_bf16 ? bf16_bits_to_f32(p.bias[col + j]) : f16_bits_to_f32(p.bias[col + j]);
    asm("" : "+v"(v));
    o[j] = p.out_bf16 ? f32_to_bf16_bits(v) : f32_to_f16_bits(v);
  }
  *reinterpret_cast<u16x4*>(p.c + i) = o;
}

template <int WM, int WN, int STAGES, bool BF16OUT>
static int launch_fp8_large_t(const Fp8LargeParams& p, int grid_x, hipStream_t st) {
  constexpr size_t lds = (size_t)STAGES * (128 * WM + 64 * WN) * 128;
  static_assert(lds <= 160 * 1024 && lds >= (size_t)WM * WN * 16384, "stage buffers must fit and cover the epilogue regions");
  static bool attr_set_dev[APHRO_MAX_DEVICES] = {}; bool& attr_set = attr_set_dev[device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)fp8_gemm_large_kernel<WM, WN, STAGES, BF16OUT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess) {
      set_error("fp8_gemm_large: cannot raise the dynamic LDS limit");
      return APHRO_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((fp8_gemm_large_kernel<WM, WN, STAGES, BF16OUT>), dim3(grid_x, p.streamk ? 1 : p.ksplit), dim3(WM * WN * 64), lds, st, p);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

template <bool BF16OUT, int V>
static int launch_fp8_large8_t(const Fp8LargeParams& p, int grid_x, hipStream_t st) {
  constexpr size_t lds = 128 * 1024;
  static bool attr_set_dev[APHRO_MAX_DEVICES] = {}; bool& attr_set = attr_set_dev[device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)fp8_gemm_large8_kernel<BF16OUT, V>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      set_error("fp8_gemm_large: cannot raise the dynamic LDS limit");
      return APHRO_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((fp8_gemm_large8_kernel<BF16OUT, V>), dim3(grid_x), dim3(512), lds, st, p);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

template <int WM, int WN, int STAGES>
static int launch_fp8_large(const Fp8LargeParams& p, int grid_x, hipStream_t st) {
  return p.out_bf16 ? launch_fp8_large_t<WM, WN, STAGES, true>(p, grid_x, st) : launch_fp8_large_t<WM, WN, STAGES, false>(p, grid_x, st);
}

}  // namespace aphro

using namespace aphro;

struct Fp8LargePlan { int wm, wn, tiles_m, tiles_n, streamk, grid, ksplit; };

static int cu_count() { return device_cu_count(); }

// Stream-K (one persistent workgroup per CU) when the biggest tile the shape allows still gives >= 128 tiles: then
// every workgroup gets at least half a tile's K loop and a tile is cut between at most 3 workgroups.  Below that: one
// workgroup per tile, narrower tiles and up to 8 K slices (fp32 slabs) so that ~200+ workgroups exist.
static Fp8LargePlan fp8_large_plan(int64_t M, int64_t N, int64_t K) {
  Fp8LargePlan pl;
  pl.wm = M > 128 ? 2 : 1;
  const int64_t rows = (M + 128 * pl.wm - 1) / (128 * pl.wm);
  const int mode = APHRO_LAB_ENV_INT("APHRO_FP8_LARGE_STREAMK", -1);
  const int64_t big_tiles = N % 256 == 0 ? rows * (N / 256) : rows * (N / 128);
  pl.streamk = (mode >= 0 ? mode : (big_tiles >= 128)) && device_coresident_cu_count() > 0;
  pl.ksplit = 1;
  if (pl.streamk) {
    pl.wn = N % 256 == 0 ? 4 : 2;
    pl.grid = cu_count();
  } else {
    pl.wn = (N % 256 == 0 && rows * (N / 256) >= 200) ? 4 : 2;
    const int64_t tiles = rows * (N / (64 * pl.wn));
    for (int s = 2; s <= 8; ++s) {
      if (tiles * pl.ksplit >= 200) break;
      if (K % (s * 128) == 0 && K / s >= 1024) pl.ksplit = s;
    }
    { const int s = APHRO_LAB_ENV_INT("APHRO_FP8_LARGE_KSPLIT", 0); if (s >= 1 && K % (s * 128) == 0) pl.ksplit = s; }
    pl.grid = (int)tiles;
  }
  pl.tiles_m = (int)rows;
  pl.tiles_n = (int)(N / (64 * pl.wn));
  return pl;
}

constexpr size_t FP8_LARGE_FLAG_BYTES = 4096;

// Bytes of scratch aphro_scaled_mm_fp8_large needs: stream-K flags + accumulator images, or the fp32 split-K slabs.
extern "C" size_t aphro_scaled_mm_fp8_large_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  const Fp8LargePlan pl = fp8_large_plan(M, N, K);
  if (pl.streamk) return FP8_LARGE_FLAG_BYTES + (size_t)pl.grid * pl.wm * pl.wn * 32 * 1024;
  return pl.ksplit > 1 ? (size_t)pl.ksplit * M * N * sizeof(float) : 0;
}

// out[M, N] = a_scales (.) (a[M, K] . b[N, K]^T) (.) b_scales + bias; e4m3 operands, any M (meant for M > 64).
// N % 128 == 0, K % 128 == 0.  Same argument meaning as aphro_scaled_mm_fp8.
extern "C" int aphro_scaled_mm_fp8_large(void* out, const void* a, const void* b, const float* a_scales,
                                         const float* b_scales, const void* bias, void* workspace,
                                         size_t workspace_bytes, int64_t M, int64_t N, int64_t K,
                                         int a_scale_per_token, int b_scale_per_channel, int out_dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APHRO_CHECK(out_dtype == APHRO_F16 || out_dtype == APHRO_BF16, "scaled_mm_fp8_large: out dtype must be f16 or bf16");
  APHRO_CHECK(N % 128 == 0 && K % 128 == 0, "scaled_mm_fp8_large: N=%ld and K=%ld must be multiples of 128", (long)N, (long)K);
  APHRO_CHECK(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0, "scaled_mm_fp8_large: operands must be 16-byte aligned");
  APHRO_CHECK((size_t)M * K < 0xffffffffull && (size_t)N * K < 0xffffffffull, "scaled_mm_fp8_large: operand exceeds 4 GiB");
  if (M == 0) return APHRO_OK;
  const Fp8LargePlan pl = fp8_large_plan(M, N, K);
  APHRO_CHECK((int64_t)pl.tiles_m * pl.tiles_n * (K / 128) < 0x7fffffff, "scaled_mm_fp8_large: too many work units");
  const size_t need = aphro_scaled_mm_fp8_large_workspace_bytes(M, N, K);
  if (need > 0 && (workspace == nullptr || workspace_bytes < need)) {
    set_error("scaled_mm_fp8_large: workspace %zu < %zu bytes", workspace_bytes, need);
    return APHRO_ERR_WORKSPACE;
  }
  Fp8LargeParams p;
  p.a = (const uint8_t*)a; p.w = (const uint8_t*)b; p.a_scales = a_scales; p.b_scales = b_scales;
  p.bias = (const uint16_t*)bias; p.c = (uint16_t*)out;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.a_per_token = a_scale_per_token; p.b_per_channel = b_scale_per_channel; p.out_bf16 = out_dtype == APHRO_BF16;
  p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n; p.streamk = pl.streamk; p.ksplit = pl.ksplit;
  p.flags = nullptr; p.partial = (float*)workspace;
  p.debug = APHRO_LAB_ENV_INT("APHRO_FP8_LARGE_DEBUG", 0);
  if (pl.streamk) {
    p.flags = (unsigned*)workspace;
    p.partial = (float*)((char*)workspace + FP8_LARGE_FLAG_BYTES);
    if (hipMemsetAsync(p.flags, 0, FP8_LARGE_FLAG_BYTES, st) != hipSuccess) {
      set_error("scaled_mm_fp8_large: cannot clear the stream-K flags");
      return APHRO_ERR_LAUNCH;
    }
  }
  int rc;
  // eight-phase schedule from 16 K tiles per output tile up: below that the longer prologue (seven half-tiles, the stagger
  // barriers) costs more than the K loop gains (K = 512 / 1024: 29.6 / 47.4 us against 25.8 / 43.6 for the two-stage kernel)
  const int eight_lab = APHRO_LAB_ENV_INT("APHRO_FP8_LARGE_8PHASE", -1);
  const int eight = eight_lab >= 0 ? eight_lab : (K >= 2048 ? 2 : 0);
  if (pl.wm == 2 && pl.wn == 4 && pl.streamk && eight) {
    switch (eight) {
#define F8_V(v) case v: rc = p.out_bf16 ? launch_fp8_large8_t<true, v>(p, pl.grid, st) : launch_fp8_large8_t<false, v>(p, pl.grid, st); break;
      F8_V(2)
#undef F8_V
      default: set_error("scaled_mm_fp8_large: unknown schedule variant %d", eight); return APHRO_ERR_INVALID;
    }
  }
  else if (pl.wm == 2) rc = pl.wn == 4 ? launch_fp8_large<2, 4, 2>(p, pl.grid, st) : launch_fp8_large<2, 2, 3>(p, pl.grid, st);
  else rc = pl.wn == 4 ? launch_fp8_large<1, 4, 3>(p, pl.grid, st) : launch_fp8_large<1, 2, 3>(p, pl.grid, st);
  if (rc != APHRO_OK) return rc;
  if (!pl.streamk && pl.ksplit > 1) {
    const int64_t mn = M * N;
    hipLaunchKernelGGL(fp8_splitk_reduce_large_kernel, dim3((unsigned)((mn / 4 + 255) / 256)), dim3(256), 0, st, p.partial, p, pl.ksplit);
    APHRO_LAUNCH_CHECK();
  }
  return APHRO_OK;
}
