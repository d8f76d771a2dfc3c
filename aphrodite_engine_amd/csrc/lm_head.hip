// Decode-time LM head for gfx950: logits = hidden[M <= 32, K] x W[V, K]^T (16-bit, fp32 accumulate) with the greedy
// argmax of every row folded in -- ONE launch for what the reference does as LogitsProcessor._get_logits (a library GEMM,
// modeling/layers/logits_processor.py:78-96) + Sampler._greedy_sample (torch.argmax, modeling/layers/sampler.py) and this
// package did until round 3 as hipBLASLt (200 us for the 1.05 GB of Llama-3's [128256, 4096] head: 5.25 TB/s) + argmax_rows
// (13 us) + 8 MB of logits written and read back.  HBM-bound: every weight byte is read once and meets 2 MFMAs.
//
//   * grid = one workgroup per CU (8 waves), each a contiguous run of 16-row vocabulary tiles; the waves split K
//     (K / 8 each) and keep their slice of the activations resident in registers as MFMA A fragments.
//   * the weight stream never touches a VGPR on its way in: `buffer_load ... lds` of 4 rows x 256 bytes per instruction
//     (full cache lines; the 16-row fragment gather straight from the rows costs 4x the L1 lookups -- see the AROW note in
//     wna16_gemm_resident.hip), XOR-swizzled by the row so that the B-fragment reads are bank-conflict free.  A wave's
//     ring holds KS 128-k segments = one tile ahead of the one it computes: 16 KiB in flight per wave, 128 KiB per CU.
//     hipcc does not order a ds_read after the LDS-DMA that fills its source: the waits are written by hand (loads return
//     in order; stores between them can only make a wait return earlier for the stores' own sake -- see below).
//   * K reduction over the waves through LDS once per tile; the thread that owns (token, column) of the tile rounds the sum
//     to the activation dtype (the value the reference's argmax sees), optionally stores it, and keeps a running
//     (max, index); at the end the workgroup's best per token goes to a partial buffer (write-through), a ticket is taken,
//     and the LAST workgroup reduces the partials: ties -> the lowest index, NaN never wins (= argmax_rows, step_ops.hip).
#include <mutex>
#include <utility>

#include "common.h"

namespace aphro {

struct LmHeadParams {
  const uint16_t* a;      // [M, lda] hidden states
  const uint16_t* w;      // [V, ldw] weights (row-major, K contiguous)
  uint16_t* logits;       // optional [M, ldl]
  int64_t* out_ids;       // [M]
  float* part_val;        // [grid][32]
  int* part_idx;          // [grid][32]
  unsigned* counter;      // one ticket, zero between launches
  int M, K, V;
  int lda, ldw, ldl;
  int tiles;              // ceil(V / 16)
};

typedef __attribute__((address_space(3))) void* lmh_lds_ptr;

template <int B, int E, typename F>
__device__ __forceinline__ void lmh_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    lmh_static_for<B + 1, E>(f);
  }
}

template <typename T>
__device__ __forceinline__ f32x4 lmh_mfma(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (__is_same(T, Half))
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence = s_waitcnt vmcnt(0):
// once per tile it would drain the ring's prefetch (the LDS-DMA loads in flight for the NEXT tile).
__device__ __forceinline__ void lmh_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ bool lmh_better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

// KS: 128-k segments per wave (K = 8 waves x KS x 128).  MT: 16-token tiles (M <= 16 MT).
template <typename T, int MT, int KS>
__global__ __launch_bounds__(512, 1) void lm_head_argmax_kernel(LmHeadParams p) {
  constexpr int NWV = 8;
  constexpr int SEGB = 16 * 256;                    // one staged segment: [16 rows][16 chunks of 8 k]
  constexpr int RING = KS * SEGB;                   // bytes per wave
  constexpr int RP = 17;                            // row pitch of the reduction tile (floats)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [NWV][RING] | red [NWV][16 MT][RP]
  __shared__ int last_flag;
  float* const red = reinterpret_cast<float*>(smem + NWV * RING);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  unsigned char* const ring = smem + wave * RING;

  // this workgroup's vocabulary tiles: blockIdx.x, + G, + 2 G, ... (LMH_CONTIG: a contiguous run instead)
  const int G = gridDim.x;
  const int t0 = blockIdx.x, t1 = p.tiles, tstep = G;
  // this wave's K slice: segments wave, wave + 8, ... (LMH_KBLOCK: KS consecutive segments) -- the eight waves' requests
  // for their s-th segment then cover 2 KiB of each row together
  auto kseg = [&](int s) { return (s * NWV + wave) * 128; };

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0,
      (uint32_t)(((size_t)(p.V - 1) * p.ldw + p.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.a), 0,
      (uint32_t)(((size_t)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);

  // ---- the activations of this K slice: A fragments, lane (g, c) = token 16 i + c, k = kseg(s) + 32 u + 8 g .. + 8 ------
  u32x4 af[KS][4][MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int vo = min(16 * i + c, p.M - 1) * p.lda * 2 + g * 16;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int u = 0; u < 4; ++u) af[s][u][i] = __builtin_amdgcn_raw_buffer_load_b128(ra, vo, (kseg(s) + 32 * u) * 2, 0);
  }
  // staging instruction `it` of a segment: lane -> slot row 4 it + lane / 16, chunk (lane % 16) ^ (slot row % 16)
  int voff_w[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int srow = 4 * it + (lane >> 4);
    voff_w[it] = srow * p.ldw * 2 + (((lane & 15) ^ (srow & 15)) << 4);
  }
  // fragment read of k-step u: row c, chunk (4 u + g) ^ c of the segment.  Through inline asm: for a ds_read the compiler
  // can see, it inserts its own conservative vmcnt waits against EVERY LDS-DMA in flight (seen in the ISA: vmcnt(8) / (4) /
  // (0) ahead of the segments of one tile -- the ring's prefetch depth gone); the ordering is the explicit wait below.
  uint32_t rd[4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
    rd[u] = (uint32_t)(uintptr_t)(lmh_lds_ptr)(ring + c * 256 + (((4 * u + g) ^ c) << 4));
  auto stage = [&](int tile, auto S_) {             // segment s of `tile` -> ring slot s (rows past V: the last row)
    constexpr int s = decltype(S_)::value;
    const int row0 = min(tile * 16, p.V - 16);      // (V >= 16; a ragged last tile re-reads rows of its neighbour)
    const int so = row0 * p.ldw * 2 + kseg(s) * 2;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int vo = voff_w[it];                    // (local copy: hipcc host-stub bug, see wna16_gemm_large.hip)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lmh_lds_ptr)(ring + s * SEGB + it * 1024), 16, vo, so, 0, 0);
    }
  };

  float best = -INFINITY;                           // thread (token = tid / 16, column = tid % 16): running argmax
  int best_i = 0x7fffffff;
  const int tok = tid >> 4, col = tid & 15;

  __builtin_amdgcn_sched_barrier(0);
  if (t0 < t1) lmh_static_for<0, KS>([&](auto S_) { stage(t0, S_); });
  __builtin_amdgcn_sched_barrier(0);

  for (int t = t0; t < t1; t += tstep) {
    const bool more = t + tstep < t1;
    f32x4 acc4[4][MT];                                // one accumulator per k-step of a segment: four independent MFMA chains
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i) acc4[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    lmh_static_for<0, KS>([&](auto S_) {
      constexpr int s = decltype(S_)::value;
      // segment s of tile t has landed when at most the loads issued after it are outstanding: the segments s+1 .. KS-1
      // of this tile and, once the previous iterations of this loop have issued them, segments 0 .. s-1 of the next one
      if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (KS - 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (KS - 1 - s)) : "memory");
      u32x4 bf[4];
      asm volatile(
          "ds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%8\n\t"
          "ds_read_b128 %2, %6 offset:%8\n\tds_read_b128 %3, %7 offset:%8\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(bf[0]), "=&v"(bf[1]), "=&v"(bf[2]), "=&v"(bf[3])
          : "v"(rd[0]), "v"(rd[1]), "v"(rd[2]), "v"(rd[3]), "n"(s * SEGB)
          : "memory");
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc4[u][i] = lmh_mfma<T>(af[s][u][i], bf[u], acc4[u][i]);
      // the slot is free (its four reads have returned): refill it with the same segment of the next tile
      __builtin_amdgcn_sched_barrier(0);
      if (more) stage(t + tstep, S_);
      __builtin_amdgcn_sched_barrier(0);
    });
    // ---- K reduction over the waves: D[token 4 g + r][column c] -------------------------------------------------------
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = (acc4[0][i] + acc4[1][i]) + (acc4[2][i] + acc4[3][i]);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 16 * MT + 16 * i + 4 * g + r) * RP + c] = acc[i][r];
    lmh_lds_barrier();
    if (tok < 16 * MT) {
      float sum = red[tok * RP + col];
#pragma unroll
      for (int w2 = 1; w2 < NWV; ++w2) sum += red[(w2 * 16 * MT + tok) * RP + col];
      const typename T::storage bits = T::from_f32(sum);
      // (a ragged last tile was loaded from rows V-16 .. V-1: column `col` holds row V - 16 + col)
      const int row0 = min(t * 16, p.V - 16);
      const int vrow = row0 + col;
      const bool fresh = vrow >= t * 16;            // not already covered by the previous tile
      if (tok < p.M && fresh) {
        if (p.logits) p.logits[(size_t)tok * p.ldl + vrow] = bits;
        const float v = T::to_f32(bits);
        if (lmh_better(v, vrow, best, best_i)) { best = v; best_i = vrow; }
      }
    }
    lmh_lds_barrier();
  }

  // ---- the workgroup's best per token -> partials; last arriver reduces --------------------------------------------------
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(best_i, o, 64);
    if (lmh_better(ob, oi, best, best_i)) { best = ob; best_i = oi; }
  }
  if (col == 0 && tok < 32) {
    float* pv = p.part_val + (size_t)blockIdx.x * 32 + tok;
    int* pi = p.part_idx + (size_t)blockIdx.x * 32 + tok;
    asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(pv), "v"(best) : "memory");
    asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(pi), "v"(best_i) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(p.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = old == (unsigned)G - 1u;
  }
  __syncthreads();
  if (!last_flag) return;
  if (tid == 0) __hip_atomic_store(p.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // 16 threads per token, each over every 16th workgroup's partial (coherent loads: the partials come from other XCDs)
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int wg = col; wg < G; wg += 16) {
    const float v = __hip_atomic_load(p.part_val + (size_t)wg * 32 + tok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int i2 = __hip_atomic_load(p.part_idx + (size_t)wg * 32 + tok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (lmh_better(v, i2, bv, bi)) { bv = v; bi = i2; }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    const float ob = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (lmh_better(ob, oi, bv, bi)) { bv = ob; bi = oi; }
  }
  if (col == 0 && tok < p.M) p.out_ids[tok] = bi == 0x7fffffff ? 0 : bi;
}

}  // namespace aphro

using namespace aphro;

// Per-device scratch of the argmax reduce: one ticket (zero between launches) + [grid][32] partial (value, index).
// hipMalloc'ed on first use, never during a stream capture.  Launches on one device are stream-ordered (see
// paged_attention.hip, split_workspace).
struct LmHeadWs { unsigned* counter = nullptr; float* val = nullptr; int* idx = nullptr; };
static LmHeadWs g_lmh_ws[APHRO_MAX_DEVICES];
static constexpr int LMH_MAX_GRID = 1024;

static LmHeadWs* lmh_workspace(hipStream_t st) {
  static std::mutex mu;                        // host threads racing on the first call allocate once
  std::lock_guard<std::mutex> lock(mu);
  LmHeadWs& ws = g_lmh_ws[device_slot()];
  if (ws.counter == nullptr) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    char* base = nullptr;
    const size_t bytes = 256 + (size_t)LMH_MAX_GRID * 32 * 8;
    if (hipMalloc((void**)&base, bytes) != hipSuccess) return nullptr;
    if (hipMemset(base, 0, bytes) != hipSuccess) { (void)hipFree(base); return nullptr; }
    ws.counter = (unsigned*)base;
    ws.val = (float*)(base + 256);
    ws.idx = (int*)(base + 256 + (size_t)LMH_MAX_GRID * 32 * 4);
  }
  return &ws;
}

extern "C" int aphro_lm_head_argmax_supported(int64_t M, int64_t K, int64_t V, int64_t ldw, int dtype) {
  if (dtype != APHRO_F16 && dtype != APHRO_BF16) return 0;
  if (M < 1 || M > 32 || V < 16 || K % 1024 != 0 || K < 1024 || K > 4096) return 0;
  if (ldw < K || ldw % 8 != 0 || ((size_t)(V - 1) * ldw + K) * 2 >= 0x7fffffffull) return 0;
  return 1;
}

extern "C" int aphro_lm_head_argmax(const void* hidden, int64_t lda, const void* weight, int64_t ldw, void* logits,
                                    int64_t ldl, int64_t* out_ids, int64_t M, int64_t K, int64_t V, int dtype,
                                    void* stream) {
  hipStream_t st = (hipStream_t)stream;
  APHRO_CHECK(aphro_lm_head_argmax_supported(M, K, V, ldw, dtype), "lm_head_argmax: M=%ld K=%ld V=%ld ldw=%ld dtype=%d is not served",
              (long)M, (long)K, (long)V, (long)ldw, dtype);
  APHRO_CHECK(((uintptr_t)hidden % 16) == 0 && ((uintptr_t)weight % 16) == 0 && lda % 8 == 0 && lda >= K && out_ids != nullptr,
              "lm_head_argmax: 16-byte aligned operands with lda %% 8 == 0 required");
  APHRO_CHECK(logits == nullptr || ldl >= V, "lm_head_argmax: logits row pitch %ld < V", (long)ldl);
  LmHeadWs* ws = lmh_workspace(st);
  if (ws == nullptr) {
    set_error("lm_head_argmax: scratch not allocated (first call under a stream capture)");
    return APHRO_ERR_WORKSPACE;
  }
  LmHeadParams p;
  p.a = (const uint16_t*)hidden; p.w = (const uint16_t*)weight; p.logits = (uint16_t*)logits; p.out_ids = out_ids;
  p.part_val = ws->val; p.part_idx = ws->idx; p.counter = ws->counter;
  p.M = (int)M; p.K = (int)K; p.V = (int)V; p.lda = (int)lda; p.ldw = (int)ldw; p.ldl = (int)ldl;
  p.tiles = (int)((V + 15) / 16);
  int grid = device_cu_count();
  if (grid > p.tiles) grid = p.tiles;
  if (grid > LMH_MAX_GRID) grid = LMH_MAX_GRID;
  const int ks = (int)(K / 1024), mt = M > 16 ? 2 : 1;
  const size_t lds = (size_t)8 * ks * 4096 + (size_t)8 * 16 * mt * 17 * sizeof(float);
#define LMH(TT, MTV, KSV)                                                                                          \
  {                                                                                                                \
    auto kern = lm_head_argmax_kernel<TT, MTV, KSV>;                                                               \
    if (lds > 64 * 1024 &&                                                                                         \
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
      set_error("lm_head_argmax: cannot raise the dynamic LDS limit to %zu", lds);                                 \
      return APHRO_ERR_LAUNCH;                                                                                     \
    }                                                                                                              \
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, p);                                         \
  }
#define LMH_KS(TT, MTV)                               \
  switch (ks) {                                       \
    case 1: LMH(TT, MTV, 1) break;                    \
    case 2: LMH(TT, MTV, 2) break;                    \
    case 3: LMH(TT, MTV, 3) break;                    \
    default: LMH(TT, MTV, 4) break;                   \
  }
  if (dtype == APHRO_F16) { if (mt == 2) LMH_KS(Half, 2) else LMH_KS(Half, 1) }
  else { if (mt == 2) LMH_KS(BFloat, 2) else LMH_KS(BFloat, 1) }
#undef LMH_KS
#undef LMH
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
