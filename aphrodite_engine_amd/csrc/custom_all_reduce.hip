// Tensor-parallel sum all-reduce over xGMI peer access -- the `_C_custom_ar::*` role (SURVEY a14;
// reference: kernels/custom_all_reduce.cu + custom_all_reduce.cuh, compiled OUT of its ROCm build,
// torch_bindings.cpp:506-536).  One process per GPU; every rank maps every peer's buffers through
// HIP IPC and reads them directly over the point-to-point xGMI links.
//
// Shapes on the hot path: [M, hidden] f16/bf16 after o_proj and down_proj -- 256 KiB (8B, bs 32) to
// 1 MiB (70B, bs 64), twice per layer.  At these sizes a ring is latency bound (2(N-1) hops); here
//   one-shot  : every rank reads all N inputs and sums them itself      (1 hop,  N x bytes per rank)
//   two-shot  : reduce-scatter then all-gather through peer reads       (2 hops, 2 x bytes per rank)
// xGMI is point to point (one link per peer pair), so in one-shot every link carries `bytes` once in
// each direction concurrently -- the cost is one link transfer, not N.
//
// MI355X-specific decisions (none of this is in the reference's CUDA design):
//  * flags live in UNCACHED fine-grained device memory (hipDeviceMallocUncached) and are accessed
//    with system-scope relaxed atomics: no release/acquire fences, because a system-scope release on
//    gfx9 writes back the whole L2 (measured: +16 us per launch for an agent-scope fence);
//    ordering comes from `s_waitcnt vmcnt(0)` between the data accesses and the flag store;
//  * peer data is read with `sc0 sc1` (system-coherent) loads so that a line cached by an earlier
//    call can never be served stale; two-shot partial sums go to an uncached (MTYPE UC) region, whose
//    stores bypass the L2, and are complete once `s_waitcnt vmcnt(0)` returns;
//  * the sum runs over ranks 0..N-1 in the same order on every rank (fp32 accumulate): all ranks
//    hold bit-identical results, which the greedy-decode parity tests rely on;
//  * barriers spin a bounded number of times and then raise an error word the host can read
//    (aphro_custom_ar_error) instead of hanging the GPU;
//  * per-block call counters live on the device, so a captured launch replays correctly in a HIP graph;
//  * the sum is followed, in every decoder layer, by residual add + RMSNorm (+ the pack of the next GEMM's A operand):
//    aphro_custom_ar_fused_add_rms_norm runs them in the all-reduce launch.  One-shot sizes:
//    one workgroup per token reads the token's row from every rank and finishes the norm on it.  Two-shot sizes: the
//    reduce-scatter is by COLUMN SLICE of every row, and the workgroup of a row gathers the slices and normalises it.
//    Bits: those of all_reduce followed by aphro_fused_add_rms_norm_pack;
//  * a LOOPBACK communicator (aphro_custom_ar_init_loopback) runs the same kernels with every "peer" pointing at this
//    rank's own buffers: bench.py --sim-tp times one rank of a TP group on a one-GPU box with the real instruction
//    stream, flags and scratch traffic (local memory instead of links), not a stream-holding stub.
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace aphro {

constexpr int AR_MAX_RANKS = 8;
constexpr int AR_MAX_BLOCKS = 64;
constexpr int AR_THREADS = 512;
constexpr int64_t AR_TICKS_PER_MS = 100000;   // wall_clock64(): 100 MHz constant clock

struct ArSignal {                       // one per rank, IPC-mapped by every peer
  uint32_t start[AR_MAX_BLOCKS][AR_MAX_RANKS];   // start[b][r] written by rank r's block b
  uint32_t mid[AR_MAX_BLOCKS][AR_MAX_RANKS];
  uint32_t end[AR_MAX_BLOCKS][AR_MAX_RANKS];
  uint32_t counter[AR_MAX_BLOCKS];      // local: calls so far, per block
  uint32_t error;                       // local: a barrier timed out
  uint32_t pad[63];
};

struct ArPeers {                        // one logical buffer as seen from this rank (device memory)
  const void* ptr[AR_MAX_RANKS];
};

struct ArParams {
  ArSignal* sig[AR_MAX_RANKS];          // every rank's signal area (own one included)
  const ArPeers* in;                    // inputs of all ranks
  ArPeers scratch;                      // two-shot partial sums of all ranks (uncached)
  void* out;
  int64_t nvec;                         // 16-byte vectors
  int64_t timeout_ticks;                // a barrier waits this long for a peer, then raises `error`
  int rank, world;
  int loopback;                         // every peer is this rank itself (one-GPU timing rig): flags go to the own columns
  int use_inline;                       // loopback: the input table travels in the kernel arguments (capture-safe)
  ArPeers in_inline;
};

__device__ __forceinline__ void st_sys(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint32_t ld_sys(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// System-coherent (sc0 sc1) 16-byte loads.  The wait for the data sits INSIDE the asm statement: an
// asm output counts as available the moment the statement ends, so with a separate `s_waitcnt` the
// compiler is free to move the destination registers before the data has arrived (seen: address
// arithmetic left in the low half of element 0).
__device__ __forceinline__ u32x4 ld_peer(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
// one input vector from every rank, all loads in flight together (one per xGMI link), one wait
template <int W>
__device__ __forceinline__ void ld_peers(u32x4 (&v)[W], const struct ArPeers& in, int64_t i);
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// All ranks' block `b` meet.  which: 0 start, 1 mid, 2 end.  `val` is this call's ticket.
__device__ __forceinline__ void ar_barrier(const ArParams& p, int which, uint32_t val) {
  wait_vm();                     // my earlier loads / stores have completed before I tell anyone
  __syncthreads();
  const int b = blockIdx.x;
  if ((int)threadIdx.x < p.world) {
    const int r = threadIdx.x;
    ArSignal* peer = p.sig[r];
    ArSignal* self = p.sig[p.rank];
    const int col = p.loopback ? r : p.rank;
    uint32_t* dst = which == 0 ? &peer->start[b][col] : which == 1 ? &peer->mid[b][col] : &peer->end[b][col];
    const uint32_t* src = which == 0 ? &self->start[b][r] : which == 1 ? &self->mid[b][r] : &self->end[b][r];
    st_sys(dst, val);
    const int64_t t0 = (int64_t)wall_clock64();
    while ((int32_t)(ld_sys(src) - val) < 0) {            // wrap-safe "not yet"
      __builtin_amdgcn_s_sleep(2);
      if ((int64_t)wall_clock64() - t0 > p.timeout_ticks) {   // bounded: an absent peer is an error, not a hang
        st_sys(&self->error, 1u);
        break;
      }
    }
  }
  __syncthreads();
}

template <typename T>
__device__ __forceinline__ void acc8(float (&a)[8], u32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[2 * i] += T::to_f32((uint16_t)(v[i] & 0xffffu));
    a[2 * i + 1] += T::to_f32((uint16_t)(v[i] >> 16));
  }
}
template <>
__device__ __forceinline__ void acc8<Float>(float (&a)[8], u32x4 v) {
  // whole-vector bit cast: casting ONE element of an ext-vector reads element [0] (hipcc bug)
  const f32x4 f = __builtin_bit_cast(f32x4, v);
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] += f[i];
}
template <typename T>
__device__ __forceinline__ u32x4 pack8(const float (&a)[8]) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    r[i] = (uint32_t)T::from_f32(a[2 * i]) | ((uint32_t)T::from_f32(a[2 * i + 1]) << 16);
  return r;
}
template <>
__device__ __forceinline__ u32x4 pack8<Float>(const float (&a)[8]) {
  const f32x4 f = {a[0], a[1], a[2], a[3]};
  return __builtin_bit_cast(u32x4, f);
}

template <>
__device__ __forceinline__ void ld_peers<2>(u32x4 (&v)[2], const ArPeers& in, int64_t i) {
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1])
                 : "v"((const u32x4*)in.ptr[0] + i), "v"((const u32x4*)in.ptr[1] + i)
                 : "memory");
}
template <>
__device__ __forceinline__ void ld_peers<4>(u32x4 (&v)[4], const ArPeers& in, int64_t i) {
    asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\tglobal_load_dwordx4 %2, %6, off sc0 sc1\n\tglobal_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                 : "v"((const u32x4*)in.ptr[0] + i), "v"((const u32x4*)in.ptr[1] + i), "v"((const u32x4*)in.ptr[2] + i), "v"((const u32x4*)in.ptr[3] + i)
                 : "memory");
}
template <>
__device__ __forceinline__ void ld_peers<6>(u32x4 (&v)[6], const ArPeers& in, int64_t i) {
    asm volatile("global_load_dwordx4 %0, %6, off sc0 sc1\n\tglobal_load_dwordx4 %1, %7, off sc0 sc1\n\tglobal_load_dwordx4 %2, %8, off sc0 sc1\n\tglobal_load_dwordx4 %3, %9, off sc0 sc1\n\tglobal_load_dwordx4 %4, %10, off sc0 sc1\n\tglobal_load_dwordx4 %5, %11, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
                 : "v"((const u32x4*)in.ptr[0] + i), "v"((const u32x4*)in.ptr[1] + i), "v"((const u32x4*)in.ptr[2] + i), "v"((const u32x4*)in.ptr[3] + i), "v"((const u32x4*)in.ptr[4] + i), "v"((const u32x4*)in.ptr[5] + i)
                 : "memory");
}
template <>
__device__ __forceinline__ void ld_peers<8>(u32x4 (&v)[8], const ArPeers& in, int64_t i) {
    asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %9, off sc0 sc1\n\tglobal_load_dwordx4 %2, %10, off sc0 sc1\n\tglobal_load_dwordx4 %3, %11, off sc0 sc1\n\tglobal_load_dwordx4 %4, %12, off sc0 sc1\n\tglobal_load_dwordx4 %5, %13, off sc0 sc1\n\tglobal_load_dwordx4 %6, %14, off sc0 sc1\n\tglobal_load_dwordx4 %7, %15, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"((const u32x4*)in.ptr[0] + i), "v"((const u32x4*)in.ptr[1] + i), "v"((const u32x4*)in.ptr[2] + i), "v"((const u32x4*)in.ptr[3] + i), "v"((const u32x4*)in.ptr[4] + i), "v"((const u32x4*)in.ptr[5] + i), "v"((const u32x4*)in.ptr[6] + i), "v"((const u32x4*)in.ptr[7] + i)
                 : "memory");
}

template <typename T, int WORLD>
__device__ __forceinline__ u32x4 reduce_vec(const ArPeers& in, int64_t i) {
  u32x4 v[WORLD];
  ld_peers<WORLD>(v, in, i);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < WORLD; ++r) acc8<T>(a, v[r]);    // rank order: identical bits on every rank
  return pack8<T>(a);
}

template <typename T, int WORLD>
__global__ __launch_bounds__(AR_THREADS) void ar_one_shot_kernel(ArParams p) {
  __shared__ uint32_t ticket;
  if (threadIdx.x == 0) {
    ArSignal* self = p.sig[p.rank];
    ticket = self->counter[blockIdx.x] + 1;
    self->counter[blockIdx.x] = ticket;
  }
  __syncthreads();
  const uint32_t val = ticket;
  ar_barrier(p, 0, val);                                  // every peer's input is in place
  const ArPeers in = p.use_inline ? p.in_inline : *p.in;
  for (int64_t i = (int64_t)blockIdx.x * AR_THREADS + threadIdx.x; i < p.nvec; i += (int64_t)gridDim.x * AR_THREADS)
    ((u32x4*)p.out)[i] = reduce_vec<T, WORLD>(in, i);
  ar_barrier(p, 2, val);                                  // nobody still reads my input when I return
}

template <typename T, int WORLD>
__global__ __launch_bounds__(AR_THREADS) void ar_two_shot_kernel(ArParams p) {
  __shared__ uint32_t ticket;
  if (threadIdx.x == 0) {
    ArSignal* self = p.sig[p.rank];
    ticket = self->counter[blockIdx.x] + 1;
    self->counter[blockIdx.x] = ticket;
  }
  __syncthreads();
  const uint32_t val = ticket;
  const int64_t part = (p.nvec + WORLD - 1) / WORLD;
  const int64_t lo = part * p.rank, hi = lo + part < p.nvec ? lo + part : p.nvec;
  ar_barrier(p, 0, val);
  const ArPeers in = p.use_inline ? p.in_inline : *p.in;
  u32x4* mine = (u32x4*)p.scratch.ptr[p.rank];
  for (int64_t i = lo + (int64_t)blockIdx.x * AR_THREADS + threadIdx.x; i < hi; i += (int64_t)gridDim.x * AR_THREADS) {
    const u32x4 s = reduce_vec<T, WORLD>(in, i);
    ((u32x4*)p.out)[i] = s;
    mine[i] = s;        // uncached (MTYPE UC) memory: the store goes through to HBM; peers gather it in shot 2
  }
  ar_barrier(p, 1, val);                                  // (waits for the write-through stores first)
#pragma unroll
  for (int rr = 1; rr < WORLD; ++rr) {
    const int r = (p.rank + rr) % WORLD;                  // start with a different peer on every rank
    const int64_t rlo = part * r, rhi = rlo + part < p.nvec ? rlo + part : p.nvec;
    const u32x4* theirs = (const u32x4*)p.scratch.ptr[r];
    for (int64_t i = rlo + (int64_t)blockIdx.x * AR_THREADS + threadIdx.x; i < rhi; i += (int64_t)gridDim.x * AR_THREADS) {
      ((u32x4*)p.out)[i] = ld_peer(theirs + i);
    }
  }
  ar_barrier(p, 2, val);
}


// ---------------------------------------------------------------------------------------------------
// all-reduce + residual add + RMSNorm (+ pack) in one launch.  Arithmetic = ar_*_kernel followed by
// add_rms_norm_pack_kernel (fused_decode.hip) on its `input` path: x = round_T(sum over ranks, fp32, rank order);
// residual' = round_T(x + residual); y = round_T(round_T(residual' * rstd) * w); same thread -> element mapping, same
// block size and the same reduction order as that kernel, so the bits agree.
struct ArNormParams {
  ArParams ar;
  uint16_t* residual;                   // [tokens, hidden] T, updated in place (may be null when !has_residual)
  const uint16_t* weight;               // [hidden] T
  uint16_t* packed;                     // fragment-major f16 A operand of the next GEMM, or null
  uint16_t* out;                        // row-major [tokens, hidden] T, or null
  float eps;
  int has_residual;
  int tokens, hidden;
  int rows_per_rank;                    // two-shot: vectors per column slice, ceil(hidden / 8 / world)
  int replicate_residual;               // (unused)
  // Workgroups nb .. gridDim.x - 1 take no part in the sum: they pull `pf_n16` 16-byte vectors at `pf` -- the packed weights
  // of the GEMM that follows the norm -- through the memory-side Infinity Cache while the nb reducing workgroups wait on
  // flags and xGMI round trips (the sum moves a few hundred KiB and is latency bound: HBM is idle).  The all-reduce
  // overlapped with the next GEMM's weight stream inside ONE launch, no stream fork / join (BASELINE north_star; the
  // side-stream form of round 3, distributed/overlap.py, measured 38-61 % SLOWER under a graph).
  const u32x4* pf;
  size_t pf_n16;
  int nb;
  // Q8 kernels (FP8 W8A8 layers under TP): the normalised row also leaves as e4m3 with its per-token scale -- the bits of
  // add_rms_norm_quant_kernel (fused_fp8.hip) on the all-reduced input: scale = max(absmax(y) / 448, 1 / (448 * 512)),
  // q = fp8(y / scale); with q8_static: q = fp8(y * (1 / *q8_static)), scale_out = *q8_static.
  uint8_t* q8_out;                      // [tokens, hidden] e4m3
  float* q8_scale_out;                  // [tokens]
  const float* q8_static;               // [1] or null
  // ROUTER kernels (sparse-MLP layers under TP: the attention block's all-reduce in front of the router norm): the router's
  // logits of the row, round_T(y . router_w[e]) for e < num_experts <= 16 -- the bits of add_rms_norm_pack_kernel<T, ROUTER>
  // (fused_decode.hip) on the all-reduced input; the normalised row leaves row-major through `out`.
  const uint16_t* router_w;             // [num_experts, hidden] T
  uint16_t* router_out;                 // [tokens, num_experts] T
  int num_experts;
};

constexpr int AR_EPI_NONE = 0, AR_EPI_Q8 = 1, AR_EPI_ROUTER = 2;

__device__ __forceinline__ void ar_prefetch_role(const ArNormParams& q) {
  u32x4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)(gridDim.x - q.nb) * blockDim.x;
  size_t i = (size_t)(blockIdx.x - q.nb) * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < q.pf_n16; i += 4 * stride) {
    const u32x4 a = q.pf[i], b = q.pf[i + stride], c = q.pf[i + 2 * stride], d = q.pf[i + 3 * stride];
    acc ^= a ^ b ^ c ^ d;
  }
  for (; i < q.pf_n16; i += stride) acc ^= q.pf[i];
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u && q.out == (uint16_t*)1) *q.out = 1;   // keeps the loads alive
}

__device__ __forceinline__ float ar_block_sum(float v, float* red) {      // == block_sum_f of fused_decode.hip
  v = wave_sum(v);
  const int nw = blockDim.x >> 6;
  if (nw == 1) return v;
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < nw; ++w) t += red[w];
  __syncthreads();
  return t;
}
__device__ __forceinline__ size_t ar_packed_chunk(int row, int k, int mtiles) {   // == packed_chunk of fused_decode.hip
  const int seg = k >> 7, g = (k & 127) >> 5, u = (k & 31) >> 3;
  const int mt = row >> 4;
  return ((((size_t)seg * 4 + u) * mtiles + mt) * 64 + g * 16 + (row & 15)) * 8;
}
template <typename T>
__device__ __forceinline__ uint16_t ar_to_f16_bits(uint16_t tbits) {
  if constexpr (__is_same(T, Half)) return tbits;
  else return bf16_bits_to_f16_bits_sat(tbits);
}

// One row of the norm on the block: xs[it] = the T-rounded sums of this thread's (up to) two 8-element chunks.  Returns
// the normalised chunks in y[it]; writes residual' to `res_row` (if non-null) and, when `res_pub` is non-null, to that
// second place too (the two-shot scratch).
template <typename T>
__device__ __forceinline__ void ar_norm_row(const u32x4 (&xs)[2], const uint16_t* __restrict__ res_in, uint16_t* res_row,
                                            uint16_t* res_pub, const u16x8 (&wv)[2], int nv, int hidden, float eps,
                                            float* red, u16x8 (&y)[2]) {
  float v[2][8];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) {
      const u16x8 a = __builtin_bit_cast(u16x8, xs[it]);
      u16x8 rs;
      if (res_in) {
        const u16x8 r = *reinterpret_cast<const u16x8*>(res_in + 8 * i);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          rs[j] = T::from_f32(T::to_f32(a[j]) + T::to_f32(r[j]));
          v[it][j] = T::to_f32(rs[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          rs[j] = a[j];
          v[it][j] = T::to_f32(a[j]);
        }
      }
      if (res_row) *reinterpret_cast<u16x8*>(res_row + 8 * i) = rs;
      if (res_pub) *reinterpret_cast<u16x8*>(res_pub + 8 * i) = rs;
      {
        // squares rounded, then added (what hipcc emits for add_rms_norm_pack_kernel's input path: v_pk_mul_f32 + adds)
#pragma clang fp contract(off)
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += v[it][j] * v[it][j];
      }
    }
  }
  ss = ar_block_sum(ss, red);
  const float inv = __frsqrt_rn(ss / (float)hidden + eps);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        y[it][j] = T::from_f32(T::to_f32(from_f32_exact<T>(v[it][j] * inv)) * T::to_f32(wv[it][j]));
    }
  }
}

template <typename T>
__device__ __forceinline__ void ar_norm_emit(const ArNormParams& q, int row, int i, u16x8 y) {
  if (q.out) *reinterpret_cast<u16x8*>(q.out + (size_t)row * q.hidden + 8 * i) = y;
  if (q.packed) {
    u16x8 yh;
#pragma unroll
    for (int j = 0; j < 8; ++j) yh[j] = ar_to_f16_bits<T>(y[j]);
    *reinterpret_cast<u16x8*>(q.packed + ar_packed_chunk(row, 8 * i, (q.tokens + 15) >> 4)) = yh;
  }
}

// The FP8 epilogue of a Q8 launch: y[it] = this thread's normalised chunks of row `row` (ar_norm_row).
template <typename T>
__device__ __forceinline__ void ar_norm_quant(const ArNormParams& q, int row, const u16x8 (&y)[2], int nv, float* red) {
  constexpr float Q8_MAX = 448.f;
  float v[2][8];
  float amax = 0.f;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[it][j] = T::to_f32(y[it][j]);
        amax = __builtin_fmaxf(amax, __builtin_fabsf(v[it][j]));
      }
    }
  }
  float scale, inv_static = 0.f;
  if (q.q8_static) {
    scale = *q.q8_static;
    inv_static = 1.0f / scale;
  } else {
    amax = wave_max(amax);
    const int nw = blockDim.x >> 6;
    if (nw > 1) {                                          // (red[] is free: ar_block_sum ended on a barrier)
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
      __syncthreads();
      float t = red[0];
      for (int w = 1; w < nw; ++w) t = __builtin_fmaxf(t, red[w]);
      amax = t;
    }
    scale = __builtin_fmaxf(amax / Q8_MAX, 1.0f / (Q8_MAX * 512.f));
  }
  if (threadIdx.x == 0) q.q8_scale_out[row] = scale;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) {
      float a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float s = q.q8_static ? v[it][j] * inv_static : v[it][j] / scale;
        a[j] = __builtin_fmaxf(-Q8_MAX, __builtin_fminf(s, Q8_MAX));
      }
      int lo = __builtin_amdgcn_cvt_pk_fp8_f32(a[0], a[1], 0, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(a[2], a[3], lo, true);
      int hi = __builtin_amdgcn_cvt_pk_fp8_f32(a[4], a[5], 0, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(a[6], a[7], hi, true);
      *reinterpret_cast<u32x2*>(q.q8_out + (size_t)row * q.hidden + 8 * i) = u32x2{(uint32_t)lo, (uint32_t)hi};
    }
  }
}

// The router epilogue of a ROUTER launch: same thread -> element mapping, accumulation order, reduce-scatter butterfly and
// cross-wave sum as add_rms_norm_pack_kernel<T, ROUTER = true> (fused_decode.hip).
// The first eight router rows of a thread's chunks: requested with the norm weights, in flight across the first barrier (a
// serial L2 round trip behind the norm otherwise: 8.1 -> 7.4 us per launch at 4 ranks, [32, 4096]).
template <bool ON>
__device__ __forceinline__ void ar_router_prefetch(const ArNormParams& q, int nv, u16x8 (&g8v)[2][8]) {
  if constexpr (ON) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = threadIdx.x + it * blockDim.x;
      if (i < nv) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < q.num_experts) g8v[it][e] = *reinterpret_cast<const u16x8*>(q.router_w + (size_t)e * q.hidden + 8 * i);
      }
    }
  }
}

// PRE: g8v holds the prefetched rows (one-shot kernel).  The two-shot kernel (1024-thread workgroups at hidden 8192: 128
// VGPRs) loads them here -- with the prefetch it measured 13.2 us against 10.6 at 8 ranks, [64, 8192].
template <typename T, bool PRE>
__device__ __forceinline__ void ar_norm_router(const ArNormParams& q, int row, const u16x8 (&y)[2], int nv, float* rred,
                                               const u16x8 (&g8v)[PRE ? 2 : 1][8]) {
  float rpart[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) rpart[e] = 0.f;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (e < q.num_experts) {
          u16x8 g8;
          if (PRE && e < 8) g8 = g8v[PRE ? it : 0][e < 8 ? e : 0];
          else g8 = *reinterpret_cast<const u16x8*>(q.router_w + (size_t)e * q.hidden + 8 * i);
          {
#pragma clang fp contract(off)
#pragma unroll
            for (int j = 0; j < 8; ++j) rpart[e] += T::to_f32(y[it][j]) * T::to_f32(g8[j]);
          }
        }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = (blockDim.x + 63) >> 6;
#define AR_RS_STEP(O, HALF)                                           \
  {                                                                   \
    const bool up = (lane & O) != 0;                                  \
    _Pragma("unroll") for (int j = 0; j < HALF; ++j) {                \
      const float send = up ? rpart[j] : rpart[j + HALF];             \
      const float keep = up ? rpart[j + HALF] : rpart[j];             \
      rpart[j] = keep + __shfl_xor(send, O, 64);                      \
    }                                                                 \
  }
  AR_RS_STEP(32, 8)
  AR_RS_STEP(16, 4)
  AR_RS_STEP(8, 2)
  AR_RS_STEP(4, 1)
#undef AR_RS_STEP
  float v2 = rpart[0];
  v2 += __shfl_xor(v2, 2, 64);
  v2 += __shfl_xor(v2, 1, 64);
  if ((lane & 3) == 0) rred[wave * 16 + ((lane >> 2) & 15)] = v2;
  __syncthreads();
  if ((int)threadIdx.x < q.num_experts) {
    float sum = 0.f;
    for (int w2 = 0; w2 < nwave; ++w2) sum += rred[w2 * 16 + threadIdx.x];
    q.router_out[(size_t)row * q.num_experts + threadIdx.x] = T::from_f32(sum);
  }
}

template <typename T, int WORLD, int EPI>
__global__ __launch_bounds__(1024) void ar_norm_one_shot_kernel(ArNormParams q) {
  __shared__ uint32_t ticket;
  __shared__ float red[16];
  __shared__ float rred[EPI == AR_EPI_ROUTER ? 16 * 16 : 1];
  if ((int)blockIdx.x >= q.nb) { ar_prefetch_role(q); return; }      // (block-uniform)
  const ArParams& p = q.ar;
  if (threadIdx.x == 0) {
    ArSignal* self = p.sig[p.rank];
    ticket = self->counter[blockIdx.x] + 1;
    self->counter[blockIdx.x] = ticket;
  }
  const int tok = blockIdx.x;
  const int nv = q.hidden >> 3;
  u16x8 wv[2];                                             // norm weights: in flight across the barrier
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) wv[it] = *reinterpret_cast<const u16x8*>(q.weight + 8 * i);
  }
  u16x8 g8v[EPI == AR_EPI_ROUTER ? 2 : 1][8];
  if constexpr (EPI == AR_EPI_ROUTER) ar_router_prefetch<true>(q, nv, g8v);
  __syncthreads();
  const uint32_t val = ticket;
  ar_barrier(p, 0, val);                                   // every peer's input is in place
  const ArPeers in = p.use_inline ? p.in_inline : *p.in;
  u32x4 xs[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) xs[it] = reduce_vec<T, WORLD>(in, (int64_t)tok * nv + i);
  }
  uint16_t* res_row = q.residual ? q.residual + (size_t)tok * q.hidden : nullptr;
  u16x8 y[2];
  ar_norm_row<T>(xs, q.has_residual ? res_row : nullptr, res_row, nullptr, wv, nv, q.hidden, q.eps, red, y);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) ar_norm_emit<T>(q, tok, i, y[it]);
  }
  if constexpr (EPI == AR_EPI_Q8) ar_norm_quant<T>(q, tok, y, nv, red);
  if constexpr (EPI == AR_EPI_ROUTER) ar_norm_router<T, true>(q, tok, y, nv, rred, g8v);
  ar_barrier(p, 2, val);                                   // nobody still reads my input when I return
}

// Two-shot sizes: reduce-scatter by COLUMN SLICE, norm after the gather.  Rank r sums vectors [r S, (r + 1) S), S =
// ceil(hidden / 8 / world), of every row (workgroup b = row b, S threads busy: the same one-round-trip reduce as
// ar_two_shot_kernel, spread over `tokens` workgroups) and publishes them in its uncached scratch; after the flag round
// every workgroup gathers ITS row -- one 16-byte load per thread from the slice's owner -- and finishes residual add +
// norm + pack on it.  Every rank ends with every row (residual included), the data crossed the links twice per element
// instead of `world` times.  (Round-5 lab, loopback communicator, [64, 8192] f16 at 8 ranks: all-reduce 8.0 us + norm 4.4 us
// as two launches 12.6 us; a reduce-scatter BY ROW with the norm before the gather -- 8 workgroups of 1024 threads doing
// all of it -- 15.3 us; this form see profiles/r5_ar_norm_fused.txt.)
template <typename T, int WORLD, int EPI>
__global__ __launch_bounds__(1024) void ar_norm_two_shot_kernel(ArNormParams q) {
  __shared__ uint32_t ticket;
  __shared__ float red[16];
  __shared__ float rred[EPI == AR_EPI_ROUTER ? 16 * 16 : 1];
  if ((int)blockIdx.x >= q.nb) { ar_prefetch_role(q); return; }      // (block-uniform)
  const ArParams& p = q.ar;
  if (threadIdx.x == 0) {
    ArSignal* self = p.sig[p.rank];
    ticket = self->counter[blockIdx.x] + 1;
    self->counter[blockIdx.x] = ticket;
  }
  const int row = blockIdx.x;
  const int nv = q.hidden >> 3;
  const int nvs = q.rows_per_rank;                          // (reused field: vectors per column slice)
  u16x8 wv[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) wv[it] = *reinterpret_cast<const u16x8*>(q.weight + 8 * i);
  }
  u16x8 g8v[1][8];                                          // (not prefetched in this form: see ar_norm_router)
  __syncthreads();
  const uint32_t val = ticket;
  ar_barrier(p, 0, val);
  const ArPeers in = p.use_inline ? p.in_inline : *p.in;
  u32x4* mine = (u32x4*)p.scratch.ptr[p.rank];
  for (int i = threadIdx.x; i < nvs; i += blockDim.x) {
    const int col = p.rank * nvs + i;
    if (col < nv) mine[(size_t)row * nv + col] = reduce_vec<T, WORLD>(in, (int64_t)row * nv + col);   // uncached: through to memory
  }
  ar_barrier(p, 1, val);                                   // (waits for the write-through stores first)
  u32x4 xs[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) xs[it] = ld_peer((const u32x4*)p.scratch.ptr[i / nvs] + (size_t)row * nv + i);
  }
  uint16_t* res_row = q.residual ? q.residual + (size_t)row * q.hidden : nullptr;
  u16x8 y[2];
  ar_norm_row<T>(xs, q.has_residual ? res_row : nullptr, res_row, nullptr, wv, nv, q.hidden, q.eps, red, y);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nv) ar_norm_emit<T>(q, row, i, y[it]);
  }
  if constexpr (EPI == AR_EPI_Q8) ar_norm_quant<T>(q, row, y, nv, red);
  if constexpr (EPI == AR_EPI_ROUTER) ar_norm_router<T, false>(q, row, y, nv, rred, g8v);
  ar_barrier(p, 2, val);
}

// ---------------------------------------------------------------------------------------------------
struct IpcKey {
  char b[sizeof(hipIpcMemHandle_t)];
  bool operator<(const IpcKey& o) const { return memcmp(b, o.b, sizeof(b)) < 0; }
};

struct CustomAr {
  int rank = 0, world = 0;
  ArSignal* sig[AR_MAX_RANKS] = {};
  ArPeers scratch = {};
  size_t scratch_bytes = 0;
  ArPeers* d_slots = nullptr;           // device array of registered buffers' peer pointers
  int slot_cap = 0, slot_used = 0;
  std::map<const void*, int> registered;            // local base pointer -> slot
  std::vector<const void*> graph_unreg;             // inputs seen while capturing, slots reserved in order
  std::map<IpcKey, void*> opened;
  void* own_signal = nullptr;
  void* own_scratch = nullptr;
  int64_t timeout_ticks = 10000 * AR_TICKS_PER_MS;   // APHRODITE_CUSTOM_AR_TIMEOUT_MS, default 10 s
  bool loopback = false;                             // aphro_custom_ar_init_loopback: every peer is this rank

  void* open_peer(const char* handle) {
    IpcKey k;
    memcpy(k.b, handle, sizeof(k.b));
    auto it = opened.find(k);
    if (it != opened.end()) return it->second;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    opened[k] = p;
    return p;
  }
};

static int fill_peers(CustomAr* fa, const void* local, const char* handles, const int64_t* offsets, ArPeers* out) {
  for (int r = 0; r < fa->world; ++r) {
    if (r == fa->rank) {
      out->ptr[r] = local;
    } else {
      char* base = (char*)fa->open_peer(handles + (size_t)r * sizeof(hipIpcMemHandle_t));
      if (!base) {
        set_error("custom_ar: cannot open the IPC handle of rank %d", r);
        return APHRO_ERR_INVALID;
      }
      out->ptr[r] = base + offsets[r];
    }
  }
  return APHRO_OK;
}


// Resolve the registered-buffer slot of `src` (or reserve one while the stream is capturing) and fill the fields every
// kernel of this file needs.
static int ar_fill_params(CustomAr* fa, const void* src, hipStream_t st, ArParams* p) {
  memset(p, 0, sizeof(*p));
  for (int r = 0; r < AR_MAX_RANKS; ++r) p->sig[r] = fa->sig[r];
  p->scratch = fa->scratch;
  p->rank = fa->rank; p->world = fa->world;
  p->timeout_ticks = fa->timeout_ticks;
  if (fa->loopback) {
    p->loopback = 1; p->use_inline = 1;
    for (int r = 0; r < AR_MAX_RANKS; ++r) p->in_inline.ptr[r] = src;
    p->in = fa->d_slots;
    return APHRO_OK;
  }
  int slot = -1;
  auto it = fa->registered.find(src);
  if (it != fa->registered.end()) {
    slot = it->second;
  } else {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cs);
    if (cs != hipStreamCaptureStatusActive) {
      set_error("custom_ar: buffer address %p is not registered!", src);
      return APHRO_ERR_INVALID;
    }
    APHRO_CHECK(fa->slot_used + (int)fa->graph_unreg.size() < fa->slot_cap, "custom_ar: rank_data is full");
    slot = fa->slot_used + (int)fa->graph_unreg.size();    // filled by register_graph_buffers
    fa->graph_unreg.push_back(src);
  }
  p->in = fa->d_slots + slot;
  return APHRO_OK;
}

}  // namespace aphro

using namespace aphro;

extern "C" int64_t aphro_custom_ar_meta_size() { return (int64_t)sizeof(ArSignal); }

extern "C" int aphro_ipc_handle_bytes() { return (int)sizeof(hipIpcMemHandle_t); }

// Peer-visible, uncached device memory (signals, two-shot scratch): zero-filled.
extern "C" int aphro_custom_ar_alloc_shared(void** ptr, size_t bytes) {
  APHRO_CHECK(ptr != nullptr && bytes > 0, "custom_ar: bad allocation request");
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("custom_ar: hipExtMallocWithFlags(%zu, uncached) failed: %s", bytes, hipGetErrorString(e));
    return APHRO_ERR_LAUNCH;
  }
  e = hipMemset(p, 0, bytes);
  if (e != hipSuccess) {
    (void)hipFree(p);
    set_error("custom_ar: hipMemset failed: %s", hipGetErrorString(e));
    return APHRO_ERR_LAUNCH;
  }
  (void)hipDeviceSynchronize();
  *ptr = p;
  return APHRO_OK;
}

extern "C" int aphro_custom_ar_free_shared(void* ptr) {
  if (ptr) (void)hipFree(ptr);
  return APHRO_OK;
}

// IPC handle of the allocation that contains `ptr` + the offset of `ptr` inside it
// (what torch's storage._share_cuda_() hands the reference: custom_all_reduce.py:193-199).
extern "C" int aphro_ipc_get_mem_handle(const void* ptr, char* handle_out, int64_t* offset_out) {
  APHRO_CHECK(ptr && handle_out && offset_out, "ipc_get_mem_handle: NULL argument");
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  hipError_t e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("ipc_get_mem_handle: hipMemGetAddressRange failed: %s", hipGetErrorString(e));
    return APHRO_ERR_LAUNCH;
  }
  hipIpcMemHandle_t h;
  e = hipIpcGetMemHandle(&h, (void*)base);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("ipc_get_mem_handle: hipIpcGetMemHandle failed: %s", hipGetErrorString(e));
    return APHRO_ERR_LAUNCH;
  }
  memcpy(handle_out, &h, sizeof(h));
  *offset_out = (int64_t)((const char*)ptr - (const char*)base);
  return APHRO_OK;
}

// init_custom_ar: `signal` / `scratch` are this rank's aphro_custom_ar_alloc_shared areas (meta_size
// and scratch_bytes big), `*_handles` = world x aphro_ipc_handle_bytes() bytes in rank order with
// their offsets, `rank_data` = device memory for `rank_data_bytes / 64` registered-buffer slots.
extern "C" int aphro_custom_ar_init(void** fa_out, void* signal, const char* signal_handles,
                                    const int64_t* signal_offsets, void* scratch, size_t scratch_bytes,
                                    const char* scratch_handles, const int64_t* scratch_offsets, void* rank_data,
                                    size_t rank_data_bytes, int rank, int world) {
  APHRO_CHECK(fa_out && signal && scratch && rank_data, "custom_ar_init: NULL argument");
  APHRO_CHECK(world >= 2 && world <= AR_MAX_RANKS && world % 2 == 0, "custom_ar: world size %d not in {2,4,6,8}", world);
  APHRO_CHECK(rank >= 0 && rank < world, "custom_ar: invalid rank %d", rank);
  APHRO_CHECK(rank_data_bytes >= sizeof(ArPeers) && scratch_bytes % 16 == 0, "custom_ar: bad buffer sizes");
  CustomAr* fa = new CustomAr();
  fa->rank = rank; fa->world = world;
  fa->own_signal = signal; fa->own_scratch = scratch; fa->scratch_bytes = scratch_bytes;
  if (knobs().ar_timeout_ms > 0) fa->timeout_ticks = (int64_t)knobs().ar_timeout_ms * AR_TICKS_PER_MS;
  fa->d_slots = (ArPeers*)rank_data;
  fa->slot_cap = (int)(rank_data_bytes / sizeof(ArPeers));
  ArPeers sp;
  int rc = fill_peers(fa, signal, signal_handles, signal_offsets, &sp);
  if (rc == APHRO_OK) rc = fill_peers(fa, scratch, scratch_handles, scratch_offsets, &fa->scratch);
  if (rc != APHRO_OK) {
    delete fa;
    return rc;
  }
  for (int r = 0; r < world; ++r) fa->sig[r] = (ArSignal*)sp.ptr[r];
  *fa_out = fa;
  return APHRO_OK;
}

extern "C" int aphro_custom_ar_dispose(void* fa_) {
  CustomAr* fa = (CustomAr*)fa_;
  if (!fa) return APHRO_OK;
  for (auto& kv : fa->opened) (void)hipIpcCloseMemHandle(kv.second);
  delete fa;
  return APHRO_OK;
}

// register_buffer: a user buffer every rank allocated at the same point (handles of all ranks).
extern "C" int aphro_custom_ar_register_buffer(void* fa_, const void* local_ptr, const char* handles,
                                               const int64_t* offsets) {
  CustomAr* fa = (CustomAr*)fa_;
  APHRO_CHECK(fa && local_ptr && handles && offsets, "custom_ar_register_buffer: NULL argument");
  APHRO_CHECK(fa->slot_used < fa->slot_cap, "custom_ar: rank_data is full (%d buffers)", fa->slot_cap);
  ArPeers pe = {};
  int rc = fill_peers(fa, local_ptr, handles, offsets, &pe);
  if (rc != APHRO_OK) return rc;
  APHRO_CHECK(hipMemcpy(fa->d_slots + fa->slot_used, &pe, sizeof(pe), hipMemcpyHostToDevice) == hipSuccess,
              "custom_ar: hipMemcpy of the peer table failed");
  fa->registered[local_ptr] = fa->slot_used++;
  return APHRO_OK;
}

extern "C" int aphro_custom_ar_should_one_shot(int world, size_t bytes) {
  // one link transfer of `bytes` vs two transfers of bytes / world plus a second flag round trip
  // (APHRO_CUSTOM_AR_ONE_SHOT_MAX=<bytes>: override, any world size -- the tests use it to reach the two-shot forms with
  //  two ranks; must be the same on every rank)
  const long forced = knobs().ar_one_shot_max;
  if (forced >= 0) return bytes <= (size_t)forced;
  if (world <= 2) return 1;
  return bytes <= (world <= 4 ? 512u * 1024u : 256u * 1024u);
}

// all_reduce_reg (reg_buffer == NULL: `inp` itself is registered, or -- while the stream is capturing
// -- gets a slot that aphro_custom_ar_register_graph_buffers fills after the capture) and
// all_reduce_unreg (reg_buffer = a registered staging buffer: copy, then reduce).
extern "C" int aphro_custom_ar_all_reduce(void* fa_, const void* inp, void* out, int64_t numel, int dtype,
                                          void* reg_buffer, size_t reg_buffer_bytes, void* stream) {
  CustomAr* fa = (CustomAr*)fa_;
  hipStream_t st = (hipStream_t)stream;
  APHRO_CHECK(fa && inp && out, "custom_ar_all_reduce: NULL argument");
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16 || dtype == APHRO_F32, "custom_ar: unsupported dtype %d", dtype);
  const size_t esz = dtype == APHRO_F32 ? 4 : 2;
  const size_t bytes = (size_t)numel * esz;
  APHRO_CHECK(bytes % 16 == 0 && ((uintptr_t)inp % 16) == 0 && ((uintptr_t)out % 16) == 0,
              "custom all reduce currently requires input length to be multiple of 16 bytes");
  if (numel == 0) return APHRO_OK;
  const void* src = inp;
  if (reg_buffer != nullptr) {
    APHRO_CHECK(bytes <= reg_buffer_bytes, "custom_ar: registered buffer is too small (%zu > %zu)", bytes, reg_buffer_bytes);
    APHRO_CHECK(hipMemcpyAsync(reg_buffer, inp, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess,
                "custom_ar: staging copy failed");
    src = reg_buffer;
  }
  ArParams p;
  int rc = ar_fill_params(fa, src, st, &p);
  if (rc != APHRO_OK) return rc;
  const bool one_shot = aphro_custom_ar_should_one_shot(fa->world, bytes) != 0;
  APHRO_CHECK(one_shot || bytes <= fa->scratch_bytes, "custom_ar: %zu bytes exceed the two-shot scratch", bytes);
  p.out = out;
  p.nvec = (int64_t)(bytes / 16);
  int64_t work = one_shot ? p.nvec : (p.nvec + fa->world - 1) / fa->world;
  int blocks = (int)((work + AR_THREADS - 1) / AR_THREADS);
  blocks = blocks < 1 ? 1 : (blocks > AR_MAX_BLOCKS ? AR_MAX_BLOCKS : blocks);
#define AR_LAUNCH(TT, W)                                                                                     \
  {                                                                                                          \
    if (one_shot) hipLaunchKernelGGL((ar_one_shot_kernel<TT, W>), dim3(blocks), dim3(AR_THREADS), 0, st, p);   \
    else hipLaunchKernelGGL((ar_two_shot_kernel<TT, W>), dim3(blocks), dim3(AR_THREADS), 0, st, p);           \
  }
#define AR_WORLD(TT)                                                                  \
  switch (fa->world) {                                                               \
    case 2: AR_LAUNCH(TT, 2) break;                                                   \
    case 4: AR_LAUNCH(TT, 4) break;                                                   \
    case 6: AR_LAUNCH(TT, 6) break;                                                   \
    default: AR_LAUNCH(TT, 8) break;                                                  \
  }
  if (dtype == APHRO_F16) AR_WORLD(Half)
  else if (dtype == APHRO_BF16) AR_WORLD(BFloat)
  else AR_WORLD(Float)
#undef AR_WORLD
#undef AR_LAUNCH
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

// get_graph_buffer_ipc_meta: handles + offsets of the inputs recorded while capturing (count of them
// in *count; cap = room in the output arrays, in buffers).
extern "C" int aphro_custom_ar_get_graph_buffer_ipc_meta(void* fa_, char* handles_out, int64_t* offsets_out, int cap,
                                                         int* count) {
  CustomAr* fa = (CustomAr*)fa_;
  APHRO_CHECK(fa && count, "custom_ar_get_graph_buffer_ipc_meta: NULL argument");
  const int n = (int)fa->graph_unreg.size();
  *count = n;
  if (handles_out == nullptr) return APHRO_OK;          // size query
  APHRO_CHECK(cap >= n, "custom_ar: %d graph buffers, room for %d", n, cap);
  for (int i = 0; i < n; ++i) {
    int rc = aphro_ipc_get_mem_handle(fa->graph_unreg[i], handles_out + (size_t)i * sizeof(hipIpcMemHandle_t),
                                      offsets_out + i);
    if (rc != APHRO_OK) return rc;
  }
  return APHRO_OK;
}

// register_graph_buffers: handles[r][i], offsets[r][i] (rank-major) for the `count` recorded inputs.
extern "C" int aphro_custom_ar_register_graph_buffers(void* fa_, const char* handles, const int64_t* offsets, int count) {
  CustomAr* fa = (CustomAr*)fa_;
  APHRO_CHECK(fa, "custom_ar_register_graph_buffers: NULL handle");
  APHRO_CHECK(count == (int)fa->graph_unreg.size(), "custom_ar: %d graph buffers recorded, %d registered",
              (int)fa->graph_unreg.size(), count);
  const size_t hb = sizeof(hipIpcMemHandle_t);
  std::vector<char> hrow((size_t)fa->world * hb);
  std::vector<int64_t> orow(fa->world);
  for (int i = 0; i < count; ++i) {
    for (int r = 0; r < fa->world; ++r) {
      memcpy(hrow.data() + r * hb, handles + ((size_t)r * count + i) * hb, hb);
      orow[r] = offsets[(size_t)r * count + i];
    }
    ArPeers pe = {};
    int rc = fill_peers(fa, fa->graph_unreg[i], hrow.data(), orow.data(), &pe);
    if (rc != APHRO_OK) return rc;
    APHRO_CHECK(hipMemcpy(fa->d_slots + fa->slot_used, &pe, sizeof(pe), hipMemcpyHostToDevice) == hipSuccess,
                "custom_ar: hipMemcpy of the peer table failed");
    fa->registered[fa->graph_unreg[i]] = fa->slot_used++;
  }
  fa->graph_unreg.clear();
  return APHRO_OK;
}

// 1 if a barrier of this rank ever timed out (the results of that call are garbage); clears it.
extern "C" int aphro_custom_ar_error(void* fa_) {
  CustomAr* fa = (CustomAr*)fa_;
  if (!fa) return 0;
  ArSignal* s = (ArSignal*)fa->own_signal;
  uint32_t e = 0;
  if (hipMemcpy(&e, &s->error, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (e) {
    uint32_t z = 0;
    (void)hipMemcpy(&s->error, &z, sizeof(z), hipMemcpyHostToDevice);
  }
  return (int)e;
}

// Loopback communicator for one-GPU timing (bench.py --sim-tp): `world` ranks that are all this process -- every signal
// area, scratch region and input resolves to the local buffers, so each kernel of this file runs its real instruction
// stream (flag writes / polls through uncached memory, `world` input reads per element, scratch round trip) with local
// memory in place of the xGMI links.  The RESULT is the sum of `world` copies of the local input: timing only.
extern "C" int aphro_custom_ar_init_loopback(void** fa_out, void* signal, void* scratch, size_t scratch_bytes,
                                             void* rank_data, size_t rank_data_bytes, int world) {
  APHRO_CHECK(fa_out && signal && scratch && rank_data, "custom_ar_init_loopback: NULL argument");
  APHRO_CHECK(world >= 2 && world <= AR_MAX_RANKS && world % 2 == 0, "custom_ar: world size %d not in {2,4,6,8}", world);
  APHRO_CHECK(rank_data_bytes >= sizeof(ArPeers) && scratch_bytes % 16 == 0, "custom_ar: bad buffer sizes");
  CustomAr* fa = new CustomAr();
  fa->rank = 0; fa->world = world; fa->loopback = true;
  fa->own_signal = signal; fa->own_scratch = scratch; fa->scratch_bytes = scratch_bytes;
  fa->d_slots = (ArPeers*)rank_data;
  fa->slot_cap = (int)(rank_data_bytes / sizeof(ArPeers));
  for (int r = 0; r < world; ++r) {
    fa->sig[r] = (ArSignal*)signal;
    fa->scratch.ptr[r] = scratch;
  }
  *fa_out = fa;
  return APHRO_OK;
}

// 1 when aphro_custom_ar_fused_add_rms_norm will run the one-shot form for `tokens` x `hidden` elements of `esz` bytes,
// 0 for the two-shot (column-slice) form.
extern "C" int aphro_custom_ar_fused_norm_one_shot(int world, int64_t tokens, int hidden, int esz) {
  return aphro_custom_ar_should_one_shot(world, (size_t)tokens * hidden * esz);
}

// all_reduce(inp) -> fused_add_rms_norm(residual) [-> pack] in ONE launch: x = sum over the ranks of inp [tokens, hidden]
// (rounded to the dtype, as the all-reduce returns it), residual' = x + residual (in place; has_residual = 0: residual'
// = x, written if `residual` is given), y = rms_norm(residual') * weight -> `packed` (the W4A16 / FP8 decode GEMMs'
// fragment-major f16 A operand, aphro_wna16_packed_a_bytes) and / or row-major `out`.  Same bits as
// aphro_custom_ar_all_reduce followed by aphro_fused_add_rms_norm_pack (reference call sites: the row-parallel linear's
// all-reduce, modeling/layers/linear.py:1142-1143, then models/llama.py's fused_add_rms_norm).  Every rank ends with every
// row.  prefetch: see ArNormParams.  reg_buffer as in aphro_custom_ar_all_reduce.
static int ar_fused_norm_launch(void* fa_, const void* inp, void* residual, int has_residual,
                                const void* weight, float eps, void* packed, void* out,
                                void* q8_out, float* q8_scale_out, const float* q8_static,
                                const void* router_w, void* router_out, int num_experts,
                                int64_t tokens, int hidden, int dtype,
                                const void* prefetch, size_t prefetch_bytes,
                                void* reg_buffer, size_t reg_buffer_bytes, void* stream) {
  CustomAr* fa = (CustomAr*)fa_;
  hipStream_t st = (hipStream_t)stream;
  APHRO_CHECK(fa && inp && weight, "custom_ar_fused_add_rms_norm: NULL argument");
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "custom_ar_fused_add_rms_norm: dtype must be f16 or bf16");
  APHRO_CHECK(hidden % 8 == 0 && hidden > 0 && hidden <= 16384, "custom_ar_fused_add_rms_norm: hidden=%d unsupported", hidden);
  APHRO_CHECK(packed == nullptr || hidden % 128 == 0, "custom_ar_fused_add_rms_norm: packing needs hidden %% 128 == 0");
  APHRO_CHECK(!has_residual || residual != nullptr, "custom_ar_fused_add_rms_norm: residual missing");
  APHRO_CHECK(packed != nullptr || out != nullptr || q8_out != nullptr, "custom_ar_fused_add_rms_norm: no output requested");
  APHRO_CHECK(q8_out == nullptr || q8_scale_out != nullptr, "custom_ar_fused_add_rms_norm: e4m3 output without a scale output");
  APHRO_CHECK(((uintptr_t)q8_out % 8) == 0, "custom_ar_fused_add_rms_norm: the e4m3 output must be 8-byte aligned");
  APHRO_CHECK(tokens >= 0 && tokens <= (int64_t)AR_MAX_BLOCKS, "custom_ar_fused_add_rms_norm: %lld tokens (at most %d: decode batches)",
              (long long)tokens, AR_MAX_BLOCKS);
  APHRO_CHECK((((uintptr_t)inp | (uintptr_t)residual | (uintptr_t)weight | (uintptr_t)packed | (uintptr_t)out) % 16) == 0,
              "custom_ar_fused_add_rms_norm: input, residual, weight and outputs must be 16-byte aligned");
  if (tokens == 0) return APHRO_OK;
  const size_t bytes = (size_t)tokens * hidden * 2;
  const void* src = inp;
  if (reg_buffer != nullptr) {
    APHRO_CHECK(bytes <= reg_buffer_bytes, "custom_ar: registered buffer is too small (%zu > %zu)", bytes, reg_buffer_bytes);
    APHRO_CHECK(hipMemcpyAsync(reg_buffer, inp, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess,
                "custom_ar: staging copy failed");
    src = reg_buffer;
  }
  ArNormParams q;
  memset(&q, 0, sizeof(q));
  int rc = ar_fill_params(fa, src, st, &q.ar);
  if (rc != APHRO_OK) return rc;
  const bool one_shot = aphro_custom_ar_should_one_shot(fa->world, bytes) != 0;
  APHRO_CHECK(one_shot || bytes <= fa->scratch_bytes,
              "custom_ar_fused_add_rms_norm: %zu bytes exceed the two-shot scratch", bytes);
  q.residual = (uint16_t*)residual; q.weight = (const uint16_t*)weight;
  q.packed = (uint16_t*)packed; q.out = (uint16_t*)out;
  q.eps = eps; q.has_residual = has_residual ? 1 : 0;
  q.tokens = (int)tokens; q.hidden = hidden;
  q.rows_per_rank = (hidden / 8 + fa->world - 1) / fa->world;
  q.replicate_residual = 0;
  q.q8_out = (uint8_t*)q8_out; q.q8_scale_out = q8_scale_out; q.q8_static = q8_static;
  q.router_w = (const uint16_t*)router_w; q.router_out = (uint16_t*)router_out; q.num_experts = num_experts;
  // the block size of aphro_fused_add_rms_norm_pack / aphro_fused_add_rms_norm_quant_fp8 (same thread -> element mapping, same reduction order)
  int nv = hidden / 8, t = nv <= 1024 ? nv : (nv + 1) / 2;
  t = (t + 63) / 64 * 64;
  t = t < 64 ? 64 : (t > 1024 ? 1024 : t);
  int blocks = (int)tokens;
  q.nb = blocks;
  if (prefetch != nullptr && prefetch_bytes >= 16) {
    // (a hint: an unaligned pointer is rounded up; a modest number of extra workgroups streams at the HBM rate)
    const uintptr_t a = ((uintptr_t)prefetch + 15) & ~(uintptr_t)15;
    const size_t skip = (size_t)(a - (uintptr_t)prefetch);
    if (prefetch_bytes > skip + 16) {
      q.pf = (const u32x4*)a;
      q.pf_n16 = (prefetch_bytes - skip) / 16;
      const int max_pf = APHRO_LAB_ENV_INT("APHRO_AR_PREFETCH_BLOCKS", 192);
      size_t want = (q.pf_n16 + (size_t)t * 4 - 1) / ((size_t)t * 4);
      blocks += (int)(want > (size_t)max_pf ? (size_t)max_pf : want);
    }
  }
#define ARN_LAUNCH(TT, W, Q)                                                                              \
  {                                                                                                       \
    if (one_shot) hipLaunchKernelGGL((ar_norm_one_shot_kernel<TT, W, Q>), dim3(blocks), dim3(t), 0, st, q); \
    else hipLaunchKernelGGL((ar_norm_two_shot_kernel<TT, W, Q>), dim3(blocks), dim3(t), 0, st, q);         \
  }
#define ARN_WORLD(TT, Q)                                                              \
  switch (fa->world) {                                                               \
    case 2: ARN_LAUNCH(TT, 2, Q) break;                                               \
    case 4: ARN_LAUNCH(TT, 4, Q) break;                                               \
    case 6: ARN_LAUNCH(TT, 6, Q) break;                                               \
    default: ARN_LAUNCH(TT, 8, Q) break;                                              \
  }
  if (q8_out) {
    if (dtype == APHRO_F16) ARN_WORLD(Half, AR_EPI_Q8)
    else ARN_WORLD(BFloat, AR_EPI_Q8)
  } else if (router_out) {
    if (dtype == APHRO_F16) ARN_WORLD(Half, AR_EPI_ROUTER)
    else ARN_WORLD(BFloat, AR_EPI_ROUTER)
  } else {
    if (dtype == APHRO_F16) ARN_WORLD(Half, AR_EPI_NONE)
    else ARN_WORLD(BFloat, AR_EPI_NONE)
  }
#undef ARN_WORLD
#undef ARN_LAUNCH
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" int aphro_custom_ar_fused_add_rms_norm(void* fa_, const void* inp, void* residual, int has_residual,
                                                  const void* weight, float eps, void* packed, void* out,
                                                  int64_t tokens, int hidden, int dtype,
                                                  const void* prefetch, size_t prefetch_bytes,
                                                  void* reg_buffer, size_t reg_buffer_bytes, void* stream) {
  return ar_fused_norm_launch(fa_, inp, residual, has_residual, weight, eps, packed, out, nullptr, nullptr, nullptr, nullptr,
                              nullptr, 0, tokens, hidden, dtype, prefetch, prefetch_bytes, reg_buffer, reg_buffer_bytes, stream);
}

// The FP8 W8A8 form: all_reduce(inp) -> fused_add_rms_norm(residual) -> per-token (or static) FP8
// quantisation of the normalised row in ONE launch -- the bits of aphro_custom_ar_all_reduce followed by
// aphro_fused_add_rms_norm_quant_fp8 on its `input` path (reference call sites: linear.py:1142-1143, models/llama.py's
// layernorm, then the scaled_fp8_quant at the head of Fp8LinearMethod.apply, quantization/fp8.py).  q_out [tokens, hidden]
// e4m3, scale_out [tokens] (static_scale [1]: q = fp8(y * (1 / s)) and every scale_out entry = s); `out`: optional
// row-major T copy of the normalised rows (the last norm of the model).
extern "C" int aphro_custom_ar_fused_add_rms_norm_quant_fp8(void* fa_, const void* inp, void* residual, int has_residual,
                                                            const void* weight, float eps, void* q_out, float* scale_out,
                                                            const float* static_scale, void* out, int64_t tokens,
                                                            int hidden, int dtype, void* reg_buffer,
                                                            size_t reg_buffer_bytes, void* stream) {
  APHRO_CHECK(q_out && scale_out, "custom_ar_fused_add_rms_norm_quant_fp8: NULL output");
  return ar_fused_norm_launch(fa_, inp, residual, has_residual, weight, eps, nullptr, out, q_out, scale_out, static_scale,
                              nullptr, nullptr, 0, tokens, hidden, dtype, nullptr, 0, reg_buffer, reg_buffer_bytes, stream);
}

// The sparse-MLP form: all_reduce(inp) -> fused_add_rms_norm(residual) -> the router's logits in ONE launch -- the bits of
// aphro_custom_ar_all_reduce followed by aphro_fused_add_rms_norm_router on its `input` path (reference call sites: the
// attention block's row-parallel all-reduce, linear.py:1142-1143; models/mixtral.py's post_attention_layernorm; the
// replicated gate linear of MixtralMoE, mixtral.py:60-110).  out [tokens, hidden] T (the experts' input), router_out
// [tokens, num_experts] T, num_experts <= 16.
extern "C" int aphro_custom_ar_fused_add_rms_norm_router(void* fa_, const void* inp, void* residual, int has_residual,
                                                         const void* weight, float eps, void* out, const void* router_w,
                                                         void* router_out, int num_experts, int64_t tokens, int hidden,
                                                         int dtype, void* reg_buffer, size_t reg_buffer_bytes,
                                                         void* stream) {
  APHRO_CHECK(out && router_w && router_out, "custom_ar_fused_add_rms_norm_router: NULL argument");
  APHRO_CHECK(num_experts >= 1 && num_experts <= 16, "custom_ar_fused_add_rms_norm_router: 1..16 experts (got %d)", num_experts);
  APHRO_CHECK((((uintptr_t)router_w) % 16) == 0, "custom_ar_fused_add_rms_norm_router: router weights must be 16-byte aligned");
  return ar_fused_norm_launch(fa_, inp, residual, has_residual, weight, eps, nullptr, out, nullptr, nullptr, nullptr, router_w,
                              router_out, num_experts, tokens, hidden, dtype, nullptr, 0, reg_buffer, reg_buffer_bytes, stream);
}
