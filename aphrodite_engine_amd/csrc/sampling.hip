// Fused random sampling for the decode step (sampling inside the HIP graph):
//   temperature -> top-k -> top-p -> min-p -> softmax -> multinomial (argmax(probs / q), q ~ Exp(1))
// one launch, one workgroup per row, no sort.  Semantics of aphrodite/modeling/layers/sampler.py:
// logits.div_(t) (:256-262), _apply_top_k_top_p (:865-891), _apply_min_p (:894-908), _multinomial
// (:1273-1292), in fp32.
//
// The reference sorts every row (128k logits) and scatters back; here the two thresholds are found by
// a 3-pass radix select (11 + 11 + 10 bits) over order-preserving integer keys with histograms in LDS:
//   top-k : element counts, walked from the largest key down until k elements are covered;
//   top-p : FIXED-POINT probability mass (exp(x - max) * 2^32 as u64), walked from the smallest key up
//           until the mass exceeds (1 - p) * total.  Integer adds are associative, so the threshold --
//           hence the sample -- does not depend on the order in which LDS atomics land.
// Elements equal to a threshold are kept as a group (the reference cuts inside a tie group by sort
// position; the kept set here can only be larger, by tied values).  The row (256 KiB of f16 logits) is
// re-read from L2 once per pass, at most 8 passes; nothing but the sampled index is written.
#include "common.h"

namespace aphro {

constexpr int SP_THREADS = 1024;
constexpr int SP_BINS = 2048;

struct SampleParams {
  int64_t* out;
  const void* logits;
  int64_t row_stride;        // elements
  const float* temperature;  // [rows] or NULL (1.0)
  const int32_t* top_k;      // [rows] or NULL (disabled); <= 0 or >= vocab: disabled
  const float* top_p;        // [rows] or NULL (disabled)
  const float* min_p;        // [rows] or NULL (disabled): drop tokens with prob < min_p * max prob (:894-908)
  const float* q;            // [rows, vocab] Exp(1) noise, or NULL: drawn in the kernel from `seeds`
  int64_t q_stride;
  const int64_t* seeds;      // [rows], used when q == NULL
  float* logprobs;           // [rows] or NULL: log-softmax of the masked, temperature-scaled row at the sampled token
  int vocab;
};

__device__ __forceinline__ uint32_t key_of(float x) {       // ascending, order preserving
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  return (u >> 31) ? ~u : (u | 0x80000000u);
}

template <typename T>
__device__ __forceinline__ float load_logit(const void* base, int64_t i) {
  if constexpr (__is_same(T, Float)) return ((const float*)base)[i];
  else return T::to_f32(((const uint16_t*)base)[i]);
}

// Visit every logit of the row as (index, logit / t).  VEC: 16-byte loads, four in flight per thread
// before any is consumed (the scalar loop is latency bound: one dependent L2 round trip per element,
// measured 630 us per launch at 32 x 128256); indices ascend within a thread.
template <typename T, bool VEC, typename F>
__device__ __forceinline__ void for_each_logit(const void* lrow, int V, float t, F&& f) {
  const int tid = threadIdx.x;
  const bool unit = t == 1.0f;                    // x / 1 == x: skip the IEEE division
  if constexpr (VEC) {
    constexpr int E = __is_same(T, Float) ? 4 : 8;   // elements per 16-byte vector
    constexpr int UN = 4;
    const int nvec = V / E;
    for (int c0 = tid; c0 < nvec; c0 += SP_THREADS * UN) {
      u32x4 v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int c = c0 + u * SP_THREADS;
        v[u] = c < nvec ? ((const u32x4*)lrow)[c] : u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int c = c0 + u * SP_THREADS;
        if (c >= nvec) break;
        if constexpr (__is_same(T, Float)) {
          const f32x4 fv = __builtin_bit_cast(f32x4, v[u]);   // whole-vector cast (element casts are miscompiled)
#pragma unroll
          for (int j = 0; j < 4; ++j) f(c * 4 + j, unit ? fv[j] : fv[j] / t);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float x = T::to_f32((uint16_t)(v[u][j >> 1] >> (16 * (j & 1))));
            f(c * 8 + j, unit ? x : x / t);
          }
        }
      }
    }
    for (int i = nvec * E + tid; i < V; i += SP_THREADS) {
      const float x = load_logit<T>(lrow, i);
      f(i, unit ? x : x / t);
    }
  } else {
    for (int i = tid; i < V; i += SP_THREADS) {
      const float x = load_logit<T>(lrow, i);
      f(i, unit ? x : x / t);
    }
  }
}

// Exp(1) noise from a counter hash when the caller supplies none (splitmix64 finaliser, 24-bit uniform)
__device__ __forceinline__ float exp_noise(int64_t seed, int i) {
  uint64_t z = (uint64_t)seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = ((float)(uint32_t)(z >> 40) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
  return -__logf(u) + 1e-10f;
}

// wave 0: which digit does the running sum (from the top, or from the bottom) cross `target` in?
// returns the digit and the sum accumulated BEFORE it.  W = u32 counts or u64 masses.
// `hist` holds SP_SUB interleaved copies of every bin (lanes spread over them to cut same-address
// serialisation of the LDS atomics: random-model logits share a handful of exponents); a bin's value
// is the sum of its copies.
constexpr int SP_SUB = 4;
template <typename W>
struct SubHist {
  const W* h;
  __device__ __forceinline__ W operator[](int b) const {
    W s = 0;
#pragma unroll
    for (int j = 0; j < SP_SUB; ++j) s += h[b * SP_SUB + j];
    return s;
  }
};

template <typename W, bool FROM_TOP>
__device__ __forceinline__ void pick_digit(const SubHist<W> hist, int nbins, W before, W target, int* digit_out, W* before_out) {
  const int lane = threadIdx.x;                 // called by wave 0 only
  const int per = nbins / 64;                   // bins per lane, contiguous; lane 0 owns the FIRST bins walked
  W mine = 0;
  for (int j = 0; j < per; ++j) {
    const int b = FROM_TOP ? nbins - 1 - (lane * per + j) : lane * per + j;
    mine += hist[b];
  }
  W incl = mine;                                // inclusive scan over lanes
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    W o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  const W excl = before + incl - mine;
  // FROM_TOP: first position where the sum reaches target (>=); else where it exceeds target (>)
  const bool hit = FROM_TOP ? (excl + mine >= target) : (excl + mine > target);
  const uint64_t ballot = __ballot(hit);
  int d_found = FROM_TOP ? 0 : nbins - 1;       // nothing crosses (p <= 0 ...): fall to the extreme digit
  W b_found = before;
  bool found = false;
  if (ballot != 0) {
    const int first = __builtin_ctzll(ballot);
    if (lane == first) {
      W acc = excl;
      for (int j = 0; j < per; ++j) {
        const int b = FROM_TOP ? nbins - 1 - (lane * per + j) : lane * per + j;
        const W h = hist[b];
        if (FROM_TOP ? (acc + h >= target) : (acc + h > target)) {
          d_found = b; b_found = acc; found = true;
          break;
        }
        acc += h;
      }
    }
    d_found = __shfl(d_found, first);
    b_found = __shfl(b_found, first);
    (void)found;
  } else if (!FROM_TOP) {
    // mass never exceeds the target: keep only the largest keys -> walk to the top digit that is populated
    int top = -1;
    for (int j = 0; j < per; ++j) {
      const int b = lane * per + j;
      if (hist[b] != 0) top = b;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const int o = __shfl_xor(top, d);
      top = o > top ? o : top;
    }
    d_found = top < 0 ? 0 : top;
    W below = 0;                                 // mass strictly below that digit
    for (int j = 0; j < per; ++j) {
      const int b = lane * per + j;
      if (b < d_found) below += hist[b];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) below += __shfl_xor(below, d);
    b_found = before + below;
  }
  if (lane == 0) {
    *digit_out = d_found;
    *before_out = b_found;
  }
}

template <typename T, bool VEC>
__global__ __launch_bounds__(SP_THREADS) void sample_kernel(SampleParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long hist_raw[];   // SP_BINS * SP_SUB u64 (64 KiB)
  unsigned long long* mass = hist_raw;
  uint32_t* cnt = (uint32_t*)hist_raw;          // the count histogram reuses the same storage (top-k runs first)
  const int sub = threadIdx.x & (SP_SUB - 1);
  __shared__ float redf[SP_THREADS / 64];
  __shared__ int redi[SP_THREADS / 64];
  __shared__ int sh_digit;
  __shared__ uint32_t sh_cbefore;
  __shared__ unsigned long long sh_mbefore, sh_target;
  const int row = blockIdx.x, tid = threadIdx.x, V = p.vocab;
  const void* lrow = (const char*)p.logits + (size_t)row * p.row_stride * (__is_same(T, Float) ? 4 : 2);
  float t = p.temperature ? p.temperature[row] : 1.0f;
  if (t < 1e-5f) t = 1.0f;

  // ---- pass 1: row maximum (it always survives both filters: the softmax shift) ------------------
  float mx = -INFINITY;
  for_each_logit<T, VEC>(lrow, V, t, [&](int, float x) { mx = __builtin_fmaxf(mx, x); });
  mx = wave_max(mx);
  if ((tid & 63) == 0) redf[tid >> 6] = mx;
  __syncthreads();
  mx = redf[0];
  for (int w = 1; w < SP_THREADS / 64; ++w) mx = __builtin_fmaxf(mx, redf[w]);
  __syncthreads();

  // ---- top-k: key of the k-th largest element -----------------------------------------------------
  int k = p.top_k ? p.top_k[row] : 0;
  uint32_t kmin = 0;                              // survivors: key >= kmin
  if (k > 0 && k < V) {
    uint32_t prefix = 0;
    uint32_t before = 0;
    int shift = 32;
    for (int pass = 0; pass < 3; ++pass) {
      const int bits = pass == 2 ? 10 : 11;
      shift -= bits;
      for (int b = tid; b < SP_BINS * SP_SUB; b += SP_THREADS) cnt[b] = 0;
      __syncthreads();
      const uint32_t hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + bits));
      for_each_logit<T, VEC>(lrow, V, t, [&](int, float x) {
        const uint32_t key = key_of(x);
        if ((key & hi_mask) == prefix) atomicAdd(&cnt[((key >> shift) & ((1u << bits) - 1)) * SP_SUB + sub], 1u);
      });
      __syncthreads();
      if (tid < 64)
        pick_digit<uint32_t, true>(SubHist<uint32_t>{cnt}, 1 << bits, before, (uint32_t)k, &sh_digit, &sh_cbefore);
      __syncthreads();
      prefix |= (uint32_t)sh_digit << shift;
      before = sh_cbefore;
      __syncthreads();
    }
    kmin = prefix;
  }

  // ---- top-p: smallest key whose inclusive ascending mass exceeds (1 - p) * total ------------------
  const float pp = p.top_p ? p.top_p[row] : 1.0f;
  if (pp < 1.0f) {
    uint32_t prefix = 0;
    unsigned long long before = 0;
    int shift = 32;
    for (int pass = 0; pass < 3; ++pass) {
      const int bits = pass == 2 ? 10 : 11;
      shift -= bits;
      for (int b = tid; b < SP_BINS * SP_SUB; b += SP_THREADS) mass[b] = 0;
      __syncthreads();
      const uint32_t hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + bits));
      for_each_logit<T, VEC>(lrow, V, t, [&](int, float x) {
        const uint32_t key = key_of(x);
        if (key >= kmin && (key & hi_mask) == prefix) {
          const unsigned long long f = (unsigned long long)((double)expf(x - mx) * 4294967296.0);
          atomicAdd(&mass[((key >> shift) & ((1u << bits) - 1)) * SP_SUB + sub], f);
        }
      });
      __syncthreads();
      if (pass == 0) {                              // total mass of the top-k survivors -> the target
        if (tid < 64) {
          unsigned long long s = 0;
          for (int b = tid; b < SP_BINS * SP_SUB; b += 64) s += mass[b];
#pragma unroll
          for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
          if (tid == 0) sh_target = (unsigned long long)((1.0 - (double)pp) * (double)s);
        }
        __syncthreads();
      }
      if (tid < 64)
        pick_digit<unsigned long long, false>(SubHist<unsigned long long>{mass}, 1 << bits, before, sh_target, &sh_digit,
                                              &sh_mbefore);
      __syncthreads();
      prefix |= (uint32_t)sh_digit << shift;
      before = sh_mbefore;
      __syncthreads();
    }
    kmin = prefix > kmin ? prefix : kmin;
  }

  // ---- multinomial: argmax over the survivors of exp(x - max) / q, lowest index on ties ------------
  float best = -1.0f;
  int best_i = 0x7fffffff;
  const float* qrow = p.q ? p.q + (size_t)row * p.q_stride : nullptr;
  const int64_t seed = p.seeds ? p.seeds[row] : 0;
  // min-p acts on the probabilities of the row as masked so far: p_i < min_p * p_max  <=>  exp(x_i - max) < min_p
  const float mp = p.min_p ? p.min_p[row] : 0.0f;
  // logprobs: the reference takes log_softmax of the row AFTER the masks (sampler.py:545): normaliser = the survivors
  const bool want_lp = p.logprobs != nullptr;
  float zsum = 0.f;
  for_each_logit<T, VEC>(lrow, V, t, [&](int i, float x) {
    if (key_of(x) >= kmin) {
      const float e = expf(x - mx);
      if (e < mp && x != mx) return;
      zsum += e;
      const float qq = qrow ? qrow[i] : exp_noise(seed, i);
      const float s = e / qq;
      if (s > best) { best = s; best_i = i; }        // ascending i: first maximum wins inside the thread
    }
  });
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const float ob = __shfl_xor(best, d);
    const int oi = __shfl_xor(best_i, d);
    if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
  }
  if ((tid & 63) == 0) { redf[tid >> 6] = best; redi[tid >> 6] = best_i; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < SP_THREADS / 64; ++w)
      if (redf[w] > best || (redf[w] == best && redi[w] < best_i)) { best = redf[w]; best_i = redi[w]; }
    best_i = best_i == 0x7fffffff ? 0 : best_i;
    p.out[row] = best_i;
    redi[0] = best_i;
  }
  if (want_lp) {                                  // fixed reduction order: thread, wave tree, waves in sequence
    __syncthreads();
    zsum = wave_sum(zsum);
    if ((tid & 63) == 0) redf[tid >> 6] = zsum;
    __syncthreads();
    if (tid == 0) {
      float z = 0.f;
      for (int w = 0; w < SP_THREADS / 64; ++w) z += redf[w];
      const float xs = load_logit<T>(lrow, redi[0]);
      p.logprobs[row] = ((t == 1.0f ? xs : xs / t) - mx) - logf(z);
    }
  }
}

}  // namespace aphro

using namespace aphro;

extern "C" int aphro_sample_top_k_top_p(int64_t* out, const void* logits, int64_t row_stride, const float* temperature,
                                        const int32_t* top_k, const float* top_p, const float* min_p, const float* q,
                                        int64_t q_stride, const int64_t* seeds, float* logprobs_out, int64_t rows,
                                        int64_t vocab, int dtype, void* stream) {
  APHRO_CHECK(out && logits, "sample_top_k_top_p: NULL argument");
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16 || dtype == APHRO_F32, "sample_top_k_top_p: unsupported dtype %d", dtype);
  APHRO_CHECK(vocab > 0 && vocab < (1ll << 31), "sample_top_k_top_p: vocab=%lld unsupported", (long long)vocab);
  APHRO_CHECK(q != nullptr || seeds != nullptr, "sample_top_k_top_p: either the Exp(1) noise q or per-row seeds are required");
  if (rows == 0) return APHRO_OK;
  SampleParams p;
  p.out = out; p.logits = logits; p.row_stride = row_stride; p.temperature = temperature; p.top_k = top_k;
  p.top_p = top_p; p.min_p = min_p; p.q = q; p.q_stride = q_stride; p.seeds = seeds; p.logprobs = logprobs_out; p.vocab = (int)vocab;
  dim3 grid((unsigned)rows), block(SP_THREADS);
  const size_t esz = dtype == APHRO_F32 ? 4 : 2;
  const bool vec = ((uintptr_t)logits % 16) == 0 && ((size_t)row_stride * esz) % 16 == 0;   // every row 16-byte aligned
  const size_t lds = (size_t)SP_BINS * SP_SUB * sizeof(unsigned long long);
#define SP_LAUNCH1(TT, VV)                                                                                  \
  {                                                                                                         \
    auto kern = sample_kernel<TT, VV>;                                                                      \
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
    hipLaunchKernelGGL(kern, grid, block, lds, (hipStream_t)stream, p);                                      \
  }
#define SP_LAUNCH(TT) { if (vec) SP_LAUNCH1(TT, true) else SP_LAUNCH1(TT, false) }
  if (dtype == APHRO_F16) SP_LAUNCH(Half)
  else if (dtype == APHRO_BF16) SP_LAUNCH(BFloat)
  else SP_LAUNCH(Float)
#undef SP_LAUNCH
#undef SP_LAUNCH1
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
