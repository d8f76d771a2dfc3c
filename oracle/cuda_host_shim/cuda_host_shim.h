// TEST INFRASTRUCTURE ONLY (oracle/): a minimal "CUDA on the host" so that the reference's own GPTQ kernels
// (kernels/quantization/gptq/q_gemm.cu, qdq_4.cuh, matrix_view.cuh -- compiled from where they lie, never copied)
// can be EXECUTED on the CPU as the pin of oracle/quant.py.  Nothing here restates reference arithmetic: it only
// supplies what nvcc supplies -- the half / half2 types with IEEE round-to-nearest-even arithmetic, the thread /
// block index variables, __syncthreads (cooperative fibers, one per emulated thread) and atomicAdd.
#pragma once
#include <ucontext.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };
extern uint3_ threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
struct int4 { int x, y, z, w; };
using std::max;
using std::min;

// ---- IEEE binary16 with round-to-nearest-even conversions (software: no F16C / _Float16 dependency) -------------
namespace cuemu {
inline float h2f(uint16_t h) {
  const uint32_t s = (uint32_t)(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ff, u;
  if (e == 0) {
    if (m == 0) u = s;
    else { int sh = 0; while (!(m & 0x400)) { m <<= 1; ++sh; } m &= 0x3ff; u = s | ((uint32_t)(113 - sh) << 23) | (m << 13); }
  } else if (e == 31) u = s | 0x7f800000u | (m << 13);
  else u = s | ((e + 112) << 23) | (m << 13);
  float f; std::memcpy(&f, &u, 4); return f;
}
// double -> binary16, one rounding (RNE), overflow to inf like __double2half / cvt.rn.f16
inline uint16_t d2h(double d) {
  uint64_t u; std::memcpy(&u, &d, 8);
  const uint16_t s = (uint16_t)((u >> 48) & 0x8000u);
  const int e = (int)((u >> 52) & 0x7ff);
  const uint64_t m = u & 0xfffffffffffffull;
  if (e == 0x7ff) return (uint16_t)(s | 0x7c00u | (m ? 0x200u : 0u));
  if (e == 0) return s;                                   // double subnormal: far below half's range
  const int he = e - 1023 + 15;
  if (he >= 31) return (uint16_t)(s | 0x7c00u);
  uint64_t sig = m | (1ull << 52);                        // 53-bit significand
  int shift;                                              // bits to drop so that 11 (normal) or fewer remain
  if (he >= 1) shift = 42; else { shift = 42 + (1 - he); if (shift > 63) return s; }
  const uint64_t keep = sig >> shift, rem = sig & ((1ull << shift) - 1), half = 1ull << (shift - 1);
  uint64_t r = keep;
  if (rem > half || (rem == half && (keep & 1))) ++r;
  uint32_t out;
  if (he >= 1) { out = ((uint32_t)(he - 1) << 10) + (uint32_t)r; if (out >= 0x7c00u) out = 0x7c00u; }   // r carries into the exponent
  else out = (uint32_t)r;                                 // subnormal (may round up into the smallest normal)
  return (uint16_t)(s | out);
}
}  // namespace cuemu

struct __half_raw { uint16_t x; };
struct half {
  uint16_t x;
  half() = default;
  half(const __half_raw& r) : x(r.x) {}
  operator __half_raw() const { return __half_raw{x}; }
};
struct half2 { half x, y; };
static inline double __h2d(half h) { return (double)cuemu::h2f(h.x); }
static inline half __d2h(double d) { half h; h.x = cuemu::d2h(d); return h; }

static inline float __half2float(half h) { return cuemu::h2f(h.x); }
static inline half __float2half_rn(float f) { return __d2h((double)f); }
static inline half __float2half(float f) { return __d2h((double)f); }
static inline half __int2half_rn(int i) { return __d2h((double)i); }
static inline half __ushort_as_half(unsigned short u) { half h; h.x = u; return h; }
static inline unsigned short __half_as_ushort(half h) { return h.x; }
static inline half __hadd(half a, half b) { return __d2h(__h2d(a) + __h2d(b)); }       // exact in double, one rounding
static inline half __hsub(half a, half b) { return __d2h(__h2d(a) - __h2d(b)); }
static inline half __hmul(half a, half b) { return __d2h(__h2d(a) * __h2d(b)); }
static inline half __hfma(half a, half b, half c) { return __d2h(std::fma(__h2d(a), __h2d(b), __h2d(c))); }
static inline half2 __halves2half2(half a, half b) { half2 r; r.x = a; r.y = b; return r; }
static inline half2 __half2half2(half a) { half2 r; r.x = a; r.y = a; return r; }
static inline half __low2half(half2 a) { return a.x; }
static inline half __high2half(half2 a) { return a.y; }
static inline float __low2float(half2 a) { return __half2float(a.x); }
static inline float __high2float(half2 a) { return __half2float(a.y); }
static inline half2 __hadd2(half2 a, half2 b) { return __halves2half2(__hadd(a.x, b.x), __hadd(a.y, b.y)); }
static inline half2 __hmul2(half2 a, half2 b) { return __halves2half2(__hmul(a.x, b.x), __hmul(a.y, b.y)); }
static inline half2 __hfma2(half2 a, half2 b, half2 c) { return __halves2half2(__hfma(a.x, b.x, c.x), __hfma(a.y, b.y, c.y)); }
static inline unsigned __funnelshift_rc(unsigned lo, unsigned hi, unsigned shift) {
  shift = shift > 32 ? 32 : shift;
  return shift == 32 ? hi : (unsigned)((((uint64_t)hi << 32) | lo) >> shift);
}
// the emulator runs threads one at a time, so an atomic add is an add (in emulated-thread order)
static inline void atomicAdd(half* p, half v) { *p = __hadd(*p, v); }
static inline void atomicAdd(half2* p, half2 v) { *p = __hadd2(*p, v); }

// ---- kernel launch emulation: one fiber per thread, round-robin between barriers --------------------------------
namespace cuemu {
void syncthreads();
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
}
static inline void __syncthreads() { cuemu::syncthreads(); }
