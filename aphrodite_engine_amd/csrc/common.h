// Common device helpers for the gfx950 (CDNA4) kernels.  wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/aphrodite_mi355x.h"

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));

#define WAVE 64

namespace aphro {

void set_error(const char* fmt, ...);

#define APHRO_CHECK(cond, ...)                      \
  do {                                              \
    if (!(cond)) {                                  \
      ::aphro::set_error(__VA_ARGS__);              \
      return APHRO_ERR_INVALID;                     \
    }                                               \
  } while (0)

#define APHRO_LAUNCH_CHECK()                                          \
  do {                                                                \
    hipError_t e__ = hipGetLastError();                               \
    if (e__ != hipSuccess) {                                          \
      ::aphro::set_error("launch failed: %s", hipGetErrorString(e__)); \
      return APHRO_ERR_LAUNCH;                                        \
    }                                                                 \
  } while (0)

// Host-side per-device state (function attributes, CU counts) is indexed by the current HIP device: a process that
// drives a second GPU must not inherit the first one's "already set" flag.
constexpr int APHRO_MAX_DEVICES = 64;
inline int device_slot() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= APHRO_MAX_DEVICES) d = 0;
  return d;
}
inline int device_cu_count() {
  static int n[APHRO_MAX_DEVICES] = {};
  const int d = device_slot();
  if (n[d] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) v = 256;
    n[d] = v;
  }
  return n[d];
}
// ---- the environment, read ONCE --------------------------------------------------------------------------------------
// Every switch the library honours is a field of Knobs, filled from the environment the first time knobs() is called and
// again by aphro_reload_env() (tests flip a switch inside one process: ops.knob(...)); nothing on a launch path calls
// getenv.  These are the switches INTEGRATION.md documents.  Plan-forcing / ablation overrides of the kernel labs are NOT
// product switches: APHRO_LAB_ENV_INT(name, default) reads them only in a -DAPHRO_LAB build (make LAB=1 -> the library the
// tools/ scripts load through APHRODITE_MI355X_LIB) and is the constant `default` in the product.
struct Knobs {
  int pa_splits;              // APHRO_PA_SPLITS=<n>: force the in-launch KV split count of decode attention (0: planned)
  int fa_v4_min_keys;         // APHRO_FA_V4_MIN_KEYS=<n>: prefill attention, fourth generation from n keys (default 4096)
  int fa_no_xcd;              // APHRO_FA_NO_XCD=1: prefill attention without the kv-head -> XCD placement (same bits)
  int fp8_stream_all;         // APHRO_FP8_STREAM_ALL=1: the LDS-DMA FP8 decode kernel on every shape it tiles
  int wna16_stream;           // APHRO_WNA16_STREAM=0: the two-pass resident kernel instead of the single-pass stream kernel
  int wna16_op_no_resident;   // APHRO_WNA16_OP_NO_RESIDENT=1: op-level gptq_gemm as pack + GEMM + reduce (three launches)
  int wna16_large_8phase;     // APHRO_WNA16_LARGE_8PHASE=0/1: prefill W4A16 schedule (-1: by K)
  int wna16_large_two_pass;   // APHRO_WNA16_LARGE_TWO_PASS=0/1: prefill W4A16 dequantise-once form off / wherever it applies (-1: from 4096 rows)
  int wna16_mid_waves;        // APHRO_WNA16_MID_WAVES=4/8: K waves of the 33..64-row kernel (0: planned)
  int res_cfg[4];             // APHRO_WNA16_RES_CFG="nwv,nseg,np4,rem": force a resident-kernel plan (tests)
  int res_cfg_set;
  long ar_one_shot_max;       // APHRO_CUSTOM_AR_ONE_SHOT_MAX=<bytes>: one- / two-shot crossover of the peer-access all-reduce (-1: planned)
  long ar_timeout_ms;         // APHRODITE_CUSTOM_AR_TIMEOUT_MS=<ms>: bound on a peer wait (0: the built-in default)
  int cu_masked;              // HSA_CU_MASK / ROC_GLOBAL_CU_MASK present: not every reported CU is usable
};
const Knobs& knobs();
#ifdef APHRO_LAB
inline int aphro_lab_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#define APHRO_LAB_ENV_INT(name, dflt) ([]() -> int { static const int v_ = aphro::aphro_lab_env_int(name, dflt); return v_; }())
#else
#define APHRO_LAB_ENV_INT(name, dflt) (dflt)
#endif

// Persistent grids whose workgroups WAIT on one another (the stream-K owners of the prompt-sized GEMMs spin on flags
// of higher-index workgroups) are only correct when the whole grid is co-resident.  One workgroup per reported CU is,
// unless the process runs under a CU mask (the runtime still reports every CU): then those plans are not used at all
// and the shape takes the one-workgroup-per-tile + split-K plan, which has no cross-workgroup wait.
inline int device_coresident_cu_count() { return knobs().cu_masked ? 0 : device_cu_count(); }

// Pair-major 16-bit activations between a fused producer and the AQ GEMM: element (row m, column k) of [M, K] sits at
//   ((((k / 64) * mtiles + m / 16) * 2 + (k % 16) / 8) * 64 + ((k % 64) / 16) * 16 + m % 16) * 8 + k % 8      (16-bit elements)
// i.e. for every (k-pair, 16-row tile, half) the 64 lanes' 16-byte pieces are one contiguous KiB in lane order: the consumer's
// A loads are lane-linear like its weight loads (a row-major gather touches 16 cache lines per instruction, twice per
// fragment for 16-bit data).  mtiles = ceil(M / 16); rows >= M of the last tile are never written and never read back.
__host__ __device__ __forceinline__ size_t aq_pair_offset(int m, int k, int mtiles) {
  return ((((size_t)(k >> 6) * mtiles + (m >> 4)) * 2 + ((k & 15) >> 3)) * 64 + ((k & 63) >> 4) * 16 + (m & 15)) * 8 + (k & 7);
}

// ---- scalar conversions -----------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) {
  return __builtin_bit_cast(float, (uint32_t)b << 16);
}
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {  // RNE
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float f16_bits_to_f32(uint16_t b) {
  return (float)__builtin_bit_cast(f16, b);
}
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
  return __builtin_bit_cast(uint16_t, (f16)f);
}

// bf16 -> f16 for the int4 kernels' MFMA operands (the reference's GPTQ / AWQ kernels are fp16-only, gptq.py:54-55;
// we accept bf16 activations by widening them).  SATURATING: a bf16 activation beyond the f16 range (|x| > 65504)
// becomes +-65504 instead of inf -- an inf would turn into NaN in the MFMA and poison the whole output row, a clamped
// outlier costs accuracy on that one element only.  (Values below 2^-24 flush to zero: f16 subnormal range.)
// NaN stays NaN (fminf / fmaxf return the non-NaN operand: without the test an upstream NaN would silently become a finite
// -65504 and the fault would surface as garbage logits instead of NaN -- ADVICE r2).
__device__ __forceinline__ uint16_t bf16_bits_to_f16_bits_sat(uint16_t b) {
  const float f = bf16_bits_to_f32(b);
  const uint16_t h = f32_to_f16_bits(__builtin_fminf(__builtin_fmaxf(f, -65504.f), 65504.f));
  return (f != f) ? (uint16_t)0x7e00 : h;
}

// Storage-type traits: T is a tag for the 16-bit (or 32-bit) activation dtype.
struct Half {
  typedef uint16_t storage;
  static __device__ __forceinline__ float to_f32(uint16_t b) { return f16_bits_to_f32(b); }
  static __device__ __forceinline__ uint16_t from_f32(float f) { return f32_to_f16_bits(f); }
};
struct BFloat {
  typedef uint16_t storage;
  static __device__ __forceinline__ float to_f32(uint16_t b) { return bf16_bits_to_f32(b); }
  static __device__ __forceinline__ uint16_t from_f32(float f) { return f32_to_bf16_bits(f); }
};
struct Float {
  typedef float storage;
  static __device__ __forceinline__ float to_f32(float b) { return b; }
  static __device__ __forceinline__ float from_f32(float f) { return f; }
};

// hipcc selects fptrunc(fmul a, b) as v_fma_mixlo_f16: ONE rounding of the exact product, where the
// reference rounds the fp32 product first and converts second (they differ when the fp32 product
// lands on an f16 tie, ~2^-13 of the elements).  Kernels that promise bit-exactness against an op
// sequence round through this: the asm makes the fp32 value opaque to that selection.
template <typename T>
__device__ __forceinline__ typename T::storage from_f32_exact(float f) {
  asm("" : "+v"(f));
  return T::from_f32(f);
}

// ---- fp8 (OCP e4m3fn / e5m2, native on gfx950) ---------------------------------
// word: 4 packed fp8; returns elements (2*hi_pair, 2*hi_pair+1) as f32.
template <bool E5M2>
__device__ __forceinline__ f32x2 fp8x2_to_f32(uint32_t word, bool hi) {
  if constexpr (E5M2) {
    return hi ? __builtin_amdgcn_cvt_pk_f32_bf8(word, true)
              : __builtin_amdgcn_cvt_pk_f32_bf8(word, false);
  } else {
    return hi ? __builtin_amdgcn_cvt_pk_f32_fp8(word, true)
              : __builtin_amdgcn_cvt_pk_f32_fp8(word, false);
  }
}
// Two packed fp8 (low or high half of a dword) -> a packed pair of T in ONE instruction: gfx950's
// v_cvt_scalef32_pk_{f16,bf16}_{fp8,bf8} with scale 1.0.  Exact (every e4m3 / e5m2 value, subnormals
// included, is representable in f16 and in bf16), and 3-4x fewer VALU than fp8 -> f32 -> T.
template <typename T, bool E5M2, bool HI>
__device__ __forceinline__ uint32_t fp8x2_to_T(uint32_t w) {
  if constexpr (__is_same(T, Half)) {
    if constexpr (E5M2) return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(w, 1.0f, HI));
    else return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, HI));
  } else {
    if constexpr (E5M2) return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(w, 1.0f, HI));
    else return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, HI));
  }
}

template <bool E5M2>
__device__ __forceinline__ float fp8_to_f32(uint8_t b) {
  if constexpr (E5M2) return __builtin_amdgcn_cvt_f32_bf8((uint32_t)b, 0);
  else return __builtin_amdgcn_cvt_f32_fp8((uint32_t)b, 0);
}
// saturating RNE encode of two floats -> low 16 bits of the result.
template <bool E5M2>
__device__ __forceinline__ uint32_t f32x2_to_fp8(float a, float b) {
  if constexpr (E5M2) {
    a = __builtin_fmaxf(-57344.f, __builtin_fminf(a, 57344.f));
    b = __builtin_fmaxf(-57344.f, __builtin_fminf(b, 57344.f));
    return (uint32_t)__builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false) & 0xffffu;
  } else {
    a = __builtin_fmaxf(-448.f, __builtin_fminf(a, 448.f));
    b = __builtin_fmaxf(-448.f, __builtin_fminf(b, 448.f));
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
  }
}

template <bool E5M2, int BYTE>
__device__ __forceinline__ float fp8_byte_to_f32(uint32_t word) {
  if constexpr (E5M2) return __builtin_amdgcn_cvt_f32_bf8(word, BYTE);
  else return __builtin_amdgcn_cvt_f32_fp8(word, BYTE);
}

// (hi16(lo) | hi16(hi) << 16): two f32 that are exactly representable in bf16 ->
// packed bf16 pair.  One v_perm_b32.  NOTE (hipcc 7.2, gfx950): bit-casting
// element [1] of a __builtin_amdgcn_cvt_pk_f32_fp8/bf8 result to an integer is
// MISCOMPILED (element [0]'s register is read instead; float uses of [1] are
// fine) -- feed this helper from the per-byte converts (fp8_byte_to_f32) only.
__device__ __forceinline__ uint32_t f32x2_hi16(float lo, float hi) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x07060302u);
}

// One rotary pair in fp32 with a FIXED operation order (no fp contraction), shared by
// rotary_kernel and rope_cache_kernel so the fused and unfused paths agree bit
// for bit (left to the compiler, fp contraction differs between kernels).
__device__ __forceinline__ void rope_pair(float x, float y, float c, float s, float& xo, float& yo) {
#pragma clang fp contract(off)
  const float xc = x * c, ys = y * s, yc = y * c, xs = x * s;
  xo = xc - ys;
  yo = yc + xs;
}

// ---- wave reductions ------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// round(round(silu(gate)) * up) in the activation type T, on T-rounded inputs
// (activation_kernels.cu:14-17, 25-28).  Shared by silu_and_mul, silu_and_mul_pack and the
// fused GEMM epilogue so that the three paths are bit-identical.
template <typename T>
__device__ __forceinline__ uint16_t silu_mul_bits(float gate, float up) {
  const float s = T::to_f32(T::from_f32(gate / (1.0f + __expf(-gate))));
  return from_f32_exact<T>(s * up);
}

// two fp32 -> packed pair of 16-bit floats with the hardware converters (v_cvt_pk_bf16_f32 is RNE like f32_to_bf16_bits;
// it differs only in the NaN payload it produces)
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
template <bool BF16>
__device__ __forceinline__ uint32_t pack2_16(float a, float b) {
  if constexpr (BF16) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2_hw));
  } else {
    return __builtin_bit_cast(uint32_t, f16x2{(f16)a, (f16)b});
  }
}

// ---- epilogue of the prefill-sized GEMMs (wna16_gemm_large.hip, fp8_gemm_large.hip) ------------------------------
// A wave owns a 128 (m) x 64 (n) tile of 16-bit results, one output ROW per lane (32x32 MFMA C layout: lane = row
// l & 31, 4 consecutive columns 8 q + 4 (l >> 5) per accumulator quad).  Stored straight from that layout every
// instruction scatters 8-byte pieces over 32 rows: measured 1.7 TB/s, a quarter of the kernel at K = 4096.  Instead
// the tile goes through a wave-private 16 KiB LDS region (XOR-swizzled 8-byte slots, conflict-free both ways) and
// leaves as full 128-byte row segments, 16 bytes per lane.
__device__ __forceinline__ void epi_put(unsigned char* region, int row, int c8, u32x2 v) {
  // row: 0..127 inside the wave tile; c8: 8-byte column chunk 0..15
  *reinterpret_cast<u32x2*>(region + row * 128 + ((c8 ^ (((row >> 1) & 7) << 1)) << 3)) = v;
}
__device__ __forceinline__ void epi_flush(const unsigned char* region, uint16_t* c, int64_t ldc, int rows_valid, int lane) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = i * 8 + (lane >> 3), c16 = lane & 7;
    const u32x4 v = *reinterpret_cast<const u32x4*>(region + row * 128 + ((c16 ^ ((row >> 1) & 7)) << 4));
    if (row < rows_valid) *reinterpret_cast<u32x4*>(c + (int64_t)row * ldc + c16 * 8) = v;
  }
}

// The same flush through a buffer descriptor: 32-bit offsets (no 64-bit address per row to keep alive -- with those hipcc
// spilled the row pointers and reloaded them from scratch in front of every store, and a scratch reload is a VMEM load:
// its vmcnt(0) drained the PREVIOUS store, so the 16 stores of a wave went out one acknowledged round trip at a time,
// 16.5k of 93k cycles per 256 x 256 x 4096 tile), rows past the end of C dropped by the descriptor's bounds check instead of
// a branch per store.  c_tile: the wave tile's origin; bytes_left: from there to the end of C.
__device__ __forceinline__ void epi_flush_buf(const unsigned char* region, uint16_t* c_tile, int64_t ldc, int64_t bytes_left, int lane) {
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
      c_tile, 0, (uint32_t)(bytes_left > 0xffffffffll ? 0xffffffffll : (bytes_left < 0 ? 0 : bytes_left)), 0x00020000);
  const int r8 = lane >> 3, c16 = lane & 7;
  const uint32_t voff = (uint32_t)r8 * (uint32_t)ldc * 2u + (uint32_t)c16 * 16u;
  const uint32_t step = 8u * (uint32_t)ldc * 2u;
  u32x4 v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = i * 8 + r8;
    v[i] = *reinterpret_cast<const u32x4*>(region + row * 128 + ((c16 ^ ((row >> 1) & 7)) << 4));
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) __builtin_amdgcn_raw_buffer_store_b128(v[i], rc, voff, (uint32_t)i * step, 0);
}

}  // namespace aphro
