// Probes (1) whether v_mfma_f32_16x16x32_f16 consumes f16 SUBNORMAL inputs exactly,
// (2) issue cost of the VALU ops the int4 unpack can be built from.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void mfma_subnormal(float* out) {
  int lane = threadIdx.x;
  // A[m][k] = 1 + m/16 (exact f16), B[k][n] = (k%16 + n%3) * 2^-24 as raw subnormal bits
  f16x8 a; u32x4 braw;
  for (int j = 0; j < 8; ++j) a[j] = (_Float16)(1.0f + (lane & 15) / 16.0f);
  int g = lane >> 4, n = lane & 15;
  for (int j = 0; j < 4; ++j) {
    uint32_t lo = ((8 * g + 2 * j) % 16 + n % 3) & 0xf, hi = ((8 * g + 2 * j + 1) % 16 + n % 3) & 0xf;
    braw[j] = lo | (hi << 16);
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(f16x8, braw), c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + n] = c[r] * 16777216.0f;
}

template <int OP>
__global__ void valu_cost(uint32_t* out, uint32_t seed, long long* cycles) {
  uint32_t x0 = seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 11, x5 = x0 * 13, x6 = x0 * 17, x7 = x0 * 19;
  uint32_t m = seed | 0x000f000f, k = 0x64006400;
  asm volatile("" : "+v"(k));
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 256; ++it) {
#define R8(EXPR) x0 = EXPR(x0); x1 = EXPR(x1); x2 = EXPR(x2); x3 = EXPR(x3); x4 = EXPR(x4); x5 = EXPR(x5); x6 = EXPR(x6); x7 = EXPR(x7);
    if (OP == 0) {
#define E0(x) ((x & m) + 1)
      R8(E0) R8(E0) R8(E0) R8(E0)
    } else if (OP == 1) {
#define E1(x) ((x >> 4) ^ m)
      R8(E1) R8(E1) R8(E1) R8(E1)
    } else if (OP == 2) {  // v_pk_add_f16
#define E2(x) __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, x) + __builtin_bit_cast(f16x2, m))
      R8(E2) R8(E2) R8(E2) R8(E2)
    } else if (OP == 3) {  // v_pk_fma_f16
#define E3(x) __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, x) * __builtin_bit_cast(f16x2, m) + __builtin_bit_cast(f16x2, k))
      R8(E3) R8(E3) R8(E3) R8(E3)
    } else if (OP == 4) {  // v_fma_f32
#define E4(x) __builtin_bit_cast(uint32_t, __builtin_fmaf(__builtin_bit_cast(float, x), 1.0001f, 0.5f))
      R8(E4) R8(E4) R8(E4) R8(E4)
    } else if (OP == 5) {  // v_and_or_b32
      uint32_t r;
#define E5(x) ({ asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(m), "v"(k)); r; })
      R8(E5) R8(E5) R8(E5) R8(E5)
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
  if (threadIdx.x == 0) cycles[0] = t1 - t0;
}

int main() {
  float* d; CHECK(hipMalloc(&d, 256 * 4));
  hipLaunchKernelGGL(mfma_subnormal, dim3(1), dim3(64), 0, 0, d);
  float h[256]; CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  int bad = 0;
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
    double ref = 0; for (int k = 0; k < 32; ++k) ref += (1.0 + m / 16.0) * ((k % 16 + n % 3) & 0xf);
    if (h[m * 16 + n] != (float)ref) { if (bad < 5) printf("mismatch m=%d n=%d got %g ref %g\n", m, n, h[m * 16 + n], ref); ++bad; }
  }
  printf("MFMA f16 subnormal inputs: %s (%d mismatches)\n", bad ? "NOT exact" : "EXACT", bad);
  uint32_t* o; long long* cyc; CHECK(hipMalloc(&o, 4096)); CHECK(hipMalloc(&cyc, 8));
  const char* names[] = {"v_and+v_add (2 ops)", "v_lshr+v_xor (2 ops)", "v_pk_add_f16", "v_pk_fma_f16", "v_fma_f32", "v_and_or_b32"};
  for (int op = 0; op < 6; ++op) {
    for (int waves = 1; waves <= 2; ++waves) {
      switch (op) {
        case 0: hipLaunchKernelGGL(valu_cost<0>, dim3(1), dim3(256 * waves), 0, 0, o, 12345u, cyc); break;
        case 1: hipLaunchKernelGGL(valu_cost<1>, dim3(1), dim3(256 * waves), 0, 0, o, 12345u, cyc); break;
        case 2: hipLaunchKernelGGL(valu_cost<2>, dim3(1), dim3(256 * waves), 0, 0, o, 12345u, cyc); break;
        case 3: hipLaunchKernelGGL(valu_cost<3>, dim3(1), dim3(256 * waves), 0, 0, o, 12345u, cyc); break;
        case 4: hipLaunchKernelGGL(valu_cost<4>, dim3(1), dim3(256 * waves), 0, 0, o, 12345u, cyc); break;
        default: hipLaunchKernelGGL(valu_cost<5>, dim3(1), dim3(256 * waves), 0, 0, o, 12345u, cyc); break;
      }
      long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
      printf("%-22s waves/SIMD=%d : %6.2f cycles per (8 independent) source statement group/8 = %.2f per statement\n",
             names[op], waves, (double)c / (256.0 * 4), (double)c / (256.0 * 32));
    }
  }
  return 0;
}
