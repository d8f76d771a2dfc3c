// How fast can ONE CU stream from HBM?  Launch G workgroups (<= 256: one per CU) of W waves,
// each wave streaming `kib` KiB contiguously with U x 1 KiB loads in flight.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int U>
__global__ __launch_bounds__(1024) void stream(const uint8_t* __restrict__ base, int kib, uint32_t* out) {
  const int lane = threadIdx.x & 63;
  const size_t wave_global = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint8_t* p = base + wave_global * (size_t)kib * 1024 + lane * 16;
  u32x4 acc = {0, 0, 0, 0};
  for (int s = 0; s < kib; s += U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + (size_t)(s + u) * 1024));
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[wave_global] = 1;
}
template <int U>
void run(const uint8_t* buf, uint32_t* out, int G, int W, int kib) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((stream<U>), dim3(G), dim3(W * 64), 0, 0, buf, kib, out);
  CHECK(hipDeviceSynchronize());
  const int iters = 5;
  CHECK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((stream<U>), dim3(G), dim3(W * 64), 0, 0, buf + (size_t)(i + 1) * 67108864, kib, out);
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  double us = ms * 1e3 / iters, bytes = (double)G * W * kib * 1024;
  printf("G=%4d waves/WG=%2d U=%2d KiB/wave=%4d : %8.2f us  total %7.1f GB/s  per-WG %6.1f GB/s\n", G, W, U, kib, us, bytes / us / 1e3, bytes / us / 1e3 / G);
}
int main() {
  uint8_t* buf; uint32_t* out;
  CHECK(hipMalloc(&buf, (size_t)3 << 29)); CHECK(hipMemset(buf, 1, (size_t)3 << 29)); CHECK(hipMalloc(&out, 1 << 22));
  for (int G : {8, 64, 128, 256, 512}) { run<4>(buf, out, G, 8, 256); }
  for (int G : {64, 256}) { run<4>(buf, out, G, 4, 256); run<8>(buf, out, G, 4, 256); run<16>(buf, out, G, 4, 256); run<16>(buf, out, G, 8, 256); run<4>(buf, out, G, 16, 128); run<16>(buf, out, G, 16, 128); }
  // short per-wave streams like the GEMM (32 KiB per wave)
  for (int G : {256, 512}) { run<4>(buf, out, G, 4, 32); run<8>(buf, out, G, 4, 32); run<4>(buf, out, G, 8, 32); run<8>(buf, out, G, 8, 32); }
  run<8>(buf, out, 448, 4, 32); run<8>(buf, out, 448, 8, 16); run<8>(buf, out, 1792, 4, 8); run<8>(buf, out, 1792, 1, 32);
  return 0;
}
