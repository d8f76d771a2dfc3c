// Round-5 lab: how many bytes per clock can ONE CU pull through its vector-memory path when the data is on chip?  The stamps
// of the W4A16 stream kernels (profiles/r5_decode_experiments.txt (2)) price their prologues -- 14-20 loads of 1 KB per wave,
// four waves, nothing to wait for -- at ~32 B/clk per CU; this probe measures the same thing without a GEMM around it.
// 256 workgroups (one per CU), W waves each; every wave re-reads a private window of `win` bytes with 16 B per lane loads, 8 in
// flight, `iters` times: win = 8 KB per wave stays in the CU's 32 KB L1, 64 KB per wave lives in the XCD's L2.  Shader-clock
// stamps per wave; B/clk/CU = waves x bytes per wave / cycles.
// Build: hipcc --offload-arch=gfx950 -O3 tools/l2rate_probe.hip -o tools/bin/l2rate_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>      // 0 plain, 1 nt, 2 sc1 (L1 bypass)
__global__ __launch_bounds__(1024) void rate_kernel(const u32x4* __restrict__ buf, int win16, int iters, uint32_t* __restrict__ stamps,
                                                    uint32_t* __restrict__ sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const u32x4* p = buf + ((size_t)blockIdx.x * nw + wave) * win16 + lane;
  const int steps = win16 / 64;                       // 1 KB instructions per pass over the window (multiple of 8)
  u32x4 acc = {0, 0, 0, 0};
  // one warm pass (fills L1 / L2), then the timed passes
  for (int s = 0; s < steps; ++s) acc ^= p[(size_t)s * 64];
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    for (int s = 0; s < steps; s += 8) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const u32x4* q = p + (size_t)(s + u) * 64;
        if (MODE == 1) v[u] = __builtin_nontemporal_load(q);
        else if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u]) : "v"(q) : "memory");
        else v[u] = *q;
      }
      if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[blockIdx.x] = 1;
  if (lane == 0) stamps[blockIdx.x * nw + wave] = (uint32_t)(t1 - t0);
}

template <int MODE>
static void run(const char* mode, const u32x4* buf, int waves, int win_kb, int iters, uint32_t* stamps, uint32_t* sink) {
  const int wgs = 256, win16 = win_kb * 1024 / 16;
  hipLaunchKernelGGL((rate_kernel<MODE>), dim3(wgs), dim3(waves * 64), 0, 0, buf, win16, iters, stamps, sink);
  CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL((rate_kernel<MODE>), dim3(wgs), dim3(waves * 64), 0, 0, buf, win16, iters, stamps, sink);
  CHECK(hipDeviceSynchronize());
  std::vector<uint32_t> h((size_t)wgs * waves);
  CHECK(hipMemcpy(h.data(), stamps, 4 * h.size(), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double bytes_per_wave = (double)win_kb * 1024 * iters, cyc = h[h.size() / 2];
  printf("{\"loads\": \"%s\", \"waves_per_cu\": %d, \"window_KB_per_wave\": %d, \"window_KB_per_cu\": %d, \"cycles_p50\": %.0f, "
         "\"B_per_clk_per_cu\": %.1f, \"cycles_per_1KB_instr_per_cu\": %.1f}\n",
         mode, waves, win_kb, win_kb * waves, cyc, waves * bytes_per_wave / cyc, cyc / (waves * bytes_per_wave / 1024));
  fflush(stdout);
}

int main() {
  u32x4* buf;
  uint32_t *stamps, *sink;
  const size_t bytes = (size_t)256 * 16 * 64 * 1024;          // up to 16 waves x 64 KB per workgroup
  CHECK(hipMalloc((void**)&buf, bytes));
  CHECK(hipMemset(buf, 1, bytes));
  CHECK(hipMalloc((void**)&stamps, 1 << 20));
  CHECK(hipMalloc((void**)&sink, 1 << 16));
  // (the L2 of an XCD is 4 MB for 32 CUs: keep a CU's footprint <= 64 KB so that the windows stay resident)
  run<0>("plain", buf, 1, 8, 256, stamps, sink);        // 8 KB per CU: L1
  run<0>("plain", buf, 4, 8, 256, stamps, sink);        // 32 KB per CU: the whole L1
  run<0>("plain", buf, 1, 64, 32, stamps, sink);        // 64 KB per CU: L2
  run<0>("plain", buf, 4, 16, 128, stamps, sink);
  run<0>("plain", buf, 8, 8, 256, stamps, sink);
  run<1>("nt", buf, 4, 16, 128, stamps, sink);
  run<2>("sc1", buf, 4, 16, 128, stamps, sink);
  run<1>("nt", buf, 8, 8, 256, stamps, sink);
  return 0;
}
