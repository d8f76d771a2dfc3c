// Round-5 lab: what does instruction fetch cost a decode-sized launch?  The W4A16 stream kernels are 20-30 KB of straight-line
// code that every wave executes ONCE (fully unrolled k loop), each launch starts with a cold instruction cache, and the
// graph alternates between seven different kernels per layer.  Torch-free.  Two forms of the same instruction stream
// (N KB of dependent v_add_u32 / v_add3_u32, 256 workgroups x 4 waves = one wave per SIMD):
//   line<KB, W>  straight line, executed once
//   loop<KB, W>  a KB/16 body executed 16 times (same instruction count; the body is fetched once)
// W = 4: 4-byte VOP2 instructions, W = 8: 8-byte VOP3.  Every wave stamps entry -> end (shader clock); launches alternate
// between TWO different kernels of the same kind (a / b twin) so that no launch finds its own code in the instruction cache
// even if the cache survived a launch boundary.  Reported: in-kernel cycles (median over workgroups, last launch), cycles
// per instruction, us per launch in a 24-launch graph.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ifetch_probe.hip -o tools/bin/ifetch_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define STR2(x) #x
#define STR(x) STR2(x)

#define BODY4(n) ".rept " STR(n) "\n v_add_u32 %0, %0, %1\n .endr\n"
#define BODY8(n) ".rept " STR(n) "\n v_add3_u32 %0, %0, %1, %1\n .endr\n"

template <int KB, int W, int TWIN>
__global__ __launch_bounds__(256) void line_kernel(uint32_t* __restrict__ out, uint32_t* __restrict__ stamps, uint32_t seed) {
  const uint64_t t0 = __builtin_readcyclecounter();
  uint32_t x = threadIdx.x + TWIN, y = seed;
  if constexpr (W == 4) asm volatile(BODY4(256) : "+v"(x) : "v"(y));     // (1 KB per statement)
  else asm volatile(BODY8(128) : "+v"(x) : "v"(y));
#pragma unroll
  for (int i = 1; i < KB; ++i) {
    if constexpr (W == 4) asm volatile(BODY4(256) : "+v"(x) : "v"(y));
    else asm volatile(BODY8(128) : "+v"(x) : "v"(y));
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (x == 0x12345678u) out[blockIdx.x] = x;
  if (threadIdx.x == 0) stamps[blockIdx.x] = (uint32_t)(t1 - t0);
}

template <int KB, int W, int TWIN>
__global__ __launch_bounds__(256) void loop_kernel(uint32_t* __restrict__ out, uint32_t* __restrict__ stamps, uint32_t seed) {
  const uint64_t t0 = __builtin_readcyclecounter();
  uint32_t x = threadIdx.x + TWIN, y = seed;
#pragma unroll 1
  for (int it = 0; it < 16; ++it) {
#pragma unroll
    for (int i = 0; i < KB; ++i) {                                         // KB x 64 B per pass = KB / 16 KB
      if constexpr (W == 4) asm volatile(BODY4(16) : "+v"(x) : "v"(y));
      else asm volatile(BODY8(8) : "+v"(x) : "v"(y));
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (x == 0x12345678u) out[blockIdx.x] = x;
  if (threadIdx.x == 0) stamps[blockIdx.x] = (uint32_t)(t1 - t0);
}

typedef void (*kern_t)(uint32_t*, uint32_t*, uint32_t);

static void run(const char* name, int kb, int w, kern_t ka, kern_t kb_, int wgs, hipStream_t st, uint32_t* out, uint32_t* stamps) {
  const int L = 24;
  hipGraph_t graph;
  hipGraphExec_t ex;
  CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < L; ++i) hipLaunchKernelGGL((i & 1) ? kb_ : ka, dim3(wgs), dim3(256), 0, st, out, stamps, (uint32_t)i);
  CHECK(hipStreamEndCapture(st, &graph));
  CHECK(hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CHECK(hipGraphLaunch(ex, st));
  CHECK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  double best = 1e30;
  for (int r = 0; r < 5; ++r) {
    CHECK(hipEventRecord(e0, st));
    CHECK(hipGraphLaunch(ex, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms * 1e3 / L);
  }
  std::vector<uint32_t> h(wgs);
  CHECK(hipMemcpy(h.data(), stamps, 4 * wgs, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double ninstr = (double)kb * 1024 / w;
  printf("{\"kernel\": \"%s\", \"KB\": %d, \"instr_bytes\": %d, \"workgroups\": %d, \"us_per_launch\": %.2f, \"cycles_p10\": %u, \"cycles_p50\": %u, "
         "\"cycles_p90\": %u, \"cycles_per_instr_p50\": %.2f, \"cycles_per_64B_line_p50\": %.1f}\n",
         name, kb, w, wgs, best, h[wgs / 10], h[wgs / 2], h[wgs * 9 / 10], h[wgs / 2] / ninstr, h[wgs / 2] / (kb * 16.0));
  fflush(stdout);
  CHECK(hipGraphExecDestroy(ex));
  CHECK(hipGraphDestroy(graph));
}

#define RUN(KB, W)                                                                                         \
  run("line", KB, W, line_kernel<KB, W, 0>, line_kernel<KB, W, 1>, wgs, st, out, stamps);                  \
  run("loop", KB, W, loop_kernel<KB, W, 0>, loop_kernel<KB, W, 1>, wgs, st, out, stamps);

int main(int argc, char** argv) {
  uint32_t *out, *stamps;
  CHECK(hipMalloc((void**)&out, 1 << 16));
  CHECK(hipMalloc((void**)&stamps, 1 << 16));
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  for (int wgs : {256, 8}) {
    RUN(4, 4) RUN(16, 4) RUN(32, 4) RUN(64, 4)
    RUN(4, 8) RUN(16, 8) RUN(32, 8) RUN(64, 8)
  }
  return 0;
}
