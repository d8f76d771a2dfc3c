// Round-5 lab ("what's weak" 1a): how much of a weight-streaming decode launch's "entry -> first data" time is
// address translation rather than HBM latency?  Torch-free.  NB buffers of BYTES each (default 96 x 60 MB = 5.8 GB: the
// footprint one configs[1] decode step cycles through, so a buffer's translations are as cold when its turn comes again as
// a layer's weights are in the real step).  A HIP graph of L launches streams buffer after buffer the way the W4A16 decode
// GEMMs do (256 workgroups x 4 waves, 16 B per lane, 8 loads in flight); every workgroup stamps (shader clock) its entry,
// the arrival of its first load and its end.  Arms:
//   none          stream launches only
//   dummy         a touch launch before every stream launch that reads ONE line (same launch count as the touch arms)
//   touch S       a touch launch before every stream launch reads one 64 B line per S bytes of THAT launch's buffer from
//                 every XCD (8 x k workgroups, round-robin over the XCDs) -- translations warm, data (all but 64 B per S) cold
//   ahead S       the touch launch before stream launch i warms buffer i + 1: what a product kernel would do for its
//                 successor (the touch overlaps nothing here; it shows whether warmth survives a launch boundary)
//   inline S      no touch launch: the first 32 workgroups of stream launch i read one line per S bytes of buffer i + 1 at
//                 their entry (the form a product kernel would carry)
//   warm          every launch streams the SAME buffer (data in the Infinity Cache / L2 and translations warm)
// Build: hipcc --offload-arch=gfx950 -O3 tools/tlb_probe.hip -o tools/bin/tlb_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Stamp { uint32_t first, end; };

__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ buf, size_t n16, Stamp* __restrict__ stamps,
                                                     uint32_t* __restrict__ sink, const uint8_t* __restrict__ next, size_t next_bytes,
                                                     size_t next_stride) {
  const uint64_t t0 = __builtin_readcyclecounter();
  uint32_t tacc = 0;
  if (next != nullptr && blockIdx.x < 32) {          // "inline" arm: 4 workgroups per XCD warm the NEXT launch's translations
    const size_t pages = (next_bytes + next_stride - 1) / next_stride;
    for (size_t pg = (size_t)(blockIdx.x >> 3) * 256 + threadIdx.x; pg < pages; pg += 4 * 256)
      tacc ^= __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(next + pg * next_stride));
  }
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const size_t per_wave = n16 / ((size_t)gridDim.x * 4);           // (multiple of 64 * 8 by construction)
  const u32x4* p = buf + (size_t)wave * per_wave + lane;
  const int steps = (int)(per_wave / 64);
  u32x4 acc = {0, 0, 0, 0};
  u32x4 v0 = __builtin_nontemporal_load(p);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  acc ^= v0;
  const uint64_t t1 = __builtin_readcyclecounter();
  for (int s = 1; s + 8 <= steps; s += 8) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + (size_t)(s + u) * 64);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u];
  }
  const uint64_t t2 = __builtin_readcyclecounter();
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3] ^ tacc) == 0x12345678u) sink[wave] = 1;
  if (threadIdx.x == 0) stamps[blockIdx.x] = Stamp{(uint32_t)(t1 - t0), (uint32_t)(t2 - t0)};
}

// one 64 B line per `stride` bytes; every workgroup group (blockIdx % 8 = XCD) covers the whole buffer
__global__ __launch_bounds__(64) void touch_kernel(const uint8_t* __restrict__ buf, size_t bytes, size_t stride, uint32_t* __restrict__ sink) {
  const int per_xcd = gridDim.x >> 3, slot = blockIdx.x >> 3;
  const size_t pages = (bytes + stride - 1) / stride;
  uint32_t acc = 0;
  for (size_t pg = (size_t)slot * 64 + threadIdx.x; pg < pages; pg += (size_t)per_xcd * 64)
    acc ^= __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(buf + pg * stride));
  if (acc == 0x12345678u) sink[blockIdx.x] = 1;
}

static double median(std::vector<uint32_t> v, double q = 0.5) {
  std::sort(v.begin(), v.end());
  return v.empty() ? 0.0 : (double)v[(size_t)(q * (v.size() - 1))];
}

int main(int argc, char** argv) {
  const int NB = argc > 1 ? atoi(argv[1]) : 96;
  const size_t MB = argc > 2 ? (size_t)atoi(argv[2]) : 60;
  const int L = 24, WGS = 256;
  const size_t bytes = MB << 20, n16 = bytes / 16 / (WGS * 4 * 64 * 8) * (WGS * 4 * 64 * 8);
  std::vector<uint8_t*> bufs(NB);
  for (int i = 0; i < NB; ++i) {
    CHECK(hipMalloc((void**)&bufs[i], bytes));
    CHECK(hipMemset(bufs[i], i + 1, bytes));
  }
  Stamp* stamps;
  uint32_t* sink;
  CHECK(hipMalloc((void**)&stamps, sizeof(Stamp) * WGS * L));
  CHECK(hipMalloc((void**)&sink, 1 << 20));
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  struct Arm { const char* name; int mode; size_t stride; int touch_wgs; };
  std::vector<Arm> arms = {
      {"none", 0, 0, 0},           {"dummy", 1, 0, 8},           {"touch 2M", 2, 2u << 20, 8},  {"touch 64K", 2, 64u << 10, 8},
      {"touch 4K", 2, 4u << 10, 32}, {"ahead 2M", 3, 2u << 20, 8}, {"ahead 64K", 3, 64u << 10, 8}, {"ahead 4K", 3, 4u << 10, 32},
      {"inline 2M", 5, 2u << 20, 0}, {"inline 64K", 5, 64u << 10, 0}, {"inline 4K", 5, 4u << 10, 0},
      {"warm", 4, 0, 0},           {"none", 0, 0, 0}};
  for (const Arm& a : arms) {
    // graphs[g] covers launches g * L .. g * L + L - 1 of the cycle over the NB buffers
    const int ngraph = NB / L > 0 ? NB / L : 1;
    std::vector<hipGraphExec_t> execs;
    for (int g = 0; g < ngraph; ++g) {
      hipGraph_t graph;
      CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < L; ++i) {
        const int b = a.mode == 4 ? 0 : (g * L + i) % NB;
        if (a.mode == 1) hipLaunchKernelGGL(touch_kernel, dim3(a.touch_wgs), dim3(64), 0, st, (const uint8_t*)sink, (size_t)64, (size_t)64, sink + 4096);
        if (a.mode == 2) hipLaunchKernelGGL(touch_kernel, dim3(a.touch_wgs), dim3(64), 0, st, bufs[b], bytes, a.stride, sink + 4096);
        if (a.mode == 3) hipLaunchKernelGGL(touch_kernel, dim3(a.touch_wgs), dim3(64), 0, st, bufs[(b + 1) % NB], bytes, a.stride, sink + 4096);
        hipLaunchKernelGGL(stream_kernel, dim3(WGS), dim3(256), 0, st, (const u32x4*)bufs[b], n16, stamps + (size_t)i * WGS, sink,
                           a.mode == 5 ? (const uint8_t*)bufs[(b + 1) % NB] : (const uint8_t*)nullptr, bytes, a.stride ? a.stride : (size_t)4096);
      }
      CHECK(hipStreamEndCapture(st, &graph));
      hipGraphExec_t ex;
      CHECK(hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0));
      execs.push_back(ex);
      CHECK(hipGraphDestroy(graph));
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) for (auto ex : execs) CHECK(hipGraphLaunch(ex, st));
    CHECK(hipStreamSynchronize(st));
    double best = 1e30, sum = 0;
    const int reps = 5;
    for (int r = 0; r < reps; ++r) {
      CHECK(hipEventRecord(e0, st));
      for (auto ex : execs) CHECK(hipGraphLaunch(ex, st));
      CHECK(hipEventRecord(e1, st));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / (execs.size() * L);
      best = std::min(best, us); sum += us;
    }
    std::vector<Stamp> h((size_t)WGS * L);
    CHECK(hipMemcpy(h.data(), stamps, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
    std::vector<uint32_t> first, end;
    for (auto& s : h) { first.push_back(s.first); end.push_back(s.end); }
    printf("{\"arm\": \"%s\", \"buffers\": %d, \"MB\": %zu, \"us_per_pair\": %.2f, \"us_mean\": %.2f, \"first_cycles_p10\": %.0f, \"first_cycles_p50\": %.0f, "
           "\"first_cycles_p90\": %.0f, \"end_cycles_p50\": %.0f, \"end_cycles_p90\": %.0f}\n",
           a.name, NB, MB, best, sum / reps, median(first, 0.1), median(first), median(first, 0.9), median(end), median(end, 0.9));
    fflush(stdout);
    for (auto ex : execs) CHECK(hipGraphExecDestroy(ex));
  }
  return 0;
}
