// Round-4 probe: can DEPENDENT streaming kernels overlap their launch gap and first-data latency on MI355X without a
// persistent kernel?  A chain of weight-streaming launches (256 workgroups x 4 waves, 256 KB per CU each, distinct
// weights per launch) where launch k must not consume "activations" before launch k - 1 has completely finished:
//   serial      one stream, plain launches: the dependency is the stream order (today's decode step);
//   two-stream  launches alternate between two streams of one captured graph (no event between them except the fork at
//               the head and the join at the tail); launch k spins on a device counter that every workgroup of launch k - 1
//               bumps after its last (write-through) store -- so its workgroups may start, and stream their weights,
//               as soon as CUs free up, with no kernel boundary in between.
// Prints us per launch for both.  Spins are bounded (a protocol bug must not hang the GPU).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                             \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                            \
    }                                                                                     \
  } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct P {
  const uint32_t* w;       // [256][4][steps][2] KiB
  const uint32_t* act;     // 64 KB written by the previous launch (read after the dependency is met)
  uint32_t* out;           // 64 KB this launch writes (write-through)
  unsigned* dep;           // counter of the previous launch (NULL: none / stream order)
  unsigned* done;          // this launch's counter
  unsigned dep_target;
  int steps;
  uint32_t wbytes;
  unsigned* timeouts;
};

template <int D>
__global__ __launch_bounds__(256, 1) void chain_kernel(P p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.w), 0, p.wbytes, 0x00020000);
  const int wbase = (blockIdx.x * 4 + wave) * p.steps * 2048;
  const int vo = lane * 16;
  u32x4 ring[D][2];
  u32x4 acc = {0, 0, 0, 0};
  // weights first: they do not depend on the previous launch
#pragma unroll
  for (int d = 0; d < D; ++d) {
    ring[d][0] = __builtin_amdgcn_raw_buffer_load_b128(rw, vo, wbase + d * 2048, 2);
    ring[d][1] = __builtin_amdgcn_raw_buffer_load_b128(rw, vo, wbase + d * 2048 + 1024, 2);
  }
  // the dependency: every workgroup of the previous launch has bumped its counter after its last store
  if (p.dep != nullptr) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
      int spins = 0;
      while (__hip_atomic_load(p.dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.dep_target && spins < (1 << 22)) {
        __builtin_amdgcn_s_sleep(2);
        ++spins;
      }
      if (spins >= (1 << 22)) atomicAdd(p.timeouts, 1u);
      ok = 1;
    }
    __syncthreads();
  }
  // "activations": coherent loads of what the previous launch wrote (a few KB per workgroup)
  uint32_t a;
  asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(p.act + (blockIdx.x * 256 + threadIdx.x) % 16384) : "memory");
  acc[0] ^= a;
  for (int s0 = 0; s0 < p.steps; s0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      acc ^= ring[d][0];
      acc ^= ring[d][1];
      if (s0 + d + D < p.steps) {
        ring[d][0] = __builtin_amdgcn_raw_buffer_load_b128(rw, vo, wbase + (s0 + d + D) * 2048, 2);
        ring[d][1] = __builtin_amdgcn_raw_buffer_load_b128(rw, vo, wbase + (s0 + d + D) * 2048 + 1024, 2);
      }
    }
  }
  // output: write-through, then the counter
  const uint32_t v = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  uint32_t* dst = p.out + (blockIdx.x * 256 + threadIdx.x) % 16384;
  asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(dst), "v"(v) : "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void zero_kernel(unsigned* c, int n) {
  if ((int)threadIdx.x < n) c[threadIdx.x] = 0;
}

int main() {
  const int steps = 32, L = 24, NW = 11;
  hipStream_t s0, s1;
  CK(hipStreamCreate(&s0));
  CK(hipStreamCreate(&s1));
  hipEvent_t e0, e1, fork, join;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  std::vector<uint32_t*> w(NW);
  const size_t wb = (size_t)256 * 4 * steps * 2048;
  for (int i = 0; i < NW; ++i) { CK(hipMalloc(&w[i], wb)); CK(hipMemset(w[i], i + 1, wb)); }
  uint32_t* act[2];
  for (int i = 0; i < 2; ++i) { CK(hipMalloc(&act[i], 65536)); CK(hipMemset(act[i], 0, 65536)); }
  unsigned *cnt, *timeouts;
  CK(hipMalloc(&cnt, 64 * sizeof(unsigned)));
  CK(hipMalloc(&timeouts, sizeof(unsigned)));
  CK(hipMemset(timeouts, 0, sizeof(unsigned)));
  auto time_graph = [&](hipGraphExec_t ge) {
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s0));
    CK(hipStreamSynchronize(s0));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s0));
      for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s0));
      CK(hipEventRecord(e1, s0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms * 1e3f / (10 * L));
    }
    return best;
  };
  for (int mode = 0; mode < 3; ++mode) {
    // 0: serial, no counters; 1: serial WITH the counter protocol (its cost alone); 2: two streams + counters
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(64), 0, s0, cnt, 64);
    if (mode == 2) {
      CK(hipEventRecord(fork, s0));
      CK(hipStreamWaitEvent(s1, fork, 0));
    }
    for (int k = 0; k < L; ++k) {
      P p{w[k % NW], act[(k + 1) & 1], act[k & 1], (mode > 0 && k > 0) ? cnt + (k - 1) : nullptr, cnt + k, 256u, steps, (uint32_t)wb, timeouts};
      hipLaunchKernelGGL(chain_kernel<8>, dim3(256), dim3(256), 0, (mode == 2 && (k & 1)) ? s1 : s0, p);
    }
    if (mode == 2) {
      CK(hipEventRecord(join, s1));
      CK(hipStreamWaitEvent(s0, join, 0));
    }
    CK(hipStreamEndCapture(s0, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const float us = time_graph(ge);
    unsigned to = 0;
    CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost));
    printf("{\"mode\": \"%s\", \"us_per_launch\": %.2f, \"MB_per_launch\": %.1f, \"TBps\": %.2f, \"spin_timeouts\": %u}\n",
           mode == 0 ? "serial" : (mode == 1 ? "serial + counters" : "two streams + counters"), us, wb / 1e6, wb / us / 1e6, to);
    fflush(stdout);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  // eager launches (no graph): is the two-stream cost the graph's cross-stream machinery or the hardware's?
  for (int mode = 0; mode < 2; ++mode) {
    auto run = [&]() {
      hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(64), 0, s0, cnt, 64);
      if (mode == 1) { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
      for (int k = 0; k < L; ++k) {
        P p{w[k % NW], act[(k + 1) & 1], act[k & 1], k > 0 ? cnt + (k - 1) : nullptr, cnt + k, 256u, steps, (uint32_t)wb, timeouts};
        hipLaunchKernelGGL(chain_kernel<8>, dim3(256), dim3(256), 0, (mode == 1 && (k & 1)) ? s1 : s0, p);
      }
      if (mode == 1) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
    };
    for (int i = 0; i < 3; ++i) run();
    CK(hipStreamSynchronize(s0));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s0));
      for (int i = 0; i < 10; ++i) run();
      CK(hipEventRecord(e1, s0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms * 1e3f / (10 * L));
    }
    unsigned to = 0;
    CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost));
    printf("{\"mode\": \"eager %s\", \"us_per_launch\": %.2f, \"spin_timeouts\": %u}\n", mode == 0 ? "serial + counters" : "two streams + counters", best, to);
  }
  return 0;
}
