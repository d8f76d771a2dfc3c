// Round-4 probe: what one CU can pull IN per microsecond when a weight stream from HBM (unique bytes, `nt`) and an
// activation stream from L2 (the same 256 KB for every workgroup) share its vector-memory path -- the traffic mix of the
// W4A16 decode GEMM (gate_up: 229 KB of weights + 256 KB of A fragments per CU; down: 115 + 229).  One workgroup per CU,
// NWV waves, each wave `steps` steps of WL weight loads + AL activation loads (1 KiB per load instruction), D steps in
// flight, into registers or straight into LDS (`buffer_load ... lds`).  Launches cycle over > 640 MB of weights.
//   inbound_probe            -> one JSON line per variant
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
typedef __attribute__((address_space(3))) void* lds_ptr;

struct P {
  const uint32_t* w;   // this launch's weights: [workgroups][NWV][steps][WL] KiB
  const uint32_t* a;   // [NWV_total_segments...] 256 KB, shared
  uint32_t* sink;
  int steps;
  uint32_t wbytes, abytes;
};

template <int NWV, int WL, int AL, int D, bool DMA>
__global__ __launch_bounds__(NWV * 64, NWV <= 4 ? 1 : 2) void probe(P p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.w), 0, p.wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.a), 0, p.abytes, 0x00020000);
  const int wbase = (blockIdx.x * NWV + wave) * p.steps * WL * 1024;
  const int abase = wave * p.steps * AL * 1024;      // every workgroup reads the same A; the waves split it (K ranges)
  const int vo = lane * 16;
  constexpr int PER = WL + AL;
  u32x4 acc = {0, 0, 0, 0};
  if constexpr (!DMA) {
    u32x4 ring[D][PER > 0 ? PER : 1];
    auto issue = [&](int s, int slot) {
#pragma unroll
      for (int j = 0; j < WL; ++j) ring[slot][j] = __builtin_amdgcn_raw_buffer_load_b128(rw, vo, wbase + (s * WL + j) * 1024, 2);
#pragma unroll
      for (int j = 0; j < AL; ++j) ring[slot][WL + j] = __builtin_amdgcn_raw_buffer_load_b128(ra, vo, abase + (s * AL + j) * 1024, 0);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, d);
    for (int s0 = 0; s0 < p.steps; s0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int j = 0; j < PER; ++j) acc ^= ring[d][j];
        if (s0 + d + D < p.steps) issue(s0 + d + D, d);
      }
    }
  } else {
    unsigned char* ring = smem + wave * D * PER * 1024;
    auto issue = [&](int s, int slot) {
#pragma unroll
      for (int j = 0; j < WL; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(ring + (slot * PER + j) * 1024), 16, vo, wbase + (s * WL + j) * 1024, 0, 2);
#pragma unroll
      for (int j = 0; j < AL; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(ring + (slot * PER + WL + j) * 1024), 16, vo, abase + (s * AL + j) * 1024, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, d);
    for (int s0 = 0; s0 < p.steps; s0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        // the oldest step has landed when at most (D - 1) steps' loads are still out
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * PER > 63 ? 63 : (D - 1) * PER) : "memory");
        if (s0 + d + D < p.steps) issue(s0 + d + D, d);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    acc[0] = *reinterpret_cast<const uint32_t*>(ring + lane * 4);
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) p.sink[threadIdx.x] = acc[0];
}

static hipStream_t st;
static hipEvent_t e0, e1;
static std::vector<uint32_t*> wbuf;
static uint32_t *abuf, *sink;

template <int NWV, int WL, int AL, int D, bool DMA>
static void run(const char* what, int steps) {
  const int ncopy = (int)wbuf.size();
  const size_t wb = (size_t)256 * NWV * steps * WL * 1024, ab = (size_t)NWV * steps * AL * 1024;
  if (wb > ((size_t)64 << 20) || ab > ((size_t)1 << 20)) { printf("{\"skip\": \"%s\"}\n", what); return; }
  auto kern = probe<NWV, WL, AL, D, DMA>;
  const size_t lds = DMA ? (size_t)NWV * D * (WL + AL) * 1024 : 0;
  if (lds > 160 * 1024) { printf("{\"skip\": \"%s lds\"}\n", what); return; }
  if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int L = 24;
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < L; ++i) {
    P p{wbuf[i % ncopy], abuf, sink, steps, (uint32_t)std::max<size_t>(wb, 16), (uint32_t)std::max<size_t>(ab, 16)};
    hipLaunchKernelGGL(kern, dim3(256), dim3(NWV * 64), lds, st, p);
  }
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 2; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms * 1e3f / (10 * L));
  }
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  printf("{\"probe\": \"%s\", \"waves\": %d, \"WL\": %d, \"AL\": %d, \"depth_steps\": %d, \"dma\": %d, \"steps\": %d, \"W_KB_per_cu\": %.0f, "
         "\"A_KB_per_cu\": %.0f, \"inflight_KB_per_cu\": %d, \"us\": %.2f, \"W_TBps\": %.2f, \"inbound_GBps_per_cu\": %.1f}\n",
         what, NWV, WL, AL, D, (int)DMA, steps, wb / 256.0 / 1024, ab / 1024.0, NWV * D * (WL + AL), best, wb / best / 1e6,
         (wb / 256.0 + ab) / best / 1e3);
  fflush(stdout);
}

int main() {
  CK(hipStreamCreate(&st));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 11; ++i) {
    uint32_t* w;
    CK(hipMalloc(&w, (size_t)64 << 20));
    CK(hipMemset(w, i + 1, (size_t)64 << 20));
    wbuf.push_back(w);
  }
  CK(hipMalloc(&abuf, 1 << 20));
  CK(hipMemset(abuf, 7, 1 << 20));
  CK(hipMalloc(&sink, 4096));
  // gate_up-like: 4 waves x 32 steps x (2 KiB W + 2 KiB A) = 256 KB W + 256 KB A per CU
#define SET(NWV, STEPS)                                        \
  run<NWV, 2, 0, 4, false>("W only", STEPS);                   \
  run<NWV, 2, 0, 8, false>("W only", STEPS);                   \
  run<NWV, 2, 0, 16, false>("W only", STEPS);                  \
  run<NWV, 0, 2, 8, false>("A only", STEPS);                   \
  run<NWV, 0, 2, 16, false>("A only", STEPS);                  \
  run<NWV, 2, 2, 2, false>("W+A", STEPS);                      \
  run<NWV, 2, 2, 4, false>("W+A", STEPS);                      \
  run<NWV, 2, 2, 8, false>("W+A", STEPS);                      \
  run<NWV, 2, 2, 16, false>("W+A", STEPS);                     \
  run<NWV, 1, 2, 8, false>("W+2A (down)", STEPS);              \
  run<NWV, 2, 0, 8, true>("W only", STEPS);                    \
  run<NWV, 2, 0, 16, true>("W only", STEPS);                   \
  run<NWV, 0, 2, 8, true>("A only", STEPS);                    \
  run<NWV, 2, 2, 4, true>("W+A", STEPS);                       \
  run<NWV, 2, 2, 8, true>("W+A", STEPS);                       \
  run<NWV, 1, 2, 8, true>("W+2A (down)", STEPS);
  SET(4, 32)
  SET(8, 16)
  run<8, 2, 2, 4, true>("W+A", 16);
  run<4, 2, 2, 8, false>("W+A long", 128);
  run<4, 2, 0, 8, false>("W only long", 128);
  run<4, 2, 2, 8, true>("W+A long", 128);
  run<4, 2, 0, 8, true>("W only long", 128);
  return 0;
}
