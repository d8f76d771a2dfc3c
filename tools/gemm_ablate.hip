// Ablation driver for the int4 decode GEMM (not part of the product build).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DABL_NO_A ...] tools/gemm_ablate.hip -o /tmp/abl && /tmp/abl K N M
// Includes the kernel translation unit directly and launches the fast kernel
// on random data, cycling over enough weight copies to defeat the Infinity Cache.
#include "../aphrodite_engine_amd/csrc/wna16_gemm.hip"
#include <vector>
namespace aphro { void set_error(const char*, ...) {} }
#include <stdio.h>
#include <stdlib.h>
#ifndef ABL_VEC
#define ABL_VEC 4
#endif
#ifndef ABL_NSEG
#define ABL_NSEG 8
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
  int K = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 28672, M = argc > 3 ? atoi(argv[3]) : 32;
  const int G = K / 128, copies = 8;
  const int ksplit = K / (128 * aphro::FNW * ABL_NSEG);
  size_t wbytes = (size_t)K / 8 * N * 4;
  std::vector<uint32_t*> qw(copies);
  std::vector<uint32_t> h(wbytes / 4);
  for (auto& x : h) x = (uint32_t)rand() * 2654435761u;
  for (int i = 0; i < copies; ++i) { CK(hipMalloc(&qw[i], wbytes)); CK(hipMemcpy(qw[i], h.data(), wbytes, hipMemcpyHostToDevice)); }
  uint32_t* qz; uint16_t *sc, *apk, *c; float* part;
  CK(hipMalloc(&qz, (size_t)G * N / 2)); CK(hipMemset(qz, 0x77, (size_t)G * N / 2));
  CK(hipMalloc(&sc, (size_t)G * N * 2)); CK(hipMemset(sc, 0x1c, (size_t)G * N * 2));
  int mtiles = (M + 15) / 16;
  CK(hipMalloc(&apk, (size_t)mtiles * 16 * K * 2)); CK(hipMemset(apk, 0x3c, (size_t)mtiles * 16 * K * 2));
  CK(hipMalloc(&c, (size_t)M * N * 2));
  CK(hipMalloc(&part, (size_t)ksplit * M * N * 4 + 4096));
  aphro::Wna16Params p{};
  p.a = nullptr; p.apk = apk; p.qz = qz; p.sc = sc; p.c = c; p.partial = part;
  p.M = M; p.N = N; p.K = K; p.lda = K; p.group_size = 128; p.ksteps_per_split = 4 * aphro::FNW * ABL_NSEG;
  p.ksplit = ksplit; p.zero_offset = 1; p.gshift = 0; p.force_partial = 0;
  constexpr int MT = 2;
  dim3 grid(N / (16 * ABL_VEC), ksplit, (M + 16 * MT - 1) / (16 * MT));
  size_t lds = (size_t)aphro::FNW * MT * ABL_VEC * 64 * 4 * sizeof(float);
#ifdef ABL_TRACE
  uint64_t* tb; CK(hipMalloc(&tb, (size_t)grid.x * grid.y * aphro::FNW * 16 * 8)); CK(hipMemset(tb, 0, (size_t)grid.x * grid.y * aphro::FNW * 16 * 8));
  p.a = (const uint16_t*)tb;
#endif
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](int i) {
    p.qw = qw[i % copies];
    hipLaunchKernelGGL((aphro::wna16_gemm_kernel<aphro::Half, ABL_VEC, MT, ABL_NSEG>), grid, dim3(aphro::FNW * 64), lds, 0, p);
  };
  for (int i = 0; i < 8; ++i) run(i);
  CK(hipDeviceSynchronize());
  const int iters = 40;
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) run(i);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double us = ms * 1e3 / iters;
#ifdef ABL_TRACE
  {
    size_t nw = (size_t)grid.x * grid.y * aphro::FNW;
    run(3); CK(hipDeviceSynchronize());
    std::vector<uint64_t> ht(nw * 16); CK(hipMemcpy(ht.data(), tb, nw * 16 * 8, hipMemcpyDeviceToHost));
    uint64_t t0 = ~0ull; for (size_t w = 0; w < nw; ++w) t0 = ht[w * 16] < t0 ? ht[w * 16] : t0;
    const int NP = ABL_NSEG + 3;
    { double sc = 0, sw = 0; for (size_t w = 0; w < nw; ++w) { sc += (double)ht[w * 16 + 14]; sw += (double)(ht[w * 16 + NP - 1] - ht[w * 16]); }
      printf("shader cycles per wave %.0f, wall ticks %.1f -> effective clock %.0f MHz\n", sc / nw, sw / nw, sc / sw * 100.0); }
    if (getenv("ABL_DUMP")) {
      FILE* f = fopen(getenv("ABL_DUMP"), "w");
      for (size_t w = 0; w < nw; ++w) {
        fprintf(f, "%zu %llu", w, (unsigned long long)ht[w * 16 + 15]);
        for (int k = 0; k < NP; ++k) fprintf(f, " %.2f", (ht[w * 16 + k] - t0) * 0.01);
        fprintf(f, "\n");
      }
      fclose(f);
    }
    printf("trace (us rel. to first wave start; 100 MHz clock): point  min  mean  max\n");
    for (int k = 0; k < NP; ++k) {
      double mn = 1e9, mx = 0, sm = 0;
      for (size_t w = 0; w < nw; ++w) { double v = (ht[w * 16 + k] - t0) * 0.01; mn = v < mn ? v : mn; mx = v > mx ? v : mx; sm += v; }
      printf("  p%-2d %7.2f %7.2f %7.2f\n", k, mn, sm / nw, mx);
    }
  }
#endif
  printf("%-40s K=%d N=%d M=%d grid=(%d,%d,%d): %7.2f us  %7.1f GB/s (weights)\n", ABL_NAME, K, N, M, grid.x, grid.y, grid.z, us, wbytes / us / 1e3);
  return 0;
}
