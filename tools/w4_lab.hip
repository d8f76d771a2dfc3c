// Torch-free lab for the prefill-sized W4A16 GEMM (csrc/wna16_gemm_large.hip): host fp64 reference on sampled rows / columns,
// bitwise comparison of the eight-phase kernel with the two / three-stage kernel, run-to-run race screen, interleaved timing.
// Build: tools/run_w4_lab.sh ; run on the GPU box: tools/bin/w4_lab
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <vector>

#include "../include/aphrodite_mi355x.h"

extern "C" size_t aphro_wna16_gemm_large_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t groups, int dtype);
extern "C" int aphro_wna16_gemm_large(const void* a, const uint32_t* q_weight, const uint32_t* qzeros, const void* scales, void* c,
                                      void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K, int64_t groups,
                                      int64_t lda, int zero_offset, int dtype, void* stream);
extern "C" const char* aphro_last_error(void);
extern "C" void aphro_reload_env(void);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float h2f(uint16_t b) { _Float16 h; memcpy(&h, &b, 2); return (float)h; }
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t b; memcpy(&b, &h, 2); return b; }

struct Shape { int M, N, K; };

int main(int argc, char** argv) {
  std::vector<Shape> shapes = {{8192, 4096, 4096}, {8192, 28672, 4096}, {8192, 4096, 14336}};
  const int GS = 128;
  std::mt19937 rng(4321);
  std::normal_distribution<float> nd(0.f, 1.f);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (const Shape& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K, G = K / GS;
    std::vector<uint16_t> ha((size_t)M * K), hs((size_t)G * N);
    std::vector<uint32_t> hq((size_t)(K / 8) * N), hz((size_t)G * (N / 8));
    for (auto& x : ha) x = f2h(nd(rng));
    for (auto& x : hq) x = rng();
    for (auto& x : hz) x = rng();
    for (auto& x : hs) x = f2h(0.005f + (rng() % 1000) * 1e-5f);
    uint16_t *da, *ds, *dc, *dc2; uint32_t *dq, *dz; void* ws;
    CK(hipMalloc(&da, ha.size() * 2)); CK(hipMalloc(&ds, hs.size() * 2)); CK(hipMalloc(&dq, hq.size() * 4)); CK(hipMalloc(&dz, hz.size() * 4));
    CK(hipMalloc(&dc, (size_t)M * N * 2)); CK(hipMalloc(&dc2, (size_t)M * N * 2));
    const size_t wsb = aphro_wna16_gemm_large_workspace_bytes(M, N, K, G, APHRO_F16);
    CK(hipMalloc(&ws, wsb ? wsb : 16));
    CK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dz, hz.data(), hz.size() * 4, hipMemcpyHostToDevice));
    auto run = [&](uint16_t* out, int eight) {
      setenv("APHRO_WNA16_LARGE_8PHASE", eight ? "1" : "0", 1);
      aphro_reload_env();
      const int rc = aphro_wna16_gemm_large(da, dq, dz, ds, out, ws, wsb, M, N, K, G, K, 1, APHRO_F16, st);
      if (rc != 0) { printf("launch rc=%d: %s\n", rc, aphro_last_error()); exit(1); }
    };
    std::vector<uint16_t> hc((size_t)M * N), hc2((size_t)M * N);
    CK(hipMemsetAsync(dc, 0xff, (size_t)M * N * 2, st));
    run(dc, 0);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(hc.data(), dc, hc.size() * 2, hipMemcpyDeviceToHost));
    {   // host reference: w[k][n] = f16((q - (z + 1)) * s) (one rounding), fp64 dot
      auto wval = [&](int k, int n) {
        const uint32_t d = hq[(size_t)(k / 8) * N + n];
        const int e = k & 7;
        const int q = (d >> ((e & 1) * 16 + (e >> 1) * 4)) & 15;
        const int z = ((hz[(size_t)(k / GS) * (N / 8) + n / 8] >> ((n & 7) * 4)) & 15) + 1;
        return h2f(f2h((float)(q - z) * h2f(hs[(size_t)(k / GS) * N + n])));
      };
      double worst = 0; long bad = 0;
      auto check = [&](int m, int n) {
        double acc = 0;
        for (int k = 0; k < K; ++k) acc += (double)h2f(ha[(size_t)m * K + k]) * wval(k, n);
        const double got = h2f(hc[(size_t)m * N + n]);
        const double err = fabs(got - acc), tol = 2e-3 * fabs(acc) + 2e-3 * sqrt((double)K) * 0.05;
        if (!(err <= tol)) { if (bad < 5) printf("  MISMATCH m=%d n=%d got %g ref %g\n", m, n, got, acc); ++bad; }
        worst = std::max(worst, err);
      };
      const int rows[] = {0, 1, 31, 32, 63, 64, 127, 128, 255, 256, 257, M / 2 + 3, M - 257, M - 256, M - 2, M - 1};
      const int cols[] = {0, 3, 4, 31, 32, 33, 63, 64, 65, 255, 256, N / 2 + 5, N - 257, N - 33, N - 2, N - 1};
      const int nrow = K > 2048 ? 3 : 8;
      for (int i = 0; i < nrow; ++i) { const int m = rows[(i * 5) % 16]; for (int n = 0; n < N; ++n) check(m, n); }
      for (int i = 0; i < nrow; ++i) { const int n = cols[(i * 5) % 16]; for (int m = 0; m < M; ++m) check(m, n); }
      printf("M=%d N=%d K=%d: staged kernel vs host fp64: %ld bad, worst abs %.3e\n", M, N, K, bad, worst);
    }
    long diff = 0;
    for (int r = 0; r < 4; ++r) {
      CK(hipMemsetAsync(dc2, 0xff, (size_t)M * N * 2, st));
      run(dc2, 1);
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(hc2.data(), dc2, hc2.size() * 2, hipMemcpyDeviceToHost));
      long d = 0;
      for (size_t i = 0; i < hc.size(); ++i) d += hc[i] != hc2[i];
      if (d && r == 0) { for (size_t i = 0, shown = 0; i < hc.size() && shown < 5; ++i) if (hc[i] != hc2[i]) { printf("   first diffs: m=%zu n=%zu old %g new %g\n", i / N, i % N, h2f(hc[i]), h2f(hc2[i])); ++shown; } }
      diff += d;
    }
    printf("   eight-phase: %ld elements differ from the staged kernel over 4 runs\n", diff);
    double best[2] = {1e9, 1e9}, sum[2] = {0, 0};
    const int rounds = 4, iters = K * (double)N > 1e8 ? 5 : 20;
    for (int r = 0; r < rounds; ++r)
      for (int v = 0; v < 2; ++v) {
        run(dc, v);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) run(dc, v);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        best[v] = std::min(best[v], us); sum[v] += us;
      }
    for (int v = 0; v < 2; ++v)
      printf("   %-12s best %8.1f us %7.1f TF   mean %8.1f us %7.1f TF\n", v ? "eight-phase" : "staged", best[v], 2.0 * M * N * K / best[v] * 1e-6, sum[v] / rounds,
             2.0 * M * N * K / (sum[v] / rounds) * 1e-6);
    if (wsb >= 4096) {
      run(dc, 1);
      CK(hipStreamSynchronize(st));
      std::vector<unsigned long long> stm(256);
      CK(hipMemcpy(stm.data(), (char*)ws + 2048, 2048, hipMemcpyDeviceToHost));
      for (int g = 0; g < 2; ++g)
        for (int sg = 0; sg < 3; ++sg) {
          const unsigned long long* t = &stm[((0 * 2 + g) * 4 + sg) * 8];
          if (!t[0]) continue;
          printf("   stamps wg 0 group %d seg %d: landed +%6lld  loop done +%7lld  stores issued +%7lld  acked +%7lld   (%.2f us, %.0f MHz)\n", g, sg,
                 (long long)(t[1] - t[0]), (long long)(t[2] - t[0]), (long long)(t[4] - t[0]), (long long)(t[5] - t[0]), (t[7] - t[6]) * 0.01,
                 t[7] > t[6] ? (double)(t[5] - t[0]) / ((t[7] - t[6]) * 0.01) : 0.0);
        }
    }
    fflush(stdout);
    CK(hipFree(da)); CK(hipFree(ds)); CK(hipFree(dq)); CK(hipFree(dz)); CK(hipFree(dc)); CK(hipFree(dc2)); CK(hipFree(ws));
  }
  return 0;
}
