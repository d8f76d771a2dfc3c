// Torch-free lab for the prefill-sized FP8 GEMM (csrc/fp8_gemm_large.hip): correctness against a host fp64 reference on
// sampled rows / columns, bitwise comparison of the eight-phase kernel with the two-stage kernel, run-to-run race screen,
// and interleaved A/B timing.  Build + run: tools/run_gemm8_lab.sh (hipcc tools/gemm8_lab.hip csrc/fp8_gemm_large.hip
// csrc/runtime.hip).   gemm8_lab [quick]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <vector>

#include "../include/aphrodite_mi355x.h"

extern "C" size_t aphro_scaled_mm_fp8_large_workspace_bytes(int64_t M, int64_t N, int64_t K);
extern "C" int aphro_scaled_mm_fp8_large(void* out, const void* a, const void* b, const float* a_scales, const float* b_scales,
                                         const void* bias, void* workspace, size_t workspace_bytes, int64_t M, int64_t N,
                                         int64_t K, int a_scale_per_token, int b_scale_per_channel, int out_dtype, void* stream);
extern "C" const char* aphro_last_error(void);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float e4m3_to_f32(uint8_t b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 0) v = ldexpf((float)m, -9);
  else v = ldexpf(1.f + m / 8.f, e - 7);
  return s ? -v : v;
}
static float bf16_to_f32(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

struct Shape { int M, N, K; };

static void set_mode(int eight, int skew) {
  char buf[32];
  setenv("APHRO_FP8_LARGE_SKEW_MODE", skew < 0 ? "1" : "0", 1);
  if (skew < 0) skew = -skew;
  snprintf(buf, sizeof buf, "%d", eight); setenv("APHRO_FP8_LARGE_8PHASE", buf, 1);
  snprintf(buf, sizeof buf, "%d", skew); setenv("APHRO_FP8_LARGE_SKEW", buf, 1);
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  std::vector<Shape> shapes = {{2048, 4096, 512}, {8192, 4096, 4096}, {8192, 6144, 4096}, {8192, 28672, 4096}, {8192, 4096, 14336},
                               {4000, 8192, 1024}};
  if (quick) shapes = {{2048, 4096, 512}, {8192, 4096, 4096}};
  std::mt19937 rng(1234);
  float lut[256];
  for (int i = 0; i < 256; ++i) lut[i] = e4m3_to_f32((uint8_t)i);
  auto rand_fp8 = [&](std::vector<uint8_t>& v) {
    for (auto& x : v) {   // sign random, exponent 3..9 (2^-4 .. 2^2), mantissa random: full-range toggling, no NaN
      const uint32_t r = rng();
      x = (uint8_t)(((r & 1) << 7) | ((3 + (r >> 1) % 7) << 3) | ((r >> 8) & 7));
    }
  };
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  for (const Shape& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K;
    std::vector<uint8_t> ha((size_t)M * K), hw((size_t)N * K);
    rand_fp8(ha); rand_fp8(hw);
    std::vector<float> hsa(M), hsb(N);
    for (auto& x : hsa) x = 0.05f + (rng() % 1000) * 1e-4f;
    for (auto& x : hsb) x = 0.005f + (rng() % 1000) * 1e-5f;
    uint8_t *da, *dw; float *dsa, *dsb; uint16_t *dc, *dc2; void* ws;
    CK(hipMalloc(&da, ha.size())); CK(hipMalloc(&dw, hw.size()));
    CK(hipMalloc(&dsa, M * 4)); CK(hipMalloc(&dsb, N * 4));
    CK(hipMalloc(&dc, (size_t)M * N * 2)); CK(hipMalloc(&dc2, (size_t)M * N * 2));
    const size_t wsb = aphro_scaled_mm_fp8_large_workspace_bytes(M, N, K);
    CK(hipMalloc(&ws, wsb ? wsb : 16));
    CK(hipMemcpy(da, ha.data(), ha.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, hsa.data(), M * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsb, hsb.data(), N * 4, hipMemcpyHostToDevice));
    auto run = [&](uint16_t* out) {
      const int rc = aphro_scaled_mm_fp8_large(out, da, dw, dsa, dsb, nullptr, ws, wsb, M, N, K, 1, 1, APHRO_BF16, st);
      if (rc != 0) { printf("launch rc=%d: %s\n", rc, aphro_last_error()); exit(1); }
    };
    std::vector<uint16_t> hc((size_t)M * N), hc2((size_t)M * N);
    // ---- correctness: the two-stage kernel vs host fp64 on sampled rows / columns; every variant bitwise vs the two-stage
    // kernel (same K order inside a tile, same stream-K cuts at the same skew) + a run-to-run race screen -----------------
    std::vector<int> variants = {1, 2, 3};
    std::vector<int> ablations = {5, 7};
    if (getenv("LAB_VARIANTS")) { variants.clear(); for (char* t = strtok(strdup(getenv("LAB_VARIANTS")), ","); t; t = strtok(nullptr, ",")) variants.push_back(atoi(t)); }
    {
      set_mode(0, 0);
      CK(hipMemsetAsync(dc, 0xff, (size_t)M * N * 2, st));
      run(dc);
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(hc.data(), dc, hc.size() * 2, hipMemcpyDeviceToHost));
      double worst = 0;
      long bad = 0;
      auto check = [&](int m, int n) {
        double acc = 0;
        const uint8_t* ar = &ha[(size_t)m * K]; const uint8_t* wr = &hw[(size_t)n * K];
        for (int k = 0; k < K; ++k) acc += (double)lut[ar[k]] * lut[wr[k]];
        const double ref = (double)hsa[m] * ((double)hsb[n] * acc);
        const double got = bf16_to_f32(hc[(size_t)m * N + n]);
        const double err = fabs(got - ref), tol = 8e-3 * fabs(ref) + 1e-3 * sqrt((double)K) * hsa[m] * hsb[n];
        if (!(err <= tol)) { if (bad < 5) printf("  MISMATCH m=%d n=%d got %g ref %g\n", m, n, got, ref); ++bad; }
        worst = std::max(worst, err / (fabs(ref) + 1e-6));
      };
      const int rows[] = {0, 1, 31, 32, 63, 64, 127, 128, 255, 256, 257, M / 2 + 3, M - 257, M - 256, M - 2, M - 1};
      const int cols[] = {0, 3, 4, 31, 32, 33, 63, 64, 65, 255, 256, N / 2 + 5, N - 257, N - 33, N - 2, N - 1};
      const int nrow = K > 2048 ? 6 : 16;
      for (int i = 0; i < nrow; ++i) { const int m = rows[(i * 5) % 16]; for (int n = 0; n < N; ++n) check(m, n); }
      for (int i = 0; i < nrow; ++i) { const int n = cols[(i * 5) % 16]; for (int m = 0; m < M; ++m) check(m, n); }
      printf("M=%d N=%d K=%d: two-stage vs host fp64: %ld bad, worst rel %.2e\n", M, N, K, bad, worst);
    }
    const int skews[3] = {0, K / 128 / 8, -(K / 128 / 4)};
    for (int si = 0; si < 3; ++si) {
      if (si >= 1) { set_mode(0, skews[si]); run(dc); CK(hipStreamSynchronize(st)); CK(hipMemcpy(hc.data(), dc, hc.size() * 2, hipMemcpyDeviceToHost)); }
      for (int v : variants) {
        long diff = 0;
        for (int r = 0; r < 3; ++r) {
          set_mode(v, skews[si]);
          CK(hipMemsetAsync(dc2, 0xff, (size_t)M * N * 2, st));
          run(dc2);
          CK(hipStreamSynchronize(st));
          CK(hipMemcpy(hc2.data(), dc2, hc2.size() * 2, hipMemcpyDeviceToHost));
          for (size_t i = 0; i < hc.size(); ++i) diff += hc[i] != hc2[i];
        }
        printf("   variant %2d skew %2d: %ld elements differ from the two-stage kernel over 3 runs\n", v, skews[si], diff);
        fflush(stdout);
      }
    }
    // ---- timing: interleaved rounds -----------------------------------------------------------------------------------
    struct Var { int eight, skew; double best, sum; };
    std::vector<Var> vars = {{0, 0, 1e9, 0}};
    for (int v : variants) vars.push_back({v, 0, 1e9, 0});
    for (int v : ablations) vars.push_back({v, 0, 1e9, 0});
    vars.push_back({2, K / 128 / 8, 1e9, 0});
    vars.push_back({2, -(K / 128 / 4), 1e9, 0});
    vars.push_back({2, -(K / 128 / 8), 1e9, 0});
    const int rounds = 4, iters = K * (double)N > 1e8 ? 5 : 20;
    for (int r = 0; r < rounds; ++r)
      for (auto& v : vars) {
        set_mode(v.eight, v.skew);
        run(dc);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) run(dc);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        v.best = std::min(v.best, us); v.sum += us;
      }
    for (auto& v : vars)
      printf("   variant %2d skew %3d  best %8.1f us %7.1f TF   mean %8.1f us %7.1f TF\n", v.eight, v.skew, v.best, 2.0 * M * N * K / v.best * 1e-6,
             v.sum / rounds, 2.0 * M * N * K / (v.sum / rounds) * 1e-6);
    fflush(stdout);
    // ---- per-wave stamps (debug == 4) of the shipped variant -----------------------------------------------------------
    if (wsb >= 4096) {
      setenv("APHRO_FP8_LARGE_DEBUG", "4", 1);
      set_mode(2, 0);
      run(dc);
      CK(hipStreamSynchronize(st));
      unsetenv("APHRO_FP8_LARGE_DEBUG");
      std::vector<unsigned long long> stm(256);
      CK(hipMemcpy(stm.data(), (char*)ws + 2048, 2048, hipMemcpyDeviceToHost));
      for (int w = 0; w < 2; ++w)
        for (int g = 0; g < 2; ++g)
          for (int sg = 0; sg < 3; ++sg) {
            const unsigned long long* t = &stm[((w * 2 + g) * 4 + sg) * 8];
            if (!t[0]) continue;
            printf("   stamps wg %d group %d seg %d: landed +%6lld  loop done +%7lld  epi_put +%7lld  stores issued +%7lld  acked +%7lld   (%.2f us, %.0f MHz)\n", w, g, sg,
                   (long long)(t[1] - t[0]), (long long)(t[2] - t[0]), t[3] ? (long long)(t[3] - t[0]) : -1, t[4] ? (long long)(t[4] - t[0]) : -1,
                   t[5] ? (long long)(t[5] - t[0]) : -1, (t[7] - t[6]) * 0.01, t[5] > t[0] && t[7] > t[6] ? (double)(t[5] - t[0]) / ((t[7] - t[6]) * 0.01) : 0.0);
          }
    }
    CK(hipFree(da)); CK(hipFree(dw)); CK(hipFree(dsa)); CK(hipFree(dsb)); CK(hipFree(dc)); CK(hipFree(dc2)); CK(hipFree(ws));
  }
  return 0;
}
