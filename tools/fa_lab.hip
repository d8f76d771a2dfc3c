// Lab harness for the prefill attention kernels: the library sources compiled standalone.
//   -DFA_LAB=<bits> (third-generation kernel only): 8 no softmax, 16 no QK^T, 32 no PV, 64 no K/V staging after the first tiles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fa_lab.hip -o tools/bin/fa_lab     (tools/run_fa_lab.sh)
// Without FA_LAB: times the third-generation (APHRO_FA_NO_V4=1) and the fourth-generation kernel on the same random
// input (T tokens, Hq 32 / Hkv 8 / hd 128, causal) and compares their outputs element by element.
#include "bin/csrc_lab/flash_attn.hip"
#include "bin/csrc_lab/flash_attn_v4.hip"
#include "bin/csrc_lab/runtime.hip"
#include <cmath>
#include <cstring>
#include <vector>
int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 8192, Hq = 32, Hkv = 8, D = 128;
  const int dt = argc > 2 ? atoi(argv[2]) : APHRO_F16;
  const size_t row = (size_t)(Hq + 2 * Hkv) * D;
  std::vector<uint16_t> h(T * row);
  uint32_t st = 12345;
  for (auto& x : h) {
    st = st * 1664525u + 1013904223u;
    const float f = (((st >> 8) & 0xffff) / 65536.f - 0.5f) * 4.f;
    if (dt == APHRO_F16) { _Float16 v = (_Float16)f; x = *(uint16_t*)&v; }
    else { uint32_t u; memcpy(&u, &f, 4); x = (uint16_t)(u >> 16); }
  }
  uint16_t *qkv, *out, *out3; int32_t* cu;
  const size_t on = (size_t)T * Hq * D;
  hipMalloc(&qkv, h.size() * 2); hipMalloc(&out, on * 2); hipMalloc(&out3, on * 2); hipMalloc(&cu, 8);
  hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int32_t hc[2] = {0, T}; hipMemcpy(cu, hc, 8, hipMemcpyHostToDevice);
  auto run = [&](uint16_t* o) { return aphro_flash_attn_varlen(o, qkv, qkv + Hq * D, qkv + (Hq + Hkv) * D, cu, 1, T, Hq, Hkv, D, row, row, row, 0.0883883f, 1, nullptr, dt, nullptr); };
  auto bench = [&](const char* name, uint16_t* o) {
    hipMemset(o, 0xff, on * 2);
    if (run(o) != 0) { printf("error: %s\n", aphro_last_error()); exit(1); }
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: kernel fault\n", name); exit(1); }
    for (int i = 0; i < 40; ++i) run(o);            // the first ~50 ms of MFMA-heavy work run below the clocks the chip then holds
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); for (int i = 0; i < 20; ++i) run(o); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
    printf("%s T=%d %s: %.3f ms  %.1f TFLOP/s (causal flops)\n", name, T, dt == APHRO_F16 ? "f16" : "bf16", ms,
           4.0 * T * T * D * Hq / 2 / (ms * 1e-3) / 1e12);
  };
#ifdef FA_LAB
  setenv("APHRO_FA_NO_V4", "1", 1);
  bench("third-generation (FA_LAB pieces removed)", out3);
#else
  setenv("APHRO_FA_NO_V4", "1", 1);
  bench("third generation ", out3);
  unsetenv("APHRO_FA_NO_V4");
  bench("fourth generation", out);
  std::vector<uint16_t> a(on), b(on);
  hipMemcpy(a.data(), out3, on * 2, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), out, on * 2, hipMemcpyDeviceToHost);
  auto tof = [&](uint16_t x) { if (dt == APHRO_F16) { _Float16 v; memcpy(&v, &x, 2); return (float)v; } uint32_t u = (uint32_t)x << 16; float f; memcpy(&f, &u, 4); return f; };
  double worst = 0; size_t bad = 0, nan = 0, wi = 0;
  for (size_t i = 0; i < on; ++i) {
    const float x = tof(a[i]), y = tof(b[i]);
    if (y != y) { ++nan; continue; }
    const double d = fabs((double)x - y);
    if (d > worst) { worst = d; wi = i; }
    if (d > 2e-2 * (dt == APHRO_F16 ? 1 : 4)) ++bad;
  }
  printf("fourth vs third generation: worst abs diff %.3e at token %zu head %zu d %zu (%.4f vs %.4f), %zu beyond tolerance, %zu NaN\n",
         worst, wi / (Hq * D), (wi / D) % Hq, wi % D, tof(a[wi]), tof(b[wi]), bad, nan);
#endif
  return 0;
}
