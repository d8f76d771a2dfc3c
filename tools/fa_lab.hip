// Lab harness for the third-generation prefill kernel: the library source compiled standalone with pieces removed
// (-DFA_LAB=<bits>: 8 no softmax, 16 no QK^T, 32 no PV, 64 no K/V staging after the first tiles; 0 = everything).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -DFA_LAB=8 tools/fa_lab.hip -o tools/bin/fa_lab_8
#include "../aphrodite_engine_amd/csrc/flash_attn.hip"
#include "../aphrodite_engine_amd/csrc/runtime.hip"
#include <vector>
int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 8192, Hq = 32, Hkv = 8, D = 128;
  const size_t row = (size_t)(Hq + 2 * Hkv) * D;
  std::vector<uint16_t> h(T * row);
  uint32_t st = 12345;
  for (auto& x : h) { st = st * 1664525u + 1013904223u; const float f = ((st >> 8) & 0xffff) / 65536.f - 0.5f; _Float16 v = (_Float16)f; x = *(uint16_t*)&v; }
  uint16_t *qkv, *out; int32_t* cu;
  hipMalloc(&qkv, h.size() * 2); hipMalloc(&out, (size_t)T * Hq * D * 2); hipMalloc(&cu, 8);
  hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int32_t hc[2] = {0, T}; hipMemcpy(cu, hc, 8, hipMemcpyHostToDevice);
  auto run = [&]() { return aphro_flash_attn_varlen(out, qkv, qkv + Hq * D, qkv + (Hq + Hkv) * D, cu, 1, T, Hq, Hkv, D, row, row, row, 0.0883883f, 1, nullptr, APHRO_F16, nullptr); };
  if (run() != 0) { printf("error: %s\n", aphro_last_error()); return 1; }
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a); for (int i = 0; i < 5; ++i) run(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  printf("FA_LAB=%d T=%d: %.3f ms  %.1f TFLOP/s (causal flops)\n",
#ifdef FA_LAB
         FA_LAB,
#else
         -1,
#endif
         T, ms, 4.0 * T * T * D * Hq / 2 / (ms * 1e-3) / 1e12);
  return 0;
}
