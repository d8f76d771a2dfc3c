// Round-4 lab harness (no torch, no Python: a fresh GPU box pays ~1-2 min for its first `import torch`): times the W4A16
// decode GEMM entry points of one or more builds of the library on the configs[1] projection shapes, weights cold
// (cycling over > 640 MB), HIP-graph replay, HIP-event timing -- the method of tools/prefetch_lab.py:timeit -- and
// compares every variant's output with the first library's (bitwise + max abs / rel difference).
//
//   reslab [--m 32] [--shapes gate_up,down,qkv,o] [--iters 10] LIB[@ENV=VAL[;ENV=VAL]...] ...
//
// The first LIB is the baseline.  ENV assignments are applied (setenv) before that variant's calls and removed after,
// e.g.  lib_a.so@APHRO_WNA16_RES_CFG=8,4,1,3  -- the resident kernel's hand-picked configuration.
// One JSON line per (variant, shape).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));      \
      exit(2);                                                                                 \
    }                                                                                          \
  } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ void fill_u32(uint32_t* p, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = hash32((uint32_t)i * 2654435761u + seed);
}
// f16 values: uniform in [lo, hi)
__global__ void fill_f16(_Float16* p, size_t n, uint32_t seed, float lo, float hi) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (_Float16)(lo + (hi - lo) * (hash32((uint32_t)i * 2654435761u + seed) >> 8) * (1.0f / 16777216.0f));
}

typedef int (*resident_fn)(const void*, const uint32_t*, const uint32_t*, const void*, void*, float*, size_t, void*, int64_t,
                           int64_t, int64_t, int64_t, int, int, int, void*);
typedef int (*relayout_fn)(const uint32_t*, uint32_t*, int64_t, int64_t, int64_t, int64_t, void*);
typedef int (*ksplit_fn)(int64_t, int64_t, int64_t, int64_t);
typedef int (*pack_a_fn)(const void*, const int32_t*, void*, int64_t, int64_t, int64_t, int, void*);
typedef size_t (*packed_bytes_fn)(int64_t, int64_t);
typedef const char* (*err_fn)(void);
typedef int (*packed_fn)(const void*, const uint32_t*, const uint32_t*, const void*, void*, float*, size_t, int64_t, int64_t,
                         int64_t, int64_t, int, int, void*);

struct Lib {
  std::string path, tag;
  std::vector<std::pair<std::string, std::string>> env;
  void* h = nullptr;
  resident_fn resident = nullptr;
  relayout_fn relayout = nullptr;
  ksplit_fn ksplit = nullptr;
  pack_a_fn pack_a = nullptr;
  packed_bytes_fn packed_bytes = nullptr;
  err_fn err = nullptr;
  packed_fn packed = nullptr;    // aphro_wna16_gemm_packed (the round-2 kernel): RESLAB_API=packed in the variant's ENV list
  ksplit_fn packed_ksplit = nullptr;
  bool use_packed = false;
  bool trace = false;            // RESLAB_TRACE=1: per-wave timeline of the last launch of a graph (TRACE instantiations)
  void (*set_trace)(void*) = nullptr;
};

struct Shape { const char* name; int K, N; bool silu; };
static const Shape SHAPES[] = {{"gate_up", 4096, 28672, true}, {"down", 14336, 4096, false}, {"qkv", 4096, 6144, false},
                               {"o", 4096, 4096, false}, {"gate_up70tp8", 8192, 7168, true}, {"down70tp8", 3584, 8192, false},
                               {"qkv70tp8", 8192, 1280, false}, {"o70tp8", 1024, 8192, false}};

int main(int argc, char** argv) {
  int M = 32, iters = 10;
  std::string shapes = "gate_up,down,qkv,o";
  std::vector<Lib> libs;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--m")) M = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--iters")) iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--shapes")) shapes = argv[++i];
    else {
      Lib l;
      std::string a = argv[i];
      l.tag = a;
      size_t at = a.find('@');
      l.path = a.substr(0, at);
      if (at != std::string::npos) {
        std::string rest = a.substr(at + 1);
        size_t pos = 0;
        while (pos < rest.size()) {
          size_t sc = rest.find(';', pos);
          if (sc == std::string::npos) sc = rest.size();
          std::string kv = rest.substr(pos, sc - pos);
          size_t eq = kv.find('=');
          if (eq != std::string::npos) l.env.push_back({kv.substr(0, eq), kv.substr(eq + 1)});
          pos = sc + 1;
        }
      }
      libs.push_back(l);
    }
  }
  if (libs.empty()) { fprintf(stderr, "usage: reslab [--m M] [--shapes a,b] LIB[@ENV=VAL;ENV=VAL] ...\n"); return 1; }
  for (auto& l : libs) {
    l.h = dlopen(l.path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!l.h) { fprintf(stderr, "dlopen %s: %s\n", l.path.c_str(), dlerror()); return 1; }
    l.resident = (resident_fn)dlsym(l.h, "aphro_wna16_gemm_resident");
    l.relayout = (relayout_fn)dlsym(l.h, "aphro_wna16_strip_relayout");
    l.ksplit = (ksplit_fn)dlsym(l.h, "aphro_wna16_resident_ksplit");
    l.pack_a = (pack_a_fn)dlsym(l.h, "aphro_wna16_pack_a");
    l.packed_bytes = (packed_bytes_fn)dlsym(l.h, "aphro_wna16_packed_a_bytes");
    l.err = (err_fn)dlsym(l.h, "aphro_last_error");
    l.packed = (packed_fn)dlsym(l.h, "aphro_wna16_gemm_packed");
    l.packed_ksplit = (ksplit_fn)dlsym(l.h, "aphro_wna16_ksplit");
    for (auto& kv : l.env) if (kv.first == "RESLAB_API" && kv.second == "packed") l.use_packed = true;
    for (auto& kv : l.env) if (kv.first == "RESLAB_TRACE") l.trace = true;
    l.set_trace = (void (*)(void*))dlsym(l.h, "aphro_wna16_resident_set_trace");
    if (l.use_packed && (!l.packed || !l.packed_ksplit)) { fprintf(stderr, "%s: no aphro_wna16_gemm_packed\n", l.path.c_str()); return 1; }
    if (!l.err || (!l.use_packed && (!l.resident || !l.relayout || !l.ksplit))) { fprintf(stderr, "%s: missing symbols\n", l.path.c_str()); return 1; }
  }
  if (!libs[0].pack_a) { fprintf(stderr, "baseline library lacks aphro_wna16_pack_a\n"); return 1; }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  for (const Shape& sh : SHAPES) {
    if ((("," + shapes + ",").find(std::string(",") + sh.name + ",")) == std::string::npos) continue;
    const int K = sh.K, N = sh.N, G = K / 128;
    const size_t wwords = (size_t)(K / 8) * N, zwords = (size_t)G * (N / 8), swords = (size_t)G * N;
    const size_t wbytes = wwords * 4;
    const int ncopy = (int)std::max<size_t>(2, ((size_t)640 << 20) / wbytes);
    std::vector<uint32_t*> qw(ncopy), qws(ncopy), qz(ncopy);
    std::vector<_Float16*> sc(ncopy);
    for (int i = 0; i < ncopy; ++i) {
      CK(hipMalloc(&qw[i], wbytes));
      CK(hipMalloc(&qws[i], wbytes));
      CK(hipMalloc(&qz[i], zwords * 4));
      CK(hipMalloc(&sc[i], swords * 2));
      fill_u32<<<1024, 256, 0, st>>>(qw[i], wwords, 0x1000u + i);
      fill_u32<<<256, 256, 0, st>>>(qz[i], zwords, 0x2000u + i);
      fill_f16<<<256, 256, 0, st>>>(sc[i], swords, 0x3000u + i, 0.001f, 0.011f);
    }
    _Float16* a;
    CK(hipMalloc(&a, (size_t)M * K * 2));
    fill_f16<<<256, 256, 0, st>>>(a, (size_t)M * K, 0x4000u, -1.f, 1.f);
    const size_t pb = libs[0].packed_bytes(M, K);
    void* apk;
    CK(hipMalloc(&apk, pb));
    if (libs[0].pack_a(a, nullptr, apk, M, K, K, 0, st) != 0) { fprintf(stderr, "pack_a: %s\n", libs[0].err()); return 1; }
    // outputs: silu form -> act_packed [packed_a_bytes(M, N/2)]; slab form -> [ksplit][M][N] fp32
    const size_t out_bytes = sh.silu ? libs[0].packed_bytes(M, N / 2) : (size_t)16 * M * N * 4;
    void *out, *out_ref;
    CK(hipMalloc(&out, out_bytes));
    CK(hipMalloc(&out_ref, out_bytes));
    std::vector<unsigned char> h_ref(out_bytes), h_out(out_bytes);
    const double nb = (double)wbytes + swords * 2 + zwords * 4 + (double)M * K * 2;

    for (size_t li = 0; li < libs.size(); ++li) {
      Lib& l = libs[li];
      for (auto& kv : l.env) setenv(kv.first.c_str(), kv.second.c_str(), 1);
      const int ks = l.use_packed ? (sh.silu ? 0 : l.packed_ksplit(M, N, K, G)) : l.ksplit(M, N, K, G);
      if (ks <= 0 || (sh.silu && ks != 1)) {
        printf("{\"shape\": \"%s\", \"variant\": \"%s\", \"skipped\": \"ksplit %d\"}\n", sh.name, l.tag.c_str(), ks);
        for (auto& kv : l.env) unsetenv(kv.first.c_str());
        continue;
      }
      for (int i = 0; i < ncopy && !l.use_packed; ++i)
        if (l.relayout(qw[i], qws[i], M, N, K, G, st) != 0) { fprintf(stderr, "relayout: %s\n", l.err()); return 1; }
      int nuse = ncopy;              // RESLAB_NCOPY=n: cycle over the first n copies only (1: weights warm in the Infinity Cache)
      for (auto& kv : l.env) if (kv.first == "RESLAB_NCOPY") nuse = std::max(1, std::min(ncopy, atoi(kv.second.c_str())));
      auto call = [&](int i, void* o) -> int {
        if (l.use_packed) return l.packed(apk, qw[i], qz[i], sc[i], nullptr, (float*)o, out_bytes, M, N, K, G, 1, 0, st);
        if (sh.silu) return l.resident(apk, qws[i], qz[i], sc[i], nullptr, nullptr, 0, o, M, N, K, G, 1, 0, 1, st);
        return l.resident(apk, qws[i], qz[i], sc[i], nullptr, (float*)o, out_bytes, nullptr, M, N, K, G, 1, 0, 1, st);
      };
      CK(hipMemsetAsync(out, 0, out_bytes, st));
      if (call(0, out) != 0) { fprintf(stderr, "%s: %s\n", l.tag.c_str(), l.err()); return 1; }
      CK(hipStreamSynchronize(st));
      // the slab form's result = sum over the K slices (variants may slice differently)
      CK(hipMemcpy(h_out.data(), out, out_bytes, hipMemcpyDeviceToHost));
      std::vector<float> val;
      if (sh.silu) {
        const _Float16* p = (const _Float16*)h_out.data();
        val.resize(out_bytes / 2);
        for (size_t i = 0; i < val.size(); ++i) val[i] = (float)p[i];
      } else {
        const float* p = (const float*)h_out.data();
        val.assign((size_t)M * N, 0.f);
        for (int z = 0; z < ks; ++z)
          for (size_t i = 0; i < (size_t)M * N; ++i) val[i] += p[(size_t)z * M * N + i];
      }
      static std::vector<float> ref;
      size_t nbad = 0;
      double maxabs = 0, maxref = 0;
      if (li == 0) ref = val;
      else {
        for (size_t i = 0; i < val.size(); ++i) {
          const double d = std::fabs((double)val[i] - (double)ref[i]);
          if (!(d == 0)) ++nbad;     // (NaN counts)
          if (d > maxabs || d != d) maxabs = d;
          if (std::fabs(ref[i]) > maxref) maxref = std::fabs(ref[i]);
        }
      }
      // timing: L launches per graph over the cold copies
      const int L = std::max(24, ncopy);
      hipGraph_t g;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int i = 0; i < L; ++i) call(i % nuse, out);
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int i = 0; i < 2; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      float best = 1e30f, tot = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const float us = ms * 1e3f / (iters * L);
        best = std::min(best, us);
        tot += us;
      }
      CK(hipGraphExecDestroy(ge));
      CK(hipGraphDestroy(g));
      if (l.trace && l.set_trace) {
        // a graph of L launches over the cold copies; only the last one stamps.  [workgroup][wave][20] u64
        const int NWG = 1024, NW = 8;
        unsigned long long* tr;
        CK(hipMalloc(&tr, (size_t)NWG * NW * 20 * 8));
        CK(hipMemset(tr, 0, (size_t)NWG * NW * 20 * 8));
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < L; ++i) { l.set_trace(i == L - 1 ? tr : nullptr); call(i % nuse, out); }
        l.set_trace(nullptr);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> h((size_t)NWG * NW * 20);
        CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<std::vector<double>> col(16);
        double wmin = 1e30, wmax = 0, emax = 0;
        int nw = 0;
        for (size_t w = 0; w < (size_t)NWG * NW; ++w) {
          const unsigned long long* t = &h[w * 20];
          if (t[17] == 0) continue;
          ++nw;
          for (int i = 0; i < 16; ++i) col[i].push_back(t[i] ? (double)(t[i] - t[0]) : -1.0);
          wmin = std::min(wmin, (double)t[16]); emax = std::max(emax, (double)t[16]); wmax = std::max(wmax, (double)t[17]);
        }
        printf("{\"shape\": \"%s\", \"variant\": \"%s\", \"trace_waves\": %d, \"wall_span_us\": %.2f, \"entry_skew_us\": %.2f, \"median_cycles\": [",
               sh.name, l.tag.c_str(), nw, (wmax - wmin) / 100.0, (emax - wmin) / 100.0);
        for (int i = 0; i < 16; ++i) {
          std::sort(col[i].begin(), col[i].end());
          printf("%s%.0f", i ? ", " : "", col[i].empty() ? -1.0 : col[i][col[i].size() / 2]);
        }
        printf("], \"p90_cycles\": [");
        for (int i = 0; i < 16; ++i) printf("%s%.0f", i ? ", " : "", col[i].empty() ? -1.0 : col[i][col[i].size() * 9 / 10]);
        printf("]}\n");
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
        CK(hipFree(tr));
      }
      printf("{\"shape\": \"%s\", \"M\": %d, \"variant\": \"%s\", \"ksplit\": %d, \"us\": %.2f, \"us_mean\": %.2f, \"TBps\": %.3f, "
             "\"mismatch\": %zu, \"max_abs_diff\": %.3g, \"max_ref\": %.3g}\n",
             sh.name, M, l.tag.c_str(), ks, best, tot / 3, nb / best / 1e6, nbad, maxabs, maxref);
      fflush(stdout);
      for (auto& kv : l.env) unsetenv(kv.first.c_str());
    }
    for (int i = 0; i < ncopy; ++i) { CK(hipFree(qw[i])); CK(hipFree(qws[i])); CK(hipFree(qz[i])); CK(hipFree(sc[i])); }
    CK(hipFree(a)); CK(hipFree(apk)); CK(hipFree(out)); CK(hipFree(out_ref));
  }
  return 0;
}
