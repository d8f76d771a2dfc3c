// Round-2 experiment bench for the int4 decode GEMM (not part of the product build).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.hip -o /tmp/gemm_lab && /tmp/gemm_lab [shape ...]
// One process times every variant interleaved (same inputs, same clocks) and checks each against the
// shipped kernel's output.  Questions it answers (DESIGN.md 8.1 / VERDICT r1 item 2):
//   * where does a wave's time go (per-segment timeline, s_memrealtime + HW_ID, MODE_TRACE)
//   * is the A-fragment cost an L2 CHANNEL HOT SPOT?  All workgroups sweep the same 256 KiB packed-A
//     image from segment 0 upwards in lock step, and the four waves of a workgroup start 64 KiB apart
//     (= the same channel if channels interleave at 4 KiB): MODE_ROT starts every workgroup at a
//     different segment, MODE_ILV gives the four waves adjacent segments, MODE_AREP reads one of 8
//     staggered copies of A.  probe_a isolates the A stream, probe_l2 measures the interleave itself.
//   * how far do the small shapes (qkv / o) move with every load issued up front and 8-16 waves.
#include "bin/csrc_lab/wna16_gemm.hip"
#include <vector>
#include <string>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
namespace aphro { void set_error(const char*, ...) {} }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

enum { MODE_ROT = 1, MODE_ILV = 2, MODE_TRACE = 4, MODE_AREP = 8, MODE_NOA = 16, MODE_NORS = 32, MODE_NOMFMA = 64, MODE_NOW = 128, MODE_RSPRE = 256, MODE_PRESC = 512, MODE_ZEND = 1024 };

namespace aphro {

struct LabParams {
  Wna16Params p;
  unsigned long long* trace;   // [waves][16]
  int arep_stride;             // bytes between A replicas (MODE_AREP)
  const float* rs;             // MODE_RSPRE: precomputed row sums [K/128][mtiles*16]
};

template <int NWV, int NSEG, int DEPTH, int MODE>
__global__ __launch_bounds__(NWV * 64, (NWV >= 16) ? 4 : 2) void lab_gemm_kernel(LabParams lp) {
  const Wna16Params& p = lp.p;
  using T = Half;
  constexpr int VEC = 4, MT = 2;
  constexpr int NBUF = DEPTH + 1;
  constexpr int NA = 2;
  constexpr int AUX_NT = 2;
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4;
  const int c = lane & 15;
  const int n0 = blockIdx.x * 64;
  const int m0 = blockIdx.z * 32;
  const int ncol = n0 + VEC * c;
  const int rot = (MODE & MODE_ROT) ? (int)((blockIdx.x * 5u + blockIdx.y) % NSEG) : 0;
  // segment visited at step s
  auto seg_of = [&](int s) -> int {
    int sr = s + rot;
    if (sr >= NSEG) sr -= NSEG;
    if constexpr (MODE & MODE_ILV) return blockIdx.y * (NWV * NSEG) + sr * NWV + wave;
    else return (blockIdx.y * NWV + wave) * NSEG + sr;
  };
  unsigned long long ts[NSEG + 3];
  if constexpr (MODE & MODE_TRACE) ts[0] = __builtin_amdgcn_s_memrealtime();

  const int mtiles = (p.M + 15) >> 4;
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.qw, (uint32_t)((size_t)(p.K >> 3) * p.N * 4));
  const uint16_t* abase = p.apk;
  if constexpr (MODE & MODE_AREP) abase = (const uint16_t*)((const char*)abase + (size_t)(blockIdx.x & 7) * lp.arep_stride);
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(abase, (uint32_t)((size_t)(p.K >> 7) * 4 * mtiles * 1024));
  const int ngroups = p.K / p.group_size;
  const __amdgpu_buffer_rsrc_t rs_ = make_rsrc(p.sc, (uint32_t)((size_t)ngroups * p.N * 2));
  const __amdgpu_buffer_rsrc_t rz = make_rsrc(p.qz, (uint32_t)((size_t)ngroups * (p.N >> 3) * 4));
  const int roww = p.N * 4;
  const int voff_w = (4 * g * p.N + ncol) * 4;
  int voff_a[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) voff_a[i] = (min((m0 >> 4) + i, mtiles - 1) * 64 + lane) * 16;
  const int abytes = mtiles * 1024;
  const int voff_s = ncol * 2;
  const int voff_z = (ncol >> 3) * 4;
  const int zshift = (ncol & 7) * 4;
  const float zoff = (float)p.zero_offset;
  const f16x8 ones = {(f16)1.f, (f16)1.f, (f16)16.f, (f16)16.f, (f16)1.f, (f16)1.f, (f16)16.f, (f16)16.f};
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  f32x4 cacc[MT][VEC];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < VEC; ++t) cacc[i][t] = zero4;

  SegMeta<VEC> meta[2];
  uint32_t w[NBUF][4][VEC];
  u32x4 af[NA][4][MT];
  f32x4 rsp[2][MT];
  const __amdgpu_buffer_rsrc_t rr = make_rsrc(lp.rs, (uint32_t)((size_t)(p.K >> 7) * mtiles * 64));
  auto load_rs = [&](f32x4 (&r)[MT], int s) {
    const int sg = seg_of(s);
#pragma unroll
    for (int i = 0; i < MT; ++i)
      r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, (min((m0 >> 4) + i, mtiles - 1) * 16 + 4 * g) * 4, sg * mtiles * 64, 0));
  };

  auto load_meta = [&](SegMeta<VEC>& m, int s) {
    const int grp = seg_of(s) >> p.gshift;
    m.zw = __builtin_amdgcn_raw_buffer_load_b32(rz, voff_z, grp * (p.N >> 3) * 4, 0);
    u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_, voff_s, grp * p.N * 2, 0);
    m.sc[0] = v[0]; m.sc[1] = v[1];
  };
  auto load_w = [&](uint32_t (&wd)[4][VEC], int s) {
    const int sg = seg_of(s);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int soff = (sg * 16 + u) * roww;
      u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, voff_w, soff, AUX_NT);
      wd[u][0] = v[0]; wd[u][1] = v[1]; wd[u][2] = v[2]; wd[u][3] = v[3];
    }
  };
  auto load_a = [&](u32x4 (&ad)[4][MT], int s) {
    const int sg = seg_of(s);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i)
        ad[u][i] = __builtin_amdgcn_raw_buffer_load_b128(ra, voff_a[i], (sg * 4 + u) * abytes, 0);
  };

  load_meta(meta[0], 0);
  if constexpr (MODE & MODE_RSPRE) load_rs(rsp[0], 0);
  if constexpr (!(MODE & MODE_NOA)) load_a(af[0], 0);
  if constexpr (!(MODE & MODE_NOW)) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_w(w[d], d);
  } else {
#pragma unroll
    for (int d = 0; d < NBUF; ++d)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < VEC; ++t) w[d][u][t] = 0x12345678u * (lane + u + t + d);
  }
  __builtin_amdgcn_sched_barrier(0);

#pragma unroll
  for (int s = 0; s < NSEG; ++s) {
    if constexpr (MODE & MODE_TRACE) ts[1 + s] = __builtin_amdgcn_s_memrealtime();
    if constexpr (!(MODE & MODE_NOA)) { if (s + 1 < NSEG) load_a(af[(s + 1) % NA], s + 1); }
    if constexpr (!(MODE & MODE_NOW)) { if (s + DEPTH < NSEG) load_w(w[(s + DEPTH) % NBUF], s + DEPTH); }
    if (s + 1 < NSEG) load_meta(meta[(s + 1) & 1], s + 1);
    if constexpr (MODE & MODE_RSPRE) { if (s + 1 < NSEG) load_rs(rsp[(s + 1) & 1], s + 1); }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[MT][VEC];
    f32x4 rs[MT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f16x8 a[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        u32x4 av = af[(MODE & MODE_NOA) ? 0 : (s % NA)][u][i];
        if constexpr (MODE & MODE_NOA) { asm volatile("" : "+v"(av)); }
        if constexpr (!(MODE & MODE_PRESC)) {
          asm("v_pk_mul_f16 %0, %1, %2" : "=v"(av[1]) : "v"(av[1]), "s"(0x2c002c00u));
          asm("v_pk_mul_f16 %0, %1, %2" : "=v"(av[3]) : "v"(av[3]), "s"(0x2c002c00u));
        }
        a[i] = __builtin_bit_cast(f16x8, av);
        if constexpr (MODE & MODE_RSPRE) { if (u == 0) rs[i] = rsp[s & 1][i]; }
        else if constexpr (!(MODE & (MODE_NORS | MODE_NOMFMA)))
          rs[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], ones, u == 0 ? zero4 : rs[i], 0, 0, 0);
        else rs[i] = zero4;
      }
#pragma unroll
      for (int t = 0; t < VEC; ++t) {
        uint32_t wv = w[s % NBUF][u][t];
        if constexpr (MODE & MODE_NOW) asm volatile("" : "+v"(wv));
        const uint32_t w8 = wv >> 8;
        u32x4 bq = {wv & 0x000f000fu, wv & 0x00f000f0u, w8 & 0x000f000fu, w8 & 0x00f000f0u};
        const f16x8 b = __builtin_bit_cast(f16x8, bq);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if constexpr (MODE & MODE_NOMFMA) {   // keep every operand live, no matrix op
            asm volatile("" :: "v"(a[i]), "v"(b));
            acc[i][t] = zero4;
          } else {
            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b, u == 0 ? zero4 : acc[i][t], 0, 0, 0);
          }
        }
      }
    }
    const SegMeta<VEC>& m = meta[s & 1];
#pragma unroll
    for (int t = 0; t < VEC; ++t) {
      const float z = (float)((m.zw >> (zshift + 4 * t)) & 0xf) + zoff;
      const float sf = T::to_f32((uint16_t)(m.sc[t >> 1] >> (16 * (t & 1))));
      const float s24 = sf * 16777216.f;
      const float nzs = -z * sf;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        cacc[i][t] = __builtin_elementwise_fma(acc[i][t], f32x4{s24, s24, s24, s24}, cacc[i][t]);
        if constexpr (!(MODE & MODE_ZEND)) cacc[i][t] = __builtin_elementwise_fma(rs[i], f32x4{nzs, nzs, nzs, nzs}, cacc[i][t]);
        else asm volatile("" :: "v"(nzs), "v"(rs[i]));
      }
    }
  }
  if constexpr (MODE & MODE_ZEND) {
    // zero-point term  -sum_g (s_g z_g)[n] rs_g[m]  as (hi, lo) f16 MFMAs over k' = group index
    f16x8 bz[VEC];
#pragma unroll
    for (int t = 0; t < VEC; ++t) { u32x4 q = {meta[0].sc[t >> 1], meta[1].sc[t >> 1], meta[0].zw, meta[1].zw}; bz[t] = __builtin_bit_cast(f16x8, q); }
#pragma unroll
    for (int rep = 0; rep < 3; ++rep)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        f16x8 ar = __builtin_bit_cast(f16x8, rsp[rep & 1][i]);
#pragma unroll
        for (int t = 0; t < VEC; ++t) cacc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ar, bz[t], cacc[i][t], 0, 0, 0);
      }
  }
  if constexpr (MODE & MODE_TRACE) ts[NSEG + 1] = __builtin_amdgcn_s_memrealtime();
  wna16_epilogue<T, VEC, MT, NWV>(p, red, cacc, lane, wave, g, m0, ncol);
  if constexpr (MODE & MODE_TRACE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[NSEG + 2] = __builtin_amdgcn_s_memrealtime();
    const size_t wg = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NWV + wave;
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < NSEG + 3; ++k) lp.trace[wg * 16 + k] = ts[k];
      lp.trace[wg * 16 + 14] = __builtin_amdgcn_s_getreg(63492);   // HW_ID
      lp.trace[wg * 16 + 15] = __builtin_amdgcn_s_getreg(63508);   // XCC_ID
    }
  }
}

// A stream alone: the shipped kernel's A-fragment address pattern (8 x 1 KiB per segment per wave), no weights.
template <int NWV, int NSEG, int MODE>
__global__ __launch_bounds__(NWV * 64) void probe_a_kernel(const uint16_t* apk, int K, int mtiles, uint32_t* out, int arep_stride) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rot = (MODE & MODE_ROT) ? (int)((blockIdx.x * 5u) % NSEG) : 0;
  const uint16_t* abase = apk;
  if constexpr (MODE & MODE_AREP) abase = (const uint16_t*)((const char*)abase + (size_t)(blockIdx.x & 7) * arep_stride);
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(abase, (uint32_t)((size_t)(K >> 7) * 4 * mtiles * 1024));
  const int abytes = mtiles * 1024;
  u32x4 x = {0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < NSEG; ++s) {
    int sr = s + rot; if (sr >= NSEG) sr -= NSEG;
    const int sg = (MODE & MODE_ILV) ? sr * NWV + wave : wave * NSEG + sr;
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i) v[u * 2 + i] = __builtin_amdgcn_raw_buffer_load_b128(ra, (i * 64 + lane) * 16, (sg * 4 + u) * abytes, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) x ^= v[j];
  }
  if ((x[0] ^ x[1] ^ x[2] ^ x[3]) == 0x12345679u) out[blockIdx.x] = 1;
}

// L2 interleave probe: every wave reads 1 KiB (sc1: not from L1) ITERS times from region (wave_global * 5 + it) % R of
// size 1 KiB at spacing `stride`.  R = 1: one hot KiB; larger R at spacing 256 B ... 64 KiB shows the channel interleave.
__global__ __launch_bounds__(256) void probe_l2_kernel(const uint8_t* buf, int R, int stride, int iters, uint32_t* out) {
  const int lane = threadIdx.x & 63;
  const int wg = blockIdx.x * 4 + (threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t r = make_rsrc(buf, 0x7fffffffu);
  u32x4 x = {0, 0, 0, 0};
  for (int it = 0; it < iters; it += 4) {
    u32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int reg = __builtin_amdgcn_readfirstlane((wg * 5 + it + j) % R);
      v[j] = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, reg * stride, 16 /* sc1 */);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) x ^= v[j];
  }
  if ((x[0] ^ x[1] ^ x[2] ^ x[3]) == 0x12345679u) out[blockIdx.x] = 1;
}
}  // namespace aphro

struct Ctx {
  int K, N, M, G, mtiles;
  std::vector<uint32_t*> qw;
  uint32_t* qz; uint16_t *sc, *apk, *c; float* part; unsigned long long* trace; uint32_t* out;
  int arep_stride;
  std::vector<float> ref;   // full-K sums of the reference kernel
};

template <typename F>
static double time_us(F&& launch, int iters = 30) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 6; ++i) launch(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch(i);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1e3 / iters;
}

static void fetch_sum(Ctx& cx, int ksplit, std::vector<float>& dst) {
  const size_t mn = (size_t)cx.M * cx.N;
  dst.assign(mn, 0.f);
  if (ksplit == 1) {
    std::vector<uint16_t> h(mn);
    CK(hipMemcpy(h.data(), cx.c, mn * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < mn; ++i) dst[i] = (float)__builtin_bit_cast(_Float16, h[i]);
  } else {
    std::vector<float> h(mn * ksplit);
    CK(hipMemcpy(h.data(), cx.part, mn * ksplit * 4, hipMemcpyDeviceToHost));
    for (int k = 0; k < ksplit; ++k) for (size_t i = 0; i < mn; ++i) dst[i] += h[(size_t)k * mn + i];
  }
}

static aphro::Wna16Params base_params(Ctx& cx, int ksplit, int nwv, int nseg) {
  aphro::Wna16Params p{};
  p.a = nullptr; p.apk = cx.apk; p.qz = cx.qz; p.sc = cx.sc; p.c = cx.c; p.partial = cx.part;
  p.M = cx.M; p.N = cx.N; p.K = cx.K; p.lda = cx.K; p.group_size = 128; p.ksteps_per_split = 4 * nwv * nseg;
  p.ksplit = ksplit; p.zero_offset = 1; p.gshift = 0; p.force_partial = 0;
  return p;
}

static std::vector<std::pair<std::string, double>> results;

template <int NWV, int NSEG, int DEPTH, int MODE>
static void run_variant(Ctx& cx, const char* name) {
  if (cx.K % (128 * NWV * NSEG) != 0) return;
  const int ksplit = cx.K / (128 * NWV * NSEG);
  if (ksplit > 8) return;
  aphro::LabParams lp{};
  lp.p = base_params(cx, ksplit, NWV, NSEG);
  lp.trace = cx.trace; lp.arep_stride = cx.arep_stride; lp.rs = (const float*)cx.part;
  dim3 grid(cx.N / 64, ksplit, (cx.M + 31) / 32);
  const size_t lds = (size_t)NWV * 2 * 4 * 64 * 4 * sizeof(float);
  auto kern = aphro::lab_gemm_kernel<NWV, NSEG, DEPTH, MODE>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  auto launch = [&](int i) { lp.p.qw = cx.qw[i % cx.qw.size()]; hipLaunchKernelGGL(kern, grid, dim3(NWV * 64), lds, 0, lp); };
  // correctness vs the shipped kernel (weights copy 0)
  CK(hipMemset(cx.c, 0, (size_t)cx.M * cx.N * 2)); CK(hipMemset(cx.part, 0, (size_t)8 * cx.M * cx.N * 4));
  launch(0); CK(hipDeviceSynchronize());
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { printf("  %-34s launch failed: %s\n", name, hipGetErrorString(le)); return; }
  const char* verdict = "";
  if (!(MODE & (MODE_NOA | MODE_NORS | MODE_NOMFMA | MODE_NOW | MODE_RSPRE | MODE_PRESC | MODE_ZEND))) {
    std::vector<float> got; fetch_sum(cx, ksplit, got);
    double maxd = 0, maxr = 0;
    for (size_t i = 0; i < got.size(); ++i) { maxd = fmax(maxd, fabs((double)cx.ref[i] - got[i])); maxr = fmax(maxr, fabs((double)cx.ref[i])); }
    verdict = maxd <= 2e-3 * maxr + 1e-6 ? "ok" : "MISMATCH";
  }
  const double us = time_us(launch);
  const double wbytes = (double)cx.K / 8 * cx.N * 4;
  printf("  %-34s wg=%4d x %2dw ksplit=%d nseg=%d depth=%d : %7.2f us  %6.0f GB/s  %s\n", name, grid.x * grid.y, NWV, ksplit, NSEG, DEPTH,
         us, wbytes / us / 1e3, verdict);
  fflush(stdout);
  if constexpr (MODE & MODE_TRACE) {
    const size_t nw = (size_t)grid.x * grid.y * NWV;
    CK(hipMemset(cx.trace, 0, nw * 16 * 8));
    launch(3); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> ht(nw * 16); CK(hipMemcpy(ht.data(), cx.trace, nw * 16 * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull; for (size_t w = 0; w < nw; ++w) t0 = std::min(t0, ht[w * 16]);
    const int NP = NSEG + 3;
    // waves per CU (HW_ID: cu_id bits 11:8, sh_id 12, se_id 15:13 ; XCC_ID bits 3:0)
    std::vector<int> cu_of(nw); std::vector<int> cnt(16 * 256, 0);
    for (size_t w = 0; w < nw; ++w) {
      const unsigned hw = (unsigned)ht[w * 16 + 14], xcc = (unsigned)ht[w * 16 + 15] & 15;
      const int cu = (int)(xcc * 256 + ((hw >> 8) & 0xff));
      cu_of[w] = cu; cnt[cu]++;
    }
    printf("    trace %s (us since first wave start, 100 MHz clock): point = start, seg0..seg%d top, loop end, kernel end\n", name, NSEG - 1);
    for (int cls = 0; cls < 2; ++cls) {   // waves on CUs hosting <= NWV waves vs more
      std::vector<double> mean(NP, 0.0), mx(NP, 0.0); size_t n = 0;
      for (size_t w = 0; w < nw; ++w) {
        const bool heavy = cnt[cu_of[w]] > NWV;
        if ((int)heavy != cls) continue;
        ++n;
        for (int k = 0; k < NP; ++k) { double v = (ht[w * 16 + k] - t0) * 0.01; mean[k] += v; mx[k] = std::max(mx[k], v); }
      }
      if (!n) continue;
      printf("    %s CUs (%zu waves): mean", cls ? "2-WG" : "1-WG", n);
      for (int k = 0; k < NP; ++k) printf(" %6.2f", mean[k] / n);
      printf("\n    %*s max ", 20, "");
      for (int k = 0; k < NP; ++k) printf(" %6.2f", mx[k]);
      printf("\n");
    }
    if (getenv("LAB_DUMP")) {
      std::string nm = name; for (auto& ch : nm) if (ch == '/' || ch == ' ' || ch == '|') ch = '_';
      std::string fn = std::string(getenv("LAB_DUMP")) + "_" + std::to_string(cx.K) + "_" + nm + ".txt";
      FILE* f = fopen(fn.c_str(), "w");
      if (f)
      for (size_t w = 0; f && w < nw; ++w) {
        fprintf(f, "%zu %d %d", w, cu_of[w], cnt[cu_of[w]]);
        for (int k = 0; k < NP; ++k) fprintf(f, " %.2f", (ht[w * 16 + k] - t0) * 0.01);
        fprintf(f, "\n");
      }
      if (f) fclose(f);
    }
  }
}

static void run_shipped(Ctx& cx, int nseg, int ksplit) {
  aphro::Wna16Params p = base_params(cx, ksplit, aphro::FNW, nseg);
  dim3 grid(cx.N / 64, ksplit, (cx.M + 31) / 32);
  const size_t lds = (size_t)aphro::FNW * 2 * 4 * 64 * 4 * sizeof(float);
  auto launch = [&](int i) {
    p.qw = cx.qw[i % cx.qw.size()];
    switch (nseg) {
      case 8: hipLaunchKernelGGL((aphro::wna16_gemm_kernel<aphro::Half, 4, 2, 8>), grid, dim3(256), lds, 0, p); break;
      case 7: hipLaunchKernelGGL((aphro::wna16_gemm_kernel<aphro::Half, 4, 2, 7>), grid, dim3(256), lds, 0, p); break;
      case 4: hipLaunchKernelGGL((aphro::wna16_gemm_kernel<aphro::Half, 4, 2, 4>), grid, dim3(256), lds, 0, p); break;
      case 2: hipLaunchKernelGGL((aphro::wna16_gemm_kernel<aphro::Half, 4, 2, 2>), grid, dim3(256), lds, 0, p); break;
      default: hipLaunchKernelGGL((aphro::wna16_gemm_kernel<aphro::Half, 4, 2, 1>), grid, dim3(256), lds, 0, p); break;
    }
  };
  CK(hipMemset(cx.c, 0, (size_t)cx.M * cx.N * 2)); CK(hipMemset(cx.part, 0, (size_t)8 * cx.M * cx.N * 4));
  launch(0); CK(hipDeviceSynchronize());
  fetch_sum(cx, ksplit, cx.ref);
  const double us = time_us(launch);
  const double wbytes = (double)cx.K / 8 * cx.N * 4;
  printf("  %-34s wg=%4d x %2dw ksplit=%d nseg=%d depth=2 : %7.2f us  %6.0f GB/s  (reference)\n", "SHIPPED", grid.x * grid.y, 4, ksplit, nseg,
         us, wbytes / us / 1e3);
  fflush(stdout);
}

static void shape(int K, int N, int M, int ship_nseg, int ship_ksplit) {
  Ctx cx; cx.K = K; cx.N = N; cx.M = M; cx.G = K / 128; cx.mtiles = (M + 15) / 16;
  const size_t wbytes = (size_t)K / 8 * N * 4;
  const int copies = (int)std::max<size_t>(2, (600u << 20) / wbytes + 1);
  std::vector<uint32_t> h(wbytes / 4);
  for (auto& x : h) x = (uint32_t)rand() * 2654435761u;
  cx.qw.resize(std::min(copies, 48));
  for (auto& q : cx.qw) { CK(hipMalloc(&q, wbytes)); CK(hipMemcpy(q, h.data(), wbytes, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&cx.qz, (size_t)cx.G * N / 2)); CK(hipMemset(cx.qz, 0x77, (size_t)cx.G * N / 2));
  CK(hipMalloc(&cx.sc, (size_t)cx.G * N * 2)); CK(hipMemset(cx.sc, 0x1c, (size_t)cx.G * N * 2));
  const size_t abytes = (size_t)cx.mtiles * 16 * K * 2;
  cx.arep_stride = (int)(abytes + 4096 + 256);    // replicas staggered by one 4 KiB page + 256 B
  CK(hipMalloc(&cx.apk, (size_t)cx.arep_stride * 8));
  {
    std::vector<uint16_t> ha(abytes / 2);
    for (auto& x : ha) { _Float16 v = (_Float16)((rand() % 2048) / 1024.0f - 1.0f); x = __builtin_bit_cast(uint16_t, v); }
    for (int r = 0; r < 8; ++r) CK(hipMemcpy((char*)cx.apk + (size_t)r * cx.arep_stride, ha.data(), abytes, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&cx.c, (size_t)M * N * 2));
  CK(hipMalloc(&cx.part, (size_t)8 * M * N * 4 + 4096));
  CK(hipMalloc(&cx.trace, (size_t)(N / 64) * 8 * 16 * 16 * 8));
  CK(hipMalloc(&cx.out, 1 << 20));
  printf("== K=%d N=%d M=%d  (%.1f MB of packed weights, %zu copies cycled)\n", K, N, M, wbytes / 1e6, cx.qw.size());
  run_shipped(cx, ship_nseg, ship_ksplit);
#define V(NWV, NSEG, DEPTH, MODE) run_variant<NWV, NSEG, DEPTH, MODE>(cx, #NWV "w/" #NSEG "s/d" #DEPTH "/" #MODE)
  if (K == 4096 && N > 8192) {
    V(4, 8, 2, 0);
    V(4, 8, 2, MODE_RSPRE | MODE_PRESC | MODE_ZEND);
  } else if (K == 4096) {
    V(4, 8, 2, 0);
    V(4, 4, 2, 0);
    V(4, 4, 4, 0);
    V(4, 2, 2, 0);
    V(4, 1, 1, 0);
    V(8, 4, 2, 0);
    V(8, 4, 4, 0);
    V(8, 2, 2, 0);
    V(8, 1, 1, 0);
    V(4, 2, 2, MODE_NOA);
    V(4, 2, 2, MODE_NOA | MODE_NOW);
    V(4, 2, 2, MODE_NOA | MODE_NOW | MODE_NOMFMA);
    V(4, 4, 4, MODE_TRACE);
    V(4, 2, 2, MODE_TRACE);
    V(8, 1, 1, MODE_TRACE);
  } else {
    V(4, 7, 2, 0);
    V(4, 7, 2, MODE_RSPRE | MODE_PRESC);
    V(4, 7, 2, MODE_RSPRE | MODE_PRESC | MODE_ZEND);
    V(4, 7, 2, MODE_NOA | MODE_NOW);
    V(4, 7, 2, MODE_RSPRE | MODE_PRESC | MODE_ZEND | MODE_NOA | MODE_NOW);
  }
#undef V
  // A stream alone, and the same with the fixes
  if (K == 4096) {
    auto pa = [&](const char* nm, auto kern) {
      auto launch = [&](int) { hipLaunchKernelGGL(kern, dim3(N / 64), dim3(256), 0, 0, cx.apk, K, cx.mtiles, cx.out, cx.arep_stride); };
      const double us = time_us(launch);
      printf("  probe_a %-26s %4d WGs x 4 waves x 64 KiB of A : %7.2f us  %6.0f GB/s (L2->L1)\n", nm, N / 64, us, (double)(N / 64) * 256 * 1024 / us / 1e3);
    };
    pa("lockstep", aphro::probe_a_kernel<4, 8, 0>);
    pa("rot", aphro::probe_a_kernel<4, 8, MODE_ROT>);
    pa("ilv", aphro::probe_a_kernel<4, 8, MODE_ILV>);
    pa("rot+ilv", aphro::probe_a_kernel<4, 8, MODE_ROT | MODE_ILV>);
    pa("arep", aphro::probe_a_kernel<4, 8, MODE_AREP>);
    pa("arep+rot+ilv", aphro::probe_a_kernel<4, 8, MODE_AREP | MODE_ROT | MODE_ILV>);
  }
  for (auto q : cx.qw) CK(hipFree(q));
  CK(hipFree(cx.qz)); CK(hipFree(cx.sc)); CK(hipFree(cx.apk)); CK(hipFree(cx.c)); CK(hipFree(cx.part)); CK(hipFree(cx.trace)); CK(hipFree(cx.out));
}

static void l2_probe() {
  uint8_t* buf; uint32_t* out;
  CK(hipMalloc(&buf, 64 << 20)); CK(hipMemset(buf, 1, 64 << 20)); CK(hipMalloc(&out, 1 << 20));
  printf("== L2 interleave probe: 2048 WGs x 4 waves, 64 x 1 KiB sc1 reads per wave from R regions at spacing S\n");
  const int Rs[] = {1, 2, 4, 8, 16, 16, 16, 16, 16, 16, 16, 64, 64, 256};
  const int Ss[] = {1024, 4096, 4096, 4096, 1024, 2048, 4096, 8192, 16384, 65536, 4096 + 256, 4096, 1024, 4096};
  for (int i = 0; i < 14; ++i) {
    auto launch = [&](int) { hipLaunchKernelGGL(aphro::probe_l2_kernel, dim3(2048), dim3(256), 0, 0, buf, Rs[i], Ss[i], 64, out); };
    const double us = time_us(launch, 10);
    printf("  R=%3d S=%6d : %8.2f us  %7.0f GB/s\n", Rs[i], Ss[i], us, 2048.0 * 4 * 64 * 1024 / us / 1e3);
  }
  CK(hipFree(buf)); CK(hipFree(out));
}

int main(int argc, char** argv) {
  std::vector<std::string> which;
  for (int i = 1; i < argc; ++i) which.push_back(argv[i]);
  auto want = [&](const char* s) { return which.empty() || std::find(which.begin(), which.end(), std::string(s)) != which.end(); };
  if (want("l2")) l2_probe();
  if (want("gate_up")) shape(4096, 28672, 32, 8, 1);
  if (want("down")) shape(14336, 4096, 32, 7, 4);
  if (want("qkv")) shape(4096, 6144, 32, 4, 2);
  if (want("o")) shape(4096, 4096, 32, 2, 4);
  return 0;
}
