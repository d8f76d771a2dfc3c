// Where do direct-to-LDS buffer loads land?  (gfx950: buffer_load_dwordx4 ... lds, M0 = LDS base)
// Each lane loads 16 bytes from its own buffer offset; this probe prints, for a few lanes, which source
// dwords are found at LDS base + 16 * lane -- the layout a weights-through-LDS GEMM pipeline would rely on.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_direct_probe.hip -o /tmp/ldsp && /tmp/ldsp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* in, unsigned* out, int n) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(in), 0, n * 4, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned* base = lds + wave * 512;                    // 2 KiB per wave
  // lane l reads source bytes [64 * l, 64 * l + 16) (a strided gather, so the landing spot is unambiguous)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)base, 16, lane * 64, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(base + 256), 16, lane * 64, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512 * (int)(blockDim.x >> 6); i += blockDim.x) out[i] = lds[i];
}
int main() {
  const int n = 64 * 16 + 64;
  std::vector<unsigned> h(n);
  for (int i = 0; i < n; ++i) h[i] = i;                  // value = source dword index
  unsigned *din, *dout;
  hipMalloc(&din, n * 4); hipMalloc(&dout, 2 * 512 * 4);
  hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 2 * 2048, 0, din, dout, n);
  std::vector<unsigned> o(1024);
  if (hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 1; }
  bool ok = true;
  for (int w = 0; w < 2; ++w)
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        ok &= o[w * 512 + 4 * l + j] == (unsigned)(16 * l + j);            // first load: source dwords 16l .. 16l+3
        ok &= o[w * 512 + 256 + 4 * l + j] == (unsigned)(16 * l + 4 + j);  // second load: soffset 16 bytes -> +4 dwords
      }
  printf("lane 0: %u %u %u %u | lane 1: %u %u %u %u | lane 63: %u %u %u %u | 2nd load lane 1: %u %u %u %u\n", o[0], o[1], o[2], o[3],
         o[4], o[5], o[6], o[7], o[252], o[253], o[254], o[255], o[260], o[261], o[262], o[263]);
  printf("direct-to-LDS b128: lane l lands at base + 16*l, per wave -> %s\n", ok ? "CONFIRMED" : "NOT as assumed");
  return 0;
}
