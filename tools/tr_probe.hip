// Probe of ds_read_b64_tr_b16 (gfx950): which element does lane l / slot j receive?
// LDS image: M[r][c] = r * 100 + c, r < 16 rows, 16 cols (row stride 16 halfs = 32 B).
// Hypothesis (cdna_hip_programming.md T10): within each 16-lane group, lane i supplies the address
// of 4 contiguous halfs  &blk[i / 4][4 * (i % 4)]  of a [4][16] block; lane c receives column c:
// {blk[0][c], blk[1][c], blk[2][c], blk[3][c]}.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 16];
  for (int i = threadIdx.x; i < 64 * 16; i += 64) lds[i] = (uint16_t)((i / 16) * 100 + (i % 16));
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
  // group g reads block rows 4g .. 4g+3 (all 16 columns)
  const uint16_t* addr = &lds[(4 * g + i / 4) * 16 + 4 * (i % 4)];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)addr);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", h[l * 4 + j]); printf("\n"); }
  return 0;
}
