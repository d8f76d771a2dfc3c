// AWQ 4-bit ops.
//   awq_dequantize  kernels/quantization/awq/gemm_kernels.cu:340-403, 720-776
//   awq_gemm        kernels/quantization/awq/gemm_kernels.cu:20-335, 784-841
//                   (on ROCm today: Triton, aphrodite/quantization/awq_triton.py)
// AWQ packs along N: word c of row k holds columns 8c + {0,2,4,6,1,3,5,7}[p]
// at nibble p.  The MFMA B fragment wants 8 consecutive k of one column, so
// the op-level awq_gemm first transposes the nibbles into the CDNA4 K-packed
// exllama layout (awq_repack_kernel, one pass over the int4 bytes) and then
// runs the shared W4A16 kernel with AWQ's un-offset zero points.  The
// load-time path (AWQ "marlin role", aphro_awq_repack) does the repack once
// and calls the fast kernel directly.
#include "common.h"

namespace aphro {

// one thread = one packed word = 8 columns of one row; bit-exact vs the
// reference: (q - z) exact in fp16, one rounding in the multiply.
template <typename T>
__global__ void awq_dequantize_kernel(const uint32_t* __restrict__ qw, const uint16_t* __restrict__ sc,
                                      const uint32_t* __restrict__ qz, uint16_t* __restrict__ out, int K,
                                      int N, int group_size) {
  const int wc = blockIdx.x * blockDim.x + threadIdx.x;  // word column
  const int k = blockIdx.y;
  const int words = N >> 3;
  if (wc >= words) return;
  const uint32_t w = qw[(size_t)k * words + wc];
  const int grp = k / group_size;
  const uint32_t z = qz[(size_t)grp * words + wc];
  u16x8 s = *reinterpret_cast<const u16x8*>(sc + (size_t)grp * N + 8 * wc);
  u16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int shift = 4 * (((j & 1) << 2) | (j >> 1));  // [0,4,1,5,2,6,3,7][j]
    const int q = (int)((w >> shift) & 0xf) - (int)((z >> shift) & 0xf);
    o[j] = T::from_f32((float)q * T::to_f32(s[j]));
  }
  *reinterpret_cast<u16x8*>(out + (size_t)k * N + 8 * wc) = o;
}

}  // namespace aphro

using namespace aphro;

extern "C" int aphro_awq_dequantize(const uint32_t* qweight, const void* scales, const uint32_t* qzeros,
                                    void* out, int64_t K, int64_t N, int64_t groups, int dtype,
                                    void* stream) {
  APHRO_CHECK(dtype == APHRO_F16 || dtype == APHRO_BF16, "awq_dequantize: dtype must be f16 or bf16");
  APHRO_CHECK(N % 8 == 0 && groups > 0 && K % groups == 0, "awq_dequantize: bad shape");
  if (K == 0 || N == 0) return APHRO_OK;
  dim3 grid((unsigned)((N / 8 + 127) / 128), (unsigned)K);
  int gs = (int)(K / groups);
  if (dtype == APHRO_F16)
    hipLaunchKernelGGL((awq_dequantize_kernel<Half>), grid, dim3(128), 0, (hipStream_t)stream, qweight,
                       (const uint16_t*)scales, qzeros, (uint16_t*)out, (int)K, (int)N, gs);
  else
    hipLaunchKernelGGL((awq_dequantize_kernel<BFloat>), grid, dim3(128), 0, (hipStream_t)stream, qweight,
                       (const uint16_t*)scales, qzeros, (uint16_t*)out, (int)K, (int)N, gs);
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}

extern "C" size_t aphro_awq_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t groups) {
  size_t rep = (size_t)(K / 8) * N * 4 + (size_t)groups * (N / 8) * 4;
  rep = (rep + 255) / 256 * 256;
  return rep + aphro_wna16_workspace_bytes(M, N, K);
}

extern "C" int aphro_awq_gemm(const void* a, const uint32_t* qweight, const void* scales,
                              const uint32_t* qzeros, void* c, void* workspace, size_t workspace_bytes,
                              int64_t M, int64_t N, int64_t K, int64_t groups, int64_t lda, int dtype,
                              void* stream) {
  APHRO_CHECK(K % 8 == 0 && N % 8 == 0, "awq_gemm: K and N must be multiples of 8");
  size_t need = aphro_awq_gemm_workspace_bytes(M, N, K, groups);
  if (!workspace || workspace_bytes < need) {
    set_error("awq_gemm: workspace %zu < %zu bytes", workspace_bytes, need);
    return APHRO_ERR_WORKSPACE;
  }
  uint32_t* rq = (uint32_t*)workspace;
  uint32_t* rz = rq + (size_t)(K / 8) * N;
  size_t rep = ((size_t)(K / 8) * N * 4 + (size_t)groups * (N / 8) * 4 + 255) / 256 * 256;
  int rc = aphro_awq_repack(qweight, rq, K, N, stream);
  if (rc != APHRO_OK) return rc;
  rc = aphro_awq_repack_zeros(qzeros, rz, groups, N, stream);
  if (rc != APHRO_OK) return rc;
  return aphro_gptq_gemm(a, rq, rz, scales, nullptr, nullptr, c, (char*)workspace + rep, workspace_bytes - rep,
                         M, N, K, groups, lda, /*zero_offset=*/0, dtype, stream);
}
