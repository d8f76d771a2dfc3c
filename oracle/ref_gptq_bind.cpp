// TEST INFRASTRUCTURE ONLY (oracle/): runs the REFERENCE's own GPTQ CUDA kernels on the host.
//
// oracle/_ref/q_gemm_host.inc is the reference's kernels/quantization/gptq/q_gemm.cu with its host-side launchers
// removed (oracle/gen_ref_gptq.py, at build time, from the file where it lies); its kernels -- shuffle_4bit_kernel,
// make_sequential_4bit_kernel, gemm_half_q_half_gptq_4bit_kernel, reconstruct_exllama_4bit_kernel,
// reconstruct_gptq_kernel -- and the headers they include (qdq_4.cuh, matrix_view.cuh, ...) are compiled UNMODIFIED
// against oracle/cuda_host_shim/.  This file supplies (a) the launch emulator: every emulated thread is a fiber, a
// block's fibers run round-robin between __syncthreads(); and (b) C entry points that launch the kernels with the
// grid / block shapes of the launchers that were removed (cited below).  What comes out are the reference's bits:
// the pin for oracle/quant.py's gptq_shuffle / gptq_dequant / gptq_gemm restatements.
#include "cuda_host_shim.h"

uint3_ threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace cuemu {
namespace {
struct Fiber {
  ucontext_t ctx;
  bool done;
  uint3_ tid;
};
ucontext_t g_sched;
Fiber* g_cur = nullptr;
const std::function<void()>* g_body = nullptr;
std::vector<char> g_stacks;
constexpr size_t kStack = 256 * 1024;

void fiber_main() {
  (*g_body)();
  g_cur->done = true;
  swapcontext(&g_cur->ctx, &g_sched);
}
}  // namespace

void syncthreads() { swapcontext(&g_cur->ctx, &g_sched); }

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  const size_t nthreads = (size_t)block.x * block.y * block.z;
  if (g_stacks.size() < nthreads * kStack) g_stacks.resize(nthreads * kStack);
  std::vector<Fiber> fibers(nthreads);
  gridDim = grid;
  blockDim = block;
  g_body = &body;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = uint3_{bx, by, bz};
        size_t t = 0;
        for (unsigned tz = 0; tz < block.z; ++tz)
          for (unsigned ty = 0; ty < block.y; ++ty)
            for (unsigned tx = 0; tx < block.x; ++tx, ++t) {
              Fiber& f = fibers[t];
              getcontext(&f.ctx);
              f.ctx.uc_stack.ss_sp = g_stacks.data() + t * kStack;
              f.ctx.uc_stack.ss_size = kStack;
              f.ctx.uc_link = nullptr;
              f.done = false;
              f.tid = uint3_{tx, ty, tz};
              makecontext(&f.ctx, fiber_main, 0);
            }
        size_t live = nthreads;
        while (live) {          // one round = every live thread runs to its next barrier (or to the end)
          for (size_t i = 0; i < nthreads; ++i) {
            if (fibers[i].done) continue;
            g_cur = &fibers[i];
            threadIdx = fibers[i].tid;
            swapcontext(&g_sched, &fibers[i].ctx);
            if (fibers[i].done) --live;
          }
        }
      }
}
}  // namespace cuemu

#include "q_gemm_host.inc"

using namespace aphrodite::gptq;

extern "C" {

// aphrodite::gptq::shuffle_exllama_weight (q_gemm.cu:1822-1872), 4-bit, one expert: act-order rows made sequential
// through q_perm (make_sequential_4bit_kernel), then every word shuffled (shuffle_4bit_kernel).  In place.
void ref_gptq_shuffle(uint32_t* q_weight, const int* q_perm, int height, int width) {
  const int bit = 4;
  if (q_perm) {
    std::vector<uint32_t> new_qweight((size_t)height / 32 * bit * width);
    dim3 block(THREADS_X, 1, 1), grid(DIVIDE(width, THREADS_X), height / 32 * bit, 1);
    uint32_t* nw = new_qweight.data();
    cuemu::launch(grid, block, [&] { make_sequential_4bit_kernel(q_weight, nw, q_perm, height / 32 * bit, width); });
    std::memcpy(q_weight, nw, new_qweight.size() * 4);
  }
  dim3 block(THREADS_X, 1, 1), grid(DIVIDE(width, THREADS_X), 1, 1);
  cuemu::launch(grid, block, [&] { shuffle_4bit_kernel(q_weight, height, width); });
}

// gemm_half_q_half_cuda's exllama branch for size_m <= MAX_Q_GEMM_ROWS (q_gemm.cu:1545-1562) through
// gemm_half_q_half_cuda_part (:737-757): chunks of BLOCK_M_SIZE_MAX rows, grid (N/512, M/m_count, K/128), 128 threads.
// b_q_weight is the SHUFFLED weight, b_q_perm the act-order permutation or NULL.  c must be zeroed by the caller
// exactly as torch::empty + the kernel's own "if (blockIdx.z == 0) zero" leave it: the kernel zeroes it itself.
static void gemm_part(const half* a, const uint32_t* qw, const uint32_t* qz, const half* sc, const int* perm, half* c,
                      int size_m, int size_n, int size_k, int m_count, int groups) {
  dim3 block(BLOCK_KN_SIZE, 1, 1);
  dim3 grid(DIVIDE(size_n, BLOCK_KN_SIZE * 4), DIVIDE(size_m, m_count), DIVIDE(size_k, BLOCK_KN_SIZE));
  fp_gemm_half_q_half_gptq_kernel kernel = pick_gemm_half_q_half_gptq_kernel(true, m_count, 4);
  cuemu::launch(grid, block, [&] { kernel(a, qw, qz, sc, c, size_m, size_n, size_k, groups, perm); });
}
int ref_gptq_gemm_exllama(const uint16_t* a, const uint32_t* b_q_weight, const uint32_t* b_gptq_qzeros,
                          const uint16_t* b_gptq_scales, const int* b_q_perm, uint16_t* c, int size_m, int size_n,
                          int size_k, int groups) {
  if (size_m > MAX_Q_GEMM_ROWS) return -1;      // the reference reconstructs + calls cuBLAS above 50 rows
  const half* ah = reinterpret_cast<const half*>(a);
  half* ch = reinterpret_cast<half*>(c);
  const half* sh = reinterpret_cast<const half*>(b_gptq_scales);
  const int max_chunks = size_m / BLOCK_M_SIZE_MAX;
  const int last_chunk = max_chunks * BLOCK_M_SIZE_MAX;
  const int last_chunk_size = size_m - last_chunk;
  if (max_chunks) gemm_part(ah, b_q_weight, b_gptq_qzeros, sh, b_q_perm, ch, last_chunk, size_n, size_k, BLOCK_M_SIZE_MAX, groups);
  if (last_chunk_size)
    gemm_part(ah + (size_t)last_chunk * size_k, b_q_weight, b_gptq_qzeros, sh, b_q_perm, ch + (size_t)last_chunk * size_n,
              last_chunk_size, size_n, size_k, last_chunk_size, groups);
  return 0;
}

// reconstruct_exllama (q_gemm.cu:1157-1182): fp16 [K, N] from the SHUFFLED weight; grid (N/128... see below), 128 threads
void ref_gptq_reconstruct_exllama(const uint32_t* b_q_weight, const int* b_q_perm, const uint32_t* b_gptq_qzeros,
                                  const uint16_t* b_gptq_scales, int height, int width, int groups, uint16_t* out) {
  dim3 block(BLOCK_KN_SIZE, 1, 1), grid(DIVIDE(width, BLOCK_KN_SIZE), DIVIDE(height, BLOCK_KN_SIZE), 1);
  cuemu::launch(grid, block, [&] {
    reconstruct_exllama_4bit_kernel(b_q_weight, b_q_perm, b_gptq_qzeros, reinterpret_cast<const half*>(b_gptq_scales), height,
                                    width, groups, reinterpret_cast<half*>(out));
  });
}

// reconstruct_gptq (q_gemm.cu:1480-1505), 4-bit: fp16 [K, N] from the checkpoint-order (UNshuffled) weight + g_idx
void ref_gptq_reconstruct(const uint32_t* b_q_weight, const uint32_t* b_gptq_qzeros, const uint16_t* b_gptq_scales,
                          const int* b_g_idx, int height, int width, int groups, uint16_t* out) {
  const int bit = 4;
  dim3 block(BLOCK_KN_SIZE, 1, 1), grid(DIVIDE(width, BLOCK_KN_SIZE), DIVIDE(height, 32 / bit), 1);
  cuemu::launch(grid, block, [&] {
    reconstruct_gptq_kernel<MatrixView_q4_row, 4>(b_q_weight, reinterpret_cast<const half*>(b_gptq_scales), b_gptq_qzeros, b_g_idx,
                                                  height, width, groups, reinterpret_cast<half*>(out));
  });
}

// reconstruct_gptq (q_gemm.cu:1480-1505) for the other widths: <MatrixView_q2_row, 2>, reconstruct_gptq_3bit_kernel,
// <MatrixView_q8_row, 8> -- the DEFINITION of the 2 / 3 / 8-bit checkpoint layout (values and zero points laid end to end,
// 3-bit values 10 and 21 straddling words) that oracle/quant.py's bitstring pack / unpack restates.
int ref_gptq_reconstruct_bits(const uint32_t* b_q_weight, const uint32_t* b_gptq_qzeros, const uint16_t* b_gptq_scales,
                              const int* b_g_idx, int height, int width, int groups, int bit, uint16_t* out) {
  const half* sc = reinterpret_cast<const half*>(b_gptq_scales);
  half* o = reinterpret_cast<half*>(out);
  dim3 block(BLOCK_KN_SIZE, 1, 1), grid(DIVIDE(width, BLOCK_KN_SIZE), bit == 3 ? DIVIDE(height, 32) : DIVIDE(height, 32 / bit), 1);
  if (bit == 2)
    cuemu::launch(grid, block, [&] { reconstruct_gptq_kernel<MatrixView_q2_row, 2>(b_q_weight, sc, b_gptq_qzeros, b_g_idx, height, width, groups, o); });
  else if (bit == 3)
    cuemu::launch(grid, block, [&] { reconstruct_gptq_3bit_kernel(b_q_weight, sc, b_gptq_qzeros, b_g_idx, height, width, groups, o); });
  else if (bit == 8)
    cuemu::launch(grid, block, [&] { reconstruct_gptq_kernel<MatrixView_q8_row, 8>(b_q_weight, sc, b_gptq_qzeros, b_g_idx, height, width, groups, o); });
  else
    return -1;
  return 0;
}

// fp32 -> binary16 bits with the shim's RNE conversion (so that fixtures can be made without torch)
uint16_t ref_f32_to_f16(float f) { return cuemu::d2h((double)f); }
float ref_f16_to_f32(uint16_t h) { return cuemu::h2f(h); }
}
