// Error reporting + ABI version for libaphrodite_mi355x.so
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace aphro {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace aphro

extern "C" const char* aphro_last_error(void) { return aphro::g_err; }
extern "C" int aphro_abi_version(void) { return 1; }

// A launch that holds its stream for `us` microseconds and touches no memory: the stand-in for a collective when ONE rank
// of a tensor-parallel group is timed on a one-GPU box (bench.py --sim-tp: the all-reduce is replaced by its measured
// latency).  One wave, polling the 100 MHz wall clock with s_sleep in between.
namespace aphro {
__global__ void spin_kernel(long long ticks) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}
}  // namespace aphro
extern "C" int aphro_spin_us(double us, void* stream) {
  if (us <= 0) return APHRO_OK;
  hipLaunchKernelGGL(aphro::spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)(us * 100.0));
  APHRO_LAUNCH_CHECK();
  return APHRO_OK;
}
