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

// ---- the environment (common.h: Knobs) ------------------------------------------------------------------------------------
namespace aphro {
static int env_i(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static long env_l(const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; }
static Knobs read_knobs() {
  Knobs k;
  k.pa_splits = env_i("APHRO_PA_SPLITS", 0);
  k.fa_v4_min_keys = env_i("APHRO_FA_V4_MIN_KEYS", 4096);
  k.fa_no_xcd = env_i("APHRO_FA_NO_XCD", 0);
  k.fp8_stream_all = env_i("APHRO_FP8_STREAM_ALL", 0);
  k.wna16_stream = env_i("APHRO_WNA16_STREAM", 1);
  k.wna16_op_no_resident = env_i("APHRO_WNA16_OP_NO_RESIDENT", 0);
  k.wna16_large_8phase = env_i("APHRO_WNA16_LARGE_8PHASE", -1);
  k.wna16_large_two_pass = env_i("APHRO_WNA16_LARGE_TWO_PASS", -1);
  k.wna16_mid_waves = env_i("APHRO_WNA16_MID_WAVES", 0);
  k.res_cfg_set = 0;
  k.res_cfg[0] = k.res_cfg[1] = k.res_cfg[2] = k.res_cfg[3] = 0;
  if (const char* e = getenv("APHRO_WNA16_RES_CFG"))
    k.res_cfg_set = sscanf(e, "%d,%d,%d,%d", &k.res_cfg[0], &k.res_cfg[1], &k.res_cfg[2], &k.res_cfg[3]) == 4 ? 1 : -1;
  k.ar_one_shot_max = env_l("APHRO_CUSTOM_AR_ONE_SHOT_MAX", -1);
  k.ar_timeout_ms = env_l("APHRODITE_CUSTOM_AR_TIMEOUT_MS", 0);
  k.cu_masked = (getenv("HSA_CU_MASK") || getenv("ROC_GLOBAL_CU_MASK")) ? 1 : 0;
  return k;
}
static Knobs g_knobs = read_knobs();
const Knobs& knobs() { return g_knobs; }
}  // namespace aphro
// Re-read the environment switches (a test or a tool that changed one inside the process; never needed on a serving path).
extern "C" void aphro_reload_env(void) { aphro::g_knobs = aphro::read_knobs(); }

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
