// Error plumbing shared by every launcher.
#include "rg_common.hip.h"
#include "../../include/recguru_hip.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

int rg_set_error(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: HIP error %d (%s)", where, (int)e, hipGetErrorString(e));
  return RG_ERR_HIP;
}
int rg_set_error_msg(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
extern "C" const char* rg_last_error(void) { return g_err; }
extern "C" int rg_version(void) { return 1; }

// ---- deterministic-reduction build (rg_det.hip.h): the arenas' descriptors live once per translation unit (static __device__), so
// every unit that accumulates registers a setter / a fault reader here when the library is loaded.
#define RG_DET_MAX_TU 16
static int (*g_det_set[RG_DET_MAX_TU])(const void*);
static int (*g_det_fault[RG_DET_MAX_TU])(int*, int);
static int g_det_ntu = 0;
static int g_det_dropped = 0;      // units that found the table full: their accumulators would fall back to float atomics UNREPORTED
extern "C" __attribute__((visibility("hidden"))) void rg_det_register_tu(int (*set)(const void* arenas), int (*fault)(int* out, int clear)) {
  if (g_det_ntu < RG_DET_MAX_TU) { g_det_set[g_det_ntu] = set; g_det_fault[g_det_ntu] = fault; ++g_det_ntu; }
  else ++g_det_dropped;
}
extern "C" int rg_det_enabled(void) { return g_det_ntu; }
extern "C" int rg_det_set_arenas(void* fbase0, void* sbase0, unsigned long long bytes0, int bits0,
                                 void* fbase1, void* sbase1, unsigned long long bytes1, int bits1) {
  if (!g_det_ntu) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "rg_det_set_arenas: this library accumulates with float atomics (load librecguru_hip_det.so)");
  struct { unsigned long long fbase, sbase, bytes; int bits; int pad; } a[2] = {
      {(unsigned long long)fbase0, (unsigned long long)sbase0, bytes0, bits0, 0}, {(unsigned long long)fbase1, (unsigned long long)sbase1, bytes1, bits1, 0}};
  if (bits0 < 0 || bits0 > 60 || bits1 < 0 || bits1 > 60) return rg_set_error_msg(RG_ERR_INVALID, "rg_det_set_arenas: bits outside 0..60");
  if (g_det_dropped) return rg_set_error_msg(RG_ERR_UNSUPPORTED, "rg_det_set_arenas: more accumulating translation units than RG_DET_MAX_TU (csrc/rg_error.hip): "
                                             "the units beyond it would add with float atomics and report no fault -- raise RG_DET_MAX_TU");
  for (int i = 0; i < g_det_ntu; ++i) {
    const int e = g_det_set[i](a);
    if (e) return rg_set_error((hipError_t)e, "rg_det_set_arenas");
  }
  return 0;
}
extern "C" int rg_det_fault(int clear) {
  int v = g_det_dropped ? 1 : 0;                        // an unregistered unit adds outside the shadows by construction
  for (int i = 0; i < g_det_ntu; ++i) {
    const int e = g_det_fault[i](&v, clear);
    if (e) { rg_set_error((hipError_t)e, "rg_det_fault"); return -1; }
  }
  return v;
}
