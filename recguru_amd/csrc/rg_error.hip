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
