// gkg_api.hip — version / error-string part of the C ABI (include/gkg_hip.h).
#include <stdio.h>
#include <string.h>

#include "gkg_common.h"

static thread_local char g_err[256] = "";

int gkg_fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int gkg_fail_hip(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return (int)e;
}

extern "C" int gkg_version(void) { return GKG_ABI_VERSION; }

extern "C" const char* gkg_last_error_string(void) { return g_err; }
