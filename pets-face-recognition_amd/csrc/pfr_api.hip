// pfr_api.hip — C-ABI housekeeping: thread-local error string, version, device probe.
#include "pfr_common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

thread_local hipEvent_t pfr_tls_stop_event = nullptr;

void pfr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* pfr_last_error(void) { return g_err; }
extern "C" int pfr_version(void) { return 100; }

// returns the gcnArchName of the current device into buf (e.g. "gfx950:sramecc+:xnack-")
extern "C" int pfr_device_arch(char* buf, int buflen) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { pfr_set_error("pfr_device_arch: no HIP device"); return PFR_ERR_HIP; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { pfr_set_error("pfr_device_arch: hipGetDeviceProperties failed"); return PFR_ERR_HIP; }
  strncpy(buf, prop.gcnArchName, buflen - 1);
  buf[buflen - 1] = 0;
  return PFR_OK;
}
