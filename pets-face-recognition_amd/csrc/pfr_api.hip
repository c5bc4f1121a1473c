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

// ---- tuning knobs (pfr_common.h: PfrKnob) ----
static const struct { const char* name; int def; } kKnobs[KNOB_COUNT] = {
  {"igemm_p", 1}, {"igemm_ptile", -1}, {"igemm_pkch", 8}, {"igemm_ppf", 0}, {"igemm_tile", -1}, {"igemm_kch", 0}, {"igemm_big", 1},
  {"sconv", 1}, {"sconv3", 1}, {"bnb", 0}, {"swgrad", 1}, {"wgrad_big", 0}, {"wgrad_tile", -1}, {"wgrad_splits", 0}, {"wgrad9", 1},
  {"wgrad9_slots", 256}, {"attn_mfma", 1}, {"bnb_tile3", 0}, {"slin", 1}, {"slin_np", 0}, {"match_order", 1},
  {"ln_rb", 4}, {"sstem", 1}, {"igemm_dma", 0x101}, {"igemm_krot", 0},
};
int g_pfr_knob[KNOB_COUNT];
static const bool g_knobs_ready = [] { for (int k = 0; k < KNOB_COUNT; ++k) g_pfr_knob[k] = kKnobs[k].def; return true; }();   // (order = enum PfrKnob)
static int g_tuning_epoch = 0;
extern "C" int pfr_tuning_epoch(void) { return g_tuning_epoch; }
extern "C" int pfr_set_tuning(const char* key, int value) {
  PFR_CHECK_ARG(key, "pfr_set_tuning: null key");
  for (int k = 0; k < KNOB_COUNT; ++k)
    if (!strcmp(kKnobs[k].name, key)) {
      if (g_pfr_knob[k] != value) { g_pfr_knob[k] = value; ++g_tuning_epoch; }
      return PFR_OK;
    }
  pfr_set_error("pfr_set_tuning: unknown key %s", key);
  return PFR_ERR_ARG;
}
extern "C" int pfr_get_tuning(const char* key, int* value) {
  PFR_CHECK_ARG(key && value, "pfr_get_tuning: null pointer");
  for (int k = 0; k < KNOB_COUNT; ++k)
    if (!strcmp(kKnobs[k].name, key)) { *value = g_pfr_knob[k]; return PFR_OK; }
  pfr_set_error("pfr_get_tuning: unknown key %s", key);
  return PFR_ERR_ARG;
}
