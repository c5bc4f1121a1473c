// pfr_plan.hip — C-side executor of a pre-built launch plan.
//
// A training / inference step of the engines (models/_fe_engine.py, models/_swin_engine.py) is a fixed list of C-ABI calls with
// fixed device pointers plus the fork / join edges between the main stream and the weight-gradient side stream.  Replaying that
// list from Python costs ~20 µs per entry (ctypes marshalling + interpreter): ~16 ms per ResNet-50 step, 2/3 of the GPU time.
// Here the list lives in C: an entry is {thunk, flat 64-bit argument slots, role}; pfr_plan_run walks it with ~0.3 µs of host
// time per launch.  It replaces, on the host side, what the reference gets from PyTorch's autograd engine + PL's training loop
// (/root/reference/engine/trainer.py:403-425): nothing numeric — only WHO calls the kernels.
//
// Roles (kind): 0 launch on the main stream | 1 launch on the side stream (main when the side stream is disabled) |
//   2 fork: record event ev on main, side waits for it | 3 record event ev on side | 4 main waits for event ev |
//   5 as 4, but only when the caller asked for hook stops (gradient-bucket marks of data-parallel training) |
//   6 hook stop: pfr_plan_run returns its index so that the host can run its callback and resume after it.
#include "pfr_common.h"
#include <stdint.h>
#include <string.h>
#include <vector>

// every entry point a thunk may call
#include "../../include/pfr_hip.h"

struct PlanThunk {
  const char* name;
  int (*fn)(const unsigned long long*, void*);
};
static inline float u2f_(unsigned long long v) {
  const uint32_t b = (uint32_t)v;
  float f;
  memcpy(&f, &b, 4);
  return f;
}
#include "pfr_thunks_gen.inc"

struct PlanOp {
  int kind, thunk, ev, arg;
  unsigned long long a[30];
};
struct Plan {
  std::vector<PlanOp> ops;
  std::vector<hipEvent_t> events;
};

extern "C" int pfr_plan_thunk_index(const char* name) {
  const int n = (int)(sizeof(g_thunks) / sizeof(g_thunks[0]));
  for (int i = 0; i < n; ++i)
    if (!strcmp(g_thunks[i].name, name)) return i;
  return -1;
}

extern "C" void* pfr_plan_create(int n_events) {
  Plan* p = new Plan();
  p->events.resize(n_events > 0 ? n_events : 0);
  for (auto& e : p->events)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      delete p;
      pfr_set_error("pfr_plan_create: hipEventCreate failed");
      return nullptr;
    }
  return p;
}

extern "C" int pfr_plan_destroy(void* plan) {
  Plan* p = reinterpret_cast<Plan*>(plan);
  if (!p) return PFR_OK;
  for (auto& e : p->events) (void)hipEventDestroy(e);
  delete p;
  return PFR_OK;
}

// appends one entry; args = nargs 64-bit slots (ignored for kinds >= 2, where `ev` is the event index / hook argument)
extern "C" int pfr_plan_append(void* plan, int kind, int thunk, int ev, const unsigned long long* args, int nargs) {
  Plan* p = reinterpret_cast<Plan*>(plan);
  PFR_CHECK_ARG(p && kind >= 0 && kind <= 6, "pfr_plan_append: bad plan / kind");
  PFR_CHECK_ARG(kind >= 2 || (thunk >= 0 && thunk < (int)(sizeof(g_thunks) / sizeof(g_thunks[0])) && nargs >= 0 && nargs <= 30 && args),
                "pfr_plan_append: bad thunk / argument count");
  PFR_CHECK_ARG(kind < 2 || kind == 6 || (ev >= 0 && ev < (int)p->events.size()), "pfr_plan_append: bad event index");
  PlanOp op;
  op.kind = kind; op.thunk = thunk; op.ev = ev; op.arg = nargs;
  if (kind < 2) memcpy(op.a, args, sizeof(unsigned long long) * nargs);
  p->ops.push_back(op);
  return PFR_OK;
}

extern "C" int pfr_plan_size(void* plan) { return plan ? (int)reinterpret_cast<Plan*>(plan)->ops.size() : 0; }

// runs entries [begin, end) (end < 0: to the end); returns -1 when it reached the end, the index of a hook stop (kind 6,
// only with hook_stops != 0) when it stopped there, or a value <= -2 on error (pfr_last_error).
extern "C" int pfr_plan_run(void* plan, int begin, int end, pfr_stream_t main_stream, pfr_stream_t side_stream, int hook_stops) {
  Plan* p = reinterpret_cast<Plan*>(plan);
  if (!p) { pfr_set_error("pfr_plan_run: null plan"); return -2; }
  const int n = (int)p->ops.size();
  if (end < 0 || end > n) end = n;
  hipStream_t ms = (hipStream_t)main_stream, ss = (hipStream_t)side_stream;
  const bool use_side = ss != nullptr;
  static const int apply_thunk = pfr_plan_thunk_index("pfr_bn_bwd_apply");
  int attached = -1;
  for (int i = begin; i < end; ++i) {
    const PlanOp& op = p->ops[i];
    switch (op.kind) {
      case 0:
      case 1: {
        // a main-stream launch directly followed by a fork: let the kernel's own completion signal be the fork event
        if (op.kind == 0 && use_side && op.thunk == apply_thunk && i + 1 < end && p->ops[i + 1].kind == 2) {
          pfr_tls_stop_event = p->events[p->ops[i + 1].ev];
          attached = i + 1;
        }
        const int rc = g_thunks[op.thunk].fn(op.a, (op.kind == 1 && use_side) ? (void*)ss : (void*)ms);
        // the launch clears the thread-local when it attaches the event; if it returned before that point (argument error) the
        // event is NOT attached: disarm it so that no later launch of this thread picks up a foreign plan's event
        if (pfr_tls_stop_event != nullptr) {
          pfr_tls_stop_event = nullptr;
          attached = -1;
        }
        if (rc != PFR_OK) return rc <= -2 ? rc : -2 + (rc < 0 ? rc : 0) - 1;
        break;
      }
      case 2:
        if (use_side) {
          if ((attached != i && hipEventRecord(p->events[op.ev], ms) != hipSuccess) || hipStreamWaitEvent(ss, p->events[op.ev], 0) != hipSuccess) {
            pfr_set_error("pfr_plan_run: fork failed");
            return -2;
          }
        }
        break;
      case 3:
        if (use_side && hipEventRecord(p->events[op.ev], ss) != hipSuccess) { pfr_set_error("pfr_plan_run: record failed"); return -2; }
        break;
      case 5:
        if (hook_stops != 1) break;   // 2: the hook makes its own (communication) stream wait for the side stream
        [[fallthrough]];
      case 4:
        if (use_side && hipStreamWaitEvent(ms, p->events[op.ev], 0) != hipSuccess) { pfr_set_error("pfr_plan_run: wait failed"); return -2; }
        break;
      case 6:
        if (hook_stops) return i;
        break;
    }
  }
  return -1;
}
