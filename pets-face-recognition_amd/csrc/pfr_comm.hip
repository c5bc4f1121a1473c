// pfr_comm.hip — gradient all-reduce over RCCL behind the C-ABI (pfr_comm_{init,allreduce,destroy}).
//
// Replaces, for a host that binds libpfr_hip.so directly, what the reference gets from
// `DistributedDataParallel` (/root/reference/utils/__init__.py:114-119): an in-place mean (or sum) all-reduce of a flat gradient
// range on a stream of the caller's.  The Python host of this repo uses torch.distributed (backend "nccl" = RCCL) for the same
// collective (engine/ddp.py); both end in ncclAllReduce over xGMI.
//
// RCCL is resolved at FIRST USE with dlopen (the copy the process already holds — torch ships its own — else the system one):
// the library has no load-time dependency on RCCL, so single-GPU users never touch it and two RCCL copies cannot clash.
#include "pfr_common.h"
#include <dlfcn.h>
#include <string.h>

namespace {
struct NcclUid { char internal[128]; };
typedef int (*fn_uid)(NcclUid*);
typedef int (*fn_init)(void**, int, NcclUid, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);
struct Rccl {
  void* h = nullptr;
  fn_uid uid = nullptr;
  fn_init init = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_destroy destroy = nullptr;
  fn_errstr errstr = nullptr;
};
Rccl g_rccl;

bool rccl_load() {
  if (g_rccl.h) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {   // a copy that is already mapped (torch's) wins
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
    if (h) break;
  }
  if (!h)
    for (const char* n : names) {
      h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
  if (!h) { pfr_set_error("pfr_comm: librccl.so not found (%s)", dlerror()); return false; }
  g_rccl.uid = (fn_uid)dlsym(h, "ncclGetUniqueId");
  g_rccl.init = (fn_init)dlsym(h, "ncclCommInitRank");
  g_rccl.allreduce = (fn_allreduce)dlsym(h, "ncclAllReduce");
  g_rccl.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
  g_rccl.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
  if (!g_rccl.uid || !g_rccl.init || !g_rccl.allreduce || !g_rccl.destroy) { pfr_set_error("pfr_comm: RCCL symbols missing"); return false; }
  g_rccl.h = h;
  return true;
}
int rccl_fail(const char* what, int rc) {
  pfr_set_error("%s: %s (%d)", what, g_rccl.errstr ? g_rccl.errstr(rc) : "RCCL error", rc);
  return PFR_ERR_HIP;
}
}  // namespace

// 128-byte rendezvous id: rank 0 creates it, the host hands it to every rank (file, socket, MPI, torchrun store, ...)
extern "C" int pfr_comm_unique_id(void* id128) {
  PFR_CHECK_ARG(id128, "pfr_comm_unique_id: null pointer");
  if (!rccl_load()) return PFR_ERR_UNSUPPORTED;
  NcclUid u;
  const int rc = g_rccl.uid(&u);
  if (rc) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(id128, &u, sizeof(u));
  return PFR_OK;
}
// The handle: the RCCL communicator plus a stream of its OWN.  Collectives are enqueued on that stream and tied to the caller's stream by
// two events (caller -> own before, own -> caller after): for the caller the call is stream-ordered exactly as before, but RCCL never sees
// a stream that carries other work.  Measured (round 6, world 1, profiles/r06_ddp_w1.txt): with ncclAllReduce enqueued directly on the
// engine's communication stream the train step took 26.8 instead of 17.6 ms whenever the weight-gradient side stream was on — the
// communication stream and the side stream ended up multiplexed on one hardware queue — while torch.distributed, which runs its
// collectives on an internal stream in just this way, did not.
struct PfrComm {
  void* nccl = nullptr;
  hipStream_t own = nullptr;
  hipEvent_t before = nullptr, after = nullptr;
};
// one communicator per process and GPU (the current HIP device); returns an opaque handle or NULL (pfr_last_error)
extern "C" void* pfr_comm_init(int rank, int world, const void* id128) {
  if (!id128 || world < 1 || rank < 0 || rank >= world) { pfr_set_error("pfr_comm_init: bad arguments"); return nullptr; }
  if (!rccl_load()) return nullptr;
  NcclUid u;
  memcpy(&u, id128, sizeof(u));
  PfrComm* c = new PfrComm();
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // (hi = the numerically lowest = highest priority)
  if (hipStreamCreateWithPriority(&c->own, hipStreamNonBlocking, hi) != hipSuccess ||
      hipEventCreateWithFlags(&c->before, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->after, hipEventDisableTiming) != hipSuccess) {
    pfr_set_error("pfr_comm_init: stream / event creation failed");
    delete c;
    return nullptr;
  }
  const int rc = g_rccl.init(&c->nccl, world, u, rank);
  if (rc) { rccl_fail("ncclCommInitRank", rc); (void)hipStreamDestroy(c->own); (void)hipEventDestroy(c->before); (void)hipEventDestroy(c->after); delete c; return nullptr; }
  return c;
}
// in-place all-reduce of `count` elements (dtype PFR_F32 | PFR_BF16) on `stream`; average != 0: mean over the ranks (what DDP does)
extern "C" int pfr_comm_allreduce(void* comm, void* buf, size_t count, int dtype, int average, hipStream_t stream) {
  PFR_CHECK_ARG(comm && buf, "pfr_comm_allreduce: null pointer");
  PFR_CHECK_ARG(dtype == PFR_F32 || dtype == PFR_BF16, "pfr_comm_allreduce: bad dtype %d", dtype);
  if (!rccl_load()) return PFR_ERR_UNSUPPORTED;
  const int nccl_dtype = dtype == PFR_F32 ? 7 : 9;   // ncclFloat32, ncclBfloat16
  const int nccl_op = average ? 4 : 0;               // ncclAvg, ncclSum
  PfrComm* c = reinterpret_cast<PfrComm*>(comm);
  if (hipEventRecord(c->before, stream) != hipSuccess || hipStreamWaitEvent(c->own, c->before, 0) != hipSuccess) {
    pfr_set_error("pfr_comm_allreduce: stream hand-over failed");
    return PFR_ERR_HIP;
  }
  const int rc = g_rccl.allreduce(buf, buf, count, nccl_dtype, nccl_op, c->nccl, c->own);
  if (rc) return rccl_fail("ncclAllReduce", rc);
  if (hipEventRecord(c->after, c->own) != hipSuccess || hipStreamWaitEvent(stream, c->after, 0) != hipSuccess) {
    pfr_set_error("pfr_comm_allreduce: stream hand-back failed");
    return PFR_ERR_HIP;
  }
  return PFR_OK;
}
extern "C" int pfr_comm_destroy(void* comm) {
  if (!comm) return PFR_OK;
  if (!rccl_load()) return PFR_ERR_UNSUPPORTED;
  PfrComm* c = reinterpret_cast<PfrComm*>(comm);
  (void)hipStreamSynchronize(c->own);
  const int rc = g_rccl.destroy(c->nccl);
  (void)hipStreamDestroy(c->own);
  (void)hipEventDestroy(c->before);
  (void)hipEventDestroy(c->after);
  delete c;
  if (rc) return rccl_fail("ncclCommDestroy", rc);
  return PFR_OK;
}
