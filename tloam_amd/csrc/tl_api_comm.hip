// tl_api_comm.hip -- the multi-GPU exchange of the C ABI (include/tloam_hip.h): RCCL loaded at run time, the caller's own
// all-reduce, and the library's one-shot peer exchange over xGMI (mailbox); SURVEY 8(e).  One process per GPU; the sharded
// frame's only data-path exchange is the 48 doubles of a GN sweep (+ the cap prefix counts and the cost sums per outer iteration).
#include "tl_ctx.hpp"

using namespace tl;

struct Uid128 { char bytes[128]; };  // == ncclUniqueId (rccl.h: char internal[128]), passed BY VALUE

namespace {
// ---- RCCL, loaded at run time so the library also loads where librccl is absent ----------------
struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Uid128, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;       // (optional: what the communicator itself says its size / this rank is)
  int (*CommUserRank)(void*, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
bool load_rccl(std::string* err) {
  if (g_rccl.handle) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* nm : names) {
    h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);  // prefer the copy already in the process
    if (h) break;
  }
  if (!h)
    for (const char* nm : names) {
      h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
  if (!h) { if (err) *err = std::string("dlopen librccl: ") + dlerror(); return false; }
  g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void**, int, Uid128, int))dlsym(h, "ncclCommInitRank");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
  g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  g_rccl.CommCount = (int (*)(void*, int*))dlsym(h, "ncclCommCount");
  g_rccl.CommUserRank = (int (*)(void*, int*))dlsym(h, "ncclCommUserRank");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
    if (err) *err = "librccl: missing symbols";
    return false;
  }
  g_rccl.handle = h;
  return true;
}
constexpr int kNcclFloat64 = 8;  // ncclDataType_t ncclFloat64 (rccl.h)
constexpr int kNcclSum = 0;      // ncclRedOp_t ncclSum
}  // namespace

namespace tlh {
// sum all-reduce of a small device buffer of doubles across the ranks of this context
int allreduce(tloam_ctx* c, double* dev, int count) {
  if (!exchanging(c) || c->comm == COMM_NONE) return TLOAM_OK;
  if (c->comm == COMM_MAILBOX) {
    if (count > 64) { c->last_error = "mailbox exchange: more than 64 values"; return TLOAM_E_INVALID; }
    launch_mbox_allreduce(dev, count, c->mbox, c->stream);
    return TLOAM_OK;
  }
  if (c->comm == COMM_CALLBACK) {
    const int rc = c->cb(c->cb_user, dev, count, (void*)c->stream);
    if (rc != 0) { c->last_error = "allreduce callback failed"; return TLOAM_E_RCCL; }
    return TLOAM_OK;
  }
  const int rc = g_rccl.AllReduce(dev, dev, (size_t)count, kNcclFloat64, kNcclSum, c->nccl_comm, c->stream);
  if (rc != 0) {
    c->last_error = std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return TLOAM_E_RCCL;
  }
  return TLOAM_OK;
}

void comm_rccl_info(const tloam_ctx* c, int32_t* count, int32_t* user_rank) {
  int n = -1, r = -1;
  if (c->nccl_comm && g_rccl.CommCount && g_rccl.CommCount(c->nccl_comm, &n) != 0) n = -1;
  if (c->nccl_comm && g_rccl.CommUserRank && g_rccl.CommUserRank(c->nccl_comm, &r) != 0) r = -1;
  if (count) *count = n;
  if (user_rank) *user_rank = r;
}

// a sharded set-up that failed half way, or is being replaced: the context is a single rank again (a later sharded call must not
// launch exchange kernels over peers that are not mapped)
static void comm_reset(tloam_ctx* c) {
  for (int r = 0; r < kMaxRanks; ++r)
    if (c->mbox_opened[r]) { (void)hipIpcCloseMemHandle(c->mbox_opened[r]); c->mbox_opened[r] = nullptr; }
  memset(&c->mbox, 0, sizeof(c->mbox));
  c->comm = COMM_NONE;
  c->rank = 0;
  c->nranks = 1;
  c->loopback = false;
}

// what tloam_destroy releases of the exchange
void comm_release(tloam_ctx* c) {
  if (c->nccl_comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->nccl_comm);
  for (int r = 0; r < kMaxRanks; ++r)
    if (c->mbox_opened[r]) (void)hipIpcCloseMemHandle(c->mbox_opened[r]);
  if (c->mbox_local) (void)hipFree(c->mbox_local);
  c->mbox_ctr.release();
}
}  // namespace tlh

extern "C" {

// ---- multi-GPU -------------------------------------------------------------------------------------
int tloam_rccl_unique_id(void* out128) {
  if (!out128) return TLOAM_E_INVALID;
  std::string err;
  if (!load_rccl(&err)) return TLOAM_E_RCCL;
  Uid128 id;
  memset(&id, 0, sizeof(id));
  if (g_rccl.GetUniqueId(&id) != 0) return TLOAM_E_RCCL;
  memcpy(out128, &id, sizeof(id));
  return TLOAM_OK;
}

int tloam_comm_init_rccl(tloam_ctx* c, int rank, int nranks, const void* unique_id128) {
  if (!c || !unique_id128 || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  if (!load_rccl(&c->last_error)) return TLOAM_E_RCCL;
  Uid128 id;
  memcpy(&id, unique_id128, sizeof(id));
  void* comm = nullptr;
  const int rc = g_rccl.CommInitRank(&comm, nranks, id, rank);
  if (rc != 0) {
    c->last_error = std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return TLOAM_E_RCCL;
  }
  if (c->nccl_comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->nccl_comm);   // (a second set-up on the same context)
  c->nccl_comm = comm;
  c->rank = rank;
  c->nranks = nranks;
  c->comm = COMM_RCCL;
  c->loopback = nranks == 1;   // a one-rank communicator still carries the all-reduce (tl_ctx.hpp)
  return TLOAM_OK;
}

int tloam_comm_init_callback(tloam_ctx* c, int rank, int nranks, tloam_allreduce_fn fn, void* user) {
  if (!c || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks || (nranks > 1 && !fn)) return TLOAM_E_INVALID;
  c->rank = rank;
  c->nranks = nranks;
  c->cb = fn;
  c->cb_user = user;
  c->comm = COMM_CALLBACK;
  c->loopback = false;
  return TLOAM_OK;
}

// (c) one-shot peer exchange over xGMI, no collective library on the data path: every rank exports a small
//     fine-grained buffer through HIP IPC, maps its peers', and from then on a sharded GN iteration is the sweep
//     (its last block stores the 48 doubles into every rank's buffer) and the step (adds them in rank order).
static_assert(sizeof(hipIpcMemHandle_t) == 64, "tloam_comm_mailbox_export hands out 64 bytes");
int tloam_comm_mailbox_export(tloam_ctx* c, void* handle64) {
  if (!c || !handle64) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  if (!c->mbox_local) {
    const size_t bytes = sizeof(double) * kMboxDoubles;
    void* p = nullptr;
    // uncached fine-grained device memory: peers' stores land in memory, local polls read memory
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    }
    HIPC(c, e);
    HIPC(c, hipMemset(p, 0, bytes));
    c->mbox_local = (double*)p;
  } else {
    // A second set-up on this context.  The export is the first step of a set-up and every rank takes it before
    // any rank can finish tloam_comm_init_mailbox (the launcher all-gathers the handles in between), so no exchange of the NEW
    // session can have been posted here yet; the exchanges of the OLD one are over on every rank once its last sharded call has
    // returned everywhere, which a caller that sets up again has seen.  What is in the buffer now are the old session's rows with
    // their exchange ids -- and the new session counts its ids from 1 again: left in place, new exchange 1 or 2 would find a
    // matching id and add stale rows without waiting.  So: drain this context's own work, then clear the buffer.
    HIPC(c, hipStreamSynchronize(c->stream));
    HIPC(c, hipMemset(c->mbox_local, 0, sizeof(double) * kMboxDoubles));
  }
  hipIpcMemHandle_t h;
  HIPC(c, hipIpcGetMemHandle(&h, c->mbox_local));
  memcpy(handle64, &h, sizeof(h));
  return TLOAM_OK;
}

int tloam_comm_init_mailbox(tloam_ctx* c, int rank, int nranks, const void* handles64) {
  if (!c || !handles64 || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return TLOAM_E_INVALID;
  if (!c->mbox_local) return TLOAM_E_NOT_READY;  // export first
  HIPC(c, hipSetDevice(c->device));
  comm_reset(c);   // (a second set-up on the same context: the first one's mappings go; until this one is complete the context is one rank)
  for (int r = 0; r < nranks; ++r) {
    if (r == rank) { c->mbox.peer[r] = c->mbox_local; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles64 + 64 * (size_t)r, sizeof(h));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      c->last_error = std::string("hipIpcOpenMemHandle(rank ") + std::to_string(r) + "): " + hipGetErrorString(e);
      (void)hipGetLastError();
      comm_reset(c);   // the handles opened so far are closed, the context stays a single rank
      return TLOAM_E_RCCL;
    }
    c->mbox_opened[r] = p;
    c->mbox.peer[r] = (double*)p;
  }
  {
    hipError_t e = c->mbox_ctr.reserve(4);
    if (e == hipSuccess) e = hipMemset(c->mbox_ctr.p, 0, 4 * sizeof(unsigned long long));
    if (e != hipSuccess) {
      c->last_error = std::string("mailbox exchange counter: ") + hipGetErrorString(e);
      comm_reset(c);
      return TLOAM_E_HIP;
    }
  }
  c->mbox.ctr = c->mbox_ctr.p;
  c->mbox.rank = rank;
  c->mbox.nranks = nranks;
  c->rank = rank;
  c->nranks = nranks;
  c->comm = COMM_MAILBOX;
  c->loopback = nranks == 1;   // the rank posts to, and gathers from, its own buffer (tl_ctx.hpp)
  return TLOAM_OK;
}

}  // extern "C"
