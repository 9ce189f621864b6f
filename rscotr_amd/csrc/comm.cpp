// Gradient exchange on RCCL, called directly (include/rscotr.h: rscotr_comm_*).
//
// Replaces, for the data path of the exchange, torch DDP's bucket all-reduces (the reference wraps MTL in
// MMDistributedDataParallel: mtl/apis/train.py:37-46) — and c10d's ProcessGroupNCCL underneath them.  c10d leaves every
// asynchronous collective with a watchdog thread that polls the work's events; on this torch / ROCm pair that thread
// sometimes polls an event last recorded in a CAPTURING stream and aborts the process (hipErrorCapturedEvent: 2 of 8 runs of
// the overlapped exchange in round 4).  Here a bucket's all-reduce is one ncclAllReduce(ncclAvg) on the stream the caller
// names: the caller orders it against backward with its own events (fork / join inside a hipGraph capture), nothing polls
// anything, and torch.distributed stays what carries the control traffic (the unique id, plan hashes, the graph-or-eager
// agreement).
//
// RCCL is resolved at run time: the symbols of the instance that is already in the process (torch loads its bundled
// librccl.so) or, failing that, librccl.so.1 / librccl.so from the loader path — this library does not link against a
// second copy.
#include "common.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace {

typedef struct { char internal[128]; } NcclUniqueId;  // NCCL_UNIQUE_ID_BYTES
typedef void* NcclComm;
// ncclResult_t / ncclDataType_t / ncclRedOp_t are C enums passed as int: ncclSuccess = 0, ncclFloat32 = 7, ncclAvg = 4
typedef int (*GetUniqueId_t)(NcclUniqueId*);
typedef int (*CommInitRank_t)(NcclComm*, int, NcclUniqueId, int);
typedef int (*AllReduce_t)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*CommDestroy_t)(NcclComm);
typedef const char* (*GetErrorString_t)(int);

struct Rccl {
  GetUniqueId_t get_unique_id = nullptr;
  CommInitRank_t comm_init_rank = nullptr;
  AllReduce_t all_reduce = nullptr;
  CommDestroy_t comm_destroy = nullptr;
  GetErrorString_t error_string = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclAllReduce")) {
      h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) return;
    }
    r.get_unique_id = reinterpret_cast<GetUniqueId_t>(dlsym(h, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<CommInitRank_t>(dlsym(h, "ncclCommInitRank"));
    r.all_reduce = reinterpret_cast<AllReduce_t>(dlsym(h, "ncclAllReduce"));
    r.comm_destroy = reinterpret_cast<CommDestroy_t>(dlsym(h, "ncclCommDestroy"));
    r.error_string = reinterpret_cast<GetErrorString_t>(dlsym(h, "ncclGetErrorString"));
    r.ok = r.get_unique_id && r.comm_init_rank && r.all_reduce && r.comm_destroy;
  });
  return r;
}

int nccl_fail(const char* what, int rc) {
  Rccl& r = rccl();
  return rscotr::fail(RSCOTR_E_LAUNCH, "%s: RCCL error %d (%s)", what, rc, r.error_string ? r.error_string(rc) : "?");
}

}  // namespace

// 1 if an RCCL instance could be resolved (its symbols already in the process, or librccl.so on the loader path).
extern "C" int rscotr_comm_available(void) { return rccl().ok ? 1 : 0; }

// id: 128 bytes written by ONE rank and handed to all ranks (by whatever carries the job's control traffic).
extern "C" int rscotr_comm_unique_id(void* id128) {
  if (!id128) return rscotr::fail(RSCOTR_E_ARG, "rscotr_comm_unique_id: null pointer");
  if (!rccl().ok) return rscotr::fail(RSCOTR_E_ARCH, "rscotr_comm_unique_id: no RCCL in this process (librccl.so not found)");
  NcclUniqueId id;
  if (int rc = rccl().get_unique_id(&id)) return nccl_fail("ncclGetUniqueId", rc);
  memcpy(id128, id.internal, 128);
  return RSCOTR_OK;
}

// Collective over all ranks: every rank calls it with the same id, its rank and the rank count, with ITS device current.
extern "C" int rscotr_comm_init(const void* id128, int rank, int nranks, void** comm_out) {
  if (!id128 || !comm_out) return rscotr::fail(RSCOTR_E_ARG, "rscotr_comm_init: null pointer");
  if (nranks < 1 || rank < 0 || rank >= nranks) return rscotr::fail(RSCOTR_E_ARG, "rscotr_comm_init: rank %d of %d", rank, nranks);
  if (!rccl().ok) return rscotr::fail(RSCOTR_E_ARCH, "rscotr_comm_init: no RCCL in this process (librccl.so not found)");
  NcclUniqueId id;
  memcpy(id.internal, id128, 128);
  NcclComm c = nullptr;
  if (int rc = rccl().comm_init_rank(&c, nranks, id, rank)) return nccl_fail("ncclCommInitRank", rc);
  *comm_out = c;
  return RSCOTR_OK;
}

// buf[0 .. count) = mean over the ranks of buf, in place, on `stream` (capturable; ordered like any launch on that stream).
extern "C" int rscotr_comm_allreduce_avg(void* comm, float* buf, int64_t count, void* stream) {
  if (count < 0) return rscotr::fail(RSCOTR_E_SHAPE, "rscotr_comm_allreduce_avg: negative count");
  if (count == 0) return RSCOTR_OK;
  if (!comm || !buf) return rscotr::fail(RSCOTR_E_ARG, "rscotr_comm_allreduce_avg: null pointer");
  if (int rc = rccl().all_reduce(buf, buf, (size_t)count, /*ncclFloat32*/ 7, /*ncclAvg*/ 4, comm, (hipStream_t)stream))
    return nccl_fail("ncclAllReduce", rc);
  return RSCOTR_OK;
}

extern "C" int rscotr_comm_destroy(void* comm) {
  if (!comm) return RSCOTR_OK;
  if (!rccl().ok) return rscotr::fail(RSCOTR_E_ARCH, "rscotr_comm_destroy: no RCCL in this process");
  if (int rc = rccl().comm_destroy(comm)) return nccl_fail("ncclCommDestroy", rc);
  return RSCOTR_OK;
}
