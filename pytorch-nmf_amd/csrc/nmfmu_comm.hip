// nmfmu_comm_*: the collective of the column-sharded path as C entry points over RCCL.
//
// The H half-step of a sharded fit sums one packed fp32 buffer [numerators | denominators] over the ranks
// (engine.py: DenseMU.h_step).  The Python host side does that through torch.distributed (backend "nccl" = RCCL); a host
// written in another language binds these entries instead.  RCCL is resolved at run time (dlopen of librccl.so.1 -- the
// copy already loaded into the process if there is one, e.g. PyTorch's), so libnmfmu.so has no link-time dependency on
// it and single-GPU users never load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <new>

#include "../../include/nmfmu.h"

namespace {

// the slice of rccl.h these entries need (ABI-stable since NCCL 2)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
constexpr int kNcclFloat = 7, kNcclSum = 0;

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*CommCount)(const ncclComm_t, int*) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names)
      if (!x.handle) x.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);   // a copy the process already has (PyTorch's)
    for (const char* n : names)
      if (!x.handle) x.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!x.handle) x.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!x.handle) return x;
    auto sym = [&](const char* s) { return dlsym(x.handle, s); };
    x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
    x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
    x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(sym("ncclCommInitAll"));
    x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(sym("ncclAllReduce"));
    x.CommCount = reinterpret_cast<decltype(x.CommCount)>(sym("ncclCommCount"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
    x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
    x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
    x.ok = x.GetUniqueId && x.CommInitRank && x.CommInitAll && x.AllReduce && x.CommCount && x.CommDestroy && x.GroupStart &&
           x.GroupEnd;
    return x;
  }();
  return r;
}

// ncclResult_t -> this library's convention (> 0 would read as a hipError_t): RCCL failures become 10000 + code
inline int rc(int nccl_result) { return nccl_result == 0 ? NMFMU_OK : 10000 + nccl_result; }

}  // namespace

struct nmfmu_comm {
  ncclComm_t comm;
  int nranks;
};

extern "C" {

int nmfmu_comm_available(void) { return rccl().ok ? 1 : 0; }

int nmfmu_comm_unique_id(void* id128) {
  if (!id128) return NMFMU_ERR_ARG;
  if (!rccl().ok) return NMFMU_ERR_UNSUPPORTED;
  ncclUniqueId id;
  const int e = rccl().GetUniqueId(&id);
  if (e) return rc(e);
  memcpy(id128, id.internal, sizeof(id.internal));
  return NMFMU_OK;
}

int nmfmu_comm_init_rank(nmfmu_comm** comm, int nranks, const void* id128, int rank) {
  if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return NMFMU_ERR_ARG;
  if (!rccl().ok) return NMFMU_ERR_UNSUPPORTED;
  ncclUniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  ncclComm_t c = nullptr;
  const int e = rccl().CommInitRank(&c, nranks, id, rank);   // binds to the calling thread's current device
  if (e) return rc(e);
  nmfmu_comm* out = new (std::nothrow) nmfmu_comm{c, nranks};
  if (!out) {                       // do not leak the communicator RCCL has just built
    rccl().CommDestroy(c);
    return NMFMU_ERR_ALLOC;
  }
  *comm = out;
  return NMFMU_OK;
}

int nmfmu_comm_init_all(nmfmu_comm** comms, int ndev, const int* devices) {
  if (!comms || ndev < 1 || ndev > 64) return NMFMU_ERR_ARG;
  if (!rccl().ok) return NMFMU_ERR_UNSUPPORTED;
  ncclComm_t cs[64];
  const int e = rccl().CommInitAll(cs, ndev, devices);
  if (e) return rc(e);
  for (int i = 0; i < ndev; ++i) comms[i] = new (std::nothrow) nmfmu_comm{cs[i], ndev};
  for (int i = 0; i < ndev; ++i) {
    if (comms[i]) continue;
    // an allocation failed: hand back nothing -- destroy every communicator (wrapped or not) and clear the output array
    for (int k = 0; k < ndev; ++k) {
      rccl().CommDestroy(cs[k]);
      delete comms[k];
      comms[k] = nullptr;
    }
    return NMFMU_ERR_ALLOC;
  }
  return NMFMU_OK;
}

int nmfmu_comm_nranks(const nmfmu_comm* comm) {
  if (!comm) return NMFMU_ERR_ARG;
  if (!rccl().ok) return NMFMU_ERR_UNSUPPORTED;
  int n = 0;
  const int e = rccl().CommCount(comm->comm, &n);
  return e ? -rc(e) : n;
}

int nmfmu_comm_allreduce_sum_f32(nmfmu_comm* comm, float* buf, size_t count, void* stream) {
  if (!comm || !buf) return NMFMU_ERR_ARG;
  if (!rccl().ok) return NMFMU_ERR_UNSUPPORTED;
  if (count == 0) return NMFMU_OK;
  return rc(rccl().AllReduce(buf, buf, count, kNcclFloat, kNcclSum, comm->comm, reinterpret_cast<hipStream_t>(stream)));
}

int nmfmu_comm_allreduce_sum_f32_multi(nmfmu_comm* const* comms, float* const* bufs, size_t count, void* const* streams,
                                       int ndev) {
  if (!comms || !bufs || !streams || ndev < 1) return NMFMU_ERR_ARG;
  if (!rccl().ok) return NMFMU_ERR_UNSUPPORTED;
  int e = rccl().GroupStart();
  for (int i = 0; i < ndev && !e; ++i) {
    if (!comms[i] || !bufs[i]) { e = -1; break; }
    e = rccl().AllReduce(bufs[i], bufs[i], count, kNcclFloat, kNcclSum, comms[i]->comm, reinterpret_cast<hipStream_t>(streams[i]));
  }
  const int e2 = rccl().GroupEnd();
  if (e == -1) return NMFMU_ERR_ARG;
  return rc(e ? e : e2);
}

int nmfmu_comm_destroy(nmfmu_comm* comm) {
  if (!comm) return NMFMU_ERR_ARG;
  const int e = rccl().ok ? rccl().CommDestroy(comm->comm) : 0;
  delete comm;
  return rc(e);
}

}  // extern "C"
