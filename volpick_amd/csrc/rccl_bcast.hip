// Multi-GPU bring-up of the path: ONE collective, the start-up broadcast of the flat fp32
// weight blob (1.08 MB PhaseNet / 1.52 MB EQTransformer) from the root rank over RCCL (xGMI inside a node).
// The payload is latency-bound (~10 us of one 153 GB/s link), so a single flat ncclBroadcast is the right call;
// windows are then partitioned with no further communication.
//
// RCCL is bound at run time (dlopen of librccl.so.1) so that the library loads on hosts without it and so that a
// process that already carries an RCCL (PyTorch-ROCm bundles one with the same SONAME) keeps exactly one copy.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

#include "vp_common.h"

namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

std::string g_load_error;  // why the binding failed (first dlopen message / missing symbol), set once

// Binds RCCL once (std::call_once: the first callers may come from several threads).
Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.so) break;
      const char* e = dlerror();  // read once: dlerror() clears the message
      if (g_load_error.empty()) g_load_error = e ? e : "dlopen failed";
    }
    if (!r.so) return;
    g_load_error.clear();
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.so, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.so, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.so, "ncclCommDestroy"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(r.so, "ncclBroadcast"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.so, "ncclCommCount"));
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.so, "ncclCommUserRank"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.so, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Broadcast || !r.GetErrorString || !r.CommCount || !r.CommUserRank) {
      g_load_error = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclBroadcast / ncclGetErrorString / "
                     "ncclCommCount / ncclCommUserRank";
      dlclose(r.so);
      r.so = nullptr;
    }
  });
  return r.so ? &r : nullptr;
}

#define VP_NEED_RCCL(r) VP_REQUIRE((r) != nullptr, "RCCL not available: %s", g_load_error.c_str())

#define VP_RCCL(R, call)                                                                \
  do {                                                                                  \
    ncclResult_t e_ = (call);                                                           \
    if (e_ != ncclSuccess) {                                                            \
      vp::set_error("%s failed: %s (%s:%d)", #call, (R)->GetErrorString(e_), __FILE__, __LINE__); \
      return VP_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)

}  // namespace

extern "C" {

int vp_rccl_available(void) {
  if (rccl() != nullptr) return 1;
  vp::set_error("RCCL not available: %s", g_load_error.c_str());
  return 0;
}

int vp_rccl_unique_id(void* id128) {
  VP_REQUIRE(id128 != nullptr, "null id buffer");
  Rccl* r = rccl();
  VP_NEED_RCCL(r);
  ncclUniqueId id;
  VP_RCCL(r, r->GetUniqueId(&id));
  static_assert(sizeof(id) == VP_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
  memcpy(id128, &id, sizeof(id));
  return VP_OK;
}

int vp_rccl_comm_init(int device_id, int n_ranks, const void* id128, int rank, void** comm) {
  VP_REQUIRE(id128 && comm, "null argument");
  VP_REQUIRE(n_ranks > 0 && rank >= 0 && rank < n_ranks, "rank %d outside [0, %d)", rank, n_ranks);
  Rccl* r = rccl();
  VP_NEED_RCCL(r);
  VP_HIP(hipSetDevice(device_id));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  VP_RCCL(r, r->CommInitRank(&c, n_ranks, id, rank));
  *comm = c;
  return VP_OK;
}

// What RCCL itself says about the communicator: the number of ranks it spans and this process's rank in it.
int vp_rccl_comm_info(void* comm, int* n_ranks, int* rank) {
  VP_REQUIRE(comm && n_ranks && rank, "null argument");
  Rccl* r = rccl();
  VP_NEED_RCCL(r);
  VP_RCCL(r, r->CommCount(static_cast<ncclComm_t>(comm), n_ranks));
  VP_RCCL(r, r->CommUserRank(static_cast<ncclComm_t>(comm), rank));
  return VP_OK;
}

int vp_rccl_comm_destroy(void* comm) {
  if (!comm) return VP_OK;
  Rccl* r = rccl();
  VP_NEED_RCCL(r);
  VP_RCCL(r, r->CommDestroy(static_cast<ncclComm_t>(comm)));
  return VP_OK;
}

int vp_rccl_library_path(char* buf, size_t cap) {
  VP_REQUIRE(buf != nullptr && cap > 0, "null / empty buffer");
  buf[0] = 0;
  Rccl* r = rccl();
  if (!r) {
    vp::set_error("RCCL not available: %s", g_load_error.c_str());
    return VP_ERR_UNSUPPORTED;
  }
  Dl_info info;
  if (!dladdr(reinterpret_cast<void*>(r->Broadcast), &info) || !info.dli_fname) {
    vp::set_error("dladdr(ncclBroadcast) found no object");
    return VP_ERR_UNSUPPORTED;
  }
  snprintf(buf, cap, "%s", info.dli_fname);
  return VP_OK;
}

// In place on every rank: the root sends weights_dev, the others receive into it.  Runs on a stream of its own on the
// calling thread's current device and returns when the data has arrived (start-up path, once per model).
int vp_bcast_weights(void* rccl_comm, float* weights_dev, size_t n_floats, int root) {
  VP_REQUIRE(rccl_comm && weights_dev && n_floats > 0, "null / empty argument");
  Rccl* r = rccl();
  VP_NEED_RCCL(r);
  hipStream_t s = nullptr;
  VP_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  ncclResult_t e = r->Broadcast(weights_dev, weights_dev, n_floats, ncclFloat32, root, static_cast<ncclComm_t>(rccl_comm), s);
  hipError_t he = hipStreamSynchronize(s);
  (void)hipStreamDestroy(s);
  if (e != ncclSuccess) {
    vp::set_error("ncclBroadcast failed: %s", r->GetErrorString(e));
    return VP_ERR_HIP;
  }
  VP_HIP(he);
  return VP_OK;
}

}  // extern "C"
