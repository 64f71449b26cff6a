// tn_allreduce_grads: the data-parallel gradient exchange (pipelines/base_pipeline.py:281-283: torch DDP's mean all-reduce over NCCL) for a host that
// binds only this C ABI -- SURVEY.md 8b lists it among the exports.  RCCL is NOT a link-time dependency of the library: its entry points are
// resolved at the first call, from the RCCL the process has already loaded (PyTorch ships and loads its own librccl.so; two copies in one process
// would each keep their own state) or, in a process without one, from librccl.so on the loader path.  The Python package exchanges through
// torch.distributed (parallel.py), which is the same library underneath; these entry points exist so that a compiled trainer needs nothing else.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "tn_common.h"

namespace {
// the few RCCL types / constants used (rccl.h: NCCL_UNIQUE_ID_BYTES 128; ncclFloat32 = 7; ncclSum = 0, ncclAvg = 4)
struct UniqueId { char internal[128]; };
typedef int (*get_unique_id_t)(UniqueId*);
typedef int (*comm_init_rank_t)(void**, int, UniqueId, int);
typedef int (*comm_destroy_t)(void*);
typedef int (*all_reduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*get_error_string_t)(int);
struct Rccl {
  get_unique_id_t get_unique_id = nullptr;
  comm_init_rank_t comm_init_rank = nullptr;
  comm_destroy_t comm_destroy = nullptr;
  all_reduce_t all_reduce = nullptr;
  get_error_string_t error_string = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
const Rccl& rccl() {
  std::call_once(g_rccl_once, [] {
    void* h = RTLD_DEFAULT;  // the RCCL this process has loaded already (torch's), if any
    if (dlsym(h, "ncclAllReduce") == nullptr) {
      h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (h == nullptr) return;
    }
    g_rccl.get_unique_id = reinterpret_cast<get_unique_id_t>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.comm_init_rank = reinterpret_cast<comm_init_rank_t>(dlsym(h, "ncclCommInitRank"));
    g_rccl.comm_destroy = reinterpret_cast<comm_destroy_t>(dlsym(h, "ncclCommDestroy"));
    g_rccl.all_reduce = reinterpret_cast<all_reduce_t>(dlsym(h, "ncclAllReduce"));
    g_rccl.error_string = reinterpret_cast<get_error_string_t>(dlsym(h, "ncclGetErrorString"));
    g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.all_reduce;
  });
  return g_rccl;
}
int fail(const char* what, int rc) {
  const Rccl& r = rccl();
  tn_set_error("%s: RCCL error %d (%s)", what, rc, r.error_string ? r.error_string(rc) : "?");
  return TN_ELAUNCH;
}
}  // namespace

extern "C" int tn_comm_unique_id(void* unique_id_out) {
  TN_REQUIRE(unique_id_out != nullptr, "tn_comm_unique_id: null pointer");
  const Rccl& r = rccl();
  if (!r.ok) { tn_set_error("tn_comm_unique_id: no RCCL in this process and none on the loader path"); return TN_ELAUNCH; }
  const int rc = r.get_unique_id(reinterpret_cast<UniqueId*>(unique_id_out));
  return rc ? fail("tn_comm_unique_id", rc) : TN_OK;
}

extern "C" int tn_comm_create(const void* unique_id, int32_t world_size, int32_t rank, void** comm_out) {
  TN_REQUIRE(unique_id != nullptr && comm_out != nullptr, "tn_comm_create: null pointer");
  TN_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "tn_comm_create: bad world_size=%d rank=%d", world_size, rank);
  const Rccl& r = rccl();
  if (!r.ok) { tn_set_error("tn_comm_create: no RCCL in this process and none on the loader path"); return TN_ELAUNCH; }
  UniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  void* comm = nullptr;
  const int rc = r.comm_init_rank(&comm, world_size, id, rank);  // collective over the ranks; binds the CURRENT device
  if (rc) return fail("tn_comm_create", rc);
  *comm_out = comm;
  return TN_OK;
}

extern "C" int tn_comm_destroy(void* comm) {
  if (comm == nullptr) return TN_OK;
  const Rccl& r = rccl();
  if (!r.ok) { tn_set_error("tn_comm_destroy: no RCCL"); return TN_ELAUNCH; }
  const int rc = r.comm_destroy(comm);
  return rc ? fail("tn_comm_destroy", rc) : TN_OK;
}

extern "C" int tn_allreduce_grads(void* comm, float* grads, int64_t count, int32_t average, tn_stream_t stream) {
  TN_REQUIRE(comm != nullptr && grads != nullptr && count >= 0, "tn_allreduce_grads: bad argument");
  if (count == 0) return TN_OK;
  const Rccl& r = rccl();
  if (!r.ok) { tn_set_error("tn_allreduce_grads: no RCCL"); return TN_ELAUNCH; }
  const int rc = r.all_reduce(grads, grads, (size_t)count, /*ncclFloat32*/ 7, average ? /*ncclAvg*/ 4 : /*ncclSum*/ 0, comm, tn_s(stream));
  return rc ? fail("tn_allreduce_grads", rc) : TN_OK;
}
