// rccl_dyn.cpp — RCCL bound at run time (dlopen), so a single-GPU process never
// needs librccl and a multi-GPU process shares the one copy already loaded by
// its launcher (torch.distributed loads the same SONAME, librccl.so.1).
//
// The reference has no collective at all (ROS2 DDS only); this is
// the build's N5 row: one exchange of MI355_ICP_NSUMS scalars per job per ICP
// iteration over xGMI - an ncclAllGather of the per-rank rows, which comm.hip then
// adds in rank order (bit-identical on every rank, whatever algorithm RCCL picks) -
// plus the exact integer ncclAllReduce calls of slab-sharded reference mode.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "icp_internal.h"

namespace mi355 {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static_assert(sizeof(ncclUniqueId) == MI355_ICP_UNIQUE_ID_BYTES, "ncclUniqueId size");

static RcclApi g_api;
static const char* g_api_err = nullptr;
static std::once_flag g_once;

static void load_api() {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (const char* n : names) {
    lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  if (!lib) {
    g_api_err = "RCCL not loadable (librccl.so.1 not found)";
    return;
  }
  g_api.lib = lib;
  g_api.GetUniqueId = reinterpret_cast<decltype(g_api.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  g_api.CommInitRank = reinterpret_cast<decltype(g_api.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  g_api.AllReduce = reinterpret_cast<decltype(g_api.AllReduce)>(dlsym(lib, "ncclAllReduce"));
  g_api.AllGather = reinterpret_cast<decltype(g_api.AllGather)>(dlsym(lib, "ncclAllGather"));
  g_api.CommCount = reinterpret_cast<decltype(g_api.CommCount)>(dlsym(lib, "ncclCommCount"));
  g_api.CommUserRank = reinterpret_cast<decltype(g_api.CommUserRank)>(dlsym(lib, "ncclCommUserRank"));
  g_api.CommDestroy = reinterpret_cast<decltype(g_api.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  g_api.GetErrorString = reinterpret_cast<decltype(g_api.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  if (!g_api.GetUniqueId || !g_api.CommInitRank || !g_api.AllReduce || !g_api.AllGather || !g_api.CommCount || !g_api.CommUserRank ||
      !g_api.CommDestroy || !g_api.GetErrorString) {
    g_api_err = "RCCL loaded but a required symbol is missing";
    g_api.lib = nullptr;
  }
}

const RcclApi* rccl_api(const char** why_not) {
  std::call_once(g_once, load_api);
  if (!g_api.lib) {
    if (why_not) *why_not = g_api_err ? g_api_err : "RCCL unavailable";
    return nullptr;
  }
  return &g_api;
}

int rccl_unique_id(unsigned char* out128, const char** err) {
  const RcclApi* api = rccl_api(err);
  if (!api) return MI355_ICP_ERR_COMM;
  ncclUniqueId id;
  ncclResult_t r = api->GetUniqueId(&id);
  if (r != ncclSuccess) {
    if (err) *err = api->GetErrorString(r);
    return MI355_ICP_ERR_COMM;
  }
  std::memcpy(out128, &id, sizeof(id));
  return MI355_ICP_OK;
}

int rccl_comm_init(void** comm, const unsigned char* id128, int rank, int n_ranks, const char** err) {
  const RcclApi* api = rccl_api(err);
  if (!api) return MI355_ICP_ERR_COMM;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  ncclResult_t r = api->CommInitRank(&c, n_ranks, id, rank);
  if (r != ncclSuccess) {
    if (err) *err = api->GetErrorString(r);
    return MI355_ICP_ERR_COMM;
  }
  *comm = c;
  return MI355_ICP_OK;
}

int rccl_comm_query(void* comm, int* rank, int* n_ranks, const char** err) {
  const RcclApi* api = rccl_api(err);
  if (!api) return MI355_ICP_ERR_COMM;
  ncclResult_t r = ncclSuccess;
  if (rank) r = api->CommUserRank(static_cast<ncclComm_t>(comm), rank);
  if (r == ncclSuccess && n_ranks) r = api->CommCount(static_cast<ncclComm_t>(comm), n_ranks);
  if (r != ncclSuccess) {
    if (err) *err = api->GetErrorString(r);
    return MI355_ICP_ERR_COMM;
  }
  return MI355_ICP_OK;
}

static int done(const RcclApi* api, ncclResult_t r, const char** err) {
  if (r == ncclSuccess) return MI355_ICP_OK;
  if (err) *err = api->GetErrorString(r);
  return MI355_ICP_ERR_COMM;
}

int rccl_allgather_bytes(void* comm, const void* dev_send, void* dev_recv, size_t bytes_per_rank, void* stream, const char** err) {
  const RcclApi* api = rccl_api(err);
  if (!api) return MI355_ICP_ERR_COMM;
  return done(api, api->AllGather(dev_send, dev_recv, bytes_per_rank, ncclUint8, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)), err);
}

int rccl_allreduce_u32_sum(void* comm, uint32_t* dev_buf, size_t count, void* stream, const char** err) {
  const RcclApi* api = rccl_api(err);
  if (!api) return MI355_ICP_ERR_COMM;
  return done(api, api->AllReduce(dev_buf, dev_buf, count, ncclUint32, ncclSum, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)), err);
}

int rccl_allreduce_u64_min(void* comm, unsigned long long* dev_buf, size_t count, void* stream, const char** err) {
  const RcclApi* api = rccl_api(err);
  if (!api) return MI355_ICP_ERR_COMM;
  return done(api, api->AllReduce(dev_buf, dev_buf, count, ncclUint64, ncclMin, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)), err);
}

void rccl_comm_destroy(void* comm) {
  const RcclApi* api = rccl_api(nullptr);
  if (api && comm) api->CommDestroy(static_cast<ncclComm_t>(comm));
}

}  // namespace mi355
