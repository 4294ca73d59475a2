// sos_comm.hip -- RCCL (xGMI) communicator of the multi-GPU path: one process per GPU, every rank
// holds the same keyframes and its own shard of the points; per Gauss-Newton iteration there is exactly ONE
// all-reduce (the packed fp32 accumulator blocks) and one all-gather (the newest-frame energies behind
// frameEnergyTH), both enqueued by the library on the same stream as its kernels, so the exchange is part of the
// prefetched accumulate chain and needs no host round trip.
//
// librccl is bound at run time (dlopen) so that the library has no link-time dependency on it and shares the copy
// the host process already uses (e.g. the one PyTorch ships: pass its path to sos_rccl_load).
#include "sos_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;

int rccl_check(ncclResult_t r, const char *what) {
  if (r == ncclSuccess) return SOS_OK;
  fprintf(stderr, "[sos_slam_hip] %s -> %s\n", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
  return SOS_ERR_HIP;
}
}  // namespace

struct sos_comm {
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0, device = 0;
};

extern "C" int sos_rccl_load(const char *path) {
  if (g_rccl.lib) return SOS_OK;
  void *h = dlopen(path && path[0] ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    fprintf(stderr, "[sos_slam_hip] cannot load librccl: %s\n", dlerror());
    return SOS_ERR_STATE;
  }
#define BIND(field, sym)                                                    \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, sym));   \
  if (!g_rccl.field) { fprintf(stderr, "[sos_slam_hip] librccl lacks %s\n", sym); return SOS_ERR_STATE; }
  BIND(GetUniqueId, "ncclGetUniqueId")
  BIND(CommInitRank, "ncclCommInitRank")
  BIND(CommDestroy, "ncclCommDestroy")
  BIND(AllReduce, "ncclAllReduce")
  BIND(AllGather, "ncclAllGather")
  BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
  g_rccl.lib = h;
  return SOS_OK;
}

extern "C" int sos_rccl_unique_id(void *id128) {
  if (!id128) return SOS_ERR_ARG;
  if (!g_rccl.lib) return SOS_ERR_STATE;
  ncclUniqueId id;
  int rc = rccl_check(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  if (rc) return rc;
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id128, &id, sizeof(id));
  return SOS_OK;
}

extern "C" int sos_comm_create(const void *id128, int nranks, int rank, int device, sos_comm **out) {
  if (!id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) return SOS_ERR_ARG;
  if (!g_rccl.lib) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  sos_comm *c = new sos_comm();
  c->nranks = nranks; c->rank = rank; c->device = device;
  int rc = rccl_check(g_rccl.CommInitRank(&c->comm, nranks, id, rank), "ncclCommInitRank");
  if (rc) { delete c; return rc; }
  *out = c;
  return SOS_OK;
}

extern "C" int sos_comm_destroy(sos_comm *c) {
  if (!c) return SOS_OK;
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  delete c;
  return SOS_OK;
}

extern "C" int sos_comm_size(const sos_comm *c) { return c ? c->nranks : 1; }
extern "C" int sos_comm_rank(const sos_comm *c) { return c ? c->rank : 0; }

// internal (sos_common.h): collectives on the caller's stream
int sos_comm_allreduce_sum_f32(sos_comm *c, float *buf, size_t count, hipStream_t st) {
  return rccl_check(g_rccl.AllReduce(buf, buf, count, ncclFloat32, ncclSum, c->comm, st), "ncclAllReduce(sum,f32)");
}
int sos_comm_allreduce_max_i32(sos_comm *c, int *buf, size_t count, hipStream_t st) {
  return rccl_check(g_rccl.AllReduce(buf, buf, count, ncclInt32, ncclMax, c->comm, st), "ncclAllReduce(max,i32)");
}
int sos_comm_allreduce_sum_f64(sos_comm *c, double *buf, size_t count, hipStream_t st) {
  return rccl_check(g_rccl.AllReduce(buf, buf, count, ncclFloat64, ncclSum, c->comm, st), "ncclAllReduce(sum,f64)");
}
int sos_comm_allgather_f32(sos_comm *c, const float *send, float *recv, size_t sendcount, hipStream_t st) {
  return rccl_check(g_rccl.AllGather(send, recv, sendcount, ncclFloat32, c->comm, st), "ncclAllGather(f32)");
}
