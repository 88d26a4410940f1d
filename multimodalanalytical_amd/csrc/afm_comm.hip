// Data-parallel gradient exchange behind the C ABI (afm_allreduce_bucket): the ONE collective of the training path,
// all-reduce(sum) of a bucket of the flat fp32 gradient buffer over RCCL (xGMI inside a node), enqueued on the caller's stream.
// librccl.so is resolved at run time with dlopen (the copy already mapped by the host process -- torch ships its own -- is
// preferred, so the process keeps ONE RCCL), no link-time dependency: a single-GPU user never needs the library.
#include "afm_common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <mutex>
#include <string.h>

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
  ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
  ncclResult_t (*comm_count)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*comm_user_rank)(const ncclComm_t, int*) = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
  const char* names[] = {getenv("AFM_RCCL_PATH"), "librccl.so", "librccl.so.1"};
  for (int pass = 0; pass < 2 && !g_rccl.h; ++pass)      // pass 0: a copy that is already mapped; pass 1: load one
    for (const char* n : names) {
      if (!n || !*n) continue;
      g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (g_rccl.h) break;
    }
  if (!g_rccl.h) return;
  g_rccl.get_unique_id = (decltype(g_rccl.get_unique_id))dlsym(g_rccl.h, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (decltype(g_rccl.comm_init_rank))dlsym(g_rccl.h, "ncclCommInitRank");
  g_rccl.all_reduce = (decltype(g_rccl.all_reduce))dlsym(g_rccl.h, "ncclAllReduce");
  g_rccl.comm_destroy = (decltype(g_rccl.comm_destroy))dlsym(g_rccl.h, "ncclCommDestroy");
  g_rccl.comm_count = (decltype(g_rccl.comm_count))dlsym(g_rccl.h, "ncclCommCount");
  g_rccl.comm_user_rank = (decltype(g_rccl.comm_user_rank))dlsym(g_rccl.h, "ncclCommUserRank");
  g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.all_reduce && g_rccl.comm_destroy && g_rccl.comm_count &&
              g_rccl.comm_user_rank;
}
bool rccl() {
  std::call_once(g_once, load_rccl);
  return g_rccl.ok;
}
}  // namespace

struct afm_comm {
  ncclComm_t comm;
  int rank, world;
};

extern "C" int afm_comm_unique_id(void* out128) {
  if (!out128) return AFM_ERR_ARG;
  if (!rccl()) return AFM_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (g_rccl.get_unique_id(&id) != ncclSuccess) return AFM_ERR_LAUNCH;
  memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return AFM_OK;
}

extern "C" int afm_comm_create(afm_comm** out, const void* id128, int32_t rank, int32_t world) {
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return AFM_ERR_ARG;
  if (!rccl()) return AFM_ERR_UNSUPPORTED;
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t c;
  if (g_rccl.comm_init_rank(&c, world, id, rank) != ncclSuccess) return AFM_ERR_LAUNCH;
  *out = new afm_comm{c, rank, world};
  return AFM_OK;
}

extern "C" int afm_allreduce_bucket(afm_comm* c, float* buf, int64_t n, void* stream) {
  if (!c || !buf || n < 0) return AFM_ERR_ARG;
  if (n == 0) return AFM_OK;
  if (g_rccl.all_reduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream) != ncclSuccess) return AFM_ERR_LAUNCH;
  return AFM_OK;
}

extern "C" int afm_comm_count(afm_comm* c, int32_t* rank, int32_t* world) {
  if (!c || !rank || !world) return AFM_ERR_ARG;
  int r = -1, w = -1;
  if (g_rccl.comm_user_rank(c->comm, &r) != ncclSuccess || g_rccl.comm_count(c->comm, &w) != ncclSuccess) return AFM_ERR_LAUNCH;
  *rank = r; *world = w;
  return AFM_OK;
}

extern "C" int afm_comm_destroy(afm_comm* c) {
  if (!c) return AFM_ERR_ARG;
  const ncclResult_t r = g_rccl.comm_destroy(c->comm);
  delete c;
  return r == ncclSuccess ? AFM_OK : AFM_ERR_LAUNCH;
}
