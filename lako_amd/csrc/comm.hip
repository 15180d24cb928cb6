// Data-parallel communication behind the C-ABI (SURVEY.md §8 b2 / e; the reference's scaffolding: src/slurm.py:157-160 init_process_group,
// src/util.py:248-275 the reductions it performs): thin calls into RCCL on the CALLER's stream — rendezvous id, communicator, in-place SUM
// all-reduce of a gradient buffer over xGMI — so that a host that is not Python can drive the gradient all-reduce of the data-parallel
// reader through include/lako_hip.h alone.  (The Python host keeps using torch.distributed, backend "nccl" = the same RCCL: lako_amd/dist.py.)
// RCCL is resolved at the first call (dlopen of the library the process already has — torch ships one — or of the system's): liblako_hip.so
// itself carries no link-time dependency on it and loads where RCCL is absent; the entry points then return LAKO_E_UNSUPPORTED.
// No global state: the communicator is a caller-owned handle.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "common.h"

namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  char why[256] = "";      // the loader's message, captured where it happened (dlerror() clears itself when read)
};

// resolved once per process (function pointers of a shared library: not tuning state, nothing a caller could want two of)
const Rccl& rccl() {
  static const Rccl r = [] {
    Rccl x;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      x.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (x.h) break;
      const char* e = dlerror();
      snprintf(x.why, sizeof(x.why), "%s", e ? e : "dlopen failed");
    }
    if (!x.h) return x;
    x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(x.h, "ncclGetUniqueId"));
    x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(x.h, "ncclCommInitRank"));
    x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(x.h, "ncclAllReduce"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.h, "ncclCommDestroy"));
    x.CommCount = reinterpret_cast<decltype(x.CommCount)>(dlsym(x.h, "ncclCommCount"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.h, "ncclGetErrorString"));
    x.ok = x.GetUniqueId && x.CommInitRank && x.AllReduce && x.CommDestroy && x.GetErrorString;
    if (!x.ok) snprintf(x.why, sizeof(x.why), "the library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy / ncclGetErrorString");
    return x;
  }();
  return r;
}

int need_rccl(const char* fn) {
  if (rccl().ok) return 0;
  lako_set_error("%s: RCCL (librccl.so.1) is not available in this process: %s", fn, rccl().why);
  return LAKO_E_UNSUPPORTED;
}

int check(ncclResult_t r, const char* fn, const char* what) {
  if (r == ncclSuccess) return LAKO_OK;
  lako_set_error("%s: %s failed: %s", fn, what, rccl().GetErrorString(r));
  return LAKO_E_LAUNCH;
}

}  // namespace

struct lako_comm {
  ncclComm_t comm;
  int rank, world;
};

extern "C" int lako_comm_unique_id(uint8_t id[LAKO_COMM_ID_BYTES]) {
  LAKO_CHECK_ARG(id != nullptr, "lako_comm_unique_id: null id");
  if (int rc = need_rccl("lako_comm_unique_id")) return rc;
  static_assert(sizeof(ncclUniqueId) == LAKO_COMM_ID_BYTES, "rendezvous id size");
  ncclUniqueId u;
  if (int rc = check(rccl().GetUniqueId(&u), "lako_comm_unique_id", "ncclGetUniqueId")) return rc;
  memcpy(id, &u, sizeof(u));
  return LAKO_OK;
}

extern "C" int lako_comm_init(lako_comm_t** comm, int rank, int world, const uint8_t id[LAKO_COMM_ID_BYTES]) {
  LAKO_CHECK_ARG(comm != nullptr && id != nullptr, "lako_comm_init: null argument");
  LAKO_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "lako_comm_init: rank %d of %d", rank, world);
  *comm = nullptr;
  if (int rc = need_rccl("lako_comm_init")) return rc;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  lako_comm* c = new (std::nothrow) lako_comm{nullptr, rank, world};
  LAKO_CHECK_ARG(c != nullptr, "lako_comm_init: out of host memory");
  if (int rc = check(rccl().CommInitRank(&c->comm, world, u, rank), "lako_comm_init", "ncclCommInitRank")) {
    delete c;
    return rc;
  }
  *comm = c;
  return LAKO_OK;
}

extern "C" int lako_comm_world_size(const lako_comm_t* comm) {
  if (!comm) return LAKO_E_BADARG;
  int n = comm->world;
  if (rccl().ok && rccl().CommCount && rccl().CommCount(comm->comm, &n) != ncclSuccess) return LAKO_E_LAUNCH;
  return n;          // what RCCL itself reports (bench.py prints it beside WORLD_SIZE)
}

extern "C" int lako_allreduce(lako_comm_t* comm, void* buf, int64_t count, int dtype, lako_stream_t stream) {
  LAKO_CHECK_ARG(comm != nullptr && buf != nullptr && count >= 0, "lako_allreduce: bad argument");
  LAKO_CHECK_ARG(dtype == LAKO_F32 || dtype == LAKO_BF16, "lako_allreduce: dtype must be LAKO_F32 or LAKO_BF16");
  LAKO_CHECK_ALIGN(buf, 16);
  if (int rc = need_rccl("lako_allreduce")) return rc;
  if (count == 0) return LAKO_OK;
  // in place, SUM: the 1 / world factor is folded into lako_adamw_step's grad_scale (lako_amd/dist.py does the same)
  return check(rccl().AllReduce(buf, buf, (size_t)count, dtype == LAKO_F32 ? ncclFloat32 : ncclBfloat16, ncclSum, comm->comm, (hipStream_t)stream),
               "lako_allreduce", "ncclAllReduce");
}

extern "C" int lako_comm_destroy(lako_comm_t* comm) {
  if (!comm) return LAKO_OK;
  int rc = LAKO_OK;
  if (rccl().ok) rc = check(rccl().CommDestroy(comm->comm), "lako_comm_destroy", "ncclCommDestroy");
  delete comm;
  return rc;
}

// Caller scratch an entry point wants for the given arguments (0 = none).  Every kernel of this library works in caller buffers and
// accumulates straight into its outputs; the ONE optional scratch is the slab reduction of lako_gemm_tn_grouped (bit-reproducible weight
// gradients), whose own query this forwards to.
extern "C" int64_t lako_workspace_bytes(int op, const void* args) {
  if (op == LAKO_WS_GEMM_TN_GROUPED) {
    if (!args) return LAKO_E_BADARG;
    const lako_ws_gemm_tn_grouped_t* g = static_cast<const lako_ws_gemm_tn_grouped_t*>(args);
    return lako_gemm_tn_grouped_workspace(g->items, g->n_items, g->K, g->in_dtype, g->split_k, g->tuning);
  }
  return 0;
}
