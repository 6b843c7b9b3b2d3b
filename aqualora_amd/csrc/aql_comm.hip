// aql_comm_* -- the data-parallel exchange of the PPFT step as C-ABI entry points over RCCL.
//
// Reference behaviour replaced: accelerate's DDP wrap of the trainable modules (train/ppft_train.py:905-912: rank 0's parameters
// are broadcast at construction) and DDP's bucketed gradient all-reduce(mean) fired from accelerator.backward (:1058), plus the
// logged-loss gather (:1054).  One process per GPU; every call is enqueued on the CALLER's hipStream_t, so the trainer can
// (a) put a collective on a forked side stream under the remaining backward / weight-gradient kernels and (b) capture it into
// the step's hipGraph -- neither is possible through torch.distributed's ProcessGroupNCCL, which owns its stream.
//
// RCCL is bound at run time (dlopen "librccl.so.1"): a process that already loaded PyTorch-ROCm gets the very RCCL instance
// torch's own "nccl" backend uses (same SONAME), next to the same HIP runtime; nothing here links against torch.
// Host code only -- the collectives' kernels are RCCL's (xGMI rings/trees); there is no device code of ours in this file.
#include "aql_common.h"
#include <dlfcn.h>
#include "../../include/aqualora_abi.h"   // AQL_ABI_VERSION: the ONE definition (include/aqualora_hip.h includes the same file)

extern "C" int aql_abi_version(void) { return AQL_ABI_VERSION; }

// The handful of RCCL (NCCL API 2.x) types this file passes through, declared here so that the library builds without RCCL's
// headers: RCCL is a run-time dependency only (dlopen below).  Values from the published nccl.h ABI, which RCCL keeps.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
// plain ints, not one-enumerator enums: RCCL returns codes (and takes datatype / op values) outside a mirrored enum's value range, and
// a C++ compiler may assume a value of an enum type lies within it
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
enum : int { ncclSuccess = 0 };
enum : int { ncclUint8 = 1, ncclFloat32 = 7 };
enum : int { ncclSum = 0, ncclAvg = 4 };
}

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl g_rccl;

bool rccl_load() {
  if (g_rccl.ok) return true;
  if (!g_rccl.handle) {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (g_rccl.handle) break;
    }
  }
  if (!g_rccl.handle) {
    aql_set_error("aql_comm: librccl.so.1 not found (%s)", dlerror());
    return false;
  }
#define AQL_SYM(field, sym)                                                 \
  *(void**)(&g_rccl.field) = dlsym(g_rccl.handle, sym);                     \
  if (!g_rccl.field) {                                                      \
    aql_set_error("aql_comm: symbol %s missing from librccl", sym);         \
    return false;                                                           \
  }
  AQL_SYM(GetUniqueId, "ncclGetUniqueId")
  AQL_SYM(CommInitRank, "ncclCommInitRank")
  AQL_SYM(CommDestroy, "ncclCommDestroy")
  AQL_SYM(CommAbort, "ncclCommAbort")
  AQL_SYM(CommCount, "ncclCommCount")
  AQL_SYM(AllReduce, "ncclAllReduce")
  AQL_SYM(ReduceScatter, "ncclReduceScatter")
  AQL_SYM(AllGather, "ncclAllGather")
  AQL_SYM(Broadcast, "ncclBroadcast")
  AQL_SYM(GetErrorString, "ncclGetErrorString")
#undef AQL_SYM
  g_rccl.ok = true;
  return true;
}

int rccl_status(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return AQL_OK;
  aql_set_error("%s: RCCL error %d (%s)", what, (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return AQL_ERR_HIP;
}

}  // namespace

// 1 when RCCL and every entry point used here resolve in this process, else 0 (reason in aql_last_error).  A count-like answer,
// not a status: the host side lets all ranks agree on it BEFORE anyone enters the collective aql_comm_init, so that a rank whose
// dlopen fails cannot leave the others blocked inside ncclCommInitRank.
extern "C" int aql_comm_available(void) { return rccl_load() ? 1 : 0; }

// 128 opaque bytes that identify one communicator; rank 0 creates them, the host side hands them to every rank (any side
// channel: the launcher's TCP store, an MPI broadcast ...), then every rank calls aql_comm_init with the same bytes.
extern "C" int aql_comm_unique_id(void* id128) {
  AQL_CHECK_ARG(id128 != nullptr, "aql_comm_unique_id: null output");
  if (!rccl_load()) return AQL_ERR_HIP;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  return rccl_status(g_rccl.GetUniqueId((ncclUniqueId*)id128), "ncclGetUniqueId");
}

// Collective over all ranks: builds the communicator of the CURRENT HIP device (one process per GPU).
extern "C" int aql_comm_init(const void* id128, int nranks, int rank, void** comm_out) {
  AQL_CHECK_ARG(id128 != nullptr && comm_out != nullptr, "aql_comm_init: null argument");
  AQL_CHECK_ARG(nranks >= 1 && rank >= 0 && rank < nranks, "aql_comm_init: rank %d of %d", rank, nranks);
  if (!rccl_load()) return AQL_ERR_HIP;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  const int rc = rccl_status(g_rccl.CommInitRank(&comm, nranks, id, rank), "ncclCommInitRank");
  *comm_out = (void*)comm;
  return rc;
}

extern "C" int aql_comm_size(void* comm) {   // number of ranks, or -1
  if (!comm || !rccl_load()) return -1;
  int n = -1;
  return g_rccl.CommCount((ncclComm_t)comm, &n) == ncclSuccess ? n : -1;
}

// buf[0..n) <- sum (average != 0: mean) over ranks, in place, fp32: the LoRA + mapper gradient exchange (ppft_train.py:1058).
extern "C" int aql_comm_all_reduce_f32(void* comm, float* buf, long n, int average, hipStream_t stream) {
  AQL_CHECK_ARG(comm != nullptr && (buf != nullptr || n == 0) && n >= 0, "aql_comm_all_reduce_f32: bad argument");
  if (n == 0) return AQL_OK;
  if (!rccl_load()) return AQL_ERR_HIP;
  return rccl_status(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, average ? ncclAvg : ncclSum, (ncclComm_t)comm, stream),
                     "ncclAllReduce");
}

// recv[0..recv_n) <- this rank's slice of the sum / mean of send[0..recv_n * nranks): the first half of a ring all-reduce; with
// the optimizer run on the slice and aql_comm_all_gather of the updated parameters it is the ZeRO-1 form of the exchange.
extern "C" int aql_comm_reduce_scatter_f32(void* comm, const float* send, float* recv, long recv_n, int average,
                                           hipStream_t stream) {
  AQL_CHECK_ARG(comm != nullptr && send != nullptr && recv != nullptr && recv_n > 0, "aql_comm_reduce_scatter_f32: bad argument");
  if (!rccl_load()) return AQL_ERR_HIP;
  return rccl_status(g_rccl.ReduceScatter(send, recv, (size_t)recv_n, ncclFloat32, average ? ncclAvg : ncclSum, (ncclComm_t)comm,
                                          stream), "ncclReduceScatter");
}

// recv[rank * send_bytes ...) <- send of every rank (bytes: any element type).
extern "C" int aql_comm_all_gather(void* comm, const void* send, void* recv, long send_bytes, hipStream_t stream) {
  AQL_CHECK_ARG(comm != nullptr && send != nullptr && recv != nullptr && send_bytes > 0, "aql_comm_all_gather: bad argument");
  if (!rccl_load()) return AQL_ERR_HIP;
  return rccl_status(g_rccl.AllGather(send, recv, (size_t)send_bytes, ncclUint8, (ncclComm_t)comm, stream), "ncclAllGather");
}

// buf[0..nbytes) of rank `root` overwrites every rank's copy: DDP's construction-time parameter sync (ppft_train.py:905-912).
extern "C" int aql_comm_broadcast(void* comm, void* buf, long nbytes, int root, hipStream_t stream) {
  AQL_CHECK_ARG(comm != nullptr && buf != nullptr && nbytes > 0 && root >= 0, "aql_comm_broadcast: bad argument");
  if (!rccl_load()) return AQL_ERR_HIP;
  return rccl_status(g_rccl.Broadcast(buf, buf, (size_t)nbytes, ncclUint8, root, (ncclComm_t)comm, stream), "ncclBroadcast");
}

extern "C" int aql_comm_abort(void* comm) {   // tear down without waiting for outstanding collectives (a hung self-test)
  if (!comm) return AQL_OK;
  if (!rccl_load()) return AQL_ERR_HIP;
  return rccl_status(g_rccl.CommAbort((ncclComm_t)comm), "ncclCommAbort");
}

extern "C" int aql_comm_destroy(void* comm) {
  if (!comm) return AQL_OK;
  if (!rccl_load()) return AQL_ERR_HIP;
  return rccl_status(g_rccl.CommDestroy((ncclComm_t)comm), "ncclCommDestroy");
}
