// RCCL behind the C ABI (msclip_comm_init(rank, world, uid), allgather_feats): the feature all-gather of the
// contrastive head (reference lib/utils/comm.py:140-154 via M.py:3138-3141) and the loss / gradient all-reduce (reference
// lib/utils/utils.py:66-73) as stream-ordered calls on the CALLER's stream -- ncclAllGather / ncclAllReduce enqueue their kernel
// on that stream, so the collective is an ordinary entry of a launch plan (msclip_plan_*), is captured by a hipGraph capture, and
// can be driven by a host that has no Python.  The unique id is created by rank 0 (msclip_comm_unique_id) and handed to the other
// ranks by whatever rendezvous the host has (the Python side uses the torch.distributed store that already exists).
//
// librccl is bound at run time (dlopen): the rest of the library loads and works on a box without RCCL, and inside a PyTorch
// process the communicator lives in the RCCL instance torch has already loaded (same SONAME) instead of a second copy.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "common.h"
#include "plan.h"

namespace {

typedef void* nccl_comm_t;
typedef struct { char internal[128]; } nccl_uid_t;           // NCCL_UNIQUE_ID_BYTES
enum { NCCL_INT8 = 0, NCCL_UINT8 = 1, NCCL_INT32 = 2, NCCL_FLOAT32 = 7, NCCL_BFLOAT16 = 9 };   // ncclDataType_t (rccl.h)
enum { NCCL_SUM = 0, NCCL_MAX = 2 };                                                            // ncclRedOp_t

struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(nccl_uid_t*) = nullptr;
  int (*CommInitRank)(nccl_comm_t*, int, nccl_uid_t, int) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*CommGetAsyncError)(nccl_comm_t, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};

Rccl load_rccl() {
  Rccl r;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names) {                                // an instance that is already in the process (torch's) first
    r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (r.lib) break;
  }
  if (!r.lib)
    for (const char* n : names) {
      r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
  if (!r.lib) {
    fprintf(stderr, "msclip_comm: librccl could not be loaded: %s\n", dlerror());
    return r;
  }
#define SYM(field, name) *(void**)(&r.field) = dlsym(r.lib, name)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllGather, "ncclAllGather");
  SYM(AllReduce, "ncclAllReduce");
  SYM(CommGetAsyncError, "ncclCommGetAsyncError");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.AllReduce;
  return r;
}

Rccl& rccl() {
  static Rccl r = load_rccl();                                 // (one load, thread-safe: C++11 static initialisation)
  return r;
}

int nccl_dtype(int dtype, size_t* esize) {
  switch (dtype) {
    case 0: *esize = 2; return NCCL_BFLOAT16;
    case 1: *esize = 4; return NCCL_FLOAT32;
    case 2: *esize = 1; return NCCL_UINT8;
    case 3: *esize = 4; return NCCL_INT32;
    default: return -1;
  }
}

int check(int rc, const char* what) {
  if (rc == 0) return MSCLIP_OK;
  Rccl& r = rccl();
  fprintf(stderr, "msclip_comm: %s failed: %s\n", what, r.GetErrorString ? r.GetErrorString(rc) : "?");
  return MSCLIP_ELAUNCH;
}

}  // namespace

// 128 bytes identifying a new communicator; call on ONE rank and distribute the bytes (reference: the rendezvous behind
// torch.distributed.init_process_group, lib/utils/comm.py:12-62 only reads rank / world from it).
extern "C" int msclip_comm_unique_id(void* id128) {
  if (!id128) return MSCLIP_EINVAL;
  Rccl& r = rccl();
  if (!r.ok) return MSCLIP_ELAUNCH;
  nccl_uid_t id;
  const int rc = check(r.GetUniqueId(&id), "ncclGetUniqueId");
  if (rc == MSCLIP_OK) memcpy(id128, &id, sizeof(id));
  return rc;
}

// Collective over all ranks: every rank calls it with the same id, on the device it will launch from (hipSetDevice first).
extern "C" int msclip_comm_init(int rank, int world, const void* id128, void** comm) {
  if (!id128 || !comm || world <= 0 || rank < 0 || rank >= world) return MSCLIP_EINVAL;
  Rccl& r = rccl();
  if (!r.ok) return MSCLIP_ELAUNCH;
  nccl_uid_t id;
  memcpy(&id, id128, sizeof(id));
  nccl_comm_t c = nullptr;
  const int rc = check(r.CommInitRank(&c, world, id, rank), "ncclCommInitRank");
  if (rc == MSCLIP_OK) *comm = c;
  return rc;
}

extern "C" int msclip_comm_destroy(void* comm) {
  if (!comm) return MSCLIP_EINVAL;
  Rccl& r = rccl();
  if (!r.ok) return MSCLIP_ELAUNCH;
  return check(r.CommDestroy((nccl_comm_t)comm), "ncclCommDestroy");
}

// 0 = healthy; otherwise the asynchronous error RCCL has recorded for the communicator (a peer died, a link failed).
extern "C" int msclip_comm_async_error(void* comm) {
  if (!comm) return MSCLIP_EINVAL;
  Rccl& r = rccl();
  if (!r.ok || !r.CommGetAsyncError) return MSCLIP_ELAUNCH;
  int err = 0;
  if (r.CommGetAsyncError((nccl_comm_t)comm, &err) != 0) return MSCLIP_ELAUNCH;
  return err;
}

// recv[rank r][count] = rank r's send[count], rank-major (reference gather_tensors, lib/utils/comm.py:150-153: slot order = rank
// order; the local-rows-keep-their-gradient splice is the caller's, the data is the same).  dtype 0 = bf16, 1 = fp32, 2 = bytes,
// 3 = int32.  Stream-ordered on `stream`; send may alias recv + rank * count (in place).
extern "C" int msclip_allgather_feats(void* comm, const void* send, void* recv, long long count, int dtype, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_allgather_feats, stream, comm, send, recv, count, dtype);
  size_t es = 0;
  const int dt = nccl_dtype(dtype, &es);
  if (!comm || !send || !recv || count <= 0 || dt < 0) return MSCLIP_EINVAL;
  Rccl& r = rccl();
  if (!r.ok) return MSCLIP_ELAUNCH;
  return check(r.AllGather(send, recv, (size_t)count, dt, (nccl_comm_t)comm, (hipStream_t)stream), "ncclAllGather");
}

// recv = sum (op 0) / max (op 1) over ranks of send, element-wise (the loss partials of SURVEY s8(e) option B; the gradient
// buckets of the training step: reference lib/utils/utils.py:66-73).  In place when send == recv.
extern "C" int msclip_allreduce(void* comm, const void* send, void* recv, long long count, int dtype, int op, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_allreduce, stream, comm, send, recv, count, dtype, op);
  size_t es = 0;
  const int dt = nccl_dtype(dtype, &es);
  if (!comm || !send || !recv || count <= 0 || dt < 0 || op < 0 || op > 1) return MSCLIP_EINVAL;
  Rccl& r = rccl();
  if (!r.ok) return MSCLIP_ELAUNCH;
  return check(r.AllReduce(send, recv, (size_t)count, dt, op == 0 ? NCCL_SUM : NCCL_MAX, (nccl_comm_t)comm, (hipStream_t)stream),
               "ncclAllReduce");
}
