// Multi-GPU exchange of the C-ABI: thin RCCL wrappers (included at the end of fisr_api.hip).
//
// The path shards without any exchange of intermediate activations: what moves between
// GPUs is the 32-px input halo of a tile (an all-gather of border strips when the inputs are produced
// sharded) and the output tiles / frames (an all-gather to the rank that stitches and writes).  Hosts
// that run under torch.distributed use its "nccl" backend (= RCCL); these entry points give the same two
// collectives to a host that has no torch (examples/c_host.c style).  librccl is opened lazily with
// dlopen, so the single-GPU library has no RCCL dependency.
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: nothing is linked

namespace {

struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string why;
};

RcclApi& rccl() {
  static RcclApi api = [] {
    RcclApi a;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
      if ((a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!a.handle) { a.why = std::string("dlopen(librccl): ") + dlerror(); return a; }
    auto sym = [&](const char* s) { void* p = dlsym(a.handle, s); if (!p && a.why.empty()) a.why = std::string("librccl lacks ") + s; return p; };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
    a.Send = (decltype(a.Send))sym("ncclSend");
    a.Recv = (decltype(a.Recv))sym("ncclRecv");
    a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    return a;
  }();
  return api;
}

int rccl_fail(const char* what, ncclResult_t r) {
  return fail(nullptr, FISR_EHIP, std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error"));
}

}  // namespace

struct fisr_comm {
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0, dev = 0;
};

extern "C" {

int fisr_comm_unique_id(void* id_out) {
  if (!id_out) return fail(nullptr, FISR_EINVAL, "fisr_comm_unique_id: null");
  static_assert(sizeof(ncclUniqueId) == FISR_COMM_ID_BYTES, "id size");
  if (!rccl().why.empty()) return fail(nullptr, FISR_ESTATE, rccl().why);
  ncclUniqueId id;
  const ncclResult_t r = rccl().GetUniqueId(&id);
  if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
  memcpy(id_out, &id, sizeof id);
  return 0;
}

int fisr_comm_init(fisr_comm** out, const void* id, int nranks, int rank, int device_id) {
  if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(nullptr, FISR_EINVAL, "fisr_comm_init: bad argument");
  if (!rccl().why.empty()) return fail(nullptr, FISR_ESTATE, rccl().why);
  DeviceGuard guard(device_id);
  HIP_OK(nullptr, guard.err);
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  fisr_comm* c = new fisr_comm;
  c->nranks = nranks; c->rank = rank; c->dev = device_id;
  const ncclResult_t r = rccl().CommInitRank(&c->comm, nranks, uid, rank);
  if (r != ncclSuccess) { delete c; return rccl_fail("ncclCommInitRank", r); }
  *out = c;
  return 0;
}

int fisr_comm_allgather(fisr_comm* c, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
  if (!c || !send || !recv) return fail(nullptr, FISR_EINVAL, "fisr_comm_allgather: null");
  if (bytes_per_rank == 0) return 0;
  DeviceGuard guard(c->dev);
  HIP_OK(nullptr, guard.err);
  const ncclResult_t r = rccl().AllGather(send, recv, bytes_per_rank, ncclUint8, c->comm, (hipStream_t)stream);
  return r == ncclSuccess ? 0 : rccl_fail("ncclAllGather", r);
}

int fisr_comm_sendrecv(fisr_comm* c, const void* send, void* recv, size_t bytes, int peer, void* stream) {
  if (!c || !send || !recv || peer < 0 || peer >= c->nranks) return fail(nullptr, FISR_EINVAL, "fisr_comm_sendrecv: bad argument");
  if (bytes == 0) return 0;
  DeviceGuard guard(c->dev);
  HIP_OK(nullptr, guard.err);
  ncclResult_t r = rccl().GroupStart();
  if (r == ncclSuccess) r = rccl().Send(send, bytes, ncclUint8, peer, c->comm, (hipStream_t)stream);
  if (r == ncclSuccess) r = rccl().Recv(recv, bytes, ncclUint8, peer, c->comm, (hipStream_t)stream);
  const ncclResult_t r2 = rccl().GroupEnd();
  if (r == ncclSuccess) r = r2;
  return r == ncclSuccess ? 0 : rccl_fail("ncclSend/ncclRecv", r);
}

int fisr_comm_rank(const fisr_comm* c) { return c ? c->rank : -1; }
int fisr_comm_size(const fisr_comm* c) { return c ? c->nranks : 0; }

void fisr_comm_destroy(fisr_comm* c) {
  if (!c) return;
  DeviceGuard guard(c->dev);
  if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);
  delete c;
}

}  // extern "C"
