#include "comm.h"
#include <dlfcn.h>
#include <stdexcept>
#include <mutex>
#include <cstring>

namespace isca {

namespace {
struct UniqueId { char internal[Comm::UNIQUE_ID_BYTES]; };      // layout of ncclUniqueId (rccl.h)
constexpr int kDouble = 8, kSum = 0;                             // ncclDouble, ncclSum
using comm_t = void *;
struct Api {
  int (*GetUniqueId)(UniqueId *) = nullptr;
  int (*CommInitRank)(comm_t *, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  int (*Send)(const void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
};
Api &api() {
  static Api a;
  static std::once_flag once;
  static std::string err;
  std::call_once(once, [] {
    void *lib = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) { err = std::string("RCCL not found (dlopen librccl.so.1): ") + dlerror(); return; }
    auto sym = [&](const char *n) { void *p = dlsym(lib, n); if (!p && err.empty()) err = std::string("RCCL symbol missing: ") + n; return p; };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    a.Send = (decltype(a.Send))sym("ncclSend");
    a.Recv = (decltype(a.Recv))sym("ncclRecv");
    a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
    a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
  });
  if (!err.empty()) throw std::runtime_error(err);
  return a;
}
void ck(int rc, const char *what) {
  if (rc != 0) {
    const char *m = api().GetErrorString ? api().GetErrorString(rc) : "?";
    throw std::runtime_error(std::string("RCCL ") + what + " failed: " + m);
  }
}
}  // namespace

class RcclComm final : public Comm {
 public:
  RcclComm(const void *id128, int rank, int world);
  ~RcclComm() override;
  const char *kind() const override { return "rccl"; }
  void all_to_all(const double *send, double *recv, size_t count, hipStream_t s) override;
  void halo(const double *send_lo, const double *send_hi, double *recv_lo, double *recv_hi, size_t count, hipStream_t s) override;
  void all_to_all_with_halo(const double *send, double *recv, size_t count, const double *send_lo, const double *send_hi,
                            double *recv_lo, double *recv_hi, size_t halo_count, hipStream_t s) override;
  void all_reduce_sum(double *buf, size_t count, hipStream_t s) override;

 private:
  void *comm_ = nullptr;
};

void Comm::unique_id(void *id128) {
  if (ipc_id_requested()) { ipc_unique_id(id128); return; }
  if (peer_id_requested()) { peer_unique_id(id128); return; }
  UniqueId id;
  ck(api().GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(id128, &id, sizeof(id));
}

Comm *Comm::create(const void *id128, int rank, int world) {
  if (is_ipc_id(id128)) return make_ipc_comm(id128, rank, world);
  if (is_peer_id(id128)) return make_peer_comm(id128, rank, world);
  return new RcclComm(id128, rank, world);
}

RcclComm::RcclComm(const void *id128, int rank, int world) : Comm(rank, world) {
  UniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  comm_t c = nullptr;
  ck(api().CommInitRank(&c, world, id, rank), "ncclCommInitRank");
  comm_ = c;
}

RcclComm::~RcclComm() {
  if (comm_) api().CommDestroy(comm_);
}

void RcclComm::all_to_all(const double *send, double *recv, size_t count, hipStream_t s) {
  Api &a = api();
  ck(a.GroupStart(), "ncclGroupStart");
  for (int p = 0; p < world_; ++p) {
    ck(a.Send(send + (size_t)p * count, count, kDouble, p, comm_, s), "ncclSend");
    ck(a.Recv(recv + (size_t)p * count, count, kDouble, p, comm_, s), "ncclRecv");
  }
  ck(a.GroupEnd(), "ncclGroupEnd");
}

void RcclComm::halo(const double *send_lo, const double *send_hi, double *recv_lo, double *recv_hi, size_t count, hipStream_t s) {
  if (world_ == 1 || count == 0) return;
  Api &a = api();
  ck(a.GroupStart(), "ncclGroupStart");
  if (rank_ > 0) {
    ck(a.Send(send_lo, count, kDouble, rank_ - 1, comm_, s), "ncclSend");
    ck(a.Recv(recv_lo, count, kDouble, rank_ - 1, comm_, s), "ncclRecv");
  }
  if (rank_ < world_ - 1) {
    ck(a.Send(send_hi, count, kDouble, rank_ + 1, comm_, s), "ncclSend");
    ck(a.Recv(recv_hi, count, kDouble, rank_ + 1, comm_, s), "ncclRecv");
  }
  ck(a.GroupEnd(), "ncclGroupEnd");
}

void RcclComm::all_to_all_with_halo(const double *send, double *recv, size_t count, const double *send_lo, const double *send_hi,
                                double *recv_lo, double *recv_hi, size_t halo_count, hipStream_t s) {
  Api &a = api();
  ck(a.GroupStart(), "ncclGroupStart");
  for (int p = 0; p < world_; ++p) {
    ck(a.Send(send + (size_t)p * count, count, kDouble, p, comm_, s), "ncclSend");
    ck(a.Recv(recv + (size_t)p * count, count, kDouble, p, comm_, s), "ncclRecv");
  }
  if (halo_count > 0 && rank_ > 0) {
    ck(a.Send(send_lo, halo_count, kDouble, rank_ - 1, comm_, s), "ncclSend");
    ck(a.Recv(recv_lo, halo_count, kDouble, rank_ - 1, comm_, s), "ncclRecv");
  }
  if (halo_count > 0 && rank_ < world_ - 1) {
    ck(a.Send(send_hi, halo_count, kDouble, rank_ + 1, comm_, s), "ncclSend");
    ck(a.Recv(recv_hi, halo_count, kDouble, rank_ + 1, comm_, s), "ncclRecv");
  }
  ck(a.GroupEnd(), "ncclGroupEnd");
}

void RcclComm::all_reduce_sum(double *buf, size_t count, hipStream_t s) {
  ck(api().AllReduce(buf, buf, count, kDouble, kSum, comm_, s), "ncclAllReduce");
}

}  // namespace isca
