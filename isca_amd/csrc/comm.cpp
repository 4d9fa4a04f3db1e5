#include "comm.h"
#include <dlfcn.h>
#include <stdexcept>
#include <mutex>
#include <cstring>
#include <cstdlib>
#include <chrono>
#include <thread>

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
  int (*CommGetAsyncError)(comm_t, int *) = nullptr;
  int (*CommAbort)(comm_t) = nullptr;
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
    a.CommGetAsyncError = (decltype(a.CommGetAsyncError))sym("ncclCommGetAsyncError");
    a.CommAbort = (decltype(a.CommAbort))sym("ncclCommAbort");
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
  void synchronize(hipStream_t s) override;
  void abort() noexcept override;

 private:
  void poll_async_error(const char *where);       // ncclCommGetAsyncError: an error of the communicator's proxy / transport becomes an exception (after ncclCommAbort)
  void *comm_ = nullptr;
  double timeout_s_ = 120.0;
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

void Comm::synchronize(hipStream_t s) {
  const hipError_t e = hipStreamSynchronize(s);
  if (e != hipSuccess) throw std::runtime_error(std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
}
bool Comm::fault_here() {
  static const char *spec = getenv("ISCA_FAULT_EXCHANGE");
  const long n = exchanges_++;
  if (!spec) return false;
  const char *colon = std::strchr(spec, ':');
  return colon && atoi(spec) == rank_ && atol(colon + 1) == n;
}

RcclComm::RcclComm(const void *id128, int rank, int world) : Comm(rank, world) {
  if (const char *t = getenv("ISCA_EXCHANGE_TIMEOUT_S")) timeout_s_ = atof(t);
  UniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  comm_t c = nullptr;
  ck(api().CommInitRank(&c, world, id, rank), "ncclCommInitRank");
  comm_ = c;
}

RcclComm::~RcclComm() {
  if (comm_) api().CommDestroy(comm_);
}
void RcclComm::abort() noexcept {
  if (comm_) { try { api().CommAbort(comm_); } catch (...) {} comm_ = nullptr; }
}
// ncclInProgress (7) is not an error; anything else but success: the communicator is torn down (its queued kernels end) and the caller gets the text
void RcclComm::poll_async_error(const char *where) {
  if (!comm_) throw std::runtime_error("RCCL communicator already aborted");
  int st = 0;
  if (api().CommGetAsyncError(comm_, &st) != 0 || (st != 0 && st != 7)) {
    const char *m = api().GetErrorString ? api().GetErrorString(st) : "?";
    abort();
    throw std::runtime_error(std::string("RCCL asynchronous error (") + where + "): " + m + "; communicator aborted");
  }
}
// The host's wait for the step's stream: poll the stream, the communicator's error state and the deadline instead of blocking in the runtime --
// a peer that never joins an exchange would otherwise hold this rank until the job's own limit.
void RcclComm::synchronize(hipStream_t s) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0;; ++spin) {
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) return;
    if (q != hipErrorNotReady) throw std::runtime_error(std::string("hipStreamQuery: ") + hipGetErrorString(q));
    if (spin < 4000) continue;                                    // (the usual wait is shorter than this spin)
    if ((spin & 63) == 0) {
      poll_async_error("while waiting for the step's stream");
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s_) {
        abort();
        throw std::runtime_error("an exchange of the sharded step did not complete within " + std::to_string((int)timeout_s_) +
                                 " s (ISCA_EXCHANGE_TIMEOUT_S): a peer that never joined?  RCCL communicator aborted");
      }
    }
    std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
}

void RcclComm::all_to_all(const double *send, double *recv, size_t count, hipStream_t s) {
  Api &a = api();
  if (fault_here()) return;
  ck(a.GroupStart(), "ncclGroupStart");
  for (int p = 0; p < world_; ++p) {
    ck(a.Send(send + (size_t)p * count, count, kDouble, p, comm_, s), "ncclSend");
    ck(a.Recv(recv + (size_t)p * count, count, kDouble, p, comm_, s), "ncclRecv");
  }
  ck(a.GroupEnd(), "ncclGroupEnd");
  poll_async_error("after ncclGroupEnd");
}

void RcclComm::halo(const double *send_lo, const double *send_hi, double *recv_lo, double *recv_hi, size_t count, hipStream_t s) {
  if (world_ == 1 || count == 0) return;
  Api &a = api();
  if (fault_here()) return;
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
  poll_async_error("after ncclGroupEnd");
}

void RcclComm::all_to_all_with_halo(const double *send, double *recv, size_t count, const double *send_lo, const double *send_hi,
                                double *recv_lo, double *recv_hi, size_t halo_count, hipStream_t s) {
  Api &a = api();
  if (fault_here()) return;
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
  poll_async_error("after ncclGroupEnd");
}

void RcclComm::all_reduce_sum(double *buf, size_t count, hipStream_t s) {
  if (fault_here()) return;
  ck(api().AllReduce(buf, buf, count, kDouble, kSum, comm_, s), "ncclAllReduce");
  poll_async_error("after ncclAllReduce");
}

}  // namespace isca
