// Host-staged exchange between processes that may share ONE GPU: the second implementation of isca::Comm (comm.h).
//
// What it is for: RCCL refuses two ranks on one device, so on a one-GPU box the C++ sharded step loop (api.hip: sharded_step --
// halo exchange, lat -> m all-to-all, m -> lat all-to-all, all-reduce, the RAW filter's third exchange) could only ever run with
// world = 1.  With ISCA_COMM=ipc the same loop runs with 2, 4 or 8 processes: every exchange copies the send buffer to a file
// mapped by all ranks (one "outbox" per rank), meets the other ranks at a barrier in that mapping, and copies the blocks addressed
// to it from the other ranks' outboxes to its receive buffer.  Same interface, same buffers, same order of calls as the RCCL
// implementation; replaces mpp_transmit / mpp_update_domains / mpp_sum of the reference (transforms.F90:970-1056,
// fv_advection.F90:161-162, transforms.F90:1059-1077) for verification runs.  Every call synchronises its stream: not a fast path.
//
// ISCA_IPC_SERIALIZE=1 (measurement): the ranks' device work between two exchanges runs ONE RANK AT A TIME -- a rank returns from an exchange only
// when the rank before it has finished its next segment (its next exchange's stream synchronise passes the turn on).  The kernels of a rank then
// run alone on the shared GPU, as they would on a GPU of their own, and their HIP-event durations are a 1/P shard's compute times
// (bench.py: shard_compute_ms).  The wall clock of such a run means nothing.
//
// Layout: a header file (barrier words, abort flag) drawn by the rank that makes the id, and one outbox file per rank:
//   [ reduce area 4 KB | halo to rank-1 | halo to rank+1 | all-to-all blocks [world][count] ]
// Files live in /dev/shm when it has room (they are sparse: only touched pages exist), else in /tmp; they are unlinked as soon as
// every rank has mapped them.
#include "comm.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/statvfs.h>
#include <unistd.h>

namespace isca {

namespace {

constexpr char kMagic[8] = {'I', 'S', 'C', 'A', 'I', 'P', 'C', '1'};
constexpr size_t kReduceBytes = 4096, kHeaderBytes = 4096;

struct Header {                         // zero-filled by ftruncate
  std::atomic<uint32_t> arrived;        // sense-reversing barrier
  std::atomic<uint32_t> generation;
  std::atomic<uint32_t> aborted;        // 1 + rank of the first rank that gave up
  std::atomic<uint32_t> attached;
  std::atomic<uint64_t> turn;           // ISCA_IPC_SERIALIZE: whose kernels may run (round * world + rank)
  std::atomic<uint32_t> free_run;       // ... until a rank leaves (its communicator is destroyed): nobody passes the turn on after its last exchange
};

struct IpcId {                          // the 128 bytes the ranks share
  char magic[8];
  uint64_t halo_bytes;                  // per direction
  uint64_t a2a_bytes;
  char path[104];                       // header file; outbox of rank r: path + ".r<r>"
};
static_assert(sizeof(IpcId) == Comm::UNIQUE_ID_BYTES, "id size");

void hip_ck(hipError_t e, const char *what) {
  if (e != hipSuccess) throw std::runtime_error(std::string("ipc comm: ") + what + ": " + hipGetErrorString(e));
}

double env_num(const char *name, double dflt) {
  const char *v = getenv(name);
  return v && *v ? atof(v) : dflt;
}

void *map_file(const std::string &path, size_t bytes, bool create) {
  int fd = open(path.c_str(), create ? (O_RDWR | O_CREAT | O_EXCL) : O_RDWR, 0600);
  if (fd < 0) throw std::runtime_error("ipc comm: cannot open " + path + ": " + strerror(errno));
  if (create && ftruncate(fd, (off_t)bytes) != 0) { close(fd); throw std::runtime_error("ipc comm: cannot size " + path); }
  void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) throw std::runtime_error("ipc comm: cannot map " + path + ": " + strerror(errno));
  return p;
}

class IpcComm final : public Comm {
 public:
  IpcComm(const IpcId &id, int rank, int world) : Comm(rank, world), id_(id) {
    timeout_s_ = env_num("ISCA_IPC_TIMEOUT_S", 120.0);
    serialize_ = env_num("ISCA_IPC_SERIALIZE", 0.0) != 0.0;
    slot_bytes_ = kReduceBytes + 2 * id.halo_bytes + id.a2a_bytes;
    hdr_ = (Header *)map_file(id.path, kHeaderBytes, false);
    box_.assign(world, nullptr);
    try {
      box_[rank] = (char *)map_file(box_path(rank), slot_bytes_, true);
      hdr_->attached.fetch_add(1);
      barrier("attach");                                    // every outbox exists
      for (int q = 0; q < world; ++q)
        if (q != rank) box_[q] = (char *)map_file(box_path(q), slot_bytes_, false);
      barrier("map");                                       // every rank has every mapping: the names can go
      unlink(box_path(rank).c_str());
      if (rank == 0) unlink(id.path);
      wait_turn("start");
    } catch (...) {
      abort();
      unmap_all();
      throw;
    }
  }
  ~IpcComm() override {
    if (serialize_ && hdr_) hdr_->free_run.store(1, std::memory_order_release);
    unmap_all();
  }
  const char *kind() const override { return "ipc"; }

  void all_to_all(const double *send, double *recv, size_t count, hipStream_t s) override {
    all_to_all_with_halo(send, recv, count, nullptr, nullptr, nullptr, nullptr, 0, s);
  }
  void halo(const double *send_lo, const double *send_hi, double *recv_lo, double *recv_hi, size_t count, hipStream_t s) override {
    if (world_ == 1 || count == 0) return;
    all_to_all_with_halo(nullptr, nullptr, 0, send_lo, send_hi, recv_lo, recv_hi, count, s);
  }
  void all_to_all_with_halo(const double *send, double *recv, size_t count, const double *send_lo, const double *send_hi,
                            double *recv_lo, double *recv_hi, size_t halo_count, hipStream_t s) override {
    if (fault_here()) never_join("exchange");
    const size_t blk = count * sizeof(double), hb = halo_count * sizeof(double);
    if (blk * world_ > id_.a2a_bytes) throw std::runtime_error("ipc comm: all-to-all larger than the outbox (raise ISCA_IPC_A2A_MB)");
    if (hb > id_.halo_bytes) throw std::runtime_error("ipc comm: halo rows larger than the outbox (raise ISCA_IPC_HALO_MB)");
    char *mine = box_[rank_];
    if (blk) hip_ck(hipMemcpyAsync(mine + a2a_off(), send, blk * world_, hipMemcpyDeviceToHost, s), "copy out (all-to-all)");
    if (hb && rank_ > 0) hip_ck(hipMemcpyAsync(mine + halo_off(0), send_lo, hb, hipMemcpyDeviceToHost, s), "copy out (halo)");
    if (hb && rank_ < world_ - 1) hip_ck(hipMemcpyAsync(mine + halo_off(1), send_hi, hb, hipMemcpyDeviceToHost, s), "copy out (halo)");
    hip_ck(hipStreamSynchronize(s), "synchronize");
    pass_turn();
    barrier("exchange: data out");
    for (int q = 0; blk && q < world_; ++q)
      hip_ck(hipMemcpyAsync(recv + (size_t)q * count, box_[q] + a2a_off() + (size_t)rank_ * blk, blk, hipMemcpyHostToDevice, s), "copy in (all-to-all)");
    if (hb && rank_ > 0)               // my lower neighbour's rows "to rank+1"
      hip_ck(hipMemcpyAsync(recv_lo, box_[rank_ - 1] + halo_off(1), hb, hipMemcpyHostToDevice, s), "copy in (halo)");
    if (hb && rank_ < world_ - 1)
      hip_ck(hipMemcpyAsync(recv_hi, box_[rank_ + 1] + halo_off(0), hb, hipMemcpyHostToDevice, s), "copy in (halo)");
    hip_ck(hipStreamSynchronize(s), "synchronize");
    barrier("exchange: data in");                           // the outboxes may be overwritten again
    wait_turn("exchange: turn");
  }
  void all_reduce_sum(double *buf, size_t count, hipStream_t s) override {
    if (fault_here()) never_join("all-reduce");
    if (count * sizeof(double) > kReduceBytes) throw std::runtime_error("ipc comm: all-reduce larger than its area");
    hip_ck(hipMemcpyAsync(box_[rank_], buf, count * sizeof(double), hipMemcpyDeviceToHost, s), "copy out (all-reduce)");
    hip_ck(hipStreamSynchronize(s), "synchronize");
    pass_turn();
    barrier("all-reduce: data out");
    double tot[kReduceBytes / sizeof(double)];
    for (size_t i = 0; i < count; ++i) tot[i] = 0.0;
    for (int q = 0; q < world_; ++q) {                      // rank order on every rank: every rank holds the same bits
      const double *v = (const double *)box_[q];
      for (size_t i = 0; i < count; ++i) tot[i] += v[i];
    }
    hip_ck(hipMemcpyAsync(buf, tot, count * sizeof(double), hipMemcpyHostToDevice, s), "copy in (all-reduce)");
    hip_ck(hipStreamSynchronize(s), "synchronize");
    barrier("all-reduce: data in");
    wait_turn("all-reduce: turn");
  }
  void abort() noexcept override {
    if (hdr_) { uint32_t none = 0; hdr_->aborted.compare_exchange_strong(none, (uint32_t)rank_ + 1); }
  }

 private:
  std::string box_path(int r) const { return std::string(id_.path) + ".r" + std::to_string(r); }
  size_t halo_off(int dir) const { return kReduceBytes + (size_t)dir * id_.halo_bytes; }
  size_t a2a_off() const { return kReduceBytes + 2 * id_.halo_bytes; }
  void unmap_all() {
    for (auto &b : box_) if (b) { munmap(b, slot_bytes_); b = nullptr; }
    if (hdr_) { munmap(hdr_, kHeaderBytes); hdr_ = nullptr; }
  }
  // ISCA_FAULT_EXCHANGE (comm.h): this rank does not join the exchange -- the equivalent of a rank whose ncclRecv is never posted.  It waits, like a
  // rank stuck behind a device-side exchange, until a peer gives up (their barrier times out and sets the abort word) or its own limit passes.
  [[noreturn]] void never_join(const char *what) {
    const auto t0 = std::chrono::steady_clock::now();
    while (!hdr_->aborted.load(std::memory_order_acquire) && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2 * timeout_s_)
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
    abort();
    throw std::runtime_error(std::string("ipc comm: fault injected (ISCA_FAULT_EXCHANGE): this rank did not join its ") + what);
  }
  // ISCA_IPC_SERIALIZE: my segment of device work is over (the stream has just been synchronised): the next rank may run its own
  void pass_turn() {
    if (!serialize_) return;
    (void)hipDeviceSynchronize();          // (the side stream's kernels too: nothing of mine runs into the next rank's turn)
    hdr_->turn.store(round_ * (uint64_t)world_ + (uint64_t)rank_ + 1, std::memory_order_release);
    ++round_;
  }
  // ... and I go on (return to the caller, who queues the next segment) when the rank before me has passed it on
  void wait_turn(const char *where) {
    if (!serialize_) return;
    const uint64_t mine = round_ * (uint64_t)world_ + (uint64_t)rank_;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0; hdr_->turn.load(std::memory_order_acquire) != mine && !hdr_->free_run.load(std::memory_order_acquire); ++spin) {
      const uint32_t ab = hdr_->aborted.load(std::memory_order_acquire);
      if (ab) throw std::runtime_error(std::string("ipc comm: rank ") + std::to_string(ab - 1) + " stopped with an error (" + where + ")");
      if (spin < 200) std::this_thread::yield();
      else std::this_thread::sleep_for(std::chrono::microseconds(20));
      if ((spin & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s_) {
        abort();
        throw std::runtime_error(std::string("ipc comm: timed out waiting for the turn (") + where + ")");
      }
    }
  }
  // every rank arrives, the last one opens the next generation; a rank that gave up (abort) or never comes (timeout) is an error
  void barrier(const char *where) {
    const uint32_t gen = hdr_->generation.load(std::memory_order_acquire);
    if (hdr_->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world_) {
      hdr_->arrived.store(0, std::memory_order_relaxed);
      hdr_->generation.store(gen + 1, std::memory_order_release);
      return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0; hdr_->generation.load(std::memory_order_acquire) == gen; ++spin) {
      const uint32_t ab = hdr_->aborted.load(std::memory_order_acquire);
      if (ab) throw std::runtime_error(std::string("ipc comm: rank ") + std::to_string(ab - 1) + " stopped with an error (" + where + ")");
      if (spin < 200) std::this_thread::yield();
      else std::this_thread::sleep_for(std::chrono::microseconds(50));
      if ((spin & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s_) {
        abort();
        throw std::runtime_error(std::string("ipc comm: timed out waiting for the other ranks (") + where + ")");
      }
    }
  }

  IpcId id_;
  Header *hdr_ = nullptr;
  std::vector<char *> box_;
  size_t slot_bytes_ = 0;
  double timeout_s_ = 120.0;
  bool serialize_ = false;
  uint64_t round_ = 0;
};

}  // namespace

bool ipc_id_requested() {
  const char *v = getenv("ISCA_COMM");
  return v && std::strcmp(v, "ipc") == 0;
}

void ipc_unique_id(void *id128) {
  IpcId id;
  std::memset(&id, 0, sizeof(id));
  std::memcpy(id.magic, kMagic, sizeof(kMagic));
  id.halo_bytes = (uint64_t)(env_num("ISCA_IPC_HALO_MB", 64.0) * 1048576.0);
  id.a2a_bytes = (uint64_t)(env_num("ISCA_IPC_A2A_MB", 1024.0) * 1048576.0);
  const char *dir = getenv("ISCA_IPC_DIR");
  std::string d = dir && *dir ? dir : "/dev/shm";
  struct statvfs vfs;
  if (!(dir && *dir) && (statvfs(d.c_str(), &vfs) != 0 || (double)vfs.f_bavail * vfs.f_frsize < 4.0e9)) d = "/tmp";   // a container's 64 MB /dev/shm
  snprintf(id.path, sizeof(id.path), "%s/isca_ipc_%d_%llx", d.c_str(), (int)getpid(),
           (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
  munmap(map_file(id.path, kHeaderBytes, true), kHeaderBytes);                                // zero-filled header
  std::memcpy(id128, &id, sizeof(id));
}

bool is_ipc_id(const void *id128) { return std::memcmp(id128, kMagic, sizeof(kMagic)) == 0; }

Comm *make_ipc_comm(const void *id128, int rank, int world) {
  IpcId id;
  std::memcpy(&id, id128, sizeof(id));
  id.path[sizeof(id.path) - 1] = 0;
  return new IpcComm(id, rank, world);
}

}  // namespace isca
