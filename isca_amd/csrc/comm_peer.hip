// Device-resident exchange between the ranks of one node: the third implementation of isca::Comm (comm.h), ISCA_COMM=peer.
//
// Why: at the headline size a rank's eighth of the grid is ~10 kernels of 5-10 us (bench.py: shard_compute_ms), and the step has four dependent
// exchanges -- halo rows, lat -> m all-to-all, m -> lat all-to-all, the all-reduce of the fixer sums (transforms.F90:970-1056, fv_advection.F90:161-162,
// transforms.F90:1059-1077 in the reference).  Through RCCL each is a grouped send/recv or an all-reduce with its own proxy hand-shakes: tens of
// microseconds apiece, i.e. more than the compute they separate.  Here every exchange is ONE kernel on the step's stream and nothing else:
//   * every rank's receive buffers (the spectral side's Fourier buffer, the grid side's, the halo rows) and a small block of flags are exported once
//     with hipIpcGetMemHandle and opened by the peers, so a rank's kernel STORES its blocks straight into the peers' receive buffers (over xGMI on a
//     node: one hop per peer, all links at once -- the all-to-all is a full mesh of point-to-point writes, no ring);
//   * per exchange and peer two flags in the receiver's memory: `ready` (the receiver has reached this exchange, so everything that read the buffer
//     before is done: written by the receiver into the SENDER's block) and `done` (the sender's block has landed: system-scope release after the
//     data); the kernel posts ready, waits for the peers', copies, signals done, waits for the peers' done -- when it ends the data is there;
//   * the all-reduce is the same with 16 doubles per peer and a sum in rank order (every rank holds the same bits), in one block.
// No host synchronisation, no proxy thread, no second launch.  A peer that never arrives: the spin loops give up after ISCA_PEER_TIMEOUT_S (60 s)
// and leave a word in pinned host memory which the next synchronisation point turns into the error.
//
// Verified like the host-staged implementation (comm_ipc.cpp): 2, 4 and 8 PROCESSES sharing one GPU run the library's sharded step loop through it
// (tests/test_gpu_parity.py::test_sharded_native_loop[...peer...]) and land on the one-rank run.  On one GPU the peers share an L2, so what those
// tests cannot show is cross-GPU visibility -- the design relies on system-scope release / acquire on fine-grained flag memory and on the
// system-scope acquire of the kernels launched after the exchange; it has NOT run over xGMI (no multi-GPU node was available): RCCL stays the default.
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
#include <unistd.h>

namespace isca {

namespace {

constexpr char kPeerMagic[8] = {'I', 'S', 'C', 'A', 'P', 'E', 'E', 'R'};
constexpr int kMaxPeers = 16, kKinds = 4;                  // exchange kinds: 0 forward all-to-all, 1 inverse, 2 halo, 3 all-reduce
constexpr int kRedMax = 16;

// ---- the flag block of a rank (device memory, fine-grained when the runtime gives it), written by the peers
struct PeerFlags {
  unsigned long long ready[kKinds][kMaxPeers];             // [kind][q], written BY PEER q: q has reached exchange seq of this kind -- its receive buffer is mine to write
  unsigned long long done[kKinds][kMaxPeers];              // [kind][q], written BY PEER q: q's block of exchange seq is in my receive buffer
  unsigned int arrive[kKinds][kMaxPeers];                  // local: slices of my block for peer q that have been stored (the last one signals done)
  double mailbox[kMaxPeers][kRedMax];                      // all-reduce contributions, by sender
};

struct SetupHeader {                                       // host side, a mapped file: only for the set-up
  std::atomic<uint32_t> arrived, generation, aborted, pad;
  hipIpcMemHandle_t handle[kMaxPeers][4];                  // [rank][0 fwd recv, 1 inv recv, 2 halo recv, 3 flags]
  int device[kMaxPeers];
};
struct PeerId {
  char magic[8];
  char path[120];
};
static_assert(sizeof(PeerId) == Comm::UNIQUE_ID_BYTES, "id size");

void pk(hipError_t e, const char *what) {
  if (e != hipSuccess) throw std::runtime_error(std::string("peer comm: ") + what + ": " + hipGetErrorString(e));
}
double penv(const char *name, double dflt) { const char *v = getenv(name); return v && *v ? atof(v) : dflt; }

struct PeerArgs {
  int me, world, kind, npeers, count_red;
  int peers[kMaxPeers];                                    // the ranks this exchange involves (all of them, or the neighbours)
  unsigned long long seq;
  const double *send[kMaxPeers];                           // my block for peer i (local)
  double *dst[kMaxPeers];                                  // where it goes: inside peer i's receive buffer (remote; local for myself)
  size_t count[kMaxPeers];                                 // doubles
  PeerFlags *flags[kMaxPeers];                             // flag block of rank r (remote; mine is flags[me])
  double *red;                                             // all-reduce: my values in, the sum out
  int *err;                                                // pinned host word: 1 + kind when a wait timed out
  long long timeout_ticks;                                 // wall_clock64 ticks (100 MHz)
};

__device__ __forceinline__ bool spin_until(const unsigned long long *flag, unsigned long long seq, const PeerArgs &a) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
    __builtin_amdgcn_s_sleep(32);
    if (wall_clock64() - t0 > a.timeout_ticks) { *a.err = 1 + a.kind; return false; }
  }
  return true;
}

// one exchange: blockIdx.y = index into a.peers, blockIdx.x = slice of that peer's block
__global__ __launch_bounds__(256) void k_peer_exchange(PeerArgs a) {
  const int q = a.peers[blockIdx.y], nb = gridDim.x;
  PeerFlags *mine = a.flags[a.me];
  __shared__ int ok;
  if (q == a.me) {                                         // my own block: a local copy
    const double *src = a.send[blockIdx.y]; double *dst = a.dst[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.count[blockIdx.y]; i += (size_t)nb * 256) dst[i] = src[i];
    return;
  }
  if (threadIdx.x == 0) {
    if (blockIdx.x == 0) __hip_atomic_store(&a.flags[q]->ready[a.kind][a.me], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // I am here: q may write into my buffer
    ok = spin_until(&mine->ready[a.kind][q], a.seq, a) ? 1 : 0;                                                                       // q is there: I may write into its
  }
  __syncthreads();
  if (!ok) return;
  {
    const double *src = a.send[blockIdx.y]; double *dst = a.dst[blockIdx.y];
    const size_t n = a.count[blockIdx.y];
    if (((((size_t)src) | ((size_t)dst)) & 15) == 0 && (n & 1) == 0) {
      const double2 *s2 = (const double2 *)src; double2 *d2 = (double2 *)dst;
      for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n / 2; i += (size_t)nb * 256) d2[i] = s2[i];
    } else
      for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)nb * 256) dst[i] = src[i];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int before = atomicAdd(&mine->arrive[a.kind][q], 1u);
    if (before == (unsigned)nb - 1) {                      // the last slice of my block for q: it has landed
      mine->arrive[a.kind][q] = 0;
      __threadfence_system();
      __hip_atomic_store(&a.flags[q]->done[a.kind][a.me], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (blockIdx.x == 0) (void)spin_until(&mine->done[a.kind][q], a.seq, a);       // ... and q's block for me: the kernel ends when everything is here
  }
}

// all-reduce (sum) of a.count_red <= 16 doubles: one block, thread p < world talks to rank p
__global__ __launch_bounds__(64) void k_peer_all_reduce(PeerArgs a) {
  PeerFlags *mine = a.flags[a.me];
  const int p = threadIdx.x;
  __shared__ int bad;
  if (p == 0) bad = 0;
  __syncthreads();
  if (p < a.world) {
    if (p == a.me) {
      for (int i = 0; i < a.count_red; ++i) mine->mailbox[a.me][i] = a.red[i];
    } else {
      __hip_atomic_store(&a.flags[p]->ready[3][a.me], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (spin_until(&mine->ready[3][p], a.seq, a)) {
        for (int i = 0; i < a.count_red; ++i) a.flags[p]->mailbox[a.me][i] = a.red[i];
        __threadfence_system();
        __hip_atomic_store(&a.flags[p]->done[3][a.me], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (!spin_until(&mine->done[3][p], a.seq, a)) bad = 1;
      } else bad = 1;
    }
  }
  __syncthreads();
  if (bad) return;
  if (p < a.count_red) {                                   // rank order on every rank: the same bits everywhere
    double s = 0.0;
    for (int r = 0; r < a.world; ++r) s += ((volatile double *)mine->mailbox[r])[p];
    a.red[p] = s;
  }
}

void *map_setup(const std::string &path, bool create) {
  int fd = open(path.c_str(), create ? (O_RDWR | O_CREAT | O_EXCL) : O_RDWR, 0600);
  if (fd < 0) throw std::runtime_error("peer comm: cannot open " + path + ": " + strerror(errno));
  if (create && ftruncate(fd, (off_t)sizeof(SetupHeader)) != 0) { close(fd); throw std::runtime_error("peer comm: cannot size " + path); }
  void *p = mmap(nullptr, sizeof(SetupHeader), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) throw std::runtime_error("peer comm: cannot map " + path);
  return p;
}

class PeerComm final : public Comm {
 public:
  PeerComm(const PeerId &id, int rank, int world) : Comm(rank, world), id_(id) {
    if (world > kMaxPeers) throw std::runtime_error("peer comm: at most 16 ranks");
    timeout_s_ = penv("ISCA_PEER_TIMEOUT_S", 60.0);
    hdr_ = (SetupHeader *)map_setup(id.path, false);
    pk(hipGetDevice(&dev_), "hipGetDevice");
    // the flag block: fine-grained device memory (remote atomics and the peers' polling see each other's writes), plain device memory if the
    // runtime refuses; zeroed
    if (hipExtMallocWithFlags((void **)&flags_, sizeof(PeerFlags), hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      pk(hipMalloc((void **)&flags_, sizeof(PeerFlags)), "hipMalloc (flags)");
      // Coarse-grained flags are only known to work between processes that share ONE device (the peers then poll the same L2).  Across GPUs the
      // system-scope polling is not guaranteed to see remote writes to coarse-grained memory, and every exchange would end in its time-out:
      // said loudly here instead (RCCL is the default driver; ISCA_COMM=peer is opt-in).
      fprintf(stderr, "isca peer comm (rank %d): fine-grained device memory for the flag block was refused; falling back to coarse-grained memory, "
                      "which is only valid for ranks sharing one device -- use RCCL (the default) across GPUs\n", rank);
    }
    pk(hipMemset(flags_, 0, sizeof(PeerFlags)), "hipMemset (flags)");
    pk(hipHostMalloc((void **)&err_, sizeof(int), hipHostMallocMapped), "hipHostMalloc");
    *err_ = 0;
    for (int q = 0; q < kMaxPeers; ++q) { r_flags_[q] = nullptr; for (auto &b : r_buf_) b[q] = nullptr; }
    if (world == 1) unlink(id.path);                        // (nothing to set up with anybody)
  }
  ~PeerComm() override {
    (void)hipDeviceSynchronize();
    for (int q = 0; q < world_; ++q) {
      if (q == rank_) continue;
      for (auto &b : r_buf_) if (b[q]) (void)hipIpcCloseMemHandle(b[q]);
      if (r_flags_[q]) (void)hipIpcCloseMemHandle(r_flags_[q]);
    }
    if (flags_) (void)hipFree(flags_);
    if (err_) (void)hipHostFree(err_);
    if (hdr_) munmap(hdr_, sizeof(SetupHeader));
  }
  const char *kind() const override { return "peer"; }

  // collective: the receive buffers of the sharded step (each the base of an allocation of its own), exported to and opened from every peer
  void attach(double *recv_fwd, double *recv_inv, double *recv_halo, size_t halo_half) override {
    local_[0] = recv_fwd; local_[1] = recv_inv; local_[2] = recv_halo; halo_half_ = halo_half;
    hdr_->device[rank_] = dev_;
    for (int b = 0; b < 3; ++b)
      if (local_[b]) pk(hipIpcGetMemHandle(&hdr_->handle[rank_][b], local_[b]), "hipIpcGetMemHandle (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
    if (hipIpcGetMemHandle(&hdr_->handle[rank_][3], flags_) != hipSuccess) {       // (a runtime that does not export fine-grained memory: plain device memory)
      (void)hipGetLastError();
      (void)hipFree(flags_); flags_ = nullptr;
      pk(hipMalloc((void **)&flags_, sizeof(PeerFlags)), "hipMalloc (flags)");
      pk(hipMemset(flags_, 0, sizeof(PeerFlags)), "hipMemset (flags)");
      pk(hipIpcGetMemHandle(&hdr_->handle[rank_][3], flags_), "hipIpcGetMemHandle (flags)");
    }
    host_barrier("attach: handles out");
    for (int q = 0; q < world_; ++q) {
      if (q == rank_) { for (int b = 0; b < 3; ++b) r_buf_[b][q] = local_[b]; r_flags_[q] = flags_; continue; }
      if (hdr_->device[q] != dev_) {                        // another GPU of the node: its memory over xGMI
        const hipError_t e = hipDeviceEnablePeerAccess(hdr_->device[q], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) pk(e, "hipDeviceEnablePeerAccess");
        (void)hipGetLastError();
      }
      for (int b = 0; b < 3; ++b)
        if (local_[b]) pk(hipIpcOpenMemHandle((void **)&r_buf_[b][q], hdr_->handle[q][b], hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle");
      pk(hipIpcOpenMemHandle((void **)&r_flags_[q], hdr_->handle[q][3], hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle (flags)");
    }
    host_barrier("attach: handles in");
    if (rank_ == 0) unlink(id_.path);
    attached_ = true;
  }

  void all_to_all(const double *send, double *recv, size_t count, hipStream_t s) override {
    if (world_ == 1) { pk(hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, s), "copy"); return; }
    const int b = slot_of(recv);
    PeerArgs a = base_args(b);
    a.npeers = world_;
    for (int q = 0; q < world_; ++q) {
      a.peers[q] = q; a.send[q] = send + (size_t)q * count; a.count[q] = count;
      a.dst[q] = r_buf_[b][q] + (size_t)rank_ * count;
    }
    launch(a, count, s);
  }
  void halo(const double *send_lo, const double *send_hi, double *recv_lo, double *recv_hi, size_t count, hipStream_t s) override {
    if (world_ == 1 || count == 0) return;
    need_attached();
    if (recv_lo != local_[2] || recv_hi != local_[2] + halo_half_ || count > halo_half_)
      throw std::runtime_error("peer comm: halo rows must arrive in the registered buffer");
    PeerArgs a = base_args(2);
    int n = 0;
    if (rank_ > 0) { a.peers[n] = rank_ - 1; a.send[n] = send_lo; a.count[n] = count; a.dst[n] = r_buf_[2][rank_ - 1] + halo_half_; ++n; }      // its rows from above
    if (rank_ < world_ - 1) { a.peers[n] = rank_ + 1; a.send[n] = send_hi; a.count[n] = count; a.dst[n] = r_buf_[2][rank_ + 1]; ++n; }        // its rows from below
    a.npeers = n;
    if (n) launch(a, count, s);
  }
  void all_to_all_with_halo(const double *send, double *recv, size_t count, const double *send_lo, const double *send_hi, double *recv_lo,
                            double *recv_hi, size_t halo_count, hipStream_t s) override {
    all_to_all(send, recv, count, s);
    halo(send_lo, send_hi, recv_lo, recv_hi, halo_count, s);
  }
  void all_reduce_sum(double *buf, size_t count, hipStream_t s) override {
    if (world_ == 1) return;
    need_attached();
    if (count > (size_t)kRedMax) throw std::runtime_error("peer comm: all-reduce of more than 16 values");
    PeerArgs a = base_args(3);
    a.red = buf; a.count_red = (int)count;
    hipLaunchKernelGGL(k_peer_all_reduce, dim3(1), dim3(64), 0, s, a);
    pk(hipGetLastError(), "launch (all-reduce)");
  }
  void abort() noexcept override { if (hdr_) { uint32_t none = 0; hdr_->aborted.compare_exchange_strong(none, (uint32_t)rank_ + 1); } }
  void check() override {                                   // at a synchronisation point: did a wait inside an exchange give up?
    if (err_ && *err_) {
      static const char *const names[] = {"lat -> m all-to-all", "m -> lat all-to-all", "halo exchange", "all-reduce"};
      const int k = *err_ - 1;
      throw std::runtime_error(std::string("peer comm: a peer did not arrive in the ") + names[k < 0 || k > 3 ? 0 : k] + " within ISCA_PEER_TIMEOUT_S");
    }
  }

 private:
  void need_attached() const { if (!attached_) throw std::runtime_error("peer comm: the receive buffers have not been attached (isca_dyn_comm_init does it)"); }
  int slot_of(const double *recv) const {
    need_attached();
    for (int b = 0; b < 2; ++b) if (recv == local_[b]) return b;
    throw std::runtime_error("peer comm: an all-to-all must arrive in one of the two registered Fourier buffers");
  }
  PeerArgs base_args(int kind) {
    PeerArgs a;
    std::memset(&a, 0, sizeof(a));
    a.me = rank_; a.world = world_; a.kind = kind; a.seq = ++seq_[kind];
    for (int q = 0; q < world_; ++q) a.flags[q] = r_flags_[q];
    a.err = err_;
    a.timeout_ticks = (long long)(timeout_s_ * 1.0e8);
    return a;
  }
  void launch(const PeerArgs &a, size_t count, hipStream_t s) {
    // slices per peer: enough blocks to move a block at link / HBM rate, few enough that every block of every rank is resident at once (they wait
    // for each other): 8 peers x 16 slices = 128 blocks of 256 threads
    int nb = (int)std::min<size_t>(16, std::max<size_t>(1, count / 4096));
    hipLaunchKernelGGL(k_peer_exchange, dim3(nb, a.npeers), dim3(256), 0, s, a);
    pk(hipGetLastError(), "launch (exchange)");
  }
  void host_barrier(const char *where) {
    const uint32_t gen = hdr_->generation.load(std::memory_order_acquire);
    if (hdr_->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world_) {
      hdr_->arrived.store(0, std::memory_order_relaxed);
      hdr_->generation.store(gen + 1, std::memory_order_release);
      return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0; hdr_->generation.load(std::memory_order_acquire) == gen; ++spin) {
      const uint32_t ab = hdr_->aborted.load(std::memory_order_acquire);
      if (ab) throw std::runtime_error(std::string("peer comm: rank ") + std::to_string(ab - 1) + " stopped with an error (" + where + ")");
      if (spin < 200) std::this_thread::yield();
      else std::this_thread::sleep_for(std::chrono::microseconds(50));
      if ((spin & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s_) {
        abort();
        throw std::runtime_error(std::string("peer comm: timed out waiting for the other ranks (") + where + ")");
      }
    }
  }

  PeerId id_;
  SetupHeader *hdr_ = nullptr;
  PeerFlags *flags_ = nullptr, *r_flags_[kMaxPeers];
  double *local_[3] = {nullptr, nullptr, nullptr}, *r_buf_[3][kMaxPeers];
  size_t halo_half_ = 0;
  int *err_ = nullptr, dev_ = 0;
  bool attached_ = false;
  unsigned long long seq_[kKinds] = {0, 0, 0, 0};
  double timeout_s_ = 60.0;
};

}  // namespace

bool peer_id_requested() {
  const char *v = getenv("ISCA_COMM");
  return v && std::strcmp(v, "peer") == 0;
}
void peer_unique_id(void *id128) {
  PeerId id;
  std::memset(&id, 0, sizeof(id));
  std::memcpy(id.magic, kPeerMagic, sizeof(kPeerMagic));
  const char *dir = getenv("ISCA_IPC_DIR");
  snprintf(id.path, sizeof(id.path), "%s/isca_peer_%d_%llx", dir && *dir ? dir : "/tmp", (int)getpid(),
           (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
  munmap(map_setup(id.path, true), sizeof(SetupHeader));
  std::memcpy(id128, &id, sizeof(id));
}
bool is_peer_id(const void *id128) { return std::memcmp(id128, kPeerMagic, sizeof(kPeerMagic)) == 0; }
Comm *make_peer_comm(const void *id128, int rank, int world) {
  PeerId id;
  std::memcpy(&id, id128, sizeof(id));
  id.path[sizeof(id.path) - 1] = 0;
  return new PeerComm(id, rank, world);
}

}  // namespace isca
