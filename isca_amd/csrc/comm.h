// RCCL (the ROCm NCCL) bound at run time with dlopen: the library has no link-time dependency on it, single-GPU use never
// loads it, and inside a PyTorch process the already loaded librccl.so.1 is the one that answers.
// Replaces, for the sharded step, mpp_transmit in transpose_fourier / reverse_transpose_fourier
// (atmos_spectral/tools/transforms.F90:990-1054), mpp_update_domains in fv_advection.F90:161-162,259 and the global
// sums behind area_weighted_global_mean (transforms.F90:1059-1077).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <cstddef>

namespace isca {

// Three implementations behind one interface, chosen by the 128-byte id the ranks share:
//  * RCCL (default): grouped ncclSend/ncclRecv + ncclAllReduce on the step's stream, one GPU per rank;
//  * "ipc" (ISCA_COMM=ipc when the id is drawn; comm_ipc.cpp): host-staged exchange through files mapped by every rank -- N processes
//    that may SHARE one GPU run the same C++ exchange schedule (isca_dyn_step's sharded loop) without RCCL, which refuses two ranks on
//    one device.  A verification vehicle for 1-GPU boxes, not a fast path: every exchange synchronises the stream and the host.
//  * "peer" (ISCA_COMM=peer; comm_peer.hip): the ranks of ONE node write into each other's receive buffers (hipIpc), one kernel per exchange, flags
//    in device memory instead of host hand-shakes.  Verified with N processes on one GPU; not yet run over xGMI.
class Comm {
 public:
  static constexpr int UNIQUE_ID_BYTES = 128;
  static void unique_id(void *id128);                                    // throws std::runtime_error
  static Comm *create(const void *id128, int rank, int world);           // collective over all ranks
  virtual ~Comm() {}
  int rank() const { return rank_; }
  int world() const { return world_; }
  virtual const char *kind() const = 0;
  // equal blocks: block p of `send` goes to rank p, block q of `recv` comes from rank q (count doubles each)
  virtual void all_to_all(const double *send, double *recv, size_t count, hipStream_t s) = 0;
  // rows for the neighbouring latitude bands: lo <-> rank-1, hi <-> rank+1 (no wrap-around)
  virtual void halo(const double *send_lo, const double *send_hi, double *recv_lo, double *recv_hi, size_t count, hipStream_t s) = 0;
  // both of the above in one group (RCCL: one fused send/recv kernel instead of two)
  virtual void all_to_all_with_halo(const double *send, double *recv, size_t count, const double *send_lo, const double *send_hi,
                                    double *recv_lo, double *recv_hi, size_t halo_count, hipStream_t s) = 0;
  virtual void all_reduce_sum(double *buf, size_t count, hipStream_t s) = 0;         // in place
  // this rank cannot go on (an exception on its way to the caller): peers blocked in an exchange stop with an error instead of waiting
  virtual void abort() noexcept {}
  // collective, once: the receive buffers of the sharded step -- the spectral side's Fourier buffer (lat -> m), the grid side's (m -> lat), the
  // tracer's halo rows (two halves of halo_half doubles: from below, from above; null without tracer) -- for an implementation that writes into
  // its peers' memory (comm_peer.hip); the others ignore it
  virtual void attach(double * /*recv_fwd*/, double * /*recv_inv*/, double * /*recv_halo*/, size_t /*halo_half*/) {}
  // at a host synchronisation point: an exchange that gave up on the device (a peer that never arrived) becomes the error here
  virtual void check() {}
  // the host's wait for the step's stream.  An exchange whose peer never joins would make hipStreamSynchronize wait for ever: an implementation
  // whose exchanges run on the device without a host hand-shake (RCCL) polls instead -- the stream, the communicator's asynchronous error state,
  // and a deadline (ISCA_EXCHANGE_TIMEOUT_S, default 120 s) after which it aborts the communicator and throws.  Default: hipStreamSynchronize.
  virtual void synchronize(hipStream_t s);
  // ISCA_FAULT_EXCHANGE="<rank>:<n>" (tests): that rank never joins its n-th exchange (counted per communicator, from 0) -- what its peers do about it
  // is the error path under test.  True when this call is the one to skip.
  bool fault_here();

 protected:
  Comm(int rank, int world) : rank_(rank), world_(world) {}
  int rank_, world_;
  long exchanges_ = 0;
};

// comm_peer.hip: ISCA_COMM=peer -- one kernel per exchange that stores into the peers' receive buffers (hipIpc-mapped), flags instead of host hand-shakes
bool peer_id_requested();
void peer_unique_id(void *id128);
bool is_peer_id(const void *id128);
Comm *make_peer_comm(const void *id128, int rank, int world);

// comm_ipc.cpp
bool ipc_id_requested();                       // ISCA_COMM=ipc in the environment of the rank that draws the id
void ipc_unique_id(void *id128);
bool is_ipc_id(const void *id128);
Comm *make_ipc_comm(const void *id128, int rank, int world);

}  // namespace isca
