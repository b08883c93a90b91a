// tsim_dist.hip - the one collective of the path: shots are sharded over the GPUs of a node with no
// data-path exchange, and only the finished detector/observable rows travel - an RCCL gather (or an
// all-to-all that spreads the roots) over xGMI, issued here, from the library, on a HIP stream the
// caller orders after the sampling kernels.  No PyTorch anywhere: the host process exchanges the 128-byte
// ncclUniqueId however it likes (tsim_amd/dist.py: a file or TCP rendezvous) and calls tsim_dist_init.
//
// The reference has no multi-device path at all (src/tsim/sampler.py:310 uses jax.devices()[0]); the
// partition is the north star's (BASELINE.json): shot dimension over the GPUs, gather of the bit strings.
#include "tsim_internal.hip.h"

#include <rccl/rccl.h>

struct tsim_dist {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = -1;
  hipStream_t stream = nullptr;   // own stream for the host-value helpers (barrier / max)
  double *d_scalar = nullptr;
  hipEvent_t marks[TSIM_DIST_MARKS] = {};  // tsim_dist_mark / tsim_dist_wait_mark
};

#define NCCL_TRY(expr)                                                                              \
  do {                                                                                              \
    ncclResult_t r_ = (expr);                                                                       \
    if (r_ != ncclSuccess) return tsim_fail(TSIM_EHIP, "%s failed: %s", #expr, ncclGetErrorString(r_)); \
  } while (0)

static_assert(NCCL_UNIQUE_ID_BYTES == TSIM_DIST_ID_BYTES, "unique id size");

extern "C" int tsim_dist_unique_id(uint8_t id[TSIM_DIST_ID_BYTES]) {
  if (!id) return tsim_fail(TSIM_EINVAL, "id is NULL");
  ncclUniqueId u;
  NCCL_TRY(ncclGetUniqueId(&u));
  memcpy(id, u.internal, TSIM_DIST_ID_BYTES);
  return TSIM_OK;
}

extern "C" int tsim_dist_init(int32_t device, const uint8_t id[TSIM_DIST_ID_BYTES], int32_t rank, int32_t world,
                              tsim_dist **out) {
  if (!out || !id) return tsim_fail(TSIM_EINVAL, "NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return tsim_fail(TSIM_EINVAL, "bad rank %d of %d", rank, world);
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return tsim_fail(TSIM_EINVAL, "device %d out of range (%d visible)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  tsim_dist *d = new (std::nothrow) tsim_dist();
  if (!d) return tsim_fail(TSIM_ENOMEM, "out of host memory");
  d->rank = rank;
  d->world = world;
  d->device = device;
  ncclUniqueId u;
  memcpy(u.internal, id, TSIM_DIST_ID_BYTES);
  ncclResult_t r = ncclCommInitRank(&d->comm, world, u, rank);
  if (r != ncclSuccess) {
    delete d;
    return tsim_fail(TSIM_EHIP, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
  }
  hipError_t e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc((void **)&d->d_scalar, 64);
  if (e == hipSuccess) e = hipMemset(d->d_scalar, 0, 64);
  if (e != hipSuccess) {
    tsim_dist_destroy(d);
    return tsim_fail(TSIM_EHIP, "tsim_dist_init: %s", hipGetErrorString(e));
  }
  *out = d;
  return TSIM_OK;
}

extern "C" void tsim_dist_destroy(tsim_dist *d) {
  if (!d) return;
  if (d->device >= 0) (void)hipSetDevice(d->device);
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  if (d->comm) (void)ncclCommDestroy(d->comm);
  if (d->d_scalar) (void)hipFree(d->d_scalar);
  for (hipEvent_t e : d->marks)
    if (e) (void)hipEventDestroy(e);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
}

extern "C" int tsim_dist_info(const tsim_dist *d, int32_t *rank, int32_t *world) {
  if (!d) return tsim_fail(TSIM_EINVAL, "communicator is NULL");
  if (rank) *rank = d->rank;
  if (world) *world = d->world;
  return TSIM_OK;
}

static int dist_ready(tsim_dist *d) {
  if (!d || !d->comm) return tsim_fail(TSIM_EINVAL, "communicator is NULL");
  HIP_TRY(hipSetDevice(d->device));
  return 0;
}

extern "C" int tsim_dist_gather_rows(tsim_dist *d, const void *d_send, int64_t nbytes, void *d_recv, int32_t root,
                                     void *stream) {
  if (int r = dist_ready(d)) return r;
  if (nbytes < 0 || root < 0 || root >= d->world) return tsim_fail(TSIM_EINVAL, "bad gather arguments");
  if (nbytes == 0) return TSIM_OK;
  if (!d_send || (d->rank == root && !d_recv)) return tsim_fail(TSIM_EINVAL, "NULL buffer");
  NCCL_TRY(ncclGather(d_send, d_recv, (size_t)nbytes, ncclUint8, root, d->comm, stream ? (hipStream_t)stream : d->stream));
  return TSIM_OK;
}

extern "C" int tsim_dist_alltoall_rows(tsim_dist *d, const void *d_send, void *d_recv, int64_t nbytes_per_peer, void *stream) {
  if (int r = dist_ready(d)) return r;
  if (nbytes_per_peer < 0) return tsim_fail(TSIM_EINVAL, "negative size");
  if (nbytes_per_peer == 0) return TSIM_OK;
  if (!d_send || !d_recv) return tsim_fail(TSIM_EINVAL, "NULL buffer");
  NCCL_TRY(ncclAllToAll(d_send, d_recv, (size_t)nbytes_per_peer, ncclUint8, d->comm, stream ? (hipStream_t)stream : d->stream));
  return TSIM_OK;
}

extern "C" int tsim_dist_allreduce_max(tsim_dist *d, double *value) {
  if (int r = dist_ready(d)) return r;
  if (!value) return tsim_fail(TSIM_EINVAL, "value is NULL");
  HIP_TRY(hipMemcpyAsync(d->d_scalar, value, 8, hipMemcpyHostToDevice, d->stream));
  NCCL_TRY(ncclAllReduce(d->d_scalar, d->d_scalar, 1, ncclDouble, ncclMax, d->comm, d->stream));
  HIP_TRY(hipMemcpyAsync(value, d->d_scalar, 8, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  return TSIM_OK;
}

extern "C" int tsim_dist_barrier(tsim_dist *d) {
  // an all-reduce of the device scalar as it is (its value is of no interest): no host copies either side
  if (int r = dist_ready(d)) return r;
  NCCL_TRY(ncclAllReduce(d->d_scalar, d->d_scalar, 1, ncclDouble, ncclMax, d->comm, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  return TSIM_OK;
}

extern "C" int tsim_dist_stream_wait(tsim_dist *d, void *waiting_stream, void *signalling_stream) {
  // order `waiting_stream` after the work queued so far on `signalling_stream` (NULL: the communicator's own)
  if (int r = dist_ready(d)) return r;
  hipEvent_t ev;
  HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, signalling_stream ? (hipStream_t)signalling_stream : d->stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(waiting_stream ? (hipStream_t)waiting_stream : d->stream, ev, 0);
  (void)hipEventDestroy(ev);
  if (e != hipSuccess) return tsim_fail(TSIM_EHIP, "tsim_dist_stream_wait: %s", hipGetErrorString(e));
  return TSIM_OK;
}

extern "C" int tsim_dist_mark(tsim_dist *d, int32_t mark, void *stream) {
  if (int r = dist_ready(d)) return r;
  if (mark < 0 || mark >= TSIM_DIST_MARKS) return tsim_fail(TSIM_EINVAL, "mark %d out of range", mark);
  if (!d->marks[mark]) HIP_TRY(hipEventCreateWithFlags(&d->marks[mark], hipEventDisableTiming));
  HIP_TRY(hipEventRecord(d->marks[mark], stream ? (hipStream_t)stream : d->stream));
  return TSIM_OK;
}

extern "C" int tsim_dist_wait_mark(tsim_dist *d, int32_t mark, void *stream) {
  if (int r = dist_ready(d)) return r;
  if (mark < 0 || mark >= TSIM_DIST_MARKS) return tsim_fail(TSIM_EINVAL, "mark %d out of range", mark);
  if (!d->marks[mark]) return TSIM_OK;  // never recorded: nothing to wait for
  HIP_TRY(hipStreamWaitEvent(stream ? (hipStream_t)stream : d->stream, d->marks[mark], 0));
  return TSIM_OK;
}

extern "C" int tsim_device_synchronize(int32_t device) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipDeviceSynchronize());
  return TSIM_OK;
}
