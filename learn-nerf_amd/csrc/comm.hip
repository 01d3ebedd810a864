// comm.hip — the data-parallel exchange step of the C ABI: one RCCL all-reduce (sum) of the
// flat fp32 gradient per training step.  The reference is single-device (train.py:78-112 has no collective); this
// is the exchange that the data-parallel form of TrainLoop.step_fn inserts between jax.grad and optax.adam.
// The communicator is the only persistent object the library hands out, as an opaque handle.
#include <rccl/rccl.h>
#include <string.h>

#include <new>

#include "common.h"

struct lnrf_comm {
  ncclComm_t comm;
  int rank, world;
};

namespace lnrf {
static int nccl_fail(ncclResult_t r, const char* what) {
  set_error("%s: RCCL error %d (%s)", what, (int)r, ncclGetErrorString(r));
  return (int)r;  // > 0: ncclResult_t passthrough
}
}  // namespace lnrf

using namespace lnrf;

static_assert(LNRF_COMM_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

extern "C" int lnrf_comm_get_unique_id(void* unique_id) {
  LNRF_CHECK_ARG(unique_id, "null pointer");
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId");
  memcpy(unique_id, id.internal, NCCL_UNIQUE_ID_BYTES);
  return LNRF_OK;
}

extern "C" int lnrf_comm_init(const void* unique_id, int32_t rank, int32_t world, lnrf_comm_t* comm_out) {
  LNRF_CHECK_ARG(unique_id && comm_out, "null pointer");
  LNRF_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "need 0 <= rank < world");
  ncclUniqueId id;
  memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t c;
  ncclResult_t r = ncclCommInitRank(&c, world, id, rank);  // binds to the calling thread's current HIP device
  if (r != ncclSuccess) return nccl_fail(r, "ncclCommInitRank");
  lnrf_comm* h = new (std::nothrow) lnrf_comm{c, rank, world};
  if (!h) {
    ncclCommDestroy(c);
    set_error("lnrf_comm_init: out of host memory");
    return LNRF_ERR_ARG;
  }
  *comm_out = h;
  return LNRF_OK;
}

extern "C" int lnrf_comm_allreduce(lnrf_comm_t comm, float* buf, int64_t n, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(comm, "null communicator");
  LNRF_CHECK_ARG(n >= 0, "bad n");
  if (n == 0) return LNRF_OK;
  LNRF_CHECK_ARG(buf, "null pointer");
  ncclResult_t r = ncclAllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, comm->comm, as_stream(stream));
  if (r != ncclSuccess) return nccl_fail(r, "ncclAllReduce");
  return LNRF_OK;
}

extern "C" int lnrf_comm_info(lnrf_comm_t comm, int32_t* rank, int32_t* world) {
  LNRF_CHECK_ARG(comm, "null communicator");
  if (rank) *rank = comm->rank;
  if (world) *world = comm->world;
  return LNRF_OK;
}

extern "C" int lnrf_comm_destroy(lnrf_comm_t comm) {
  if (!comm) return LNRF_OK;
  ncclResult_t r = ncclCommDestroy(comm->comm);
  delete comm;
  if (r != ncclSuccess) return nccl_fail(r, "ncclCommDestroy");
  return LNRF_OK;
}
