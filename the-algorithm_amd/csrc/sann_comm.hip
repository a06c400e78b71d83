// sann_comm.hip -- the multi-GPU exchange step of the sharded SimClusters-ANN path, over RCCL (xGMI inside a node).
//
// One process per GPU; GPU g holds the tweet-hash shard g of every posting list (sann_index_options_t). Every rank
// answers the WHOLE batch on its shard, the merge kernel writes query q's per-shard result into the packed message of
// q's owner rank (sann_batch_bind_outputs_chunked), and this file moves the messages: ONE all-to-all per batch,
// expressed as a group of point-to-point ncclSend / ncclRecv -- xGMI is point-to-point, and an all-to-all is exactly
// one direct transfer per pair of GPUs.  The owner then merges the N per-shard lists exactly
// (sann_merge_shards / sann_merge_shards_cut): the reference's shard pattern, ComposedQueryable
// (ann/src/main/scala/com/twitter/ann/common/ShardApi.scala:71-87: query every shard, concatenate, sort, take k),
// which is exact here because all of a tweet's postings live in one shard.
//
// Nothing in this file needs Python or torch: a worker calls sann_comm_unique_id on rank 0, ships the 128 bytes to the
// other ranks by whatever control channel it has, and every rank calls sann_comm_create.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>
#include <string>

#include "../../include/simclusters_ann.h"
#include "sann_host.h"

using sann_host::fail;
#include "abi_guard.h"
#define ABI_CATCH catch (...) { return abi_guard::caught(sann_host::fail, SANN_ENOMEM, SANN_EINTERNAL); }

struct sann_comm {
  ncclComm_t comm = nullptr;
  int device = 0, rank = 0, world = 1;
};

#define NCCL_TRY(expr)                                                                                              \
  do {                                                                                                              \
    ncclResult_t r_ = (expr);                                                                                       \
    if (r_ != ncclSuccess) return fail(SANN_EDEVICE, std::string(#expr) + ": " + ncclGetErrorString(r_));            \
  } while (0)

extern "C" {

int sann_comm_unique_id(void *id128) try {
  if (!id128) return fail(SANN_EINVAL, "id128 is NULL");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  NCCL_TRY(ncclGetUniqueId(&id));
  memcpy(id128, &id, sizeof id);
  return SANN_OK;
} ABI_CATCH

int sann_comm_create(int32_t device, int32_t rank, int32_t world, const void *id128, sann_comm_t **out) try {
  if (!out) return fail(SANN_EINVAL, "out is NULL");
  *out = nullptr;
  if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(SANN_EINVAL, "bad rank / world / id");
  HIP_TRY(hipSetDevice(device));
  sann_comm *c = new (std::nothrow) sann_comm();
  if (!c) return fail(SANN_ENOMEM, "out of host memory");
  c->device = device;
  c->rank = rank;
  c->world = world;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(SANN_EDEVICE, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  }
  *out = c;
  return SANN_OK;
} ABI_CATCH

int sann_comm_info(const sann_comm_t *c, int32_t *rank, int32_t *world) try {
  if (!c) return fail(SANN_EINVAL, "comm is NULL");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return SANN_OK;
} ABI_CATCH

int sann_comm_destroy(sann_comm_t *c) try {
  if (!c) return SANN_OK;
  (void)hipSetDevice(c->device);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  delete c;
  return SANN_OK;
} ABI_CATCH

int sann_exchange_to_owners(sann_comm_t *c, void *hip_stream, const void *d_send, void *d_recv, int64_t chunk_bytes) try {
  if (!c) return fail(SANN_EINVAL, "comm is NULL");
  if (chunk_bytes < 0 || (chunk_bytes > 0 && (!d_send || !d_recv))) return fail(SANN_EINVAL, "bad buffers");
  if (chunk_bytes == 0) return SANN_OK;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)hip_stream;
  const char *s = (const char *)d_send;
  char *r = (char *)d_recv;
  // chunk `peer` of the send buffer goes to rank `peer`; chunk `peer` of the receive buffer comes from rank `peer`
  NCCL_TRY(ncclGroupStart());
  for (int peer = 0; peer < c->world; peer++) {
    ncclResult_t a = ncclSend(s + (int64_t)peer * chunk_bytes, (size_t)chunk_bytes, ncclUint8, peer, c->comm, st);
    ncclResult_t b = a == ncclSuccess ? ncclRecv(r + (int64_t)peer * chunk_bytes, (size_t)chunk_bytes, ncclUint8, peer, c->comm, st) : a;
    if (b != ncclSuccess) {
      (void)ncclGroupEnd();
      return fail(SANN_EDEVICE, std::string("ncclSend/ncclRecv: ") + ncclGetErrorString(b));
    }
  }
  NCCL_TRY(ncclGroupEnd());
  return SANN_OK;
} ABI_CATCH

// The cluster-id-range deployment's exchange: postings (16 bytes each) to the GPU their tweet hashes to -- a grouped ncclSend /
// ncclRecv round with a different count per peer; both sides know the counts (they were exchanged as a fixed-size block first).
int sann_exchange_postings_by_tweet_hash(sann_comm_t *c, void *hip_stream, const void *d_send, const int64_t *send_counts, void *d_recv,
                                         const int64_t *recv_counts) try {
  if (!c || !send_counts || !recv_counts) return fail(SANN_EINVAL, "NULL argument");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)hip_stream;
  int64_t so = 0, ro = 0;
  for (int peer = 0; peer < c->world; peer++)
    if (send_counts[peer] < 0 || recv_counts[peer] < 0) return fail(SANN_EINVAL, "negative count");
  NCCL_TRY(ncclGroupStart());
  for (int peer = 0; peer < c->world; peer++) {
    ncclResult_t a = ncclSuccess;
    if (send_counts[peer] > 0) a = ncclSend((const char *)d_send + so * 16, (size_t)send_counts[peer] * 16, ncclUint8, peer, c->comm, st);
    if (a == ncclSuccess && recv_counts[peer] > 0) a = ncclRecv((char *)d_recv + ro * 16, (size_t)recv_counts[peer] * 16, ncclUint8, peer, c->comm, st);
    if (a != ncclSuccess) {
      (void)ncclGroupEnd();
      return fail(SANN_EDEVICE, std::string("ncclSend/ncclRecv: ") + ncclGetErrorString(a));
    }
    so += send_counts[peer];
    ro += recv_counts[peer];
  }
  NCCL_TRY(ncclGroupEnd());
  return SANN_OK;
} ABI_CATCH

int sann_owner_message_layout(int32_t queries_per_owner, int32_t stride, int64_t *chunk_bytes, int64_t *off_scores,
                              int64_t *off_counts, int64_t *off_map_sizes) try {
  if (queries_per_owner < 0 || stride < 1) return fail(SANN_EINVAL, "bad sizes");
  // [ids int64[n][stride] | score bits fp64[n][stride] | counts int32[n] | map sizes int32[n]]: a multiple of 8 bytes
  const int64_t arr = (int64_t)queries_per_owner * stride * 8;
  if (off_scores) *off_scores = arr;
  if (off_counts) *off_counts = 2 * arr;
  if (off_map_sizes) *off_map_sizes = 2 * arr + 4 * (int64_t)queries_per_owner;
  if (chunk_bytes) *chunk_bytes = 2 * arr + 8 * (int64_t)queries_per_owner;
  return SANN_OK;
} ABI_CATCH

}  // extern "C"
