// sann_device.h -- data layout shared by the host side and the gfx950 kernels.
//
// HBM layout of a cluster -> top-tweets index shard (see DESIGN.md "Data layout"):
//
//   postings[n_postings]        {int64 tweet_id; double score}  16 B, AoS so that one lane = one
//                               global_load_dwordx4 and a wave instruction covers 1 KiB contiguous
//   ranks[n_postings]           uint32: position of the posting in the cluster's full list as the
//                               store returned it (the `i` of ApproximateCosineSimilarity.scala:87)
//   sub_offsets[n_rows*P + 1]   uint32 CSR: sub-list (row, p) = postings of cluster `row` whose
//                               tweet hashes to partition p (and to this shard), in rank order
//
// Every posting of one tweet lives in the same (shard, partition), so a (query, partition) work
// unit can aggregate, normalise and select on its own; partitions and shards are merged by an
// exact top-k merge (the ComposedQueryable pattern, ann/.../common/ShardApi.scala:71-87).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sann {

struct Posting {
  int64_t id;
  double score;
};

// Prepared query (host side applies the SimClustersEmbedding / fetchCandidates semantics).
struct QueryHdr {
  int64_t src_excl;   // tweet id to exclude when excl_enabled
  int64_t earliest;   // ApproximateCosineSimilarity.scala:66-70
  int64_t latest;     // :71-72
  double l2norm;      // sourceEmbedding.l2norm   (full embedding, SimClustersEmbedding.scala:61)
  double lognorm;     // sourceEmbedding.logNorm  (:63)
  double min_score;   // config.minScore
  int32_t scan_begin; // offset into scan_row / scan_w
  int32_t n_scan;     // clusters to scan, in accumulation order
  int32_t M;          // max(maxTopTweetsPerCluster, 0)
  int32_t k;          // min(max(maxNumResults,0), 1000)
  int32_t alg;        // SANN_ALG_* (1..4; the offline forms 5 / 6 arrive as 3 / 4 with use_norms)
  int32_t excl_enabled;
  // offline job (scio/bq_generation/sql/tweets_ann.sql:10-15,50-51): the candidate's normaliser is its tweet's FULL
  // embedding norm -- the per-posting norms column of the index -- instead of the sum of squares over scanned clusters
  int32_t use_norms;
  // (float)(1 / l2norm), (float)(1 / lognorm): what the unit kernel's fp32 pre-filter multiplies by -- worked out once per
  // query by the preparation instead of by every wave of every unit (an fp64 division: a dozen instructions)
  float inv_l2_32;
  float inv_ln_32;
  int32_t reserved;
};
static_assert(sizeof(QueryHdr) == 88, "QueryHdr layout");

struct IndexView {
  const Posting *postings;
  const uint32_t *ranks;
  const uint32_t *sub_offsets;
  const double *norms;  // [n_postings] sum of squares of the posting's tweet's FULL embedding, or NULL (online index)
  int32_t n_rows;
  int32_t P;      // partitions (power of two)
  int32_t log2P;
  uint32_t n_postings;  // postings held (< 2^32 per shard): device-written positions are checked against it before use
};

// entries of a unit's descriptor row (BatchView::desc): 64 while every query of the batch scans <= 64 clusters, else 128
__host__ __device__ constexpr int desc_row_stride(int max_n_scan) { return max_n_scan <= 64 ? 64 : 128; }

// A fast unit's candidate list is handed to the merge kernel in one of three forms per entry (cand_key / cand_id):
//   key >= 2           final: (score_key, tweet id)
//   key == CAND_DEFERRED   id = cluster sequence number << 32 | posting position: the merge kernel fetches the posting and
//                          does the exact fp64 arithmetic (ApproximateCosineSimilarity.scala:92-125) itself
//   key == CAND_DROPPED    nothing (a candidate that failed `score >= minScore`)
// (score_key never yields 0 or 1 for a score that passed `>= minScore`: they are the images of two negative NaNs.)
constexpr unsigned long long CAND_DEFERRED = 0ull, CAND_DROPPED = 1ull;

// Unit flags
enum : uint32_t {
  UNIT_OK = 0,
  UNIT_OVERFLOW = 1,   // fast path could not hold the unit (table, dup list or scan list too big)
  UNIT_TRUNCATED = 2,  // unit withheld qualifying candidates: all of them have key < unit_thr
};

struct BatchView {
  const QueryHdr *hdr;
  const int32_t *scan_row;
  const double *scan_w;
  // [nq] per query, written by the merge kernel when not NULL: x = largest unit_T, y = sum of unit_T (postings with
  // rank < M scanned in this shard), z = scanned clusters, w = 0.  Lets the host learn a batch's shape without
  // having prepared it (device-side preparation, sann_prep.hip).
  uint4 *q_stat;
  // Descriptor rows at a FIXED stride: unit u's row is desc_stride (64 or 128) pairs (sub-list start, exclusive prefix
  // of the lengths) at desc + 2 * u * desc_stride, padded behind the query's n_scan clusters with (0, unit_T); the
  // query's cluster weights lie at scan_wq + q * desc_stride, padded with 0.  The unit kernel finds both from its block
  // index alone -- the row at a compact offset needed the query header first: one more dependent trip to memory.
  uint32_t *desc;
  double *scan_wq;
  int32_t desc_stride;
  int32_t *unit_T;          // [n_units] postings with rank < M the unit scans
  uint32_t *unit_pre;       // [n_units] cosine forms: the cluster-level cut of the unit (fp32 key; 0 = none), from the descriptor kernel
  const uint32_t *cut[4];   // cached cut tables ([n_rows*P], see sann_index::cut_cache) ...
  int32_t cut_M[4];         // ... for these values of M (-1 = unused slot)
  int32_t nq;
  int32_t cap;              // entries per unit in cand_key/cand_id (fast path)
  int32_t cap2;             // entries per unit in cand_key2/cand_id2 (general path, >= max k)
  // per unit candidate lists.  Unit u's list is at cand_*[u*cap] when unit_fb[u] < 0, else at
  // cand_*2[unit_fb[u]*cap2].
  uint64_t *cand_key;       // monotone score key
  int64_t *cand_id;
  uint64_t *cand_key2;
  int64_t *cand_id2;
  int32_t *unit_fb;         // [n_units]
  int32_t *cand_cnt;        // [n_units]
  int32_t *unit_unique;     // [n_units] distinct tweets accumulated (candidateScoresMap.size share)
  uint32_t *unit_flags;     // [n_units]
  uint64_t *unit_thr;       // [n_units*2] (hi, lo): every withheld candidate has key < thr
  // batch status: [0] = number of overflowed units, [1] = number of inexact queries,
  // [2..2+nq) = list of inexact queries
  int32_t *status;
  int32_t *overflow_units;  // [n_units] list of overflowed unit ids (first status[0] entries)
  // final per-query outputs
  int64_t *out_ids;         // [nq*stride]
  double *out_scores;       // [nq*stride]
  int32_t *out_counts;      // [nq]
  int32_t *out_map_sizes;   // [nq]
  int32_t stride;
  // query q's outputs live in chunk q / out_chunk_q, out_chunk_pitch bytes after the previous chunk's, at
  // position q % out_chunk_q within it (all four arrays; default: one chunk)
  int32_t out_chunk_q;
  int64_t out_chunk_pitch;
  // debug only: per-unit s_memtime stamps at phase boundaries ([n_units*16]); NULL in normal runs
  unsigned long long *prof;
  // [nq] scratch of the merge launch: 1 = merge_wave_kernel finished the query, the workgroup-per-query kernel skips it
  int32_t *merge_done;
};

// Workspace of the general (global-memory table) path, one region per listed unit.
struct GeneralWs {
  const int32_t *units;     // unit ids to process (NULL = identity)
  const int64_t *ws_off;    // [n] entry offset of the unit's table
  const uint32_t *ws_slots; // [n] power-of-two slot count S (table has S+1 entries)
  int64_t *keys;
  double *dot;
  double *nsq;
};

constexpr int64_t kEmptyKey = -1;  // table sentinel (memset 0xFF); a real tweet id of -1 uses slot S

__host__ __device__ inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}
// tweet -> (shard, partition).  Shard from the high bits, partition from the low bits.
__host__ __device__ inline uint32_t tweet_shard(uint64_t h, uint32_t n_shards) { return (uint32_t)((h >> 40) % n_shards); }
__host__ __device__ inline uint32_t tweet_partition(uint64_t h, uint32_t P) { return (uint32_t)(h & (P - 1)); }
__host__ __device__ inline uint32_t tweet_slot(uint64_t h) { return (uint32_t)(h >> 12); }
// Cheap in-table hash (the partition hash above is fixed at index build; this one only places a
// key inside one unit's table): fold to 32 bits, Fibonacci multiply, take the top bits.
__host__ __device__ inline uint32_t table_hash(int64_t id, int log2S) {
  uint32_t x = (uint32_t)((uint64_t)id ^ ((uint64_t)id >> 32));
  x ^= x >> 15;
  return (x * 0x9E3779B1u) >> (32 - log2S);
}

}  // namespace sann
