/*
 * ann_codec.h -- the reference's wire and on-disk formats either side of the hot path, host only.
 *
 * Everything here is Apache Thrift's TBinaryProtocol (big-endian; field = type byte, i16 id, value; struct ends with a
 * 0 byte; string/binary = i32 length + bytes; list = element type byte, i32 size; unknown fields are skipped), the
 * protocol the reference's own (de)serialisers name:
 *   ann/src/main/java/com/twitter/ann/hnsw/HnswIndexIOUtil.java:55,66,81,122      TSerializer / TBinaryProtocol
 *   ann/src/main/scala/com/twitter/ann/serialization/ThriftIteratorIO.scala:19,44  back-to-back structs until END_OF_FILE
 * org.apache.thrift itself is a third-party dependency that is not in the tree; the encoding is its published
 * specification, PARITY UNPINNED (the tree holds no serialised fixture).  Structures:
 *   simclusters-ann/thrift/src/main/thrift/simClustersAnn.thrift:8-27     Query, SimClustersANNTweetCandidate,
 *                                                                         SimClustersANNConfig, getTweetCandidates :49-57
 *   src/thrift/com/twitter/simclusters_v2/identifier.thrift               SimClustersEmbeddingId, InternalId (union)
 *   ann/src/main/thrift/com/twitter/ann/common/ann_common.thrift:65-83    HnswIndexMetadata, HnswInternalIndexMetadata,
 *                                                                         HnswGraphEntry;  :118-144 NearestNeighborResult
 * Not covered: `hnsw_embedding_mapping` and NearestNeighborQuery carry com/twitter/ml/api/embedding.thrift's Embedding,
 * an IDL that is not in the tree -- vectors are handed to hnsw_index_load_directory as a flat array instead; and
 * `hnsw_index_metadata` is written by mediaservices' ThriftByteBufferCodec (not in the tree): TBinaryProtocol assumed.
 * Keys of long-keyed indexes are AnnInjections.LongInjection = 8 bytes big-endian (ann/.../common/AnnInjections.scala:8).
 *
 * All functions return 0 or a negative ANNC_* code; ann_codec_last_error() gives the text (thread-local).
 */
#ifndef ANN_CODEC_H_
#define ANN_CODEC_H_

#include <stdint.h>

#include "hnsw_ann.h"
#include "simclusters_ann.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ANNC_OK 0
#define ANNC_EINVAL -1    /* bad argument */
#define ANNC_ETRUNC -2    /* input ends inside a value */
#define ANNC_EFORMAT -3   /* not what the IDL says (wrong type for a known field, required field missing, ...) */
#define ANNC_ESPACE -4    /* output buffer too small; *len tells the size needed where applicable */
#define ANNC_EIO -5       /* file could not be read / written */
#define ANNC_ENOMEM -6    /* host allocation failed */
#define ANNC_EINTERNAL -7 /* an unexpected C++ exception was caught at the ABI */

const char *ann_codec_last_error(void);

/* ---- simClustersAnn.thrift ----------------------------------------------------------------------------------------- */
/* InternalId is a union: `kind` is the field id set (1 tweetId, 2 userId, 3 entityId, 5 clusterId carry `value`; the
 * string and struct variants 4, 6-11 are kept as their encoded VALUE bytes -- raw points into the decoded buffer). */
typedef struct sann_wire_query {
  int32_t embedding_type;  /* identifier.thrift EmbeddingType */
  int32_t model_version;   /* online_store.thrift ModelVersion */
  int32_t internal_id_kind;
  int32_t internal_id_type; /* thrift type byte of the variant (10 i64, 8 i32, 11 string, 12 struct) */
  int64_t internal_id_value;
  const uint8_t *internal_id_raw;
  int64_t internal_id_raw_len;
  sann_config_t config;    /* field for field, simClustersAnn.thrift:18-27 */
} sann_wire_query_t;

/* a Query struct (no message envelope) */
int sann_wire_encode_query(const sann_wire_query_t *q, uint8_t *buf, int64_t cap, int64_t *len);
int sann_wire_decode_query(const uint8_t *buf, int64_t n, sann_wire_query_t *q, int64_t *consumed);
/* the value of a list<SimClustersANNTweetCandidate> (element type byte, size, structs) */
int sann_wire_encode_candidates(int32_t count, const int64_t *tweet_ids, const double *scores, uint8_t *buf, int64_t cap, int64_t *len);
int sann_wire_decode_candidates(const uint8_t *buf, int64_t n, int32_t cap, int64_t *tweet_ids, double *scores, int32_t *count,
                                int64_t *consumed);
/* SimClustersANNService.getTweetCandidates over a strict TBinaryProtocol message: CALL "getTweetCandidates" seqid
 * {1: Query} and REPLY seqid {0: list<SimClustersANNTweetCandidate>}.  (Finagle's TTwitter upgrade headers are
 * control plane and not produced or accepted.) */
int sann_wire_encode_call(int32_t seqid, const sann_wire_query_t *q, uint8_t *buf, int64_t cap, int64_t *len);
int sann_wire_decode_call(const uint8_t *buf, int64_t n, int32_t *seqid, sann_wire_query_t *q, int64_t *consumed);
int sann_wire_encode_reply(int32_t seqid, int32_t count, const int64_t *tweet_ids, const double *scores, uint8_t *buf, int64_t cap,
                           int64_t *len);
int sann_wire_decode_reply(const uint8_t *buf, int64_t n, int32_t *seqid, int32_t cap, int64_t *tweet_ids, double *scores,
                           int32_t *count, int64_t *consumed);

/* ---- ann_common.thrift: the files of an HNSW index directory ------------------------------------------------------- */
typedef struct hnsw_internal_metadata {  /* HnswInternalIndexMetadata, ann_common.thrift:71-77 */
  int32_t max_level;
  int32_t has_entry_point;  /* field 2 is optional: absent for an empty index (HnswIndexIOUtil.java:51-53) */
  int64_t entry_point;      /* LongInjection: 8 bytes big-endian */
  int32_t ef_construction;
  int32_t max_m;
  int32_t num_elements;     /* number of graph ENTRIES (HnswIndex.java:630-639), not of vectors */
} hnsw_internal_metadata_t;
int hnsw_codec_encode_internal_metadata(const hnsw_internal_metadata_t *m, uint8_t *buf, int64_t cap, int64_t *len);
int hnsw_codec_decode_internal_metadata(const uint8_t *buf, int64_t n, hnsw_internal_metadata_t *m);
/* HnswIndexMetadata, ann_common.thrift:65-69; distance_metric is the thrift enum (L2 0, Cosine 1, InnerProduct 2) */
int hnsw_codec_encode_index_metadata(int32_t dimension, int32_t distance_metric, int32_t num_elements, uint8_t *buf, int64_t cap,
                                     int64_t *len);
int hnsw_codec_decode_index_metadata(const uint8_t *buf, int64_t n, int32_t *dimension, int32_t *distance_metric,
                                     int32_t *num_elements);
/* hnsw_internal_graph: HnswGraphEntry structs back to back until the end of the input (HnswIndexIOUtil.java:76-104,
 * 111-132).  Entries in the flat form hnsw_index_build takes (keys and neighbours as longs). */
int hnsw_codec_encode_graph(int64_t n_entries, const int32_t *entry_level, const int64_t *entry_key, const int64_t *entry_offsets,
                            const int64_t *entry_neighbours, uint8_t *buf, int64_t cap, int64_t *len);
/* pass NULL arrays to size: *n_entries / *n_neighbours are always written */
int hnsw_codec_decode_graph(const uint8_t *buf, int64_t n, int64_t cap_entries, int64_t cap_neighbours, int32_t *entry_level,
                            int64_t *entry_key, int64_t *entry_offsets, int64_t *entry_neighbours, int64_t *n_entries,
                            int64_t *n_neighbours);
/* NearestNeighborResult (ann_common.thrift:118-144): ids as long keys, distances typed by the metric's union arm
 * (Cosine -> 1 cosineDistance, L2 -> 2 l2Distance, InnerProduct -> 3 innerProductDistance); with_distance 0 omits them */
int ann_wire_encode_neighbor_result(int32_t distance_metric, int32_t count, const int64_t *ids, const float *distances,
                                    int32_t with_distance, uint8_t *buf, int64_t cap, int64_t *len);
int ann_wire_decode_neighbor_result(const uint8_t *buf, int64_t n, int32_t cap, int64_t *ids, double *distances, int32_t *arms,
                                    int32_t *count, int64_t *consumed);

/* ---- a whole index directory (SerializableHnsw.scala:170-190, HnswCommon.scala:16-20,42-50) ------------------------ */
/* Writes <dir>/hnsw_index_metadata, <dir>/hnsw_internal_index/{hnsw_internal_metadata, hnsw_internal_graph} and
 * <dir>/_SUCCESS from a built or loaded index (keys = the index's ids, or positions when it has none). */
int hnsw_index_save_directory(const hnsw_index_t *index, int32_t ef_construction, const char *dir);
/* Reads the same three files and builds a searchable index over `vectors` (row i belongs to key ids[i], or to key i
 * when ids is NULL).  dimension / metric must agree with hnsw_index_metadata, as SerializableHnsw.validateMetadata
 * demands (:84-101). */
int hnsw_index_load_directory(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids,
                              const char *dir, hnsw_index_t **out);

#ifdef __cplusplus
}
#endif
#endif
