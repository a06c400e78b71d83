// ann_codec.cpp -- Thrift TBinaryProtocol codecs for the structures either side of the hot path (include/ann_codec.h).
// Host only.  The protocol is restated from Apache Thrift's published specification (thrift-binary-protocol.md):
// org.apache.thrift is a third-party dependency of the reference and not in its tree.
#include "../../include/ann_codec.h"

#include <sys/stat.h>

#include <cerrno>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "abi_guard.h"
#define ABI_CATCH catch (...) { return abi_guard::caught(fail, ANNC_ENOMEM, ANNC_EINTERNAL); }

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

// TType
enum : uint8_t { T_STOP = 0, T_BOOL = 2, T_BYTE = 3, T_DOUBLE = 4, T_I16 = 6, T_I32 = 8, T_I64 = 10, T_STRING = 11, T_STRUCT = 12, T_MAP = 13, T_SET = 14, T_LIST = 15 };
// TMessageType
enum : int32_t { M_CALL = 1, M_REPLY = 2, M_EXCEPTION = 3 };
constexpr uint32_t VERSION_1 = 0x80010000u, VERSION_MASK = 0xffff0000u;

// ---- writer: counts always, stores while there is room (so a too-small buffer still learns the size) ----------------
struct W {
  uint8_t *p;
  int64_t cap, n = 0;
  W(uint8_t *buf, int64_t c) : p(buf), cap(buf ? c : 0) {}
  void u8(uint8_t v) { if (n < cap) p[n] = v; n++; }
  void i16(int16_t v) { u8((uint8_t)((uint16_t)v >> 8)); u8((uint8_t)v); }
  void i32(int32_t v) { for (int s = 24; s >= 0; s -= 8) u8((uint8_t)((uint32_t)v >> s)); }
  void i64(int64_t v) { for (int s = 56; s >= 0; s -= 8) u8((uint8_t)((uint64_t)v >> s)); }
  void f64(double v) { int64_t b; std::memcpy(&b, &v, 8); i64(b); }
  void bytes(const uint8_t *b, int64_t len) { for (int64_t i = 0; i < len; i++) u8(b[i]); }
  void binary(const uint8_t *b, int32_t len) { i32(len); bytes(b, len); }
  void field(uint8_t type, int16_t id) { u8(type); i16(id); }
  void stop() { u8(T_STOP); }
  int finish(int64_t *len) const {
    if (len) *len = n;
    return n <= cap ? ANNC_OK : fail(ANNC_ESPACE, "output buffer too small: " + std::to_string(n) + " bytes needed");
  }
};

// ---- reader ---------------------------------------------------------------------------------------------------------
struct R {
  const uint8_t *p;
  int64_t n, o = 0;
  bool bad = false;  // ran off the end
  R(const uint8_t *buf, int64_t len) : p(buf), n(len) {}
  bool need(int64_t k) { if (bad || k < 0 || o + k > n) { bad = true; return false; } return true; }
  uint8_t u8() { if (!need(1)) return 0; return p[o++]; }
  int16_t i16() { if (!need(2)) return 0; uint16_t v = (uint16_t)((p[o] << 8) | p[o + 1]); o += 2; return (int16_t)v; }
  int32_t i32() { if (!need(4)) return 0; uint32_t v = 0; for (int i = 0; i < 4; i++) v = (v << 8) | p[o + i]; o += 4; return (int32_t)v; }
  int64_t i64() { if (!need(8)) return 0; uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[o + i]; o += 8; return (int64_t)v; }
  double f64() { int64_t b = i64(); double v; std::memcpy(&v, &b, 8); return v; }
  // skip a value of the given type (TProtocolUtil.skip); depth-limited like the library (64)
  bool skip(uint8_t type, int depth = 0) {
    if (depth > 64) return false;
    switch (type) {
      case T_BOOL: case T_BYTE: return need(1) && (o += 1, true);
      case T_I16: return need(2) && (o += 2, true);
      case T_I32: return need(4) && (o += 4, true);
      case T_I64: case T_DOUBLE: return need(8) && (o += 8, true);
      case T_STRING: { int32_t len = i32(); return !bad && len >= 0 && need(len) && (o += len, true); }
      case T_STRUCT:
        for (;;) {
          uint8_t t = u8();
          if (bad) return false;
          if (t == T_STOP) return true;
          i16();
          if (!skip(t, depth + 1)) return false;
        }
      case T_MAP: {
        uint8_t kt = u8(), vt = u8();
        int32_t sz = i32();
        if (bad || sz < 0) return false;
        for (int32_t i = 0; i < sz; i++)
          if (!skip(kt, depth + 1) || !skip(vt, depth + 1)) return false;
        return true;
      }
      case T_SET: case T_LIST: {
        uint8_t et = u8();
        int32_t sz = i32();
        if (bad || sz < 0) return false;
        for (int32_t i = 0; i < sz; i++)
          if (!skip(et, depth + 1)) return false;
        return true;
      }
      default: return false;
    }
  }
};

int trunc_or_format(const R &r, const char *what) {
  return r.bad ? fail(ANNC_ETRUNC, std::string(what) + ": input ends inside a value")
               : fail(ANNC_EFORMAT, std::string(what) + ": malformed value");
}

// ---- simClustersAnn.thrift ------------------------------------------------------------------------------------------
void write_query(W &w, const sann_wire_query_t &q) {
  // 1: required SimClustersEmbeddingId sourceEmbeddingId
  w.field(T_STRUCT, 1);
  w.field(T_I32, 1); w.i32(q.embedding_type);
  w.field(T_I32, 2); w.i32(q.model_version);
  w.field(T_STRUCT, 3);  // InternalId: a union is a struct with exactly one field set
  const uint8_t vt = (uint8_t)q.internal_id_type;
  w.field(vt, (int16_t)q.internal_id_kind);
  if (vt == T_I64) w.i64(q.internal_id_value);
  else if (vt == T_I32) w.i32((int32_t)q.internal_id_value);
  else w.bytes(q.internal_id_raw, q.internal_id_raw_len);  // string / struct variants: their encoded value
  w.stop();  // InternalId
  w.stop();  // SimClustersEmbeddingId
  // 2: required SimClustersANNConfig config
  const sann_config_t &c = q.config;
  w.field(T_STRUCT, 2);
  w.field(T_I32, 1); w.i32(c.max_num_results);
  w.field(T_DOUBLE, 2); w.f64(c.min_score);
  w.field(T_I32, 3); w.i32(c.candidate_embedding_type);
  w.field(T_I32, 4); w.i32(c.max_top_tweets_per_cluster);
  w.field(T_I32, 5); w.i32(c.max_scan_clusters);
  w.field(T_I32, 6); w.i32(c.max_tweet_candidate_age_hours);
  w.field(T_I32, 7); w.i32(c.min_tweet_candidate_age_hours);
  w.field(T_I32, 8); w.i32(c.ann_algorithm);
  w.stop();
  w.stop();  // Query
}

int check_query(const sann_wire_query_t *q) {
  if (!q) return fail(ANNC_EINVAL, "query is NULL");
  const int t = q->internal_id_type;
  if (t != T_I64 && t != T_I32 && t != T_STRING && t != T_STRUCT) return fail(ANNC_EINVAL, "internal_id_type must be 8, 10, 11 or 12");
  if ((t == T_STRING || t == T_STRUCT) && (!q->internal_id_raw || q->internal_id_raw_len < 1)) return fail(ANNC_EINVAL, "a string / struct InternalId needs its raw value");
  if (q->internal_id_kind < 1 || q->internal_id_kind > 32767) return fail(ANNC_EINVAL, "internal_id_kind must be a field id");
  return ANNC_OK;
}

// reads a struct's fields; `on_field(id, type)` returns 1 handled, 0 unknown (skipped here), < 0 error
template <class F>
int read_struct(R &r, const char *what, F on_field) {
  for (;;) {
    const uint8_t t = r.u8();
    if (r.bad) return trunc_or_format(r, what);
    if (t == T_STOP) return ANNC_OK;
    const int16_t id = r.i16();
    if (r.bad) return trunc_or_format(r, what);
    const int rc = on_field(id, t);
    if (rc < 0) return rc;
    if (rc == 0 && !r.skip(t)) return trunc_or_format(r, what);
    if (r.bad) return trunc_or_format(r, what);
  }
}

// A nested struct read inside a field callback: the innermost error code is kept in `inner` (captured by reference),
// every enclosing level reports the sentinel -1000, and the entry point turns it back into `inner`.
#define NESTED(call)                          \
  do {                                        \
    const int rc_ = (call);                   \
    if (rc_) {                                \
      if (rc_ != -1000) inner = rc_;          \
      return -1000;                           \
    }                                         \
    return 1;                                 \
  } while (0)

int wrong_type(const char *what, int id) { return fail(ANNC_EFORMAT, std::string(what) + ": field " + std::to_string(id) + " has the wrong type"); }

int read_query(R &r, sann_wire_query_t *q) {
  std::memset(q, 0, sizeof(*q));
  unsigned seen = 0, seen_cfg = 0, seen_id = 0;
  int inner = ANNC_OK;
  int rc = read_struct(r, "Query", [&](int id, uint8_t t) -> int {
    if (id == 1) {
      if (t != T_STRUCT) return wrong_type("Query", id);
      seen |= 1;
      NESTED(read_struct(r, "SimClustersEmbeddingId", [&](int id2, uint8_t t2) -> int {
        if (id2 == 1 || id2 == 2) {
          if (t2 != T_I32) return wrong_type("SimClustersEmbeddingId", id2);
          (id2 == 1 ? q->embedding_type : q->model_version) = r.i32();
          seen_id |= 1u << id2;
          return 1;
        }
        if (id2 == 3) {
          if (t2 != T_STRUCT) return wrong_type("SimClustersEmbeddingId", id2);
          seen_id |= 8;
          int set = 0;
          int rc3 = read_struct(r, "InternalId", [&](int id3, uint8_t t3) -> int {
            set++;
            q->internal_id_kind = id3;
            q->internal_id_type = t3;
            if (t3 == T_I64) { q->internal_id_value = r.i64(); return 1; }
            if (t3 == T_I32) { q->internal_id_value = r.i32(); return 1; }
            q->internal_id_raw = r.p + r.o;
            const int64_t at = r.o;
            if (!r.skip(t3)) return trunc_or_format(r, "InternalId");
            q->internal_id_raw_len = r.o - at;
            return 1;
          });
          if (rc3) return rc3;
          if (set != 1) return fail(ANNC_EFORMAT, "InternalId: a union must have exactly one field set");
          return 1;
        }
        return 0;
      }));
    }
    if (id == 2) {
      if (t != T_STRUCT) return wrong_type("Query", id);
      seen |= 2;
      sann_config_t &c = q->config;
      NESTED(read_struct(r, "SimClustersANNConfig", [&](int id2, uint8_t t2) -> int {
        if (id2 < 1 || id2 > 8) return 0;
        if (t2 != (id2 == 2 ? T_DOUBLE : T_I32)) return wrong_type("SimClustersANNConfig", id2);
        seen_cfg |= 1u << id2;
        switch (id2) {
          case 1: c.max_num_results = r.i32(); break;
          case 2: c.min_score = r.f64(); break;
          case 3: c.candidate_embedding_type = r.i32(); break;
          case 4: c.max_top_tweets_per_cluster = r.i32(); break;
          case 5: c.max_scan_clusters = r.i32(); break;
          case 6: c.max_tweet_candidate_age_hours = r.i32(); break;
          case 7: c.min_tweet_candidate_age_hours = r.i32(); break;
          default: c.ann_algorithm = r.i32(); break;
        }
        return 1;
      }));
    }
    return 0;
  });
  if (rc) return rc == -1000 ? inner : rc;
  if (seen != 3) return fail(ANNC_EFORMAT, "Query: a required field is missing");
  if (seen_id != (2u | 4u | 8u)) return fail(ANNC_EFORMAT, "SimClustersEmbeddingId: a required field is missing");
  if (seen_cfg != 0x1feu) return fail(ANNC_EFORMAT, "SimClustersANNConfig: a required field is missing");
  return ANNC_OK;
}

void write_candidates(W &w, int32_t count, const int64_t *ids, const double *scores) {
  w.u8(T_STRUCT);
  w.i32(count);
  for (int32_t i = 0; i < count; i++) {
    w.field(T_I64, 1); w.i64(ids[i]);
    w.field(T_DOUBLE, 2); w.f64(scores[i]);
    w.stop();
  }
}

int read_candidates(R &r, int32_t cap, int64_t *ids, double *scores, int32_t *count) {
  const uint8_t et = r.u8();
  const int32_t sz = r.i32();
  if (r.bad) return trunc_or_format(r, "list<SimClustersANNTweetCandidate>");
  if (et != T_STRUCT || sz < 0) return fail(ANNC_EFORMAT, "list<SimClustersANNTweetCandidate>: not a list of structs");
  if (count) *count = sz;
  for (int32_t i = 0; i < sz; i++) {
    unsigned seen = 0;
    int64_t id = 0;
    double sc = 0;
    int rc = read_struct(r, "SimClustersANNTweetCandidate", [&](int fid, uint8_t t) -> int {
      if (fid == 1) { if (t != T_I64) return wrong_type("SimClustersANNTweetCandidate", fid); id = r.i64(); seen |= 1; return 1; }
      if (fid == 2) { if (t != T_DOUBLE) return wrong_type("SimClustersANNTweetCandidate", fid); sc = r.f64(); seen |= 2; return 1; }
      return 0;
    });
    if (rc) return rc;
    if (seen != 3) return fail(ANNC_EFORMAT, "SimClustersANNTweetCandidate: a required field is missing");
    if (i < cap && ids && scores) { ids[i] = id; scores[i] = sc; }
  }
  if (sz > cap && (ids || scores)) return fail(ANNC_ESPACE, "more candidates than the output arrays hold");
  return ANNC_OK;
}

const char kMethod[] = "getTweetCandidates";

void write_message_begin(W &w, int32_t type, int32_t seqid) {  // strict write
  w.i32((int32_t)(VERSION_1 | (uint32_t)type));
  w.binary(reinterpret_cast<const uint8_t *>(kMethod), (int32_t)sizeof(kMethod) - 1);
  w.i32(seqid);
}

int read_message_begin(R &r, int32_t want_type, int32_t *seqid) {
  const int32_t first = r.i32();
  if (r.bad) return trunc_or_format(r, "message");
  std::string name;
  int32_t type;
  if (first < 0) {  // strict: version | type, name, seqid
    if (((uint32_t)first & VERSION_MASK) != VERSION_1) return fail(ANNC_EFORMAT, "message: bad protocol version");
    type = first & 0xff;
    const int32_t len = r.i32();
    if (r.bad || len < 0 || !r.need(len)) return trunc_or_format(r, "message");
    name.assign(reinterpret_cast<const char *>(r.p + r.o), (size_t)len);
    r.o += len;
  } else {  // old style (strictRead is off by default): name length first, then name, type byte, seqid
    if (!r.need(first)) return trunc_or_format(r, "message");
    name.assign(reinterpret_cast<const char *>(r.p + r.o), (size_t)first);
    r.o += first;
    type = r.u8();
  }
  const int32_t sid = r.i32();
  if (r.bad) return trunc_or_format(r, "message");
  if (name != kMethod) return fail(ANNC_EFORMAT, "message: method '" + name + "' is not getTweetCandidates");
  if (type != want_type) return fail(ANNC_EFORMAT, "message: type " + std::to_string(type) + ", expected " + std::to_string(want_type));
  if (seqid) *seqid = sid;
  return ANNC_OK;
}

// ---- ann_common.thrift ----------------------------------------------------------------------------------------------
void be64(int64_t v, uint8_t out[8]) { for (int i = 0; i < 8; i++) out[i] = (uint8_t)((uint64_t)v >> (56 - 8 * i)); }
bool read_long_key(R &r, int64_t *v) {  // binary holding Injection.long2BigEndian
  const int32_t len = r.i32();
  if (r.bad || len != 8 || !r.need(8)) return false;
  *v = r.i64();
  return !r.bad;
}

int read_file(const std::string &path, std::vector<uint8_t> &out) {
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) return fail(ANNC_EIO, "cannot open " + path + ": " + std::strerror(errno));
  out.clear();
  uint8_t buf[1 << 16];
  size_t k;
  while ((k = std::fread(buf, 1, sizeof(buf), f)) > 0) out.insert(out.end(), buf, buf + k);
  const bool err = std::ferror(f) != 0;
  std::fclose(f);
  return err ? fail(ANNC_EIO, "read error on " + path) : ANNC_OK;
}
int write_file(const std::string &path, const uint8_t *p, size_t n) {
  FILE *f = std::fopen(path.c_str(), "wb");
  if (!f) return fail(ANNC_EIO, "cannot create " + path + ": " + std::strerror(errno));
  const size_t k = n ? std::fwrite(p, 1, n, f) : 0;
  const bool ok = std::fclose(f) == 0 && k == n;
  return ok ? ANNC_OK : fail(ANNC_EIO, "write error on " + path);
}

}  // namespace

extern "C" {

const char *ann_codec_last_error(void) { return g_err.c_str(); }

int sann_wire_encode_query(const sann_wire_query_t *q, uint8_t *buf, int64_t cap, int64_t *len) try {
  if (int rc = check_query(q)) return rc;
  W w(buf, cap);
  write_query(w, *q);
  return w.finish(len);
} ABI_CATCH

int sann_wire_decode_query(const uint8_t *buf, int64_t n, sann_wire_query_t *q, int64_t *consumed) try {
  if (!buf || n < 0 || !q) return fail(ANNC_EINVAL, "NULL argument");
  R r(buf, n);
  if (int rc = read_query(r, q)) return rc;
  if (consumed) *consumed = r.o;
  return ANNC_OK;
} ABI_CATCH

int sann_wire_encode_candidates(int32_t count, const int64_t *ids, const double *scores, uint8_t *buf, int64_t cap, int64_t *len) try {
  if (count < 0 || (count > 0 && (!ids || !scores))) return fail(ANNC_EINVAL, "bad candidate arrays");
  W w(buf, cap);
  write_candidates(w, count, ids, scores);
  return w.finish(len);
} ABI_CATCH

int sann_wire_decode_candidates(const uint8_t *buf, int64_t n, int32_t cap, int64_t *ids, double *scores, int32_t *count,
                                int64_t *consumed) try {
  if (!buf || n < 0) return fail(ANNC_EINVAL, "NULL argument");
  R r(buf, n);
  if (int rc = read_candidates(r, cap, ids, scores, count)) return rc;
  if (consumed) *consumed = r.o;
  return ANNC_OK;
} ABI_CATCH

int sann_wire_encode_call(int32_t seqid, const sann_wire_query_t *q, uint8_t *buf, int64_t cap, int64_t *len) try {
  if (int rc = check_query(q)) return rc;
  W w(buf, cap);
  write_message_begin(w, M_CALL, seqid);
  w.field(T_STRUCT, 1);  // getTweetCandidates_args { 1: required Query query }
  write_query(w, *q);
  w.stop();
  return w.finish(len);
} ABI_CATCH

int sann_wire_decode_call(const uint8_t *buf, int64_t n, int32_t *seqid, sann_wire_query_t *q, int64_t *consumed) try {
  if (!buf || n < 0 || !q) return fail(ANNC_EINVAL, "NULL argument");
  R r(buf, n);
  if (int rc = read_message_begin(r, M_CALL, seqid)) return rc;
  bool seen = false;
  int inner = ANNC_OK;
  int rc = read_struct(r, "getTweetCandidates_args", [&](int id, uint8_t t) -> int {
    if (id != 1) return 0;
    if (t != T_STRUCT) return wrong_type("getTweetCandidates_args", id);
    seen = true;
    inner = read_query(r, q);
    return inner ? -1000 : 1;
  });
  if (rc) return rc == -1000 ? inner : rc;
  if (!seen) return fail(ANNC_EFORMAT, "getTweetCandidates_args: query is missing");
  if (consumed) *consumed = r.o;
  return ANNC_OK;
} ABI_CATCH

int sann_wire_encode_reply(int32_t seqid, int32_t count, const int64_t *ids, const double *scores, uint8_t *buf, int64_t cap,
                           int64_t *len) try {
  if (count < 0 || (count > 0 && (!ids || !scores))) return fail(ANNC_EINVAL, "bad candidate arrays");
  W w(buf, cap);
  write_message_begin(w, M_REPLY, seqid);
  w.field(T_LIST, 0);  // getTweetCandidates_result { 0: success }
  write_candidates(w, count, ids, scores);
  w.stop();
  return w.finish(len);
} ABI_CATCH

int sann_wire_decode_reply(const uint8_t *buf, int64_t n, int32_t *seqid, int32_t cap, int64_t *ids, double *scores, int32_t *count,
                           int64_t *consumed) try {
  if (!buf || n < 0) return fail(ANNC_EINVAL, "NULL argument");
  R r(buf, n);
  if (int rc = read_message_begin(r, M_REPLY, seqid)) return rc;
  bool seen = false;
  int inner = ANNC_OK;
  int rc = read_struct(r, "getTweetCandidates_result", [&](int id, uint8_t t) -> int {
    if (id != 0) return 0;  // 1..3 are the declared exceptions: skipped, reported below as "no success field"
    if (t != T_LIST) return wrong_type("getTweetCandidates_result", id);
    seen = true;
    inner = read_candidates(r, cap, ids, scores, count);
    return inner ? -1000 : 1;
  });
  if (rc) return rc == -1000 ? inner : rc;
  if (!seen) return fail(ANNC_EFORMAT, "getTweetCandidates_result: no success field (the server answered with an exception)");
  if (consumed) *consumed = r.o;
  return ANNC_OK;
} ABI_CATCH

// ---- HNSW index files -----------------------------------------------------------------------------------------------
int hnsw_codec_encode_internal_metadata(const hnsw_internal_metadata_t *m, uint8_t *buf, int64_t cap, int64_t *len) try {
  if (!m) return fail(ANNC_EINVAL, "metadata is NULL");
  W w(buf, cap);
  w.field(T_I32, 1); w.i32(m->max_level);
  if (m->has_entry_point) {
    uint8_t k[8];
    be64(m->entry_point, k);
    w.field(T_STRING, 2); w.binary(k, 8);
  }
  w.field(T_I32, 3); w.i32(m->ef_construction);
  w.field(T_I32, 4); w.i32(m->max_m);
  w.field(T_I32, 5); w.i32(m->num_elements);
  w.stop();
  return w.finish(len);
} ABI_CATCH

int hnsw_codec_decode_internal_metadata(const uint8_t *buf, int64_t n, hnsw_internal_metadata_t *m) try {
  if (!buf || n < 0 || !m) return fail(ANNC_EINVAL, "NULL argument");
  std::memset(m, 0, sizeof(*m));
  R r(buf, n);
  return read_struct(r, "HnswInternalIndexMetadata", [&](int id, uint8_t t) -> int {
    if (id == 2) {
      if (t != T_STRING) return wrong_type("HnswInternalIndexMetadata", id);
      if (!read_long_key(r, &m->entry_point)) return fail(ANNC_EFORMAT, "HnswInternalIndexMetadata: entryPoint is not an 8-byte long key");
      m->has_entry_point = 1;
      return 1;
    }
    if (id == 1 || (id >= 3 && id <= 5)) {
      if (t != T_I32) return wrong_type("HnswInternalIndexMetadata", id);
      const int32_t v = r.i32();
      (id == 1 ? m->max_level : id == 3 ? m->ef_construction : id == 4 ? m->max_m : m->num_elements) = v;
      return 1;
    }
    return 0;
  });
} ABI_CATCH

int hnsw_codec_encode_index_metadata(int32_t dimension, int32_t metric, int32_t num_elements, uint8_t *buf, int64_t cap, int64_t *len) try {
  W w(buf, cap);
  w.field(T_I32, 1); w.i32(dimension);
  w.field(T_I32, 2); w.i32(metric);  // enums travel as i32
  w.field(T_I32, 3); w.i32(num_elements);
  w.stop();
  return w.finish(len);
} ABI_CATCH

int hnsw_codec_decode_index_metadata(const uint8_t *buf, int64_t n, int32_t *dimension, int32_t *metric, int32_t *num_elements) try {
  if (!buf || n < 0) return fail(ANNC_EINVAL, "NULL argument");
  R r(buf, n);
  int32_t v[4] = {0, 0, 0, 0};
  int rc = read_struct(r, "HnswIndexMetadata", [&](int id, uint8_t t) -> int {
    if (id < 1 || id > 3) return 0;
    if (t != T_I32) return wrong_type("HnswIndexMetadata", id);
    v[id] = r.i32();
    return 1;
  });
  if (rc) return rc;
  if (dimension) *dimension = v[1];
  if (metric) *metric = v[2];
  if (num_elements) *num_elements = v[3];
  return ANNC_OK;
} ABI_CATCH

int hnsw_codec_encode_graph(int64_t n_entries, const int32_t *level, const int64_t *key, const int64_t *offsets,
                            const int64_t *neighbours, uint8_t *buf, int64_t cap, int64_t *len) try {
  if (n_entries < 0 || (n_entries > 0 && (!level || !key || !offsets))) return fail(ANNC_EINVAL, "bad graph arrays");
  W w(buf, cap);
  uint8_t k[8];
  for (int64_t e = 0; e < n_entries; e++) {
    const int64_t b = offsets[e], en = offsets[e + 1];
    if (en < b || en - b > 0x7fffffff || (en > b && !neighbours)) return fail(ANNC_EINVAL, "bad entry_offsets");
    w.field(T_I32, 1); w.i32(level[e]);
    be64(key[e], k);
    w.field(T_STRING, 2); w.binary(k, 8);
    w.field(T_LIST, 3); w.u8(T_STRING); w.i32((int32_t)(en - b));
    for (int64_t j = b; j < en; j++) { be64(neighbours[j], k); w.binary(k, 8); }
    w.stop();
  }
  return w.finish(len);
} ABI_CATCH

int hnsw_codec_decode_graph(const uint8_t *buf, int64_t n, int64_t cap_entries, int64_t cap_neighbours, int32_t *level, int64_t *key,
                            int64_t *offsets, int64_t *neighbours, int64_t *n_entries, int64_t *n_neighbours) try {
  if (!buf || n < 0) return fail(ANNC_EINVAL, "NULL argument");
  const bool store = level && key && offsets;
  R r(buf, n);
  int64_t ne = 0, nn = 0;
  bool overflow = false;
  while (r.o < n) {  // "until END_OF_FILE": a struct either starts at the end of the input or not at all
    int32_t lv = 0;
    int64_t k = 0;
    bool have_key = false;
    const int64_t nn0 = nn;
    int rc = read_struct(r, "HnswGraphEntry", [&](int id, uint8_t t) -> int {
      if (id == 1) { if (t != T_I32) return wrong_type("HnswGraphEntry", id); lv = r.i32(); return 1; }
      if (id == 2) {
        if (t != T_STRING) return wrong_type("HnswGraphEntry", id);
        if (!read_long_key(r, &k)) return fail(ANNC_EFORMAT, "HnswGraphEntry: key is not an 8-byte long key");
        have_key = true;
        return 1;
      }
      if (id == 3) {
        if (t != T_LIST) return wrong_type("HnswGraphEntry", id);
        const uint8_t et = r.u8();
        const int32_t sz = r.i32();
        if (r.bad) return trunc_or_format(r, "HnswGraphEntry");
        if (et != T_STRING || sz < 0) return fail(ANNC_EFORMAT, "HnswGraphEntry: neighbours is not a list of binaries");
        for (int32_t i = 0; i < sz; i++) {
          int64_t v;
          if (!read_long_key(r, &v)) return r.bad ? trunc_or_format(r, "HnswGraphEntry") : fail(ANNC_EFORMAT, "HnswGraphEntry: a neighbour is not an 8-byte long key");
          if (store && neighbours && nn < cap_neighbours) neighbours[nn] = v;
          else if (store) overflow = true;
          nn++;
        }
        return 1;
      }
      return 0;
    });
    if (rc) return rc;
    if (!have_key) return fail(ANNC_EFORMAT, "HnswGraphEntry: key is missing");
    if (store && ne < cap_entries) { level[ne] = lv; key[ne] = k; offsets[ne] = nn0; }
    else if (store) overflow = true;
    ne++;
  }
  if (store && ne <= cap_entries) offsets[ne] = nn;
  if (n_entries) *n_entries = ne;
  if (n_neighbours) *n_neighbours = nn;
  if (overflow) return fail(ANNC_ESPACE, "graph arrays too small: " + std::to_string(ne) + " entries, " + std::to_string(nn) + " neighbours");
  return ANNC_OK;
} ABI_CATCH

int ann_wire_encode_neighbor_result(int32_t metric, int32_t count, const int64_t *ids, const float *distances, int32_t with_distance,
                                    uint8_t *buf, int64_t cap, int64_t *len) try {
  if (count < 0 || (count > 0 && (!ids || (with_distance && !distances)))) return fail(ANNC_EINVAL, "bad neighbour arrays");
  const int arm = metric == HNSW_METRIC_COSINE ? 1 : metric == HNSW_METRIC_L2 ? 2 : metric == HNSW_METRIC_INNER_PRODUCT ? 3 : 0;
  if (with_distance && !arm) return fail(ANNC_EINVAL, "unknown metric");
  W w(buf, cap);
  uint8_t k[8];
  w.field(T_LIST, 1); w.u8(T_STRUCT); w.i32(count);  // 1: required list<NearestNeighbor> nearestNeighbors
  for (int32_t i = 0; i < count; i++) {
    be64(ids[i], k);
    w.field(T_STRING, 1); w.binary(k, 8);            // 1: required binary id
    if (with_distance) {
      w.field(T_STRUCT, 2);                            // 2: optional Distance distance (union)
      w.field(T_STRUCT, (int16_t)arm);                 //    1 cosineDistance / 2 l2Distance / 3 innerProductDistance
      w.field(T_DOUBLE, 1); w.f64((double)distances[i]);  //  1: required double distance
      w.stop();
      w.stop();
    }
    w.stop();
  }
  w.stop();
  return w.finish(len);
} ABI_CATCH

int ann_wire_decode_neighbor_result(const uint8_t *buf, int64_t n, int32_t cap, int64_t *ids, double *distances, int32_t *arms,
                                    int32_t *count, int64_t *consumed) try {
  if (!buf || n < 0) return fail(ANNC_EINVAL, "NULL argument");
  R r(buf, n);
  bool seen = false;
  int32_t total = 0;
  int inner = ANNC_OK;
  int rc = read_struct(r, "NearestNeighborResult", [&](int id, uint8_t t) -> int {
    if (id != 1) return 0;
    if (t != T_LIST) return wrong_type("NearestNeighborResult", id);
    seen = true;
    const uint8_t et = r.u8();
    const int32_t sz = r.i32();
    if (r.bad) return trunc_or_format(r, "NearestNeighborResult");
    if (et != T_STRUCT || sz < 0) return fail(ANNC_EFORMAT, "NearestNeighborResult: not a list of structs");
    total = sz;
    for (int32_t i = 0; i < sz; i++) {
      int64_t key = 0;
      double dist = 0;
      int arm = 0;
      bool have = false;
      int rc2 = read_struct(r, "NearestNeighbor", [&](int id2, uint8_t t2) -> int {
        if (id2 == 1) {
          if (t2 != T_STRING) return wrong_type("NearestNeighbor", id2);
          if (!read_long_key(r, &key)) return fail(ANNC_EFORMAT, "NearestNeighbor: id is not an 8-byte long key");
          have = true;
          return 1;
        }
        if (id2 == 2) {
          if (t2 != T_STRUCT) return wrong_type("NearestNeighbor", id2);
          NESTED(read_struct(r, "Distance", [&](int id3, uint8_t t3) -> int {
            if (id3 < 1 || id3 > 3) return 0;  // (4 = EditDistance: skipped, arm stays 0)
            if (t3 != T_STRUCT) return wrong_type("Distance", id3);
            arm = id3;
            NESTED(read_struct(r, "distance arm", [&](int id4, uint8_t t4) -> int {
              if (id4 != 1) return 0;
              if (t4 != T_DOUBLE) return wrong_type("distance arm", id4);
              dist = r.f64();
              return 1;
            }));
          }));
        }
        return 0;
      });
      if (rc2) return rc2;
      if (!have) return fail(ANNC_EFORMAT, "NearestNeighbor: id is missing");
      if (i < cap) {
        if (ids) ids[i] = key;
        if (distances) distances[i] = dist;
        if (arms) arms[i] = arm;
      }
    }
    return 1;
  });
  if (rc) return rc == -1000 ? inner : rc;
  if (!seen) return fail(ANNC_EFORMAT, "NearestNeighborResult: nearestNeighbors is missing");
  if (count) *count = total;
  if (consumed) *consumed = r.o;
  if (total > cap && (ids || distances || arms)) return fail(ANNC_ESPACE, "more neighbours than the output arrays hold");
  return ANNC_OK;
} ABI_CATCH

// ---- directories ----------------------------------------------------------------------------------------------------
int hnsw_index_save_directory(const hnsw_index_t *index, int32_t ef_construction, const char *dir) try {
  if (!index || !dir) return fail(ANNC_EINVAL, "NULL argument");
  int64_t n = 0, ne = 0, nn = 0, entry = -1;
  int32_t d = 0, metric = 0, max_m = 0, max_level = 0;
  if (hnsw_index_info(index, &n, &d, &metric, &max_m) || hnsw_index_graph_size(index, &ne, &nn, &entry, &max_level))
    return fail(ANNC_EINVAL, std::string("index: ") + hnsw_last_error());
  std::vector<int32_t> lv((size_t)ne + 1);
  std::vector<int64_t> it((size_t)ne + 1), off((size_t)ne + 2), nb((size_t)nn + 1), ids((size_t)n + 1);
  if (hnsw_index_graph(index, lv.data(), it.data(), off.data(), nb.data()) || hnsw_index_get_ids(index, ids.data()))
    return fail(ANNC_EINVAL, std::string("index: ") + hnsw_last_error());
  for (int64_t e = 0; e < ne; e++) it[(size_t)e] = ids[(size_t)it[(size_t)e]];  // positions -> keys
  for (int64_t j = 0; j < nn; j++) nb[(size_t)j] = ids[(size_t)nb[(size_t)j]];
  const std::string root(dir), inner = root + "/hnsw_internal_index";
  if (mkdir(root.c_str(), 0777) && errno != EEXIST) return fail(ANNC_EIO, "cannot create " + root + ": " + std::strerror(errno));
  if (mkdir(inner.c_str(), 0777) && errno != EEXIST) return fail(ANNC_EIO, "cannot create " + inner + ": " + std::strerror(errno));
  std::vector<uint8_t> buf;
  int64_t len = 0;
  // hnsw_internal_graph
  hnsw_codec_encode_graph(ne, lv.data(), it.data(), off.data(), nb.data(), nullptr, 0, &len);
  buf.resize((size_t)len + 1);
  if (int rc = hnsw_codec_encode_graph(ne, lv.data(), it.data(), off.data(), nb.data(), buf.data(), len, &len)) return rc;
  if (int rc = write_file(inner + "/hnsw_internal_graph", buf.data(), (size_t)len)) return rc;
  // hnsw_internal_metadata (HnswIndex.toDirectory: numElements = the number of graph entries written, :630-641)
  hnsw_internal_metadata_t im;
  im.max_level = entry >= 0 ? max_level : -1;  // HnswMeta starts at (-1, empty), HnswIndex.java:97
  im.has_entry_point = entry >= 0;
  im.entry_point = entry >= 0 ? ids[(size_t)entry] : 0;
  im.ef_construction = ef_construction;
  im.max_m = max_m;
  im.num_elements = (int32_t)ne;
  buf.resize(64);
  if (int rc = hnsw_codec_encode_internal_metadata(&im, buf.data(), 64, &len)) return rc;
  if (int rc = write_file(inner + "/hnsw_internal_metadata", buf.data(), (size_t)len)) return rc;
  // hnsw_index_metadata (HnswIOUtil.saveIndexMetadata: dimension, metric, number of vectors)
  if (int rc = hnsw_codec_encode_index_metadata(d, metric, (int32_t)n, buf.data(), 64, &len)) return rc;
  if (int rc = write_file(root + "/hnsw_index_metadata", buf.data(), (size_t)len)) return rc;
  return write_file(root + "/_SUCCESS", nullptr, 0);  // HnswCommon.isValidHnswIndex wants hasSuccessFile
} ABI_CATCH

int hnsw_index_load_directory(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids,
                              const char *dir, hnsw_index_t **out) try {
  if (!dir || !out) return fail(ANNC_EINVAL, "NULL argument");
  const std::string root(dir), inner = root + "/hnsw_internal_index";
  std::vector<uint8_t> buf;
  if (int rc = read_file(root + "/hnsw_index_metadata", buf)) return rc;
  int32_t f_dim = 0, f_metric = 0, f_n = 0;
  if (int rc = hnsw_codec_decode_index_metadata(buf.data(), (int64_t)buf.size(), &f_dim, &f_metric, &f_n)) return rc;
  if (f_dim != d) return fail(ANNC_EFORMAT, "Dimensions do not match. requested: " + std::to_string(d) + " existing: " + std::to_string(f_dim));
  if (f_metric != metric) return fail(ANNC_EFORMAT, "DistanceMetric do not match. requested: " + std::to_string(metric) + " existing: " + std::to_string(f_metric));
  if (int rc = read_file(inner + "/hnsw_internal_metadata", buf)) return rc;
  hnsw_internal_metadata_t im;
  if (int rc = hnsw_codec_decode_internal_metadata(buf.data(), (int64_t)buf.size(), &im)) return rc;
  if (int rc = read_file(inner + "/hnsw_internal_graph", buf)) return rc;
  int64_t ne = 0, nn = 0;
  if (int rc = hnsw_codec_decode_graph(buf.data(), (int64_t)buf.size(), 0, 0, nullptr, nullptr, nullptr, nullptr, &ne, &nn)) return rc;
  std::vector<int32_t> lv((size_t)ne + 1);
  std::vector<int64_t> it((size_t)ne + 1), off((size_t)ne + 2), nb((size_t)nn + 1);
  if (int rc = hnsw_codec_decode_graph(buf.data(), (int64_t)buf.size(), ne, nn, lv.data(), it.data(), off.data(), nb.data(), &ne, &nn)) return rc;
  // keys -> positions
  std::unordered_map<int64_t, int64_t> pos;
  if (ids) {
    pos.reserve((size_t)n * 2);
    for (int64_t i = 0; i < n; i++)
      if (!pos.emplace(ids[i], i).second) return fail(ANNC_EINVAL, "ids: key " + std::to_string(ids[i]) + " appears twice");
  }
  auto locate = [&](int64_t key, int64_t *p) -> bool {
    if (!ids) { *p = key; return key >= 0 && key < n; }
    auto f = pos.find(key);
    if (f == pos.end()) return false;
    *p = f->second;
    return true;
  };
  for (int64_t e = 0; e < ne; e++)
    if (!locate(it[(size_t)e], &it[(size_t)e])) return fail(ANNC_EFORMAT, "graph entry " + std::to_string(e) + ": its key has no vector");
  for (int64_t j = 0; j < nn; j++)
    if (!locate(nb[(size_t)j], &nb[(size_t)j])) return fail(ANNC_EFORMAT, "neighbour " + std::to_string(j) + ": its key has no vector");
  int64_t entry = -1;
  if (im.has_entry_point && !locate(im.entry_point, &entry)) return fail(ANNC_EFORMAT, "the entry point's key has no vector");
  if (hnsw_index_build(device, metric, n, d, vectors, ids, im.max_m, entry, im.max_level < 0 ? 0 : im.max_level, ne, lv.data(), it.data(),
                       off.data(), nb.data(), out))
    return fail(ANNC_EINVAL, std::string("hnsw_index_build: ") + hnsw_last_error());
  return ANNC_OK;
} ABI_CATCH

}  // extern "C"
