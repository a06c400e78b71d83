// sann_corpus.hip -- synthetic SimClusters corpus generated ON THE DEVICE and built straight into
// the index layout (same law as the numpy generator in corpus.py, which
// serves the small test corpora).  A 100M-tweet corpus is ~2.2e9 (tweet, cluster, score) postings;
// only those that can reach a cluster's top `index_cap` are ever materialised.
//
// This is also the device form of the reference's posting-list materialisation
//   src/scala/com/twitter/simclusters_v2/summingbird/stores/TopKTweetsForClusterReadableStore.scala:211-229
//   (keep score > 0, sort by score descending, take maxResults)
// i.e. SURVEY "next" row N1 for the synthetic source: filter -> per-cluster sort -> cap ->
// partition, without a host round trip.
//
// Law (all constants from the reference, see corpus.py):
//   tweet t:  n_t = min(max_per_tweet, 1 + Geom(1/mean)) distinct clusters, cluster rank
//             r = floor((C+1)^u) (Zipf s=1 as the log-uniform law), cluster id = 1 + (r*A mod C);
//             score = max(exp(N(-2,1)), 0.001); id = Snowflake id spread over the window, unique.
//   cluster c: all its (tweet, score) sorted by (score desc, tweet id asc), first index_cap kept.
//
// Pipeline
//   1. host: expected list length L_c per cluster -> score threshold tau_c such that the
//      expected number of postings above it is keep_c = min(L_c, 1.3*cap + 100): a posting below
//      tau_c can only matter if fewer than `cap` postings of c lie above it, which for
//      Poisson(keep_c) >= 2700 is a > 13 sigma event; clusters with L_c <= keep_c keep everything.
//   2. gen_kernel: one thread per tweet, counter-based RNG; surviving postings are appended to
//      their cluster's bucket with one atomicAdd.
//   3. sort_kernel: one workgroup per cluster, bitonic sort in LDS by (score desc, id asc), cap,
//      count the survivors per (cluster, partition) for this shard.
//   4. host: exclusive scan of the (cluster, partition) counts -> sub_offsets.
//   5. scatter_kernel: stable partition of every sorted list into the index layout.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <new>
#include <vector>

#include "sann_host.h"
#include "sann_math.h"

using namespace sann;
using sann_host::DevBuf;
using sann_host::fail;
#include "abi_guard.h"
#define ABI_CATCH catch (...) { return abi_guard::caught(sann_host::fail, SANN_ENOMEM, SANN_EINTERNAL); }

namespace {

constexpr int SORT_MAX = 4096;  // entries a cluster bucket may hold (LDS bitonic capacity)

struct GenParams {
  int64_t n_tweets;
  int32_t n_clusters;
  int32_t max_per_tweet;
  float inv_log1mp;   // 1 / ln(1 - 1/mean)
  double log_c1;      // ln(C + 1)
  uint32_t perm_mul;  // cluster id = 1 + (rank * perm_mul) % C, gcd(perm_mul, C) = 1
  uint64_t seed;
  int64_t ms_begin;   // window start (ms since unix epoch)
  int64_t ms_span;
};

__device__ inline uint64_t rng(uint64_t seed, uint64_t t, uint32_t k) {
  return mix64(seed ^ mix64(t * 0x9E3779B97F4A7C15ull + (uint64_t)k * 0xD1B54A32D192ED03ull + 0x632BE59BD9B4E019ull));
}
__device__ inline double u01(uint64_t x) { return ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

__host__ __device__ inline int64_t synth_tweet_id(int64_t t, int64_t n_tweets, int64_t ms_begin, int64_t ms_span) {
  // evenly spread in time; the low 22 bits are a bijection of t's low 22 bits, so ids are unique
  const int64_t ms = ms_begin + (int64_t)(((__int128)t * ms_span) / n_tweets);
  return (int64_t)(((uint64_t)(ms - 1288834974657ll) << 22) | (((uint64_t)t * 2654435761ull) & 0x3FFFFFull));
}

// One thread per tweet.
__global__ __launch_bounds__(256) void gen_kernel(GenParams g, const float *zthr /*[C+1] by cluster id*/,
                                                  const uint32_t *bucket_off /*[C+2]*/, uint32_t *cursor /*[C+1]*/,
                                                  Posting *buckets, unsigned int *overflow_flag) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= g.n_tweets) return;
  const uint64_t r0 = rng(g.seed, (uint64_t)t, 0);
  // n_t = min(max, 1 + Geom(p)) : floor(ln(u)/ln(1-p)) failures before the first success
  int n_t = 1 + (int)(__logf((float)u01(r0)) * g.inv_log1mp);
  if (n_t > g.max_per_tweet) n_t = g.max_per_tweet;
  if (n_t < 1) n_t = 1;
  const int64_t id = synth_tweet_id(t, g.n_tweets, g.ms_begin, g.ms_span);
  int32_t seen[64];
  int n_seen = 0;
  for (int j = 0; j < n_t; j++) {
    const uint64_t a = rng(g.seed, (uint64_t)t, 1 + 3 * j);
    // fp64: in fp32 the grid of u * ln(C+1) is coarser than one rank near r = C and the tail
    // clusters would get visibly uneven mass
    int r = (int)exp(u01(a) * g.log_c1);
    r = r < 1 ? 1 : (r > g.n_clusters ? g.n_clusters : r);
    const int c = 1 + (int)(((uint64_t)(uint32_t)r * g.perm_mul) % (uint32_t)g.n_clusters);
    bool dup = false;
    for (int i = 0; i < n_seen; i++) dup = dup || (seen[i] == c);
    if (dup) continue;  // distinct clusters per tweet: a repeated draw is dropped
    if (n_seen < 64) seen[n_seen++] = c;
    // z ~ N(0,1) by Box-Muller; cheap fp32 screen against the cluster's threshold first
    const double u1 = u01(rng(g.seed, (uint64_t)t, 2 + 3 * j)), u2 = u01(rng(g.seed, (uint64_t)t, 3 + 3 * j));
    const float zt = zthr[c];
    const float z32 = sqrtf(-2.0f * __logf((float)u1)) * __cosf(6.2831853f * (float)u2);
    if (z32 < zt - 0.01f) continue;
    const double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    if (z < (double)zt) continue;
    double s = exp(-2.0 + z);
    s = s < 0.001 ? 0.001 : s;
    const uint32_t o = atomicAdd(&cursor[c], 1u);
    const uint32_t cap = bucket_off[c + 1] - bucket_off[c];
    if (o < cap) {
      Posting p;
      p.id = id;
      p.score = s;
      buckets[bucket_off[c] + o] = p;
    } else {
      atomicOr(overflow_flag, 1u);
    }
  }
}

// The full SimClusters embedding of tweet t -- every (cluster, score) the generator draws for it,
// with the same RNG stream, de-duplication and arithmetic as gen_kernel (which only materialises
// the postings that can reach a cluster's cap).  Returns the number of entries (<= 64).
__device__ inline int tweet_embedding(const GenParams &g, int64_t t, int32_t *cl, double *sc) {
  const uint64_t r0 = rng(g.seed, (uint64_t)t, 0);
  int n_t = 1 + (int)(__logf((float)u01(r0)) * g.inv_log1mp);
  if (n_t > g.max_per_tweet) n_t = g.max_per_tweet;
  if (n_t < 1) n_t = 1;
  int n = 0;
  for (int j = 0; j < n_t; j++) {
    const uint64_t a = rng(g.seed, (uint64_t)t, 1 + 3 * j);
    int r = (int)exp(u01(a) * g.log_c1);
    r = r < 1 ? 1 : (r > g.n_clusters ? g.n_clusters : r);
    const int c = 1 + (int)(((uint64_t)(uint32_t)r * g.perm_mul) % (uint32_t)g.n_clusters);
    bool dup = false;
    for (int i = 0; i < n; i++) dup = dup || (cl[i] == c);
    if (dup || n >= 64) continue;
    const double u1 = u01(rng(g.seed, (uint64_t)t, 2 + 3 * j)), u2 = u01(rng(g.seed, (uint64_t)t, 3 + 3 * j));
    const double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    double s = exp(-2.0 + z);
    s = s < 0.001 ? 0.001 : s;
    cl[n] = c;
    sc[n] = s;
    n++;
  }
  return n;
}

// Debug / test export: embeddings of tweets [t0, t0 + n) at a fixed stride of 64 entries.
__global__ __launch_bounds__(256) void embed_kernel(GenParams g, int64_t t0, int n, int32_t *counts, int32_t *cl, double *sc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int32_t c[64];
  double s[64];
  const int m = tweet_embedding(g, t0 + i, c, s);
  counts[i] = m;
  for (int j = 0; j < m; j++) { cl[(int64_t)i * 64 + j] = c[j]; sc[(int64_t)i * 64 + j] = s[j]; }
}

// Exact full-embedding cosine of every tweet against nq source embeddings -- what SANN
// approximates (simclusters-ann/README.md:18-46); the "quality" recall@k truth of SURVEY 8(d).
// wtab[q*(C+1) + c] = weight of cluster c in query q (0 when absent), unorm[q] = its l2 norm.
// mode 0: histogram of the cosines (HB bins over (0,1]); mode 1: emit (tweet, cosine) with
// cosine >= thr[q].  One thread per tweet.
constexpr int EXACT_HB = 4096;
__global__ __launch_bounds__(256) void exact_cosine_kernel(GenParams g, int nq, const double *wtab, const double *unorm,
                                                           int mode, uint32_t *hist, const double *thr, int64_t *out_ids,
                                                           double *out_cos, uint32_t *out_cnt, uint32_t out_cap) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= g.n_tweets) return;
  int32_t c[64];
  double s[64];
  const int m = tweet_embedding(g, t, c, s);
  double tsq = 0.0;
  for (int j = 0; j < m; j++) tsq += s[j] * s[j];
  const double tn = sqrt(tsq);
  for (int q = 0; q < nq; q++) {
    const double *w = wtab + (int64_t)q * (g.n_clusters + 1);
    double dot = 0.0;
    for (int j = 0; j < m; j++) dot += s[j] * w[c[j]];
    if (!(dot > 0.0)) continue;
    const double cosv = dot / (unorm[q] * tn);
    if (mode == 0) {
      int bin = (int)(cosv * EXACT_HB);
      bin = bin < 0 ? 0 : (bin >= EXACT_HB ? EXACT_HB - 1 : bin);
      atomicAdd(&hist[(int64_t)q * EXACT_HB + bin], 1u);
    } else if (cosv >= thr[q]) {
      const uint32_t o = atomicAdd(&out_cnt[q], 1u);
      if (o < out_cap) {
        out_ids[(int64_t)q * out_cap + o] = synth_tweet_id(t, g.n_tweets, g.ms_begin, g.ms_span);
        out_cos[(int64_t)q * out_cap + o] = cosv;
      }
    }
  }
}

// One workgroup per cluster id c in [1, C]: sort the bucket, cap, count per partition.
// scaled != NULL: the buckets hold RAW store entries (value, with scaled[i] = its DecayedValue.scaledTime): every value is
// first decayed to now_scaled (decay_to_timestamp, sann_math.h) and entries that are not > 0 afterwards are dropped --
// TopKTweetsForClusterReadableStore.scala:51-71 and :222-223 -- before the sort (:227) and the cap (:258-259).
__global__ __launch_bounds__(256) void sort_kernel(int n_clusters, int index_cap, int P, int n_shards, int shard_id,
                                                   const uint32_t *bucket_off, const uint32_t *cursor, Posting *buckets,
                                                   uint32_t *kept /*[C+1]*/, uint32_t *cnt_cp /*[C*P]*/,
                                                   const double *scaled, double now_scaled, int filter_positive) {
  __shared__ uint64_t s_hi[SORT_MAX], s_lo[SORT_MAX];
  __shared__ unsigned s_cnt[256];
  __shared__ unsigned s_valid;
  const int c = blockIdx.x + 1;
  const int tid = threadIdx.x;
  const uint32_t base = bucket_off[c];
  uint32_t n = cursor[c];
  const uint32_t cap = bucket_off[c + 1] - base;
  if (n > cap) n = cap;
  int np = 2;
  while (np < (int)n) np <<= 1;
  if (tid == 0) s_valid = 0;
  for (int i = tid; i < 256; i += 256) s_cnt[i] = 0;
  __syncthreads();
  unsigned mine = 0;
  for (int i = tid; i < np; i += 256) {
    uint64_t hi = 0, lo = 0;  // (sorts behind every kept entry: a positive score's key has the top bit set)
    if (i < (int)n) {
      const Posting p = buckets[base + i];
      double v = p.score;
      if (scaled) v = decay_to_timestamp(v, scaled[base + i], now_scaled);
      if (!filter_positive || v > 0.0) {
        hi = score_key(v);
        lo = id_key(p.id);
        mine++;
      }
    }
    s_hi[i] = hi;
    s_lo[i] = lo;
  }
  if (mine) atomicAdd(&s_valid, mine);
  __syncthreads();
  n = s_valid;
  for (int size = 2; size <= np; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (np >> 1); t += 256) {
        const int i = 2 * t - (t & (stride - 1));
        const int j = i + stride;
        const bool desc = ((i & size) == 0);
        const uint64_t ah = s_hi[i], al = s_lo[i], bh = s_hi[j], bl = s_lo[j];
        const bool a_lt_b = ah < bh || (ah == bh && al < bl);
        const bool a_gt_b = ah > bh || (ah == bh && al > bl);
        if (desc ? a_lt_b : a_gt_b) {
          s_hi[i] = bh; s_lo[i] = bl;
          s_hi[j] = ah; s_lo[j] = al;
        }
      }
      __syncthreads();
    }
  }
  const uint32_t k = n < (uint32_t)index_cap ? n : (uint32_t)index_cap;
  for (uint32_t i = tid; i < k; i += 256) {
    Posting p;
    p.id = key_id(s_lo[i]);
    p.score = key_score(s_hi[i]);
    buckets[base + i] = p;  // rank order
    const uint64_t h = mix64((uint64_t)p.id);
    if (tweet_shard(h, (uint32_t)n_shards) == (uint32_t)shard_id) atomicAdd(&s_cnt[tweet_partition(h, (uint32_t)P)], 1u);
  }
  __syncthreads();
  if (tid == 0) kept[c] = k;
  for (int p = tid; p < P; p += 256) cnt_cp[(int64_t)(c - 1) * P + p] = s_cnt[p];
}

// One workgroup per cluster: stable partition of the rank-ordered list into the index layout.
__global__ __launch_bounds__(256) void scatter_kernel(int P, int n_shards, int shard_id, const uint32_t *bucket_off,
                                                      const uint32_t *kept, const Posting *buckets,
                                                      const uint32_t *sub_offsets, Posting *postings, uint32_t *ranks) {
  __shared__ uint8_t s_part[SORT_MAX];  // partition of rank i, 255 = other shard
  const int c = blockIdx.x + 1;
  const int tid = threadIdx.x;
  const uint32_t base = bucket_off[c];
  const uint32_t k = kept[c];
  for (uint32_t i = tid; i < k; i += 256) {
    const uint64_t h = mix64((uint64_t)buckets[base + i].id);
    s_part[i] = tweet_shard(h, (uint32_t)n_shards) == (uint32_t)shard_id ? (uint8_t)tweet_partition(h, (uint32_t)P) : 255;
  }
  __syncthreads();
  // thread p walks the list in rank order and copies its partition's entries
  for (int p = tid; p < P; p += 256) {
    uint32_t o = sub_offsets[(int64_t)(c - 1) * P + p];
    for (uint32_t i = 0; i < k; i++) {
      if (s_part[i] == (uint8_t)p) {
        postings[o] = buckets[base + i];
        ranks[o] = i;
        o++;
      }
    }
  }
}

// Lists that arrive already in list order (sann_index_build_from_device_postings): per cluster, the entries per partition.
__global__ __launch_bounds__(256) void count_parts_kernel(int P, int n_shards, int shard_id, const uint32_t *bucket_off, const Posting *buckets,
                                                          uint32_t *kept /*[C+1]*/, uint32_t *cnt_cp /*[C*P]*/) {
  __shared__ unsigned s_cnt[256];
  const int c = blockIdx.x + 1, tid = threadIdx.x;
  const uint32_t base = bucket_off[c], k = bucket_off[c + 1] - base;
  s_cnt[tid] = 0;
  __syncthreads();
  for (uint32_t i = tid; i < k; i += 256) {
    const uint64_t h = mix64((uint64_t)buckets[base + i].id);
    if (tweet_shard(h, (uint32_t)n_shards) == (uint32_t)shard_id) atomicAdd(&s_cnt[tweet_partition(h, (uint32_t)P)], 1u);
  }
  __syncthreads();
  if (tid == 0) kept[c] = k;
  for (int p = tid; p < P; p += 256) cnt_cp[(int64_t)(c - 1) * P + p] = s_cnt[p];
}

// ---- the cluster-id-range deployment's sender side --------------------------------------
// One workgroup per requested cluster row: the list's top-M prefix (its postings with rank < M; ranks are the positions of
// ApproximateCosineSimilarity.scala:87, all present when the index holds whole lists) is laid out BY RANK in LDS -- so no sort
// is needed --, every posting is routed to GPU tweet_shard(hash(tweet id), n_ranks), and the postings of each destination are
// counted (write = 0) or written, in rank order, at seg_off[cluster][dest] of `out` (write = 1).
constexpr int EXPORT_MAX = 4096;  // M the export handles (the index caps lists at 2000; maxTopTweetsPerCluster is 800 in production)
__global__ __launch_bounds__(256) void export_prefix_kernel(IndexView ix, const int32_t *rows, int M, int n_ranks, int write,
                                                            uint32_t *counts /*[n][n_ranks]*/, const int64_t *seg_off /*[n][n_ranks]*/,
                                                            Posting *out) {
  __shared__ Posting s_p[EXPORT_MAX];
  __shared__ uint8_t s_dest[EXPORT_MAX];
  __shared__ unsigned s_cnt[16];
  __shared__ unsigned s_wave[4];
  const int tid = threadIdx.x, c = blockIdx.x, row = rows[c];
  if (tid < 16) s_cnt[tid] = 0;
  for (int i = tid; i < EXPORT_MAX; i += 256) s_dest[i] = 255;
  __syncthreads();
  // the P sub-lists hold the list's postings with ascending ranks: walk each up to the first rank >= M
  const int P = ix.P;
  for (int p = 0; p < P; p++) {
    const uint32_t b = ix.sub_offsets[(int64_t)row * P + p], e = ix.sub_offsets[(int64_t)row * P + p + 1];
    for (uint32_t i = b + tid; i < e; i += 256) {
      const uint32_t r = ix.ranks[i];
      if (r < (uint32_t)M && r < (uint32_t)EXPORT_MAX) {
        const Posting q = ix.postings[i];
        s_p[r] = q;
        s_dest[r] = (uint8_t)tweet_shard(mix64((uint64_t)q.id), (uint32_t)n_ranks);
      }
    }
  }
  __syncthreads();
  const int n = M < EXPORT_MAX ? M : EXPORT_MAX;
  if (!write) {
    for (int i = tid; i < n; i += 256)
      if (s_dest[i] != 255) atomicAdd(&s_cnt[s_dest[i]], 1u);
    __syncthreads();
    if (tid < n_ranks) counts[(int64_t)c * n_ranks + tid] = s_cnt[tid];
    return;
  }
  // stable compaction per destination: ranks ascending within a destination's segment
  for (int d = 0; d < n_ranks; d++) {
    unsigned run = 0;
    for (int i0 = 0; i0 < n; i0 += 256) {
      const int i = i0 + tid;
      const bool mine = i < n && s_dest[i] == (uint8_t)d;
      const unsigned long long m = __ballot(mine);
      const int lane = tid & 63, wv = tid >> 6;
      if (lane == 0) s_wave[wv] = (unsigned)__popcll(m);
      __syncthreads();
      unsigned before = 0;
      for (int w = 0; w < wv; w++) before += s_wave[w];
      const unsigned total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
      if (mine) out[seg_off[(int64_t)c * n_ranks + d] + run + before + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = s_p[i];
      run += total;
      __syncthreads();
    }
  }
}

// inverse normal CDF (Acklam's rational approximation, |rel err| < 1.2e-9)
double inv_norm_cdf(double p) {
  static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                             1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
  static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                             6.680131188771972e+01, -1.328068155288572e+01};
  static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                             -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
  static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00,
                             3.754408661907416e+00};
  const double plow = 0.02425, phigh = 1 - plow;
  if (p < plow) {
    double q = std::sqrt(-2 * std::log(p));
    return (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) /
           ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
  }
  if (p > phigh) {
    double q = std::sqrt(-2 * std::log(1 - p));
    return -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) /
           ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
  }
  double q = p - 0.5, r = q * q;
  return (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q /
         (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1);
}

uint64_t gcd_u64(uint64_t a, uint64_t b) { return b ? gcd_u64(b, a % b) : a; }

uint32_t perm_multiplier(int C) {
  uint32_t m = 2654435761u % (uint32_t)C;
  if (m == 0) m = 1;
  while (gcd_u64(m, (uint64_t)C) != 1) m++;
  return m;
}

GenParams gen_params(const sann_synth_params_t *sp) {
  GenParams g;
  const double p_geo = 1.0 / sp->mean_clusters;
  g.n_tweets = sp->n_tweets;
  g.n_clusters = sp->n_clusters;
  g.max_per_tweet = sp->max_clusters_per_tweet;
  g.inv_log1mp = (float)(1.0 / std::log(1.0 - p_geo));
  g.log_c1 = std::log((double)sp->n_clusters + 1.0);
  g.perm_mul = perm_multiplier(sp->n_clusters);
  g.seed = sp->seed;
  g.ms_span = (int64_t)sp->window_hours * 3600000ll;
  g.ms_begin = sp->now_ms - g.ms_span;
  return g;
}

}  // namespace

extern "C" {

int sann_index_build_synthetic(const sann_index_options_t *opts, const sann_synth_params_t *sp, sann_index_t **out) try {
  if (!out) return fail(SANN_EINVAL, "out is NULL");
  *out = nullptr;
  if (!opts || !sp) return fail(SANN_EINVAL, "NULL argument");
  const int P = opts->n_partitions == 0 ? 32 : opts->n_partitions;
  if (P < 1 || P > 128 || (P & (P - 1))) return fail(SANN_EINVAL, "n_partitions must be a power of two in [1,128]");
  const int n_shards = opts->n_shards <= 0 ? 1 : opts->n_shards;
  if (opts->shard_id < 0 || opts->shard_id >= n_shards) return fail(SANN_EINVAL, "shard_id out of range");
  const int C = sp->n_clusters;
  if (sp->n_tweets < 1 || C < 1 || sp->index_cap < 1) return fail(SANN_EINVAL, "bad synthetic parameters");
  if (sp->max_clusters_per_tweet < 1 || sp->max_clusters_per_tweet > 64 || !(sp->mean_clusters > 1.0f))
    return fail(SANN_EINVAL, "bad clusters-per-tweet parameters");
  const double keep_cap = 1.3 * sp->index_cap + 100.0;
  if (keep_cap + 8 * std::sqrt(keep_cap) + 32 > SORT_MAX)
    return fail(SANN_ELIMIT, "index_cap too large for the device builder (max ~2900)");

  // ---- 1. thresholds and bucket capacities -----------------------------------------------------
  const double p_geo = 1.0 / sp->mean_clusters;
  // E[min(max, 1+Geom)] = (1 - (1-p)^max) / p
  const double mean_nt = (1.0 - std::pow(1.0 - p_geo, sp->max_clusters_per_tweet)) / p_geo;
  const double log_c1 = std::log((double)C + 1.0);
  const uint32_t perm_mul = perm_multiplier(C);
  std::vector<float> zthr((size_t)C + 1, -1e30f);
  std::vector<uint32_t> boff((size_t)C + 2, 0);
  std::vector<double> expect((size_t)C + 1, 0.0);
  for (int r = 1; r <= C; r++) {
    const int c = 1 + (int)(((uint64_t)(uint32_t)r * perm_mul) % (uint32_t)C);
    const double pr = std::log(1.0 + 1.0 / r) / log_c1;
    // expected number of tweets that CONTAIN the cluster (repeated draws inside a tweet are dropped):
    // P = 1 - E[(1-pr)^n], n = min(N, 1 + Geom(p_geo)); with g = 1 - p_geo, x = 1 - pr:
    // E[x^n] = p_geo x (1 - (g x)^(N-1)) / (1 - g x) + g^(N-1) x^N.  Counting draws instead would
    // over-estimate hot clusters (the rank-1 cluster is in ~70 % of all tweets) and cut them short.
    const double gq = 1.0 - p_geo, xq = 1.0 - pr;
    const int Nmax = sp->max_clusters_per_tweet;
    const double exn = p_geo * xq * (1.0 - std::pow(gq * xq, Nmax - 1)) / (1.0 - gq * xq) +
                       std::pow(gq, Nmax - 1) * std::pow(xq, Nmax);
    const double L = (double)sp->n_tweets * (1.0 - exn);
    (void)mean_nt;
    double keep = L;
    if (L > keep_cap) {
      keep = keep_cap;
      zthr[(size_t)c] = (float)inv_norm_cdf(1.0 - keep_cap / L);
    }
    expect[(size_t)c] = keep;
  }
  uint64_t run = 0;
  for (int c = 1; c <= C; c++) {
    boff[(size_t)c] = (uint32_t)run;
    // the fp32 threshold is rounded, so allow a little more than Poisson noise
    const double cap = expect[(size_t)c] * 1.02 + 8.0 * std::sqrt(expect[(size_t)c] + 1.0) + 32.0;
    run += (uint64_t)std::min<double>(cap, SORT_MAX);
    if (run > 0xfffffff0ull) return fail(SANN_ELIMIT, "synthetic corpus too large for one build");
  }
  boff[(size_t)C + 1] = (uint32_t)run;

  sann_index *ix = new (std::nothrow) sann_index();
  if (!ix) return fail(SANN_ENOMEM, "out of host memory");
  struct Guard { sann_index *p; ~Guard() { delete p; } } guard{ix};
  ix->device = opts->device;
  ix->P = P;
  ix->log2P = 0;
  while ((1 << ix->log2P) < P) ix->log2P++;
  ix->shard_id = opts->shard_id;
  ix->n_shards = n_shards;
  ix->cluster_ids.resize((size_t)C);
  for (int c = 1; c <= C; c++) ix->cluster_ids[(size_t)c - 1] = c;

  HIP_TRY(hipSetDevice(ix->device));
  DevBuf d_zthr, d_boff, d_cursor, d_buckets, d_flag, d_kept, d_cnt;
  HIP_TRY(d_zthr.alloc(((size_t)C + 1) * 4));
  HIP_TRY(d_boff.alloc(((size_t)C + 2) * 4));
  HIP_TRY(d_cursor.alloc(((size_t)C + 1) * 4));
  HIP_TRY(d_kept.alloc(((size_t)C + 1) * 4));
  HIP_TRY(d_cnt.alloc((size_t)C * P * 4));
  HIP_TRY(d_flag.alloc(4));
  HIP_TRY(d_buckets.alloc(std::max<uint64_t>(run, 1) * sizeof(Posting)));
  HIP_TRY(hipMemcpy(d_zthr.p, zthr.data(), zthr.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_boff.p, boff.data(), boff.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(d_cursor.p, 0, ((size_t)C + 1) * 4));
  HIP_TRY(hipMemset(d_flag.p, 0, 4));

  // ---- 2. generate --------------------------------------------------------------------------------
  const GenParams g = gen_params(sp);
  const int64_t gen_blocks = (sp->n_tweets + 255) / 256;
  if (gen_blocks > 0x7fffffffll) return fail(SANN_ELIMIT, "too many tweets");
  hipLaunchKernelGGL(gen_kernel, dim3((unsigned)gen_blocks), dim3(256), 0, 0, g, d_zthr.as<float>(), d_boff.as<uint32_t>(),
                     d_cursor.as<uint32_t>(), d_buckets.as<Posting>(), d_flag.as<unsigned int>());
  HIP_TRY(hipGetLastError());
  unsigned int flag = 0;
  HIP_TRY(hipMemcpy(&flag, d_flag.p, 4, hipMemcpyDeviceToHost));
  if (flag) {
    std::vector<uint32_t> cur((size_t)C + 1);
    HIP_TRY(hipMemcpy(cur.data(), d_cursor.p, cur.size() * 4, hipMemcpyDeviceToHost));
    int worst = 1, n_over = 0;
    double worst_ratio = 0.0;
    for (int c = 1; c <= C; c++) {
      const double cap = (double)(boff[(size_t)c + 1] - boff[(size_t)c]);
      if (cur[(size_t)c] > cap) n_over++;
      if (cur[(size_t)c] / cap > worst_ratio) { worst_ratio = cur[(size_t)c] / cap; worst = c; }
    }
    char msg[256];
    snprintf(msg, sizeof msg,
             "synthetic corpus: %d cluster buckets overflowed; worst cluster %d got %u postings for a bucket of %u "
             "(expected %.0f, z threshold %.3f)",
             n_over, worst, cur[(size_t)worst], boff[(size_t)worst + 1] - boff[(size_t)worst], expect[(size_t)worst],
             (double)zthr[(size_t)worst]);
    return fail(SANN_EINTERNAL, msg);
  }

  // ---- 3. sort + cap + count ----------------------------------------------------------------------
  hipLaunchKernelGGL(sort_kernel, dim3((unsigned)C), dim3(256), 0, 0, C, sp->index_cap, P, n_shards, ix->shard_id,
                     d_boff.as<uint32_t>(), d_cursor.as<uint32_t>(), d_buckets.as<Posting>(), d_kept.as<uint32_t>(),
                     d_cnt.as<uint32_t>(), (const double *)nullptr, 0.0, 0);
  HIP_TRY(hipGetLastError());
  std::vector<uint32_t> cnt((size_t)C * P), kept((size_t)C + 1);
  HIP_TRY(hipMemcpy(cnt.data(), d_cnt.p, cnt.size() * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(kept.data(), d_kept.p, kept.size() * 4, hipMemcpyDeviceToHost));

  // ---- 4. offsets -----------------------------------------------------------------------------------
  ix->h_sub_offsets.resize((size_t)C * P + 1);
  uint64_t tot = 0;
  for (size_t i = 0; i < (size_t)C * P; i++) {
    ix->h_sub_offsets[i] = (uint32_t)tot;
    tot += cnt[i];
    if (tot > 0xfffffff0ull) return fail(SANN_ELIMIT, "more than 2^32 postings in one shard");
  }
  ix->h_sub_offsets[(size_t)C * P] = (uint32_t)tot;
  ix->n_postings = (int64_t)tot;
  int64_t total_all = 0;
  int32_t max_len = 0;
  for (int c = 1; c <= C; c++) {
    total_all += kept[(size_t)c];
    max_len = std::max<int32_t>(max_len, (int32_t)kept[(size_t)c]);
  }
  ix->n_postings_total = total_all;
  ix->max_list_len = max_len;
  HIP_TRY(ix->postings.alloc(std::max<uint64_t>(tot, 1) * sizeof(Posting)));
  HIP_TRY(ix->ranks.alloc(std::max<uint64_t>(tot, 1) * 4));
  HIP_TRY(ix->sub_offsets.alloc(ix->h_sub_offsets.size() * 4));
  HIP_TRY(hipMemcpy(ix->sub_offsets.p, ix->h_sub_offsets.data(), ix->h_sub_offsets.size() * 4, hipMemcpyHostToDevice));

  // ---- 5. scatter ------------------------------------------------------------------------------------
  hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)C), dim3(256), 0, 0, P, n_shards, ix->shard_id, d_boff.as<uint32_t>(),
                     d_kept.as<uint32_t>(), d_buckets.as<Posting>(), ix->sub_offsets.as<uint32_t>(),
                     ix->postings.as<Posting>(), ix->ranks.as<uint32_t>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  guard.p = nullptr;
  *out = ix;
  return SANN_OK;
} ABI_CATCH

// The cluster -> top tweets provider on the device: raw store entries in, index out.
int sann_index_build_from_postings(const sann_index_options_t *opts, int32_t n_lists, const int32_t *cluster_ids,
                                   const int64_t *list_offsets, const int64_t *tweet_ids, const double *values,
                                   const double *scaled_times, int64_t now_ms, int64_t half_life_ms, int32_t max_results,
                                   sann_index_t **out) try {
  if (!out) return fail(SANN_EINVAL, "out is NULL");
  *out = nullptr;
  if (!opts) return fail(SANN_EINVAL, "opts is NULL");
  if (n_lists < 0 || (n_lists > 0 && (!cluster_ids || !list_offsets))) return fail(SANN_EINVAL, "bad list arrays");
  const int P = opts->n_partitions == 0 ? 32 : opts->n_partitions;
  if (P < 1 || P > 128 || (P & (P - 1))) return fail(SANN_EINVAL, "n_partitions must be a power of two in [1,128]");
  const int n_shards = opts->n_shards <= 0 ? 1 : opts->n_shards;
  if (opts->shard_id < 0 || opts->shard_id >= n_shards) return fail(SANN_EINVAL, "shard_id out of range");
  if (max_results < 0) max_results = 0;
  if (scaled_times && half_life_ms <= 0) return fail(SANN_EINVAL, "half_life_ms must be positive");
  for (int32_t i = 1; i < n_lists; i++)
    if (cluster_ids[i] <= cluster_ids[i - 1]) return fail(SANN_EINVAL, "cluster_ids must be ascending and unique");
  const int64_t o0 = n_lists ? list_offsets[0] : 0;
  int64_t total = 0;
  {
    std::vector<int64_t> tmp;
    for (int32_t r = 0; r < n_lists; r++) {
      const int64_t len = list_offsets[r + 1] - list_offsets[r];
      if (len < 0) return fail(SANN_EINVAL, "list_offsets must be non-decreasing");
      if (len > SORT_MAX) return fail(SANN_ELIMIT, "a raw list holds more than 4096 entries (the store keeps <= 1.2 x 1600: Monoids.scala:440-449)");
      total += len;
      if (len >= 2) {  // keys of a Map in the store: unique
        tmp.assign(tweet_ids + list_offsets[r], tweet_ids + list_offsets[r + 1]);
        std::sort(tmp.begin(), tmp.end());
        if (std::adjacent_find(tmp.begin(), tmp.end()) != tmp.end())
          return fail(SANN_EINVAL, "cluster " + std::to_string(cluster_ids[r]) + ": a tweet id appears twice in one list");
      }
    }
  }
  if (total > 0xfffffff0ll) return fail(SANN_ELIMIT, "more than 2^32 raw postings");
  if (total > 0 && (!tweet_ids || !values)) return fail(SANN_EINVAL, "tweet_ids/values are NULL");
  const int C = n_lists;

  sann_index *ix = new (std::nothrow) sann_index();
  if (!ix) return fail(SANN_ENOMEM, "out of host memory");
  struct Guard { sann_index *p; ~Guard() { delete p; } } guard{ix};
  ix->device = opts->device;
  ix->P = P;
  ix->log2P = 0;
  while ((1 << ix->log2P) < P) ix->log2P++;
  ix->shard_id = opts->shard_id;
  ix->n_shards = n_shards;
  ix->cluster_ids.assign(cluster_ids, cluster_ids + C);

  // buckets = the raw lists as given (1-based cluster slots, as the kernels index them)
  std::vector<uint32_t> boff((size_t)C + 2, 0), cursor((size_t)C + 1, 0);
  for (int c = 1; c <= C; c++) {
    boff[(size_t)c] = (uint32_t)(list_offsets[c - 1] - o0);
    cursor[(size_t)c] = (uint32_t)(list_offsets[c] - list_offsets[c - 1]);
  }
  boff[(size_t)C + 1] = (uint32_t)total;
  std::vector<Posting> raw((size_t)total);
  for (int64_t i = 0; i < total; i++) { raw[(size_t)i].id = tweet_ids[o0 + i]; raw[(size_t)i].score = values[o0 + i]; }

  HIP_TRY(hipSetDevice(ix->device));
  DevBuf d_boff, d_cursor, d_buckets, d_scaled, d_kept, d_cnt;
  HIP_TRY(d_boff.alloc(boff.size() * 4));
  HIP_TRY(d_cursor.alloc(cursor.size() * 4));
  HIP_TRY(d_kept.alloc(((size_t)C + 1) * 4));
  HIP_TRY(d_cnt.alloc(std::max<size_t>((size_t)C * P, 1) * 4));
  HIP_TRY(d_buckets.alloc(std::max<size_t>((size_t)total, 1) * sizeof(Posting)));
  HIP_TRY(hipMemcpy(d_boff.p, boff.data(), boff.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_cursor.p, cursor.data(), cursor.size() * 4, hipMemcpyHostToDevice));
  if (total) HIP_TRY(hipMemcpy(d_buckets.p, raw.data(), (size_t)total * sizeof(Posting), hipMemcpyHostToDevice));
  if (scaled_times && total) {
    HIP_TRY(d_scaled.alloc((size_t)total * 8));
    HIP_TRY(hipMemcpy(d_scaled.p, scaled_times + o0, (size_t)total * 8, hipMemcpyHostToDevice));
  }
  // DecayedValue.build(0.0, now, halfLife).scaledTime = now * math.log(2.0) / halfLife  (algebird; log(2.0) = M_LN2 exactly)
  const double now_scaled = scaled_times ? (double)now_ms * 0.69314718055994530942 / (double)half_life_ms : 0.0;

  if (C > 0) {
    hipLaunchKernelGGL(sort_kernel, dim3((unsigned)C), dim3(256), 0, 0, C, max_results, P, n_shards, ix->shard_id,
                       d_boff.as<uint32_t>(), d_cursor.as<uint32_t>(), d_buckets.as<Posting>(), d_kept.as<uint32_t>(),
                       d_cnt.as<uint32_t>(), d_scaled.as<double>(), now_scaled, 1);
    HIP_TRY(hipGetLastError());
  }
  std::vector<uint32_t> cnt((size_t)C * P), kept((size_t)C + 1, 0);
  if (C > 0) {
    HIP_TRY(hipMemcpy(cnt.data(), d_cnt.p, cnt.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(kept.data(), d_kept.p, kept.size() * 4, hipMemcpyDeviceToHost));
  }
  ix->h_sub_offsets.resize((size_t)C * P + 1);
  uint64_t tot = 0;
  for (size_t i = 0; i < (size_t)C * P; i++) {
    ix->h_sub_offsets[i] = (uint32_t)tot;
    tot += cnt[i];
  }
  ix->h_sub_offsets[(size_t)C * P] = (uint32_t)tot;
  ix->n_postings = (int64_t)tot;
  int64_t total_all = 0;
  int32_t max_len = 0;
  for (int c = 1; c <= C; c++) {
    total_all += kept[(size_t)c];
    max_len = std::max<int32_t>(max_len, (int32_t)kept[(size_t)c]);
  }
  ix->n_postings_total = total_all;
  ix->max_list_len = max_len;
  HIP_TRY(ix->postings.alloc(std::max<uint64_t>(tot, 1) * sizeof(Posting)));
  HIP_TRY(ix->ranks.alloc(std::max<uint64_t>(tot, 1) * 4));
  HIP_TRY(ix->sub_offsets.alloc(ix->h_sub_offsets.size() * 4));
  HIP_TRY(hipMemcpy(ix->sub_offsets.p, ix->h_sub_offsets.data(), ix->h_sub_offsets.size() * 4, hipMemcpyHostToDevice));
  if (C > 0) {
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)C), dim3(256), 0, 0, P, n_shards, ix->shard_id, d_boff.as<uint32_t>(),
                       d_kept.as<uint32_t>(), d_buckets.as<Posting>(), ix->sub_offsets.as<uint32_t>(),
                       ix->postings.as<Posting>(), ix->ranks.as<uint32_t>());
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipDeviceSynchronize());
  guard.p = nullptr;
  *out = ix;
  return SANN_OK;
} ABI_CATCH

// An index from lists that are ALREADY on the device and in list order (the cluster-id-range deployment's receiver: the
// postings the other GPUs sent for this GPU's tweets).  Order inside a list is kept as given; position = rank.
int sann_index_build_from_device_postings(const sann_index_options_t *opts, int32_t n_lists, const int32_t *cluster_ids,
                                          const int64_t *list_offsets, const void *d_postings, sann_index_t **out) try {
  if (!out) return fail(SANN_EINVAL, "out is NULL");
  *out = nullptr;
  if (!opts) return fail(SANN_EINVAL, "opts is NULL");
  if (n_lists < 0 || (n_lists > 0 && (!cluster_ids || !list_offsets))) return fail(SANN_EINVAL, "bad list arrays");
  const int P = opts->n_partitions == 0 ? 32 : opts->n_partitions;
  if (P < 1 || P > 128 || (P & (P - 1))) return fail(SANN_EINVAL, "n_partitions must be a power of two in [1,128]");
  const int n_shards = opts->n_shards <= 0 ? 1 : opts->n_shards;
  if (opts->shard_id < 0 || opts->shard_id >= n_shards) return fail(SANN_EINVAL, "shard_id out of range");
  for (int32_t i = 1; i < n_lists; i++)
    if (cluster_ids[i] <= cluster_ids[i - 1]) return fail(SANN_EINVAL, "cluster_ids must be ascending and unique");
  const int C = n_lists;
  const int64_t o0 = C ? list_offsets[0] : 0;
  const int64_t total = C ? list_offsets[C] - o0 : 0;
  int32_t max_len = 0;
  for (int32_t r = 0; r < C; r++) {
    const int64_t len = list_offsets[r + 1] - list_offsets[r];
    if (len < 0) return fail(SANN_EINVAL, "list_offsets must be non-decreasing");
    if (len > SORT_MAX) return fail(SANN_ELIMIT, "a list holds more than 4096 entries");
    max_len = std::max<int32_t>(max_len, (int32_t)len);
  }
  if (total > 0xfffffff0ll) return fail(SANN_ELIMIT, "more than 2^32 postings");
  if (total > 0 && !d_postings) return fail(SANN_EINVAL, "d_postings is NULL");
  sann_index *ix = new (std::nothrow) sann_index();
  if (!ix) return fail(SANN_ENOMEM, "out of host memory");
  struct Guard { sann_index *p; ~Guard() { delete p; } } guard{ix};
  ix->device = opts->device;
  ix->P = P;
  ix->log2P = 0;
  while ((1 << ix->log2P) < P) ix->log2P++;
  ix->shard_id = opts->shard_id;
  ix->n_shards = n_shards;
  ix->cluster_ids.assign(cluster_ids, cluster_ids + C);
  ix->n_postings_total = total;
  ix->max_list_len = max_len;
  std::vector<uint32_t> boff((size_t)C + 2, 0);
  for (int c = 1; c <= C; c++) boff[(size_t)c] = (uint32_t)(list_offsets[c - 1] - o0);
  boff[(size_t)C + 1] = (uint32_t)total;
  HIP_TRY(hipSetDevice(ix->device));
  DevBuf d_boff, d_kept, d_cnt;
  HIP_TRY(d_boff.alloc(boff.size() * 4));
  HIP_TRY(d_kept.alloc(((size_t)C + 1) * 4));
  HIP_TRY(d_cnt.alloc(std::max<size_t>((size_t)C * P, 1) * 4));
  HIP_TRY(hipMemcpy(d_boff.p, boff.data(), boff.size() * 4, hipMemcpyHostToDevice));
  const Posting *buckets = (const Posting *)d_postings + o0;
  if (C > 0) {
    hipLaunchKernelGGL(count_parts_kernel, dim3((unsigned)C), dim3(256), 0, 0, P, n_shards, ix->shard_id, d_boff.as<uint32_t>(), buckets,
                       d_kept.as<uint32_t>(), d_cnt.as<uint32_t>());
    HIP_TRY(hipGetLastError());
  }
  std::vector<uint32_t> cnt((size_t)C * P);
  if (C > 0) HIP_TRY(hipMemcpy(cnt.data(), d_cnt.p, cnt.size() * 4, hipMemcpyDeviceToHost));
  ix->h_sub_offsets.resize((size_t)C * P + 1);
  uint64_t tot = 0;
  for (size_t i = 0; i < (size_t)C * P; i++) {
    ix->h_sub_offsets[i] = (uint32_t)tot;
    tot += cnt[i];
  }
  ix->h_sub_offsets[(size_t)C * P] = (uint32_t)tot;
  ix->n_postings = (int64_t)tot;
  HIP_TRY(ix->postings.alloc(std::max<uint64_t>(tot, 1) * sizeof(Posting)));
  HIP_TRY(ix->ranks.alloc(std::max<uint64_t>(tot, 1) * 4));
  HIP_TRY(ix->sub_offsets.alloc(ix->h_sub_offsets.size() * 4));
  HIP_TRY(hipMemcpy(ix->sub_offsets.p, ix->h_sub_offsets.data(), ix->h_sub_offsets.size() * 4, hipMemcpyHostToDevice));
  if (C > 0) {
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)C), dim3(256), 0, 0, P, n_shards, ix->shard_id, d_boff.as<uint32_t>(),
                       d_kept.as<uint32_t>(), buckets, ix->sub_offsets.as<uint32_t>(), ix->postings.as<Posting>(), ix->ranks.as<uint32_t>());
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipDeviceSynchronize());
  guard.p = nullptr;
  *out = ix;
  return SANN_OK;
} ABI_CATCH

// The sender side: counts (write = 0: counts[n_clusters * n_ranks] on the host) or the postings themselves (write = 1: into d_out at
// seg_offsets[cluster][dest], in postings).  Clusters the index does not hold count zero.
static int export_prefixes(sann_index_t *ix, hipStream_t st, int32_t n_clusters, const int32_t *clusters, int32_t M, int32_t n_ranks,
                           int write, int32_t *counts, const int64_t *seg_offsets, void *d_out) {
  if (!ix) return fail(SANN_EINVAL, "index is NULL");
  if (ix->n_shards != 1) return fail(SANN_EINVAL, "a cluster-range shard holds whole lists (n_shards = 1)");
  if (n_clusters < 0 || (n_clusters > 0 && !clusters) || n_ranks < 1 || n_ranks > 16) return fail(SANN_EINVAL, "bad clusters / n_ranks (1..16)");
  if (M < 0) M = 0;
  if (M > EXPORT_MAX) return fail(SANN_ELIMIT, "maxTopTweetsPerCluster above 4096");
  if (n_clusters == 0) return SANN_OK;
  if ((!write && !counts) || (write && (!seg_offsets || !d_out))) return fail(SANN_EINVAL, "NULL output");
  std::vector<int32_t> rows((size_t)n_clusters);
  std::vector<int32_t> held;
  held.reserve((size_t)n_clusters);
  std::vector<int32_t> at((size_t)n_clusters, -1);  // cluster i -> position among the held ones
  for (int32_t i = 0; i < n_clusters; i++) {
    const int r = ix->row_of(clusters[i]);
    if (r >= 0) {
      at[(size_t)i] = (int32_t)held.size();
      held.push_back(r);
    }
  }
  if (!write) std::fill(counts, counts + (size_t)n_clusters * n_ranks, 0);
  if (held.empty()) return SANN_OK;
  HIP_TRY(hipSetDevice(ix->device));
  const int n = (int)held.size();
  DevBuf d_rows, d_counts, d_seg;
  HIP_TRY(d_rows.alloc((size_t)n * 4));
  HIP_TRY(hipMemcpyAsync(d_rows.p, held.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  if (!write) {
    HIP_TRY(d_counts.alloc((size_t)n * n_ranks * 4));
  } else {
    std::vector<int64_t> seg((size_t)n * n_ranks);
    for (int32_t i = 0; i < n_clusters; i++)
      if (at[(size_t)i] >= 0)
        for (int d = 0; d < n_ranks; d++) seg[(size_t)at[(size_t)i] * n_ranks + d] = seg_offsets[(size_t)i * n_ranks + d];
    HIP_TRY(d_seg.alloc(seg.size() * 8));
    HIP_TRY(hipMemcpy(d_seg.p, seg.data(), seg.size() * 8, hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(export_prefix_kernel, dim3((unsigned)n), dim3(256), 0, st, ix->view(), d_rows.as<int32_t>(), M, n_ranks, write,
                     d_counts.as<uint32_t>(), d_seg.as<int64_t>(), (Posting *)d_out);
  HIP_TRY(hipGetLastError());
  if (!write) {
    std::vector<uint32_t> h((size_t)n * n_ranks);
    HIP_TRY(hipMemcpyAsync(h.data(), d_counts.p, h.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int32_t i = 0; i < n_clusters; i++)
      if (at[(size_t)i] >= 0)
        for (int d = 0; d < n_ranks; d++) counts[(size_t)i * n_ranks + d] = (int32_t)h[(size_t)at[(size_t)i] * n_ranks + d];
  } else {
    HIP_TRY(hipStreamSynchronize(st));  // (the temporaries above die here)
  }
  return SANN_OK;
}
int sann_index_export_prefix_counts(sann_index_t *index, void *hip_stream, int32_t n_clusters, const int32_t *clusters, int32_t M,
                                    int32_t n_ranks, int32_t *counts) try {
  return export_prefixes(index, (hipStream_t)hip_stream, n_clusters, clusters, M, n_ranks, 0, counts, nullptr, nullptr);
} ABI_CATCH
int sann_index_export_prefixes_device(sann_index_t *index, void *hip_stream, int32_t n_clusters, const int32_t *clusters, int32_t M,
                                      int32_t n_ranks, const int64_t *segment_offsets, void *d_out) try {
  return export_prefixes(index, (hipStream_t)hip_stream, n_clusters, clusters, M, n_ranks, 1, nullptr, segment_offsets, d_out);
} ABI_CATCH

int sann_synth_tweet_embeddings(int32_t device, const sann_synth_params_t *sp, int64_t t0, int32_t n, int32_t *counts,
                                int32_t *cluster_ids, double *scores) try {
  if (!sp || n < 0 || t0 < 0 || t0 + n > sp->n_tweets || (n > 0 && (!counts || !cluster_ids || !scores)))
    return fail(SANN_EINVAL, "bad arguments");
  if (n == 0) return SANN_OK;
  HIP_TRY(hipSetDevice(device));
  DevBuf dc, dl, ds;
  HIP_TRY(dc.alloc((size_t)n * 4));
  HIP_TRY(dl.alloc((size_t)n * 64 * 4));
  HIP_TRY(ds.alloc((size_t)n * 64 * 8));
  hipLaunchKernelGGL(embed_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, gen_params(sp), t0, n, dc.as<int32_t>(),
                     dl.as<int32_t>(), ds.as<double>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(counts, dc.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(cluster_ids, dl.p, (size_t)n * 64 * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(scores, ds.p, (size_t)n * 64 * 8, hipMemcpyDeviceToHost));
  return SANN_OK;
} ABI_CATCH

int sann_synth_exact_cosine_topk(int32_t device, const sann_synth_params_t *sp, int32_t nq, const int64_t *emb_offsets,
                                 const int32_t *emb_cluster_ids, const double *emb_scores, int32_t k, int64_t *out_ids,
                                 double *out_cos, int32_t *out_counts) try {
  if (!sp || nq < 0 || k < 1 || (nq > 0 && (!emb_offsets || !out_ids || !out_cos || !out_counts)))
    return fail(SANN_EINVAL, "bad arguments");
  if (nq == 0) return SANN_OK;
  if (nq > 64) return fail(SANN_ELIMIT, "at most 64 queries per exact-cosine call");
  const int C = sp->n_clusters;
  std::vector<double> wtab((size_t)nq * (C + 1), 0.0), unorm((size_t)nq, 0.0);
  for (int q = 0; q < nq; q++) {
    for (int64_t i = emb_offsets[q]; i < emb_offsets[q + 1]; i++) {
      const int32_t c = emb_cluster_ids[i];
      if (c < 1 || c > C) return fail(SANN_EINVAL, "query cluster id outside 1..n_clusters");
      if (emb_scores[i] > 0.0) wtab[(size_t)q * (C + 1) + c] = emb_scores[i];
    }
    double ss = 0.0;
    for (int c = 1; c <= C; c++) ss += wtab[(size_t)q * (C + 1) + c] * wtab[(size_t)q * (C + 1) + c];
    unorm[(size_t)q] = std::sqrt(ss);
    if (!(unorm[(size_t)q] > 0.0)) unorm[(size_t)q] = 1.0;
  }
  HIP_TRY(hipSetDevice(device));
  const uint32_t cap = 1u << 16;
  DevBuf d_w, d_n, d_hist, d_thr, d_ids, d_cos, d_cnt;
  HIP_TRY(d_w.alloc(wtab.size() * 8));
  HIP_TRY(d_n.alloc(unorm.size() * 8));
  HIP_TRY(d_hist.alloc((size_t)nq * EXACT_HB * 4));
  HIP_TRY(d_thr.alloc((size_t)nq * 8));
  HIP_TRY(d_ids.alloc((size_t)nq * cap * 8));
  HIP_TRY(d_cos.alloc((size_t)nq * cap * 8));
  HIP_TRY(d_cnt.alloc((size_t)nq * 4));
  HIP_TRY(hipMemcpy(d_w.p, wtab.data(), wtab.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_n.p, unorm.data(), unorm.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(d_hist.p, 0, (size_t)nq * EXACT_HB * 4));
  HIP_TRY(hipMemset(d_cnt.p, 0, (size_t)nq * 4));
  const GenParams g = gen_params(sp);
  const unsigned blocks = (unsigned)((sp->n_tweets + 255) / 256);
  hipLaunchKernelGGL(exact_cosine_kernel, dim3(blocks), dim3(256), 0, 0, g, nq, d_w.as<double>(), d_n.as<double>(), 0,
                     d_hist.as<uint32_t>(), (const double *)nullptr, (int64_t *)nullptr, (double *)nullptr,
                     (uint32_t *)nullptr, 0u);
  HIP_TRY(hipGetLastError());
  std::vector<uint32_t> hist((size_t)nq * EXACT_HB);
  HIP_TRY(hipMemcpy(hist.data(), d_hist.p, hist.size() * 4, hipMemcpyDeviceToHost));
  std::vector<double> thr((size_t)nq, 0.0);
  for (int q = 0; q < nq; q++) {
    uint64_t cum = 0;
    int b = EXACT_HB - 1;
    for (; b >= 0; b--) {
      cum += hist[(size_t)q * EXACT_HB + b];
      if (cum >= (uint64_t)k) break;
    }
    thr[(size_t)q] = b <= 0 ? 0.0 : (double)b / EXACT_HB;  // lower edge of the bin that completes the top-k
    if (cum > cap) return fail(SANN_ELIMIT, "exact cosine: too many candidates share the threshold bin");
  }
  HIP_TRY(hipMemcpy(d_thr.p, thr.data(), thr.size() * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(exact_cosine_kernel, dim3(blocks), dim3(256), 0, 0, g, nq, d_w.as<double>(), d_n.as<double>(), 1,
                     d_hist.as<uint32_t>(), d_thr.as<double>(), d_ids.as<int64_t>(), d_cos.as<double>(),
                     d_cnt.as<uint32_t>(), cap);
  HIP_TRY(hipGetLastError());
  std::vector<uint32_t> cnt((size_t)nq);
  HIP_TRY(hipMemcpy(cnt.data(), d_cnt.p, cnt.size() * 4, hipMemcpyDeviceToHost));
  std::vector<int64_t> ids(cap);
  std::vector<double> cs(cap);
  std::vector<uint32_t> order(cap);
  for (int q = 0; q < nq; q++) {
    const uint32_t n = std::min(cnt[(size_t)q], cap);
    HIP_TRY(hipMemcpy(ids.data(), d_ids.as<int64_t>() + (size_t)q * cap, (size_t)n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(cs.data(), d_cos.as<double>() + (size_t)q * cap, (size_t)n * 8, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    std::sort(order.begin(), order.begin() + n, [&](uint32_t a, uint32_t b2) {
      return cs[a] > cs[b2] || (cs[a] == cs[b2] && ids[a] < ids[b2]);
    });
    const int m = (int)std::min<uint32_t>(n, (uint32_t)k);
    for (int i = 0; i < m; i++) {
      out_ids[(size_t)q * k + i] = ids[order[(size_t)i]];
      out_cos[(size_t)q * k + i] = cs[order[(size_t)i]];
    }
    out_counts[q] = m;
  }
  return SANN_OK;
} ABI_CATCH

int64_t sann_synth_tweet_id(int64_t t, int64_t n_tweets, int64_t now_ms, int32_t window_hours) {
  const int64_t span = (int64_t)window_hours * 3600000ll;
  return synth_tweet_id(t, n_tweets, now_ms - span, span);
}

}  // extern "C"
