// sann_fast.hip -- fast path of the (query, partition) work unit, gfx950.
//
// Two kernels.
//
// desc_kernel   one thread per (query, scanned cluster, partition): sub-list start and the number
//               of its postings with rank < M (binary search in `ranks`; the `i < min(size, M)` cut
//               of ApproximateCosineSimilarity.scala:87).  The lookups are a chain of dependent
//               loads (scan_row -> sub_offsets -> ranks); doing them here, millions of independent
//               threads wide, keeps that chain out of the unit kernel, whose own chain is then just
//               descriptors -> postings.
//
// unit_fast_kernel   one workgroup = one unit; the unit's postings live in REGISTERS (U per
//               thread), LDS holds only a 64-Kbit presence bitmap and small side tables:
//   1. descriptors  coalesced read of the unit's (start, len) row; exclusive scan -> flat index
//   2. gather       flat posting index -> (cluster, position) by binary search over the scan; one
//                   16-B global load per posting, consecutive lanes = consecutive postings of a
//                   sub-list; age window and source-tweet filters (:90-91)
//   3. duplicates   a tweet can sit in several scanned clusters (all its postings are in this unit
//                   by construction of the partition hash).  Each posting sets bit hash(id) in the
//                   bitmap; finding the bit already set flags the id as "possibly seen before".
//                   Flagged ids (true duplicates and a few hash collisions) are matched against
//                   every thread's registers; real groups are summed by one thread in cluster
//                   order, so fp64 sums follow the reference's accumulation order (:83-100)
//                   whatever the thread timing.  Unflagged postings need no LDS traffic at all.
//   4. finalise     (dot, nsq) -> score (:111-119), `>= minScore` (:125), monotone 64-bit key
//   5. threshold    MSB-first radix histogram over the keys from the highest bit in which the
//                   unit's keys differ, stopping once "everything >= this digit" is between k_local
//                   and cap entries
//   6. emit         all candidates with key >= threshold (an exact upper set of the unit) and the
//                   threshold itself, so that the merge can prove the global top-k exact.
//
// Units that do not fit (too many clusters / postings / flagged ids, or an unresolvable tie
// group) flag UNIT_OVERFLOW and are re-run by unit_general_kernel.
#include <hip/hip_runtime.h>

#include "sann_device.h"
#include "sann_kernels.h"
#include "sann_math.h"
#include "sann_select.h"

namespace sann {

constexpr int NSCAN_MAX = 128;  // scanned clusters a fast unit can describe
constexpr int LCAP = 64;        // flagged ids per unit
constexpr int MCAP = 192;       // postings matching a flagged id per unit
constexpr int BM_WORDS = 2048;  // 65536-bit presence bitmap

__device__ inline double normalise_f(int alg, double dot, double nsq, double l2norm, double lognorm) {
  switch (alg) {
    case 3: return dot / lognorm / strict_log(1 + nsq);
    case 2: return dot / l2norm / sqrt(nsq);
    case 4: return dot / sqrt(nsq);
    case 1: return dot;
    default: return __builtin_nan("");
  }
}

__device__ inline int lower_bound_rank(const uint32_t *a, int n, uint32_t v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void desc_kernel(IndexView ix, BatchView b, int total_scan) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int e = t >> ix.log2P;
  const int p = t & (ix.P - 1);
  if (e >= total_scan) return;
  const int q = b.scan_q[e];
  const int M = b.hdr[q].M;
  const int scan_begin = b.hdr[q].scan_begin;
  const int n_scan = b.hdr[q].n_scan;
  const int row = b.scan_row[e];
  const uint32_t base = ix.sub_offsets[(int64_t)row * ix.P + p];
  const uint32_t end = ix.sub_offsets[(int64_t)row * ix.P + p + 1];
  const int n = (int)(end - base);
  // postings with rank < M are a prefix of the sub-list
  const uint32_t len = (n > 0 && ix.ranks[base + n - 1] < (uint32_t)M) ? (uint32_t)n
                                                                     : (uint32_t)lower_bound_rank(ix.ranks + base, n, (uint32_t)M);
  // unit-major layout: the (q, p) unit reads n_scan consecutive entries
  const int64_t o = (int64_t)scan_begin * ix.P + (int64_t)p * n_scan + (e - scan_begin);
  b.desc[2 * o] = base;
  b.desc[2 * o + 1] = len;
}

enum { CTL_NFLAG = 0, CTL_NM, CTL_UNIQ, CTL_NVALID, CTL_CNT, CTL_SEL_D, CTL_SEL_A, CTL_SEL_B };

__device__ inline unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    unsigned long long o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}
__device__ inline unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    unsigned long long o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}
__device__ inline int wave_sum_i32(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int WG, int U>
__global__ __launch_bounds__(WG) void unit_fast_kernel(IndexView ix, BatchView b, int k_local_floor) {
  __shared__ uint32_t s_bm[BM_WORDS];
  __shared__ uint32_t s_begin[NSCAN_MAX];
  __shared__ uint32_t s_pre[NSCAN_MAX + 1];
  __shared__ uint32_t s_len[NSCAN_MAX];
  __shared__ double s_w[NSCAN_MAX];
  __shared__ unsigned long long s_L[LCAP];
  __shared__ uint16_t s_Mf[MCAP], s_Mseq[MCAP];
  __shared__ double s_Msc[MCAP];
  __shared__ double s_gdot[LCAP], s_gnsq[LCAP];
  __shared__ int s_gsize[LCAP], s_grep[LCAP];
  __shared__ unsigned s_hist[256];
  __shared__ int s_ctl[8];
  __shared__ unsigned long long s_minmax[2];

  const int tid = threadIdx.x;
  const int unit = blockIdx.x;
  const int q = unit >> ix.log2P;
  const int p = unit & (ix.P - 1);
  const QueryHdr h = b.hdr[q];

  // ---- 0. clear ----------------------------------------------------------------------------
  if (tid < 8) s_ctl[tid] = 0;
  if (tid == 0) { s_minmax[0] = ~0ull; s_minmax[1] = 0ull; }
  for (int i = tid; i < BM_WORDS; i += WG) s_bm[i] = 0;
  for (int i = tid; i <= NSCAN_MAX; i += WG) s_pre[i] = 0xffffffffu;

  bool overflow = h.n_scan > NSCAN_MAX;  // uniform
  // ---- 1. descriptors ----------------------------------------------------------------------
  if (!overflow) {
    const uint32_t *d = b.desc + 2 * ((int64_t)h.scan_begin * ix.P + (int64_t)p * h.n_scan);
    for (int c = tid; c < h.n_scan; c += WG) {
      const uint2 v = *reinterpret_cast<const uint2 *>(d + 2 * c);
      s_begin[c] = v.x;
      s_len[c] = v.y;
      s_w[c] = b.scan_w[h.scan_begin + c];
    }
  }
  __syncthreads();
  if (!overflow && tid < 64) {
    const int lane = tid;
    uint32_t a0 = (2 * lane < h.n_scan) ? s_len[2 * lane] : 0;
    uint32_t a1 = (2 * lane + 1 < h.n_scan) ? s_len[2 * lane + 1] : 0;
    uint32_t s = a0 + a1, incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    uint32_t excl = incl - s;
    if (2 * lane <= h.n_scan) s_pre[2 * lane] = excl;
    if (2 * lane + 1 <= h.n_scan) s_pre[2 * lane + 1] = excl + a0;
  }
  __syncthreads();
  const uint32_t T = overflow ? 0u : s_pre[h.n_scan];
  if (!overflow && T > (uint32_t)(WG * U)) overflow = true;

  // ---- 2. gather (postings stay in registers) -------------------------------------------------
  long long id[U];
  double sc[U];     // posting score, later the candidate's monotone key bits
  int seq[U];       // cluster sequence number; -1 = no posting / filtered / consumed
  if (!overflow) {
    Posting pst[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t j = (uint32_t)(u * WG + tid);
      int c = 0;
#pragma unroll
      for (int step = NSCAN_MAX / 2; step >= 1; step >>= 1) {
        const int t = c + step;
        if (s_pre[t] <= j) c = t;  // entries past n_scan are 0xffffffff
      }
      seq[u] = (j < T) ? c : -1;
      pst[u].id = 0;
      pst[u].score = 0.0;
      if (j < T) pst[u] = ix.postings[s_begin[c] + (j - s_pre[c])];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      // NB: written as selects, not `seq[u] = -1; continue;` -- hipcc (ROCm 7.2) mis-structurised
      // that form and dropped the -1 for postings outside the age window.
      const bool have = seq[u] >= 0;
      id[u] = have ? pst[u].id : 0;
      sc[u] = have ? pst[u].score : 0.0;
      const bool excluded = h.excl_enabled != 0 && id[u] == h.src_excl;  // :90
      const bool in_window = id[u] >= h.earliest && id[u] <= h.latest;   // :91
      const bool keep = have && !excluded && in_window;
      seq[u] = keep ? seq[u] : -1;
      if (keep) {
        // ---- 3a. presence bitmap -------------------------------------------------------------
        const uint32_t hb = table_hash(id[u], 16);
        const uint32_t bit = 1u << (hb & 31);
        const uint32_t old = atomicOr(&s_bm[hb >> 5], bit);
        if (old & bit) {
          const int f = atomicAdd(&s_ctl[CTL_NFLAG], 1);
          if (f < LCAP) s_L[f] = (unsigned long long)id[u];
        }
      }
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; u++) seq[u] = -1;
  }
  __syncthreads();
  const int nflag = s_ctl[CTL_NFLAG];
  if (nflag > LCAP) overflow = true;

  // ---- 3b. resolve flagged ids ------------------------------------------------------------------
  int consumed = 0;
  if (!overflow && nflag > 0) {
    int gf[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      gf[u] = -1;
      if (seq[u] >= 0)
        for (int f = 0; f < nflag; f++)
          if (s_L[f] == (unsigned long long)id[u]) { gf[u] = f; break; }  // smallest f = canonical group
      if (gf[u] >= 0) {
        const int m = atomicAdd(&s_ctl[CTL_NM], 1);
        if (m < MCAP) {
          s_Mf[m] = (uint16_t)gf[u];
          s_Mseq[m] = (uint16_t)seq[u];
          s_Msc[m] = sc[u];
        }
      }
    }
    __syncthreads();
    const int nm = s_ctl[CTL_NM];
    if (nm > MCAP) {
      overflow = true;
    } else {
      for (int f = tid; f < nflag; f += WG) {
        int cnt = 0, rep = 0x7fffffff;
        for (int m = 0; m < nm; m++)
          if (s_Mf[m] == f) { cnt++; rep = s_Mseq[m] < rep ? s_Mseq[m] : rep; }
        double dot = 0.0, nsq = 0.0;
        if (cnt >= 2) {
          int last = -1;
          for (int r = 0; r < cnt; r++) {  // ascending cluster sequence
            int best = 0x7fffffff;
            double bs = 0.0;
            for (int m = 0; m < nm; m++)
              if (s_Mf[m] == f) {
                const int se = s_Mseq[m];
                if (se > last && se < best) { best = se; bs = s_Msc[m]; }
              }
            dot = dot + bs * s_w[best];  // :92-94
            nsq = nsq + bs * bs;         // :95-96
            last = best;
          }
        }
        s_gsize[f] = cnt;
        s_grep[f] = rep;
        s_gdot[f] = dot;
        s_gnsq[f] = nsq;
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < U; u++) {
        const bool grouped = seq[u] >= 0 && gf[u] >= 0 && s_gsize[gf[u] >= 0 ? gf[u] : 0] >= 2;
        const bool is_rep = grouped && seq[u] == s_grep[gf[u] >= 0 ? gf[u] : 0];
        // representative: carries the group's sums; the others are folded into it
        seq[u] = grouped ? (is_rep ? (0x10000 | gf[u]) : -1) : seq[u];
        consumed += (grouped && !is_rep) ? 1 : 0;
      }
    }
  }

  if (overflow) {
    if (tid == 0) {
      b.cand_cnt[unit] = 0;
      b.unit_unique[unit] = 0;
      b.unit_flags[unit] = UNIT_OVERFLOW;
      b.unit_thr[2 * (int64_t)unit] = 0;
      b.unit_thr[2 * (int64_t)unit + 1] = 0;
      const int o = atomicAdd(&b.status[0], 1);
      b.overflow_units[o] = unit;
    }
    return;
  }

  // ---- 4. finalise (registers only) ---------------------------------------------------------------
  int uniq = 0, nval = 0;
  unsigned long long kmin = ~0ull, kmax = 0ull;
#pragma unroll
  for (int u = 0; u < U; u++) {
    const bool live = seq[u] >= 0;
    if (live) {
      uniq++;
      double dot, nsq;
      if (seq[u] & 0x10000) {
        dot = s_gdot[seq[u] & 0xffff];
        nsq = s_gnsq[seq[u] & 0xffff];
      } else {
        dot = 0.0 + sc[u] * s_w[seq[u]];  // getOrElse(tweetId, 0.0) + score * sourceClusterScore
        nsq = 0.0 + sc[u] * sc[u];
      }
      const double v = normalise_f(h.alg, dot, nsq, h.l2norm, h.lognorm);
      const bool cand = v >= h.min_score;  // :125 (false for NaN)
      const unsigned long long key = score_key(v);
      sc[u] = cand ? bits_f64(key) : sc[u];
      nval += cand ? 1 : 0;
      kmin = (cand && key < kmin) ? key : kmin;
      kmax = (cand && key > kmax) ? key : kmax;
      // a non-candidate still counted in candidateScoresMap.size above
      seq[u] = cand ? seq[u] : -2;
    }
  }
  {
    const int wu = wave_sum_i32(uniq), wv = wave_sum_i32(nval);
    const unsigned long long wmin = wave_min_u64(kmin), wmax = wave_max_u64(kmax);
    if ((tid & 63) == 0) {
      if (wu) atomicAdd(&s_ctl[CTL_UNIQ], wu);
      if (wv) {
        atomicAdd(&s_ctl[CTL_NVALID], wv);
        atomicMin(&s_minmax[0], wmin);
        atomicMax(&s_minmax[1], wmax);
      }
    }
  }
  __syncthreads();
  const int n_valid = s_ctl[CTL_NVALID];
  const int cap = b.cap;

  // ---- 5. threshold ---------------------------------------------------------------------------------
  int kl;
  {
    const float share = (float)h.k / (float)ix.P;
    kl = (int)(share + 6.0f * sqrtf(share) + 8.0f);
    if (kl < k_local_floor) kl = k_local_floor;
    if (kl > h.k) kl = h.k;
    if (kl > cap) kl = cap;
  }
  unsigned long long thr = 0;
  bool give_up = false;
  const int emit_all = cap < kl + kl / 2 + 16 ? cap : kl + kl / 2 + 16;
  if (n_valid > emit_all) {
    const unsigned long long gmin = s_minmax[0], gmax = s_minmax[1];
    const unsigned long long diff = gmin ^ gmax;
    int shift = 0, width = 0;
    unsigned long long prefix = gmax;
    bool done = false;
    if (diff == 0) {
      give_up = n_valid > cap;  // every score identical: cannot cut by score
      done = true;
    } else {
      const int hbit = 63 - __clzll((long long)diff);
      shift = hbit - 7 < 0 ? 0 : hbit - 7;
      width = hbit - shift + 1;
      prefix = (hbit == 63) ? 0ull : (gmax >> (hbit + 1)) << (hbit + 1);
    }
    int need = kl, budget = cap;
    while (!done) {
      for (int i = tid; i < 256; i += WG) s_hist[i] = 0;
      __syncthreads();
      const unsigned long long hi_mask = (shift + width >= 64) ? 0ull : (~0ull << (shift + width));
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (seq[u] >= 0) {
          const unsigned long long key = f64_bits(sc[u]);
          if ((key & hi_mask) == (prefix & hi_mask))
            atomicAdd(&s_hist[(unsigned)((key >> shift) & ((1u << width) - 1))], 1u);
        }
      }
      __syncthreads();
      if (tid < 64) wave_find_digit(s_hist, need, &s_ctl[CTL_SEL_D]);  // writes D, A, B
      __syncthreads();
      const int d = s_ctl[CTL_SEL_D], A = s_ctl[CTL_SEL_A], B = s_ctl[CTL_SEL_B];
      prefix |= (unsigned long long)d << shift;
      if (A + B <= budget) {
        done = true;
      } else if (shift == 0) {
        give_up = true;  // more exactly-equal scores than the unit may emit
        done = true;
      } else {
        need -= A;
        budget -= A;
        const int ns = shift - 8 < 0 ? 0 : shift - 8;
        width = shift - ns;
        shift = ns;
      }
      __syncthreads();
    }
    thr = prefix;
  }
  if (give_up) {
    if (tid == 0) {
      b.cand_cnt[unit] = 0;
      b.unit_unique[unit] = 0;
      b.unit_flags[unit] = UNIT_OVERFLOW;
      b.unit_thr[2 * (int64_t)unit] = 0;
      b.unit_thr[2 * (int64_t)unit + 1] = 0;
      const int o = atomicAdd(&b.status[0], 1);
      b.overflow_units[o] = unit;
    }
    return;
  }

  // ---- 6. emit -----------------------------------------------------------------------------------------
  const int64_t obase = (int64_t)unit * cap;
#pragma unroll
  for (int u = 0; u < U; u++) {
    if (seq[u] >= 0) {
      const unsigned long long key = f64_bits(sc[u]);
      if (key >= thr) {
        const int o = atomicAdd(&s_ctl[CTL_CNT], 1);
        if (o < cap) {
          b.cand_key[obase + o] = key;
          b.cand_id[obase + o] = id[u];
        }
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    const int cnt = s_ctl[CTL_CNT];
    b.cand_cnt[unit] = cnt < cap ? cnt : cap;
    b.unit_unique[unit] = s_ctl[CTL_UNIQ];
    b.unit_flags[unit] = (n_valid > cnt) ? UNIT_TRUNCATED : UNIT_OK;
    b.unit_thr[2 * (int64_t)unit] = thr;
    b.unit_thr[2 * (int64_t)unit + 1] = 0;
  }
  (void)consumed;
}

template <int WG, int U>
static hipError_t launch_one(const IndexView &ix, const BatchView &b, const FastParams &fp, int n_units,
                             hipStream_t stream) {
  hipLaunchKernelGGL((unit_fast_kernel<WG, U>), dim3(n_units), dim3(WG), 0, stream, ix, b, fp.k_local);
  return hipGetLastError();
}

hipError_t launch_desc(const IndexView &ix, const BatchView &b, int total_scan, hipStream_t stream) {
  const int64_t n = (int64_t)total_scan * ix.P;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(desc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ix, b, total_scan);
  return hipGetLastError();
}

// fp.unit_capacity = postings one unit may hold = WG * U
hipError_t launch_unit_fast(const IndexView &ix, const BatchView &b, const FastParams &fp, int n_units,
                            hipStream_t stream) {
  if (n_units <= 0) return hipSuccess;
  switch (fp.unit_capacity) {
    case 256: return launch_one<64, 4>(ix, b, fp, n_units, stream);
    case 512: return launch_one<128, 4>(ix, b, fp, n_units, stream);
    case 1024: return launch_one<256, 4>(ix, b, fp, n_units, stream);
    case 2048: return launch_one<256, 8>(ix, b, fp, n_units, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace sann
