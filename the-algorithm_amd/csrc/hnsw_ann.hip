// hnsw_ann.hip -- HNSW graph search on gfx950 (include/hnsw_ann.h) + the host-side graph builder.
//
// Reference (paths relative to /root/reference/ann/src/main/java/com/twitter/ann/hnsw/):
//   HnswIndex.java:538-553   searchKnn            HnswIndex.java:447-475  bestEntryPointUntilLayer
//   HnswIndex.java:571-623   searchLayerForCandidates
//   HnswIndex.java:137-200,384-440,479-526  insert / mutuallyConnectNewElement / selectNearestNeighboursByHeuristic
//   DistancedItemQueue.java:37-43           queues = java.util.PriorityQueue on Float.compare(distance)
//
// The walk is inherently sequential -- which neighbour is admitted depends on the queue state the
// previous one left -- so the GPU does not parallelise the walk of one query; it runs ~1500 of them
// at once (one wave per query, six waves per CU) and inside a step parallelises what IS parallel:
//   * visited test-and-set for a whole neighbour list: one atomicOr per lane on a per-query bitmap;
//   * distances of the unvisited neighbours: 8 lanes per vector (16-B loads, 128 B contiguous per
//     group), 8 vectors per round, all rounds' loads issued before the first reduction;
//   * then the reference's own admission loop, in list order, on the two queues.
// The queues are java.util.PriorityQueue restated (same siftUp / siftDown), operated wave-uniformly; that makes equal
// distances come out as on the JVM.  What bounds the kernel is how many walks a CU holds -- a walk is one long dependent
// chain of heap steps and gathers, a wave issues an instruction every 3-4 cycles at best (profiles/r03_pmc_c4_dense_hnsw.txt),
// and the LDS a walk needs decides how many waves share a SIMD.  So the LDS holds what is hot and no more: the result queue
// (beam + 1 entries; every admission sifts through it twice) and the TOP of the candidate queue; the candidate queue's tail
// -- which an offer touches at one or two levels and only a poll (one per expansion) descends into -- lives in global
// memory, L2-resident.  ~10 KB per walk = 16 walks per CU at any beam (round 2: 23-39 KB at beam 800, 4-7 walks per CU).
// A query whose candidate queue outgrows the first pass's global part is re-run with a larger one.
// HBM-latency bound by nature: every expansion is a dependent gather of <= 2*maxM random 512-B rows.
//
// Distance arithmetic (the fixed order oracle/hnsw_oracle.c repeats): operands rounded to fp16, fp32
// products and sums, lane j of 8 sums chunks j, j+8, ... of 8 consecutive elements, then a pairwise
// butterfly (xor 1, 2, 4).  Compiled with -ffp-contract=off, so host (builder) == device == oracle.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hnsw_ann.h"
#include "sann_device.h"  // mix64
#include "abi_guard.h"
#define ABI_CATCH catch (...) { return abi_guard::caught(fail, HNSW_ENOMEM, HNSW_EINTERNAL); }

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &m) {
  g_err = m;
  return code;
}
#define HTRY(expr)                                                                                 \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(HNSW_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

constexpr int MAX_D = 512;
constexpr int MAX_M = 32;        // lists of <= 2*MAX_M = 64 neighbours: one lane each
constexpr int MAX_EF = 1024;
constexpr int CCAP_LDS = 1024;          // most candidate-queue entries kept in LDS (the queue's top)
constexpr int WALK_LDS_BYTES = 9 * 1024;  // dynamic LDS of a walk: result queue + candidate-queue top (16 walks per CU with the static 0.5 KB)
constexpr int CCAP_FIRST = 8192;        // global part of the candidate queue, first pass (64 KB per query)
constexpr int CCAP_GLOBAL = 1 << 17;    // ... of the re-run of a query that outgrew it
constexpr int64_t VLOG_MIN_VWORDS = 1 << 18;  // bitmaps of at least 1 MB keep an undo log
constexpr int VLOG_CAP = 1 << 16;       // visited nodes a walk remembers for cleaning up after itself (more: it wipes its whole bitmap)

struct Buf {
  void *p = nullptr;
  size_t bytes = 0;
  ~Buf() { if (p) (void)hipFree(p); }
  hipError_t reserve(size_t n) {
    if (n <= bytes && p) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&p, n ? n : 8);
    if (e == hipSuccess) bytes = n ? n : 8;
    return e;
  }
  template <class T> T *as() const { return (T *)p; }
};

// ---------------------------------------------------------------------------------------------
// shared host/device pieces: Float.compare and java.util.PriorityQueue
// ---------------------------------------------------------------------------------------------
struct HEntry {
  float dist;
  uint32_t node;
};

__host__ __device__ inline int32_t float_to_int_bits(float f) {  // Float.floatToIntBits: canonical NaN
  if (f != f) return 0x7fc00000;
#ifdef __HIP_DEVICE_COMPILE__
  return __float_as_int(f);
#else
  int32_t i;
  std::memcpy(&i, &f, 4);
  return i;
#endif
}
__host__ __device__ inline int float_compare(float a, float b) {  // java.lang.Float.compare
  if (a < b) return -1;
  if (a > b) return 1;
  const int32_t x = float_to_int_bits(a), y = float_to_int_bits(b);
  return x == y ? 0 : (x < y ? -1 : 1);
}
// comparator of a DistancedItemQueue (DistancedItemQueue.java:37-43)
template <bool MINQ>
__host__ __device__ inline int qcmp(const HEntry &a, const HEntry &b) {
  return MINQ ? float_compare(a.dist, b.dist) : float_compare(b.dist, a.dist);
}
// PriorityQueue.offer -> siftUpUsingComparator
template <bool MINQ>
__host__ __device__ inline void pq_add(HEntry *q, int &n, HEntry x) {
  int k = n++;
  while (k > 0) {
    const int parent = (k - 1) >> 1;
    const HEntry e = q[parent];
    if (qcmp<MINQ>(x, e) >= 0) break;
    q[k] = e;
    k = parent;
  }
  q[k] = x;
}
// PriorityQueue.poll -> siftDownUsingComparator
template <bool MINQ>
__host__ __device__ inline HEntry pq_poll(HEntry *q, int &n) {
  const HEntry result = q[0];
  const int s = --n;
  if (s > 0) {
    const HEntry x = q[s];
    int k = 0;
    const int half = s >> 1;
    while (k < half) {
      int child = 2 * k + 1;
      HEntry c = q[child];
      const int right = child + 1;
      if (right < s) {
        const HEntry r = q[right];
        if (qcmp<MINQ>(c, r) > 0) {
          c = r;
          child = right;
        }
      }
      if (qcmp<MINQ>(x, c) <= 0) break;
      q[k] = c;
      k = child;
    }
    q[k] = x;
  }
  return result;
}

// ---- the search kernel's queues -------------------------------------------------------------------------------------
// Entries carry the distance as an integer KEY whose signed order is Float.compare's (floatToIntBits, then the low 31
// bits flipped for negatives: -0.0 < +0.0, NaN -- canonical -- above +inf), so a heap comparison is one integer compare.
struct KEntry {
  int32_t key;
  uint32_t node;
};
__device__ __forceinline__ int32_t dist_key(float f) {
  const int32_t b = float_to_int_bits(f);
  return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float key_dist(int32_t k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }
template <bool MINQ>
__device__ __forceinline__ bool k_before(int32_t a, int32_t b) {  // comparator(a, b) < 0
  return MINQ ? a < b : a > b;
}
// Every lane of the wave runs the same queue code on the same data, but a value that comes back from LDS or global memory is
// a vector register as far as the compiler can tell, and every loop and branch that depends on it becomes exec-mask
// bookkeeping (s_and_saveexec / s_or / s_andn2 on masks that are always all-ones: ~35 instructions per sift level in the
// round-3 ISA).  v_readfirstlane says what we know -- the value is wave-uniform -- and counters, heap positions and
// comparisons move to scalar registers, branches become plain s_cbranch (~25 per level).  Measured: no change in kernel time
// (25.0 ms either way at beam 800, 1M x 256): the walk is bound by its chain of dependent LDS / memory round trips and by how
// many walks share a SIMD, not by instruction issue.  Kept because the ISA is the one a reader expects.
__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ KEntry uni(const KEntry &e) { return KEntry{uni(e.key), uni(e.node)}; }

// The visited set is an n-bit bitmap per walk -- 6.25 MB at 50M vectors -- of which a walk sets a few thousand bits.  Wiping it
// for every walk (a memset of every bitmap per search launch, or the builder's per-layer wipe) was 5 ms of a 4096-query batch
// and a third of the build at 50M.  Instead every walk LOGS the nodes it marks (VLOG_CAP entries) and clears exactly those
// words when it is done, so a bitmap that starts clean -- one memset when the buffer is allocated -- is clean again for the next
// walk in the slot.  A walk that marks more than the log holds wipes its whole bitmap.  Used from 1 MB per bitmap (8M vectors)
// up: below that the wipes are cheap and the log's stores are not (1M vectors: walk kernel 24.0 -> 25.0 ms with the log).
__device__ __forceinline__ void visited_undo(uint32_t *vis, int64_t vwords, const uint32_t *vlog, int n_log, int lane) {
  __threadfence_block();  // (the log was written by other lanes of this wave)
  if (n_log <= VLOG_CAP) {
    for (int i = lane; i < n_log; i += 64) vis[vlog[i] >> 5] = 0u;
  } else {
    for (int64_t w = lane; w < vwords; w += 64) vis[w] = 0u;
  }
  __threadfence_block();
}

// The candidate queue of a walk: entries [0, nl) in LDS, the rest in global memory (see the file header).
struct HybQ {
  KEntry *lds;
  KEntry *glb;
  int nl;
  __device__ __forceinline__ KEntry get(int i) const { return uni(i < nl ? lds[i] : glb[i - nl]); }
  __device__ __forceinline__ void set(int i, const KEntry &e) const {
    if (i < nl) lds[i] = e;
    else glb[i - nl] = e;
  }
};
template <bool MINQ>
__device__ inline void hq_add(const HybQ &q, int &n, KEntry x) {  // PriorityQueue.offer -> siftUpUsingComparator
  int k = n++;
  while (k > 0) {
    const int parent = (k - 1) >> 1;
    const KEntry e = q.get(parent);
    if (!k_before<MINQ>(x.key, e.key)) break;
    q.set(k, e);
    k = parent;
  }
  q.set(k, x);
}
template <bool MINQ>
__device__ inline KEntry hq_poll(const HybQ &q, int &n) {  // PriorityQueue.poll -> siftDownUsingComparator
  const KEntry result = q.get(0);
  const int s = --n;
  if (s > 0) {
    const KEntry x = q.get(s);
    int k = 0;
    const int half = s >> 1;
    while (k < half) {
      int child = 2 * k + 1;
      KEntry c = q.get(child);
      const int right = child + 1;
      if (right < s) {
        const KEntry r = q.get(right);
        if (k_before<MINQ>(r.key, c.key)) {  // comparator(c, r) > 0
          c = r;
          child = right;
        }
      }
      if (!k_before<MINQ>(c.key, x.key)) break;  // comparator(x, c) <= 0
      q.set(k, c);
      k = child;
    }
    q.set(k, x);
  }
  return result;
}
// plain LDS heap (the result queue): the same two operations without the LDS / global split
template <bool MINQ>
__device__ inline void kq_add(KEntry *q, int &n, KEntry x) {
  int k = n++;
  while (k > 0) {
    const int parent = (k - 1) >> 1;
    const KEntry e = uni(q[parent]);
    if (!k_before<MINQ>(x.key, e.key)) break;
    q[k] = e;
    k = parent;
  }
  q[k] = x;
}
template <bool MINQ>
__device__ inline KEntry kq_poll(KEntry *q, int &n) {
  const KEntry result = uni(q[0]);
  const int s = --n;
  if (s > 0) {
    const KEntry x = uni(q[s]);
    int k = 0;
    const int half = s >> 1;
    while (k < half) {
      int child = 2 * k + 1;
      KEntry c = uni(q[child]);
      const int right = child + 1;
      if (right < s) {
        const KEntry r = uni(q[right]);
        if (k_before<MINQ>(r.key, c.key)) {
          c = r;
          child = right;
        }
      }
      if (!k_before<MINQ>(c.key, x.key)) break;
      q[k] = c;
      k = child;
    }
    q[k] = x;
  }
  return result;
}

// distance from the fixed-order sum (see file header)
__host__ __device__ inline float finish_distance(int metric, float s) {
  return metric == HNSW_METRIC_L2 ? sqrtf(s) : 1.0f - s;
}

// ---------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct SearchArgs {
  const _Float16 *x;          // [n][dpad]
  const _Float16 *q;          // [nq][dpad] prepared queries
  const uint32_t *adj0;       // [n][m0 + 1]: count, neighbours
  const int32_t *upper_slot;  // [n]: row group of a node with levels >= 1, or -1
  const int32_t *upper_base;  // [n_upper + 1]: first row of the slot; rows = levels 1..top
  const uint32_t *upper_adj;  // [rows][m + 1]
  const int64_t *ids;         // or NULL
  const int32_t *qlist;       // queries of this launch (spill re-run) or NULL = blockIdx
  uint32_t *visited;          // [slots][vwords]: clean when a walk starts, cleaned by the walk when it ends
  uint32_t *vlog;             // [slots][VLOG_CAP]: the nodes the walk marked; NULL = small bitmaps: the host wipes them per launch
  KEntry *gc;                 // candidate-queue tails: [slots][gcap]
  float *out_dist;            // [nq][k]
  int64_t *out_ids;
  int32_t *out_counts;
  int32_t *spill;             // [nq] set when the LDS candidate queue overflowed
  unsigned long long *stats;  // [0] distance evaluations, [1] expansions, [2] largest candidate queue, [3] admissions
  int64_t vwords;
  int32_t dpad, chunks, m, m0, metric, k, ef, max_level;
  int32_t ccap_lds;           // candidate-queue entries in LDS (the top of the heap)
  int32_t gcap;               // ... and in global memory, per slot
  uint32_t entry;
};

// distances of nu nodes (ids in nodes[], LDS) to the wave's query; results to dists[] (LDS)
template <int CH, class Args>  // chunks of 8 halves per lane: dpad / 64; Args = SearchArgs or BuildArgs (x, dpad, metric)
__device__ __forceinline__ void wave_distances(const Args &a, const float (&qv)[CH][8], const uint32_t *nodes,
                                               float *dists, int nu, int lane) {
  const int g = lane >> 3, j = lane & 7;
  for (int base = 0; base < nu; base += 32) {  // up to 4 rounds of 8 vectors in flight
    half8 xv[4][CH];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = base + r * 8 + g;
      if (idx < nu) {
        const half8 *row = (const half8 *)(a.x + (size_t)nodes[idx] * a.dpad);
#pragma unroll
        for (int c = 0; c < CH; ++c) xv[r][c] = row[j + 8 * c];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = base + r * 8 + g;
      float acc = 0.0f;
      if (idx < nu) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xe = (float)xv[r][c][e];
            if (a.metric == HNSW_METRIC_L2) {
              const float t = qv[c][e] - xe;
              acc = acc + t * t;
            } else {
              // one v_fma_mix_f32 (the fp16 operand converted by the instruction) instead of cvt + mul + add -- and the same
              // bits: both factors are fp16 values, so their product (22 significant bits) is exact in fp32 and the fused
              // form rounds once exactly where `acc + q * x` rounds
              acc = __builtin_fmaf(qv[c][e], xe, acc);
            }
          }
      }
      acc = acc + __shfl_xor(acc, 1, 64);
      acc = acc + __shfl_xor(acc, 2, 64);
      acc = acc + __shfl_xor(acc, 4, 64);
      if (idx < nu && j == 0) dists[idx] = finish_distance(a.metric, acc);
    }
  }
}

template <int CH>
__global__ __launch_bounds__(64) void hnsw_search_kernel(SearchArgs a) {
  extern __shared__ KEntry dyn_s[];  // result queue [ef + 1], then the top of the candidate queue [ccap_lds]
  KEntry *wq = dyn_s;
  __shared__ uint32_t ul[64];
  __shared__ float ud[64];
  const int lane = threadIdx.x;
  const int slot = blockIdx.x;
  const int qi = a.qlist ? a.qlist[slot] : slot;
  const HybQ cq{dyn_s + a.ef + 1, a.gc + (size_t)slot * a.gcap, a.ccap_lds};
  const int ccap = a.ccap_lds + a.gcap;
  uint32_t *vis = a.visited + (size_t)slot * a.vwords;
  uint32_t *vlog = a.vlog ? a.vlog + (size_t)slot * VLOG_CAP : nullptr;  // (uniform)
  int n_log = 0;

  float qv[CH][8];
  {
    const half8 *qrow = (const half8 *)(a.q + (size_t)qi * a.dpad);
    const int j = lane & 7;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const half8 v = qrow[j + 8 * c];
#pragma unroll
      for (int e = 0; e < 8; ++e) qv[c][e] = (float)v[e];
    }
  }
  unsigned long long n_dist = 0, n_exp = 0, n_adm = 0;

  // ---- bestEntryPointUntilLayer (HnswIndex.java:447-475) ----
  uint32_t cur = a.entry;
  if (a.max_level > 0) {
    if (lane == 0) ul[0] = cur;
    __syncthreads();
    wave_distances<CH>(a, qv, ul, ud, 1, lane);
    __syncthreads();
    float cur_dist = uni(ud[0]);
    n_dist += 1;
    for (int level = a.max_level; level > 0; --level) {
      bool changed = true;
      while (changed) {
        changed = false;
        int cnt = 0;
        const int us = a.upper_slot[cur];
        if (us >= 0) {
          const int rows = a.upper_base[us + 1] - a.upper_base[us];
          if (level <= rows) {
            const uint32_t *row = a.upper_adj + (size_t)(a.upper_base[us] + level - 1) * (a.m + 1);
            cnt = (int)row[0];
            if (lane < cnt) ul[lane] = row[1 + lane];
          }
        }
        __syncthreads();
        if (cnt > 0) {
          wave_distances<CH>(a, qv, ul, ud, cnt, lane);
          __syncthreads();
          n_dist += cnt;
          // `for nn in list: if (d < curDist) take it`: the running strict minimum, in list order
          for (int i = 0; i < cnt; ++i) {
            const float t = uni(ud[i]);
            if (t < cur_dist) {
              cur_dist = t;
              cur = uni(ul[i]);
              changed = true;
            }
          }
        }
        __syncthreads();
      }
    }
  }

  // ---- searchLayerForCandidates(query, entryPoint, max(ef, k), 0) (HnswIndex.java:571-623) ----
  const int ef = a.ef;
  if (lane == 0) ul[0] = cur;
  __syncthreads();
  wave_distances<CH>(a, qv, ul, ud, 1, lane);
  __syncthreads();
  n_dist += 1;
  int cn = 0, wn = 0, peak = 1;
  bool overflow = false;
  {
    const KEntry e0{dist_key(uni(ud[0])), cur};
    hq_add<true>(cq, cn, e0);
    wq[0] = e0;  // (an offer into an empty queue)
    wn = 1;
    if (lane == 0) {
      atomicOr(&vis[cur >> 5], 1u << (cur & 31));
      if (vlog) vlog[0] = cur;
    }
    n_log = 1;
  }
  float lower = key_dist(uni(wq[0].key));
  __syncthreads();
  while (cn > 0) {
    const KEntry cand = cq.get(0);
    if (key_dist(cand.key) > lower) break;
    (void)hq_poll<true>(cq, cn);
    n_exp += 1;
    const uint32_t *row = a.adj0 + (size_t)cand.node * (a.m0 + 1);
    const int cnt = (int)uni(row[0]);
    uint32_t nn = 0;
    bool fresh = false;
    if (lane < cnt) {
      nn = row[1 + lane];
      const uint32_t bit = 1u << (nn & 31);
      fresh = (atomicOr(&vis[nn >> 5], bit) & bit) == 0;  // visited.contains / visited.add
    }
    const unsigned long long mask = __ballot(fresh);
    const int nu = __popcll(mask);
    if (fresh) {
      const int at = __popcll(mask & ((1ull << lane) - 1));
      ul[at] = nn;  // list order is kept
      if (vlog && n_log + at < VLOG_CAP) vlog[n_log + at] = nn;
    }
    n_log += nu;
    __syncthreads();
    if (nu > 0) {
      wave_distances<CH>(a, qv, ul, ud, nu, lane);
      __syncthreads();
      n_dist += nu;
      // `lower` never rises once the result queue is full, so a neighbour that fails `d < lower` now fails it when its turn
      // comes: one ballot drops those, and the admission loop -- list order kept -- only visits the rest
      unsigned long long live = __ballot(lane < nu && (wn < ef || ud[lane < nu ? lane : 0] < lower));
      while (live) {
        const int i = __builtin_ctzll(live);
        live &= live - 1;
        const float d = uni(ud[i]);
        if (wn < ef || d < lower) {  // (lower is the result queue's head at all times)
          n_adm += 1;
          if (cn >= ccap) {
            overflow = true;
            break;
          }
          const KEntry e{dist_key(d), uni(ul[i])};
          hq_add<true>(cq, cn, e);
          peak = cn > peak ? cn : peak;
          kq_add<false>(wq, wn, e);
          if (wn > ef) (void)kq_poll<false>(wq, wn);
          lower = key_dist(uni(wq[0].key));
        }
      }
    }
    __syncthreads();
    if (overflow) break;
  }
  if (lane == 0) {
    atomicAdd(&a.stats[0], n_dist);
    atomicAdd(&a.stats[1], n_exp);
    atomicAdd(&a.stats[3], n_adm);
    atomicMax(&a.stats[2], (unsigned long long)(overflow ? ccap + 1 : peak));  // largest candidate queue of the launch
  }
  if (vlog) visited_undo(vis, a.vwords, vlog, n_log, lane);
  if (overflow) {
    if (lane == 0) a.spill[qi] = 1;
    return;
  }
  // dequeueAll (descending), reverse, first k (HnswIndex.java:546-549)
  const int found = wn;
  const int m = found < a.k ? found : a.k;
  for (int pos = found - 1; pos >= 0; --pos) {
    const KEntry e = kq_poll<false>(wq, wn);
    if (pos < m && lane == 0) {
      a.out_dist[(size_t)qi * a.k + pos] = key_dist(e.key);
      a.out_ids[(size_t)qi * a.k + pos] = a.ids ? a.ids[e.node] : (int64_t)e.node;
    }
  }
  if (lane == 0) {
    a.out_counts[qi] = m;
    a.spill[qi] = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// Batched index construction on the device (hnsw_index_build_insert_gpu): every step on the GPU, and deterministic.
//
// The reference inserts one item at a time, and with several writer threads it accepts what that costs: "when using
// concurrent writers we can miss connections that we would otherwise get" (HnswIndex.java:376-380).  The device builder is
// that mode taken wide, with the interleaving FIXED so that two builds of one input give one graph (and so that
// oracle/hnsw_oracle.c can restate it: oracle_hnsw_build_batched, compared entry for entry in tests/test_hnsw_gpu_build_gpu.py):
//   order    items by (level descending, position ascending); the first is the entry point and fixes maxLevel for good
//   rounds   round r inserts the next min(batch, max(1, linked / 8)) items of the order against ONE snapshot of the graph
//   phase A  one wave per item (hnsw_build_insert_kernel): wireConnectionForAllLayers (:137-148) as the reference writes it --
//            bestEntryPointUntilLayer, then per layer searchLayerForCandidates with beam efConstruction,
//            selectNearestNeighboursByHeuristic(candidates, maxM), setConnectionList of the NEW item, neighbours.get(0)
//            as the next layer's entry -- but the back links are only recorded: keys (layer, neighbour, order index)
//   sort     the round's keys (hipcub radix sort): all additions to one (layer, node) become one run, in order-index order
//   phase B  one wave per run (hnsw_build_backlink_kernel): append while the list has room (:414-417), else ONE re-selection
//            by the heuristic over the old list followed by the additions, ascending by (Float.compare distance, position)
//            (:419-427 does it once per addition; a round does it once per node)
// No two waves ever write one list, nothing needs a lock or an atomic, nothing returns to the host between rounds.
// Items of one round do not see each other; the graph is therefore not the sequential one (nor is the reference's with two
// writers).  Two bounds the reference does not have, both counted (hnsw_index_build_stats): the candidate queue of a walk
// holds BUILD_CCAP entries -- when full, entries farther than the current bound (which can never be expanded: the walk
// stops at the first such entry, :589-591) are dropped and the heap is rebuilt in array order; a re-selection sees the first
// LINK_CAP entries of old list + additions.
// ---------------------------------------------------------------------------------------------
constexpr int BUILD_EF_MAX = 256;  // efConstruction
constexpr int BUILD_CCAP = 1024;   // candidate-queue entries of a construction walk
constexpr int LINK_CAP = 1024;     // old list + additions a re-selection can see

struct BuildArgs {
  const _Float16 *x;
  uint32_t *adj0;              // [n][m0 + 1], updated in place
  const int32_t *upper_slot;
  const int32_t *upper_base;
  uint32_t *upper_adj;         // [rows][m + 1], updated in place
  const uint32_t *order;       // [n] insertion order
  const int32_t *levels;       // [n]
  const int64_t *pair_off;     // [n + 1] by order index: first key slot of an item ((layers wired) * m slots each)
  uint64_t *keys;              // the round's keys: phase A writes (unsorted buffer), phase B reads (sorted buffer)
  uint32_t *visited;           // [round items][vwords]: clean between walks (visited_undo)
  uint32_t *vlog;              // [round items][VLOG_CAP]; NULL = small bitmaps: a walk wipes its own before it starts
  unsigned long long *bstats;  // [0] additions a re-selection did not see, [1] candidate-queue prunes, [2] candidates dropped after a prune
  int64_t vwords;              // a multiple of 256
  int32_t dpad, metric, m, m0, efc, max_level;
  int32_t ccap;                // candidate-queue entries a walk may hold: BUILD_CCAP (smaller only for tests)
  uint32_t entry, at, count;   // the round = order[at .. at + count)
  uint32_t n_keys;
};

__device__ __forceinline__ uint32_t *layer_row(const BuildArgs &a, int level, uint32_t node) {
  if (level == 0) return a.adj0 + (size_t)node * (a.m0 + 1);
  return a.upper_adj + (size_t)(a.upper_base[a.upper_slot[node]] + level - 1) * (a.m + 1);  // (a node listed on a layer has that layer)
}

// distance between stored rows ra and rb by the 8 lanes of a group (j = lane & 7), the walk's arithmetic with row ra as the
// query; every lane of the group returns the sum
template <int CH>
__device__ __forceinline__ float group_row_distance(const BuildArgs &a, uint32_t ra, uint32_t rb, int j) {
  const half8 *pa = (const half8 *)(a.x + (size_t)ra * a.dpad), *pb = (const half8 *)(a.x + (size_t)rb * a.dpad);
  float acc = 0.0f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const half8 va = pa[j + 8 * c], vb = pb[j + 8 * c];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float fa = (float)va[e], fb = (float)vb[e];
      if (a.metric == HNSW_METRIC_L2) {
        const float t = fa - fb;
        acc = acc + t * t;
      } else {
        acc = __builtin_fmaf(fa, fb, acc);  // (exact product of two fp16 values: the same bits as acc + fa * fb, see wave_distances)
      }
    }
  }
  acc = acc + __shfl_xor(acc, 1, 64);
  acc = acc + __shfl_xor(acc, 2, 64);
  acc = acc + __shfl_xor(acc, 4, 64);
  return finish_distance(a.metric, acc);
}

// `kept` (nk entries) vs candidate c at distance dc from the base: is some kept node closer to c than the base is (:508-519)
template <int CH>
__device__ __forceinline__ bool heuristic_drops(const BuildArgs &a, const uint32_t *kept, int nk, uint32_t c, float dc, int lane) {
  const int g = lane >> 3, j = lane & 7;
  bool drop = false;
  for (int k0 = 0; k0 < nk && !drop; k0 += 8) {
    const int k = k0 + g;
    bool closer = false;
    if (k < nk) closer = group_row_distance<CH>(a, kept[k], c, j) < dc;
    drop = __ballot(closer) != 0ull;
  }
  return drop;
}

template <int CH>
__global__ __launch_bounds__(64) void hnsw_build_insert_kernel(BuildArgs a) {
  __shared__ HEntry wq[BUILD_EF_MAX + 1];
  __shared__ HEntry cq[BUILD_CCAP];
  __shared__ uint32_t ul[64];
  __shared__ float ud[64];
  __shared__ uint32_t kept[MAX_M];
  const int lane = threadIdx.x;
  const uint32_t t = blockIdx.x;
  const uint32_t item = a.order[a.at + t];
  const int item_level = a.levels[item];
  uint32_t *vis = a.visited + (size_t)t * a.vwords;
  uint32_t *vlog = a.vlog ? a.vlog + (size_t)t * VLOG_CAP : nullptr;  // (uniform)
  unsigned long long n_prune = 0, n_dropped = 0;

  float qv[CH][8];  // the item's own stored row is the query (distFnIndex, item to item)
  {
    const half8 *qrow = (const half8 *)(a.x + (size_t)item * a.dpad);
    const int j = lane & 7;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const half8 v = qrow[j + 8 * c];
#pragma unroll
      for (int e = 0; e < 8; ++e) qv[c][e] = (float)v[e];
    }
  }
  // ---- bestEntryPointUntilLayer(entry, item, maxLayer, itemLevel) (:447-475), when itemLevel < maxLayer (:140-142) ----
  uint32_t cur = a.entry;
  if (item_level < a.max_level) {
    if (lane == 0) ul[0] = cur;
    __syncthreads();
    wave_distances<CH>(a, qv, ul, ud, 1, lane);
    __syncthreads();
    float cur_dist = ud[0];
    for (int level = a.max_level; level > item_level; --level) {
      bool changed = true;
      while (changed) {
        changed = false;
        const uint32_t *row = layer_row(a, level, cur);
        const int cnt = (int)row[0];
        if (lane < cnt) ul[lane] = row[1 + lane];
        __syncthreads();
        if (cnt > 0) {
          wave_distances<CH>(a, qv, ul, ud, cnt, lane);
          __syncthreads();
          for (int i = 0; i < cnt; ++i) {
            const float d = ud[i];
            if (d < cur_dist) {
              cur_dist = d;
              cur = ul[i];
              changed = true;
            }
          }
        }
        __syncthreads();
      }
    }
  }
  const int ef = a.efc;
  const int top = item_level < a.max_level ? item_level : a.max_level;
  uint64_t *keys = a.keys + (a.pair_off[a.at + t] - a.pair_off[a.at]);
  for (int level = top; level >= 0; --level) {
    // ---- searchLayerForCandidates(item, cur, efConstruction, level) (:571-623, isUpdate = false): a fresh visited set ----
    int n_log = 0;  // (with a log the bitmap is clean: the previous walk of this slot undid its marks)
    if (!vlog) {
      for (int64_t w = (int64_t)lane * 4; w < a.vwords; w += 256) *(uint4 *)(vis + w) = make_uint4(0u, 0u, 0u, 0u);
      __threadfence_block();
    }
    if (lane == 0) ul[0] = cur;
    __syncthreads();
    wave_distances<CH>(a, qv, ul, ud, 1, lane);
    __syncthreads();
    int cn = 0, wn = 0;
    {
      const HEntry e0{ud[0], cur};
      pq_add<true>(cq, cn, e0);
      pq_add<false>(wq, wn, e0);
      if (lane == 0) {
        atomicOr(&vis[cur >> 5], 1u << (cur & 31));
        if (vlog) vlog[0] = cur;
      }
      n_log = 1;
    }
    float lower = wq[0].dist;
    __syncthreads();
    while (cn > 0) {
      const HEntry cand = cq[0];
      if (cand.dist > lower) break;
      (void)pq_poll<true>(cq, cn);
      const uint32_t *row = layer_row(a, level, cand.node);
      const int cnt = (int)row[0];
      uint32_t nn = 0;
      bool fresh = false;
      if (lane < cnt) {
        nn = row[1 + lane];
        const uint32_t bit = 1u << (nn & 31);
        fresh = (atomicOr(&vis[nn >> 5], bit) & bit) == 0;  // visited.contains / visited.add
      }
      const unsigned long long mask = __ballot(fresh);
      const int nu = __popcll(mask);
      if (fresh) {
        const int at = __popcll(mask & ((1ull << lane) - 1));
        ul[at] = nn;  // list order is kept
        if (vlog && n_log + at < VLOG_CAP) vlog[n_log + at] = nn;
      }
      n_log += nu;
      __syncthreads();
      if (nu > 0) {
        wave_distances<CH>(a, qv, ul, ud, nu, lane);
        __syncthreads();
        for (int i = 0; i < nu; ++i) {
          const HEntry e{ud[i], ul[i]};
          if (wn < ef || e.dist < wq[0].dist) {
            if (cn >= a.ccap) {  // drop what can never be expanded, rebuild the heap in array order
              const int had = cn;
              cn = 0;
              for (int s = 0; s < had; ++s) {
                const HEntry x = cq[s];
                if (!(x.dist > lower)) pq_add<true>(cq, cn, x);
              }
              n_prune += 1;
            }
            if (cn < a.ccap) pq_add<true>(cq, cn, e);
            else n_dropped += 1;
            pq_add<false>(wq, wn, e);
            if (wn > ef) (void)pq_poll<false>(wq, wn);
            lower = wq[0].dist;
          }
        }
      }
      __syncthreads();
    }
    if (vlog) visited_undo(vis, a.vwords, vlog, n_log, lane);
    // ---- selectNearestNeighboursByHeuristic(candidates, maxM) (:479-526); the item itself is never among them ----
    int nk = 0;
    if (wn <= a.m) {  // (:488-491) toListWithItem: the queue's ARRAY order
      if (lane < wn) kept[lane] = wq[lane].node;
      nk = wn;
      __syncthreads();
    } else {
      cn = 0;  // candidates.reverse() (:495): re-offered in array order under the reversed comparator
      for (int i = 0; i < wn; ++i) pq_add<true>(cq, cn, wq[i]);
      __syncthreads();
      while (cn > 0 && nk < a.m) {
        const HEntry c = pq_poll<true>(cq, cn);
        if (!heuristic_drops<CH>(a, kept, nk, c.node, c.dist, lane)) {
          if (lane == 0) kept[nk] = c.node;
          nk++;
          __syncthreads();
        }
      }
    }
    // ---- setConnectionList(item, level, neighbours) (:393); the back links become keys ----
    uint32_t *row = layer_row(a, level, item);
    if (lane < nk) row[1 + lane] = kept[lane];
    if (lane == 0) row[0] = (uint32_t)nk;
    uint64_t *kslot = keys + (size_t)(top - level) * a.m;
    if (lane < a.m)
      kslot[lane] = lane < nk ? ((uint64_t)level << 56) | ((uint64_t)kept[lane] << 24) | (uint64_t)t : ~0ull;
    cur = kept[0];  // neighbours.get(0) (:439)
    __syncthreads();
  }
  if (lane == 0 && (n_prune | n_dropped)) {
    atomicAdd(&a.bstats[1], n_prune);
    atomicAdd(&a.bstats[2], n_dropped);
  }
}

template <int CH>
__global__ __launch_bounds__(64) void hnsw_build_backlink_kernel(BuildArgs a) {
  __shared__ uint32_t t_id[LINK_CAP];
  __shared__ float t_d[LINK_CAP];
  __shared__ uint32_t c_id[LINK_CAP];
  __shared__ float c_d[LINK_CAP];
  __shared__ uint32_t kept[2 * MAX_M];
  const int lane = threadIdx.x, g = lane >> 3, j = lane & 7;
  const uint32_t i0 = blockIdx.x;
  const uint64_t key = a.keys[i0];
  if (key == ~0ull) return;
  const uint64_t run = key >> 24;  // (layer, node)
  if (i0 > 0 && (a.keys[i0 - 1] >> 24) == run) return;  // not the head of its run
  const int level = (int)(key >> 56);
  const uint32_t base = (uint32_t)(run & 0xffffffffull);
  int add_n = 0;
  for (uint32_t b = i0;; b += 64) {
    const uint32_t idx = b + lane;
    const bool same = idx < a.n_keys && (a.keys[idx] >> 24) == run;
    const unsigned long long mask = __ballot(same);
    if (mask == ~0ull) {
      add_n += 64;
      continue;
    }
    add_n += __builtin_ctzll(~mask);
    break;
  }
  const int M = level == 0 ? a.m0 : a.m;
  uint32_t *row = layer_row(a, level, base);
  const int old_n = (int)row[0];
  if (old_n + add_n <= M) {  // room: append, in order-index order (:414-417)
    for (int i = lane; i < add_n; i += 64) row[1 + old_n + i] = a.order[a.at + (uint32_t)(a.keys[i0 + i] & 0xffffffull)];
    __syncthreads();
    if (lane == 0) row[0] = (uint32_t)(old_n + add_n);
    return;
  }
  int n = old_n + add_n;
  if (n > LINK_CAP) {
    if (lane == 0) atomicAdd(&a.bstats[0], (unsigned long long)(n - LINK_CAP));
    n = LINK_CAP;
  }
  for (int i = lane; i < n; i += 64) t_id[i] = i < old_n ? row[1 + i] : a.order[a.at + (uint32_t)(a.keys[i0 + (i - old_n)] & 0xffffffull)];
  __syncthreads();
  for (int r0 = 0; r0 < n; r0 += 8) {  // distances to the base, eight candidates per round
    const int i = r0 + g;
    float d = 0.0f;
    if (i < n) d = group_row_distance<CH>(a, base, t_id[i], j);
    if (i < n && j == 0) t_d[i] = d;
  }
  __syncthreads();
  for (int i = lane; i < n; i += 64) {  // ascending by (Float.compare distance, position): every lane ranks its candidates by counting
    const float d = t_d[i];
    int rank = 0;
    for (int e = 0; e < n; ++e) {
      const int c = float_compare(t_d[e], d);
      rank += (c < 0 || (c == 0 && e < i)) ? 1 : 0;
    }
    c_id[rank] = t_id[i];
    c_d[rank] = d;
  }
  __syncthreads();
  int nk = 0;  // n > M here: the heuristic proper (:495-523)
  for (int i = 0; i < n && nk < M; ++i) {
    const uint32_t c = c_id[i];
    if (c == base) continue;
    if (!heuristic_drops<CH>(a, kept, nk, c, c_d[i], lane)) {
      if (lane == 0) kept[nk] = c;
      nk++;
      __syncthreads();
    }
  }
  __syncthreads();
  for (int i = lane; i < nk; i += 64) row[1 + i] = kept[i];
  if (lane == 0) row[0] = (uint32_t)nk;
}

template <int CH>
void launch_build_round(const BuildArgs &a_ins, const BuildArgs &a_link, hipStream_t st, int which) {
  if (which == 0) hipLaunchKernelGGL((hnsw_build_insert_kernel<CH>), dim3(a_ins.count), dim3(64), 0, st, a_ins);
  else hipLaunchKernelGGL((hnsw_build_backlink_kernel<CH>), dim3(a_link.n_keys), dim3(64), 0, st, a_link);
}
void launch_build_any(int chunks, const BuildArgs &a_ins, const BuildArgs &a_link, hipStream_t st, int which) {
  switch (chunks) {
    case 1: return launch_build_round<1>(a_ins, a_link, st, which);
    case 2: return launch_build_round<2>(a_ins, a_link, st, which);
    case 3: return launch_build_round<3>(a_ins, a_link, st, which);
    case 4: return launch_build_round<4>(a_ins, a_link, st, which);
    case 5: return launch_build_round<5>(a_ins, a_link, st, which);
    case 6: return launch_build_round<6>(a_ins, a_link, st, which);
    case 7: return launch_build_round<7>(a_ins, a_link, st, which);
    default: return launch_build_round<8>(a_ins, a_link, st, which);
  }
}

// rows (fp32) -> fp16 rows padded to dpad; Cosine rows are normalised first (Hnsw.scala:149-155)
__global__ void hnsw_prep_rows(const float *__restrict__ src, int64_t n, int d, int dpad, int normalise,
                               _Float16 *__restrict__ dst) {
  int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= n) return;
  const float *x = src + row * d;
  float norm = 1.0f;
  if (normalise) {
    double ss = 0;
    for (int k = lane; k < d; k += 64) ss += (double)x[k] * (double)x[k];
    for (int o = 32; o; o >>= 1) ss += __shfl_xor(ss, o, 64);
    norm = (float)sqrt(ss);
    if (!(norm > 0.0f)) norm = 1.0f;
  }
  for (int k = lane; k < dpad; k += 64) dst[row * dpad + k] = (_Float16)(k < d ? x[k] / norm : 0.0f);
}

__global__ void hnsw_rows_to_f32(const _Float16 *__restrict__ src, int64_t i0, int64_t n, int d, int dpad,
                                 float *__restrict__ dst) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * d) return;
  dst[e] = (float)src[(i0 + e / d) * dpad + e % d];
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host: the index, the reference's insertion algorithm, the C ABI
// ---------------------------------------------------------------------------------------------
struct hnsw_index {
  int device = 0, metric = 0, d = 0, dpad = 0, m = 0, m0 = 0, max_level = 0;
  int64_t n = 0, entry = -1;
  std::vector<std::vector<std::vector<uint32_t>>> upper;  // [level - 1][slot] lists, host copy for export
  std::vector<std::vector<uint32_t>> level0;              // host copy
  std::vector<uint8_t> has0;                              // node has a level-0 entry
  std::vector<int32_t> upper_slot_h, upper_base_h;
  std::vector<std::vector<uint8_t>> has_upper;            // [slot][level - 1]: entry exists
  Buf x, adj0, upper_slot, upper_base, upper_adj, ids;
  bool has_ids = false;
  // scratch
  Buf q_in, q, visited, vlog, gc, o_dist, o_ids, o_cnt, spill, stats, qlist;
  bool visited_dirty = true;  // the bitmaps must be wiped before the next search (fresh buffer, or a search that did not finish)
  hipEvent_t ev[2] = {nullptr, nullptr};
  int64_t last_dist = 0, last_exp = 0;
  int32_t last_spilled = 0;
  int64_t last_peak = 0, last_adm = 0;
  float last_ms = 0;
  int64_t build_rounds = 0, build_truncated = 0, build_prunes = 0, build_dropped = 0;  // hnsw_index_build_insert_gpu
  ~hnsw_index() {
    for (auto &e : ev)
      if (e) (void)hipEventDestroy(e);
  }
};

namespace {

// graph under construction / being loaded.  Dense rows: item i owns rows base[i] .. base[i] + top[i] (levels
// 0..top[i]); `has` says whether the reference's map holds the key HnswNode(level, item) at all.
// One lock per stripe of items guards a row's read-modify-write, so the builder can run on several
// host threads (the reference inserts concurrently too, HnswIndex.java:150-200); at most one lock is
// ever held, so there is nothing to deadlock on.
struct HostGraph {
  int64_t n = 0;
  int m = 0, m0 = 0;
  int max_level = -1;  // HnswIndex.java:97
  int64_t entry = -1;
  std::vector<int32_t> top;
  std::vector<int64_t> base;
  std::vector<std::vector<uint32_t>> rows;
  std::vector<uint8_t> has;
  std::unique_ptr<std::mutex[]> locks;
  static constexpr size_t STRIPES = 8192;
  std::mutex meta;

  void init(int64_t n_, int m_, const std::vector<int32_t> &tops) {
    n = n_;
    m = m_;
    m0 = 2 * m_;
    top = tops;
    base.assign((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; ++i) base[(size_t)i + 1] = base[(size_t)i] + top[(size_t)i] + 1;
    rows.assign((size_t)base[(size_t)n], {});
    has.assign((size_t)base[(size_t)n], 0);
    locks.reset(new std::mutex[STRIPES]);
  }
  std::mutex &lock_of(uint32_t item) { return locks[item % STRIPES]; }
  // getConnectionListForRead: a copy of the list, empty when the key is absent
  std::vector<uint32_t> read(int level, uint32_t item) {
    if (level > top[item]) return {};
    std::lock_guard<std::mutex> lk(lock_of(item));
    return rows[(size_t)(base[item] + level)];
  }
  bool exists(int level, uint32_t item) const { return level <= top[item] && has[(size_t)(base[item] + level)]; }
  void put_locked(int level, uint32_t item, std::vector<uint32_t> list) {  // caller holds lock_of(item)
    rows[(size_t)(base[item] + level)] = std::move(list);
    has[(size_t)(base[item] + level)] = 1;
  }
  void put(int level, uint32_t item, std::vector<uint32_t> list) {
    std::lock_guard<std::mutex> lk(lock_of(item));
    put_locked(level, item, std::move(list));
  }
};

// stored rows on the host, fp16-rounded, each 64-element block transposed ([e][j] instead of [j][e]) so that
// the 8 partial sums of the fixed summation order are the lanes of one SIMD accumulator
struct HostVectors {
  std::vector<float> t;
  int dpad = 0, metric = 0;
  void load(const std::vector<float> &rows, int64_t n, int dpad_, int metric_) {
    dpad = dpad_;
    metric = metric_;
    t.resize(rows.size());
    for (int64_t i = 0; i < n; ++i)
      for (int b = 0; b < dpad / 64; ++b)
        for (int j = 0; j < 8; ++j)
          for (int e = 0; e < 8; ++e) t[(size_t)i * dpad + b * 64 + e * 8 + j] = rows[(size_t)i * dpad + b * 64 + j * 8 + e];
  }
  float distance(uint32_t a, uint32_t b) const {
    const float *q = t.data() + (size_t)a * dpad, *y = t.data() + (size_t)b * dpad;
    float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (metric == HNSW_METRIC_L2) {
      for (int k = 0; k < dpad; k += 8)
        for (int j = 0; j < 8; ++j) {
          const float d = q[k + j] - y[k + j];
          p[j] = p[j] + d * d;
        }
    } else {
      for (int k = 0; k < dpad; k += 8)
        for (int j = 0; j < 8; ++j) p[j] = p[j] + q[k + j] * y[k + j];
    }
    const float r0 = p[0] + p[1], r2 = p[2] + p[3], r4 = p[4] + p[5], r6 = p[6] + p[7];
    return finish_distance(metric, (r0 + r2) + (r4 + r6));
  }
};

// DistancedItemQueue on the host
struct JQueue {
  std::vector<HEntry> q;
  int n = 0;
  bool minq;
  uint32_t origin;
  explicit JQueue(bool mq, uint32_t o) : minq(mq), origin(o) {}
  void add(HEntry e) {
    if ((int)q.size() <= n) q.resize((size_t)n + 64);
    if (minq) pq_add<true>(q.data(), n, e);
    else pq_add<false>(q.data(), n, e);
  }
  HEntry poll() { return minq ? pq_poll<true>(q.data(), n) : pq_poll<false>(q.data(), n); }
  JQueue reverse() const {  // DistancedItemQueue.reverse: re-add in array (iterator) order
    JQueue r(!minq, origin);
    for (int i = 0; i < n; ++i) r.add(q[(size_t)i]);
    return r;
  }
};

struct Builder {  // one per host thread
  HostGraph &g;
  const HostVectors &v;
  int ef_construction;
  std::vector<uint32_t> stamp;  // visited set, epoch-stamped
  uint32_t epoch = 0;

  uint32_t best_entry(uint32_t entry, uint32_t item, int max_layer, int selected) {  // HnswIndex.java:447-475
    uint32_t cur = entry;
    if (selected < max_layer) {
      float cur_dist = v.distance(item, cur);
      for (int level = max_layer; level > selected; --level) {
        bool changed = true;
        while (changed) {
          changed = false;
          for (uint32_t nn : g.read(level, cur)) {
            const float t = v.distance(item, nn);
            if (t < cur_dist) {
              cur_dist = t;
              cur = nn;
              changed = true;
            }
          }
        }
      }
    }
    return cur;
  }

  JQueue search_layer(uint32_t item, uint32_t entry, int ef, int level) {  // HnswIndex.java:571-623, isUpdate = false
    JQueue cq(true, item);
    cq.add(HEntry{v.distance(item, entry), entry});
    JQueue wq = cq.reverse();
    ++epoch;
    stamp[entry] = epoch;
    float lower = wq.q[0].dist;
    while (cq.n > 0) {
      const HEntry cand = cq.q[0];
      if (cand.dist > lower) break;
      cq.poll();
      for (uint32_t nn : g.read(level, cand.node)) {
        if (stamp[nn] == epoch) continue;
        stamp[nn] = epoch;
        const float dist = v.distance(item, nn);
        if (wq.n < ef || dist < wq.q[0].dist) {
          cq.add(HEntry{dist, nn});
          wq.add(HEntry{dist, nn});
          if (wq.n > ef) wq.poll();
          lower = wq.q[0].dist;
        }
      }
    }
    return wq;
  }

  std::vector<uint32_t> select(const JQueue &cands, int max_conn) {  // HnswIndex.java:479-526
    const uint32_t base = cands.origin;
    std::vector<uint32_t> res;
    if (cands.n <= max_conn) {
      bool removed = false;  // List.remove(Object): first occurrence only
      for (int i = 0; i < cands.n; ++i) {
        if (!removed && cands.q[(size_t)i].node == base) {
          removed = true;
          continue;
        }
        res.push_back(cands.q[(size_t)i].node);
      }
      return res;
    }
    JQueue minq = cands.reverse();
    while (minq.n > 0) {
      if ((int)res.size() >= max_conn) break;
      const HEntry c = minq.poll();
      if (c.node == base) continue;
      bool include = true;
      for (uint32_t e : res) {
        if (v.distance(e, c.node) < c.dist) {
          include = false;
          break;
        }
      }
      if (include) res.push_back(c.node);
    }
    return res;
  }

  uint32_t connect(uint32_t item, const JQueue &cands, int level) {  // mutuallyConnectNewElement, isUpdate = false
    const std::vector<uint32_t> neighbours = select(cands, g.m);
    g.put(level, item, neighbours);
    const int M = level == 0 ? g.m0 : g.m;
    for (uint32_t nn : neighbours) {
      if (nn == item) continue;
      std::lock_guard<std::mutex> lk(g.lock_of(nn));  // the neighbour's write lock (:406-435)
      std::vector<uint32_t> conn = g.rows[(size_t)(g.base[nn] + level)];
      if ((int)conn.size() < M) {
        conn.push_back(item);
      } else {
        JQueue q(false, nn);
        for (uint32_t t : conn) q.add(HEntry{v.distance(nn, t), t});
        q.add(HEntry{v.distance(nn, item), item});
        conn = select(q, M);
      }
      g.put_locked(level, nn, std::move(conn));
    }
    return neighbours.empty() ? item : neighbours[0];
  }

  void insert(uint32_t item, int cur_level) {  // HnswIndex.java:137-200
    int64_t entry;
    int max_layer;
    bool hold_meta = false;
    g.meta.lock();
    entry = g.entry;
    max_layer = g.max_level;
    if (cur_level > max_layer) hold_meta = true;  // the global lock: this insert may move the entry point (:173-183)
    else g.meta.unlock();
    if (entry >= 0) {
      uint32_t cur = (uint32_t)entry;
      if (cur_level < max_layer) cur = best_entry(cur, item, max_layer, cur_level);
      for (int level = std::min(cur_level, max_layer); level >= 0; --level) {
        const JQueue cands = search_layer(item, cur, ef_construction, level);
        cur = connect(item, cands, level);
      }
    }
    if (hold_meta) {  // HnswMeta starts at (-1, empty): the first item always takes this branch
      g.max_level = cur_level;
      g.entry = item;
      g.meta.unlock();
    }
  }
};

int upload_graph(hnsw_index *ix, const HostGraph &g) {
  const int64_t n = ix->n;
  ix->entry = g.entry;
  ix->max_level = std::max(g.max_level, 0);
  ix->level0.assign((size_t)n, {});
  ix->has0.assign((size_t)n, 0);
  std::vector<uint32_t> adj0((size_t)std::max<int64_t>(n, 1) * (ix->m0 + 1), 0);
  for (int64_t i = 0; i < n; ++i) {
    const auto &l = g.rows[(size_t)g.base[(size_t)i]];
    ix->level0[(size_t)i] = l;
    ix->has0[(size_t)i] = g.has[(size_t)g.base[(size_t)i]];
    adj0[(size_t)i * (ix->m0 + 1)] = (uint32_t)l.size();
    std::copy(l.begin(), l.end(), adj0.begin() + (size_t)i * (ix->m0 + 1) + 1);
  }
  // upper levels: slot per node with storage above level 0, rows = its top level
  ix->upper_slot_h.assign((size_t)std::max<int64_t>(n, 1), -1);
  ix->upper_base_h.assign(1, 0);
  int top_all = 0;
  for (int64_t i = 0; i < n; ++i)
    if (g.top[(size_t)i] > 0) {
      ix->upper_slot_h[(size_t)i] = (int32_t)ix->upper_base_h.size() - 1;
      ix->upper_base_h.push_back(ix->upper_base_h.back() + g.top[(size_t)i]);
      top_all = std::max(top_all, g.top[(size_t)i]);
    }
  const int64_t rows = ix->upper_base_h.back();
  std::vector<uint32_t> uadj((size_t)std::max<int64_t>(rows, 1) * (ix->m + 1), 0);
  const size_t n_slots = ix->upper_base_h.size() - 1;
  ix->upper.assign((size_t)top_all, std::vector<std::vector<uint32_t>>(n_slots));
  ix->has_upper.assign(n_slots, {});
  for (int64_t i = 0; i < n; ++i) {
    const int32_t sl = ix->upper_slot_h[(size_t)i];
    if (sl < 0) continue;
    ix->has_upper[(size_t)sl].assign((size_t)g.top[(size_t)i], 0);
    for (int l = 1; l <= g.top[(size_t)i]; ++l) {
      const auto &list = g.rows[(size_t)(g.base[(size_t)i] + l)];
      ix->upper[(size_t)l - 1][(size_t)sl] = list;
      ix->has_upper[(size_t)sl][(size_t)l - 1] = g.has[(size_t)(g.base[(size_t)i] + l)];
      uint32_t *row = uadj.data() + (size_t)(ix->upper_base_h[(size_t)sl] + l - 1) * (ix->m + 1);
      row[0] = (uint32_t)list.size();
      std::copy(list.begin(), list.end(), row + 1);
    }
  }
  HTRY(ix->adj0.reserve(adj0.size() * 4));
  HTRY(hipMemcpy(ix->adj0.p, adj0.data(), adj0.size() * 4, hipMemcpyHostToDevice));
  HTRY(ix->upper_slot.reserve(ix->upper_slot_h.size() * 4));
  HTRY(hipMemcpy(ix->upper_slot.p, ix->upper_slot_h.data(), ix->upper_slot_h.size() * 4, hipMemcpyHostToDevice));
  HTRY(ix->upper_base.reserve(ix->upper_base_h.size() * 4));
  HTRY(hipMemcpy(ix->upper_base.p, ix->upper_base_h.data(), ix->upper_base_h.size() * 4, hipMemcpyHostToDevice));
  HTRY(ix->upper_adj.reserve(uadj.size() * 4));
  HTRY(hipMemcpy(ix->upper_adj.p, uadj.data(), uadj.size() * 4, hipMemcpyHostToDevice));
  return HNSW_OK;
}

int create_index(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids, int32_t max_m,
                 std::unique_ptr<hnsw_index> &ix, std::vector<float> *host_rows) {
  if (metric < HNSW_METRIC_L2 || metric > HNSW_METRIC_INNER_PRODUCT) return fail(HNSW_EINVAL, "unknown metric");
  if (n < 0 || n >= (int64_t)0x7fffffff) return fail(HNSW_EINVAL, "vector count out of range");
  if (d < 1 || d > MAX_D) return fail(HNSW_EINVAL, "dimension must be in 1..512");
  if (max_m < 2 || max_m > MAX_M) return fail(HNSW_EINVAL, "max_m must be in 2..32");
  if (n > 0 && !vectors) return fail(HNSW_EINVAL, "NULL vectors");
  ix.reset(new hnsw_index);
  ix->device = device;
  ix->metric = metric;
  ix->n = n;
  ix->d = d;
  ix->dpad = (d + 63) / 64 * 64;
  ix->m = max_m;
  ix->m0 = 2 * max_m;
  HTRY(hipSetDevice(device));
  for (auto &e : ix->ev) HTRY(hipEventCreate(&e));
  HTRY(ix->x.reserve((size_t)std::max<int64_t>(n, 1) * ix->dpad * sizeof(_Float16)));
  if (n > 0) {
    Buf stage;
    const int64_t chunk = std::max<int64_t>(1, (int64_t)(128u << 20) / ((int64_t)d * 4));
    HTRY(stage.reserve((size_t)std::min(chunk, n) * d * 4));
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
      const int64_t m = std::min(chunk, n - r0);
      HTRY(hipMemcpy(stage.p, vectors + r0 * d, (size_t)m * d * 4, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(hnsw_prep_rows, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, 0, stage.as<float>(), m, d, ix->dpad,
                         metric == HNSW_METRIC_COSINE ? 1 : 0, ix->x.as<_Float16>() + r0 * ix->dpad);
      HTRY(hipGetLastError());
      HTRY(hipDeviceSynchronize());
    }
  }
  if (ids && n > 0) {
    HTRY(ix->ids.reserve((size_t)n * 8));
    HTRY(hipMemcpy(ix->ids.p, ids, (size_t)n * 8, hipMemcpyHostToDevice));
    ix->has_ids = true;
  }
  if (host_rows) {  // the stored rows, as floats, for the host-side builder
    std::vector<_Float16> h((size_t)n * ix->dpad);
    if (n > 0) HTRY(hipMemcpy(h.data(), ix->x.p, h.size() * sizeof(_Float16), hipMemcpyDeviceToHost));
    host_rows->resize(h.size());
    for (size_t i = 0; i < h.size(); ++i) (*host_rows)[i] = (float)h[i];
  }
  return HNSW_OK;
}

template <int CH>
int launch_search(int blocks, const SearchArgs &a, hipStream_t st) {
  const size_t lds = (size_t)(a.ef + 1 + a.ccap_lds) * sizeof(HEntry);
  hipLaunchKernelGGL((hnsw_search_kernel<CH>), dim3(blocks), dim3(64), lds, st, a);
  return HNSW_OK;
}
int launch_search_any(int chunks, int blocks, const SearchArgs &a, hipStream_t st) {
  switch (chunks) {
    case 1: return launch_search<1>(blocks, a, st);
    case 2: return launch_search<2>(blocks, a, st);
    case 3: return launch_search<3>(blocks, a, st);
    case 4: return launch_search<4>(blocks, a, st);
    case 5: return launch_search<5>(blocks, a, st);
    case 6: return launch_search<6>(blocks, a, st);
    case 7: return launch_search<7>(blocks, a, st);
    default: return launch_search<8>(blocks, a, st);
  }
}

}  // namespace

extern "C" {

const char *hnsw_last_error(void) { return g_err.c_str(); }

int hnsw_index_build(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids,
                     int32_t max_m, int64_t entry_point, int32_t max_level, int64_t n_entries, const int32_t *entry_level,
                     const int64_t *entry_item, const int64_t *entry_offsets, const int64_t *entry_neighbours,
                     hnsw_index_t **out) try {
  if (!out) return fail(HNSW_EINVAL, "out is NULL");
  if (n_entries < 0 || (n_entries > 0 && (!entry_level || !entry_item || !entry_offsets)))
    return fail(HNSW_EINVAL, "NULL graph arrays");
  if (entry_point >= n || max_level < 0 || max_level > 64) return fail(HNSW_EINVAL, "entry point / max level out of range");
  std::unique_ptr<hnsw_index> ix;
  int rc = create_index(device, metric, n, d, vectors, ids, max_m, ix, nullptr);
  if (rc) return rc;
  std::vector<int32_t> tops((size_t)n, 0);
  for (int64_t e = 0; e < n_entries; ++e) {
    if (entry_level[e] < 0 || entry_level[e] > max_level || entry_item[e] < 0 || entry_item[e] >= n)
      return fail(HNSW_EINVAL, "graph entry out of range");
    tops[(size_t)entry_item[e]] = std::max(tops[(size_t)entry_item[e]], entry_level[e]);
  }
  HostGraph g;
  g.init(n, max_m, tops);
  g.entry = entry_point < 0 ? -1 : entry_point;
  g.max_level = max_level;
  for (int64_t e = 0; e < n_entries; ++e) {
    const int level = entry_level[e];
    const int64_t item = entry_item[e];
    const int64_t b = entry_offsets[e], en = entry_offsets[e + 1];
    if (en < b || en - b > (level == 0 ? g.m0 : g.m)) return fail(HNSW_EINVAL, "neighbour list longer than the level allows");
    std::vector<uint32_t> list;
    for (int64_t j = b; j < en; ++j) {
      if (!entry_neighbours || entry_neighbours[j] < 0 || entry_neighbours[j] >= n) return fail(HNSW_EINVAL, "neighbour out of range");
      list.push_back((uint32_t)entry_neighbours[j]);
    }
    g.put(level, (uint32_t)item, std::move(list));
  }
  rc = upload_graph(ix.get(), g);
  if (rc) return rc;
  *out = ix.release();
  return HNSW_OK;
} ABI_CATCH

static int build_insert_impl(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids,
                             int32_t max_m, int32_t ef_construction, uint64_t seed, const int32_t *given_levels,
                             int32_t n_threads, hnsw_index_t **out);

int hnsw_index_build_insert(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids,
                            int32_t max_m, int32_t ef_construction, uint64_t seed, int32_t n_threads, hnsw_index_t **out) try {
  return build_insert_impl(device, metric, n, d, vectors, ids, max_m, ef_construction, seed, nullptr, n_threads, out);
} ABI_CATCH

int hnsw_index_build_insert_levels(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors,
                                   const int64_t *ids, int32_t max_m, int32_t ef_construction, const int32_t *levels,
                                   int32_t n_threads, hnsw_index_t **out) try {
  if (!levels && n > 0) return fail(HNSW_EINVAL, "levels is NULL");
  for (int64_t i = 0; i < n; ++i)
    if (levels[i] < 0 || levels[i] > 60) return fail(HNSW_EINVAL, "a level is outside 0..60");
  return build_insert_impl(device, metric, n, d, vectors, ids, max_m, ef_construction, 0, levels, n_threads, out);
} ABI_CATCH

static int build_insert_impl(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids,
                             int32_t max_m, int32_t ef_construction, uint64_t seed, const int32_t *given_levels,
                             int32_t n_threads, hnsw_index_t **out) {
  if (!out) return fail(HNSW_EINVAL, "out is NULL");
  if (ef_construction < 1) return fail(HNSW_EINVAL, "ef_construction must be positive");
  if (n_threads < 1 || n_threads > 256) return fail(HNSW_EINVAL, "n_threads must be in 1..256");
  std::unique_ptr<hnsw_index> ix;
  std::vector<float> rows;
  int rc = create_index(device, metric, n, d, vectors, ids, max_m, ix, &rows);
  if (rc) return rc;
  const double level_mult = 1.0 / std::log(1.0 * max_m);  // HnswIndex.java:118
  std::vector<int32_t> levels((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    if (given_levels) {
      levels[(size_t)i] = given_levels[i];
      continue;
    }
    const uint64_t h = sann::mix64(seed ^ ((uint64_t)i * 0x9E3779B97F4A7C15ull));
    const double u = ((double)(h >> 11) + 1.0) * (1.0 / 9007199254740992.0);  // (0, 1]
    levels[(size_t)i] = std::min(60, (int)(-std::log(u) * level_mult));       // getRandomLevel, :369-371
  }
  HostGraph g;
  g.init(n, max_m, levels);
  HostVectors hv;
  hv.load(rows, n, ix->dpad, metric);
  rows.clear();
  rows.shrink_to_fit();
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, std::max<int64_t>(1, n / 64)));
  if (nt == 1) {
    Builder b{g, hv, ef_construction, std::vector<uint32_t>((size_t)n, 0), 0};
    for (int64_t i = 0; i < n; ++i) b.insert((uint32_t)i, levels[(size_t)i]);
  } else {
    // the first items go in one by one (an entry point must exist), the rest concurrently
    const int64_t warm = std::min<int64_t>(n, 256);
    {
      Builder b{g, hv, ef_construction, std::vector<uint32_t>((size_t)n, 0), 0};
      for (int64_t i = 0; i < warm; ++i) b.insert((uint32_t)i, levels[(size_t)i]);
    }
    std::atomic<int64_t> next(warm);
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t)
      pool.emplace_back([&]() {
        Builder b{g, hv, ef_construction, std::vector<uint32_t>((size_t)n, 0), 0};
        for (;;) {
          const int64_t i = next.fetch_add(1);
          if (i >= n) break;
          b.insert((uint32_t)i, levels[(size_t)i]);
        }
      });
    for (auto &th : pool) th.join();
  }
  rc = upload_graph(ix.get(), g);
  if (rc) return rc;
  *out = ix.release();
  return HNSW_OK;
}

// Construction on the device: see the comment above hnsw_build_insert_kernel.  The host only works out the insertion
// order and the round schedule (both functions of the levels alone) and enqueues kernels; nothing comes back before the end.
static int build_insert_gpu_impl(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids,
                                 int32_t max_m, int32_t ef_construction, uint64_t seed, const int32_t *given_levels, int32_t batch,
                                 hnsw_index_t **out) {
  if (!out) return fail(HNSW_EINVAL, "out is NULL");
  if (ef_construction < 1 || ef_construction > BUILD_EF_MAX) return fail(HNSW_EINVAL, "ef_construction must be in 1..256");
  if (batch < 0 || batch > (1 << 20)) return fail(HNSW_EINVAL, "batch must be in 0..2^20");
  if (batch == 0) batch = 4096;
  std::unique_ptr<hnsw_index> ix;
  int rc = create_index(device, metric, n, d, vectors, nullptr, max_m, ix, nullptr);  // (positions as labels while building)
  if (rc) return rc;
  const double level_mult = 1.0 / std::log(1.0 * max_m);  // HnswIndex.java:118
  std::vector<int32_t> levels((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    if (given_levels) {
      if (given_levels[i] < 0 || given_levels[i] > 60) return fail(HNSW_EINVAL, "a level is outside 0..60");
      levels[(size_t)i] = given_levels[i];
      continue;
    }
    const uint64_t h = sann::mix64(seed ^ ((uint64_t)i * 0x9E3779B97F4A7C15ull));
    const double u = ((double)(h >> 11) + 1.0) * (1.0 / 9007199254740992.0);  // (0, 1]
    levels[(size_t)i] = std::min(60, (int)(-std::log(u) * level_mult));       // getRandomLevel, :369-371
  }
  // the empty graph with every row in place (rows are a function of the levels); the kernels fill it
  HostGraph g;
  g.init(n, max_m, levels);
  std::vector<uint32_t> order((size_t)n);
  for (int64_t i = 0; i < n; ++i) order[(size_t)i] = (uint32_t)i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return levels[x] > levels[y]; });
  if (n > 0) {
    g.entry = order[0];
    g.max_level = levels[order[0]];
  }
  rc = upload_graph(ix.get(), g);
  if (rc) return rc;
  ix->build_truncated = ix->build_prunes = ix->build_dropped = ix->build_rounds = 0;
  if (n > 1) {
    HTRY(hipSetDevice(device));
    const int max_level = ix->max_level;
    std::vector<int64_t> pair_off((size_t)n + 1, 0);
    for (int64_t e = 0; e < n; ++e) pair_off[(size_t)e + 1] = pair_off[(size_t)e] + (int64_t)(std::min(levels[order[(size_t)e]], max_level) + 1) * max_m;
    // the schedule: rounds of min(batch, max(1, linked / 8)) items
    std::vector<std::pair<int64_t, int64_t>> rounds;
    int64_t max_keys = 0, max_items = 0;
    for (int64_t at = 1, linked = 1; at < n;) {
      const int64_t m = std::min<int64_t>(n - at, std::min<int64_t>(batch, std::max<int64_t>(1, linked / 8)));
      rounds.emplace_back(at, m);
      max_keys = std::max(max_keys, pair_off[(size_t)(at + m)] - pair_off[(size_t)at]);
      max_items = std::max(max_items, m);
      at += m;
      linked += m;
    }
    if (max_keys >= (int64_t)1 << 31) return fail(HNSW_EINVAL, "batch * max_m too large");
    Buf d_order, d_levels, d_pair_off, d_keys, d_sorted, d_tmp, d_bstats, d_visited, d_vlog;
    HTRY(d_order.reserve((size_t)n * 4));
    HTRY(d_levels.reserve((size_t)n * 4));
    HTRY(d_pair_off.reserve(((size_t)n + 1) * 8));
    HTRY(hipMemcpy(d_order.p, order.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HTRY(hipMemcpy(d_levels.p, levels.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HTRY(hipMemcpy(d_pair_off.p, pair_off.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice));
    HTRY(d_keys.reserve((size_t)max_keys * 8));
    HTRY(d_sorted.reserve((size_t)max_keys * 8));
    HTRY(d_bstats.reserve(4 * 8));
    HTRY(hipMemset(d_bstats.p, 0, 4 * 8));
    const int64_t vwords = ((n + 31) / 32 + 255) / 256 * 256;
    HTRY(d_visited.reserve((size_t)max_items * vwords * 4));
    const bool use_vlog = vwords >= VLOG_MIN_VWORDS || getenv("HNSW_DEBUG_VLOG") != nullptr;  // (the variable: tests force the log on small graphs)
    if (use_vlog) {
      HTRY(hipMemset(d_visited.p, 0, (size_t)max_items * vwords * 4));  // once: every walk cleans up after itself
      HTRY(d_vlog.reserve((size_t)max_items * VLOG_CAP * 4));
    }
    size_t tmp_bytes = 0;
    HTRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, d_keys.as<uint64_t>(), d_sorted.as<uint64_t>(), (int)max_keys, 0, 64, (hipStream_t)0));
    HTRY(d_tmp.reserve(tmp_bytes));
    BuildArgs a;
    a.x = ix->x.as<_Float16>();
    a.adj0 = ix->adj0.as<uint32_t>();
    a.upper_slot = ix->upper_slot.as<int32_t>();
    a.upper_base = ix->upper_base.as<int32_t>();
    a.upper_adj = ix->upper_adj.as<uint32_t>();
    a.order = d_order.as<uint32_t>();
    a.levels = d_levels.as<int32_t>();
    a.pair_off = d_pair_off.as<int64_t>();
    a.visited = d_visited.as<uint32_t>();
    a.vlog = use_vlog ? d_vlog.as<uint32_t>() : nullptr;
    a.bstats = d_bstats.as<unsigned long long>();
    a.vwords = vwords;
    a.dpad = ix->dpad;
    a.metric = ix->metric;
    a.m = ix->m;
    a.m0 = ix->m0;
    a.efc = ef_construction;
    a.max_level = max_level;
    a.ccap = BUILD_CCAP;
    if (const char *e = std::getenv("HNSW_BUILD_CCAP")) a.ccap = std::max(2, std::min(BUILD_CCAP, std::atoi(e)));  // (tests: make the prune path run)
    a.entry = (uint32_t)ix->entry;
    const int chunks = ix->dpad / 64;
    for (const auto &r : rounds) {
      a.at = (uint32_t)r.first;
      a.count = (uint32_t)r.second;
      a.n_keys = (uint32_t)(pair_off[(size_t)(r.first + r.second)] - pair_off[(size_t)r.first]);
      BuildArgs a_ins = a, a_link = a;
      a_ins.keys = d_keys.as<uint64_t>();
      a_link.keys = d_sorted.as<uint64_t>();
      launch_build_any(chunks, a_ins, a_link, 0, 0);
      size_t tb = tmp_bytes;
      HTRY(hipcub::DeviceRadixSort::SortKeys(d_tmp.p, tb, d_keys.as<uint64_t>(), d_sorted.as<uint64_t>(), (int)a.n_keys, 0, 64, (hipStream_t)0));
      launch_build_any(chunks, a_ins, a_link, 0, 1);
      HTRY(hipGetLastError());
    }
    HTRY(hipDeviceSynchronize());
    unsigned long long bs[4] = {0, 0, 0, 0};
    HTRY(hipMemcpy(bs, d_bstats.p, sizeof(bs), hipMemcpyDeviceToHost));
    ix->build_truncated = (int64_t)bs[0];
    ix->build_prunes = (int64_t)bs[1];
    ix->build_dropped = (int64_t)bs[2];
    ix->build_rounds = (int64_t)rounds.size();
    // ---- the finished graph back to the host copy (export, files).  The map holds HnswNode(level, item) for every wired item
    //      and layer; the first item was never wired (:184-186: no entry point yet) and has a key only where a back link put one
    std::vector<uint32_t> adj0((size_t)n * (ix->m0 + 1));
    HTRY(hipMemcpy(adj0.data(), ix->adj0.p, adj0.size() * 4, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) {
      const uint32_t *row = &adj0[(size_t)i * (ix->m0 + 1)];
      ix->level0[(size_t)i].assign(row + 1, row + 1 + row[0]);
      ix->has0[(size_t)i] = (row[0] > 0 || i != ix->entry) ? 1 : 0;
    }
    const int64_t urows = ix->upper_base_h.back();
    std::vector<uint32_t> uadj((size_t)std::max<int64_t>(urows, 1) * (ix->m + 1));
    if (urows > 0) HTRY(hipMemcpy(uadj.data(), ix->upper_adj.p, (size_t)urows * (ix->m + 1) * 4, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) {
      const int32_t sl = ix->upper_slot_h[(size_t)i];
      if (sl < 0) continue;
      for (int l = 1; l <= levels[(size_t)i]; ++l) {
        const uint32_t *row = uadj.data() + (size_t)(ix->upper_base_h[(size_t)sl] + l - 1) * (ix->m + 1);
        ix->upper[(size_t)l - 1][(size_t)sl].assign(row + 1, row + 1 + row[0]);
        ix->has_upper[(size_t)sl][(size_t)l - 1] = (row[0] > 0 || i != ix->entry) ? 1 : 0;
      }
    }
  }
  if (ids && n > 0) {
    HTRY(ix->ids.reserve((size_t)n * 8));
    HTRY(hipMemcpy(ix->ids.p, ids, (size_t)n * 8, hipMemcpyHostToDevice));
    ix->has_ids = true;
  }
  *out = ix.release();
  return HNSW_OK;
}

int hnsw_index_build_insert_gpu(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids,
                                int32_t max_m, int32_t ef_construction, uint64_t seed, int32_t batch, hnsw_index_t **out) try {
  return build_insert_gpu_impl(device, metric, n, d, vectors, ids, max_m, ef_construction, seed, nullptr, batch, out);
} ABI_CATCH

int hnsw_index_build_insert_gpu_levels(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids,
                                       int32_t max_m, int32_t ef_construction, const int32_t *levels, int32_t batch, hnsw_index_t **out) try {
  if (!levels && n > 0) return fail(HNSW_EINVAL, "levels is NULL");
  return build_insert_gpu_impl(device, metric, n, d, vectors, ids, max_m, ef_construction, 0, levels, batch, out);
} ABI_CATCH

int hnsw_index_build_stats(const hnsw_index_t *ix, int64_t *rounds, int64_t *unseen_additions, int64_t *queue_prunes, int64_t *dropped_candidates) try {
  if (!ix) return fail(HNSW_EINVAL, "NULL index");
  if (rounds) *rounds = ix->build_rounds;
  if (unseen_additions) *unseen_additions = ix->build_truncated;
  if (queue_prunes) *queue_prunes = ix->build_prunes;
  if (dropped_candidates) *dropped_candidates = ix->build_dropped;
  return HNSW_OK;
} ABI_CATCH

int hnsw_index_graph_size(const hnsw_index_t *ix, int64_t *n_entries, int64_t *n_neighbours, int64_t *entry_point,
                          int32_t *max_level) try {
  if (!ix) return fail(HNSW_EINVAL, "NULL index");
  int64_t ne = 0, nn = 0;
  for (int64_t i = 0; i < ix->n; ++i)
    if (ix->has0[(size_t)i]) {
      ne++;
      nn += (int64_t)ix->level0[(size_t)i].size();
    }
  for (size_t l = 0; l < ix->upper.size(); ++l)
    for (size_t s = 0; s < ix->upper[l].size(); ++s)
      if (l < ix->has_upper[s].size() && ix->has_upper[s][l]) {
        ne++;
        nn += (int64_t)ix->upper[l][s].size();
      }
  if (n_entries) *n_entries = ne;
  if (n_neighbours) *n_neighbours = nn;
  if (entry_point) *entry_point = ix->entry;
  if (max_level) *max_level = ix->max_level;
  return HNSW_OK;
} ABI_CATCH

int hnsw_index_graph(const hnsw_index_t *ix, int32_t *entry_level, int64_t *entry_item, int64_t *entry_offsets,
                     int64_t *entry_neighbours) try {
  if (!ix || !entry_level || !entry_item || !entry_offsets) return fail(HNSW_EINVAL, "NULL argument");
  int64_t e = 0, pos = 0;
  entry_offsets[0] = 0;
  auto emit = [&](int level, int64_t item, const std::vector<uint32_t> &list) {
    entry_level[e] = level;
    entry_item[e] = item;
    for (uint32_t v : list) entry_neighbours[pos++] = v;
    entry_offsets[++e] = pos;
  };
  for (int64_t i = 0; i < ix->n; ++i)
    if (ix->has0[(size_t)i]) emit(0, i, ix->level0[(size_t)i]);
  std::vector<int64_t> slot_item(ix->has_upper.size(), -1);
  for (int64_t i = 0; i < ix->n; ++i)
    if (ix->upper_slot_h[(size_t)i] >= 0) slot_item[(size_t)ix->upper_slot_h[(size_t)i]] = i;
  for (size_t l = 0; l < ix->upper.size(); ++l)
    for (size_t s = 0; s < ix->upper[l].size(); ++s)
      if (l < ix->has_upper[s].size() && ix->has_upper[s][l]) emit((int)l + 1, slot_item[s], ix->upper[l][s]);
  return HNSW_OK;
} ABI_CATCH

int hnsw_index_get_vectors(const hnsw_index_t *ix, int64_t i0, int64_t n, float *out) try {
  if (!ix || !out || i0 < 0 || n < 0 || i0 + n > ix->n) return fail(HNSW_EINVAL, "range outside the index");
  if (n == 0) return HNSW_OK;
  HTRY(hipSetDevice(ix->device));
  // in slabs: a launch stays far below 2^32 work-items (50M x 256 elements in ONE launch is 1.28e10 -- the grid wrapped and
  // most of the buffer came back unwritten: profiles/r03_hnsw_bench_50M_*) and the staging buffer below 1 GiB
  const int64_t slab = std::max<int64_t>(1, ((int64_t)1 << 28) / ix->d);  // rows per launch: <= 2^28 elements
  Buf tmp;
  HTRY(tmp.reserve((size_t)std::min(slab, n) * ix->d * 4));
  for (int64_t r0 = 0; r0 < n; r0 += slab) {
    const int64_t m = std::min(slab, n - r0), e = m * ix->d;
    hipLaunchKernelGGL(hnsw_rows_to_f32, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, 0, ix->x.as<_Float16>(), i0 + r0, m, ix->d,
                       ix->dpad, tmp.as<float>());
    HTRY(hipGetLastError());
    HTRY(hipMemcpy(out + (size_t)r0 * ix->d, tmp.p, (size_t)e * 4, hipMemcpyDeviceToHost));
  }
  return HNSW_OK;
} ABI_CATCH

int hnsw_index_info(const hnsw_index_t *ix, int64_t *n, int32_t *d, int32_t *metric, int32_t *max_m) try {
  if (!ix) return fail(HNSW_EINVAL, "NULL index");
  if (n) *n = ix->n;
  if (d) *d = ix->d;
  if (metric) *metric = ix->metric;
  if (max_m) *max_m = ix->m;
  return HNSW_OK;
} ABI_CATCH

int hnsw_index_get_ids(const hnsw_index_t *ix, int64_t *out) try {
  if (!ix || (!out && ix->n > 0)) return fail(HNSW_EINVAL, "NULL argument");
  if (ix->n == 0) return HNSW_OK;
  if (!ix->has_ids) {
    for (int64_t i = 0; i < ix->n; ++i) out[i] = i;
    return HNSW_OK;
  }
  HTRY(hipSetDevice(ix->device));
  HTRY(hipMemcpy(out, ix->ids.p, (size_t)ix->n * 8, hipMemcpyDeviceToHost));
  return HNSW_OK;
} ABI_CATCH

int hnsw_index_destroy(hnsw_index_t *ix) try {
  delete ix;
  return HNSW_OK;
} ABI_CATCH

int hnsw_search(hnsw_index_t *ix, int32_t nq, const float *queries, int32_t k, int32_t ef, float *out_dist, int64_t *out_ids,
                int32_t *out_counts) try {
  if (!ix || !queries || !out_dist || !out_ids || !out_counts) return fail(HNSW_EINVAL, "NULL argument");
  if (nq < 1) return fail(HNSW_EINVAL, "nq must be positive");
  if (k < 1 || ef < 1) return fail(HNSW_EINVAL, "k and ef must be positive");
  const int beam = std::max(ef, k);  // HnswIndex.java:545
  if (beam > MAX_EF) return fail(HNSW_ELIMIT, "max(ef, k) above 1024");
  ix->last_dist = ix->last_exp = 0;
  ix->last_spilled = 0;
  ix->last_ms = 0;
  if (ix->entry < 0) {  // metadata.getEntryPoint() absent: Collections.emptyList() (:550-552)
    for (int32_t q = 0; q < nq; ++q) out_counts[q] = 0;
    return HNSW_OK;
  }
  HTRY(hipSetDevice(ix->device));
  const int64_t vwords = (ix->n + 31) / 32;
  // concurrent queries per launch: bounded by the visited bitmaps -- n bits per query, up to 48 GiB of the 288 (at 50M
  // vectors a bitmap is 6.25 MB; the 2 GiB this was capped at until round 3 let 343 walks run at a time, 1.3 per CU,
  // and a 4096-query batch took twelve launches: profiles/r03_hnsw_bench_50M_*)
  int64_t per_launch = std::min<int64_t>(nq, std::max<int64_t>(64, (int64_t)(48ull << 30) / (vwords * 4)));
  per_launch = std::min<int64_t>(per_launch, 1 << 16);
  HTRY(ix->q_in.reserve((size_t)nq * ix->d * 4));
  HTRY(ix->q.reserve((size_t)nq * ix->dpad * sizeof(_Float16)));
  {
    const void *before = ix->visited.p;
    HTRY(ix->visited.reserve((size_t)per_launch * vwords * 4));
    if (ix->visited.p != before) ix->visited_dirty = true;
  }
  const bool use_vlog = vwords >= VLOG_MIN_VWORDS || getenv("HNSW_DEBUG_VLOG") != nullptr;  // (the variable: tests force the log on small graphs)
  HTRY(ix->o_dist.reserve((size_t)nq * k * 4));
  HTRY(ix->o_ids.reserve((size_t)nq * k * 8));
  HTRY(ix->o_cnt.reserve((size_t)nq * 4));
  HTRY(ix->spill.reserve((size_t)nq * 4));
  HTRY(ix->stats.reserve(128));
  hipStream_t st = 0;
  HTRY(hipMemcpyAsync(ix->q_in.p, queries, (size_t)nq * ix->d * 4, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(hnsw_prep_rows, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st, ix->q_in.as<float>(), (int64_t)nq, ix->d,
                     ix->dpad, ix->metric == HNSW_METRIC_COSINE ? 1 : 0, ix->q.as<_Float16>());
  HTRY(hipGetLastError());
  HTRY(hipMemsetAsync(ix->stats.p, 0, 128, st));
  HTRY(hipMemsetAsync(ix->spill.p, 0, (size_t)nq * 4, st));

  SearchArgs a;
  a.x = ix->x.as<_Float16>();
  a.adj0 = ix->adj0.as<uint32_t>();
  a.upper_slot = ix->upper_slot.as<int32_t>();
  a.upper_base = ix->upper_base.as<int32_t>();
  a.upper_adj = ix->upper_adj.as<uint32_t>();
  a.ids = ix->has_ids ? ix->ids.as<int64_t>() : nullptr;
  a.visited = ix->visited.as<uint32_t>();
  if (use_vlog) HTRY(ix->vlog.reserve((size_t)per_launch * VLOG_CAP * 4));
  a.vlog = use_vlog ? ix->vlog.as<uint32_t>() : nullptr;
  a.gc = nullptr;
  a.out_dist = ix->o_dist.as<float>();
  a.out_ids = ix->o_ids.as<int64_t>();
  a.out_counts = ix->o_cnt.as<int32_t>();
  a.spill = ix->spill.as<int32_t>();
  a.stats = ix->stats.as<unsigned long long>();
  a.vwords = vwords;
  a.dpad = ix->dpad;
  a.chunks = ix->dpad / 64;
  a.m = ix->m;
  a.m0 = ix->m0;
  a.metric = ix->metric;
  a.k = k;
  a.ef = beam;
  a.max_level = ix->max_level;
  a.entry = (uint32_t)ix->entry;
  // Every admission enters the candidate queue and stale ones stay: it reaches 2-3x the beam on clustered data and far more
  // on structureless data (i.i.d. Gaussians).  Its top is in LDS -- as much as WALK_LDS_BYTES leaves beside the result queue
  // -- and the rest in global memory: CCAP_FIRST entries per query in the first pass, CCAP_GLOBAL for the queries that
  // outgrow that (pass 2, at most 256 at a time).
  int lds_n = std::max(64, std::min(CCAP_LDS, (int)((WALK_LDS_BYTES - (size_t)(beam + 1) * sizeof(HEntry)) / sizeof(HEntry))));
  if ((size_t)(beam + 1) * sizeof(HEntry) + 64 * sizeof(HEntry) > (size_t)WALK_LDS_BYTES) lds_n = 64;
  int gcap_first = CCAP_FIRST;
  if (const char *e = getenv("HNSW_DEBUG_CCAP")) {  // tests: v > 0: v entries in LDS and v in the first pass's global part (so queries spill into pass 2); v < 0: |v| in LDS, the usual global part
    const int v = atoi(e);
    lds_n = std::min(CCAP_LDS, std::max(1, v < 0 ? -v : v));
    if (v > 0) gcap_first = v;
  }

  // the visited bitmaps: wiped when the buffer is new (or a search was cut short); every walk leaves its own clean (visited_undo)
  if (use_vlog && ix->visited_dirty) HTRY(hipMemsetAsync(ix->visited.p, 0, ix->visited.bytes, st));
  ix->visited_dirty = true;  // (until this search has run to its end)
  HTRY(hipEventRecord(ix->ev[0], st));
  std::vector<int32_t> redo, spill((size_t)nq);
  for (int pass = 0; pass < 2; ++pass) {
    const bool first = pass == 0;
    if (!first && redo.empty()) break;
    const int64_t todo = first ? nq : (int64_t)redo.size();
    const int gcap = first ? gcap_first : CCAP_GLOBAL;
    int64_t batch = std::min<int64_t>(per_launch, first ? (int64_t)1 << 16 : 256);
    batch = std::min<int64_t>(batch, todo);
    HTRY(ix->gc.reserve((size_t)batch * gcap * sizeof(HEntry)));
    if (!first) {
      HTRY(ix->qlist.reserve(redo.size() * 4));
      HTRY(hipMemcpyAsync(ix->qlist.p, redo.data(), redo.size() * 4, hipMemcpyHostToDevice, st));
    }
    for (int64_t r0 = 0; r0 < todo; r0 += batch) {
      const int64_t m = std::min<int64_t>(batch, todo - r0);
      if (!use_vlog) HTRY(hipMemsetAsync(ix->visited.p, 0, (size_t)m * vwords * 4, st));
      SearchArgs b = a;
      if (first) {
        b.q = ix->q.as<_Float16>() + r0 * ix->dpad;
        b.qlist = nullptr;
        b.out_dist = a.out_dist + r0 * k;
        b.out_ids = a.out_ids + r0 * k;
        b.out_counts = a.out_counts + r0;
        b.spill = a.spill + r0;
      } else {
        b.q = ix->q.as<_Float16>();
        b.qlist = ix->qlist.as<int32_t>() + r0;
      }
      b.gc = ix->gc.as<KEntry>();
      b.ccap_lds = lds_n;
      b.gcap = gcap;
      int rc = launch_search_any(a.chunks, (int)m, b, st);
      if (rc) return rc;
      HTRY(hipGetLastError());
    }
    HTRY(hipMemcpyAsync(spill.data(), ix->spill.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    HTRY(hipStreamSynchronize(st));
    std::vector<int32_t> still;
    if (first) {
      for (int32_t q = 0; q < nq; ++q)
        if (spill[(size_t)q]) still.push_back(q);
      ix->last_spilled = (int32_t)still.size();
    } else {
      for (int32_t q : redo)
        if (spill[(size_t)q]) still.push_back(q);
    }
    redo.swap(still);
  }
  if (!redo.empty()) return fail(HNSW_ELIMIT, "candidate queue above 131072 entries");
  HTRY(hipEventRecord(ix->ev[1], st));
  unsigned long long stats[16] = {0};
  HTRY(hipMemcpyAsync(stats, ix->stats.p, 128, hipMemcpyDeviceToHost, st));
  HTRY(hipMemcpyAsync(out_dist, ix->o_dist.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st));
  HTRY(hipMemcpyAsync(out_ids, ix->o_ids.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st));
  HTRY(hipMemcpyAsync(out_counts, ix->o_cnt.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
  HTRY(hipStreamSynchronize(st));
  ix->visited_dirty = !use_vlog;
  ix->last_dist = (int64_t)stats[0];
  ix->last_exp = (int64_t)stats[1];
  ix->last_peak = (int64_t)stats[2];
  ix->last_adm = (int64_t)stats[3];
  (void)hipEventElapsedTime(&ix->last_ms, ix->ev[0], ix->ev[1]);
  return HNSW_OK;
} ABI_CATCH

int hnsw_last_stats(const hnsw_index_t *ix, int64_t *distance_evals, int64_t *expansions, int32_t *spilled_queries,
                    float *kernel_ms) try {
  if (!ix) return fail(HNSW_EINVAL, "NULL index");
  if (distance_evals) *distance_evals = ix->last_dist;
  if (expansions) *expansions = ix->last_exp;
  if (spilled_queries) *spilled_queries = ix->last_spilled;
  if (kernel_ms) *kernel_ms = ix->last_ms;
  return HNSW_OK;
} ABI_CATCH

int hnsw_last_walk_counters(const hnsw_index_t *ix, int64_t *admissions, int64_t *largest_candidate_queue) try {
  if (!ix) return fail(HNSW_EINVAL, "NULL index");
  if (admissions) *admissions = ix->last_adm;
  if (largest_candidate_queue) *largest_candidate_queue = ix->last_peak;
  return HNSW_OK;
} ABI_CATCH

}  // extern "C"
