// sann_fast.hip -- fast path of the (query, partition) work unit, gfx950.
//
// desc_query_kernel / desc_kernel   per (query, scanned cluster, partition): sub-list start and the number of its postings
//               with rank < M (a cached cut table, or a binary search in `ranks`; the `i < min(size, M)` cut of
//               ApproximateCosineSimilarity.scala:87), the exclusive prefix over the clusters, and for the cosine forms
//               the cut of every unit (per unit and per query, see 3 below).  The lookups are a chain of dependent loads
//               (scan_row -> sub_offsets -> cut / ranks); doing them here keeps that chain out of the unit kernel.  Rows
//               and cluster weights are written at a FIXED stride (64 / 128 entries, padded), so that the unit kernel
//               finds them from its block index alone.  One workgroup per query (P >= 16), one wave per query (P = 4, 8:
//               a shard's queries), or one wave per unit (P < 4).
//
// unit_fast_kernel   one workgroup = one unit; the unit's postings live in REGISTERS (U per thread: an fp32 copy of
//               the score, the cluster's sequence number and a hash of the id -- the fp64 posting is read again only for
//               survivors, by the MERGE kernel).  What bounds it is the time a workgroup holds its slot -- a chain of
//               dependent memory trips -- at eight workgroups per CU (<= 64 VGPRs, 20 KB LDS, <= 80 SGPRs):
//   1. descriptors  header, (start, prefix) row, cluster weights and posting count in ONE trip; byte map flat posting
//                   index -> cluster; per-cluster constants (fp64 weight, fp32 weight, and for the cosine forms the fp32
//                   KEY of a single-cluster candidate: (s w) / sqrt(s s) = w, so such a candidate needs no arithmetic)
//   2. gather       one 16-B global load per posting, consecutive lanes = consecutive postings of a sub-list; all of a
//                   thread's loads are issued (inline asm: hipcc otherwise waits after each) before the single wait;
//                   age window and source-tweet filters (:90-91)
//   3. cluster cut  (cosine forms) the cut is a property of the descriptors: read from the descriptor kernel's output
//   3a/b. duplicates  a tweet can sit in several scanned clusters (all its postings are in this unit by construction of
//                   the partition hash).  Each posting ORs four hash bits into one 64-bit word of a blocked Bloom
//                   filter with ONE LDS atomic; finding all already set flags the id.  Units with a flag (one in
//                   three at the benchmark's shape) look their hashes up in the flagged filter, fetch the matching
//                   postings again and resolve the flagged ids through a small match list -- pairwise for a handful of
//                   entries, sorted in registers for 13..64 --, groups summed in cluster order, so fp64 sums follow the
//                   reference's accumulation order (:83-100) whatever the timing.
//   4. keys         APPROXIMATE fp32 score per live candidate (|approx/exact - 1| <= EPS), as an order-preserving u32
//   5a. data cut    (other forms, or when filters thinned what 3 counted on) the kl-th largest of the 256 per-thread
//                   maxima, found by an in-register bitonic sort per wave (DPP / permlane swaps) and a rank search in LDS
//   5b. compact     survivors (key >= cut) into LDS as (cluster, flat posting index): one LDS atomic per wave
//   6. hand over    the survivors (tens, out of ~1250) leave as (cluster, posting position) with key(theta), theta =
//                   cut (1 + 2 EPS); the merge kernel computes their exact fp64 scores (:111-125) and proves the global
//                   top-k exact against the units' thetas.  Representatives of multi-cluster tweets are finished here.
//
// Units that do not fit (too many clusters / postings / flagged ids / survivors, scores outside the fp32 range) flag
// UNIT_OVERFLOW and are re-run by unit_general_kernel.
//
// Build rule (csrc/Makefile, tools/check_unit_kernel_resources.py): no instantiation of unit_fast_kernel may use
// scratch.  hipcc (ROCm 7.2) was seen to place a VGPR spill store in FRONT of the `s_or_b64 exec` that re-joins a
// divergent branch, so lanes that sat the branch out reloaded garbage: results differed from run to run.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "sann_device.h"
#include "sann_kernels.h"
#include "sann_math.h"
#include "sann_select.h"
#include "sann_unit.h"
#include "sann_wave.h"

namespace sann {

// ---------------------------------------------------------------------------------------------
// Cut table for one value of M: out[row*P + p] = number of postings of sub-list (row, p) with rank < M.
__global__ __launch_bounds__(256) void cut_kernel(IndexView ix, int M, uint32_t *out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)ix.n_rows * ix.P) return;
  const uint32_t base = ix.sub_offsets[i], end = ix.sub_offsets[i + 1];
  const int n = (int)(end - base);
  out[i] = (n > 0 && ix.ranks[base + n - 1] < (uint32_t)M) ? (uint32_t)n : (uint32_t)lower_bound_rank(ix.ranks + base, n, (uint32_t)M);
}
hipError_t launch_cut(const IndexView &ix, int M, uint32_t *out, hipStream_t stream) {
  const int64_t n = (int64_t)ix.n_rows * ix.P;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(cut_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ix, M, out);
  return hipGetLastError();
}

// One WAVE per unit: lane c (and c + 64) resolves cluster c's sub-list, the wave scans the lengths,
// and the unit kernel later reads (start, exclusive prefix) pairs and the unit's posting count.
__global__ __launch_bounds__(256) void desc_kernel(IndexView ix, BatchView b, int n_units, int k_local_floor) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int unit = blockIdx.x * 4 + wave;
  // per-run state the unit and merge kernels update: cleared here instead of by two memset launches
  // (status: overflow count, inexact count, inexact list -- nq + 2 ints; one thread each)
  {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < b.nq + 2) b.status[g] = 0;
  }
  if (unit >= n_units) return;
  if (lane == 0) b.unit_fb[unit] = -1;
  const int q = unit >> ix.log2P;
  const int p = unit & (ix.P - 1);
  const int M = b.hdr[q].M;
  const int scan_begin = b.hdr[q].scan_begin;
  const int n_scan = b.hdr[q].n_scan;
  if (n_scan > NSCAN_MAX) {  // the unit kernel sends such units to the general path
    if (lane == 0) b.unit_T[unit] = 0;
    return;
  }
  const uint32_t *cut = nullptr;  // cached cut table for this query's M, if any (uniform)
#pragma unroll
  for (int j = 0; j < 4; j++)
    if (b.cut_M[j] == M) cut = b.cut[j];
  uint32_t base[2], len[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int c = lane + 64 * r;
    base[r] = 0;
    len[r] = 0;
    if (c < n_scan) {
      const int row = b.scan_row[scan_begin + c];
      base[r] = ix.sub_offsets[(int64_t)row * ix.P + p];
      const uint32_t end = ix.sub_offsets[(int64_t)row * ix.P + p + 1];
      const int n = (int)(end - base[r]);
      // postings with rank < M are a prefix of the sub-list
      if (cut) len[r] = cut[(int64_t)row * ix.P + p];
      else
        len[r] = (n > 0 && ix.ranks[base[r] + n - 1] < (uint32_t)M) ? (uint32_t)n
                                                                   : (uint32_t)lower_bound_rank(ix.ranks + base[r], n, (uint32_t)M);
    }
  }
  uint32_t incl0 = len[0], incl1 = len[1];
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t0 = __shfl_up(incl0, off, 64), t1 = __shfl_up(incl1, off, 64);
    if (lane >= off) { incl0 += t0; incl1 += t1; }
  }
  const uint32_t total0 = __shfl(incl0, 63, 64), total1 = __shfl(incl1, 63, 64);
  // the unit's row: desc_stride (start, prefix) pairs, padded behind n_scan with (0, T); the query's weights likewise
  // (written by the query's first unit)
  {
    uint2 *d = reinterpret_cast<uint2 *>(b.desc) + (int64_t)unit * b.desc_stride;
    const uint32_t T = total0 + total1;
    d[lane] = lane < n_scan ? make_uint2(base[0], incl0 - len[0]) : make_uint2(0u, T);
    if (b.desc_stride > 64) d[lane + 64] = lane + 64 < n_scan ? make_uint2(base[1], total0 + incl1 - len[1]) : make_uint2(0u, T);
    if (p == 0) {
      double *wq = b.scan_wq + (int64_t)q * b.desc_stride;
      wq[lane] = lane < n_scan ? b.scan_w[scan_begin + lane] : 0.0;
      if (b.desc_stride > 64) wq[lane + 64] = lane + 64 < n_scan ? b.scan_w[scan_begin + lane + 64] : 0.0;
    }
  }
  if (lane == 0) b.unit_T[unit] = (int32_t)(total0 + total1);
  // cluster-level cut (n_scan <= 64: lane c = cluster c): clusters by key, descending, with the low 8 bits of the key
  // replaced by c (keys 256 ulps apart tie; the cut is taken 256 ulps low anyway)
  const QueryHdr h = b.hdr[q];
  uint32_t pre = 0u;
  if (query_has_cluster_cut(h)) {  // (uniform)
    const float inv_l2_32 = h.inv_l2_32;
    const uint32_t kc = lane < n_scan ? cosine_cluster_key(h.alg, b.scan_w[scan_begin + lane], inv_l2_32) : 0u;
    const uint32_t pk = wave_sort_desc_u32(kc ? ((kc & ~0xffu) | (uint32_t)lane) : 0u);
    const int c = (int)(pk & 0xffu);
    int cum = pk ? (int)__shfl(len[0], c, 64) : 0;
    const uint32_t kc_sorted = __shfl(kc, c, 64);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(cum, off, 64);
      cum += lane >= off ? t : 0;
    }
    const unsigned long long ok = __ballot(pk != 0u && cum >= unit_kl(h.k, ix.P, k_local_floor));
    if (ok != 0ull) pre = cluster_cut_from_key(__shfl(kc_sorted, __ffsll((long long)ok) - 1, 64));
  }
  if (lane == 0) b.unit_pre[unit] = pre;
}

// Kernel arguments that only the last lines of the unit kernel need, read from the kernarg segment WHEN they are needed.
// hipcc loads every argument in the first lines of a kernel; with the 80 SGPRs that eight waves per SIMD leave a wave
// (MI355X_MICROARCH.md: 16 more are the trap handler's), the pointers of the output arrays then sat in spilled SGPRs --
// v_writelane / v_readlane pairs, a tenth of the kernel's vector instructions -- for the whole kernel.  A volatile load is
// not hoisted.  (Arguments: IndexView at 0, BatchView behind it.)
typedef const __attribute__((address_space(4))) char *kernarg_ptr;
constexpr size_t KERNARG_BATCH = (sizeof(IndexView) + alignof(BatchView) - 1) / alignof(BatchView) * alignof(BatchView);
template <class T> __device__ inline T late_kernarg(size_t off) {
  kernarg_ptr ka = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  return *(const volatile __attribute__((address_space(4))) T *)(ka + off);
}
#define LATE_ARG(field) late_kernarg<decltype(BatchView::field)>(KERNARG_BATCH + offsetof(BatchView, field))

// second launch bound = waves per SIMD: small units are asked to fit 8 waves (<= 64 VGPRs); the
// rare duplicate-resolution code may spill, the common path does not
// ABL != 0: measurement builds that stop after a phase (sann_debug_gather_probe modes 11..15; results are garbage)
#define ABLATE(n, expr)                                                                                      \
  if constexpr (ABL == (n)) {                                                                                \
    unsigned long long sink_ = (expr);                                                                       \
    if (sink_ == 0x123456789abcdefull) b.unit_thr[2 * (int64_t)unit] = sink_;                                \
    return;                                                                                                  \
  }
// NS = scanned clusters the unit's tables hold (64 when the batch's queries scan at most 64 -- the production N is 50 --
// else NSCAN_MAX): with the match list moved into the dead Bloom filter's memory the six-slot geometry then needs
// 20.0 KB of LDS and 64 registers, i.e. EIGHT workgroups per CU (the hardware's 32 waves) instead of six.
// NORMS = the batch holds queries of the offline job's forms (QueryHdr.use_norms): a candidate's normaliser is the
// norms column of the index, carried as a seventh fp32 per posting -- a separate instantiation, so that the online
// kernel's 64 registers are not touched.
// a unit the fast path cannot hold: listed for the general path (one thread); reason -> unit_thr[2u + 1], for diagnostics
__device__ inline void unit_overflowed(int unit, unsigned long long reason) {
  LATE_ARG(cand_cnt)[unit] = 0;
  LATE_ARG(unit_unique)[unit] = 0;
  LATE_ARG(unit_flags)[unit] = UNIT_OVERFLOW;
  uint64_t *const thr = LATE_ARG(unit_thr);
  thr[2 * (int64_t)unit] = 0;
  thr[2 * (int64_t)unit + 1] = reason;
  const int o = atomicAdd(&LATE_ARG(status)[0], 1);
  LATE_ARG(overflow_units)[o] = unit;
}

template <int WG, int U, int NS = NSCAN_MAX, int ABL = 0, bool NORMS = false>
__global__ __launch_bounds__(WG, (WG < 256 ? 2 : NORMS ? (U <= 8 ? 5 : 2) : U <= 6 ? 8 : U <= 8 ? 5 : U <= 12 ? 3 : 2)) void unit_fast_kernel(IndexView ix, BatchView b, int k_local_floor, int n_blocks_q8) {
  constexpr int SCAP = FAST_SCAP;
  constexpr int BW = bloom_log2(WG * U);  // log2 of the Bloom filter's 64-bit words
  constexpr int BLOOM_ALLOC = 1 << BW;
  constexpr int HB = BLOOM_POS_BITS + BW;  // hash bits: 4 x 5 bit positions, then the word index
  constexpr int FBLOOM_WORDS = BW >= 11 ? 64 : 256;  // (beside the 16 KB Bloom filter: 64, so that eight workgroups fit a CU)
  constexpr int FB = BW >= 11 ? 6 : 8;
  constexpr int MCAP = (WG * U <= 1024) ? 64 : 128;
  __shared__ unsigned long long s_bloom[BLOOM_ALLOC];
  __shared__ uint32_t s_begin[NS];
  __shared__ uint32_t s_pre[NS];
  __shared__ uint8_t s_map[WG * U];  // flat posting index -> cluster sequence number
  __shared__ double s_w[NS];
  __shared__ float s_w32[NS];
  __shared__ uint32_t s_wkey[NS];  // cosine forms: the fp32 key of a single-cluster candidate of the cluster (0 = untrusted)
  __shared__ unsigned long long s_fbloom[FBLOOM_WORDS];
  // The survivor list, the radix histogram / lane maxima and the match list of the duplicate phase are first touched
  // after the Bloom filter is dead (the barrier that ends phase 2 lies between): they live in its memory when it is
  // large enough.
  constexpr bool ALIAS = BLOOM_ALLOC >= SCAP + 128;
  constexpr int M_OFF = SCAP + 128;  // in 8-byte words: behind the survivor list and the histogram
  constexpr bool ALIAS_M = ALIAS && BLOOM_ALLOC >= M_OFF + 5 * MCAP;
  __shared__ unsigned long long s_ent_own[ALIAS ? 1 : SCAP];
  __shared__ unsigned s_hist_own[ALIAS ? 1 : 256];
  __shared__ long long s_Mid_own[ALIAS_M ? 1 : MCAP];
  __shared__ double s_Msc_own[ALIAS_M ? 1 : MCAP], s_Mdot_own[ALIAS_M ? 1 : MCAP], s_Mnsq_own[ALIAS_M ? 1 : MCAP];
  __shared__ int s_Mseq_own[ALIAS_M ? 1 : MCAP], s_Mrole_own[ALIAS_M ? 1 : MCAP];
  __shared__ double s_Mnrm[NORMS ? MCAP : 1];  // offline forms: the full norm of the match-list entry's tweet
  // survivor list: (cluster sequence number or 0x10000 | match-list entry) << 32 | flat index of the posting in the unit
  unsigned long long *const s_ent = ALIAS ? s_bloom : s_ent_own;
  unsigned *const s_hist = ALIAS ? reinterpret_cast<unsigned *>(s_bloom + SCAP) : s_hist_own;
  long long *const s_Mid = ALIAS_M ? reinterpret_cast<long long *>(s_bloom + M_OFF) : s_Mid_own;
  double *const s_Msc = ALIAS_M ? reinterpret_cast<double *>(s_bloom + M_OFF + MCAP) : s_Msc_own;
  double *const s_Mdot = ALIAS_M ? reinterpret_cast<double *>(s_bloom + M_OFF + 2 * MCAP) : s_Mdot_own;
  double *const s_Mnsq = ALIAS_M ? reinterpret_cast<double *>(s_bloom + M_OFF + 3 * MCAP) : s_Mnsq_own;
  int *const s_Mseq = ALIAS_M ? reinterpret_cast<int *>(s_bloom + M_OFF + 4 * MCAP) : s_Mseq_own;
  int *const s_Mrole = ALIAS_M ? reinterpret_cast<int *>(s_bloom + M_OFF + 4 * MCAP) + MCAP : s_Mrole_own;
  __shared__ int s_ctl[CTL_N];

  const int tid = threadIdx.x;
  // XCD-aware mapping: consecutive blocks go to different XCDs (round robin over 8), so give all
  // P units of a query the same blockIdx % 8: they then share one L2 for the query's postings.
  const int blk = blockIdx.x;
  const int x = blk & 7, r = blk >> 3;
  const int p = r & (ix.P - 1);
  const int q = ((r >> ix.log2P) << 3) + x;
  if (q >= b.nq) return;
  const int unit = q * ix.P + p;
  const QueryHdr h = b.hdr[q];
#define STAMP(i) do { if (b.prof && tid == 0) b.prof[(int64_t)unit * 16 + (i)] = (unsigned long long)clock64(); } while (0)
  STAMP(0);

  // ---- 0. clear ----------------------------------------------------------------------------
  if (tid < CTL_N) s_ctl[tid] = (tid == CTL_KMIN) ? -1 : 0;
  for (int i = tid; i < BLOOM_ALLOC; i += WG) s_bloom[i] = 0ull;
  for (int i = tid; i < FBLOOM_WORDS; i += WG) s_fbloom[i] = 0ull;

  const bool overflow_n = h.n_scan > NS;  // uniform
  // ---- 1. descriptors ----------------------------------------------------------------------
  // The unit's posting count T, its descriptor row and the query's cluster weights are found from the block index alone
  // (rows of NS entries at a fixed stride, padded behind the query's n_scan clusters with (0, T) / weight 0): they are
  // loaded together with the query header, in ONE trip to memory.  (Until round 2 the row lay at a compact offset that
  // the header had to supply first: header -> row -> postings were three dependent trips, now two.)  Nothing below
  // branches on T or on the header before these loads are issued.
  // (Tv stays a per-lane register until after the loop: a scalar copy would make hipcc wait for it right here)
  uint32_t Tv = (uint32_t)b.unit_T[unit];  // (0 for a query that scans more than NSCAN_MAX clusters)
  const float inv_l2_32 = h.inv_l2_32;
  {
    const uint2 *d = reinterpret_cast<const uint2 *>(b.desc) + (int64_t)unit * NS;
    const double *wq = b.scan_wq + (int64_t)q * NS;
    // four lanes per cluster: lane part (0..3) fills a quarter of the cluster's stretch of the map
#pragma unroll
    for (int t0 = 0; t0 < 4 * NS; t0 += WG) {
      const int t = t0 + tid;
      if (4 * NS % WG != 0 && t >= 4 * NS) break;
      const int c = t >> 2, part = t & 3;
      const uint2 v = d[c];
      const bool last = c + 1 >= NS;
      uint32_t nxt = last ? 0u : d[c + 1].y;
      if (part == 0) {
        const double w = wq[c];
        s_begin[c] = v.x;
        s_pre[c] = v.y;
        s_w[c] = w;
        s_w32[c] = (float)w;
        s_wkey[c] = cosine_cluster_key(h.alg, w, inv_l2_32);
      }
      // every posting of this cluster records its cluster in the flat map (an oversized unit stops at the map's end;
      // a padding entry's stretch is empty)
      asm volatile("" : "+v"(nxt));  // keeps the select on Tv below the loads above
      uint32_t next = last ? Tv : nxt;
      next = next < (uint32_t)(WG * U) ? next : (uint32_t)(WG * U);
      for (uint32_t i = v.y + part; i < next; i += 4) s_map[i] = (uint8_t)c;
    }
  }
  if (overflow_n) Tv = 0u;
  asm volatile("" : "+v"(Tv));
  const uint32_t T = (uint32_t)__builtin_amdgcn_readfirstlane((int)Tv);
  bool overflow = overflow_n;
  // overflow reason, for diagnostics (unit_thr[2u+1]): 1 scanned clusters, 2 postings, 3 match list, 4 key range.  Derived
  // at the exit from facts that are uniform anyway, not carried in a register (it was the last value hipcc spilled).
  if (!overflow && T > (uint32_t)(WG * U)) overflow = true;
  const bool overflow_T = overflow && !overflow_n;
  __syncthreads();
  STAMP(1);  // descriptors + map done
  ABLATE(6, (unsigned long long)T + s_map[tid] + s_begin[tid & (NS - 1)] + s_pre[tid & (NS - 1)] + s_wkey[tid & (NS - 1)] + (unsigned long long)s_w[tid & (NS - 1)]);

  // entries a unit must offer before it may withhold the rest (unit_kl), and the size up to which it simply offers
  // everything
  const int kl = unit_kl(h.k, ix.P, k_local_floor);
  const int keep_all = SCAP < kl + kl / 2 + 16 ? SCAP : kl + kl / 2 + 16;
  // Cosine forms: a single-cluster candidate's key is a constant of its cluster, so WHERE to cut was decided from the
  // descriptors alone, by the descriptor kernel (unit_pre), and the data-dependent cut (5a) is skipped.
  const bool use_norms = NORMS && h.use_norms != 0;  // uniform
  const bool pre_cut = query_has_cluster_cut(h) && !overflow;  // uniform
  const uint32_t pre_tau = pre_cut ? b.unit_pre[unit] : 0u;  // (uniform: a scalar load, long back when it is needed)

  // ---- 2. gather ---------------------------------------------------------------------------------------
  // A posting is looked at ONCE: window / source filters on its id, three Bloom bits from a hash of its id, and its
  // score rounded to fp32 for the pre-filter.  What stays in registers per posting is 12 bytes -- (s32, cluster, hash)
  // -- not the 16-byte posting: the few that survive the cut (~4 %) are fetched again (from L2) for their exact
  // fp64 arithmetic.  With the 16-byte form resident, six slots did not fit 80 registers and hipcc kept two of them
  // in scratch memory, re-reading them in every later phase.
  float nrm32[NORMS ? U : 1];  // offline forms: the tweet's full norm, fp32
  float s32[U];   // posting score, fp32
  int seq[U];     // cluster sequence number; bit 16 = group representative (low bits: match-list entry); < 0 = no candidate here
  int live = 0;
  uint32_t hsh[U];  // table_hash of the tweet id (Bloom word and bits): lives until the duplicate phase has looked at it
#pragma unroll
  for (int u = 0; u < U; u++) {
    seq[u] = -1;
    s32[u] = 0.f;
    hsh[u] = 0u;
  }
  {
    // an overflowed unit gathers nothing: with Tg = 0 every slot below is skipped, and no separate control path
    // has to be merged with the loaded registers (the merge made hipcc wait for the first slot's load at once)
    const uint32_t Tg = overflow ? 0u : T;
    // The loads are written as inline asm and waited for by hand.  Left to hipcc, the six 16-byte loads of a thread
    // were given OVERLAPPING destination registers (the id half of one under the score half of the next) with an
    // `s_waitcnt vmcnt(0)` + register copy behind every one of them: six serial trips to memory instead of one.
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 raw[U];
    // Every slot loads, unconditionally (a slot the unit does not reach re-reads the unit's last posting: one cache
    // line for the whole wave): the asm outputs then have no other definition they would have to be merged with --
    // a merge is a register copy placed right behind the load, i.e. a read of registers the load has not written yet.
    if (Tg != 0u) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t j = (uint32_t)(u * WG + tid);
        const uint32_t jj = j < Tg ? j : Tg - 1;
        const int c = (int)s_map[jj];
        seq[u] = (j < Tg) ? c : -1;
        const Posting *src = ix.postings + (s_begin[c] + (jj - s_pre[c]));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(raw[u]) : "v"(src) : "memory");
      }
    STAMP(2);  // posting loads issued
    if constexpr (U == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]) : : "memory");
    else if constexpr (U == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]) : : "memory");
    else if constexpr (U == 6)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]) : : "memory");
    else {
#pragma unroll
      for (int u0 = 0; u0 < U; u0 += 4)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[u0]), "+v"(raw[u0 + 1]), "+v"(raw[u0 + 2]), "+v"(raw[u0 + 3]) : : "memory");
    }
    if constexpr (ABL == 7) {
      unsigned long long x_ = 0;
#pragma unroll
      for (int u = 0; u < U; u++) x_ ^= ((unsigned long long)(raw[u].y ^ raw[u].w) << 32) | (raw[u].x ^ raw[u].z ^ (unsigned)seq[u]);
      ABLATE(7, x_);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if ((uint32_t)(u * WG) < Tg) {
        // NB: selects, not `seq[u] = -1; continue;` -- hipcc (ROCm 7.2) mis-structurised that form
        // and dropped the -1 for postings outside the age window.
        const bool have = seq[u] >= 0;
        const long long idv = (long long)(((unsigned long long)raw[u].y << 32) | raw[u].x);
        const double scv = __longlong_as_double((long long)(((unsigned long long)raw[u].w << 32) | raw[u].z));
        const bool excluded = h.excl_enabled != 0 && idv == h.src_excl;  // :90
        const bool in_window = idv >= h.earliest && idv <= h.latest;     // :91
        const bool keep = have && !excluded && in_window;
        bool keep_n = keep;
        if constexpr (NORMS) {
          nrm32[u] = 1.f;
          if (use_norms) {
            const int c = have ? seq[u] : 0;
            const double nrm = have ? ix.norms[s_begin[c] + ((uint32_t)(u * WG + tid) - s_pre[c])] : 0.0;
            nrm32[u] = (float)nrm;
            keep_n = keep && nrm > 0.0;  // tweets_ann.sql:14  HAVING norm > 0.0
          }
        }
        seq[u] = keep_n ? seq[u] : -1;
        s32[u] = (float)scv;
        live += __popcll(__ballot(keep_n));  // wave count, identical in all lanes
        hsh[u] = table_hash(idv, HB);
      }
    }
    }
    // ---- 3a. blocked Bloom filter: four bits of one 64-bit word, one returning LDS atomic per posting.  All of a
    // thread's atomics are issued before the first result is looked at (one LDS round trip, not U).
    unsigned long long seen[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      seen[u] = 0ull;
      if ((uint32_t)(u * WG) < Tg && seq[u] >= 0) {
        const uint32_t hv = hsh[u];
        const unsigned long long bits = bloom_bits(hv);
        seen[u] = ~atomicOr(&s_bloom[hv >> BLOOM_POS_BITS], bits) & bits;  // bits of this posting that were NOT set before
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if ((uint32_t)(u * WG) < Tg && seq[u] >= 0 && seen[u] == 0ull) {
        // possibly seen before: mark the id's bits in the (sparse) "flagged" filter
        const uint32_t hv = hsh[u];
        const unsigned long long bits = bloom_bits(hv);
        atomicOr(&s_fbloom[hv >> (HB - FB)], bits);
        s_ctl[CTL_NFLAG] = 1;
      }
    }
    if ((tid & 63) == 0 && live) atomicAdd(&s_ctl[CTL_LIVE], live);
  }
  __syncthreads();
  STAMP(3);  // postings arrived, filtered, bloom done
  {
    unsigned long long x_ = 0;
    if constexpr (ABL == 1) {
#pragma unroll
      for (int u = 0; u < U; u++) x_ ^= (unsigned long long)__float_as_uint(s32[u]) ^ (unsigned)seq[u];
    }
    ABLATE(1, x_ + (unsigned)s_ctl[CTL_NFLAG] + (unsigned)s_ctl[CTL_LIVE]);
  }

  // ---- 3b. resolve flagged ids ------------------------------------------------------------------
  // Every posting whose bits are all set in the flagged filter (the flagged posting itself, the
  // earlier postings of the same tweet, and a few hash collisions) joins the match list M; one
  // thread per M entry then settles its group by comparing ids inside M only.
  if (!overflow && s_ctl[CTL_NFLAG] != 0) {
    // a flagged unit (one in five at the benchmark's shape: tweets met in two of the scanned clusters) looks its
    // postings' hashes up in the flagged filter; only the few that match fetch their posting again (from L2) -- every
    // unit carrying the 16-byte postings through the whole kernel cost more registers than the kernel has.  (Until
    // round 2 ALL postings of a flagged unit were fetched and hashed again: 40 % of the flagged unit's extra time.)
    int mi[U];
    Posting pm[U];
#pragma unroll
    for (int u = 0; u < U; u++) {  // (all of a thread's fetches in one trip)
      const uint32_t hv = hsh[u];
      const unsigned long long bits = bloom_bits(hv);
      const bool hit = seq[u] >= 0 && (s_fbloom[hv >> (HB - FB)] & bits) == bits;
      mi[u] = hit ? 0 : -1;
      pm[u] = Posting{0, 0.0};
      if (hit) {
        const int c = seq[u];
        pm[u] = ix.postings[s_begin[c] + ((uint32_t)(u * WG + tid) - s_pre[c])];
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (mi[u] >= 0) {
        const int c = seq[u];
        const int m = atomicAdd(&s_ctl[CTL_NM], 1);
        mi[u] = m;
        if (m < MCAP) {
          s_Mid[m] = pm[u].id;
          s_Mseq[m] = c;
          s_Msc[m] = pm[u].score;
          if constexpr (NORMS) s_Mnrm[m] = use_norms ? ix.norms[s_begin[c] + ((uint32_t)(u * WG + tid) - s_pre[c])] : 0.0;
        }
      }
    }
    __syncthreads();
    STAMP(9);  // (flagged units only) matches fetched, match list written
    const int nm = s_ctl[CTL_NM];
    if (nm > MCAP) {
      overflow = true;
      if (tid == 0) s_ctl[CTL_BAD] = 8;  // (diagnostics: the match list overflowed)
    } else if (nm > 12 && nm <= 64) {
      // A duplicate-heavy corpus (1M tweets under 144k clusters: every unit, 40-60 entries): the entries are SORTED by
      // (tweet id, cluster sequence) in wave 0's registers -- groups become runs of neighbouring lanes, a run's first
      // lane is its representative and walks the run for its sums (ascending cluster sequence: the reference's order).
      // The pairwise comparison below costs nm^2 dependent LDS reads: 18 k of such a unit's 40 k cycles.
      if (tid < 64) {
        const bool have = tid < nm;
        uint64_t kh = have ? id_key(s_Mid[tid]) : 0ull;  // descending id_key = ascending id
        uint64_t kl = have ? ~(uint64_t)(((uint32_t)s_Mseq[tid] << 6) | (uint32_t)tid) : 0ull;  // (never 0 for an entry)
        wave_sort_desc_k128(kh, kl);
        const bool live = (kh | kl) != 0ull;
        const uint32_t pk = (uint32_t)~kl;  // cluster sequence << 6 | entry
        const int ee = (int)(pk & 63u);
        const uint64_t prev = __shfl_up((unsigned long long)kh, 1, 64);
        const bool start = live && (tid == 0 || prev != kh);
        const unsigned long long sm = __ballot(start);
        const int n_ent = __popcll(__ballot(live));
        const unsigned long long above = tid == 63 ? 0ull : (sm >> (tid + 1));
        const int len = start ? (above != 0ull ? __ffsll((long long)above) : n_ent - tid) : 0;
        const int role = !live ? 0 : (start ? (len >= 2 ? 1 : 0) : 2);
        const int max_len = (int)wave_max_u32((uint32_t)len);
        double dot = 0.0, nsq = 0.0;
        for (int j = 0; j < max_len; j++) {  // (uniform)
          const uint32_t pj = (uint32_t)__shfl((int)pk, tid + j < 64 ? tid + j : 63, 64);
          if (role == 1 && j < len) {
            const double bs = s_Msc[pj & 63u];
            dot = dot + bs * s_w[pj >> 6];  // :92-94
            nsq = nsq + bs * bs;            // :95-96
          }
        }
        if (live) {
          if (role == 1) {
            if constexpr (NORMS) nsq = use_norms ? s_Mnrm[ee] : nsq;  // one norm per tweet, not a sum over clusters
            s_Mdot[ee] = dot;
            s_Mnsq[ee] = nsq;
          }
          s_Mrole[ee] = role;
        }
      }
      __syncthreads();
      STAMP(10);  // groups settled
      int folded = 0;
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int role = mi[u] >= 0 ? s_Mrole[mi[u]] : 0;
        seq[u] = role == 1 ? (0x10000 | mi[u]) : (role == 2 ? -1 : seq[u]);
        folded += __popcll(__ballot(role == 2));
      }
      if ((tid & 63) == 0 && folded) {
        atomicSub(&s_ctl[CTL_LIVE], folded);
        atomicAdd(&s_ctl[CTL_FOLD], folded);
      }
      __syncthreads();
    } else {
      {
        // one thread per match-list entry, ids compared inside M (a handful of entries on a large corpus, or more than a
        // wave holds)
        for (int m = tid; m < nm; m += WG) {
          const long long my = s_Mid[m];
          const int myseq = s_Mseq[m];
          int cnt = 0, rep = myseq;
#pragma unroll 4
          for (int e2 = 0; e2 < nm; e2++) {  // (unrolled: four LDS round trips in flight, not one after the other)
            const bool same = s_Mid[e2] == my;
            const int se = s_Mseq[e2];
            cnt += same ? 1 : 0;
            rep = (same && se < rep) ? se : rep;
          }
          int role = 0;
          if (cnt >= 2) {
            role = 2;
            if (myseq == rep) {
              double dot = 0.0, nsq = 0.0;
              int last = -1;
              for (int rr = 0; rr < cnt; rr++) {  // ascending cluster sequence
                int best = 0x7fffffff;
                double bs = 0.0;
#pragma unroll 4
                for (int e2 = 0; e2 < nm; e2++) {
                  const int se = s_Mseq[e2];
                  if (s_Mid[e2] == my && se > last && se < best) { best = se; bs = s_Msc[e2]; }
                }
                dot = dot + bs * s_w[best];  // :92-94
                nsq = nsq + bs * bs;         // :95-96
                last = best;
              }
              if constexpr (NORMS) nsq = use_norms ? s_Mnrm[m] : nsq;  // one norm per tweet, not a sum over clusters
              s_Mdot[m] = dot;
              s_Mnsq[m] = nsq;
              role = 1;
            }
          }
          s_Mrole[m] = role;
        }
      }
      __syncthreads();
      STAMP(10);  // groups settled
      int folded = 0;
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int role = mi[u] >= 0 ? s_Mrole[mi[u]] : 0;
        // representative: carries the group's sums (bit 16 + its M index); the others fold into it
        seq[u] = role == 1 ? (0x10000 | mi[u]) : (role == 2 ? -1 : seq[u]);
        folded += __popcll(__ballot(role == 2));
      }
      if ((tid & 63) == 0 && folded) {
        atomicSub(&s_ctl[CTL_LIVE], folded);
        atomicAdd(&s_ctl[CTL_FOLD], folded);
      }
      __syncthreads();
    }
  }

  STAMP(4);  // duplicates resolved
  {
    unsigned long long x_ = 0;
    if constexpr (ABL == 2) {
#pragma unroll
      for (int u = 0; u < U; u++) x_ ^= (unsigned long long)__float_as_uint(s32[u]) ^ (unsigned)seq[u];
    }
    ABLATE(2, x_ + (unsigned)s_ctl[CTL_LIVE]);
  }
  // ---- 4. approximate fp32 keys ---------------------------------------------------------------------------
  // One uniform branch per algorithm around the slot loop (not a switch per slot).  Cosine and the no-source-norm
  // form need no arithmetic at all for a single-cluster candidate: (s w) / sqrt(s s) = w for s > 0, so the key is a
  // constant of the cluster (s_wkey, filled with the descriptors).  Representatives of multi-cluster tweets (rare)
  // are re-keyed afterwards by the general formula.
  uint32_t k32[U];
  {
    bool bad = false;
    if (overflow) {
#pragma unroll
      for (int u = 0; u < U; u++) k32[u] = 0u;
    } else if (use_norms) {
      // offline forms: dot / LN(1 + norm) (as alg 3 with logNorm = 1) and dot / SQRT(norm) (as alg 4), norm = the column
      if constexpr (NORMS) {
#pragma unroll
        for (int u = 0; u < U; u++) {
          const bool lv = seq[u] >= 0;
          const float n32 = nrm32[u];
          bool forced;
          // (below 1e-6 the exact form's rounding of 1 + norm matters: such units take the general path)
          const float a = approx_score(h.alg, s32[u] * s_w32[lv ? (seq[u] & (NS - 1)) : 0], n32, 1.0, 1.f, 1.f, &forced);
          bad = bad || (lv && !(seq[u] & 0x10000) && !(a > 1e-30f && a < 1e30f && n32 >= 1e-6f && n32 < 1e30f));
          k32[u] = lv ? (__float_as_uint(a) | 0x80000000u) : 0u;
        }
      }
    } else if (h.alg == 2 || h.alg == 4) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const bool lv = seq[u] >= 0;
        const uint32_t wk = s_wkey[seq[u] & (NS - 1)];  // (dead slots and representatives read some entry: unused)
        // the shortcut is only trusted for ordinary positive magnitudes (s32^2 within the fp32 range, w / l2norm too):
        // 1e-15 < s32 < 1e15 as one unsigned compare of the bit pattern (negatives, NaN and inf fall outside)
        constexpr uint32_t LO = 0x26901d7du, HI = 0x58635fa9u;  // 1e-15f, 1e15f
        const bool ordinary = (__float_as_uint(s32[u]) - (LO + 1u)) < (HI - LO - 1u);
        bad = bad || ((uint32_t)seq[u] < 0x10000u && !(ordinary && wk != 0u));
        k32[u] = lv ? wk : 0u;
      }
    } else if (h.alg == 1) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const bool lv = seq[u] >= 0;
        const float a = s32[u] * s_w32[lv ? (seq[u] & (NS - 1)) : 0];
        bad = bad || (lv && !(seq[u] & 0x10000) && !(a > 1e-30f && a < 1e30f));
        k32[u] = lv ? (__float_as_uint(a) | 0x80000000u) : 0u;
      }
    } else {
      const float invln = h.inv_ln_32;
#pragma unroll
      for (int u = 0; u < U; u++) {
        const bool lv = seq[u] >= 0;
        const float sv = s32[u];
        float a = 0.f;
        bool forced = false;
        if (lv && !(seq[u] & 0x10000)) {
          // below 1e-6 the exact form's rounding of 1 + nsq decides the score: that needs the fp64 score, fetched again
          double nsq64 = 0.0;
          if (sv * sv < 1e-6f) {
            const int c = seq[u];
            const double sd = ix.postings[s_begin[c] + ((uint32_t)(u * WG + tid) - s_pre[c])].score;
            nsq64 = sd * sd;
          }
          a = approx_score(3, sv * s_w32[seq[u] & (NS - 1)], sv * sv, nsq64, 0.f, invln, &forced);
        }
        bad = bad || (lv && !(seq[u] & 0x10000) && !forced && !(a > 1e-30f && a < 1e30f && sv > 1e-15f && sv < 1e15f));
        k32[u] = lv ? (forced ? FORCED_KEY : (__float_as_uint(a) | 0x80000000u)) : 0u;
      }
    }
    if (!overflow && s_ctl[CTL_NFLAG] != 0) {  // uniform: only units that resolved duplicates can hold representatives
      const float invl2 = h.inv_l2_32, invln = h.inv_ln_32;
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (seq[u] >= 0 && (seq[u] & 0x10000)) {
          const double nsq64 = s_Mnsq[seq[u] & 0xffff];
          const float d32 = (float)s_Mdot[seq[u] & 0xffff], n32 = (float)nsq64;
          bool forced;
          const float a = approx_score(h.alg, d32, n32, nsq64, invl2, invln, &forced);
          bad = bad || (!forced && !(a > 1e-30f && a < 1e30f && n32 > 1e-30f && n32 < 1e30f)) || (use_norms && (forced || n32 < 1e-6f));
          k32[u] = forced ? FORCED_KEY : (__float_as_uint(a) | 0x80000000u);
        }
      }
    }
    if (__ballot(bad) != 0ull && (tid & 63) == 0) atomicOr(&s_ctl[CTL_BAD], 1);
  }
  // ---- 5a. the cut: the kl-th largest LANE MAXIMUM ---------------------------------------------------------------
  // kl distinct candidates reach the kl-th largest of the 256 per-thread maxima, so cutting there keeps at least kl;
  // the few candidates that share a thread with a larger one come on top (about U * kl^2 / n_live: 48 in all at the
  // benchmark's shape), far below the SCAP the survivor list holds.  Each wave sorts its 64 maxima in registers, the
  // four sorted runs meet in LDS, and every thread ranks its own value by three 6-step searches: two barriers, no
  // LDS atomics (the radix histogram this replaces spent 25 % of the kernel serialising atomics on the few distinct
  // keys a near-tie batch has).  The cut goes 256 fp32 ulps (>= 3 EPS) BELOW the value found: everything below the
  // cut is then bounded by theta = tau (1 + 2 EPS) <= m (1 - EPS), which the candidates that tie with m (in the
  // near-tie regime: the whole cluster group the cut falls into) still reach with their exact scores -- a cut exactly
  // at m emitted none of them, and the merge could not prove the query.
  // Cosine forms take the cluster-level cut of 5a' instead and come here only if it kept too few (window / source
  // filters thinned the clusters it counted on).
  uint32_t *const s_lm = s_hist;  // [WG], in the dead Bloom filter's memory (or s_hist_own, which is WG <= 256 words)
  uint32_t tau = 0;  // survivors: k32 >= tau
  bool cut_by_data = !pre_cut;  // uniform
  if (pre_cut && s_ctl[CTL_LIVE] > keep_all) tau = pre_tau;
  for (;;) {
    if (cut_by_data) {
      const bool select = s_ctl[CTL_LIVE] > keep_all && !overflow;  // uniform
      if (select) {
        uint32_t m = 0u;
#pragma unroll
        for (int u = 0; u < U; u++) m = k32[u] > m ? k32[u] : m;
        s_lm[tid] = wave_sort_desc_u32(m);
      }
      __syncthreads();
      STAMP(5);  // approximate keys, lane maxima sorted
      {
        unsigned long long x_ = 0;
        if constexpr (ABL == 3) {
#pragma unroll
          for (int u = 0; u < U; u++) x_ ^= (unsigned long long)__float_as_uint(s32[u]) ^ (unsigned)seq[u] ^ k32[u];
        }
        ABLATE(3, x_ + s_lm[tid]);
      }
      tau = 0u;
      if (select) {
        const int lane = tid & 63, wv = tid >> 6;
        const uint32_t m = s_lm[tid];
        int rank = lane;  // position in the total order (value desc, wave asc, lane asc)
#pragma unroll
        for (int w2 = 0; w2 < WG / 64; w2++) {
          if (w2 == wv) continue;
          const uint32_t *L = s_lm + w2 * 64;
          int pos = 0;  // entries of wave w2's descending run that come before mine
#pragma unroll
          for (int step = 32; step >= 1; step >>= 1) {
            const uint32_t v = L[pos + step - 1];
            const bool before = v > m || (v == m && w2 < wv);
            pos += before ? step : 0;
          }
          {  // 64 entries = 63 reachable by the steps above, plus the last one
            const uint32_t v = L[63];
            pos += (pos == 63 && (v > m || (v == m && w2 < wv))) ? 1 : 0;
          }
          rank += pos;
        }
        // ranks are a permutation: exactly one thread writes (none when kl > WG: tau stays 0)
        if (rank == kl - 1) s_ctl[CTL_SEL_D] = (int)(m > 0x80000100u ? m - 256u : m);
        __syncthreads();
        tau = (uint32_t)s_ctl[CTL_SEL_D];
      }
    } else {
      STAMP(5);  // (cluster-level cut: the threshold was known before the keys)
    }
    STAMP(6);  // threshold found
    {
      unsigned long long x_ = 0;
      if constexpr (ABL == 4) {
#pragma unroll
        for (int u = 0; u < U; u++) x_ ^= (unsigned long long)__float_as_uint(s32[u]) ^ (unsigned)seq[u] ^ k32[u];
      }
      ABLATE(4, x_ + tau);
    }
    // ---- 5b. compact the survivors: one LDS atomic per wave --------------------------------------------------------
    // Survivors are few (tens per unit), so the bookkeeping is per THREAD, not per slot: a thread counts its own, a DPP
    // prefix sum places it in the wave, one atomic places the wave in the unit, and an entry is just (cluster, flat
    // posting index) -- the posting's address is worked out in phase 6, by the few threads that need it.  (Six rounds of
    // ballot / mbcnt / descriptor reads here were 200 of the kernel's 790 VALU instructions per wave.)
    {
      const uint32_t tau_eff = tau ? tau : 1u;  // (a dead slot's key is 0)
      int cnt = 0;
#pragma unroll
      for (int u = 0; u < U; u++) cnt += k32[u] >= tau_eff ? 1 : 0;
      const int incl = wave_incl_scan_i32(cnt);
      const int total = __builtin_amdgcn_readlane(incl, 63);
      int base = 0;
      if ((tid & 63) == 0 && total) base = atomicAdd(&s_ctl[CTL_NSURV], total);
      base = __builtin_amdgcn_readfirstlane(base);
      {
        unsigned long long x_ = 0;
        if constexpr (ABL == 8) {
#pragma unroll
          for (int u = 0; u < U; u++) x_ ^= (unsigned long long)__float_as_uint(s32[u]) ^ (unsigned)seq[u];
        }
        ABLATE(8, x_ + (unsigned)(incl + base + total));
      }
      if (total != 0 && base + total <= SCAP) {  // uniform per wave; a list that would not fit is an overflow below
        int o = base + incl - cnt;
#pragma unroll
        for (int u = 0; u < U; u++) {
          if (k32[u] >= tau_eff) {
            s_ent[o] = ((unsigned long long)(uint32_t)seq[u] << 32) | (uint32_t)(u * WG + tid);
            o++;
          }
        }
      }
    }
    __syncthreads();
    {
      unsigned long long x_ = 0;
      if constexpr (ABL == 9) {
#pragma unroll
        for (int u = 0; u < U; u++) x_ ^= (unsigned long long)__float_as_uint(s32[u]);
      }
      ABLATE(9, x_ + s_ent[tid & 63] + (unsigned)s_ctl[CTL_NSURV]);
    }
    // the cluster-level cut counted postings the filters then removed: cut by the data after all.  (With nothing
    // removed -- live + folded = T -- the count behind the cut is exact; it may then be below kl on purpose: the
    // descriptor kernel's query-level rule.)
    if (!cut_by_data && tau != 0u && s_ctl[CTL_NSURV] < kl && s_ctl[CTL_NSURV] < s_ctl[CTL_LIVE] &&
        s_ctl[CTL_LIVE] + s_ctl[CTL_FOLD] < (int)T && !s_ctl[CTL_BAD] && !overflow) {
      __syncthreads();  // (everyone has read the counters)
      if (tid == 0) s_ctl[CTL_NSURV] = 0;
      cut_by_data = true;
      continue;  // (the barrier after the lane maxima orders the reset before the next compaction)
    }
    break;
  }
  if (s_ctl[CTL_BAD] && !overflow) overflow = true;
  if (overflow) {
    if (tid == 0) unit_overflowed(unit, overflow_n ? 1ull : overflow_T ? 2ull : (s_ctl[CTL_BAD] & 8) ? 3ull : 4ull);
    return;
  }
  // A cut that keeps more than the survivor list holds (many candidates sharing threads with larger ones, or a tie
  // group of more than ~100 identical keys -- which no finer cut could split either) sends the unit to the general
  // path.  (Round 1 re-cut such units with a radix histogram over all keys; it fired too rarely to earn its registers.)
  if (s_ctl[CTL_NSURV] > SCAP) {
    if (tid == 0) unit_overflowed(unit, 5ull);
    return;
  }
  const int ns = s_ctl[CTL_NSURV] < SCAP ? s_ctl[CTL_NSURV] : SCAP;
  // theta: every candidate below the cut has approx < tau, hence exact < tau * (1 + 2 EPS)
  unsigned long long theta_key = 0ull;
  // ... unless every live candidate survived the cut (tau can be non-zero and still below all of them when the
  // need-th key sits in the lowest occupied digit): then nothing is withheld and nothing may be dropped below.
  if (tau != 0u && s_ctl[CTL_NSURV] < s_ctl[CTL_LIVE]) {
    double tau_val = (double)__uint_as_float(tau & 0x7fffffffu);
    tau_val = tau_val < 1e30 ? tau_val : 1e30;  // a cut inside the forced (+inf) band: the others are still < 1e30
    theta_key = score_key(tau_val * (1.0 + 2.0 * (double)APPROX_EPS));
  }

  STAMP(7);  // survivors compacted
  ABLATE(5, (unsigned long long)ns + theta_key + s_ent[tid % SCAP]);
  // ---- 6. hand the survivors over ----------------------------------------------------------------------------------
  // A survivor's exact fp64 score (a second fetch of its posting, a division and a square root: a chain of ~190 dependent
  // instructions in ONE wave of the workgroup while the other three waited at the barrier -- 15 % of the kernel) is
  // computed by the merge kernel, where 512 threads share a query's ~1400 survivors.  What leaves here is
  // (cluster sequence number, posting position); only representatives of multi-cluster tweets, whose sums live in this
  // workgroup's LDS, are finished here.  Nothing is dropped below theta any more: a candidate under theta can only
  // reach the top k of a query whose proof fails anyway (theta > k-th key).
  const int64_t obase = (int64_t)unit * LATE_ARG(cap);
  uint64_t *const cand_key = LATE_ARG(cand_key);
  int64_t *const cand_id = LATE_ARG(cand_id);
  int i_first = tid;
  asm volatile("" : "+v"(i_first));  // (a fresh index: hipcc otherwise keeps tid * 8 from the first lines alive, in scratch)
  for (int i = i_first; i < ns; i += WG) {
    const unsigned long long e = s_ent[i];
    const int c = (int)(e >> 32);
    unsigned long long key = CAND_DEFERRED;
    long long idv;
    if (c & 0x10000) {
      idv = s_Mid[c & 0xffff];
      // (the query's norms and minScore are read here, by the few lanes that need them, not kept in six SGPRs all along)
      const QueryHdr *hq = LATE_ARG(hdr) + q;
      const double l2 = *(const volatile double *)&hq->l2norm, ln = *(const volatile double *)&hq->lognorm;
      const double v = normalise_f(h.alg, s_Mdot[c & 0xffff], s_Mnsq[c & 0xffff], l2, ln);
      key = v >= *(const volatile double *)&hq->min_score ? score_key(v) : CAND_DROPPED;  // :125 (false for NaN)
    } else {
      const uint32_t pos = s_begin[c] + ((uint32_t)e - s_pre[c]);
      idv = (long long)(((unsigned long long)(uint32_t)c << 32) | pos);
    }
    cand_key[obase + i] = key;
    cand_id[obase + i] = idv;
  }
  if (tid == 0) {
    const int n_live = s_ctl[CTL_LIVE];
    const bool withheld = n_live > ns;  // candidates below the cut were not examined exactly
    LATE_ARG(cand_cnt)[unit] = ns;
    LATE_ARG(unit_unique)[unit] = n_live;
    LATE_ARG(unit_flags)[unit] = withheld ? UNIT_TRUNCATED : UNIT_OK;
    uint64_t *const thr = LATE_ARG(unit_thr);
    thr[2 * (int64_t)unit] = withheld ? theta_key : 0ull;
    thr[2 * (int64_t)unit + 1] = 0;
  }
  STAMP(8);  // emitted
#undef STAMP
}

template <int WG, int U>
static hipError_t launch_one(const IndexView &ix, const BatchView &b, const FastParams &fp, hipStream_t stream) {
  const int nq8 = (b.nq + 7) / 8 * 8;
  const int n_blocks = nq8 * ix.P;
  if (b.desc_stride != desc_row_stride(fp.max_n_scan)) return hipErrorInvalidValue;  // (the kernel's NS is the row stride)
  if (fp.use_norms) {
    if (fp.max_n_scan <= 64)
      hipLaunchKernelGGL((unit_fast_kernel<WG, U, 64, 0, true>), dim3(n_blocks), dim3(WG), 0, stream, ix, b, fp.k_local, n_blocks);
    else
      hipLaunchKernelGGL((unit_fast_kernel<WG, U, NSCAN_MAX, 0, true>), dim3(n_blocks), dim3(WG), 0, stream, ix, b, fp.k_local, n_blocks);
  } else if (fp.max_n_scan <= 64)
    hipLaunchKernelGGL((unit_fast_kernel<WG, U, 64>), dim3(n_blocks), dim3(WG), 0, stream, ix, b, fp.k_local, n_blocks);
  else
    hipLaunchKernelGGL((unit_fast_kernel<WG, U, NSCAN_MAX>), dim3(n_blocks), dim3(WG), 0, stream, ix, b, fp.k_local, n_blocks);
  return hipGetLastError();
}

// The same descriptors, one WORKGROUP per query (P <= 32): the n_scan x P sub-lists of a query are resolved by 256
// threads with all their loads in flight at once and partition-contiguous (coalesced) reads of the offset and cut
// tables; the per-partition prefix over the clusters runs in LDS, and the rows are written out coalesced.  One
// round of 1024 workgroups instead of four rounds of one-wave units, each a chain of three dependent trips to memory.
constexpr int DESC_Q_ITEMS = 4096;  // NSCAN_MAX x 32
// QPW = queries per workgroup: 1 (all 256 threads on one query: P >= 16, up to 4096 sub-lists) or 4 (one WAVE per
// query and no workgroup barrier: a shard's queries have 4 or 8 partitions, 200-400 sub-lists each, and N times as many
// of them -- 8192 workgroups of mostly idle threads took 30 us).
template <int QPW>
__global__ __launch_bounds__(256) void desc_query_kernel(IndexView ix, BatchView b, int ld, int k_local_floor) {
  constexpr int G = 256 / QPW;  // threads per query
  constexpr int WPG = G / 64;   // waves per query
  // s_base / s_len [p * ld + c]; s_len becomes the exclusive prefix.  Sized by the launch for the batch's largest query
  // (13 KB at 50 clusters x 32 partitions): small enough to find room on a CU that is full of unit-kernel workgroups.
  // (ld odd: the fill walks p, everything after it walks c -- both free of bank conflicts)
  extern __shared__ uint32_t s_desc[];
  const int grp = threadIdx.x / G, tid = threadIdx.x % G;  // (tid: within the query's threads)
  const int P = ix.P;
  uint32_t *const s_base = s_desc + grp * 2 * P * ld, *const s_len = s_base + P * ld;
  const int q = blockIdx.x * QPW + grp;
  {
    const int g = blockIdx.x * 256 + threadIdx.x;  // per-run state (see desc_kernel)
    if (g < b.nq + 2) b.status[g] = 0;
  }
  if (q >= b.nq) return;  // (QPW > 1: whole waves, and nothing below synchronises across a query's threads' workgroup)
  // a query's threads meet: the workgroup's barrier, or -- one wave per query -- just the order of the wave's own LDS traffic
  auto meet = [&]() {
    if constexpr (QPW == 1) __syncthreads();
    else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  };
  const int M = b.hdr[q].M;
  const int scan_begin = b.hdr[q].scan_begin;
  const int n_scan = b.hdr[q].n_scan;
  for (int p = tid; p < P; p += G) b.unit_fb[(int64_t)q * P + p] = -1;
  if (n_scan > NSCAN_MAX) {  // the unit kernel sends such units to the general path
    for (int p = tid; p < P; p += G) {
      b.unit_T[(int64_t)q * P + p] = 0;
      b.unit_pre[(int64_t)q * P + p] = 0u;
    }
    return;
  }
  // cluster-level cut: the order of the query's clusters by key is the same for all of its units -- sorted once, by
  // wave 0 (lane c = cluster c), while the other waves already chase the sub-list descriptors
  __shared__ uint32_t s_okey_all[QPW][64];  // sorted position i: the cluster's key ...
  __shared__ uint8_t s_ocl_all[QPW][64];    // ... and the cluster
  uint32_t *const s_okey = s_okey_all[grp];
  uint8_t *const s_ocl = s_ocl_all[grp];
  const QueryHdr h = b.hdr[q];
  const bool cluster_cut = query_has_cluster_cut(h);  // (uniform)
  if (cluster_cut && tid < 64) {
    const float inv_l2_32 = h.inv_l2_32;
    const uint32_t kc = tid < n_scan ? cosine_cluster_key(h.alg, b.scan_w[scan_begin + tid], inv_l2_32) : 0u;
    const uint32_t pk = wave_sort_desc_u32(kc ? ((kc & ~0xffu) | (uint32_t)tid) : 0u);
    const int c = (int)(pk & 0xffu);
    s_ocl[tid] = (uint8_t)c;
    s_okey[tid] = pk ? __shfl(kc, c, 64) : 0u;
  }
  const uint32_t *cut = nullptr;  // cached cut table for this query's M, if any (uniform)
#pragma unroll
  for (int j = 0; j < 4; j++)
    if (b.cut_M[j] == M) cut = b.cut[j];
  const int n_items = n_scan * P;
  for (int i = tid; i < n_items; i += G) {
    const int c = i >> ix.log2P, p = i & (P - 1);
    const int row = b.scan_row[scan_begin + c];
    const uint32_t base = ix.sub_offsets[(int64_t)row * P + p];
    uint32_t len;
    if (cut) {
      len = cut[(int64_t)row * P + p];
    } else {
      const int n = (int)(ix.sub_offsets[(int64_t)row * P + p + 1] - base);
      // postings with rank < M are a prefix of the sub-list
      len = (n > 0 && ix.ranks[base + n - 1] < (uint32_t)M) ? (uint32_t)n
                                                          : (uint32_t)lower_bound_rank(ix.ranks + base, n, (uint32_t)M);
    }
    s_base[p * ld + c] = base;
    s_len[p * ld + c] = len;
  }
  meet();
  // The query-level rule: clusters in key order until k plus a margin postings are covered in ALL the query's partitions
  // together.  A cluster's single-cluster candidates share one key, so a cut below that cluster keeps every candidate
  // that can reach the top k whichever partitions they fell into -- no per-unit allowance for the spread (unit_kl: the
  // share plus five sigma) is needed, and a query hands ~800 candidates to its merge instead of ~1500.  A unit takes the
  // stricter of the two cuts.  (Postings that filters or duplicates then remove are what the margin, the unit kernel's
  // data-dependent re-cut and, in the end, the merge's proof are for.)
  __shared__ uint32_t s_qpre_all[QPW];
  if (cluster_cut && tid < 64) {
    int tot = 0;
    const uint32_t kc = s_okey[tid];
    if (kc != 0u)
      for (int p = 0; p < P; p++) tot += (int)s_len[p * ld + (int)s_ocl[tid]];
    const int cum = wave_incl_scan_i32(tot);
    const int target = h.k + (h.k / 4 > 32 ? h.k / 4 : 32);
    const unsigned long long ok = __ballot(kc != 0u && cum >= target);
    if (tid == 0) s_qpre_all[grp] = ok != 0ull ? cluster_cut_from_key(s_okey[__ffsll((long long)ok) - 1]) : 0u;
  }
  if (cluster_cut) meet();  // (uniform)
  // Per partition: the unit's cut (clusters in key order until kl postings are covered) and the exclusive prefix of the
  // lengths over the clusters.  One WAVE per partition at a time -- lane = cluster, a DPP prefix sum -- instead of one
  // thread walking its partition's 50 clusters through dependent LDS reads.  A wave owns the same partitions in both
  // passes, so the second pass may overwrite what the first one read.
  const int wv = tid >> 6, lane = tid & 63;
  if (cluster_cut) {
    const int kl = unit_kl(h.k, P, k_local_floor);
    const uint32_t qpre = s_qpre_all[grp];
    const uint32_t kc = s_okey[lane];  // (lane = position in key order; keys of 0 sort last: no trusted cluster is left)
    const int oc = (int)s_ocl[lane];
    for (int p = wv; p < P; p += WPG) {  // (uniform per wave)
      const int cum = wave_incl_scan_i32(kc != 0u ? (int)s_len[p * ld + oc] : 0);
      const unsigned long long ok = __ballot(kc != 0u && cum >= kl);
      uint32_t pre = 0u;
      if (ok != 0ull) pre = cluster_cut_from_key((uint32_t)__builtin_amdgcn_readlane((int)kc, __ffsll((long long)ok) - 1));
      if (lane == 0) b.unit_pre[(int64_t)q * P + p] = pre > qpre ? pre : qpre;
    }
  } else {
    for (int p = tid; p < P; p += G) b.unit_pre[(int64_t)q * P + p] = 0u;
  }
  __shared__ uint32_t s_T_all[QPW][DESC_Q_ITEMS / NSCAN_MAX];  // (P <= 32 here)
  uint32_t *const s_T = s_T_all[grp];
  for (int p = wv; p < P; p += WPG) {  // exclusive prefix over the clusters (n_scan <= 128: two per lane), per partition
    const int l0 = lane < n_scan ? (int)s_len[p * ld + lane] : 0;
    const int l1 = lane + 64 < n_scan ? (int)s_len[p * ld + lane + 64] : 0;
    const int i0 = wave_incl_scan_i32(l0);
    const int t0 = __builtin_amdgcn_readlane(i0, 63);
    const int i1 = wave_incl_scan_i32(l1);
    const int t1 = __builtin_amdgcn_readlane(i1, 63);
    if (lane < n_scan) s_len[p * ld + lane] = (uint32_t)(i0 - l0);
    if (lane + 64 < n_scan) s_len[p * ld + lane + 64] = (uint32_t)(t0 + i1 - l1);
    if (lane == 0) {
      b.unit_T[(int64_t)q * P + p] = t0 + t1;
      s_T[p] = (uint32_t)(t0 + t1);
    }
  }
  meet();
  // unit-major rows of desc_stride (start, prefix) pairs, padded behind n_scan with (0, T); the query's weights likewise
  const int stride = b.desc_stride, log2s = stride == 64 ? 6 : 7;
  uint2 *d = reinterpret_cast<uint2 *>(b.desc) + (int64_t)q * P * stride;
  for (int o = tid; o < P * stride; o += G) {
    const int p = o >> log2s, c = o & (stride - 1);
    d[o] = c < n_scan ? make_uint2(s_base[p * ld + c], s_len[p * ld + c]) : make_uint2(0u, s_T[p]);
  }
  for (int c = tid; c < stride; c += G) b.scan_wq[(int64_t)q * stride + c] = c < n_scan ? b.scan_w[scan_begin + c] : 0.0;
}

hipError_t launch_desc(const IndexView &ix, const BatchView &b, int n_units, int max_n_scan, int k_local_floor, hipStream_t stream) {
  if (n_units <= 0) return hipSuccess;
  const int ld = (max_n_scan < NSCAN_MAX ? (max_n_scan > 0 ? max_n_scan : 1) : NSCAN_MAX) | 1;  // (odd: see the kernel)
  // (one workgroup per query from 8 partitions up: the cluster-level cut's sort is then done once per query, not once per
  // unit -- 38 us against 54 for an 8-GPU shard's 65536 units)
  if (ix.P >= 4 && ix.P * NSCAN_MAX <= DESC_Q_ITEMS) {
    if (ix.P <= 8)  // four queries per workgroup, one wave each
      hipLaunchKernelGGL((desc_query_kernel<4>), dim3((unsigned)((b.nq + 3) / 4)), dim3(256), (size_t)4 * ix.P * ld * 8, stream, ix, b, ld, k_local_floor);
    else
      hipLaunchKernelGGL((desc_query_kernel<1>), dim3((unsigned)b.nq), dim3(256), (size_t)ix.P * ld * 8, stream, ix, b, ld, k_local_floor);
  }
  else
    hipLaunchKernelGGL(desc_kernel, dim3((unsigned)((n_units + 3) / 4)), dim3(256), 0, stream, ix, b, n_units, k_local_floor);
  return hipGetLastError();
}

// Audit hook: the pre-filter's approximate score of a single-cluster candidate (posting score s, cluster weight w),
// computed by the very function the unit kernel calls, with the unit kernel's fp32 conversions.
__global__ void debug_approx_kernel(int alg, int n, const double *s, const double *w, double l2norm, double lognorm,
                                    float *out, uint8_t *out_forced) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float invl2 = (float)(1.0 / l2norm), invln = (float)(1.0 / lognorm);
  const float s32 = (float)s[i];
  bool forced;
  out[i] = approx_score(alg, s32 * (float)w[i], s32 * s32, s[i] * s[i], invl2, invln, &forced);
  out_forced[i] = forced ? 1 : 0;
}
// Audit hook: every wave sorts its 64 values with the unit kernel's wave_sort_desc_u32
__global__ void debug_wave_sort_kernel(uint32_t *v) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  v[i] = wave_sort_desc_u32(v[i]);
}
hipError_t launch_debug_wave_sort(int n_waves, uint32_t *v, hipStream_t stream) {
  if (n_waves <= 0) return hipSuccess;
  hipLaunchKernelGGL(debug_wave_sort_kernel, dim3(n_waves), dim3(64), 0, stream, v);
  return hipGetLastError();
}

hipError_t launch_unit_ablation(const IndexView &ix, const BatchView &b, const FastParams &fp, int abl, hipStream_t stream) {
  const int nq8 = (b.nq + 7) / 8 * 8;
  const int n_blocks = nq8 * ix.P;
  if (fp.unit_capacity != 1536 || b.desc_stride != 64) return hipErrorInvalidValue;
  switch (abl) {
    case 0: hipLaunchKernelGGL((unit_fast_kernel<256, 6, 64, 0>), dim3(n_blocks), dim3(256), 0, stream, ix, b, fp.k_local, n_blocks); break;
    case 1: hipLaunchKernelGGL((unit_fast_kernel<256, 6, 64, 1>), dim3(n_blocks), dim3(256), 0, stream, ix, b, fp.k_local, n_blocks); break;
    case 2: hipLaunchKernelGGL((unit_fast_kernel<256, 6, 64, 2>), dim3(n_blocks), dim3(256), 0, stream, ix, b, fp.k_local, n_blocks); break;
    case 3: hipLaunchKernelGGL((unit_fast_kernel<256, 6, 64, 3>), dim3(n_blocks), dim3(256), 0, stream, ix, b, fp.k_local, n_blocks); break;
    case 4: hipLaunchKernelGGL((unit_fast_kernel<256, 6, 64, 4>), dim3(n_blocks), dim3(256), 0, stream, ix, b, fp.k_local, n_blocks); break;
    case 5: hipLaunchKernelGGL((unit_fast_kernel<256, 6, 64, 5>), dim3(n_blocks), dim3(256), 0, stream, ix, b, fp.k_local, n_blocks); break;
    case 6: hipLaunchKernelGGL((unit_fast_kernel<256, 6, 64, 6>), dim3(n_blocks), dim3(256), 0, stream, ix, b, fp.k_local, n_blocks); break;
    case 7: hipLaunchKernelGGL((unit_fast_kernel<256, 6, 64, 7>), dim3(n_blocks), dim3(256), 0, stream, ix, b, fp.k_local, n_blocks); break;
    case 8: hipLaunchKernelGGL((unit_fast_kernel<256, 6, 64, 8>), dim3(n_blocks), dim3(256), 0, stream, ix, b, fp.k_local, n_blocks); break;
    case 9: hipLaunchKernelGGL((unit_fast_kernel<256, 6, 64, 9>), dim3(n_blocks), dim3(256), 0, stream, ix, b, fp.k_local, n_blocks); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_debug_approx(int alg, int n, const double *s, const double *w, double l2norm, double lognorm, float *out,
                               uint8_t *out_forced, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(debug_approx_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, alg, n, s, w, l2norm, lognorm, out,
                     out_forced);
  return hipGetLastError();
}

// fp.unit_capacity = postings one unit may hold = WG * U
hipError_t launch_unit_fast(const IndexView &ix, const BatchView &b, const FastParams &fp, int n_units,
                            hipStream_t stream) {
  if (n_units <= 0) return hipSuccess;
  if (b.cap < FAST_SCAP) return hipErrorInvalidValue;
  switch (fp.unit_capacity) {
    case 256: return launch_one<64, 4>(ix, b, fp, stream);
    case 512: return launch_one<128, 4>(ix, b, fp, stream);
    case 768: return launch_one<256, 3>(ix, b, fp, stream);
    case 1024: return launch_one<256, 4>(ix, b, fp, stream);
    case 1536: return launch_one<256, 6>(ix, b, fp, stream);
    case 2048: return launch_one<256, 8>(ix, b, fp, stream);
    case 3072: return launch_one<256, 12>(ix, b, fp, stream);
    case 4096: return launch_one<256, 16>(ix, b, fp, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace sann
