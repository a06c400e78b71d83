// sann_pipe.hip -- the (query, partition) unit kernel, software-pipelined: PERSISTENT workgroups that keep the next unit's
// postings in flight while they work on the current one.
//
// Why (profiles/r02_phase_table.txt, profiles/r03_rotation_sweep.txt): the one-unit-per-workgroup
// kernel of sann_fast.hip is bound by the time a workgroup holds its slot -- a chain of two dependent memory trips
// (descriptor row -> postings) followed by LDS / VALU phases -- at eight workgroups per CU; its loads are in flight for
// only a quarter of a workgroup's life, so the CU's share of HBM bandwidth is idle most of the time.  Here a workgroup
// handles a SEQUENCE of units (block g takes units g, g + G, g + 2G, ...; G = grid size, a multiple of 8, so a
// workgroup's queries stay on one XCD) as a three-stage pipeline:
//
//   D(n+2)  descriptor row, cluster weights, posting count, cluster cut of unit n+2     7 loads per thread, into registers
//   P(n+1)  the 16-byte postings of unit n+1                                            U loads per thread, into registers
//   C(n)    unit n itself: filters, Bloom filter, duplicates, keys, cut, compaction, hand-over   (sann_fast.hip's phases 2-6)
//
// D and P are issued at the top of an iteration (inline asm: the compiler never sees a pending load) and are waited for
// where their registers are needed -- D in the middle of C(n), when the tables of unit n+2 are filled (three table
// buffers in LDS), P just before the hand-over stores -- with counted `s_waitcnt vmcnt`: D is issued BEFORE P, so
// vmcnt(U) means "D has arrived, P may still be in flight".  The loads of a unit are therefore in flight for a whole
// iteration, and a workgroup's critical path per unit is its LDS / VALU work alone.  Half as many workgroups are resident
// (four per CU, 128 registers each instead of 64), each with six times the bytes in flight per unit of its lifetime.
//
// Barriers per unit: one fewer than sann_fast.hip (the table fill of unit n+2 happens in front of the compaction
// barrier of unit n, which also publishes it; no barrier sits between a unit's descriptor data and its posting loads).
// LDS hazards between consecutive units are covered by the existing barriers: the Bloom filter is cleared right after
// the barrier that ends its use (it no longer shares memory with the survivor / match lists), the control words are
// double-buffered, the flagged filter is cleared by the (rare) units that wrote it.
//
// Results are sann_fast.hip's, entry for entry (same phases, same arithmetic); it serves the geometries with 256-thread
// units, <= 64 scanned clusters and no norms column.  Everything else stays with unit_fast_kernel.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "sann_device.h"
#include "sann_kernels.h"
#include "sann_math.h"
#include "sann_unit.h"
// lane number of the wave helpers: recomputed where it is used (two instructions the optimiser can neither hoist out of the
// kernel's loop nor merge), see sann_wave.h
__device__ inline int sann_pipe_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
#define SANN_WAVE_LANE() sann_pipe_lane()
#include "sann_wave.h"

namespace sann {

namespace {

template <int NS, int CAP>
struct UnitTables {
  uint32_t begin[NS];  // sub-list start of the cluster in ix.postings
  uint32_t pre[NS];    // exclusive prefix of the sub-list lengths = flat index of the cluster's first posting in the unit
  double w[NS];        // cluster weight of the query
  float w32[NS];
  uint32_t wkey[NS];   // cosine forms: fp32 key of a single-cluster candidate (0 = untrusted)
  uint8_t map[CAP];    // flat posting index -> cluster sequence number
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

// The query header of a unit, as SCALAR loads issued by hand one unit ahead (stage H).  Left to hipcc, `b.hdr[q]` inside the
// loop became vector loads with an `s_waitcnt vmcnt(0)` at the top of every iteration (the kernel's asm statements clobber
// memory, so the compiler will not use the scalar cache): a whole exposed trip to memory per unit.
struct HdrRegs {
  u32x8 a;     // bytes [0, 32): src_excl, earliest, latest, l2norm
  u32x8 b;     // bytes [48, 80): scan_begin, n_scan, M, k, alg, excl_enabled, use_norms, inv_l2_32
  uint32_t c;  // bytes [80, 84): inv_ln_32
};
static_assert(offsetof(QueryHdr, latest) == 16 && offsetof(QueryHdr, n_scan) == 52 && offsetof(QueryHdr, k) == 60 &&
                  offsetof(QueryHdr, alg) == 64 && offsetof(QueryHdr, excl_enabled) == 68 && offsetof(QueryHdr, use_norms) == 72 &&
                  offsetof(QueryHdr, inv_l2_32) == 76 && offsetof(QueryHdr, inv_ln_32) == 80,
              "HdrRegs follows QueryHdr");
// the fields the unit's phases read
struct UnitHdr {
  int64_t src_excl, earliest, latest;
  int32_t n_scan, k, alg, excl_enabled, use_norms;
  float inv_l2_32, inv_ln_32;
};
__device__ inline UnitHdr unit_hdr(const HdrRegs &H) {
  UnitHdr h;
  h.src_excl = (int64_t)(((uint64_t)H.a[1] << 32) | H.a[0]);
  h.earliest = (int64_t)(((uint64_t)H.a[3] << 32) | H.a[2]);
  h.latest = (int64_t)(((uint64_t)H.a[5] << 32) | H.a[4]);
  h.n_scan = (int32_t)H.b[1];
  h.k = (int32_t)H.b[3];
  h.alg = (int32_t)H.b[4];
  h.excl_enabled = (int32_t)H.b[5];
  h.use_norms = (int32_t)H.b[6];
  h.inv_l2_32 = __uint_as_float(H.b[7]);
  h.inv_ln_32 = __uint_as_float(H.c);
  return h;
}
__device__ inline bool unit_has_cluster_cut(const UnitHdr &h) {  // = query_has_cluster_cut
  return (h.alg == 2 || h.alg == 4) && h.n_scan <= 64 && h.use_norms == 0;
}

// what stage D loads per thread (thread t serves cluster t / 4, quarter t % 4 of its stretch of the map)
struct DescRegs {
  u32x2 v;        // (sub-list start, exclusive prefix) of cluster c
  uint32_t nxt;   // prefix of cluster c + 1
  u32x2 w;        // the query's weight of cluster c (double)
  uint32_t T;     // postings of the unit
  uint32_t pre;   // the unit's cluster-level cut (unit_pre)
};

// Kernel arguments that only the hand-over needs are read from the kernarg segment WHEN they are needed (a volatile load
// is not hoisted): hipcc otherwise loads every argument in the kernel's first lines and keeps the output arrays' pointers
// in SGPRs across the whole loop.  (Arguments: IndexView at 0, BatchView behind it -- as unit_fast_kernel.)
typedef const __attribute__((address_space(4))) char *kernarg_ptr;
constexpr size_t KERNARG_BATCH = (sizeof(IndexView) + alignof(BatchView) - 1) / alignof(BatchView) * alignof(BatchView);
template <class T> __device__ inline T late_kernarg(size_t off) {
  kernarg_ptr ka = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  return *(const volatile __attribute__((address_space(4))) T *)(ka + off);
}
#define LATE_ARG(field) late_kernarg<decltype(BatchView::field)>(KERNARG_BATCH + offsetof(BatchView, field))

__device__ inline void pipe_unit_overflowed(int unit, unsigned long long reason) {
  asm volatile("" : "+v"(reason));  // (built here: as a constant the (0, reason) pair is materialised in front of the kernel's loop -- and spilled)
  LATE_ARG(cand_cnt)[unit] = 0;
  LATE_ARG(unit_unique)[unit] = 0;
  LATE_ARG(unit_flags)[unit] = UNIT_OVERFLOW;
  uint64_t *const thr = LATE_ARG(unit_thr);
  thr[2 * (int64_t)unit] = 0;
  thr[2 * (int64_t)unit + 1] = reason;
  const int o = atomicAdd(&LATE_ARG(status)[0], 1);
  LATE_ARG(overflow_units)[o] = unit;
}


// ---- registers that hold loads IN FLIGHT are not the compiler's ------------------------------------------------------------
// Stage D's seven results and stage P's U postings are in flight across most of an iteration.  As ordinary asm outputs
// they were the compiler's to move: hipcc placed the register copies of the loop's rotation in FRONT of the hand-written
// `s_waitcnt` (ROCm 7.2: `v_mov_b64 v[40:41], v[36:37]` two lines above it) -- a read of a register whose load may not have
// landed.  So the kernel is compiled with amdgpu_num_vgpr(PIPE_VGPRS): the register allocator owns v0 .. v94, and
// v95 .. v127 are used by the asm statements below and by nothing else.  A load names its destination literally; the
// statement that waits for it copies the value into a compiler-visible variable behind the wait.
//   v96-v97 D.v   v98-v99 D.w   v100 D.nxt   v101 D.T   v102 D.pre   v[104 + 4u .. 107 + 4u] posting slot u   (tuples 64-bit aligned)
// (the limit is a multiple of 8, the allocation granule: a limit of 97 let the allocator use v97 -- checked in the ISA by
// tools/check_unit_kernel_resources.py's caller, see csrc/Makefile)
#define PIPE_VGPRS 96
#define PIPE_CLOBBER_D "v96", "v97", "v98", "v99", "v100", "v101", "v102"
template <int S> __device__ inline void pipe_load_slot(const Posting *src);
template <int S> __device__ inline void pipe_take_slot(u32x4 &r);
#define PIPE_SLOT(S, R0, R1, R2, R3)                                                                                           \
  template <> __device__ inline void pipe_load_slot<S>(const Posting *src) {                                                     \
    asm volatile("global_load_dwordx4 v[" #R0 ":" #R3 "], %0, off" : : "v"(src) : "memory", "v" #R0, "v" #R1, "v" #R2, "v" #R3); \
  }                                                                                                                              \
  template <> __device__ inline void pipe_take_slot<S>(u32x4 &r) {                                                               \
    asm volatile("v_mov_b32 %0, v" #R0 "\n\tv_mov_b32 %1, v" #R1 "\n\tv_mov_b32 %2, v" #R2 "\n\tv_mov_b32 %3, v" #R3             \
                 : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w)                                                                    \
                 :                                                                                                               \
                 : "memory");                                                                                                    \
  }
PIPE_SLOT(0, 104, 105, 106, 107)
PIPE_SLOT(1, 108, 109, 110, 111)
PIPE_SLOT(2, 112, 113, 114, 115)
PIPE_SLOT(3, 116, 117, 118, 119)
PIPE_SLOT(4, 120, 121, 122, 123)
PIPE_SLOT(5, 124, 125, 126, 127)
#undef PIPE_SLOT
template <int U, int S = 0> __device__ inline void pipe_take_slots(u32x4 (&raw)[U]) {
  if constexpr (S < U) {
    pipe_take_slot<S>(raw[S]);
    pipe_take_slots<U, S + 1>(raw);
  }
}
// wait until at most N vector-memory operations are outstanding
template <int N> __device__ inline void pipe_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
__device__ inline void pipe_take_D(DescRegs &D) {
  asm volatile("v_mov_b32 %0, v96\n\tv_mov_b32 %1, v97\n\tv_mov_b32 %2, v100\n\tv_mov_b32 %3, v98\n\tv_mov_b32 %4, v99\n\t"
               "v_mov_b32 %5, v101\n\tv_mov_b32 %6, v102"
               : "=v"(D.v.x), "=v"(D.v.y), "=v"(D.nxt), "=v"(D.w.x), "=v"(D.w.y), "=v"(D.T), "=v"(D.pre)
               :
               : "memory");
}

}  // namespace

template <int U>
__global__ __launch_bounds__(256, 4) __attribute__((amdgpu_num_vgpr(PIPE_VGPRS / 2))) void unit_pipe_kernel(IndexView ix, BatchView b, int k_local_floor, int n_blocks) {
  constexpr int WG = 256, NS = 64, CAP = WG * U, SCAP = FAST_SCAP;
  constexpr int BW = bloom_log2(CAP);
  constexpr int BLOOM_ALLOC = 1 << BW;
  constexpr int HB = BLOOM_POS_BITS + BW;
  constexpr int FBLOOM_WORDS = BW >= 11 ? 64 : 256;
  constexpr int FB = BW >= 11 ? 6 : 8;
  constexpr int MCAP = (CAP <= 1024) ? 64 : 128;
  static_assert(4 * NS == WG, "one thread per (cluster, quarter)");
  static_assert(U <= 6, "six posting slots of manual registers");
  __shared__ unsigned long long s_bloom[BLOOM_ALLOC];
  __shared__ unsigned long long s_fbloom[FBLOOM_WORDS];
  __shared__ UnitTables<NS, CAP> s_tab[3];
  __shared__ unsigned long long s_ent[SCAP];  // survivors: (cluster sequence number or 0x10000 | match-list entry) << 32 | flat index
  __shared__ uint32_t s_lm[WG];               // lane maxima of the data-dependent cut
  __shared__ long long s_Mid[MCAP];
  __shared__ double s_Msc[MCAP], s_Mdot[MCAP], s_Mnsq[MCAP];
  __shared__ int s_Mseq[MCAP], s_Mrole[MCAP];
  __shared__ int s_ctl2[2][CTL_N];
  // debug only (sann_debug_phase_cycles): s_memtime stamps of the iteration, kept in LDS (a store to memory would count in
  // vmcnt and tighten the pipeline's counted waits) and written out behind the hand-over
  __shared__ unsigned long long s_stamp[12];

  const int tid = threadIdx.x;
  const int G = gridDim.x;
  const int P = ix.P;

  // unit of pipeline position i: block index g + i G, mapped as sann_fast.hip maps blockIdx (all P units of a query on
  // one blockIdx % 8, i.e. one XCD's L2)
  auto unit_of = [&](int i, int &q, int &unit) -> bool {
    const long long blk = (long long)blockIdx.x + (long long)i * G;
    const int x = (int)(blk & 7);
    const long long r = blk >> 3;
    const int p = (int)(r & (P - 1));
    const long long qq = ((r >> ix.log2P) << 3) + x;
    const bool valid = blk < n_blocks && qq < b.nq;
    q = valid ? (int)qq : 0;
    unit = valid ? q * P + p : 0;
    return valid;
  };

  // ---- stage D: the unit's descriptor data into the manual registers (five loads; nothing waits here) ----------------
  auto issue_D = [&](int q, int unit) {  // (five loads)
    const int c = tid >> 2;
    const int cn = c + 1 < NS ? c + 1 : c;
    const uint2 *d = reinterpret_cast<const uint2 *>(b.desc) + (int64_t)unit * NS;
    const double *wq = b.scan_wq + (int64_t)q * NS + c;
    const int32_t *pT = b.unit_T + unit;
    const uint32_t *pp = b.unit_pre + unit;
    asm volatile("global_load_dwordx2 v[96:97], %0, off\n\t"
                 "global_load_dword v100, %1, off offset:4\n\t"
                 "global_load_dwordx2 v[98:99], %2, off\n\t"
                 "global_load_dword v101, %3, off\n\t"
                 "global_load_dword v102, %4, off"
                 :
                 : "v"(d + c), "v"(d + cn), "v"(wq), "v"(pT), "v"(pp)
                 : "memory", PIPE_CLOBBER_D);
  };
  // ---- stage H: the query header of a unit into SGPRs (three scalar loads; nothing waits here) -------------------
  auto issue_H = [&](int q, HdrRegs &H) {
    const uint64_t addr = (uint64_t)(b.hdr + q);  // (uniform; said so explicitly, or the pointer arrives in vector registers)
    const QueryHdr *hq = (const QueryHdr *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(addr >> 32)) << 32) |
                                            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)addr));
    asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(H.a) : "s"(hq) : "memory");
    asm volatile("s_load_dwordx8 %0, %1, 0x30" : "=&s"(H.b) : "s"(hq) : "memory");
    asm volatile("s_load_dword %0, %1, 0x50" : "=&s"(H.c) : "s"(hq) : "memory");
  };
  // tables of a unit from its D registers (ARRIVED); T = its posting count (0 for a position without a unit)
  auto fill_tables = [&](UnitTables<NS, CAP> &t, const DescRegs &D, uint32_t T) {
    const int c = tid >> 2, part = tid & 3;
    const bool last = c + 1 >= NS;
    if (part == 0) {
      const double w = __longlong_as_double((long long)(((unsigned long long)D.w.y << 32) | D.w.x));
      t.begin[c] = D.v.x;
      t.pre[c] = D.v.y;
      t.w[c] = w;
      t.w32[c] = (float)w;  // (wkey follows at the start of the unit's own iteration, from its query header)
    }
    // every posting of this cluster records its cluster in the flat map (an oversized unit stops at the map's end; a
    // padding entry's stretch is empty)
    uint32_t next = last ? T : D.nxt;
    next = next < (uint32_t)CAP ? next : (uint32_t)CAP;
    for (uint32_t i = D.v.y + part; i < next; i += 4) t.map[i] = (uint8_t)c;
  };
  // ---- stage P: the unit's postings into the manual registers (U loads; nothing waits here).  Every slot loads,
  // unconditionally (a slot the unit does not reach re-reads the unit's last posting; a unit without postings reads posting 0).
  auto p_src = [&](const UnitTables<NS, CAP> &t, uint32_t Tg, int u) -> const Posting * {
    const uint32_t j = (uint32_t)(u * WG + tid);
    const uint32_t jj = j < Tg ? j : (Tg ? Tg - 1 : 0u);
    const int c = (int)t.map[jj];
    const uint32_t off = Tg ? t.begin[c] + (jj - t.pre[c]) : 0u;
    return ix.postings + off;
  };
  auto issue_P = [&](const UnitTables<NS, CAP> &t, uint32_t Tg) {
    pipe_load_slot<0>(p_src(t, Tg, 0));
    pipe_load_slot<1>(p_src(t, Tg, 1));
    pipe_load_slot<2>(p_src(t, Tg, 2));
    if constexpr (U > 3) pipe_load_slot<3>(p_src(t, Tg, 3));
    if constexpr (U > 4) pipe_load_slot<4>(p_src(t, Tg, 4));
    if constexpr (U > 5) pipe_load_slot<5>(p_src(t, Tg, 5));
  };

  // ---- prologue: tables of units 0 and 1, postings of unit 0 ---------------------------------------------------------
  DescRegs Dn;
  HdrRegs H_cur, H_nxt;
  u32x4 raw_c[U];
  int q_cur, unit_cur, q_nxt, unit_nxt, q_nn, unit_nn;
  bool valid_cur = unit_of(0, q_cur, unit_cur);
  bool valid_nxt = unit_of(1, q_nxt, unit_nxt);
  bool valid_nn = false;
  uint32_t T_cur, T_nxt, T_nn = 0, pre_cur, pre_nxt, pre_nn = 0;
  {
    issue_H(q_cur, H_cur);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(H_cur.a), "+s"(H_cur.b), "+s"(H_cur.c) : : "memory");
    issue_D(q_cur, unit_cur);
    pipe_wait_vm<0>();
    pipe_take_D(Dn);
    T_cur = valid_cur ? (uint32_t)__builtin_amdgcn_readfirstlane((int)Dn.T) : 0u;
    pre_cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)Dn.pre);
    fill_tables(s_tab[0], Dn, T_cur);
    issue_D(q_nxt, unit_nxt);
    pipe_wait_vm<0>();
    pipe_take_D(Dn);
    T_nxt = valid_nxt ? (uint32_t)__builtin_amdgcn_readfirstlane((int)Dn.T) : 0u;
    pre_nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)Dn.pre);
    fill_tables(s_tab[1], Dn, T_nxt);
    if (tid < CTL_N) {
      s_ctl2[0][tid] = (tid == CTL_KMIN) ? -1 : 0;
      s_ctl2[1][tid] = (tid == CTL_KMIN) ? -1 : 0;
    }
    for (int i = tid; i < BLOOM_ALLOC; i += WG) s_bloom[i] = 0ull;
    for (int i = tid; i < FBLOOM_WORDS; i += WG) s_fbloom[i] = 0ull;
    __syncthreads();
    issue_P(s_tab[0], T_cur <= (uint32_t)CAP ? T_cur : 0u);
    pipe_wait_vm<0>();
    pipe_take_slots<U>(raw_c);
  }

  const int tid_kernel = tid;
  for (int it = 0; (long long)blockIdx.x + (long long)it * G < n_blocks; it++) {
    // (the thread number, opaque per iteration: predicates on it are then computed where they are used instead of being
    // hoisted out of the loop into SGPR pairs that live -- spilled -- for the whole kernel)
    int tid = tid_kernel;
    asm volatile("" : "+v"(tid));
    UnitTables<NS, CAP> &tab = s_tab[it % 3];
    int *const s_ctl = s_ctl2[it & 1];
    const int unit = unit_cur, q = q_cur;
    const UnitHdr h = unit_hdr(H_cur);
    const uint32_t T = T_cur;
    const uint32_t pre_tau_in = pre_cur;

#define STAMP(i) do { if (b.prof && tid == 0) s_stamp[i] = (unsigned long long)clock64(); } while (0)
    STAMP(0);
    // ---- top of the iteration: put the next two stages' loads in flight (D first: see the waits below) -----------------
    valid_nn = unit_of(it + 2, q_nn, unit_nn);
    issue_D(q_nn, unit_nn);
    issue_P(s_tab[(it + 1) % 3], T_nxt <= (uint32_t)CAP ? T_nxt : 0u);
    issue_H(q_nxt, H_nxt);
    STAMP(1);  // next stages issued

    // ---- C(n): the unit itself, as sann_fast.hip ----------------------------------------------------------------------
    bool overflow = h.n_scan > NS;  // uniform (cannot happen: the launch checks the batch's largest n_scan)
    const bool overflow_n = overflow;
    if (!overflow && T > (uint32_t)CAP) overflow = true;
    const bool overflow_T = overflow && !overflow_n;
    const int kl = unit_kl(h.k, P, k_local_floor);
    const int keep_all = SCAP < kl + kl / 2 + 16 ? SCAP : kl + kl / 2 + 16;
    const bool pre_cut = unit_has_cluster_cut(h) && !overflow;  // uniform
    const uint32_t pre_tau = pre_cut ? pre_tau_in : 0u;

    // the cosine forms' per-cluster key of a single-cluster candidate (sann_fast.hip fills it with the tables; here the
    // tables are filled two units ahead, when this unit's header has not been loaded): published by the barrier below
    if ((tid & 3) == 0) tab.wkey[tid >> 2] = cosine_cluster_key(h.alg, tab.w[tid >> 2], h.inv_l2_32);

    // ---- 2. the postings (loaded during the previous iteration): filters, fp32 copy, hash --------------------------------
    float s32[U];
    int seq[U];
    int live = 0;
    uint32_t hsh[U];
    const uint32_t Tg = (overflow || !valid_cur) ? 0u : T;
#pragma unroll
    for (int u = 0; u < U; u++) {
      seq[u] = -1;
      s32[u] = 0.f;
      hsh[u] = 0u;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if ((uint32_t)(u * WG) < Tg) {
        const uint32_t j = (uint32_t)(u * WG + tid);
        const bool have = j < Tg;
        const int c = (int)tab.map[have ? j : Tg - 1];
        const long long idv = (long long)(((unsigned long long)raw_c[u].y << 32) | raw_c[u].x);
        const double scv = __longlong_as_double((long long)(((unsigned long long)raw_c[u].w << 32) | raw_c[u].z));
        const bool excluded = h.excl_enabled != 0 && idv == h.src_excl;  // ApproximateCosineSimilarity.scala:90
        const bool in_window = idv >= h.earliest && idv <= h.latest;     // :91
        const bool keep = have && !excluded && in_window;
        seq[u] = keep ? c : -1;
        s32[u] = (float)scv;
        live += __popcll(__ballot(keep));  // wave count, identical in all lanes
        hsh[u] = table_hash(idv, HB);
      }
    }
    // ---- 3a. blocked Bloom filter: four bits of one 64-bit word, one returning LDS atomic per posting ----------------------
    {
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
          const uint32_t hv = hsh[u];
          const unsigned long long bits = bloom_bits(hv);
          atomicOr(&s_fbloom[hv >> (HB - FB)], bits);
          s_ctl[CTL_NFLAG] = 1;
        }
      }
      if ((tid & 63) == 0 && live) atomicAdd(&s_ctl[CTL_LIVE], live);
    }
    STAMP(2);  // filters + Bloom atomics issued
    __syncthreads();
    STAMP(3);  // barrier X1
    // the Bloom filter is dead: cleared for the next unit here (published by the barriers below); the other unit's
    // control words likewise (their unit, n - 1, is over for every wave that has passed the barrier above)
    for (int i = tid; i < BLOOM_ALLOC; i += WG) s_bloom[i] = 0ull;
    if (tid < CTL_N) s_ctl2[(it + 1) & 1][tid] = (tid == CTL_KMIN) ? -1 : 0;
    const bool flagged = !overflow && s_ctl[CTL_NFLAG] != 0;  // uniform

    // ---- 3b. resolve flagged ids (sann_fast.hip 3b) ------------------------------------------------------------------------
    if (flagged) {
      // a flagged unit looks its postings' hashes up in the flagged filter; the few that match join the match list -- with
      // their id and fp64 score straight from the posting registers (sann_fast.hip fetches them again: its 64 registers
      // cannot keep the 16-byte postings; here they are still live, and a fetch would wait behind stage P's loads)
      int mi[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t hv = hsh[u];
        const unsigned long long bits = bloom_bits(hv);
        const bool hit = seq[u] >= 0 && (s_fbloom[hv >> (HB - FB)] & bits) == bits;
        mi[u] = -1;
        if (hit) {
          const int m = atomicAdd(&s_ctl[CTL_NM], 1);
          mi[u] = m;
          if (m < MCAP) {
            s_Mid[m] = (long long)(((unsigned long long)raw_c[u].y << 32) | raw_c[u].x);
            s_Mseq[m] = seq[u];
            s_Msc[m] = __longlong_as_double((long long)(((unsigned long long)raw_c[u].w << 32) | raw_c[u].z));
          }
        }
      }
      __syncthreads();
      // (the flagged filter is dead: cleared for the next unit, published by the barriers below)
      for (int i = tid; i < FBLOOM_WORDS; i += WG) s_fbloom[i] = 0ull;
      const int nm = s_ctl[CTL_NM];
      if (nm > MCAP) {
        overflow = true;
        if (tid == 0) s_ctl[CTL_BAD] = 8;  // (diagnostics: the match list overflowed)
      } else if (nm > 12 && nm <= 64) {
        // entries sorted by (tweet id, cluster sequence) in wave 0's registers: groups are runs of neighbouring lanes
        if (tid < 64) {
          const bool have = tid < nm;
          uint64_t kh = have ? id_key(s_Mid[tid]) : 0ull;  // descending id_key = ascending id
          uint64_t klo = have ? ~(uint64_t)(((uint32_t)s_Mseq[tid] << 6) | (uint32_t)tid) : 0ull;  // (never 0 for an entry)
          wave_sort_desc_k128(kh, klo);
          const bool lv = (kh | klo) != 0ull;
          const uint32_t pk = (uint32_t)~klo;  // cluster sequence << 6 | entry
          const int ee = (int)(pk & 63u);
          const uint64_t prev = __shfl_up((unsigned long long)kh, 1, 64);
          const bool start = lv && (tid == 0 || prev != kh);
          const unsigned long long sm = __ballot(start);
          const int n_ent = __popcll(__ballot(lv));
          const unsigned long long above = tid == 63 ? 0ull : (sm >> (tid + 1));
          const int len = start ? (above != 0ull ? __ffsll((long long)above) : n_ent - tid) : 0;
          const int role = !lv ? 0 : (start ? (len >= 2 ? 1 : 0) : 2);
          const int max_len = (int)wave_max_u32((uint32_t)len);
          double dot = 0.0, nsq = 0.0;
          for (int j = 0; j < max_len; j++) {  // (uniform)
            const uint32_t pj = (uint32_t)__shfl((int)pk, tid + j < 64 ? tid + j : 63, 64);
            if (role == 1 && j < len) {
              const double bs = s_Msc[pj & 63u];
              dot = dot + bs * tab.w[pj >> 6];  // :92-94
              nsq = nsq + bs * bs;              // :95-96
            }
          }
          if (lv) {
            if (role == 1) {
              s_Mdot[ee] = dot;
              s_Mnsq[ee] = nsq;
            }
            s_Mrole[ee] = role;
          }
        }
      } else {
        // one thread per match-list entry, ids compared inside the list
        for (int m = tid; m < nm; m += WG) {
          const long long my = s_Mid[m];
          const int myseq = s_Mseq[m];
          int cnt = 0, rep = myseq;
#pragma unroll 4
          for (int e2 = 0; e2 < nm; e2++) {
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
              int lastc = -1;
              for (int rr = 0; rr < cnt; rr++) {  // ascending cluster sequence
                int best = 0x7fffffff;
                double bs = 0.0;
#pragma unroll 4
                for (int e2 = 0; e2 < nm; e2++) {
                  const int se = s_Mseq[e2];
                  if (s_Mid[e2] == my && se > lastc && se < best) { best = se; bs = s_Msc[e2]; }
                }
                dot = dot + bs * tab.w[best];  // :92-94
                nsq = nsq + bs * bs;           // :95-96
                lastc = best;
              }
              s_Mdot[m] = dot;
              s_Mnsq[m] = nsq;
              role = 1;
            }
          }
          s_Mrole[m] = role;
        }
      }
      __syncthreads();
      if (nm <= MCAP) {
        int folded = 0;
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int role = mi[u] >= 0 ? s_Mrole[mi[u]] : 0;
          // representative: carries the group's sums (bit 16 + its list index); the others fold into it
          seq[u] = role == 1 ? (0x10000 | mi[u]) : (role == 2 ? -1 : seq[u]);
          folded += __popcll(__ballot(role == 2));
        }
        if ((tid & 63) == 0 && folded) {
          atomicSub(&s_ctl[CTL_LIVE], folded);
          atomicAdd(&s_ctl[CTL_FOLD], folded);
        }
      }
      __syncthreads();
    }

    STAMP(4);  // Bloom cleared, duplicates resolved
    // ---- 4. approximate fp32 keys (sann_fast.hip 4) ----------------------------------------------------------------------------
    uint32_t k32[U];
    {
      bool bad = false;
      if (overflow) {
#pragma unroll
        for (int u = 0; u < U; u++) k32[u] = 0u;
      } else if (h.alg == 2 || h.alg == 4) {
#pragma unroll
        for (int u = 0; u < U; u++) {
          const bool lv = seq[u] >= 0;
          const uint32_t wk = tab.wkey[seq[u] & (NS - 1)];  // (dead slots and representatives read some entry: unused)
          constexpr uint32_t LO = 0x26901d7du, HI = 0x58635fa9u;  // 1e-15f, 1e15f
          const bool ordinary = (__float_as_uint(s32[u]) - (LO + 1u)) < (HI - LO - 1u);
          bad = bad || ((uint32_t)seq[u] < 0x10000u && !(ordinary && wk != 0u));
          k32[u] = lv ? wk : 0u;
        }
      } else if (h.alg == 1) {
#pragma unroll
        for (int u = 0; u < U; u++) {
          const bool lv = seq[u] >= 0;
          const float a = s32[u] * tab.w32[lv ? (seq[u] & (NS - 1)) : 0];
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
            // below 1e-6 the exact form's rounding of 1 + nsq decides the score: that needs the fp64 score, in the registers
            double nsq64 = 0.0;
            if (sv * sv < 1e-6f) {
              const double sd = __longlong_as_double((long long)(((unsigned long long)raw_c[u].w << 32) | raw_c[u].z));
              nsq64 = sd * sd;
            }
            a = approx_score(3, sv * tab.w32[seq[u] & (NS - 1)], sv * sv, nsq64, 0.f, invln, &forced);
          }
          bad = bad || (lv && !(seq[u] & 0x10000) && !forced && !(a > 1e-30f && a < 1e30f && sv > 1e-15f && sv < 1e15f));
          k32[u] = lv ? (forced ? FORCED_KEY : (__float_as_uint(a) | 0x80000000u)) : 0u;
        }
      }
      if (flagged && !overflow) {  // uniform: only units that resolved duplicates can hold representatives
        const float invl2 = h.inv_l2_32, invln = h.inv_ln_32;
#pragma unroll
        for (int u = 0; u < U; u++) {
          if (seq[u] >= 0 && (seq[u] & 0x10000)) {
            const double nsq64 = s_Mnsq[seq[u] & 0xffff];
            const float d32 = (float)s_Mdot[seq[u] & 0xffff], n32 = (float)nsq64;
            bool forced;
            const float a = approx_score(h.alg, d32, n32, nsq64, invl2, invln, &forced);
            bad = bad || (!forced && !(a > 1e-30f && a < 1e30f && n32 > 1e-30f && n32 < 1e30f));
            k32[u] = forced ? FORCED_KEY : (__float_as_uint(a) | 0x80000000u);
          }
        }
      }
      if (__ballot(bad) != 0ull && (tid & 63) == 0) atomicOr(&s_ctl[CTL_BAD], 1);
    }

    // ---- the tables of unit n + 2: its descriptor data has had the phases above to arrive.  D was issued before P, so
    // "all but the U youngest loads are done" says exactly that D is (anything issued since only makes the wait stricter).
    // Filled in front of the compaction barrier below, which publishes them for the next iteration's stage P.
    STAMP(5);  // keys
    pipe_wait_vm<U>();
    STAMP(6);  // D arrived
    pipe_take_D(Dn);
    T_nn = valid_nn ? (uint32_t)__builtin_amdgcn_readfirstlane((int)Dn.T) : 0u;
    pre_nn = (uint32_t)__builtin_amdgcn_readfirstlane((int)Dn.pre);
    fill_tables(s_tab[(it + 2) % 3], Dn, T_nn);

    STAMP(7);  // tables of unit n + 2 filled
    // ---- 5a / 5b. the cut and the compaction (sann_fast.hip 5a, 5b) --------------------------------------------------------------
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
            {
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
      }
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
      // the cluster-level cut counted postings the filters then removed: cut by the data after all
      if (!cut_by_data && tau != 0u && s_ctl[CTL_NSURV] < kl && s_ctl[CTL_NSURV] < s_ctl[CTL_LIVE] &&
          s_ctl[CTL_LIVE] + s_ctl[CTL_FOLD] < (int)T && !s_ctl[CTL_BAD] && !overflow) {
        __syncthreads();  // (everyone has read the counters)
        if (tid == 0) s_ctl[CTL_NSURV] = 0;
        cut_by_data = true;
        continue;  // (the barrier after the lane maxima orders the reset before the next compaction)
      }
      break;
    }

    // ---- the postings of unit n + 1 have had the whole iteration to arrive; waited for in front of the hand-over's stores
    // (behind them the wait would also cover the stores' acknowledgements)
    u32x4 raw_n[U];
    STAMP(8);  // cut + compaction
    pipe_wait_vm<0>();
    STAMP(9);  // P arrived
    pipe_take_slots<U>(raw_n);

    // ---- 6. hand the survivors over (sann_fast.hip 6) -------------------------------------------------------------------------------
    if (valid_cur) {
      if (s_ctl[CTL_BAD] && !overflow) overflow = true;
      if (overflow) {
        if (tid == 0) pipe_unit_overflowed(unit, overflow_n ? 1ull : overflow_T ? 2ull : (s_ctl[CTL_BAD] & 8) ? 3ull : 4ull);
      } else if (s_ctl[CTL_NSURV] > SCAP) {
        if (tid == 0) pipe_unit_overflowed(unit, 5ull);
      } else {
        const int ns = s_ctl[CTL_NSURV];
        // theta: every candidate below the cut has approx < tau, hence exact < tau * (1 + 2 EPS) -- unless every live
        // candidate survived the cut: then nothing is withheld
        unsigned long long theta_key = 0ull;
        if (tau != 0u && s_ctl[CTL_NSURV] < s_ctl[CTL_LIVE]) {
          double tau_val = (double)__uint_as_float(tau & 0x7fffffffu);
          tau_val = tau_val < 1e30 ? tau_val : 1e30;  // a cut inside the forced (+inf) band: the others are still < 1e30
          theta_key = score_key(tau_val * (1.0 + 2.0 * (double)APPROX_EPS));
        }
        const int64_t obase = (int64_t)unit * LATE_ARG(cap);
        uint64_t *const cand_key = LATE_ARG(cand_key);
        int64_t *const cand_id = LATE_ARG(cand_id);
        for (int i = tid; i < ns; i += WG) {
          const unsigned long long e = s_ent[i];
          const int c = (int)(e >> 32);
          unsigned long long key = CAND_DEFERRED;
          long long idv;
          if (c & 0x10000) {
            idv = s_Mid[c & 0xffff];
            // (the query's norms and minScore are read here, by the few lanes that need them)
            const QueryHdr *hq = LATE_ARG(hdr) + q;
            const double l2 = *(const volatile double *)&hq->l2norm, ln = *(const volatile double *)&hq->lognorm;
            const double v = normalise_f(h.alg, s_Mdot[c & 0xffff], s_Mnsq[c & 0xffff], l2, ln);
            key = v >= *(const volatile double *)&hq->min_score ? score_key(v) : CAND_DROPPED;  // :125 (false for NaN)
          } else {
            const uint32_t pos = tab.begin[c] + ((uint32_t)e - tab.pre[c]);
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
      }
    }

    STAMP(10);  // handed over
    if (b.prof && tid == 0 && valid_cur) {
      // slots as sann_debug_phase_cycles reads them: [0] start, [1..8] eight phase ends
      unsigned long long *o = b.prof + (int64_t)unit * 16;
      o[0] = s_stamp[0]; o[1] = s_stamp[1]; o[2] = s_stamp[3]; o[3] = s_stamp[4]; o[4] = s_stamp[5]; o[5] = s_stamp[6];
      o[6] = s_stamp[7]; o[7] = s_stamp[9]; o[8] = s_stamp[10];
      o[11] = s_stamp[2]; o[12] = s_stamp[8];
    }
#undef STAMP
    // ---- rotate the pipeline ---------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < U; u++) raw_c[u] = raw_n[u];
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(H_nxt.a), "+s"(H_nxt.b), "+s"(H_nxt.c) : : "memory");
    H_cur = H_nxt;
    q_cur = q_nxt; unit_cur = unit_nxt; valid_cur = valid_nxt; T_cur = T_nxt; pre_cur = pre_nxt;
    q_nxt = q_nn; unit_nxt = unit_nn; valid_nxt = valid_nn; T_nxt = T_nn; pre_nxt = pre_nn;
  }
  // (loads of positions past the end were issued for unit 0 and have been waited for: nothing is in flight here)
}

// Workgroups per CU: four (128 registers, ~33 KB of LDS each).  SANN_PIPE_WGS overrides (measurement).
hipError_t launch_unit_pipe(const IndexView &ix, const BatchView &b, const FastParams &fp, hipStream_t stream) {
  const int nq8 = (b.nq + 7) / 8 * 8;
  const int n_blocks = nq8 * ix.P;
  if (n_blocks <= 0) return hipSuccess;
  if (b.desc_stride != 64 || fp.max_n_scan > 64 || fp.use_norms) return hipErrorInvalidValue;
  static const int wgs_per_cu = [] {
    const char *e = getenv("SANN_PIPE_WGS");
    const int v = e ? atoi(e) : 4;
    return v >= 1 && v <= 8 ? v : 4;
  }();
  int grid = 256 * wgs_per_cu;  // (a multiple of 8: a workgroup's units stay on one blockIdx % 8)
  if (grid > n_blocks) grid = n_blocks;
  switch (fp.unit_capacity) {
    case 768: hipLaunchKernelGGL((unit_pipe_kernel<3>), dim3(grid), dim3(256), 0, stream, ix, b, fp.k_local, n_blocks); break;
    case 1024: hipLaunchKernelGGL((unit_pipe_kernel<4>), dim3(grid), dim3(256), 0, stream, ix, b, fp.k_local, n_blocks); break;
    case 1536: hipLaunchKernelGGL((unit_pipe_kernel<6>), dim3(grid), dim3(256), 0, stream, ix, b, fp.k_local, n_blocks); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

bool unit_pipe_serves(const BatchView &b, const FastParams &fp) {
  static const bool on = [] {
    const char *e = getenv("SANN_PIPE");
    return !(e && e[0] == '0');
  }();
  if (!on || b.desc_stride != 64 || fp.max_n_scan > 64 || fp.use_norms) return false;
  return fp.unit_capacity == 768 || fp.unit_capacity == 1024 || fp.unit_capacity == 1536;  // (eight slots per thread do not fit 128 registers)
}

}  // namespace sann
