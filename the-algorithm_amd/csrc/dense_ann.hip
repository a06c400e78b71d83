// dense_ann.hip -- exhaustive dense nearest-neighbour search on the gfx950 matrix cores.
//
// What it replaces: BruteForceIndex.queryWithDistance (ann/src/main/scala/com/twitter/ann/brute_force/
// BruteForceIndex.scala:66-91: score every stored vector, keep the k smallest distances, return them
// ascending) with the metrics of ann/.../common/Metric.scala:88-185 and the "Cosine = normalise, then
// InnerProduct" rule of ann/.../hnsw/DistanceFunctionGenerator.scala:12-30.
//
// Shape of the computation.  score[v][q] = <x_v, q> (+ a per-vector bias: -|x_v|^2/2 for L2, so that
// a larger score is always a smaller distance) is a [N x d] * [d x nq] product: with nq ~ 1000
// queries it has ~1000 flop per index byte, far right of the ridge (2.5 PFLOP/s / 8 TB/s ~ 310), so
// the path is MFMA-bound and is laid out for v_mfma_f32_32x32x16_f16:
//   * the index is stored in HBM already in MFMA A-fragment order -- per block of 32 vectors,
//     [k-step s][lane][8 halves] with lane (r, h) holding x[r][16s + 8h .. +8] -- so a wave loads its
//     vectors with perfectly coalesced 1-KiB global_load_dwordx4 and keeps them IN REGISTERS
//     (VB blocks x S k-steps x 4 registers = all 256 AGPRs) for the whole life of the workgroup;
//   * the queries are pre-arranged once per search in the same fragment order (B operand); blocks of
//     32 queries are streamed through a three-slot LDS ring shared by the 4 waves, each wave reading
//     one conflict-free ds_read_b128 per four MFMAs;
//   * C has the query on the lane and 16 vectors in the accumulator registers, so the per-query
//     reductions of the epilogue (tile maximum, threshold test) are register-local.
// The [N x nq] score matrix is never written.  Exact top-k without it:
//   pass A  the same GEMM over a strided sample of the index writes only per-(query, 64-vector tile)
//           maxima; tau_q = the k-th largest tile maximum is a lower bound of the k-th best score
//           (k distinct vectors reach it).  The sample is sized so that ~256 sqrt(k) scores pass tau_q.
//   pass B  the GEMM over the whole index appends every score >= tau_q to a per-query buffer.
//   refine  a query whose buffer overflowed takes the k-th largest buffered score as a tighter bound
//           and pass B is repeated for it (never happens at the sizes pass A is tuned for; it is the
//           general fallback for tiny indexes / huge k).
//   select  per query: sort the <= CAP survivors by (score desc, position asc), convert to distances.
// Positions are in id order (the builder sorts by id), so ties resolve by id ascending.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/dense_ann.h"
#include "sann_device.h"  // mix64
#include "abi_guard.h"
#define ABI_CATCH catch (...) { return abi_guard::caught(fail, DANN_ENOMEM, DANN_EINTERNAL); }

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &m) {
  g_err = m;
  return code;
}
#define DTRY(expr)                                                                                \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return fail(DANN_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int W2 = 4;             // waves per workgroup: one per SIMD, the whole 512-register file each
constexpr int TILE_MAXIMA = 2 * W2;  // pass A writes two tile maxima per wave
constexpr int CAP = 8192;         // survivors kept per query
constexpr int MAX_K = 1024;
constexpr int MAX_D = 512;
constexpr int MAX_NQ = 4096;     // queries per GEMM launch

struct Survivor {
  float score;
  uint32_t pos;
};

__device__ __forceinline__ uint32_t f2key(float f) {  // order-preserving float -> uint
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// ---------------------------------------------------------------------------------------------
// rows (fp32, row-major) -> fp16 MFMA fragments.  One wave per row.  Serves the index (A operand)
// and the queries (B operand): both want "row r of a 32-row block on lane r + 32h, k = 16s + 8h + j".
// sumsq[row] = sum of the squares of the stored (rounded) halves, in fp32 arithmetic order-fixed.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ void prep_rows_kernel(const float *__restrict__ src, int64_t n, int d, int S, int normalise,
                                 int64_t row0, _Float16 *__restrict__ frag, float *__restrict__ sumsq) {
  int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= n) return;
  const float *x = src + row * d;
  double ss = 0;
  for (int k = lane; k < d; k += 64) ss += (double)x[k] * (double)x[k];
  ss = wave_sum(ss);
  float norm = 1.0f;
  if (normalise) {
    norm = (float)sqrt(ss);
    if (!(norm > 0.0f)) norm = 1.0f;
  }
  int64_t grow = row0 + row;
  int64_t g = grow >> 5;
  int r = (int)(grow & 31);
  double ss16 = 0;
  for (int k = lane; k < S * 16; k += 64) {
    float v = k < d ? x[k] / norm : 0.0f;
    _Float16 hv = (_Float16)v;
    float back = (float)hv;
    ss16 += (double)back * (double)back;
    int s = k >> 4, h = (k >> 3) & 1, j = k & 7;
    frag[(((g * S + s) * 64) + h * 32 + r) * 8 + j] = hv;
  }
  ss16 = wave_sum(ss16);
  if (lane == 0) sumsq[grow] = (float)ss16;
}

// bias[v] = 0 (InnerProduct / Cosine), -sumsq/2 (L2), -inf (padding rows)
__global__ void bias_kernel(float *__restrict__ bias, int64_t n, int64_t n_pad, int metric) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  if (i >= n) bias[i] = -INFINITY;
  else bias[i] = metric == DANN_METRIC_L2 ? -0.5f * bias[i] : 0.0f;
}

// fragments -> fp32 rows (audit)
__global__ void unfrag_kernel(const _Float16 *__restrict__ frag, int S, int d, int64_t i0, int64_t n,
                              float *__restrict__ out) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * d) return;
  int64_t row = i0 + e / d;
  int k = (int)(e % d);
  int64_t g = row >> 5;
  int r = (int)(row & 31), s = k >> 4, h = (k >> 3) & 1, j = k & 7;
  out[e] = (float)frag[(((g * S + s) * 64) + h * 32 + r) * 8 + j];
}

// synthetic index: i.i.d. N(0,1), one thread per vector
__device__ __forceinline__ float gauss(uint64_t seed, uint64_t v, uint32_t k) {
  uint64_t h = sann::mix64(seed ^ (v * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(k >> 1) << 48) ^ (k >> 1));
  float u1 = ((uint32_t)(h >> 40) + 1u) * (1.0f / 16777217.0f);
  float u2 = (uint32_t)((h >> 8) & 0xffffffu) * (1.0f / 16777216.0f);
  float rad = sqrtf(-2.0f * __logf(u1));
  float ang = 6.2831853f * u2;
  return (k & 1) ? rad * __sinf(ang) : rad * __cosf(ang);
}

__global__ void synth_kernel(_Float16 *__restrict__ frag, float *__restrict__ bias, int64_t n, int d, int S,
                             int metric, uint64_t seed) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  float norm = 1.0f;
  if (metric == DANN_METRIC_COSINE) {
    double ss = 0;
    for (int k = 0; k < d; ++k) {
      float x = gauss(seed, (uint64_t)v, (uint32_t)k);
      ss += (double)x * x;
    }
    norm = (float)sqrt(ss);
  }
  int64_t g = v >> 5;
  int r = (int)(v & 31);
  double ss16 = 0;
  for (int c = 0; c < S * 2; ++c) {  // 8-element chunk c = 2s + h
    half8 out;
    for (int j = 0; j < 8; ++j) {
      int k = c * 8 + j;
      float x = k < d ? gauss(seed, (uint64_t)v, (uint32_t)k) / norm : 0.0f;
      _Float16 hv = (_Float16)x;
      out[j] = hv;
      ss16 += (double)(float)hv * (double)(float)hv;
    }
    *(half8 *)&frag[(((g * S + (c >> 1)) * 64) + (c & 1) * 32 + r) * 8] = out;
  }
  bias[v] = (float)ss16;
}

// ---------------------------------------------------------------------------------------------
// the GEMM
// ---------------------------------------------------------------------------------------------
struct GemmArgs {
  const _Float16 *xf;    // index fragments
  const float *bias;     // per vector
  const _Float16 *qf;    // query fragments
  int nqb;               // query blocks of 32
  int64_t n;             // real vectors
  uint32_t n_wg_total;   // workgroup tiles in the index
  uint32_t n_wg_launch;  // workgroup tiles this launch covers (sample or all)
  uint32_t tile0;        // first index tile of this launch
  uint32_t col0;         // first pass-A column group of this launch
  // pass A
  float *tmax;           // [nqb*32][pitch]
  int64_t pitch;
  // pass B
  const float *tau;      // [nqb*32]
  uint32_t *cnt;         // [nqb*32]
  Survivor *surv;        // [nqb*32][CAP]
};

// ---------------------------------------------------------------------------------------------
// The GEMM: ONE wave per SIMD with the whole 512-register file, registers placed by hand.
//   * the wave's VB x 32 vectors live in the 256 AGPRs (MFMA reads srcA from AGPRs directly); the
//     MFMAs are inline asm with an "a" constraint so that the allocator keeps them there -- left to
//     itself hipcc parks them in AGPRs as spill space and copies each fragment back before use;
//   * two accumulator sets in VGPRs: while the MFMAs of query block qb fill one, the epilogue of block
//     qb-1 (maximum, threshold test) runs on the other in the MFMA shadow;
//   * query fragments come from LDS through a 4-deep register ring that runs straight across block
//     boundaries (block qb+1 is already readable during block qb, see below), one ds_read_b128 per VB
//     MFMAs: half the LDS traffic of the 8-wave geometry;
//   * query blocks arrive by LDS-DMA (global_load_lds_dwordx4 -- the fragment image is lane-linear,
//     exactly what the DMA writes) into a ring of three slots, two blocks ahead.  The DMA is issued
//     from inline asm: hipcc otherwise puts s_waitcnt vmcnt(0) in front of every ds_read that
//     follows a DMA it knows about.  Ordering is by hand: at the top of block qb every wave waits
//     vmcnt(0) -- retiring stage qb+1, issued a whole block earlier, so the wait is free -- then the
//     barrier; stage qb+2 is issued after it into the slot block qb-1 was read from.  A slot is
//     read only after a barrier that followed the wait that retired it.
// MFMA hazards the compiler cannot see through the asm: see mfma_last_step.
// ---------------------------------------------------------------------------------------------
constexpr int SB_CAP = 512;  // survivors a workgroup stages in LDS before touching global memory
struct Staged {
  float score;
  uint32_t pos, q;
};

__device__ __forceinline__ void glds16_asm(const half8 *g, uint32_t lds_byte) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g), "s"(lds_byte)
      : "memory");
}

__device__ __forceinline__ void mfma_first(float16v &d, const half8 &a, const half8 &b) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma_bias(float16v &d, const half8 &a, const half8 &b, const float16v &c) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "a"(a), "v"(b), "v"(c));
}
__device__ __forceinline__ void mfma_acc(float16v &d, const half8 &a, const half8 &b) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "v"(b));
}

// The last k-step of a query block: the VB MFMAs and the wait states their results need before any
// VALU may read them (19 after a 16-pass MFMA), in ONE asm statement.  The compiler cannot see the
// hazard through the asm and is free to read an accumulator right behind its last MFMA (it does: the
// register copies at the merge in front of the final epilogue read stale rows 24..31 of the last
// block); with the wait inside the statement there is no such place.
template <int VB>
__device__ __forceinline__ void mfma_last_step(float16v (&c)[VB], const half8 (&a)[VB], const half8 &b) {
  if constexpr (VB == 4)
    asm volatile(
        "v_mfma_f32_32x32x16_f16 %0, %4, %8, %0\n\t"
        "v_mfma_f32_32x32x16_f16 %1, %5, %8, %1\n\t"
        "v_mfma_f32_32x32x16_f16 %2, %6, %8, %2\n\t"
        "v_mfma_f32_32x32x16_f16 %3, %7, %8, %3\n\t"
        "s_nop 15\n\ts_nop 3"
        : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
        : "a"(a[0]), "a"(a[1]), "a"(a[2]), "a"(a[3]), "v"(b));
  else
    asm volatile(
        "v_mfma_f32_32x32x16_f16 %0, %2, %4, %0\n\t"
        "v_mfma_f32_32x32x16_f16 %1, %3, %4, %1\n\t"
        "s_nop 15\n\ts_nop 3"
        : "+v"(c[0]), "+v"(c[1])
        : "a"(a[0]), "a"(a[1]), "v"(b));
}

template <int S, int VB, bool EMIT, bool BIAS>
__global__ __launch_bounds__(W2 * 64) void gemm2_kernel(GemmArgs a) {
  constexpr int CH = S * 64, NB = 3, NCH = CH / (W2 * 64);
  constexpr int FOLD = (VB * 16 + S / 2 - 1) / (S / 2);  // accumulator registers folded per k-step (first half of a block)
  extern __shared__ half8 smem[];
  float4 *sbias = (float4 *)(smem + NB * CH);    // [W2][VB][8]
  uint32_t *sb_n = (uint32_t *)(sbias + W2 * VB * 8);  // survivor staging: count (+ 3 words of padding)
  Staged *sb = (Staged *)(sb_n + 4);                   // [SB_CAP]
  float *stau = (float *)(sb + SB_CAP);                // [nqb * 32]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const uint32_t tile = a.tile0 + (uint32_t)(((uint64_t)blockIdx.x * a.n_wg_total) / a.n_wg_launch);
  const int64_t g0 = ((int64_t)tile * W2 + w) * VB;
  const half8 *qsrc = (const half8 *)a.qf;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) half8 *)smem;
  const int wu = __builtin_amdgcn_readfirstlane(w);

  auto stage = [&](int qb, int slot) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      glds16_asm(qsrc + (size_t)qb * CH + (c * W2 + wu) * 64 + lane,
                 lds0 + (uint32_t)((slot * CH + (c * W2 + wu) * 64) * 16));
  };
  stage(0, 0);
  if (a.nqb > 1) stage(1, 1);

  half8 av[VB][S];
#pragma unroll
  for (int vb = 0; vb < VB; ++vb)
#pragma unroll
    for (int s = 0; s < S; ++s) av[vb][s] = *(const half8 *)&a.xf[((((g0 + vb) * S + s) * 64) + lane) * 8];
  if (BIAS && lane < VB * 8) sbias[w * VB * 8 + lane] = *(const float4 *)&a.bias[g0 * 32 + lane * 4];
  if (EMIT) {
    for (int i = t; i < a.nqb * 32; i += W2 * 64) stau[i] = a.tau[i];
    if (t == 0) *sb_n = 0;
  }
  // the compiler does not know about the DMA above: __syncthreads() alone waits for LDS traffic only,
  // and the fragment ring is primed from slot 0 right behind it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // epilogue of one finished block, split in two: fold() pieces run between the MFMAs of the next
  // block, finish() after them.
  float fm[VB];  // running maximum per 32-vector block
  const uint32_t vbase = (uint32_t)(g0 * 32) + 4u * (uint32_t)(lane >> 5);  // + 32 vb + row(i) = position
  const uint32_t n32 = (uint32_t)a.n;
  // element e of the fold order is register (e / VB) of block (e % VB): consecutive elements belong to
  // different blocks, so the v_max3 of one k-step are independent of each other (a dependent VALU chain in
  // the MFMA shadow costs ~3.4 cycles per instruction instead of ~1: tools/micro/mfma_loop.hip)
  auto fold = [&](float16v (&p)[VB], int first, int count) {
#pragma unroll
    for (int e = first; e < first + count && e < VB * 16; ++e) {
      const int vb = e % VB, i = e / VB;
      fm[vb] = fmaxf(fm[vb], p[vb][i]);
    }
  };
  auto finish = [&](float16v (&p)[VB], int qb) {
    const int q = qb * 32 + (lane & 31);
    if (!EMIT) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float m = VB == 4 ? fmaxf(fm[2 * h], fm[2 * h + 1]) : fm[h];
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (lane < 32) a.tmax[(int64_t)q * a.pitch + ((int64_t)(a.col0 + blockIdx.x) * W2 + w) * 2 + h] = m;
      }
    } else {
      // survivors are rare (a few hundred per query over the whole index): test per wave, then per
      // 32-vector block, and only then per element
      const float thr = stau[q];
      float mall = fm[0];
#pragma unroll
      for (int vb = 1; vb < VB; ++vb) mall = fmaxf(mall, fm[vb]);
      // cold: about one wave-block in six gets here; keeping it out of line keeps the two-block hot loop
      // (~25 KB of code) contiguous in the 64 KB instruction cache it shares with the neighbouring CU
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(mall >= thr) != 0, 0)) {
        // opaque copy: keeps the 64 positions (and their bound checks) from being hoisted out of the
        // block loop into 64 live registers
        uint32_t vb0 = vbase;
        asm volatile("" : "+v"(vb0));
#pragma unroll
        for (int vb = 0; vb < VB; ++vb) {
          if (__builtin_amdgcn_ballot_w64(fm[vb] >= thr) == 0) continue;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float sc = p[vb][i];
            const uint32_t v = vb0 + 32u * vb + (i & 3) + 8 * (i >> 2);
            // padding rows exist only where the bias path runs (score -inf; they pass only tau = -inf)
            if (sc >= thr && (!BIAS || v < n32)) {
              // a returning global atomic costs a round trip to L2 (as long as a whole query block)
              // and would stall all four waves at the next barrier: stage in LDS, flush at the end
              uint32_t e = atomicAdd(sb_n, 1u);
              if (e < (uint32_t)SB_CAP) {
                sb[e] = Staged{sc, v, (uint32_t)q};
              } else {
                uint32_t pos = atomicAdd(&a.cnt[q], 1u);
                if (pos < (uint32_t)CAP) a.surv[(size_t)q * CAP + pos] = Survivor{sc, v};
              }
            }
          }
        }
      }
    }
  };

  half8 bq[4];  // fragment ring: k-step s of any block sits in bq[s & 3] (S % 4 == 0)
#define DANN_LDB(SLOT, STEP) bq[(STEP) & 3] = smem[(SLOT) * CH + (STEP) * 64 + lane]

  // one query block: MFMAs into cur, epilogue of the previous block on prev
  auto block = [&](auto hp, float16v (&cur)[VB], float16v (&prev)[VB], int qb, int slot) {
    constexpr bool have_prev = decltype(hp)::value;
    const int slot2 = slot == 0 ? 2 : slot - 1;  // (qb + 2) % 3
    const int slot1 = slot == 2 ? 0 : slot + 1;  // (qb + 1) % 3
    float16v bt[VB];
    if (BIAS) {
#pragma unroll
      for (int vb = 0; vb < VB; ++vb)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          float4 bv = sbias[(w * VB + vb) * 8 + 2 * i4 + (lane >> 5)];
          bt[vb][4 * i4 + 0] = bv.x;
          bt[vb][4 * i4 + 1] = bv.y;
          bt[vb][4 * i4 + 2] = bv.z;
          bt[vb][4 * i4 + 3] = bv.w;
        }
    }
#pragma unroll
    for (int vb = 0; vb < VB; ++vb) fm[vb] = -INFINITY;
    const bool more = qb + 1 < a.nqb;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      if (s == 1) {
        // The block's one synchronisation point sits between two MFMA groups, not at the block
        // boundary: with a single wave per SIMD nothing else keeps the matrix pipe busy while this
        // wave waits, so the wait is taken with the four MFMAs of step 0 still executing.  It retires
        // stage qb+1 (issued a block ago: free) and orders the restaging of slot (qb-1) % 3 behind every
        // wave's last read of it (block qb-1 is over for whoever gets here).  What block qb itself
        // reads was retired by the previous block's barrier.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (qb + 2 < a.nqb) stage(qb + 2, slot2);
      }
      // keep the ring 3 fragments ahead; past the end of this block it runs into the next one
      // (unconditional: behind the last block it re-reads this slot -- a branch here makes the compiler
      // wait for every outstanding LDS read at the end of each block)
      if (s + 3 < S) DANN_LDB(slot, s + 3);
      else DANN_LDB(more ? slot1 : slot, s + 3 - S);
      if (s == S - 1) {
        half8 al[VB];
#pragma unroll
        for (int vb = 0; vb < VB; ++vb) al[vb] = av[vb][S - 1];
        mfma_last_step<VB>(cur, al, bq[s & 3]);
      } else {
#pragma unroll
        for (int vb = 0; vb < VB; ++vb) {
          if (s == 0) {
            if (BIAS) mfma_bias(cur[vb], av[vb][0], bq[0], bt[vb]);
            else mfma_first(cur[vb], av[vb][0], bq[0]);
          } else {
            mfma_acc(cur[vb], av[vb][s], bq[s & 3]);
          }
        }
      }
      // the previous block's epilogue rides in the shadow of this block's MFMAs: folds in the first
      // half, the threshold test right behind them
      if constexpr (have_prev) {
        if (s < S / 2) fold(prev, s * FOLD, FOLD);
        if (s == S / 2) finish(prev, qb - 1);
      }
    }
  };

  float16v acc0[VB], acc1[VB];
  DANN_LDB(0, 0);
  DANN_LDB(0, 1);
  DANN_LDB(0, 2);
  auto last = [&](float16v (&p)[VB], int qb) {  // epilogue of the final block: nothing left to hide it behind
#pragma unroll
    for (int vb = 0; vb < VB; ++vb) fm[vb] = -INFINITY;
    fold(p, 0, VB * 16);
    finish(p, qb);
  };
  auto next = [](int slot) { return slot == 2 ? 0 : slot + 1; };
  block(std::false_type{}, acc0, acc1, 0, 0);
  int slot = 1, qb = 1;
  for (; qb + 1 < a.nqb; qb += 2) {
    block(std::true_type{}, acc1, acc0, qb, slot);
    slot = next(slot);
    block(std::true_type{}, acc0, acc1, qb + 1, slot);
    slot = next(slot);
  }
  if (qb < a.nqb) {
    block(std::true_type{}, acc1, acc0, qb, slot);
    last(acc1, qb);
  } else {
    last(acc0, a.nqb - 1);
  }
  if (EMIT) {  // flush the staged survivors
    __syncthreads();
    const uint32_t ns = min(*sb_n, (uint32_t)SB_CAP);
    for (uint32_t e = t; e < ns; e += W2 * 64) {
      const Staged sv = sb[e];
      uint32_t pos = atomicAdd(&a.cnt[sv.q], 1u);
      if (pos < (uint32_t)CAP) a.surv[(size_t)sv.q * CAP + pos] = Survivor{sv.score, sv.pos};
    }
  }
#undef DANN_LDB
}

// ---------------------------------------------------------------------------------------------
// k-th largest of n floats (stride in floats), one workgroup, 4 radix passes over an LDS histogram
// ---------------------------------------------------------------------------------------------
__device__ float wg_kth_largest(const float *vals, int64_t n, int stride, int k, uint32_t *hist /*[258]*/) {
  uint32_t prefix = 0, mask = 0;
  uint32_t want = (uint32_t)k;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
      uint32_t key = f2key(vals[i * stride]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t acc = 0;
      int dgt = 255;
      for (; dgt > 0; --dgt) {
        if (acc + hist[dgt] >= want) break;
        acc += hist[dgt];
      }
      hist[256] = (uint32_t)dgt;
      hist[257] = want - acc;
    }
    __syncthreads();
    prefix |= hist[256] << shift;
    mask |= 255u << shift;
    want = hist[257];
    __syncthreads();
  }
  return key2f(prefix);
}

// tau[q] from the pass-A tile maxima; padded queries get +inf (never emit)
// (exact mode: slack > 0 lowers tau by two rounding bounds, so that the vectors the fp32 re-rank could still prefer are
// among the survivors from the first pass on; see prove_kernel)
__global__ void tau_kernel(const float *__restrict__ tmax, int64_t pitch, int64_t n_tiles, int k, int nq,
                           float *__restrict__ tau, uint32_t *__restrict__ cnt, const float *__restrict__ qsumsq, float slack_norm,
                           int metric) {
  __shared__ uint32_t hist[258];
  int q = blockIdx.x;
  float v;
  if (q >= nq) v = INFINITY;
  else if (n_tiles < k) v = -INFINITY;
  else v = wg_kth_largest(tmax + (int64_t)q * pitch, n_tiles, 1, k, hist);
  if (slack_norm > 0.0f && q < nq && v > -INFINITY)
    v -= 2.0f * (slack_norm * sqrtf(qsumsq[q]) * 1.05e-3f + (metric == DANN_METRIC_L2 ? slack_norm * slack_norm * 5.0e-4f : 0.0f));
  if (threadIdx.x == 0) {
    tau[q] = v;
    cnt[q] = 0;
  }
}

// after pass B: overflowed queries get a tighter tau and are re-armed; the others are parked at +inf.
// status: 0 done, 1 redo, 2 cannot tighten (more than CAP scores tie at the k-th)
__global__ void refine_kernel(float *__restrict__ tau, uint32_t *__restrict__ cnt, uint32_t *__restrict__ done_cnt,
                              const Survivor *__restrict__ surv, int k, int nq, int *__restrict__ status,
                              int *__restrict__ flags, float *__restrict__ tau_used) {
  __shared__ uint32_t hist[258];
  int q = blockIdx.x;
  if (q >= nq) return;
  if (done_cnt[q] != 0xffffffffu) return;  // finished in an earlier round
  uint32_t c = cnt[q];
  if (c <= (uint32_t)CAP) {
    if (threadIdx.x == 0) {
      done_cnt[q] = c;
      status[q] = 0;
      if (tau_used) tau_used[q] = tau[q];  // (exact mode: the threshold this query's survivors passed)
      tau[q] = INFINITY;
    }
    return;
  }
  float old = tau[q];
  float nt = wg_kth_largest(&surv[(size_t)q * CAP].score, CAP, 2, k, hist);
  if (threadIdx.x == 0) {
    if (nt > old) {
      tau[q] = nt;
      cnt[q] = 0;
      status[q] = 1;
      atomicOr(&flags[0], 1);
    } else {
      status[q] = 2;
      tau[q] = INFINITY;
      atomicOr(&flags[0], 2);
    }
  }
}

// per query: sort survivors by (score desc, position asc), emit the k nearest as distances
__global__ __launch_bounds__(512) void select_kernel(const Survivor *__restrict__ surv, const uint32_t *__restrict__ done_cnt,
                                                     const float *__restrict__ qsumsq, const int64_t *__restrict__ ids,
                                                     int metric, int k, float *__restrict__ out_dist,
                                                     int64_t *__restrict__ out_ids, int32_t *__restrict__ out_counts) {
  extern __shared__ unsigned long long keys[];
  const int q = blockIdx.x;
  const uint32_t c = min(done_cnt[q], (uint32_t)CAP);
  uint32_t n2 = 64;
  while (n2 < c) n2 <<= 1;
  for (uint32_t i = threadIdx.x; i < n2; i += blockDim.x) {
    unsigned long long key = 0;
    if (i < c) {
      Survivor s = surv[(size_t)q * CAP + i];
      key = ((unsigned long long)f2key(s.score) << 32) | (0xffffffffu - s.pos);
    }
    keys[i] = key;
  }
  __syncthreads();
  for (uint32_t size = 2; size <= n2; size <<= 1)
    for (uint32_t str = size >> 1; str > 0; str >>= 1) {
      for (uint32_t i = threadIdx.x; i < n2 / 2; i += blockDim.x) {
        uint32_t lo = 2 * i - (i & (str - 1));
        uint32_t hi = lo + str;
        bool desc = (lo & size) == 0;
        unsigned long long x = keys[lo], y = keys[hi];
        if ((x < y) == desc) {
          keys[lo] = y;
          keys[hi] = x;
        }
      }
      __syncthreads();
    }
  const uint32_t m = min(c, (uint32_t)k);
  for (uint32_t i = threadIdx.x; i < (uint32_t)k; i += blockDim.x) {
    float dist = 0.0f;
    int64_t id = 0;
    if (i < m) {
      unsigned long long key = keys[i];
      float sc = key2f((uint32_t)(key >> 32));
      uint32_t pos = 0xffffffffu - (uint32_t)key;
      id = ids ? ids[pos] : (int64_t)pos;
      if (metric == DANN_METRIC_L2) dist = sqrtf(fmaxf(0.0f, qsumsq[q] - 2.0f * sc));
      else dist = 1.0f - sc;
    }
    out_dist[(size_t)q * k + i] = dist;
    out_ids[(size_t)q * k + i] = id;
  }
  if (threadIdx.x == 0) out_counts[q] = (int32_t)m;
}

// ---- exact mode: the fp32 originals beside the fp16 fragments, and a re-rank of the survivors in fp32 ----------------
// BruteForceIndex is an exact fp32 scan (ann/src/main/scala/com/twitter/ann/brute_force/BruteForceIndex.scala:66-91); the
// fp16 GEMM decides which few thousand vectors per query are worth a second look, this scores them as the reference
// would: fp32 operands (Cosine: normalised in fp32), fp32 accumulation.
__global__ void store_rows_kernel(const float *__restrict__ src, int64_t n, int d, int normalise, int64_t row0,
                                  float *__restrict__ x32, float *__restrict__ xss32) {
  int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= n) return;
  const float *x = src + row * d;
  double ss = 0;
  for (int k = lane; k < d; k += 64) ss += (double)x[k] * (double)x[k];
  ss = wave_sum(ss);
  float norm = 1.0f;
  if (normalise) {
    norm = (float)sqrt(ss);
    if (!(norm > 0.0f)) norm = 1.0f;
  }
  double ss2 = 0;
  for (int k = lane; k < d; k += 64) {
    const float v = x[k] / norm;
    x32[(row0 + row) * d + k] = v;
    ss2 += (double)v * (double)v;
  }
  ss2 = wave_sum(ss2);
  if (lane == 0) xss32[row0 + row] = (float)ss2;
}

__device__ __forceinline__ float wave_sum_f32(float v) {
  for (int o = 32; o; o >>= 1) v = v + __shfl_xor(v, o, 64);
  return v;
}

// one workgroup per query: every survivor's score again, from the fp32 rows
__global__ __launch_bounds__(256) void rescore_kernel(const float *__restrict__ q_in, int d, int metric, const float *__restrict__ x32,
                                                      const float *__restrict__ xss32, Survivor *__restrict__ surv,
                                                      const uint32_t *__restrict__ done_cnt, float *__restrict__ qsumsq) {
  __shared__ float sq[MAX_D];
  __shared__ double s_ss;
  const int q = blockIdx.x, t = threadIdx.x;
  double part = 0;
  for (int k = t; k < d; k += 256) { const float v = q_in[(size_t)q * d + k]; part += (double)v * (double)v; }
  part = wave_sum(part);
  if (t == 0) s_ss = 0;
  __syncthreads();
  if ((t & 63) == 0) atomicAdd(&s_ss, part);
  __syncthreads();
  float norm = 1.0f;
  if (metric == DANN_METRIC_COSINE) {
    norm = (float)sqrt(s_ss);
    if (!(norm > 0.0f)) norm = 1.0f;
  }
  double p2 = 0;
  for (int k = t; k < d; k += 256) { const float v = q_in[(size_t)q * d + k] / norm; sq[k] = v; p2 += (double)v * (double)v; }
  p2 = wave_sum(p2);
  __syncthreads();
  if (t == 0) s_ss = 0;
  __syncthreads();
  if ((t & 63) == 0) atomicAdd(&s_ss, p2);
  __syncthreads();
  if (t == 0) qsumsq[q] = (float)s_ss;  // |q|^2 of the fp32 query: what the L2 distance is rebuilt from
  const uint32_t c = min(done_cnt[q], (uint32_t)CAP);
  const int lane = t & 63, wv = t >> 6;
  for (uint32_t i = wv; i < c; i += 4) {
    Survivor s = surv[(size_t)q * CAP + i];
    const float *x = x32 + (size_t)s.pos * d;
    float acc = 0.0f;
    for (int k = lane; k < d; k += 64) acc = acc + sq[k] * x[k];
    acc = wave_sum_f32(acc);
    if (metric == DANN_METRIC_L2) acc = acc - 0.5f * xss32[s.pos];
    if (lane == 0) surv[(size_t)q * CAP + i].score = acc;
  }
}

// A vector the fp16 pass did not emit has fp16-operand score < tau_used, hence fp32 score < tau_used + delta; it cannot
// be among the k best if the k-th best rescored survivor is at least that.  delta bounds |fp32 score - fp16 score|:
// operands rounded to 11 bits each (2^-10 |x||q| in all) plus the accumulation error of either sum, and for L2 the bias
// -|x|^2/2 taken from the rounded row.  A query that fails the test is re-armed with a tau two deltas below its k-th.
__global__ __launch_bounds__(256) void prove_kernel(const Survivor *__restrict__ surv, uint32_t *__restrict__ done_cnt,
                                                    const float *__restrict__ tau_used, const float *__restrict__ qsumsq,
                                                    float max_norm, int metric, int k, float *__restrict__ tau,
                                                    uint32_t *__restrict__ cnt, int *__restrict__ status, int *__restrict__ flags) {
  __shared__ uint32_t hist[258];
  const int q = blockIdx.x;
  const uint32_t c = min(done_cnt[q], (uint32_t)CAP);
  if (c < (uint32_t)k) return;  // every stored vector was emitted (tau = -inf): nothing is hidden
  const float kth = wg_kth_largest(&surv[(size_t)q * CAP].score, c, 2, k, hist);
  if (threadIdx.x == 0) {
    const float qn = sqrtf(qsumsq[q]);
    const float delta = max_norm * qn * 1.05e-3f + (metric == DANN_METRIC_L2 ? max_norm * max_norm * 5.0e-4f : 0.0f);
    const float used = tau_used[q];
    if (used > -INFINITY && kth < used + delta) {
      tau[q] = kth - 2.0f * delta;
      cnt[q] = 0;
      done_cnt[q] = 0xffffffffu;
      status[q] = 1;
      atomicOr(&flags[0], 1);
    }
  }
}

struct Buf {
  void *p = nullptr;
  size_t bytes = 0;
  ~Buf() { if (p) (void)hipFree(p); }
  hipError_t reserve(size_t n) {
    if (n <= bytes) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&p, n);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  template <class T> T *as() const { return (T *)p; }
};

}  // namespace

struct dann_index {
  int device = 0, metric = 0, d = 0, S = 0, VB = 0;
  int64_t n = 0, n_pad = 0;
  Buf xf, bias, ids;
  Buf x32, xss32, tau_used;  // exact mode: fp32 rows [n][d] (Cosine: unit length), their squared norms, per-query thresholds
  bool has_ids = false, exact = false;
  float max_norm = 0.0f;     // exact mode: largest |x| of the stored rows
  // per-search scratch (grown on demand, reused)
  Buf q_in, qf, qsumsq, tmax, tau, cnt, done_cnt, surv, status, flags, o_dist, o_ids, o_cnt;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  float t_a = 0, t_b = 0, t_sel = 0;
  int last_rounds = 0;  // passes B of the last search (1 unless a query overflowed its buffer or, in exact mode, failed its proof)
  ~dann_index() {
    for (auto &e : ev)
      if (e) (void)hipEventDestroy(e);
  }
};

namespace {

int pick_geometry(int d, int *S, int *VB) {
  if (d < 1 || d > MAX_D) return fail(DANN_EINVAL, "dimension must be in 1..512");
  int s = 4;
  while (s * 16 < d) s <<= 1;
  *S = s;
  *VB = s == 32 ? 2 : 4;  // 32-vector blocks per wave: VB * S * 4 = 256 AGPRs of A fragments
  return DANN_OK;
}

int alloc_index(dann_index *ix, int device, int metric, int64_t n, int d) {
  if (metric < DANN_METRIC_L2 || metric > DANN_METRIC_INNER_PRODUCT) return fail(DANN_EINVAL, "unknown metric");
  if (n < 1 || n >= (int64_t)0xffffff00u) return fail(DANN_EINVAL, "vector count out of range");
  int rc = pick_geometry(d, &ix->S, &ix->VB);
  if (rc) return rc;
  DTRY(hipSetDevice(device));
  ix->device = device;
  ix->metric = metric;
  ix->d = d;
  ix->n = n;
  const int64_t tile = (int64_t)W2 * ix->VB * 32;
  ix->n_pad = (n + tile - 1) / tile * tile;
  DTRY(ix->xf.reserve((size_t)ix->n_pad * ix->S * 16 * sizeof(_Float16)));
  DTRY(ix->bias.reserve((size_t)ix->n_pad * sizeof(float)));
  DTRY(hipMemset(ix->xf.p, 0, ix->xf.bytes));
  for (auto &e : ix->ev) DTRY(hipEventCreate(&e));
  return DANN_OK;
}

template <int S, int VB>
int launch_gemm2(bool emit, bool bias, const GemmArgs &a, hipStream_t st) {
  const size_t lds = (size_t)3 * S * 64 * 16 + (size_t)W2 * VB * 8 * 16 + 16 + (size_t)SB_CAP * sizeof(Staged) +
                     (size_t)a.nqb * 32 * 4;
  const void *fn = emit ? (bias ? (const void *)gemm2_kernel<S, VB, true, true> : (const void *)gemm2_kernel<S, VB, true, false>)
                        : (bias ? (const void *)gemm2_kernel<S, VB, false, true> : (const void *)gemm2_kernel<S, VB, false, false>);
  DTRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  void *args[] = {(void *)&a};
  DTRY(hipLaunchKernel(fn, dim3(a.n_wg_launch), dim3(W2 * 64), args, lds, st));
  return DANN_OK;
}

int launch_gemm_any(int S, bool emit, bool bias, const GemmArgs &a, hipStream_t st) {
  switch (S) {
    case 4: return launch_gemm2<4, 4>(emit, bias, a, st);
    case 8: return launch_gemm2<8, 4>(emit, bias, a, st);
    case 16: return launch_gemm2<16, 4>(emit, bias, a, st);
    default: return launch_gemm2<32, 2>(emit, bias, a, st);
  }
}

}  // namespace

extern "C" {

const char *dann_last_error(void) { return g_err.c_str(); }

static int build_impl(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids, bool exact,
                      dann_index_t **out);

int dann_index_build(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids,
                     dann_index_t **out) try {
  return build_impl(device, metric, n, d, vectors, ids, false, out);
} ABI_CATCH

int dann_index_build_exact(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids,
                           dann_index_t **out) try {
  return build_impl(device, metric, n, d, vectors, ids, true, out);
} ABI_CATCH

static int build_impl(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids, bool exact,
                      dann_index_t **out) {
  if (!vectors || !out) return fail(DANN_EINVAL, "null argument");
  std::unique_ptr<dann_index> ix(new dann_index);
  int rc = alloc_index(ix.get(), device, metric, n, d);
  if (rc) return rc;
  // positions follow id order, so that "position ascending" is "id ascending" in every tie
  std::vector<int64_t> order;
  if (ids) {
    order.resize((size_t)n);
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return ids[x] < ids[y]; });
    std::vector<int64_t> sorted((size_t)n);
    for (int64_t i = 0; i < n; ++i) sorted[(size_t)i] = ids[order[(size_t)i]];
    DTRY(ix->ids.reserve((size_t)n * sizeof(int64_t)));
    DTRY(hipMemcpy(ix->ids.p, sorted.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice));
    ix->has_ids = true;
  }
  const int64_t chunk = std::max<int64_t>(1, (int64_t)(256u << 20) / ((int64_t)d * 4));
  Buf stage;
  DTRY(stage.reserve((size_t)std::min(chunk, n) * d * sizeof(float)));
  if (exact) {
    DTRY(ix->x32.reserve((size_t)n * d * sizeof(float)));
    DTRY(ix->xss32.reserve((size_t)n * sizeof(float)));
  }
  std::vector<float> gathered;
  for (int64_t r0 = 0; r0 < n; r0 += chunk) {
    int64_t m = std::min(chunk, n - r0);
    const float *src = vectors + r0 * d;
    if (ids) {
      gathered.resize((size_t)m * d);
      for (int64_t i = 0; i < m; ++i)
        std::memcpy(&gathered[(size_t)i * d], vectors + order[(size_t)(r0 + i)] * d, (size_t)d * sizeof(float));
      src = gathered.data();
    }
    DTRY(hipMemcpy(stage.p, src, (size_t)m * d * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(prep_rows_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, 0, stage.as<float>(), m, d, ix->S,
                       metric == DANN_METRIC_COSINE ? 1 : 0, r0, ix->xf.as<_Float16>(), ix->bias.as<float>());
    DTRY(hipGetLastError());
    if (exact) {
      hipLaunchKernelGGL(store_rows_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, 0, stage.as<float>(), m, d,
                         metric == DANN_METRIC_COSINE ? 1 : 0, r0, ix->x32.as<float>(), ix->xss32.as<float>());
      DTRY(hipGetLastError());
    }
    DTRY(hipDeviceSynchronize());
  }
  if (exact) {
    std::vector<float> ss((size_t)n);
    DTRY(hipMemcpy(ss.data(), ix->xss32.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    float mx = 0.0f;
    for (float v : ss) mx = std::max(mx, v);
    ix->max_norm = std::sqrt(mx) * 1.0001f;
    ix->exact = true;
  }
  hipLaunchKernelGGL(bias_kernel, dim3((unsigned)((ix->n_pad + 255) / 256)), dim3(256), 0, 0, ix->bias.as<float>(), n,
                     ix->n_pad, metric);
  DTRY(hipGetLastError());
  DTRY(hipDeviceSynchronize());
  *out = ix.release();
  return DANN_OK;
}

int dann_index_build_synthetic(int32_t device, int32_t metric, int64_t n, int32_t d, uint64_t seed, dann_index_t **out) try {
  if (!out) return fail(DANN_EINVAL, "null argument");
  std::unique_ptr<dann_index> ix(new dann_index);
  int rc = alloc_index(ix.get(), device, metric, n, d);
  if (rc) return rc;
  hipLaunchKernelGGL(synth_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ix->xf.as<_Float16>(),
                     ix->bias.as<float>(), n, d, ix->S, metric, seed);
  DTRY(hipGetLastError());
  hipLaunchKernelGGL(bias_kernel, dim3((unsigned)((ix->n_pad + 255) / 256)), dim3(256), 0, 0, ix->bias.as<float>(), n,
                     ix->n_pad, metric);
  DTRY(hipGetLastError());
  DTRY(hipDeviceSynchronize());
  *out = ix.release();
  return DANN_OK;
} ABI_CATCH

int dann_index_get_vectors(const dann_index_t *ix, int64_t i0, int64_t n, float *out) try {
  if (!ix || !out || i0 < 0 || n < 0 || i0 + n > ix->n) return fail(DANN_EINVAL, "range outside the index");
  if (n == 0) return DANN_OK;
  DTRY(hipSetDevice(ix->device));
  // in slabs of <= 2^28 elements: a launch must stay below 2^32 work-items (hnsw_index_get_vectors found out at 50M x 256)
  const int64_t slab = std::max<int64_t>(1, ((int64_t)1 << 28) / ix->d);
  Buf tmp;
  DTRY(tmp.reserve((size_t)std::min(slab, n) * ix->d * sizeof(float)));
  for (int64_t r0 = 0; r0 < n; r0 += slab) {
    const int64_t m = std::min(slab, n - r0), e = m * ix->d;
    hipLaunchKernelGGL(unfrag_kernel, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, 0, ix->xf.as<_Float16>(), ix->S, ix->d,
                       i0 + r0, m, tmp.as<float>());
    DTRY(hipGetLastError());
    DTRY(hipMemcpy(out + (size_t)r0 * ix->d, tmp.p, (size_t)e * sizeof(float), hipMemcpyDeviceToHost));
  }
  return DANN_OK;
} ABI_CATCH

int dann_index_destroy(dann_index_t *ix) try {
  delete ix;
  return DANN_OK;
} ABI_CATCH

static int search_chunk(dann_index_t *ix, int32_t nq, const float *queries, int32_t k, float *out_dist, int64_t *out_ids,
                        int32_t *out_counts);

int dann_search(dann_index_t *ix, int32_t nq, const float *queries, int32_t k, float *out_dist, int64_t *out_ids,
                int32_t *out_counts) try {
  if (!ix || !queries || !out_dist || !out_ids || !out_counts) return fail(DANN_EINVAL, "null argument");
  if (nq < 1) return fail(DANN_EINVAL, "nq must be positive");
  if (k < 1 || k > MAX_K) return fail(DANN_EINVAL, "k must be in 1..1024");
  float ta = 0, tb = 0, ts = 0;
  for (int32_t q0 = 0; q0 < nq; q0 += MAX_NQ) {  // the GEMM keeps one threshold per query in LDS
    int32_t m = std::min<int32_t>(MAX_NQ, nq - q0);
    int rc = search_chunk(ix, m, queries + (size_t)q0 * ix->d, k, out_dist + (size_t)q0 * k, out_ids + (size_t)q0 * k,
                          out_counts + q0);
    if (rc) return rc;
    ta += ix->t_a;
    tb += ix->t_b;
    ts += ix->t_sel;
  }
  ix->t_a = ta;
  ix->t_b = tb;
  ix->t_sel = ts;
  return DANN_OK;
} ABI_CATCH

static int search_chunk(dann_index_t *ix, int32_t nq, const float *queries, int32_t k, float *out_dist, int64_t *out_ids,
                        int32_t *out_counts) {
  DTRY(hipSetDevice(ix->device));
  const int S = ix->S, d = ix->d;
  const int nqb = (nq + 31) / 32, nq_pad = nqb * 32;
  const int64_t wg_vecs = (int64_t)W2 * ix->VB * 32;
  const uint32_t n_wg = (uint32_t)(ix->n_pad / wg_vecs);

  // tiles without padding rows run without the per-vector bias for InnerProduct / Cosine (the
  // accumulators start at the inline constant 0); the last, partial tile always takes the bias
  // path, whose -inf keeps the padding rows out of everything.
  const bool l2 = ix->metric == DANN_METRIC_L2;
  const uint32_t n_full = (uint32_t)(ix->n / wg_vecs);
  const bool tail = n_wg > n_full;
  // pass-A sample (full tiles only).  A larger sample costs GEMM time (2k / E of a full pass) but gives a
  // tighter tau, i.e. fewer survivors E in pass B, where a survivor takes its wave out of the MFMA stream
  // for ~1000 cycles (and its three neighbours wait for it at the next barrier).  Measured at k = 10 on
  // 50M vectors: E = 2048 -> 0.33 + 21.9 ms, E = 405 -> 1.43 + 20.95 ms; the sum is flat in between.
  const int e_target = (int)std::min(4096.0, std::max(512.0, 256.0 * std::sqrt((double)k)));
  uint32_t n_swg = 0;
  static const bool no_sample = getenv("DANN_NO_SAMPLE") != nullptr;  // debugging: force the refinement path
  if ((int64_t)n_full * TILE_MAXIMA >= k && !no_sample) {
    uint64_t want = ((uint64_t)2 * k * n_wg + e_target - 1) / e_target;
    uint64_t floor_wg = ((uint64_t)4 * std::max(k, 64) + TILE_MAXIMA - 1) / TILE_MAXIMA;
    n_swg = (uint32_t)std::min<uint64_t>(n_full, std::max(want, floor_wg));
  }
  const int64_t pitch = (int64_t)n_swg * TILE_MAXIMA;

  DTRY(ix->q_in.reserve((size_t)nq * d * sizeof(float)));
  DTRY(ix->qf.reserve((size_t)nq_pad * S * 16 * sizeof(_Float16)));
  DTRY(ix->qsumsq.reserve((size_t)nq_pad * sizeof(float)));
  DTRY(ix->tmax.reserve((size_t)nq_pad * std::max<int64_t>(pitch, 1) * sizeof(float)));
  DTRY(ix->tau.reserve((size_t)nq_pad * sizeof(float)));
  DTRY(ix->cnt.reserve((size_t)nq_pad * sizeof(uint32_t)));
  DTRY(ix->done_cnt.reserve((size_t)nq_pad * sizeof(uint32_t)));
  DTRY(ix->status.reserve((size_t)nq_pad * sizeof(int)));
  DTRY(ix->flags.reserve(sizeof(int)));
  DTRY(ix->surv.reserve((size_t)nq_pad * CAP * sizeof(Survivor)));
  DTRY(ix->o_dist.reserve((size_t)nq * k * sizeof(float)));
  DTRY(ix->o_ids.reserve((size_t)nq * k * sizeof(int64_t)));
  DTRY(ix->o_cnt.reserve((size_t)nq * sizeof(int32_t)));

  hipStream_t st = 0;
  DTRY(hipMemcpyAsync(ix->q_in.p, queries, (size_t)nq * d * sizeof(float), hipMemcpyHostToDevice, st));
  DTRY(hipMemsetAsync(ix->qf.p, 0, (size_t)nq_pad * S * 16 * sizeof(_Float16), st));
  DTRY(hipMemsetAsync(ix->status.p, 0, (size_t)nq_pad * sizeof(int), st));
  DTRY(hipMemsetAsync(ix->done_cnt.p, 0xff, (size_t)nq_pad * sizeof(uint32_t), st));
  hipLaunchKernelGGL(prep_rows_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st, ix->q_in.as<float>(), (int64_t)nq, d,
                     S, ix->metric == DANN_METRIC_COSINE ? 1 : 0, (int64_t)0, ix->qf.as<_Float16>(), ix->qsumsq.as<float>());
  DTRY(hipGetLastError());

  GemmArgs a;
  a.xf = ix->xf.as<_Float16>();
  a.bias = ix->bias.as<float>();
  a.qf = ix->qf.as<_Float16>();
  a.nqb = nqb;
  a.n = ix->n;
  a.n_wg_total = n_full;
  a.tile0 = 0;
  a.col0 = 0;
  a.tmax = ix->tmax.as<float>();
  a.pitch = pitch;
  a.tau = ix->tau.as<float>();
  a.cnt = ix->cnt.as<uint32_t>();
  a.surv = ix->surv.as<Survivor>();

  DTRY(hipEventRecord(ix->ev[0], st));
  if (n_swg) {
    a.n_wg_launch = n_swg;
    int rc = launch_gemm_any(S, false, l2, a, st);
    if (rc) return rc;
    DTRY(hipGetLastError());
  }
  hipLaunchKernelGGL(tau_kernel, dim3(nq_pad), dim3(256), 0, st, ix->tmax.as<float>(), pitch, pitch, k, nq,
                     ix->tau.as<float>(), ix->cnt.as<uint32_t>(), ix->qsumsq.as<float>(), ix->exact ? ix->max_norm : 0.0f, ix->metric);
  DTRY(hipGetLastError());
  DTRY(hipEventRecord(ix->ev[1], st));

  if (ix->exact) DTRY(ix->tau_used.reserve((size_t)nq_pad * sizeof(float)));
  float *const d_tau_used = ix->exact ? ix->tau_used.as<float>() : nullptr;
  int flags = 0;
  for (int round = 0;; ++round) {
    if (n_full) {
      a.n_wg_total = a.n_wg_launch = n_full;
      a.tile0 = 0;
      int rc = launch_gemm_any(S, true, l2, a, st);
      if (rc) return rc;
      DTRY(hipGetLastError());
    }
    if (tail) {
      a.n_wg_total = a.n_wg_launch = 1;
      a.tile0 = n_full;
      int rc = launch_gemm_any(S, true, true, a, st);
      if (rc) return rc;
      DTRY(hipGetLastError());
    }
    if (round == 0) DTRY(hipEventRecord(ix->ev[2], st));
    DTRY(hipMemsetAsync(ix->flags.p, 0, sizeof(int), st));
    hipLaunchKernelGGL(refine_kernel, dim3(nq), dim3(256), 0, st, ix->tau.as<float>(), ix->cnt.as<uint32_t>(),
                       ix->done_cnt.as<uint32_t>(), ix->surv.as<Survivor>(), k, nq, ix->status.as<int>(), ix->flags.as<int>(),
                       d_tau_used);
    DTRY(hipGetLastError());
    DTRY(hipMemcpyAsync(&flags, ix->flags.p, sizeof(int), hipMemcpyDeviceToHost, st));
    DTRY(hipStreamSynchronize(st));
    if (flags & 2) return fail(DANN_ELIMIT, "more than 8192 stored vectors tie at the k-th distance of a query");
    if (round >= 16) return fail(DANN_ELIMIT, "threshold refinement did not converge");
    if (flags & 1) {
      continue;
    }
    if (!ix->exact) {
      ix->last_rounds = round + 1;
      break;
    }
    // exact mode: survivors scored again in fp32, then the proof that nothing below the fp16 threshold could matter
    hipLaunchKernelGGL(rescore_kernel, dim3(nq), dim3(256), 0, st, ix->q_in.as<float>(), d, ix->metric, ix->x32.as<float>(),
                       ix->xss32.as<float>(), ix->surv.as<Survivor>(), ix->done_cnt.as<uint32_t>(), ix->qsumsq.as<float>());
    DTRY(hipGetLastError());
    DTRY(hipMemsetAsync(ix->flags.p, 0, sizeof(int), st));
    const float delta_scale = getenv("DANN_DEBUG_DELTA_SCALE") ? (float)atof(getenv("DANN_DEBUG_DELTA_SCALE")) : 1.0f;  // tests: force the re-arm path
    hipLaunchKernelGGL(prove_kernel, dim3(nq), dim3(256), 0, st, ix->surv.as<Survivor>(), ix->done_cnt.as<uint32_t>(), d_tau_used,
                       ix->qsumsq.as<float>(), ix->max_norm * delta_scale, ix->metric, k, ix->tau.as<float>(), ix->cnt.as<uint32_t>(),
                       ix->status.as<int>(), ix->flags.as<int>());
    DTRY(hipGetLastError());
    DTRY(hipMemcpyAsync(&flags, ix->flags.p, sizeof(int), hipMemcpyDeviceToHost, st));
    DTRY(hipStreamSynchronize(st));
    if (!(flags & 1)) {  // every query proven
      ix->last_rounds = round + 1;
      break;
    }
  }

  DTRY(hipFuncSetAttribute((const void *)select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                           CAP * sizeof(unsigned long long)));
  hipLaunchKernelGGL(select_kernel, dim3(nq), dim3(512), CAP * sizeof(unsigned long long), st, ix->surv.as<Survivor>(),
                     ix->done_cnt.as<uint32_t>(), ix->qsumsq.as<float>(), ix->has_ids ? ix->ids.as<int64_t>() : nullptr,
                     ix->metric, k, ix->o_dist.as<float>(), ix->o_ids.as<int64_t>(), ix->o_cnt.as<int32_t>());
  DTRY(hipGetLastError());
  DTRY(hipEventRecord(ix->ev[3], st));
  DTRY(hipMemcpyAsync(out_dist, ix->o_dist.p, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, st));
  DTRY(hipMemcpyAsync(out_ids, ix->o_ids.p, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  DTRY(hipMemcpyAsync(out_counts, ix->o_cnt.p, (size_t)nq * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  DTRY(hipStreamSynchronize(st));
  (void)hipEventElapsedTime(&ix->t_a, ix->ev[0], ix->ev[1]);
  (void)hipEventElapsedTime(&ix->t_b, ix->ev[1], ix->ev[2]);
  (void)hipEventElapsedTime(&ix->t_sel, ix->ev[2], ix->ev[3]);
  return DANN_OK;
}

int dann_last_rounds(const dann_index_t *ix, int32_t *rounds) try {
  if (!ix || !rounds) return fail(DANN_EINVAL, "null argument");
  *rounds = ix->last_rounds;
  return DANN_OK;
} ABI_CATCH

int dann_last_timing(const dann_index_t *ix, float *a_ms, float *b_ms, float *sel_ms) try {
  if (!ix) return fail(DANN_EINVAL, "null index");
  if (a_ms) *a_ms = ix->t_a;
  if (b_ms) *b_ms = ix->t_b;
  if (sel_ms) *sel_ms = ix->t_sel;
  return DANN_OK;
} ABI_CATCH

int dann_compose_shards(int32_t n_shards, int32_t nq, int32_t k_in, const int64_t *ids, const float *dist, const int32_t *counts,
                        int32_t k, int64_t *out_ids, float *out_dist, int32_t *out_counts) try {
  if (n_shards < 1 || nq < 0 || k_in < 0 || k < 0) return fail(DANN_EINVAL, "n_shards >= 1, nq / k_in / k >= 0");
  if (nq > 0 && (!counts || !out_counts || (k_in > 0 && (!ids || !dist)) || (k > 0 && (!out_ids || !out_dist))))
    return fail(DANN_EINVAL, "NULL argument");
  std::vector<std::pair<float, int64_t>> all;
  for (int32_t q = 0; q < nq; ++q) {
    all.clear();
    for (int32_t s = 0; s < n_shards; ++s) {
      const int32_t c = counts[(size_t)s * nq + q];
      if (c < 0 || c > k_in) return fail(DANN_EINVAL, "a shard's count is outside 0..k_in");
      const size_t base = ((size_t)s * nq + q) * (size_t)k_in;
      for (int32_t j = 0; j < c; ++j) all.emplace_back(dist[base + j], ids[base + j]);
    }
    std::sort(all.begin(), all.end(), [](const std::pair<float, int64_t> &a, const std::pair<float, int64_t> &b) {
      return a.first < b.first || (a.first == b.first && a.second < b.second);
    });
    const int32_t m = (int32_t)std::min<size_t>(all.size(), (size_t)k);
    for (int32_t j = 0; j < m; ++j) {
      out_dist[(size_t)q * k + j] = all[(size_t)j].first;
      out_ids[(size_t)q * k + j] = all[(size_t)j].second;
    }
    out_counts[q] = m;
  }
  return DANN_OK;
} ABI_CATCH

}  // extern "C"
