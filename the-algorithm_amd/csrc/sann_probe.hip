// sann_probe.hip -- MEASUREMENT ONLY: the unit kernel's memory side on its own.
//
// Same grid, same descriptors, same (query, partition) -> posting gather as unit_fast_kernel, but the postings are
// only folded into a checksum.  What it answers: how long does the gather of a batch take (a) in the unit kernel's
// structure -- one workgroup per unit, three dependent trips to memory (header -> descriptor row -> postings) -- and
// (b) from persistent workgroups that keep the NEXT unit's postings in flight while the current unit is consumed.
// The difference between (a) and the unit kernel is what the arithmetic costs; the difference between (a) and (b) is
// what the dependent trips cost.  Nothing here is on the product path (sann_debug_gather_probe).
#include <hip/hip_runtime.h>

#include "sann_device.h"
#include "sann_kernels.h"

namespace sann {

namespace {
constexpr int NSCAN_MAX_P = 128;

template <int WG, int U>
__device__ inline void build_map(const IndexView &ix, const BatchView &b, int q, int p, const QueryHdr &h, uint32_t *s_begin,
                                 uint32_t *s_pre, uint8_t *s_map, uint32_t T) {
  const uint32_t *d = b.desc + 2 * ((int64_t)q * ix.P + p) * b.desc_stride;  // (fixed-stride rows; the first n_scan entries are real)
  for (int t = threadIdx.x; t < 4 * h.n_scan; t += WG) {
    const int c = t >> 2, part = t & 3;
    const uint2 v = *reinterpret_cast<const uint2 *>(d + 2 * c);
    const bool last = c + 1 >= h.n_scan;
    uint32_t next = last ? T : d[2 * (c + 1) + 1];
    if (part == 0) {
      s_begin[c] = v.x;
      s_pre[c] = v.y;
    }
    next = next < (uint32_t)(WG * U) ? next : (uint32_t)(WG * U);
    for (uint32_t i = v.y + part; i < next; i += 4) s_map[i] = (uint8_t)c;
  }
}

__device__ inline unsigned long long wave_xor(unsigned long long v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v ^= __shfl_xor(v, off, 64);
  return v;
}
}  // namespace

// (a) one workgroup per unit, as the unit kernel
template <int WG, int U>
__global__ __launch_bounds__(WG) void probe_unit_kernel(IndexView ix, BatchView b, unsigned long long *out) {
  __shared__ uint32_t s_begin[NSCAN_MAX_P], s_pre[NSCAN_MAX_P];
  __shared__ uint8_t s_map[WG * U];
  const int tid = threadIdx.x;
  const int blk = blockIdx.x;
  const int x = blk & 7, r = blk >> 3;
  const int p = r & (ix.P - 1);
  const int q = ((r >> ix.log2P) << 3) + x;
  if (q >= b.nq) return;
  const int unit = q * ix.P + p;
  const QueryHdr h = b.hdr[q];
  if (h.n_scan > NSCAN_MAX_P) return;
  uint32_t T = (uint32_t)b.unit_T[unit];
  T = T < (uint32_t)(WG * U) ? T : (uint32_t)(WG * U);
  build_map<WG, U>(ix, b, q, p, h, s_begin, s_pre, s_map, T);
  __syncthreads();
  Posting pst[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const uint32_t j = (uint32_t)(u * WG + tid);
    const uint32_t jj = j < T ? j : (T ? T - 1 : 0);
    const int c = (int)s_map[jj];
    pst[u] = T ? ix.postings[s_begin[c] + (jj - s_pre[c])] : Posting{0, 0.0};
  }
  unsigned long long acc = 0;
#pragma unroll
  for (int u = 0; u < U; u++)
    if ((uint32_t)(u * WG + tid) < T) acc ^= (unsigned long long)pst[u].id ^ (unsigned long long)__double_as_longlong(pst[u].score);
  acc = wave_xor(acc);
  if ((tid & 63) == 0) atomicXor(&out[unit], acc);
}

// (b) persistent workgroups: unit i+1's postings are loaded (into a second register set) before unit i is consumed;
// unit i+2's descriptor data is already on its way.
template <int WG, int U>
__global__ __launch_bounds__(WG) void probe_persistent_kernel(IndexView ix, BatchView b, unsigned long long *out, int n_blocks) {
  __shared__ uint32_t s_begin[2][NSCAN_MAX_P], s_pre[2][NSCAN_MAX_P];
  __shared__ uint8_t s_map[2][WG * U];
  const int tid = threadIdx.x;
  Posting cur[U], nxt[U];
  uint32_t T_cur = 0, T_nxt = 0;
  int unit_cur = -1, unit_nxt = -1;
  int par = 0;
  // iteration it loads unit it (into nxt) and consumes unit it-1 (cur)
  for (int blk = blockIdx.x;; blk += gridDim.x) {
    const bool have = blk < n_blocks;
    unit_nxt = -1;
    T_nxt = 0;
    if (have) {
      const int x = blk & 7, r = blk >> 3;
      const int p = r & (ix.P - 1);
      const int q = ((r >> ix.log2P) << 3) + x;
      if (q < b.nq) {
        const QueryHdr h = b.hdr[q];
        if (h.n_scan <= NSCAN_MAX_P) {
          unit_nxt = q * ix.P + p;
          uint32_t T = (uint32_t)b.unit_T[unit_nxt];
          T = T < (uint32_t)(WG * U) ? T : (uint32_t)(WG * U);
          T_nxt = T;
          build_map<WG, U>(ix, b, q, p, h, s_begin[par], s_pre[par], s_map[par], T);
        }
      }
    }
    __syncthreads();
    if (unit_nxt >= 0 && T_nxt) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t j = (uint32_t)(u * WG + tid);
        const uint32_t jj = j < T_nxt ? j : T_nxt - 1;
        const int c = (int)s_map[par][jj];
        nxt[u] = ix.postings[s_begin[par][c] + (jj - s_pre[par][c])];
      }
    }
    // consume the previous unit while the loads above are in flight
    if (unit_cur >= 0) {
      unsigned long long acc = 0;
#pragma unroll
      for (int u = 0; u < U; u++)
        if ((uint32_t)(u * WG + tid) < T_cur) acc ^= (unsigned long long)cur[u].id ^ (unsigned long long)__double_as_longlong(cur[u].score);
      acc = wave_xor(acc);
      if ((tid & 63) == 0) atomicXor(&out[unit_cur], acc);
    }
    if (!have) break;
#pragma unroll
    for (int u = 0; u < U; u++) cur[u] = nxt[u];
    unit_cur = unit_nxt;
    T_cur = T_nxt;
    par ^= 1;
  }
}

hipError_t launch_gather_probe(const IndexView &ix, const BatchView &b, int unit_capacity, int mode, int wgs_per_cu,
                               unsigned long long *out, hipStream_t stream) {
  const int nq8 = (b.nq + 7) / 8 * 8;
  const int n_blocks = nq8 * ix.P;
  if (n_blocks <= 0) return hipSuccess;
  const int grid_p = ((256 * wgs_per_cu + 7) / 8) * 8;
#define PROBE(WG, U)                                                                                                  \
  do {                                                                                                                \
    if (mode == 0) hipLaunchKernelGGL((probe_unit_kernel<WG, U>), dim3(n_blocks), dim3(WG), 0, stream, ix, b, out);     \
    else hipLaunchKernelGGL((probe_persistent_kernel<WG, U>), dim3(grid_p < n_blocks ? grid_p : n_blocks), dim3(WG), 0, stream, ix, b, out, n_blocks); \
  } while (0)
  switch (unit_capacity) {
    case 768: PROBE(256, 3); break;
    case 1024: PROBE(256, 4); break;
    case 1536: PROBE(256, 6); break;
    case 2048: PROBE(256, 8); break;
    default: return hipErrorInvalidValue;
  }
#undef PROBE
  return hipGetLastError();
}

}  // namespace sann
