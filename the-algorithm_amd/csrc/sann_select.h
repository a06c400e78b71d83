// sann_select.h -- wave-parallel pieces of the radix threshold search (device only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sann {

// hist[0..256) are digit counts.  Find the largest digit d such that the number of entries with
// digit >= d is at least `need`; A = entries with digit > d, B = hist[d].  If fewer than `need`
// entries exist, d = 0.  Must be called by all 64 lanes of ONE wave (lanes 0..63 of the
// workgroup); the result is written to out[0..2] (LDS) by one lane.  A serial scan by a single
// thread would be 256 dependent LDS reads (~25k cycles); this is 4 reads per lane + 6 shuffles.
__device__ inline void wave_find_digit(const unsigned *hist, int need, int *out) {
  const int lane = threadIdx.x & 63;
  // lane 0 holds the four highest digits, lane 63 the four lowest
  const int top = 255 - 4 * lane;
  const int c0 = (int)hist[top], c1 = (int)hist[top - 1], c2 = (int)hist[top - 2], c3 = (int)hist[top - 3];
  const int s = c0 + c1 + c2 + c3;
  int incl = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  const int excl = incl - s;
  const bool cross = excl < need && incl >= need;
  const unsigned long long m = __ballot(cross);
  const int total = __shfl(incl, 63, 64);
  if (m == 0) {
    if (lane == 0) { out[0] = 0; out[1] = total - (int)hist[0]; out[2] = (int)hist[0]; }
    return;
  }
  if (cross) {
    int cum = excl, d = top, B = c0;
    if (cum + c0 >= need) { d = top; B = c0; }
    else if (cum + c0 + c1 >= need) { cum += c0; d = top - 1; B = c1; }
    else if (cum + c0 + c1 + c2 >= need) { cum += c0 + c1; d = top - 2; B = c2; }
    else { cum += c0 + c1 + c2; d = top - 3; B = c3; }
    out[0] = d;
    out[1] = cum;
    out[2] = B;
  }
}

}  // namespace sann
