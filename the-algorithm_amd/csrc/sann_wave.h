// sann_wave.h -- wave64 primitives on the VALU for gfx950: DPP reductions / scans, lane exchange without the LDS
// crossbar, and in-register bitonic sorts (one value per lane).  Shared by the unit kernel and the merge kernel.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

// The lane number the helpers below branch on.  A kernel whose body is one big LOOP can define it, before this header is
// included, as something the optimiser cannot prove loop-invariant: the dozens of lane predicates of the sort networks are
// otherwise hoisted out of the loop and held in SGPR pairs for the kernel's whole life (round 3's pipelined unit kernel: 97
// spilled SGPRs; profiles/r03_pipelined_unit_kernel_experiment.txt).
#ifndef SANN_WAVE_LANE
#define SANN_WAVE_LANE() ((int)(threadIdx.x & 63))
#endif

namespace sann {

// ---------------------------------------------------------------------------------------------
// wave64 reductions on the VALU (DPP), result valid in every lane after the readlane
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ inline uint32_t dpp_u32(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ inline uint32_t wave_max_u32(uint32_t v) {
  v = max(v, dpp_u32<0xB1, 0xf>(0u, v));   // quad_perm [1,0,3,2]
  v = max(v, dpp_u32<0x4E, 0xf>(0u, v));   // quad_perm [2,3,0,1]
  v = max(v, dpp_u32<0x124, 0xf>(0u, v));  // row_ror 4
  v = max(v, dpp_u32<0x128, 0xf>(0u, v));  // row_ror 8
  v = max(v, dpp_u32<0x142, 0xa>(0u, v));  // row_bcast 15 -> rows 1,3
  v = max(v, dpp_u32<0x143, 0xc>(0u, v));  // row_bcast 31 -> rows 2,3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ inline uint32_t wave_min_u32(uint32_t v) { return ~wave_max_u32(~v); }
// inclusive prefix sum over the wave, on the VALU (DPP row shifts, then the two row broadcasts)
__device__ inline int wave_incl_scan_i32(int x) {
  uint32_t v = (uint32_t)x;
  v += dpp_u32<0x111, 0xf>(0u, v);  // row_shr:1
  v += dpp_u32<0x112, 0xf>(0u, v);  // row_shr:2
  v += dpp_u32<0x114, 0xf>(0u, v);  // row_shr:4
  v += dpp_u32<0x118, 0xf>(0u, v);  // row_shr:8
  v += dpp_u32<0x142, 0xa>(0u, v);  // row_bcast:15 -> rows 1, 3
  v += dpp_u32<0x143, 0xc>(0u, v);  // row_bcast:31 -> rows 2, 3
  return (int)v;
}

// Value of lane (l ^ J), without a trip through the LDS crossbar (a chain of 21 ds_bpermute round trips cost the
// sort below ~6k cycles of latency): quad permutes, row shifts, and gfx950's row / half-wave swaps.
template <int J>
__device__ inline uint32_t lane_xor(uint32_t v) {
  const int lane = SANN_WAVE_LANE();
  if constexpr (J == 1) return dpp_u32<0xB1, 0xf>(v, v);  // quad_perm [1,0,3,2]
  else if constexpr (J == 2) return dpp_u32<0x4E, 0xf>(v, v);  // quad_perm [2,3,0,1]
  else if constexpr (J == 4 || J == 8) {
    const uint32_t up = dpp_u32<0x100 + J, 0xf>(v, v);    // row_shl:J  -> lane l reads l + J
    const uint32_t down = dpp_u32<0x110 + J, 0xf>(v, v);  // row_shr:J  -> lane l reads l - J
    return (lane & J) ? down : up;
  } else if constexpr (J == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // r[0] = rows (0,0,2,2), r[1] = rows (1,1,3,3)
    return (lane & 16) ? r[0] : r[1];
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // r[0] = halves (lo,lo), r[1] = (hi,hi)
    return (lane & 32) ? r[0] : r[1];
  }
}
template <int K, int J>
__device__ inline uint32_t bitonic_step(uint32_t v) {
  const int lane = SANN_WAVE_LANE();
  const uint32_t o = lane_xor<J>(v);
  const bool keep_max = ((lane & K) == 0) == ((lane & J) == 0);
  return keep_max ? (v > o ? v : o) : (v < o ? v : o);
}
// bitonic sort of one value per lane, descending: lane i ends up with the wave's i-th largest
__device__ inline uint32_t wave_sort_desc_u32(uint32_t v) {
  v = bitonic_step<2, 1>(v);
  v = bitonic_step<4, 2>(v); v = bitonic_step<4, 1>(v);
  v = bitonic_step<8, 4>(v); v = bitonic_step<8, 2>(v); v = bitonic_step<8, 1>(v);
  v = bitonic_step<16, 8>(v); v = bitonic_step<16, 4>(v); v = bitonic_step<16, 2>(v); v = bitonic_step<16, 1>(v);
  v = bitonic_step<32, 16>(v); v = bitonic_step<32, 8>(v); v = bitonic_step<32, 4>(v); v = bitonic_step<32, 2>(v); v = bitonic_step<32, 1>(v);
  v = bitonic_step<64, 32>(v); v = bitonic_step<64, 16>(v); v = bitonic_step<64, 8>(v); v = bitonic_step<64, 4>(v); v = bitonic_step<64, 2>(v); v = bitonic_step<64, 1>(v);
  return v;
}


// ---- 128-bit keys (hi, lo), one per lane: the merge kernel's (score key, id key) ---------------------------------------
template <int J>
__device__ inline uint64_t lane_xor_u64(uint64_t v) {
  const uint32_t a = lane_xor<J>((uint32_t)v), b = lane_xor<J>((uint32_t)(v >> 32));
  return ((uint64_t)b << 32) | a;
}
template <int K, int J>
__device__ inline void bitonic_step_k128(uint64_t &hi, uint64_t &lo) {
  const int lane = SANN_WAVE_LANE();
  const uint64_t ohi = lane_xor_u64<J>(hi), olo = lane_xor_u64<J>(lo);
  const bool keep_max = ((lane & K) == 0) == ((lane & J) == 0);
  const bool o_gt = ohi > hi || (ohi == hi && olo > lo);
  const bool o_lt = ohi < hi || (ohi == hi && olo < lo);
  const bool take = keep_max ? o_gt : o_lt;
  hi = take ? ohi : hi;
  lo = take ? olo : lo;
}
// bitonic sort of one 128-bit key per lane, descending: lane i ends up with the wave's i-th largest
__device__ inline void wave_sort_desc_k128(uint64_t &hi, uint64_t &lo) {
#define S_(K, J) bitonic_step_k128<K, J>(hi, lo)
  S_(2, 1);
  S_(4, 2); S_(4, 1);
  S_(8, 4); S_(8, 2); S_(8, 1);
  S_(16, 8); S_(16, 4); S_(16, 2); S_(16, 1);
  S_(32, 16); S_(32, 8); S_(32, 4); S_(32, 2); S_(32, 1);
  S_(64, 32); S_(64, 16); S_(64, 8); S_(64, 4); S_(64, 2); S_(64, 1);
#undef S_
}

}  // namespace sann
