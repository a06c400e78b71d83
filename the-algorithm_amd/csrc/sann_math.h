// sann_math.h -- fp64 helpers whose results must be bit-identical on the host and on gfx950.
//
// Everything here is built with -ffp-contract=off: the reference's arithmetic is plain JVM
// fp64 (one rounding per operation, no fused multiply-add), and the top-k boundary of the
// partial-normalised cosine is decided by last-ulp differences (DESIGN.md "Near-tie regime").
// Division and sqrt are IEEE correctly rounded on both sides (checked by tests/test_fp64_gpu.py).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SANN_HD __host__ __device__
#else
#define SANN_HD
#endif

namespace sann {

SANN_HD inline uint64_t f64_bits(double x) {
  uint64_t u;
  memcpy(&u, &x, 8);
  return u;
}
SANN_HD inline double bits_f64(uint64_t u) {
  double x;
  memcpy(&x, &u, 8);
  return x;
}

// math.log as java.lang.StrictMath.log specifies it: the fdlibm __ieee754_log algorithm
// (argument reduction to [sqrt(2)/2, sqrt(2)), degree-14 Remez polynomial in s = f/(2+f),
// result assembled as k*ln2_hi - ((hfsq - (s*(hfsq+R) + k*ln2_lo)) - f)).
// Used for ScoringAlgorithm.LogCosineSimilarity (ApproximateCosineSimilarity.scala:112-113)
// and logNorm (CosineSimilarityUtil.scala:43-45).
SANN_HD inline double strict_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  const double two54 = 1.80143985094819840000e+16;
  const double Lg1 = 6.666666666666735130e-01;
  const double Lg2 = 3.999999999940941908e-01;
  const double Lg3 = 2.857142874366239149e-01;
  const double Lg4 = 2.222219843214978396e-01;
  const double Lg5 = 1.818357216161805012e-01;
  const double Lg6 = 1.531383769920937332e-01;
  const double Lg7 = 1.479819860511658591e-01;

  uint64_t u = f64_bits(x);
  int32_t hx = (int32_t)(u >> 32);
  uint32_t lx = (uint32_t)u;
  int32_t k = 0;
  if (hx < 0x00100000) {
    if (((hx & 0x7fffffff) | lx) == 0) return -two54 / 0.0;
    if (hx < 0) return (x - x) / 0.0;
    k -= 54;
    x *= two54;
    u = f64_bits(x);
    hx = (int32_t)(u >> 32);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  int32_t i = (hx + 0x95f64) & 0x100000;
  u = (u & 0xffffffffull) | ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32);
  x = bits_f64(u);
  k += (i >> 20);
  double f = x - 1.0;
  if ((0x000fffff & (2 + hx)) < 3) {
    if (f == 0.0) {
      if (k == 0) return 0.0;
      double dk = (double)k;
      return dk * ln2_hi + dk * ln2_lo;
    }
    double R = f * f * (0.5 - 0.33333333333333333 * f);
    if (k == 0) return f - R;
    double dk = (double)k;
    return dk * ln2_hi - ((R - dk * ln2_lo) - f);
  }
  double s = f / (2.0 + f);
  double dk = (double)k;
  double z = s * s;
  i = hx - 0x6147a;
  double w = z * z;
  int32_t j = 0x6b851 - hx;
  double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  double R = t2 + t1;
  if (i > 0) {
    double hfsq = 0.5 * f * f;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  }
  if (k == 0) return f - s * (f - R);
  return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

// Monotone map double -> uint64 that realises java.lang.Double.compare order
// (-0.0 < +0.0; NaNs never reach it: they fail `score >= minScore`).
SANN_HD inline uint64_t score_key(double s) {
  uint64_t b = f64_bits(s);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
SANN_HD inline double key_score(uint64_t k) {
  return bits_f64((k >> 63) ? (k ^ 0x8000000000000000ull) : ~k);
}
// Smaller tweet id = larger key (tie-break: tweet id ascending).
SANN_HD inline uint64_t id_key(int64_t id) { return ~((uint64_t)id ^ 0x8000000000000000ull); }
SANN_HD inline int64_t key_id(uint64_t k) { return (int64_t)((~k) ^ 0x8000000000000000ull); }

}  // namespace sann
