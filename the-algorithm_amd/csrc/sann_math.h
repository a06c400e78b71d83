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

// math.exp as java.lang.StrictMath.exp specifies it: the fdlibm __ieee754_exp algorithm (argument reduction
// x = k ln2 + r, |r| <= 0.5 ln2; exp(r) from the degree-5 Remez approximation of r (exp(r)+1)/(exp(r)-1); scaling by 2^k).
// Used to decay posting scores to "now" (algebird DecayedValueMonoid.scaledPlus, called from
// summingbird/common/ThriftDecayedValueMonoid.scala:33-38).  HotSpot's Math.exp intrinsic may differ by 1 ulp.
SANN_HD inline double strict_exp(double x) {
  const double o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02;
  const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  const double huge = 1.0e+300, twom1000 = 9.33263618503218878990e-302;
  uint64_t u = f64_bits(x);
  uint32_t hx = (uint32_t)(u >> 32);
  const int xsb = (int)((hx >> 31) & 1u);
  hx &= 0x7fffffffu;
  double hi = 0.0, lo = 0.0;
  int k = 0;
  if (hx >= 0x40862E42u) {  // |x| >= 709.78...
    if (hx >= 0x7ff00000u) {
      if (((hx & 0xfffffu) | (uint32_t)u) != 0) return x + x;  // NaN
      return xsb == 0 ? x : 0.0;                                // exp(+-inf) = {inf, 0}
    }
    if (x > o_threshold) return huge * huge;
    if (x < u_threshold) return twom1000 * twom1000;
  }
  if (hx > 0x3fd62e42u) {  // |x| > 0.5 ln2
    if (hx < 0x3FF0A2B2u) {  // and |x| < 1.5 ln2
      hi = xsb ? x + ln2HI : x - ln2HI;
      lo = xsb ? -ln2LO : ln2LO;
      k = 1 - xsb - xsb;
    } else {
      k = (int)(invln2 * x + (xsb ? -0.5 : 0.5));
      const double t = (double)k;
      hi = x - t * ln2HI;
      lo = t * ln2LO;
    }
    x = hi - lo;
  } else if (hx < 0x3e300000u) {  // |x| < 2^-28
    return 1.0 + x;
  }
  const double t = x * x;
  const double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
  double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
  if (k >= -1021) return bits_f64(f64_bits(y) + ((uint64_t)(uint32_t)k << 52));
  y = bits_f64(f64_bits(y) + ((uint64_t)(uint32_t)(k + 1000) << 52));
  return y * twom1000;
}

// algebird DecayedValueMonoid(eps = 0.0).plus(v, DecayedValue(0.0, now)) -- what
// ThriftDecayedValueMonoid.decayToTimestamp (summingbird/common/ThriftDecayedValueMonoid.scala:33-38, built with
// Implicits.scala:28 eps 0.0) does to a posting's (value, scaledTime), scaledTime = ms * ln 2 / halfLife:
//   scaledPlus(newer, older) = newer.value + exp(older.scaledTime - newer.scaledTime) * older.value, zero unless |.| > eps
// (com.twitter.algebird is not vendored and no version is pinned in the tree; this is its published DecayedValue.)
SANN_HD inline double decay_to_timestamp(double value, double scaled_time, double now_scaled) {
  const double nv = scaled_time < now_scaled ? 0.0 + strict_exp(scaled_time - now_scaled) * value
                                             : value + strict_exp(now_scaled - scaled_time) * 0.0;
  return (nv > 0.0 || nv < 0.0) ? nv : 0.0;
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
