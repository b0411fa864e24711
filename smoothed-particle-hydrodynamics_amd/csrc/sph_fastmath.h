// Shorter instruction sequences for three IEEE divisions by one denominator and for an IEEE square root, valid — bit for bit equal
// to the compiler's correctly rounded expansions (v_div_scale / v_div_fmas / v_div_fixup; the scaled v_sqrt sequence) — while the
// operands stay in ranges where those expansions neither scale nor fix anything up. Outside the ranges the functions return false
// (their results are then meaningless) and the caller uses `/` and sqrtf. Verified on MI355X against the compiler's code:
// tools/micro/exact_div_sqrt.hip (exhaustive square root over the guarded range, directed division cases for every mantissa of the
// denominator, random operand sets of the kernel's data flow; run by tests/test_gpu_parity.py), profiles/r03/exact_div_sqrt.txt.
//
// Division (gfx9 f32, denormals on): the compiler emits d' = div_scale(d), n' = div_scale(n), y0 = rcp(d'), e = fma(-d', y0, 1),
// y = fma(e, y0, y0), q0 = n' y, r0 = fma(-d', q0, n'), q1 = fma(r0, y, q0), r1 = fma(-d', q1, n'), q = div_fmas(r1, y, q1),
// div_fixup. With exponents far from the ends of the range div_scale returns its operands, div_fmas is an fma and div_fixup
// returns q: the same arithmetic as below, where y is shared by the three numerators.
//
// SCALED numerators (round 3). The residuals a - r q of the sequence are about 2^-46 |a| and must be exact, so a numerator must be
// zero or at least ~2^-100. The numerators of k_pressure_force are value * v_k with value = -(hs - r)^2 * 0.5 * (p_i + p_j) / rho*_j,
// which gets arbitrarily small as r approaches the support radius hs — in a disordered liquid almost every batch of 8 x 64
// neighbours holds such a pair, and the batch then paid for both paths (a collapsing column: 1.5 ms instead of 0.25 ms per step in
// this kernel). The kernel therefore carries a factor 2^SPH_FAST_SCALE_EXP through the fast path: the numerator's constant 0.5
// becomes 2^39 (so `value` arrives multiplied by 2^40, exactly: powers of two commute with every rounding while nothing is
// subnormal or overflows), the quotients come out multiplied by 2^40, and the accumulation `sum + q` is done as
// fma(q', 2^-40, sum) — one rounding of the same exact value. What must hold is now only that the UNSCALED value, numerators and
// quotients are normal numbers (>= 2^-126): 26 binades more room, and a pair fails the guard only within ~2e-6 hs of the support
// radius.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SPH_FAST_R_EXP_LO (-40)   // denominator 2^-40 <= r < 2^21 (CHECK_R form); k_pressure_force: 2^-39 <= r <= 1 by its d2 bounds
#define SPH_FAST_R_EXP_HI 20
#define SPH_FAST_S_EXP_LO (-60)   // square root argument 2^-60 <= x < 2^61
#define SPH_FAST_S_EXP_HI 60
#define SPH_FAST_SCALE_EXP 40     // the factor carried by value, numerators and quotients on the fast path
#define SPH_FAST_HALF_SCALED 0x1p39f   // 0.5 * 2^40: replaces the 0.5 of the numerator (sphFluid.cl:1166-1168)
#define SPH_FAST_UNSCALE 0x1p-40f
#include <math.h>
// k_pressure_force takes r = sqrtf(d2) * simulationScale: one pair of bounds on d2 (two float compares per neighbour) keeps d2 inside
// the square root's range AND r inside [2^-39, 1] (r <= 1: the unscaled quotients a / r are then at least as large as the
// numerators, hence normal). valueMin bounds the SCALED value, see sph_div3_by. rhoMin = the smallest predicted density the solver
// can produce (hs^6 * mass * Wpoly6): `value` is a quotient by rho*, and the argument that its scaled form is exact needs rho* not
// absurdly small; hScaled = h * simulationScale. A configuration for which any of this fails disables the short path (d2Min = +inf).
__host__ __device__ static inline void sph_fast_bounds(float simulationScale, float hScaled, float rhoMin, float* d2Min, float* d2Max,
                                                       float* valueMin) {
  const double s = (double)simulationScale;
  double lo = ldexp(1.0, -39) / s, hi = 1.0 / s;
  lo = lo * lo * 1.001; hi = hi * hi * 0.999;
  if (lo < ldexp(1.0, SPH_FAST_S_EXP_LO + 1)) lo = ldexp(1.0, SPH_FAST_S_EXP_LO + 1);
  if (hi > ldexp(1.0, SPH_FAST_S_EXP_HI - 1)) hi = ldexp(1.0, SPH_FAST_S_EXP_HI - 1);
  // (hScaled >= 2^-30: (hs - r)^2, a square of a difference of floats of that size, and its half are then normal numbers)
  const bool sane = s > 0.0 && lo < hi && rhoMin >= 0x1p-40f && 33.f * rhoMin <= 0x1p20f && hScaled >= 0x1p-30f && hScaled <= 1.f;
  *d2Min = sane ? (float)lo : INFINITY;
  *d2Max = sane ? (float)hi : 0.f;
  // 2 * 2^(SPH_FAST_A_EXP_LO - SPH_FAST_COORD_EXP_LO + 24) / scale, on the scaled value
  *valueMin = sane ? (float)(2.0 * ldexp(1.0, -86 + 4 + 24) / s) : INFINITY;
}

__device__ __forceinline__ bool sph_exp_in(float x, int lo, int hi) {  // 2^lo <= |x| < 2^(hi+1), by the exponent field
  const uint32_t e = (__float_as_uint(x) >> 23) & 0xffu;
  return e - (uint32_t)(lo + 127) <= (uint32_t)(hi - lo);
}

// q[k] = a_k / r for a_k = value * v_k with |v_k| <= ~r (components of a vector of length r): true if valid. On the scaled path
// `value`, a_k and q[k] all carry the factor 2^SPH_FAST_SCALE_EXP (the sequence itself does not care).
// Ranges: r by its exponent (CHECK_R) or guaranteed by the caller (sph_fast_bounds); the common factor zero or valueMin <= |value| <= 2^60 (the quotients then cannot overflow:
// |a_k / r| <= ~|value|). PRECONDITION the caller guarantees with valueMin: every numerator is zero or at least
// 2^SPH_FAST_A_EXP_LO in magnitude — i.e. its unscaled form is a normal number, so that the scaling is exact, and a fortiori the
// residuals a - r q (about 2^-46 |a|) are exact. k_pressure_force:
// v_k = (x_i - x_j).k * simulationScale; with the particle's own coordinates at least 2^SPH_FAST_COORD_EXP_LO in magnitude a
// non-zero difference is at least 2^(SPH_FAST_COORD_EXP_LO - 24) (the two floats are then of that magnitude or the difference is
// large), so valueMin = 2^(SPH_FAST_A_EXP_LO - SPH_FAST_COORD_EXP_LO + 24) / simulationScale, doubled for the roundings.
#define SPH_FAST_A_EXP_LO (-126 + SPH_FAST_SCALE_EXP)
#define SPH_FAST_COORD_EXP_LO (-4)
#define SPH_FAST_V_MAX 0x1p60f
template <bool CHECK_R>
__device__ __forceinline__ bool sph_div3_by(float a0, float a1, float a2, float value, float valueMin, float r, float q[3]) {
  const float av = __builtin_fabsf(value);
  const bool ok = (!CHECK_R || sph_exp_in(r, SPH_FAST_R_EXP_LO, SPH_FAST_R_EXP_HI)) && (value == 0.f || (av >= valueMin && av <= SPH_FAST_V_MAX));
  const float y0 = __builtin_amdgcn_rcpf(r);
  const float e = __builtin_fmaf(-r, y0, 1.f);
  const float y = __builtin_fmaf(e, y0, y0);
  const float a[3] = {a0, a1, a2};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float t = a[k] * y;
    const float r0 = __builtin_fmaf(-r, t, a[k]);
    t = __builtin_fmaf(r0, y, t);
    const float r1 = __builtin_fmaf(-r, t, a[k]);
    q[k] = __builtin_fmaf(r1, y, t);
  }
  return ok;  // (the results are computed either way — no branch — and are meaningless when this is false)
}

// a / d by the same short sequence, one numerator. k_pressure_force: value = num / rho*_j. No guard of its own is needed there:
// rho* is max(sum, hs^6) * mass * Wpoly6 with sum <= 32 hs^6, i.e. inside [rhoMin, 33 rhoMin] for ANY input (sph_fast_bounds checks
// that interval against [2^-40, 2^20] once); a numerator too small for exact residuals (< 2^-100) gives a quotient far below the
// valueMin that sph_div3_by checks next, zero gives exactly zero, and inf / NaN fail that check too.
__device__ __forceinline__ float sph_div1_by(float a, float d) {
  const float y0 = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, y0, 1.f);
  const float y = __builtin_fmaf(e, y0, y0);
  float t = a * y;
  const float r0 = __builtin_fmaf(-d, t, a);
  t = __builtin_fmaf(r0, y, t);
  const float r1 = __builtin_fmaf(-d, t, a);
  return __builtin_fmaf(r1, y, t);
}

// *out = sqrtf(x), correctly rounded, for x in the guarded range: v_sqrt_f32 (1 ulp) and the two one-ulp neighbours checked by residual
__device__ __forceinline__ bool sph_sqrt_fast(float x, float xMin, float xMax, float* out) {  // 2^-60 <= xMin <= xMax < 2^61
  const float s = __builtin_amdgcn_sqrtf(x);
  const float lo = __uint_as_float(__float_as_uint(s) - 1u), hi = __uint_as_float(__float_as_uint(s) + 1u);
  const float rlo = __builtin_fmaf(-lo, s, x), rhi = __builtin_fmaf(-hi, s, x);
  float res = s;
  if (rlo <= 0.f) res = lo;
  if (rhi > 0.f) res = hi;
  *out = res;
  return x >= xMin && x <= xMax;  // (no branch: *out is meaningless when this is false)
}
