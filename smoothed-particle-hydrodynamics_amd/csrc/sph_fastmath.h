// Shorter instruction sequences for three IEEE divisions by one denominator and for an IEEE square root, valid — bit for bit equal
// to the compiler's correctly rounded expansions (v_div_scale / v_div_fmas / v_div_fixup; the scaled v_sqrt sequence) — while the
// operands stay in ranges where those expansions neither scale nor fix anything up. Outside the ranges the functions return false
// (their results are then meaningless) and the caller uses `/` and sqrtf. Verified on MI355X against the compiler's code on 7e10 random operand sets drawn from (and
// beyond) these ranges: tools/micro/exact_div_sqrt.hip, profiles/r02/exact_div_sqrt.txt.
//
// Division (gfx9 f32, denormals on): the compiler emits d' = div_scale(d), n' = div_scale(n), y0 = rcp(d'), e = fma(-d', y0, 1),
// y = fma(e, y0, y0), q0 = n' y, r0 = fma(-d', q0, n'), q1 = fma(r0, y, q0), r1 = fma(-d', q1, n'), q = div_fmas(r1, y, q1),
// div_fixup. With exponents far from the ends of the range div_scale returns its operands, div_fmas is an fma and div_fixup
// returns q: the same arithmetic as below, where y is shared by the three numerators.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SPH_FAST_R_EXP_LO (-40)   // denominator 2^-40 <= r < 2^21
#define SPH_FAST_R_EXP_HI 20
#define SPH_FAST_S_EXP_LO (-60)   // square root argument 2^-60 <= x < 2^61
#define SPH_FAST_S_EXP_HI 60
#include <math.h>
// k_pressure_force takes r = sqrtf(d2) * simulationScale: one pair of bounds on d2 (two float compares per neighbour) keeps d2 inside
// the square root's range AND r inside the division's (2^-39 <= r <= 2^19). valueMin: see sph_div3_by. A scale for which the bounds
// are empty disables the short path (d2Min = +inf).
__host__ __device__ static inline void sph_fast_bounds(float simulationScale, float* d2Min, float* d2Max, float* valueMin) {
  const double s = (double)simulationScale;
  double lo = ldexp(1.0, -39) / s, hi = ldexp(1.0, 19) / s;
  lo = lo * lo * 1.001; hi = hi * hi * 0.999;
  if (lo < ldexp(1.0, SPH_FAST_S_EXP_LO + 1)) lo = ldexp(1.0, SPH_FAST_S_EXP_LO + 1);
  if (hi > ldexp(1.0, SPH_FAST_S_EXP_HI - 1)) hi = ldexp(1.0, SPH_FAST_S_EXP_HI - 1);
  const bool sane = s > 0.0 && lo < hi;
  *d2Min = sane ? (float)lo : INFINITY;
  *d2Max = sane ? (float)hi : 0.f;
  *valueMin = sane ? (float)(2.0 * ldexp(1.0, -100 + 4 + 24) / s) : INFINITY;  // 2 * 2^(SPH_FAST_A_EXP_LO - SPH_FAST_COORD_EXP_LO + 24) / scale
}

__device__ __forceinline__ bool sph_exp_in(float x, int lo, int hi) {  // 2^lo <= |x| < 2^(hi+1), by the exponent field
  const uint32_t e = (__float_as_uint(x) >> 23) & 0xffu;
  return e - (uint32_t)(lo + 127) <= (uint32_t)(hi - lo);
}

// q[k] = a_k / r for a_k = value * v_k with |v_k| <= ~r (components of a vector of length r): true if valid.
// Ranges: r by its exponent (CHECK_R) or guaranteed by the caller (sph_fast_bounds); the common factor zero or valueMin <= |value| <= 2^60 (the quotients then cannot overflow:
// |a_k / r| <= ~|value|). PRECONDITION the caller guarantees with valueMin: every numerator is zero or at least
// 2^SPH_FAST_A_EXP_LO in magnitude, so that the residuals a - r q (about 2^-24 |a|) stay normal numbers. k_pressure_force:
// v_k = (x_i - x_j).k * simulationScale; with the particle's own coordinates at least 2^SPH_FAST_COORD_EXP_LO in magnitude a
// non-zero difference is at least 2^(SPH_FAST_COORD_EXP_LO - 24) (the two floats are then of that magnitude or the difference is
// large), so valueMin = 2^(SPH_FAST_A_EXP_LO - SPH_FAST_COORD_EXP_LO + 24) / simulationScale, doubled for the roundings (sph_api.hip).
#define SPH_FAST_A_EXP_LO (-100)
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
