// Verification (MI355X): the short division / square-root sequences of sph_pcisph.hip (k_pressure_force) against the compiler's
// IEEE expansions (`/`, sqrtf under -fhip-fp32-correctly-rounded-divide-sqrt, denormals on), bit for bit, on random operands
// drawn from the ranges the kernel guards them to (plus the edges of those ranges and special mantissas).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero
//        -I../../smoothed-particle-hydrodynamics_amd/csrc -o exact_div_sqrt exact_div_sqrt.hip ; run: ./exact_div_sqrt [rounds=64]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include "sph_fastmath.h"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t rng(uint64_t& s) {  // splitmix64
  s += 0x9e3779b97f4a7c15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return (uint32_t)((z ^ (z >> 31)) >> 16);
}
// a float with a random mantissa (sometimes all zeros / all ones / one bit) and an exponent in [elo, ehi]
__device__ __forceinline__ float rnd_float(uint64_t& s, int elo, int ehi, bool signedv) {
  const uint32_t r = rng(s), m = rng(s);
  uint32_t mant = m & 0x7fffffu;
  const uint32_t kind = r & 15u;
  if (kind == 0) mant = 0;
  else if (kind == 1) mant = 0x7fffffu;
  else if (kind == 2) mant = 1u << ((m >> 24) % 23);
  else if (kind == 3) mant = 0x7fffffu ^ (1u << ((m >> 24) % 23));
  const int e = elo + (int)((r >> 4) % (uint32_t)(ehi - elo + 1));
  const uint32_t bits = ((uint32_t)(e + 127) << 23) | mant | ((signedv && (r >> 31)) ? 0x80000000u : 0u);
  return __uint_as_float(bits);
}

// The data flow of k_pressure_force's fast path (sph_pcisph.hip, pf_batch<true>) against its IEEE path (pf_batch<false>), one
// neighbour per iteration: d2 -> r = sqrt(d2) * scale; value = -(hs - r)^2 * 0.5 * (p_i + p_j) / rho_j (fast: 2^39 instead of 0.5, so
// everything downstream carries 2^40); numerators value * v_k; quotients by r; sum + quotient (fast: fma(q, 2^-40, sum)).
__global__ void k_check(uint64_t seed, int iters, unsigned long long* bad, unsigned long long* slow, float* example) {
  uint64_t s = seed * 0x100000001b3ull + (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9e3779b97f4a7c15ull;
  unsigned long long nb = 0, ns = 0;
  for (int it = 0; it < iters; it++) {
    const float scale = rnd_float(s, -26, 0, false);
    const float x = rnd_float(s, SPH_FAST_S_EXP_LO - 4, SPH_FAST_S_EXP_HI + 3, false);  // also beyond both ends of the bounds
    const float wantS = sqrtf(x);
    const float r = wantS * scale;
    // the support radius: usually a few r, sometimes within a few ulps .. percent of r (value then gets tiny: the guard's edge)
    const uint32_t hk = rng(s);
    const float hs = (hk & 1u) ? r * (1.f + __uint_as_float((uint32_t)(127 - (int)((hk >> 1) % 24u)) << 23)) : r * (1.f + 4.f * rnd_float(s, -3, 0, false));
    float d2Min, d2Max, valueMin;
    sph_fast_bounds(scale, hs, 1.0f, &d2Min, &d2Max, &valueMin);
    float sq;
    const bool fs = sph_sqrt_fast(x, d2Min, d2Max, &sq);
    if (!fs) ns++;
    else if (__float_as_uint(sq) != __float_as_uint(wantS)) { if (nb == 0) { example[0] = x; example[1] = 0.f; example[2] = sq; example[3] = wantS; } nb++; }
    if (!fs) continue;  // (the kernel recomputes such a batch with the compiler's code)
    const float P = (rng(s) & 63u) == 0 ? 0.f : rnd_float(s, -40, 30, false);   // p_i + p_j >= 0
    const float rho = rnd_float(s, -10, 20, false);                              // >= rhoMin = 1 would be the solver's guarantee; smaller is harsher
    const float num = -(hs - r) * (hs - r) * 0.5f * P, numS = -(hs - r) * (hs - r) * SPH_FAST_HALF_SCALED * P;
    const float value = num / rho, valueS = sph_div1_by(numS, rho);  // (the kernel's fast path: short sequence, no guard of its own)
    float v[3];
    for (int c = 0; c < 3; c++) {
      const uint32_t pick = rng(s) & 15u;
      const int down = (int)(rng(s) % 70u);  // the component is up to 2^69 times shorter than r (numerators down to denormals)
      v[c] = pick == 0 ? 0.f : rnd_float(s, 0, 0, true) * r * __uint_as_float((uint32_t)(127 - down) << 23) * (pick == 1 ? 1.f : 0.999f);
    }
    float q[3];
    bool fast = sph_div3_by<false>(valueS * v[0], valueS * v[1], valueS * v[2], valueS, valueMin, r, q);
    for (int c = 0; c < 3; c++) {  // the caller's precondition: scaled numerators zero or at least 2^SPH_FAST_A_EXP_LO
      const float a = fabsf(valueS * v[c]);
      if (a != 0.f && a < __uint_as_float((uint32_t)(SPH_FAST_A_EXP_LO + 127) << 23)) fast = false;
    }
    if (!fast) { ns++; continue; }
    for (int c = 0; c < 3; c++) {
      const float a = value * v[c];
      const float sum = (rng(s) & 3u) == 0 ? 0.f : rnd_float(s, -60, 40, true);
      const float want = sum + a / r;
      const float got = __builtin_fmaf(q[c], SPH_FAST_UNSCALE, sum);
      // (+0 and -0 are interchangeable for the kernel: the terms are added to sums that are never -0)
      const bool same = __float_as_uint(got) == __float_as_uint(want) || (got == 0.f && want == 0.f);
      if (!same) { if (nb == 0) { example[0] = a; example[1] = r; example[2] = got; example[3] = want; } nb++; }
    }
  }
  if (nb) atomicAdd(bad, nb);
  if (ns) atomicAdd(slow, ns);
}

// ---- exhaustive square root: EVERY float in [d2Min, d2Max] of a given simulationScale (the range k_pressure_force's guard lets
// through), plus the floats just outside both ends, which the guard must reject. 2^32 / 256 threads x 4096 blocks: each thread
// takes a strided share of the bit patterns.
__global__ void k_sqrt_all(uint32_t firstBits, uint32_t lastBits, float d2Min, float d2Max, unsigned long long* bad,
                           unsigned long long* checked, float* example) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long nb = 0, nc = 0;
  for (uint64_t b = (uint64_t)firstBits + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= (uint64_t)lastBits; b += stride) {
    const float x = __uint_as_float((uint32_t)b);
    float sq;
    const bool ok = sph_sqrt_fast(x, d2Min, d2Max, &sq);
    const bool inside = x >= d2Min && x <= d2Max;
    if (ok != inside) { if (nb == 0) { example[0] = x; example[1] = -1.f; example[2] = sq; example[3] = 0.f; } nb++; continue; }
    if (!ok) continue;
    nc++;
    const float want = sqrtf(x);
    if (__float_as_uint(sq) != __float_as_uint(want)) { if (nb == 0) { example[0] = x; example[1] = 0.f; example[2] = sq; example[3] = want; } nb++; }
  }
  if (nb) atomicAdd(bad, nb);
  atomicAdd(checked, nc);
}

// ---- directed division: every mantissa of r at three exponents; numerators built so that a / r lies next to a ROUNDING
// MIDPOINT of the quotient (the cases a short division sequence gets wrong first) and next to / exactly at a float.
// For a target quotient q (float) the exact products q * r and (q + ulp(q)/2) * r have <= 49 significant bits, so they are exact
// in double; the floats just below and above them are the hard numerators.
__device__ __forceinline__ float f_down(double v) {  // largest float <= v (v > 0, normal range)
  float f = (float)v;
  if ((double)f > v) f = __uint_as_float(__float_as_uint(f) - 1u);
  return f;
}
__global__ void k_div_directed(int rExp, float valueMin, unsigned long long* bad, unsigned long long* checked, float* example) {
  const uint32_t mant = blockIdx.x * blockDim.x + threadIdx.x;  // 2^23 threads: one per mantissa of r
  if (mant >= (1u << 23)) return;
  const float r = __uint_as_float(((uint32_t)(rExp + 127) << 23) | mant);
  uint64_t s = 0x51ed270b7a3ull * (mant + 1) + (uint64_t)(rExp + 200);
  unsigned long long nb = 0, nc = 0;
  const int qExps[3] = {-12, 0, 17};
  for (int t = 0; t < 12; t++) {
    const uint32_t qm = (t < 3) ? (t == 0 ? 0u : (t == 1 ? 0x7fffffu : 0x400000u)) : (rng(s) & 0x7fffffu);
    const float q = __uint_as_float(((uint32_t)(qExps[t % 3] + 127) << 23) | qm);
    const double half = 0.5 * (double)(__uint_as_float(__float_as_uint(q) + 1u) - q);  // ulp(q) / 2, exact
    const double mid = ((double)q + half) * (double)r, at = (double)q * (double)r;     // exact (<= 49 bits)
    float a[3];
    for (int pass = 0; pass < 2; pass++) {
      const float lo = f_down(pass == 0 ? mid : at);
      a[0] = lo; a[1] = __uint_as_float(__float_as_uint(lo) + 1u); a[2] = -__uint_as_float(__float_as_uint(lo) - 1u);
      float got[3];
      const bool ok = sph_div3_by<true>(a[0], a[1], a[2], 1.0f, valueMin, r, got);
      if (!ok) continue;
      for (int c = 0; c < 3; c++) {
        const float want = a[c] / r;
        nc++;
        if (__float_as_uint(got[c]) != __float_as_uint(want)) { if (nb == 0) { example[0] = a[c]; example[1] = r; example[2] = got[c]; example[3] = want; } nb++; }
      }
    }
  }
  if (nb) atomicAdd(bad, nb);
  atomicAdd(checked, nc);
}

static int report(const char* what, unsigned long long* dBad, unsigned long long* dN, float* dEx, unsigned long long atLeast) {
  CHECK(hipDeviceSynchronize());
  unsigned long long bad, n; float ex[4];
  CHECK(hipMemcpy(&bad, dBad, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&n, dN, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ex, dEx, 16, hipMemcpyDeviceToHost));
  printf("%s: %llu checked, %llu differ from the IEEE expansions\n", what, n, bad);
  if (bad) printf("  first difference: a or x = %a, r = %a, got %a, want %a\n", ex[0], ex[1], ex[2], ex[3]);
  if (n < atLeast) { printf("  only %llu cases ran (expected at least %llu): the check is void\n", n, atLeast); return 1; }
  CHECK(hipMemset(dBad, 0, 8)); CHECK(hipMemset(dN, 0, 8));
  return bad ? 1 : 0;
}

// usage: exact_div_sqrt [rounds=64]                       random operand sets (4096 x 256 x 1024 per round)
//        exact_div_sqrt suite <simulationScale> [rounds=1]  exhaustive square root over the guarded range of that scale, directed
//                                                         division cases for every mantissa of r, then `rounds` random rounds
int main(int argc, char** argv) {
  unsigned long long *dBad, *dSlow; float* dEx;
  CHECK(hipMalloc(&dBad, 8)); CHECK(hipMalloc(&dSlow, 8)); CHECK(hipMalloc(&dEx, 16));
  CHECK(hipMemset(dBad, 0, 8)); CHECK(hipMemset(dSlow, 0, 8)); CHECK(hipMemset(dEx, 0, 16));
  int rounds = 64, failed = 0;
  if (argc > 2 && !strcmp(argv[1], "suite")) {
    const float scale = (float)atof(argv[2]);
    rounds = argc > 3 ? atoi(argv[3]) : 1;
    float d2Min, d2Max, valueMin;
    sph_fast_bounds(scale, 3.34f * scale, 100.f, &d2Min, &d2Max, &valueMin);
    uint32_t lo, hi; memcpy(&lo, &d2Min, 4); memcpy(&hi, &d2Max, 4);
    printf("simulationScale %a: guarded d2 range [%a, %a] = %llu floats, valueMin %a\n", scale, d2Min, d2Max, (unsigned long long)hi - lo + 1, valueMin);
    if (!(d2Min < d2Max)) { printf("empty range\n"); return 1; }
    hipLaunchKernelGGL(k_sqrt_all, dim3(4096), dim3(256), 0, 0, lo - 64u, hi + 64u, d2Min, d2Max, dBad, dSlow, dEx);  // 64 floats beyond each edge
    failed |= report("square root, every float of the guarded range (+ guard edges)", dBad, dSlow, dEx, (unsigned long long)hi - lo + 1);
    // r = distance * simulationScale: its exponent in the shipped configuration (~2^-20 .. 2^-17), and two far ones
    float rTypical = 3.34f * scale; int e0; frexpf(rTypical, &e0);
    const int rExps[3] = {e0 - 1, -38, 17};
    for (int k = 0; k < 3; k++) {
      hipLaunchKernelGGL(k_div_directed, dim3((1u << 23) / 256), dim3(256), 0, 0, rExps[k], valueMin, dBad, dSlow, dEx);
      char what[160];
      snprintf(what, sizeof(what), "division, every mantissa of r at 2^%d, quotients at / next to rounding midpoints and floats", rExps[k]);
      failed |= report(what, dBad, dSlow, dEx, 12ull * 6ull * (1ull << 23));
    }
  } else if (argc > 1) {
    rounds = atoi(argv[1]);
  }
  const int blocks = 4096, threads = 256, iters = 1024;
  for (int r = 0; r < rounds; r++) hipLaunchKernelGGL(k_check, dim3(blocks), dim3(threads), 0, 0, (uint64_t)(r + 1), iters, dBad, dSlow, dEx);
  CHECK(hipDeviceSynchronize());
  unsigned long long bad, slow; float ex[4];
  CHECK(hipMemcpy(&bad, dBad, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&slow, dSlow, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ex, dEx, 16, hipMemcpyDeviceToHost));
  const double n = (double)rounds * blocks * threads * iters;
  printf("%.3g operand sets (a square root, then 3 quotients by r = root * scale): %llu results differ from the IEEE expansions; %llu times a guard sent the set to the slow path\n", n, bad, slow);
  if (bad) printf("first difference: a or x = %a, r = %a, got %a, want %a\n", ex[0], ex[1], ex[2], ex[3]);
  return (bad || failed) ? 1 : 0;
}
