// Verification (MI355X): the short division / square-root sequences of sph_pcisph.hip (k_pressure_force) against the compiler's
// IEEE expansions (`/`, sqrtf under -fhip-fp32-correctly-rounded-divide-sqrt, denormals on), bit for bit, on random operands
// drawn from the ranges the kernel guards them to (plus the edges of those ranges and special mantissas).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero
//        -I../../smoothed-particle-hydrodynamics_amd/csrc -o exact_div_sqrt exact_div_sqrt.hip ; run: ./exact_div_sqrt [rounds=64]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
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

__global__ void k_check(uint64_t seed, int iters, unsigned long long* bad, unsigned long long* slow, float* example) {
  uint64_t s = seed * 0x100000001b3ull + (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9e3779b97f4a7c15ull;
  unsigned long long nb = 0, ns = 0;
  for (int it = 0; it < iters; it++) {
    // the kernel's data flow: a length scale, the bounds the host derives from it, d2 -> r = sqrt(d2) * scale -> numerators value * v_k
    const float scale = rnd_float(s, -26, 0, false);
    float d2Min, d2Max, valueMin;
    sph_fast_bounds(scale, &d2Min, &d2Max, &valueMin);
    const float x = rnd_float(s, SPH_FAST_S_EXP_LO - 4, SPH_FAST_S_EXP_HI + 3, false);  // also beyond both ends of the bounds
    float sq;
    const bool fs = sph_sqrt_fast(x, d2Min, d2Max, &sq);
    const float wantS = sqrtf(x);
    if (!fs) ns++;
    else if (__float_as_uint(sq) != __float_as_uint(wantS)) { if (nb == 0) { example[0] = x; example[1] = 0.f; example[2] = sq; example[3] = wantS; } nb++; }
    if (!fs) continue;  // (the kernel recomputes such a batch with the compiler's code)
    const float r = wantS * scale;
    const float value = (rng(s) & 63u) == 0 ? 0.f : rnd_float(s, -120, 62, true);
    float v[3];
    for (int c = 0; c < 3; c++) {
      const uint32_t pick = rng(s) & 15u;
      const int down = (int)(rng(s) % 70u);  // the component is up to 2^69 times shorter than r (numerators down to denormals)
      v[c] = pick == 0 ? 0.f : rnd_float(s, 0, 0, true) * r * __uint_as_float((uint32_t)(127 - down) << 23) * (pick == 1 ? 1.f : 0.999f);
    }
    float q[3];
    bool fast = sph_div3_by<false>(value * v[0], value * v[1], value * v[2], value, valueMin, r, q);
    for (int c = 0; c < 3; c++) {  // the caller's precondition: numerators zero or at least 2^SPH_FAST_A_EXP_LO
      const float a = fabsf(value * v[c]);
      if (a != 0.f && a < __uint_as_float((uint32_t)(SPH_FAST_A_EXP_LO + 127) << 23)) fast = false;
    }
    if (!fast) { ns++; continue; }
    for (int c = 0; c < 3; c++) {
      const float a = value * v[c];
      const float want = a / r;
      // (+0 and -0 are interchangeable for the kernel: the terms are added to sums that are never -0)
      const bool same = __float_as_uint(q[c]) == __float_as_uint(want) || (q[c] == 0.f && want == 0.f);
      if (!same) { if (nb == 0) { example[0] = a; example[1] = r; example[2] = q[c]; example[3] = want; } nb++; }
    }
  }
  if (nb) atomicAdd(bad, nb);
  if (ns) atomicAdd(slow, ns);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 64;
  unsigned long long *dBad, *dSlow; float* dEx;
  CHECK(hipMalloc(&dBad, 8)); CHECK(hipMalloc(&dSlow, 8)); CHECK(hipMalloc(&dEx, 16));
  CHECK(hipMemset(dBad, 0, 8)); CHECK(hipMemset(dSlow, 0, 8)); CHECK(hipMemset(dEx, 0, 16));
  const int blocks = 4096, threads = 256, iters = 1024;
  for (int r = 0; r < rounds; r++) hipLaunchKernelGGL(k_check, dim3(blocks), dim3(threads), 0, 0, (uint64_t)(r + 1), iters, dBad, dSlow, dEx);
  CHECK(hipDeviceSynchronize());
  unsigned long long bad, slow; float ex[4];
  CHECK(hipMemcpy(&bad, dBad, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&slow, dSlow, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ex, dEx, 16, hipMemcpyDeviceToHost));
  const double n = (double)rounds * blocks * threads * iters;
  printf("%.3g operand sets (a square root, then 3 quotients by r = root * scale): %llu results differ from the IEEE expansions; %llu times a guard sent the set to the slow path\n", n, bad, slow);
  if (bad) printf("first difference: a or x = %a, r = %a, got %a, want %a\n", ex[0], ex[1], ex[2], ex[3]);
  return bad ? 1 : 0;
}
