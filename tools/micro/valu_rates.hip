// Issue rates of the vector instructions the neighbour search is made of (gfx950): cycles per wave-instruction at 1, 2 and 4 waves
// per SIMD, dependent chains broken by independent accumulators. Build: make -C tools/micro valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, float seed) {
  f32x2 a[8], b = {seed, seed * 1.5f}, c = {0.5f * seed, 0.25f};
  float s[8];
  uint32_t m[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { a[i] = f32x2{seed + i, seed - i}; s[i] = seed * i; m[i] = (uint32_t)i; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (KIND == 0) a[i] = __builtin_elementwise_fma(a[i], b, c);                 // v_pk_fma_f32
      if (KIND == 1) s[i] = __builtin_fmaf(s[i], b.x, c.x);                        // v_fma_f32
      if (KIND == 2) a[i] = a[i] - b;                                              // v_pk_add_f32 (neg)
      if (KIND == 3) m[i] = __builtin_amdgcn_alignbit(m[i], __float_as_uint(b.x), 31);  // v_alignbit_b32
      if (KIND == 4) { a[i] = __builtin_elementwise_fma(a[i], b, c); s[i] = __builtin_fmaf(s[i], b.x, c.x); }  // mix
      if (KIND == 5) s[i] = s[i] * b.x;                                            // v_mul_f32
      if (KIND == 6) a[i] = a[i] * b;                                              // v_pk_mul_f32
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) r += a[i].x + a[i].y + s[i] + (float)m[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
static void run(const char* name, int perIter) {
  float* out; CHECK(hipMalloc(&out, 4 * 256 * 1024 * 4));
  const int iters = 4096;
  for (int wavesPerSimd = 1; wavesPerSimd <= 4; wavesPerSimd *= 2) {
    const int blocks = 256 * wavesPerSimd;  // 256 CUs x (wavesPerSimd blocks of 4 waves)
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, 16, 1.0001f);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instrPerSimd = (double)iters * 8 * perIter * wavesPerSimd;   // wave-instructions each SIMD issues
    printf("%-28s %d wave(s)/SIMD: %.3f ms -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, wavesPerSimd, ms,
           ms * 1e6 / instrPerSimd, ms * 1e6 / instrPerSimd * 2.4);
  }
  CHECK(hipFree(out));
}

int main() {
  run<0>("v_pk_fma_f32", 1);
  run<1>("v_fma_f32", 1);
  run<2>("v_pk_add_f32", 1);
  run<3>("v_alignbit_b32", 1);
  run<4>("v_pk_fma_f32 + v_fma_f32", 2);
  run<5>("v_mul_f32", 1);
  run<6>("v_pk_mul_f32", 1);
  return 0;
}
