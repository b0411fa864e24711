// Micro-benchmark (MI355X): what does a 64-lane 12-byte gather cost as a function of the lanes that take part?
//   mode 0: global_load_dwordx3, every lane a random record in a 512-record neighbourhood
//   mode 1: raw buffer load, a fraction of the lanes out of range (dropped by the address unit)
//   mode 2: global load under an exec mask (if), same fraction of lanes switched off
//   mode 3: global load, the same fraction of lanes redirected to their own (coalesced) record
//   mode 4: all lanes, records packed at a 12-byte stride
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_lanes gather_lanes.hip ; run: ./gather_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ rec, const int* __restrict__ nbr, float* __restrict__ out, int n, int keepPermille) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= n) return;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)rec, 0, n * 16, 0x00020000);
  float s = 0.f;
#pragma unroll 1
  for (int b = 0; b < 4; b++) {
    int j[8];
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) j[k2] = nbr[(size_t)(b * 8 + k2) * n + id];
    float x[8], y[8], z[8];
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) {
      const int jj = j[k2] >> 10;
      const bool keep = (j[k2] & 1023) < keepPermille;   // the low 10 bits of the entry: a per-(particle, slot) random number
      if (MODE == 0) { const float4 v = rec[jj]; x[k2] = v.x; y[k2] = v.y; z[k2] = v.z; }
      if (MODE == 1) {
        const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rsrc, keep ? (unsigned)jj * 16u : 0xffffffffu, 0, 0);
        x[k2] = __uint_as_float(v.x); y[k2] = __uint_as_float(v.y); z[k2] = __uint_as_float(v.z);
      }
      if (MODE == 2) { x[k2] = y[k2] = z[k2] = 0.f; if (keep) { const float4 v = rec[jj]; x[k2] = v.x; y[k2] = v.y; z[k2] = v.z; } }
      if (MODE == 3) { const float4 v = rec[keep ? jj : id]; x[k2] = v.x; y[k2] = v.y; z[k2] = v.z; }
      if (MODE == 4) {  // records packed at a 12-byte stride (10.7 per 128-byte line instead of 8)
        const float* p3 = reinterpret_cast<const float*>(rec) + (size_t)jj * 3;
        x[k2] = p3[0]; y[k2] = p3[1]; z[k2] = p3[2];
      }
    }
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) s += x[k2] * y[k2] + z[k2];
  }
  out[id] = s;
}

int main() {
  const int n = 1 << 20;
  std::vector<float4> rec(n);
  std::vector<int> nbr((size_t)32 * n);
  srand(1);
  for (int i = 0; i < n; i++) rec[i] = make_float4(i, 1.f, 2.f, 3.f);
  for (int s = 0; s < 32; s++)
    for (int i = 0; i < n; i++) {
      int j = i - 256 + rand() % 512;
      j = j < 0 ? 0 : (j >= n ? n - 1 : j);
      nbr[(size_t)s * n + i] = (j << 10) | (rand() & 1023);
    }
  float4* dRec; int* dNbr; float* dOut;
  CHECK(hipMalloc(&dRec, n * sizeof(float4))); CHECK(hipMalloc(&dNbr, nbr.size() * 4)); CHECK(hipMalloc(&dOut, n * 4));
  CHECK(hipMemcpy(dRec, rec.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dNbr, nbr.data(), nbr.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int keeps[5] = {1024, 768, 512, 340, 0};
  for (int mode = 0; mode < 5; mode++)
    for (int ki = 0; ki < 5; ki++) {
      if ((mode == 0 || mode == 4) && ki) continue;
      const int keep = keeps[ki];
      for (int rep = 0; rep < 23; rep++) {
        if (rep == 3) CHECK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(n / 256), dim3(256), 0, 0, dRec, dNbr, dOut, n, keep);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(n / 256), dim3(256), 0, 0, dRec, dNbr, dOut, n, keep);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(n / 256), dim3(256), 0, 0, dRec, dNbr, dOut, n, keep);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(n / 256), dim3(256), 0, 0, dRec, dNbr, dOut, n, keep);
        if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(n / 256), dim3(256), 0, 0, dRec, dNbr, dOut, n, keep);
      }
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("mode %d keep %4d/1024 lanes: %.4f ms per launch (1 M particles x 32 gathers)\n", mode, keep, ms / 20);
    }
  return 0;
}
