// Slab decomposition support (include/sphmi.h "Spatial decomposition"; no reference counterpart — SURVEY.md 8e):
// select the authoritative particles after a step, build the halo messages for the two neighbouring slabs, and rebuild
// the local particle set (kept + received) sorted by global id. Everything stays in HBM; the caller moves the two
// messages with RCCL send/recv (smoothed-particle-hydrodynamics_amd/sphmi/slab.py).
#include "sph_common.h"

#define REC SPH_SLAB_RECORD_WORDS

__device__ __forceinline__ void put_record(uint32_t* msg, uint32_t j, const float4 p, const float4 v, uint32_t g) {
  uint32_t* r = msg + (size_t)j * REC;
  r[0] = __float_as_uint(p.x); r[1] = __float_as_uint(p.y); r[2] = __float_as_uint(p.z); r[3] = __float_as_uint(p.w);
  r[4] = __float_as_uint(v.x); r[5] = __float_as_uint(v.y); r[6] = __float_as_uint(v.z); r[7] = __float_as_uint(v.w);
  r[8] = g;
}

// Owned particles (flag set at the last rebuild) are the authoritative ones. All of them stay in the local set (they move
// less than one cell layer per step); those now within W layers of a cut are also copied into the neighbour's message.
// Order is irrelevant here (the rebuild sorts by global id), so slots are handed out by counters: LDS counters inside a
// workgroup of 2048 particles, then ONE global atomic per workgroup and destination (a single hot word serves only
// ~88 atomics/us on this part: one atomic per wave cost 0.19 ms per step at 1 M particles).
#define PACK_ITEMS 8
__global__ __launch_bounds__(SPH_BLOCK) void k_slab_pack(SphDev d, sph_slab slab, uint32_t* __restrict__ counts,
                                                         uint32_t* __restrict__ msgDown, uint32_t* __restrict__ msgUp,
                                                         int capRecords) {
  __shared__ uint32_t local[3], base[3];
  if (threadIdx.x < 3) local[threadIdx.x] = 0u;
  __syncthreads();
  const int first = blockIdx.x * (SPH_BLOCK * PACK_ITEMS) + threadIdx.x;
  uint32_t slotKeep[PACK_ITEMS], slotDown[PACK_ITEMS], slotUp[PACK_ITEMS];
#pragma unroll
  for (int u = 0; u < PACK_ITEMS; u++) {
    const int i = first + u * SPH_BLOCK;
    slotKeep[u] = slotDown[u] = slotUp[u] = 0xffffffffu;
    if (i < d.N && d.owned[i]) {
      const int layer = (int)(d.posOrig[i].z * d.cellSizeInv);  // the z cell coordinate hashParticles uses (sphFluid.cl:199)
      slotKeep[u] = atomicAdd(&local[0], 1u);
      if (slab.hasLower && layer < slab.layerLo + slab.ghostLayers) slotDown[u] = atomicAdd(&local[1], 1u);
      if (slab.hasUpper && layer >= slab.layerHi - slab.ghostLayers) slotUp[u] = atomicAdd(&local[2], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) base[threadIdx.x] = local[threadIdx.x] ? atomicAdd(&counts[threadIdx.x], local[threadIdx.x]) : 0u;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PACK_ITEMS; u++) {
    if (slotKeep[u] == 0xffffffffu) continue;
    const int i = first + u * SPH_BLOCK;
    const float4 p = d.posOrig[i], v = d.velOrig[i];
    const uint32_t g = d.gid[i];
    const uint32_t k = base[0] + slotKeep[u];
    d.sortedPos[k] = p; d.sortedVel[k] = v; d.keys[k] = g;  // sorted* / keys are free between two steps: staging area
    if (slotDown[u] != 0xffffffffu && (int)(base[1] + slotDown[u]) < capRecords) put_record(msgDown, base[1] + slotDown[u], p, v, g);
    if (slotUp[u] != 0xffffffffu && (int)(base[2] + slotUp[u]) < capRecords) put_record(msgUp, base[2] + slotUp[u], p, v, g);
  }
}

int sphk_slab_pack(sph_solver* s, uint32_t* msgDown, uint32_t* msgUp, int capRecords) {
  SPH_HIP(hipMemsetAsync(s->slabCounts, 0, sizeof(uint32_t) * 4, s->stream));
  hipLaunchKernelGGL(k_slab_pack, dim3(sph_blocks(s->d.N, SPH_BLOCK * PACK_ITEMS)), dim3(SPH_BLOCK), 0, s->stream, s->d, s->slab,
                     s->slabCounts, msgDown, msgUp, capRecords);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

__global__ __launch_bounds__(SPH_BLOCK) void k_slab_append(SphDev d, const uint32_t* __restrict__ msg, int n, int base) {
  const int t = blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (t >= n) return;
  const uint32_t* r = msg + (size_t)t * REC;
  d.sortedPos[base + t] = make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
  d.sortedVel[base + t] = make_float4(__uint_as_float(r[4]), __uint_as_float(r[5]), __uint_as_float(r[6]), __uint_as_float(r[7]));
  d.keys[base + t] = r[8];
}

__global__ __launch_bounds__(SPH_BLOCK) void k_iota(uint32_t* __restrict__ v, int n) {
  const int i = blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (i < n) v[i] = (uint32_t)i;
}

// after the sort by global id: vals[i] = staging index of the i-th particle
__global__ __launch_bounds__(SPH_BLOCK) void k_slab_gather(SphDev d, sph_slab slab, int n) {
  const int i = blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint32_t src = d.vals[i];
  const float4 p = d.sortedPos[src];
  d.posOrig[i] = p;
  d.velOrig[i] = d.sortedVel[src];
  d.gid[i] = d.keys[i];
  const int layer = (int)(p.z * d.cellSizeInv);
  d.owned[i] = (layer >= slab.layerLo && layer < slab.layerHi) ? 1u : 0u;
}

// staging area already holds `kept` records (from k_slab_pack or sph_slab_init)
int sphk_slab_rebuild(sph_solver* s, const uint32_t* recvDown, int nDown, const uint32_t* recvUp, int nUp, int kept) {
  const int total = kept + nDown + nUp;
  if (nDown) hipLaunchKernelGGL(k_slab_append, dim3(sph_blocks(nDown)), dim3(SPH_BLOCK), 0, s->stream, s->d, recvDown, nDown, kept);
  if (nUp) hipLaunchKernelGGL(k_slab_append, dim3(sph_blocks(nUp)), dim3(SPH_BLOCK), 0, s->stream, s->d, recvUp, nUp, kept + nDown);
  hipLaunchKernelGGL(k_iota, dim3(sph_blocks(total)), dim3(SPH_BLOCK), 0, s->stream, s->d.vals, total);
  int rc = sphk_sort_pairs(s, total, s->slab.globalIdBits);  // stable LSD radix sort by global id (keys), vals follow
  if (rc != SPH_OK) return rc;
  hipLaunchKernelGGL(k_slab_gather, dim3(sph_blocks(total)), dim3(SPH_BLOCK), 0, s->stream, s->d, s->slab, total);
  SPH_HIP(hipGetLastError());
  s->d.N = total;
  s->progress = 0;
  return SPH_OK;
}
