// K1 clearBuffers + K5 findNeighbors (sphFluid.cl:64-92, 94-184, 207-329).
//
// Semantics reproduced exactly: per particle, two passes over the same 8 cells in the order
// own, x, y, z, xy, xz, yz, xyz (towards the nearer half of the cell on each axis); pass 0 builds a 30-bin radial
// histogram of candidates with d^2 <= h^2, from which r_thr is chosen so that at most 32 remain (or 31h/30 if fewer
// than 32 exist); pass 1 appends candidates with d^2 <= r_thr^2 in traversal order, at most 32.
#include "sph_common.h"

__global__ __launch_bounds__(SPH_BLOCK) void k_clear_neighbors(int32_t* __restrict__ nbrId, float* __restrict__ nbrDist,
                                                                size_t n4) {
  const size_t i = (size_t)blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (i >= n4) return;
  reinterpret_cast<int4*>(nbrId)[i] = make_int4(-1, -1, -1, -1);
  reinterpret_cast<float4*>(nbrDist)[i] = make_float4(-1.f, -1.f, -1.f, -1.f);
}

int sphk_clear_neighbors(sph_solver* s) {
  const size_t n4 = (size_t)s->numTiles * 64 * 8;
  hipLaunchKernelGGL(k_clear_neighbors, dim3((unsigned)((n4 + SPH_BLOCK - 1) / SPH_BLOCK)), dim3(SPH_BLOCK), 0, s->stream,
                     s->d.nbrId, s->d.nbrDist, n4);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

__device__ __forceinline__ int wrap_cell(int c, int G) {  // searchCell, sphFluid.cl:94-112
  if (c < 0) c += G;
  if (c >= G) c -= G;
  return c;
}

// v1: one lane per sorted particle, candidates read straight from global memory (L1/L2), histogram in LDS.
__global__ __launch_bounds__(SPH_BLOCK) void k_find_neighbors(SphDev d) {
  __shared__ uint32_t hist[SPH_RSEG][SPH_BLOCK];  // [bin][thread]: conflict-free, 30 KB
  const int tid = threadIdx.x;
  const int id = blockIdx.x * SPH_BLOCK + tid;
  if (id >= d.N) return;
  const float4 me = d.sortedPos[id];
  const int myCell = (int)d.keys[id];  // == (int)sortedPosition.w & mask of the reference (sphFluid.cl:229)
#pragma unroll
  for (int b = 0; b < SPH_RSEG; b++) hist[b][tid] = 0u;

  // which neighbour in x/y/z: -1 if the particle sits in the low half of its cell (sphFluid.cl:253-271)
  const float px = me.x - d.xmin, py = me.y - d.ymin, pz = me.z - d.zmin;
  const float cfx = (float)(int)(me.x * d.cellSizeInv) * d.cellSize;
  const float cfy = (float)(int)(me.y * d.cellSizeInv) * d.cellSize;
  const float cfz = (float)(int)(me.z * d.cellSizeInv) * d.cellSize;
  const int dx = ((px - cfx) < d.h) ? -1 : 1;
  const int dy = ((py - cfy) < d.h) ? -1 : 1;
  const int dz = ((pz - cfz) < d.h) ? -1 : 1;
  const int sy = dy * d.gx, sz = dz * d.gx * d.gy;
  int cells[8];
  cells[0] = myCell;
  cells[1] = wrap_cell(myCell + dx, d.G);
  cells[2] = wrap_cell(myCell + sy, d.G);
  cells[3] = wrap_cell(myCell + sz, d.G);
  cells[4] = wrap_cell(myCell + dx + sy, d.G);
  cells[5] = wrap_cell(myCell + dx + sz, d.G);
  cells[6] = wrap_cell(myCell + sy + sz, d.G);
  cells[7] = wrap_cell(myCell + dx + sy + sz, d.G);
  int lo[8], hi[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int c = min(max(cells[k], 0), d.G - 1);  // no-op for particles inside the box; keeps the table read in range
    lo[k] = (int)d.cellStart[c];
    hi[k] = (int)d.cellStart[c + 1];
  }

  // ---- pass 0: radial histogram of candidates within h
  const float h2 = d.h * d.h;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    for (int j = lo[k]; j < hi[k]; j++) {
      if (j == id) continue;
      const float4 o = d.sortedPos[j];
      const float ex = me.x - o.x, ey = me.y - o.y, ez = me.z - o.z;
      const float d2 = ex * ex + ey * ey + ez * ez;
      if (d2 <= h2) {
        const float dist = sqrtf(d2);
        const int bin = (int)(dist * (float)SPH_RSEG / d.h);
        if (bin < SPH_RSEG) hist[bin][tid] += 1u;
      }
    }
  }
  // ---- threshold (sphFluid.cl:310-323)
  int jb = 0, sum = 0;
  while (jb < SPH_RSEG) {
    sum += (int)hist[jb][tid];
    if (sum == SPH_MAXN) break;
    if (sum > SPH_MAXN) { jb--; break; }
    jb++;
  }
  const float r_thr = (float)(jb + 1) * d.h / (float)SPH_RSEG;
  const float r2 = r_thr * r_thr;

  // ---- pass 1: keep the first 32 candidates within r_thr, traversal order
  int found = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    // `if(spaceLeft>0)` guards the whole cell; the inner `break` leaves only this cell's loop (sphFluid.cl:145,169)
    if (found >= SPH_MAXN) continue;
    for (int j = lo[k]; j < hi[k]; j++) {
      if (j == id) continue;
      const float4 o = d.sortedPos[j];
      const float ex = me.x - o.x, ey = me.y - o.y, ez = me.z - o.z;
      const float d2 = ex * ex + ey * ey + ez * ez;
      if (d2 <= r2) {
        if (found >= SPH_MAXN) break;
        const size_t idx = nbr_index(id, found);
        d.nbrId[idx] = j;
        d.nbrDist[idx] = sqrtf(d2) * d.simScale;
        found++;
      }
    }
  }
  for (int k = found; k < SPH_MAXN; k++) {  // K1 folded in: unused slots = (-1, -1)
    const size_t idx = nbr_index(id, k);
    d.nbrId[idx] = -1;
    d.nbrDist[idx] = -1.f;
  }
}

int sphk_find_neighbors(sph_solver* s) {
  hipLaunchKernelGGL(k_find_neighbors, dim3(sph_blocks(s->d.N)), dim3(SPH_BLOCK), 0, s->stream, s->d);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}
