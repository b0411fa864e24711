// K1 clearBuffers + K5 findNeighbors (sphFluid.cl:64-92, 94-184, 207-329) for gfx950.
//
// Semantics reproduced exactly: per particle, two passes over the same 8 cells in the order
// own, x, y, z, xy, xz, yz, xyz (towards the nearer half of the cell on each axis); pass 0 builds a 30-bin radial
// histogram of candidates with d^2 <= h^2, from which r_thr is chosen so that at most 32 remain (or 31h/30 if fewer
// than 32 exist); pass 1 appends candidates with d^2 <= r_thr^2 in traversal order, at most 32.
//
// MI355X design: the reference walks ~640 candidates twice per particle and does the expensive part
// (sqrt, IEEE divide, histogram / store) under a branch that ~7 % of the lanes take but ~99 % of the waves execute.
// Here a 256-thread workgroup owns 256 consecutive sorted particles, stages the <= 9 contiguous runs of sorted particles
// that can contain their candidates (cells are contiguous along x in the sorted order) into LDS once, and each lane
//   1. walks its 8 cells ONCE with a cheap filter d^2 <= max(h, 31h/30)^2 (a superset of both reference passes) and
//      appends the LDS slot of every hit to a private list in LDS (compaction: ~45 of ~640 candidates survive),
//   2. replays pass 0 and pass 1 of the reference over that short list, in traversal order, with every lane busy,
//      using exactly the reference's float expressions (IEEE sqrt / divide), so r_thr, slot order and distances are
//      bit-identical.
// Lanes whose list overflows or whose cells are not fully staged (wrapped / aliased cells, LDS capacity) fall back to
// the literal two-pass walk over global memory (`find_neighbors_slow`), so the result is exact in every case.
#include <stdlib.h>
#include <string.h>

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

// The 8 cells of a particle in the reference's traversal order, as sorted-index ranges [lo, hi).
struct CellSet { int lo[8], hi[8], row[8], cell[8]; };

__device__ __forceinline__ void particle_cells(const SphDev& d, const float4 me, int myCell, CellSet& cs) {
  // which neighbour in x/y/z: -1 if the particle sits in the low half of its cell (sphFluid.cl:253-271)
  const float px = me.x - d.xmin, py = me.y - d.ymin, pz = me.z - d.zmin;
  const float cfx = (float)(int)(me.x * d.cellSizeInv) * d.cellSize;
  const float cfy = (float)(int)(me.y * d.cellSizeInv) * d.cellSize;
  const float cfz = (float)(int)(me.z * d.cellSizeInv) * d.cellSize;
  const int dx = ((px - cfx) < d.h) ? -1 : 1;
  const int dy = ((py - cfy) < d.h) ? -1 : 1;
  const int dz = ((pz - cfz) < d.h) ? -1 : 1;
  const int sy = dy * d.gx, sz = dz * d.gx * d.gy;
  const int off[8] = {0, dx, sy, sz, dx + sy, dx + sz, sy + sz, dx + sy + sz};
  const int ry = dy + 1, rz = dz + 1;  // staged-row ids: row = (y step + 1) + 3 * (z step + 1)
  const int rows[8] = {4, 4, ry + 3, 1 + 3 * rz, ry + 3, 1 + 3 * rz, ry + 3 * rz, ry + 3 * rz};
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int raw = myCell + off[k];
    int c = wrap_cell(raw, d.G);
    cs.row[k] = (c == raw) ? rows[k] : -1;  // wrapped cells are never staged
    c = min(max(c, 0), d.G - 1);            // no-op for particles inside the box; keeps the table read in range
    cs.cell[k] = c;
    cs.lo[k] = (int)d.cellStart[c];
    cs.hi[k] = (int)d.cellStart[c + 1];
  }
}

// Literal restatement of the reference's two passes for one particle, candidates read from global memory.
// `hist` points at 30 private counters with stride `hstride` (LDS).
__device__ __forceinline__ void find_neighbors_slow(const SphDev& d, int id, uint32_t* hist, int hstride) {
  const float4 me = d.sortedPos[id];
  CellSet cs;
  particle_cells(d, me, (int)d.keys[id], cs);
  for (int b = 0; b < SPH_RSEG; b++) hist[b * hstride] = 0u;
  const float h2 = d.h * d.h;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    for (int j = cs.lo[k]; j < cs.hi[k]; j++) {
      if (j == id) continue;
      const float4 o = d.sortedPos[j];
      const float ex = me.x - o.x, ey = me.y - o.y, ez = me.z - o.z;
      const float d2 = ex * ex + ey * ey + ez * ez;
      if (d2 <= h2) {
        const float dist = sqrtf(d2);
        const int bin = (int)(dist * (float)SPH_RSEG / d.h);
        if (bin < SPH_RSEG) hist[bin * hstride] += 1u;
      }
    }
  }
  int jb = 0, sum = 0;  // threshold, sphFluid.cl:310-323
  while (jb < SPH_RSEG) {
    sum += (int)hist[jb * hstride];
    if (sum == SPH_MAXN) break;
    if (sum > SPH_MAXN) { jb--; break; }
    jb++;
  }
  const float r_thr = (float)(jb + 1) * d.h / (float)SPH_RSEG;
  const float r2 = r_thr * r_thr;
  int found = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (found >= SPH_MAXN) continue;  // `if(spaceLeft>0)`, sphFluid.cl:145
    for (int j = cs.lo[k]; j < cs.hi[k]; j++) {
      if (j == id) continue;
      const float4 o = d.sortedPos[j];
      const float ex = me.x - o.x, ey = me.y - o.y, ez = me.z - o.z;
      const float d2 = ex * ex + ey * ey + ez * ez;
      if (d2 <= r2) {
        if (found >= SPH_MAXN) break;  // sphFluid.cl:169
        const size_t idx = nbr_index(id, found);
        d.nbrId[idx] = j;
        d.nbrDist[idx] = sqrtf(d2) * d.simScale;
        found++;
      }
    }
  }
  for (int k = found; k < SPH_MAXN; k++) {
    const size_t idx = nbr_index(id, k);
    d.nbrId[idx] = -1;
    d.nbrDist[idx] = -1.f;
  }
}

#define FN_PART 256         // particles per workgroup
#define FN_THREADS 512      // two lanes per particle: 8 waves = 2 per SIMD with one workgroup per CU
#define FN_CAND_CAP 5632    // staged candidates per workgroup (float4: 88 KB)
#define FN_CAND_PAD 4       // the 4-wide walk may read (never use) up to 3 slots past a cell
#define FN_LIST_CAP 48      // compaction list entries per lane (u16: 48 KB per workgroup)
#define FN_HIST_WORDS 15    // 30 bins as packed u16 pairs, one histogram per particle (15 KB per workgroup)

struct FnShared {
  float4 cand[FN_CAND_CAP + FN_CAND_PAD];         // x, y, z, sorted index (int bits)
  uint16_t list[FN_LIST_CAP][FN_THREADS];         // [entry][lane] LDS slots of the filter hits, traversal order
  uint32_t hist[FN_HIST_WORDS][FN_PART];          // [bin pair][particle]
  int rowLo[9], rowHi[9], rowBase[9];             // staged runs: sorted-index range and first LDS slot
};

// d.dbg layout: [0] particles handed to the fallback kernel because a cell was not staged, [1] because a list
// overflowed, [2] sum of staged candidates, [3] candidate runs dropped for LDS capacity, [4] fallback queue length.
//
// Lane pair (2p, 2p+1) serves particle p: lane `half` walks the cells k = half, half+2, half+4, half+6 of the
// reference's order, so the merged traversal order is A0 B1 A2 B3 A4 B5 A6 B7 and only four per-cell hit counts have
// to cross lanes (one DPP swap each) to place every neighbour in the reference's slot.
__global__ __launch_bounds__(FN_THREADS, 2) void k_find_neighbors(SphDev d, uint32_t* __restrict__ slowQueue, int experiment) {
  extern __shared__ __align__(16) unsigned char fn_smem[];
  FnShared& sh = *reinterpret_cast<FnShared*>(fn_smem);
  const int tid = threadIdx.x;
  const int p = tid >> 1, half = tid & 1;
  const int p0 = blockIdx.x * FN_PART;
  const int id = p0 + p;
  const bool alive = id < d.N;

  // ---- stage the candidate runs. Cells of this workgroup lie in [cLo, cHi]; row r = (sy+1) + 3*(sz+1) holds the cells
  // [cLo - 1, cHi + 1] shifted by sy*gx + sz*gx*gy, which is one contiguous run of sorted particles.
  if (tid < 9) {
    const int cLo = (int)d.keys[p0], cHi = (int)d.keys[min(p0 + FN_PART, d.N) - 1];
    const int sy = tid % 3 - 1, sz = tid / 3 - 1;
    const int shift = sy * d.gx + sz * d.gx * d.gy;
    const int a = max(cLo - 1 + shift, 0), b = min(cHi + 1 + shift, d.G - 1);
    int lo = 0, hi = 0;
    if (a <= b) { lo = (int)d.cellStart[a]; hi = (int)d.cellStart[b + 1]; }
    sh.rowLo[tid] = lo; sh.rowHi[tid] = hi;
  }
  __syncthreads();
  if (tid == 0) {
    int base = 0;
    // the own row (4) first, then the others: if LDS runs out, the most-used runs are the ones that are staged
    const int order[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8};
    for (int q = 0; q < 9; q++) {
      const int r = order[q];
      const int n = sh.rowHi[r] - sh.rowLo[r];
      sh.rowBase[r] = base;
      if (base + n > FN_CAND_CAP) { sh.rowHi[r] = sh.rowLo[r]; atomicAdd(&d.dbg[3], 1u); continue; }  // not staged
      base += n;
    }
    atomicAdd(&d.dbg[2], (uint32_t)base);
  }
  __syncthreads();
#pragma unroll 1
  for (int r = 0; r < 9; r++) {
    const int lo = sh.rowLo[r], n = sh.rowHi[r] - lo, base = sh.rowBase[r];
    for (int t = tid; t < n; t += FN_THREADS) {
      float4 q = d.sortedPos[lo + t];
      q.w = __int_as_float(lo + t);
      sh.cand[base + t] = q;
    }
  }
  __syncthreads();

  if (experiment == 1) return;  // timing experiment: staging only
  float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
  bool slow = false;
  int ldsLo[4], num[4];
#pragma unroll
  for (int i = 0; i < 4; i++) { num[i] = 0; ldsLo[i] = 0; }
  if (alive) {
    me = d.sortedPos[id];
    CellSet cs;
    particle_cells(d, me, (int)d.keys[id], cs);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int n = cs.hi[k] - cs.lo[k];
      int base = 0;
      if (n > 0) {
        const int r = cs.row[k];
        if (r >= 0 && cs.lo[k] >= sh.rowLo[r] && cs.hi[k] <= sh.rowHi[r]) base = sh.rowBase[r] + (cs.lo[k] - sh.rowLo[r]);
        else slow = true;  // a non-empty cell of this particle is not in LDS (both lanes of the pair see this)
      }
      if ((k & 1) == half) { num[k >> 1] = n; ldsLo[k >> 1] = base; }
    }
  }
  if (slow) {
#pragma unroll
    for (int i = 0; i < 4; i++) num[i] = 0;
    if (half == 0) atomicAdd(&d.dbg[0], 1u);
  }

  // ---- 1. single walk with the cheap filter; r_max covers pass 0 (h) and every possible pass-1 radius (<= 31h/30).
  // Four candidates per trip: the four LDS reads are issued together so their latency overlaps.
  const float rA = d.h, rB = (float)(SPH_RSEG + 1) * d.h / (float)SPH_RSEG;
  const float r2max = fmaxf(rA * rA, rB * rB);
  const int selfId = (half == 0) ? id : -1;  // only cell k = 0 (lane 0 of the pair, first cell) can hold the particle itself
  int cnt = 0;
  int segEnd[4];
#define FN_VISIT(o, u)                                                                        \
  {                                                                                           \
    const float ex = me.x - (o).x, ey = me.y - (o).y, ez = me.z - (o).z;                      \
    const float d2 = ex * ex + ey * ey + ez * ez;                                             \
    bool hit = (d2 <= r2max) && (t + (u) < n);                                                \
    if (i == 0) hit = hit && (__float_as_int((o).w) != selfId);                               \
    if (hit) {                                                                                \
      if (cnt < FN_LIST_CAP) sh.list[cnt][tid] = (uint16_t)(base + t + (u));                  \
      cnt++;                                                                                  \
    }                                                                                         \
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int n = num[i], base = ldsLo[i];
    for (int t = 0; t < n; t += 4) {
      const float4 o0 = sh.cand[base + t], o1 = sh.cand[base + t + 1], o2 = sh.cand[base + t + 2], o3 = sh.cand[base + t + 3];
      FN_VISIT(o0, 0) FN_VISIT(o1, 1) FN_VISIT(o2, 2) FN_VISIT(o3, 3)
    }
    segEnd[i] = cnt;
  }
#undef FN_VISIT
  if (experiment == 2) { if (cnt == 12345) d.dbg[15] = 1; return; }  // timing experiment: staging + walk
  bool over = cnt > FN_LIST_CAP;
  over = over || (__shfl_xor((int)over, 1) != 0);
  if (over) {
    if (half == 0 && !slow) atomicAdd(&d.dbg[1], 1u);
    slow = true;
#pragma unroll
    for (int i = 0; i < 4; i++) segEnd[i] = 0;
  }

  // ---- 2. replay of the reference's two passes over the short lists, entirely in registers.
  // The list (<= 48 LDS slots per lane) is expanded once into d2[] / idx[] with static indices: all list reads, then all
  // candidate reads are in flight together, and pass 0 (histogram), the threshold and pass 1 (selection + slot
  // assignment) never touch LDS for candidate data again.
  const int total = segEnd[3];
  float d2v[FN_LIST_CAP];
  int idxv[FN_LIST_CAP];
#pragma unroll
  for (int c0 = 0; c0 < FN_LIST_CAP; c0 += 16) {
    if (__any(c0 < total)) {  // wave-uniform skip of empty 16-entry chunks
      uint32_t slot[16];
#pragma unroll
      for (int u = 0; u < 16; u++) slot[u] = min((uint32_t)sh.list[c0 + u][tid], (uint32_t)(FN_CAND_CAP + FN_CAND_PAD - 1));
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const float4 o = sh.cand[slot[u]];
        const float ex = me.x - o.x, ey = me.y - o.y, ez = me.z - o.z;
        d2v[c0 + u] = ex * ex + ey * ey + ez * ez;
        idxv[c0 + u] = __float_as_int(o.w);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 16; u++) { d2v[c0 + u] = 0.f; idxv[c0 + u] = -1; }
    }
  }
  // ---- 2a. pass 0: per-particle 30-bin histogram of the hits with d^2 <= h^2 (sphFluid.cl:157-161)
  for (int w = half; w < FN_HIST_WORDS; w += 2) sh.hist[w][p] = 0u;
  const float h2 = d.h * d.h;
#pragma unroll
  for (int e = 0; e < FN_LIST_CAP; e++) {
    if (e < total && d2v[e] <= h2) {
      const float dist = sqrtf(d2v[e]);
      const int bin = (int)(dist * (float)SPH_RSEG / d.h);
      if (bin < SPH_RSEG) atomicAdd(&sh.hist[bin >> 1][p], 1u << ((bin & 1) << 4));
    }
  }
  // ---- threshold (sphFluid.cl:310-323); both lanes of a pair read the same histogram (same wave: LDS ops are ordered)
  int jb = 0, sum = 0;
  while (jb < SPH_RSEG) {
    sum += (int)((sh.hist[jb >> 1][p] >> ((jb & 1) << 4)) & 0xffffu);
    if (sum == SPH_MAXN) break;
    if (sum > SPH_MAXN) { jb--; break; }
    jb++;
  }
  const float r_thr = (float)(jb + 1) * d.h / (float)SPH_RSEG;
  const float r2 = r_thr * r_thr;
  if (experiment == 3) { if (r2 == 12345.f) d.dbg[15] = 1; return; }  // timing experiment: + histogram pass

  // ---- 2b. pass 1: hits with d^2 <= r_thr^2. Count them per cell, swap the four counts inside the pair, then write
  // every hit at (hits in earlier cells of the merged order) + (rank inside its cell); slots >= 32 are dropped, which is
  // what the reference's `break` / `spaceLeft` logic amounts to (sphFluid.cl:145,168-169).
  int mine[4] = {0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < FN_LIST_CAP; e++) {
    const int acc = (e < total && d2v[e] <= r2) ? 1 : 0;
    mine[0] += (e < segEnd[0]) ? acc : 0;
    mine[1] += (e >= segEnd[0] && e < segEnd[1]) ? acc : 0;
    mine[2] += (e >= segEnd[1] && e < segEnd[2]) ? acc : 0;
    mine[3] += (e >= segEnd[2]) ? acc : 0;
  }
  int theirs[4];
#pragma unroll
  for (int i = 0; i < 4; i++) theirs[i] = __shfl_xor(mine[i], 1);
  if (alive && !slow) {
    int start[4], run = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {  // merged order: (lane 0, i), (lane 1, i), (lane 0, i+1), ...
      start[i] = run + (half ? theirs[i] : 0);
      run += mine[i] + theirs[i];
    }
    if (experiment == 5) { if (run == 12345) d.dbg[15] = 1; return; }  // timing experiment: no write loop
    // position of entry e = start of its cell + hits before it inside the cell
    int pos = start[0];
#pragma unroll
    for (int e = 0; e < FN_LIST_CAP; e++) {
      if (e == segEnd[0]) pos = start[1];  // (segments may be empty: later assignments win, in order)
      if (e == segEnd[1]) pos = start[2];
      if (e == segEnd[2]) pos = start[3];
      if (e < total && d2v[e] <= r2) {
        if (pos < SPH_MAXN) {
          const size_t idx = nbr_index(id, pos);
          if (experiment == 4) { if (sqrtf(d2v[e]) * d.simScale == 12345.f) d.dbg[15] = idxv[e]; }
          else {
            d.nbrId[idx] = idxv[e];
            d.nbrDist[idx] = sqrtf(d2v[e]) * d.simScale;
          }
        }
        pos++;
      }
    }
    for (int k = min(run, SPH_MAXN) + half; k < SPH_MAXN; k += 2) {  // K1 folded in: unused slots = (-1, -1)
      const size_t idx = nbr_index(id, k);
      d.nbrId[idx] = -1;
      d.nbrDist[idx] = -1.f;
    }
  }
  // ---- the rare particles the fast path cannot serve are queued for k_find_neighbors_fallback (exact, any input)
  if (alive && slow && half == 0) slowQueue[atomicAdd(&d.dbg[4], 1u)] = (uint32_t)id;
}

// Literal two-pass walk for the queued particles, at full occupancy (grid-stride over the queue).
__global__ __launch_bounds__(SPH_BLOCK) void k_find_neighbors_fallback(SphDev d, const uint32_t* __restrict__ slowQueue) {
  __shared__ uint32_t hist[SPH_RSEG][SPH_BLOCK];
  const uint32_t count = d.dbg[4];
  for (uint32_t q = blockIdx.x * SPH_BLOCK + threadIdx.x; q < count; q += gridDim.x * SPH_BLOCK)
    find_neighbors_slow(d, (int)slowQueue[q], &hist[0][threadIdx.x], SPH_BLOCK);
}

// Reference-shaped kernel (one lane per particle, everything from global memory); kept for A/B timing
// (SPHMI_FIND_NEIGHBORS=v1) and as the semantic baseline of the fast kernel.
__global__ __launch_bounds__(SPH_BLOCK) void k_find_neighbors_v1(SphDev d) {
  __shared__ uint32_t hist[SPH_RSEG][SPH_BLOCK];
  const int id = blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (id >= d.N) return;
  find_neighbors_slow(d, id, &hist[0][threadIdx.x], SPH_BLOCK);
}

int sphk_find_neighbors(sph_solver* s) {
  static int variant = -1, experiment = 0;
  if (variant < 0) {
    const char* x = getenv("SPHMI_FN_EXPERIMENT");
    experiment = x ? atoi(x) : 0;
    const char* e = getenv("SPHMI_FIND_NEIGHBORS");
    variant = (e && !strcmp(e, "v1")) ? 1 : 2;
  }
  if (variant == 1) {
    hipLaunchKernelGGL(k_find_neighbors_v1, dim3(sph_blocks(s->d.N)), dim3(SPH_BLOCK), 0, s->stream, s->d);
  } else {
    static bool attrSet = false;
    if (!attrSet) {
      SPH_HIP(hipFuncSetAttribute((const void*)k_find_neighbors, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(FnShared)));
      attrSet = true;
    }
    // the fallback queue reuses keysAlt (N words, idle between the sort and the next step's sort)
    SPH_HIP(hipMemsetAsync(&s->d.dbg[4], 0, sizeof(uint32_t), s->stream));
    hipLaunchKernelGGL(k_find_neighbors, dim3(sph_blocks(s->d.N, FN_PART)), dim3(FN_THREADS), sizeof(FnShared), s->stream, s->d, s->d.keysAlt, experiment);
    hipLaunchKernelGGL(k_find_neighbors_fallback, dim3(min(sph_blocks(s->d.N), 2048)), dim3(SPH_BLOCK), 0, s->stream, s->d, s->d.keysAlt);
  }
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}
