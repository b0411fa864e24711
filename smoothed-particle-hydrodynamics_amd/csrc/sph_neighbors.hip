// K1 clearBuffers + K5 findNeighbors (sphFluid.cl:64-92, 94-184, 207-329) for gfx950.
//
// Semantics reproduced exactly: per particle, two passes over the same 8 cells in the order
// own, x, y, z, xy, xz, yz, xyz (towards the nearer half of the cell on each axis); pass 0 builds a 30-bin radial
// histogram of candidates with d^2 <= h^2, from which r_thr is chosen so that at most 32 remain (or 31h/30 if fewer
// than 32 exist); pass 1 appends candidates with d^2 <= r_thr^2 in traversal order, at most 32.
//
// MI355X design. The reference walks ~640 candidates twice per particle and does the expensive part
// (sqrt, IEEE divide, histogram / store) under a branch that ~7 % of the lanes take but ~99 % of the waves execute.
// Here a 256-thread workgroup owns 128 consecutive sorted particles, TWO lanes per particle (FN_LANES; four were measured:
// slower), stages the <= 9 contiguous runs of sorted particles that can contain their candidates (cells are contiguous along x
// in the sorted order) into LDS once as SoA x/y/z, and each lane
//   1. walks four of the particle's 8 cells ONCE (~280 candidates), four candidates per trip (16-byte aligned LDS reads,
//      packed-f32 math, next quad prefetched), with the cheap filter d^2 <= max(h, 31h/30)^2 — a superset of both reference
//      passes — and appends the LDS slot of every hit to a private u16 list in LDS (~20 of ~280 survive),
//   2. replays pass 0 / threshold / pass 1 of the reference over that short list in registers (48 entries), in traversal
//      order, with exactly the reference's float expressions, so r_thr, slot order and distances are bit-identical. Pass 0
//      does not build the histogram: the cumulative counts it needs are "d^2 < U[j]" tests against 30 thresholds computed
//      exactly on the host (binU), searched by bisection; counts are combined inside the particle's lanes with DPP adds,
//   3. stores the row: ids as 16-bit offsets (sph_common.h, SPH_N16_*) and d^2, staged per wave in LDS as [particle][slot],
//      leave as 8- and 16-byte vectors per lane (square roots taken on the way out).
// Particles whose list overflows or whose cells are not fully staged (wrapped / aliased cells, LDS capacity) are served at
// the end of the batch by the literal two-pass walk, one WAVE per particle, reading the staged candidates from LDS where
// they are staged and from global memory otherwise — so the result is exact for any input and no second kernel or queue
// is needed.
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <type_traits>

#include "sph_common.h"

__global__ __launch_bounds__(SPH_BLOCK) void k_clear_neighbors(int32_t* __restrict__ nbrId, float* __restrict__ nbrDist,
                                                                uint16_t* __restrict__ nbr16, size_t n4) {
  const size_t i = (size_t)blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (i >= n4) return;
  reinterpret_cast<int4*>(nbrId)[i] = make_int4(-1, -1, -1, -1);
  reinterpret_cast<float4*>(nbrDist)[i] = make_float4(-1.f, -1.f, -1.f, -1.f);
  if (i < n4 / 2) reinterpret_cast<int4*>(nbr16)[i] = make_int4(-1, -1, -1, -1);  // 2 bytes per slot: all SPH_N16_EMPTY
}

int sphk_clear_neighbors(sph_solver* s) {
  const size_t n4 = (size_t)((s->d.N + SPH_TILE - 1) / SPH_TILE) * 64 * 8;
  hipLaunchKernelGGL(k_clear_neighbors, dim3((unsigned)((n4 + SPH_BLOCK - 1) / SPH_BLOCK)), dim3(SPH_BLOCK), 0, s->stream,
                     s->d.nbrId, s->d.nbrDist, s->d.nbr16, n4);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

// What the search kernel needs of SphDev (a by-value SphDev costs ~100 SGPRs, half of them spilled). The arrays are separate
// `__restrict__` kernel arguments so that loads with a workgroup-uniform address become scalar loads.
struct FnParams {
  int G, gx, gy, rangeLo, rangeHi;
  float h, cellSize, cellSizeInv, simScale, xmin, ymin, zmin;
};
struct FnArrays {  // (what the exact walk needs; built inside the kernel from its arguments)
  const float4* sortedPos;
  const uint32_t *keys, *cellStart;
  int32_t* nbrId;
  float* nbrDist;
  uint16_t* nbr16;
};

__device__ __forceinline__ int wrap_cell(int c, int G) {  // searchCell, sphFluid.cl:94-112
  if (c < 0) c += G;
  if (c >= G) c -= G;
  return c;
}

#ifndef FN_PART
#define FN_PART 128               // particles per workgroup (A/B: 256 particles / 512 threads / one workgroup per CU)
#endif
#ifndef FN_LANES
#define FN_LANES 2                // lanes per particle: 2 (a pair; whole cells) or 4 (a DPP quad: pairs x cell halves). A/B on MI355X,
#endif                            // config #2: 2 lanes 0.44 ms, 4 lanes 0.58 ms (more waves per SIMD, but 30 % more instructions)
#define FN_LOG_LANES (FN_LANES == 4 ? 2 : 1)
#define FN_THREADS (FN_LANES * FN_PART)
#define FN_PER_WAVE (64 / FN_LANES)  // particles per wave
#define FN_WAVES (FN_THREADS / 64)
#ifndef FN_CAND_CAP
#define FN_CAND_CAP 4096          // staged candidates per workgroup (SoA x/y/z: 48 KB; + lists 24 KB -> two workgroups per CU)
#endif
#ifndef FN_STAGE_PER
#define FN_STAGE_PER 8           // candidate records per thread and staging round, all loads of a round in flight (A/B: 8 0.429, 12 0.440, 16 0.436 ms)
#endif
#define FN_WIN 16                 // cell-table window per row: covers batches that span up to 13 cells (else: direct table reads)
#define FN_CAND_PAD 16            // the aligned, prefetching 4-wide walk reads (never uses) up to 15 slots past a piece
#ifndef FN_LIST_CAP
#define FN_LIST_CAP (96 / FN_LANES)  // compaction list entries per lane (u16 [entry][lane]); 96 per particle
#endif
#ifndef FN_CHUNK
#define FN_CHUNK 8                 // list entries per wave-uniform skip test in the replay loops (A/B: 8 0.419 ms, 4 0.445 ms)
#endif
#define FN_SLOTS_PER_LANE (SPH_MAXN / FN_LANES)  // map slots every lane of a particle finishes (square root + store)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FN_DST_STRIDE 36           // floats per particle in the d^2 staging area (32 slots + 16 B: keeps rows 16-byte aligned)
// rows (of 64 u16) of a wave's list area: the lists, and later per particle FN_DST_STRIDE floats of d^2 + 32 u16 of encoded ids
#define FN_STAGE_BYTES (FN_PER_WAVE * (FN_DST_STRIDE * 4 + SPH_MAXN * 2))
#define FN_LIST_ROWS ((FN_STAGE_BYTES + 127) / 128 > FN_LIST_CAP ? (FN_STAGE_BYTES + 127) / 128 : FN_LIST_CAP)

struct FnShared {
  float x[FN_CAND_CAP + FN_CAND_PAD], y[FN_CAND_CAP + FN_CAND_PAD], z[FN_CAND_CAP + FN_CAND_PAD];  // staged candidates
  // per wave: [entry][lane] LDS slots of the filter hits, traversal order. Once a wave has expanded its lists into registers the
  // same 3 KB hold the d^2 of the accepted neighbours, [particle of the wave][slot], on their way to 16-byte map stores.
  uint16_t list[FN_WAVES][FN_LIST_ROWS][64];  // (FN_LIST_CAP rows of list; a few more so that the d^2 and id staging of the store phase fit)
  float binU[64];                                 // U[j]: d^2 < U[j]  <=>  counted in histogram bins 0..j; r_thr^2 table; filter radius (see SphDev::binU)
  int rowLo[9], rowHi[9], rowBase[9];             // staged runs: sorted-index range and first LDS slot
  int win[9][FN_WIN];                             // cellStart[] of the cells cLo-1 .. of every row (see the staging code)
  uint32_t hist[FN_WAVES][32];                    // radial histograms of the wave-per-particle walk
#ifdef FN_STAMPS
  uint32_t stamps[16];                            // diagnostic build: cycles per phase, summed over the workgroup's waves
#endif
};
static_assert(FN_LIST_ROWS >= FN_LIST_CAP, "the wave's list area holds the lists first, the staging of the store phase afterwards");
static_assert(FN_LANES == 2 || FN_LANES == 4, "two or four lanes per particle");

// DPP moves inside a quad (4 consecutive lanes = the lanes of one particle): value of lane (l ^ 1), (l ^ 2)
__device__ __forceinline__ int quad_xor1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true); }  // quad_perm [1,0,3,2]
__device__ __forceinline__ int quad_xor2(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true); }  // quad_perm [2,3,0,1]
// the lanes of one particle: value held by the lane of the OTHER PAIR (same half) / by the lane of the other HALF (same pair; none with two lanes)
__device__ __forceinline__ int grp_other_pair(int v) { return FN_LANES == 4 ? quad_xor2(v) : quad_xor1(v); }
__device__ __forceinline__ int grp_other_half(int v) { return FN_LANES == 4 ? quad_xor1(v) : 0; }
typedef typename std::conditional<(FN_LIST_CAP > 32), unsigned long long, uint32_t>::type fn_mask_t;  // one bit per list entry
__device__ __forceinline__ int fn_popc(uint32_t v) { return __popc(v); }
__device__ __forceinline__ int fn_popc(unsigned long long v) { return __popcll(v); }

// d.dbg layout: [0] particles handed to the exact wave-per-particle walk because a cell was not staged, [1] because a list
// overflowed, [2] rows without a 16-bit copy because an offset was out of range, [3] candidate runs dropped for LDS capacity;
// [6] non-finite coordinates seen by the hash kernel (sph_api.hip: check_finite_state; not cleared by sph_reset_stage_times);
// [8] (wave, batch) pairs of k_pressure_force that left the short division / square-root path (sph_pcisph.hip). With FN_STAMPS (diagnostic build only): [16..25] cycles / 64 per phase.
#ifdef FN_STAMPS
#define FN_STAMP(ph)                                                                                      \
  {                                                                                                       \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                           \
    if ((threadIdx.x & 63) == 0) atomicAdd(&sh.stamps[(ph)], (uint32_t)(t_ - stamp_));                    \
    stamp_ = __builtin_amdgcn_s_memtime();                                                                \
  }
#else
#define FN_STAMP(ph)
#endif

// The literal reference walk for ONE particle by one wave (the 64 lanes take consecutive candidates of a cell): pass 0 fills a
// 30-bin histogram (LDS atomics), every lane derives the same r_thr, pass 1 assigns slots in traversal order with a ballot
// prefix (lane order == candidate order inside a chunk) and stops at 32. Candidates come from the staged LDS copy where the
// cell is staged (the usual case: a list overflowed) and from global memory otherwise (wrapped / aliased / dropped cells).
__device__ __forceinline__ void fn_exact_walk(const FnParams& d, const FnArrays& g, FnShared& sh, int id, int wave, int lane) {
  const float4 me = g.sortedPos[id];
  const int myCell = (int)g.keys[id];
  int lo[8], hi[8], ldsBase[8];  // sorted-index range of each cell; first LDS slot or -1 if the cell is not staged
  {
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
      const int r = (c == raw) ? rows[k] : -1;  // wrapped cells are never staged
      c = min(max(c, 0), d.G - 1);              // no-op for particles inside the box; keeps the table read in range
      lo[k] = (int)g.cellStart[c];
      hi[k] = (int)g.cellStart[c + 1];
      if (hi[k] - lo[k] > (1 << 20)) hi[k] = lo[k];  // a million particles in one cell: a blown-up state (counted by k_hash); do not walk it
      ldsBase[k] = -1;
#ifndef FN_EXACT_GLOBAL
      if (r >= 0 && hi[k] > lo[k] && lo[k] >= sh.rowLo[r] && hi[k] <= sh.rowHi[r]) ldsBase[k] = sh.rowBase[r] + (lo[k] - sh.rowLo[r]);
#endif
    }
  }
  if (lane < 32) sh.hist[wave][lane] = 0u;
  const unsigned long long ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const float h2 = d.h * d.h;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    for (int j0 = lo[k]; j0 < hi[k]; j0 += 64) {
      const int j = j0 + lane;
      if (j < hi[k] && j != id) {
        float ox, oy, oz;
        if (ldsBase[k] >= 0) { const int sl = ldsBase[k] + (j - lo[k]); ox = sh.x[sl]; oy = sh.y[sl]; oz = sh.z[sl]; }
        else { const float4 o = g.sortedPos[j]; ox = o.x; oy = o.y; oz = o.z; }
        const float ex = me.x - ox, ey = me.y - oy, ez = me.z - oz;
        const float d2 = ex * ex + ey * ey + ez * ez;
        if (d2 <= h2) {
          const float dist = sqrtf(d2);
          const int bin = (int)(dist * (float)SPH_RSEG / d.h);
          if (bin < SPH_RSEG) atomicAdd(&sh.hist[wave][bin], 1u);
        }
      }
    }
  }
  int jb = 0, sum = 0;  // threshold, sphFluid.cl:310-323 (LDS ops of one wave retire in order: the adds are visible)
  while (jb < SPH_RSEG) {
    sum += (int)sh.hist[wave][jb];
    if (sum == SPH_MAXN) break;
    if (sum > SPH_MAXN) { jb--; break; }
    jb++;
  }
  const float r_thr = (float)(jb + 1) * d.h / (float)SPH_RSEG;
  const float r2 = r_thr * r_thr;
  int found = 0;  // wave-uniform
#pragma unroll
  for (int k = 0; k < 8; k++) {
    for (int j0 = lo[k]; j0 < hi[k] && found < SPH_MAXN; j0 += 64) {
      const int j = j0 + lane;
      float d2 = 0.f;
      bool hit = false;
      if (j < hi[k] && j != id) {
        float ox, oy, oz;
        if (ldsBase[k] >= 0) { const int sl = ldsBase[k] + (j - lo[k]); ox = sh.x[sl]; oy = sh.y[sl]; oz = sh.z[sl]; }
        else { const float4 o = g.sortedPos[j]; ox = o.x; oy = o.y; oz = o.z; }
        const float ex = me.x - ox, ey = me.y - oy, ez = me.z - oz;
        d2 = ex * ex + ey * ey + ez * ez;
        hit = d2 <= r2;
      }
      const unsigned long long m = __ballot(hit);
      const int pos = found + __popcll(m & ltMask);
      if (hit && pos < SPH_MAXN) {
        const size_t idx = nbr_index(id, pos);
        g.nbrId[idx] = j;
        g.nbrDist[idx] = sqrtf(d2) * d.simScale;
      }
      found += __popcll(m);
    }
  }
  if (lane >= min(found, SPH_MAXN) && lane < SPH_MAXN) {
    const size_t idx = nbr_index(id, lane);
    g.nbrId[idx] = -1;
    g.nbrDist[idx] = -1.f;
  }
  if (lane == 0) g.nbr16[nbr_index(id, 0)] = (uint16_t)SPH_N16_WIDE;  // no 16-bit copy of this row: readers take the 32-bit ids
}

// Quad (4p .. 4p+3) serves particle p. Lanes 0,1 (pair A) walk the cells k = 0,5,6,7 of the reference's order and lanes 2,3
// (pair B) the cells 1,2,3,4; inside a pair, lane `sub` = 0 takes the first half of every cell (in sorted-index order) and
// lane 1 the second half. The merged traversal order is therefore
//   A0.0 A0.1 | B1.0 B1.1 B2.0 B2.1 B3.0 B3.1 B4.0 B4.1 | A5.0 A5.1 A6.0 A6.1 A7.0 A7.1
// and four per-piece hit counts per lane (one packed word, two DPP moves) place every neighbour in the reference's slot.
// Pair A gets 54 % of the hits (own cell ~42 %, edges 5 % each, corner 2 %), pair B 46 % (faces ~14 % each, one edge).
//
// Synchronisation: ONE workgroup barrier per batch, between the staging of the candidates and their use. Everything before it
// (batch bounds, the 9 candidate runs, their LDS layout) is computed redundantly by every wave from loads with workgroup-uniform
// addresses, everything after it is private to a wave (its lists, its d^2 staging area, the exact walks of its own particles).
__global__ __launch_bounds__(FN_THREADS, (FN_LANES == 4 ? 4 : (FN_PART > 128 ? 1 : 2))) void k_find_neighbors(FnParams d, const float4* __restrict__ sortedPos,
                                                                   const uint32_t* __restrict__ keys,
                                                                   const uint32_t* __restrict__ cellStart,
                                                                   const float* __restrict__ binU, int32_t* __restrict__ nbrId,
                                                                   float* __restrict__ nbrDist, uint16_t* __restrict__ nbr16,
                                                                   int32_t* __restrict__ nbrBase, uint32_t* __restrict__ dbg,
                                                                   uint32_t* __restrict__ trace) {  // trace: FN_STAMPS builds only
  extern __shared__ __align__(16) unsigned char fn_smem[];
  FnShared& sh = *reinterpret_cast<FnShared*>(fn_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = tid >> FN_LOG_LANES, quadLane = tid & (FN_LANES - 1);  // quadLane: lane inside the particle's group
  const int pairB = FN_LANES == 4 ? quadLane >> 1 : quadLane, sub = FN_LANES == 4 ? (quadLane & 1) : 0;
  const int rangeBegin = (int)cellStart[d.rangeLo], rangeEnd = (int)cellStart[d.rangeHi];  // all particles, or fewer ghost layers
  // (an XCD-aware tile order — each XCD one contiguous eighth of the tiles — was measured: 0.441 vs 0.430 ms on config #2, no change at 16.5 M)
  const int p0 = rangeBegin + blockIdx.x * FN_PART;
  if (p0 >= rangeEnd) return;  // uniform
  const int id = p0 + p;
  const bool alive = id < rangeEnd;
  FnArrays g;
  g.sortedPos = sortedPos; g.keys = keys; g.cellStart = cellStart; g.nbrId = nbrId; g.nbrDist = nbrDist; g.nbr16 = nbr16;
#ifdef FN_STAMPS
  const unsigned long long wgStart_ = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  if (tid < 16) sh.stamps[tid] = 0u;
  __syncthreads();
  unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
#endif

  // ---- stage the candidate runs. Cells of a batch of particles lie in [cLo, cHi]; row r = (sy+1) + 3*(sz+1) holds the
  // cells [cLo - 1, cHi + 1] shifted by sy*gx + sz*gx*gy, which is one contiguous run of sorted particles. Normally the
  // batch is the whole workgroup (128 particles). Where those runs would not fit the LDS budget — the workgroup straddles
  // two x-rows of cells (the runs then span whole rows) or sits in very dense cells — it serves its particles in smaller
  // batches instead: at most 32 particles, never across an x-row boundary.
  const int blockHi = min(p0 + FN_PART, rangeEnd);
  // own record and cell: issued now, needed only after the staging
  const float4 myPos = alive ? sortedPos[id] : make_float4(0.f, 0.f, 0.f, 0.f);
  const int myCell = alive ? (int)keys[id] : 0;
  int batchLo = p0, retry = 0;  // uniform. retry 1: batches end at x-row boundaries; retry 2: and hold at most 32 particles
  while (batchLo < blockHi) {
  int batchHi = blockHi;
  if (retry) {  // every wave finds the end of the batch itself (same data, same answer)
    const unsigned row0 = keys[batchLo] / (unsigned)d.gx;
    // the workgroup's <= FN_PART particles, 64 per round: the first one that must not join ends the batch
    int firstStop = FN_PART;
#pragma unroll
    for (int q = FN_PART / 64 - 1; q >= 0; q--) {
      const int iq = batchLo + 64 * q + lane;
      const bool stopq = (iq >= blockHi) || (keys[min(iq, blockHi - 1)] / (unsigned)d.gx != row0) || (retry == 2 && (q > 0 || lane >= 32));
      const unsigned long long mq = __ballot(stopq);
      if (mq) firstStop = 64 * q + __ffsll((long long)mq) - 1;
    }
    batchHi = min(batchLo + firstStop, blockHi);
  }
  // The dependent global round trips below are on the workgroup's critical path: (1) the batch's first / last cell, (2) the
  // two ends of the 9 runs, (3) all candidate records at once, flat over the concatenated runs, together with the cell-table
  // window the particles will look their cells up in. (1) and (2) have uniform addresses.
  const int cLo = (int)keys[batchLo], cHi = (int)keys[batchHi - 1];
  const int nc = cHi - cLo + 3;  // cells per row: cLo-1 .. cHi+1; the window holds nc+1 table entries
  const bool winOk = nc + 1 <= FN_WIN;
  // run table: the own row (4) first, then the others — if LDS still runs out, the most-used runs are the ones that are staged
  constexpr int order[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8};
  int runLo[9], runN[9], total = 0;
  {
    // the 18 run ends in ONE round trip: lane 2q + e loads end e of run q, the values are then broadcast with v_readlane
    const int q = min(lane >> 1, 8);
    const int r = (int)((0x862071534ull >> (4 * q)) & 15ull);  // order[q]
    const int shift = (r % 3 - 1) * d.gx + (r / 3 - 1) * d.gx * d.gy;
    const int cell = ((lane & 1) ? cHi + 2 : cLo - 1) + shift;
    const int v = (int)cellStart[min(max(cell, 0), d.G)];  // (clamped table reads: an out-of-grid run is empty)
#pragma unroll
    for (int qq = 0; qq < 9; qq++) {
      const int lo = __builtin_amdgcn_readlane(v, 2 * qq), hi = __builtin_amdgcn_readlane(v, 2 * qq + 1);
      runLo[qq] = lo; runN[qq] = max(hi - lo, 0);
      total += runN[qq];
    }
  }
  if (total > FN_CAND_CAP) {
    // redo in smaller batches (nothing was written to LDS yet): first one batch per x-row of cells — a workgroup that straddles two
    // rows has runs that span whole rows —, then batches of at most 32 particles (very dense cells)
    if (retry < 2) { retry++; continue; }
    total = 0;  // a small batch that still does not fit (very dense cells): drop runs (rare; their particles take the exact walk)
#pragma unroll
    for (int q = 0; q < 9; q++) {
      if (total + runN[q] > FN_CAND_CAP) { runN[q] = 0; if (tid == 0) atomicAdd(&dbg[3], 1u); }
      total += runN[q];
    }
  }
  int se[9], sd[9];  // flat staging: LDS slots < se[q] belong to run q; sorted index = slot + sd[q]
  {
    int base = 0;
#pragma unroll
    for (int q = 0; q < 9; q++) { sd[q] = runLo[q] - base; base += runN[q]; se[q] = base; }
  }
  if (tid == 0) {  // the tables the per-particle setup and the exact walk index by row
#pragma unroll
    for (int q = 0; q < 9; q++) { sh.rowLo[order[q]] = runLo[q]; sh.rowHi[order[q]] = runLo[q] + runN[q]; sh.rowBase[order[q]] = se[q] - runN[q]; }
  }
  if (winOk && tid < 9 * FN_WIN) {
    const int r = tid / FN_WIN, i = tid % FN_WIN;
    const int shift = (r % 3 - 1) * d.gx + (r / 3 - 1) * d.gx * d.gy;
    if (i <= nc) sh.win[r][i] = (int)cellStart[min(max(cLo - 1 + shift + i, 0), d.G)];
  }
  if (tid >= 192 && tid < 256) sh.binU[tid - 192] = binU[tid - 192];
  FN_STAMP(0)
  {
    constexpr int PER = FN_LANES == 4 ? 6 : FN_STAGE_PER;  // records per thread and round, all loads of a round in flight together
#pragma unroll 1
    for (int f0 = 0; f0 < total; f0 += PER * FN_THREADS) {
      float4 rec[PER];
#pragma unroll
      for (int u = 0; u < PER; u++) {
        const int f = f0 + tid + u * FN_THREADS;
        int delta = sd[0];
#pragma unroll
        for (int q = 0; q < 8; q++) delta = (f >= se[q]) ? sd[q + 1] : delta;  // runs are laid out in order of q
        if (f < total) rec[u] = sortedPos[f + delta];
      }
#pragma unroll
      for (int u = 0; u < PER; u++) {
        const int f = f0 + tid + u * FN_THREADS;
        if (f < total) { sh.x[f] = rec[u].x; sh.y[f] = rec[u].y; sh.z[f] = rec[u].z; }
      }
    }
  }
  __syncthreads();
  FN_STAMP(3)
  const bool mine_now = alive && id >= batchLo && id < batchHi;  // this lane's particle belongs to the current batch
  bool slow = false;

  if (mine_now) {  // (whole quads: the four lanes of a particle agree)
  // (opaque to the optimiser: everything derived from the particle's record is otherwise hoisted out of the batch loop,
  // which runs once for almost every workgroup, and then spilled)
  float4 me = myPos;
  int myCellNow = myCell, idNow = id;
  asm volatile("" : "+v"(me.x), "+v"(me.y), "+v"(me.z), "+v"(myCellNow), "+v"(idNow));
  int pLo[4], pHi[4], absDelta[4];  // this lane's four pieces as LDS slot ranges; sorted index = LDS slot + absDelta
  int selfSlot = -1;
  int zLoMine = 0;  // pair B: first sorted index of cell 3, the z neighbour (base of the flagged 16-bit offsets)
  {  // the lane's four cells (sphFluid.cl:253-308) with the cell table read from the LDS window where possible
    const float px = me.x - d.xmin, py = me.y - d.ymin, pz = me.z - d.zmin;
    const float cfx = (float)(int)(me.x * d.cellSizeInv) * d.cellSize;
    const float cfy = (float)(int)(me.y * d.cellSizeInv) * d.cellSize;
    const float cfz = (float)(int)(me.z * d.cellSizeInv) * d.cellSize;
    const int dx = ((px - cfx) < d.h) ? -1 : 1;  // -1: the particle sits in the low half of its cell
    const int dy = ((py - cfy) < d.h) ? -1 : 1;
    const int dz = ((pz - cfz) < d.h) ? -1 : 1;
    const int sy = dy * d.gx, sz = dz * d.gx * d.gy;
    const int ry = dy + 1, rz = dz + 1;  // staged-row ids: row = (y step + 1) + 3 * (z step + 1)
    // pair A: cells 0 (own), 5 (xz), 6 (yz), 7 (xyz); pair B: cells 1 (x), 2 (y), 3 (z), 4 (xy)
    const bool stepX[4] = {pairB != 0, pairB == 0, false, true};
    const bool stepY[4] = {false, pairB != 0, pairB == 0, true};
    const bool stepZ[4] = {false, pairB == 0, true, pairB == 0};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int xs = stepX[i] ? dx : 0;
      const int raw = myCellNow + xs + (stepY[i] ? sy : 0) + (stepZ[i] ? sz : 0);
      const int row = (stepY[i] ? ry : 1) + 3 * (stepZ[i] ? rz : 1);
      const int c = wrap_cell(raw, d.G);
      int lo, hi;
      const bool staged = c == raw;  // wrapped cells are never staged: only their emptiness matters
      if (staged && winOk) {
        const int w = myCellNow - cLo + 1 + xs;  // in [0, nc): the batch's cells are cLo..cHi, the window starts at cLo-1
        lo = sh.win[row][w];
        hi = sh.win[row][w + 1];
      } else {  // wrapped cells and batches wider than the window
        const int cc = min(max(c, 0), d.G - 1);
        lo = (int)cellStart[cc];
        hi = (int)cellStart[cc + 1];
      }
      const int n = hi - lo;
      int base = 0;
      if (n > 0) {
        if (staged && lo >= sh.rowLo[row] && hi <= sh.rowHi[row]) base = sh.rowBase[row] + (lo - sh.rowLo[row]);
        else slow = true;  // a non-empty cell of this particle is not in LDS
      }
      // halves (four lanes per particle): the split point is rounded to a 16-byte boundary of the SoA arrays so the second
      // half starts aligned; with two lanes per particle a lane walks whole cells
      const int mid = FN_LANES == 4 ? min(base + n, max(base, (base + (n >> 1) + 2) & ~3)) : base + n;
      pLo[i] = sub ? mid : base;
      pHi[i] = sub ? base + n : mid;
      absDelta[i] = lo - base;
      if (i == 2) zLoMine = lo;
      if (i == 0 && pairB == 0) selfSlot = base + (idNow - lo);  // the particle itself sits in its own cell
    }
  }
  {  // both pairs must agree (inside a pair the lanes see the same cells). The DPP move is executed by ALL four lanes — under a
     // short-circuit `||` the lanes that are already slow would sit it out and their partners would read 0 from them.
    const int otherPair = grp_other_pair((int)slow);
    slow = slow || (otherPair != 0);
  }
  if (slow) {
#pragma unroll
    for (int i = 0; i < 4; i++) pHi[i] = pLo[i];
    if (quadLane == 0) atomicAdd(&dbg[0], 1u);
  }
#ifdef DIAG_NO_WALK  // timing only: nothing is walked (every list stays empty)
#pragma unroll
  for (int i = 0; i < 4; i++) pHi[i] = pLo[i];
#endif
  FN_STAMP(4)

  // ---- 1. single walk with the cheap filter; r_max covers pass 0 (h) and every possible pass-1 radius (<= 31h/30).
  // Four candidates per trip from 16-byte-aligned SoA reads (the walk starts at the aligned slot at or before the piece and
  // masks what lies outside it); the arithmetic is packed f32 (v_pk_*), IEEE per component.
  int cnt = 0;
  int segEnd[4];
  const f32x2 mx = {me.x, me.x}, my = {me.y, me.y}, mz = {me.z, me.z};
  // The filter only has to be a superset of both reference passes (the replay below decides with the reference's exact
  // expressions), so it evaluates d^2 - R^2 as one chain of fused multiply-adds starting at -R^2, R^2 = r_max^2 (1 + 2^-20): the
  // chain differs from the reference's separately rounded d^2 by < 2^-21 R^2 (three roundings of <= 2^-24 max(d^2, R^2) each, plus
  // the reference's own three), so every candidate with d^2 <= r_max^2 comes out strictly negative. A hit is the SIGN BIT of that
  // value, shifted into a per-lane bit mask with one v_alignbit per candidate; every 32 candidates, and at the end of a piece,
  // the few set bits are turned into list entries in traversal order.
  const float r2f = sh.binU[63];  // max(h, 31h/30)^2 * (1 + 2^-20), computed on the host
  const f32x2 nthr = {-r2f, -r2f};
  // m - c as fma(c, -1, m): the same value; v_pk_fma_f32 measures faster than v_pk_add_f32 in isolation (tools/micro/valu_rates.hip:
  // 5.5 vs 6.7 cycles per wave instruction at two waves per SIMD), in this kernel it is worth 0.6 % — what counts here is the
  // NUMBER of vector instructions (SQ_ACTIVE_INST_VALU = one slot each, 72 % of the SIMD's time), not their kind. The -1 is opaque
  // so that the compiler does not turn the fma back into a subtraction.
  float neg1s = -1.f;
  asm volatile("" : "+s"(neg1s));  // (a scalar pair; held in a vector pair instead: the same time)
  const f32x2 neg1 = {neg1s, neg1s};
  uint16_t (*const myList)[64] = sh.list[wave];
#ifdef FN_SUB_BY_ADD  // A/B: the differences as v_pk_add_f32 with negated operands (6.7 cycles each against 5.5 for v_pk_fma_f32)
#define FN_DIFF(m, c) ((m) - (c))
#else
#define FN_DIFF(m, c) __builtin_elementwise_fma((c), neg1, (m))
#endif
#define FN_TEST(X, Y, Z)                                                                        \
  {                                                                                             \
    const f32x2 ex0 = FN_DIFF(mx, X.xy), ex1 = FN_DIFF(mx, X.zw), ey0 = FN_DIFF(my, Y.xy), ey1 = FN_DIFF(my, Y.zw); \
    const f32x2 ez0 = FN_DIFF(mz, Z.xy), ez1 = FN_DIFF(mz, Z.zw);                               \
    const f32x2 s0 = __builtin_elementwise_fma(ez0, ez0, __builtin_elementwise_fma(ey0, ey0, __builtin_elementwise_fma(ex0, ex0, nthr))); \
    const f32x2 s1 = __builtin_elementwise_fma(ez1, ez1, __builtin_elementwise_fma(ey1, ey1, __builtin_elementwise_fma(ex1, ex1, nthr))); \
    acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(s0.x), 31);                            \
    acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(s0.y), 31);                            \
    acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(s1.x), 31);                            \
    acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(s1.y), 31);                            \
  }
  // k candidates (slots aBase .. aBase+k-1, first one in the highest of the k low bits of acc) -> list entries
#define FN_FLUSH(k)                                                                             \
  {                                                                                             \
    uint32_t m = ((k) == 0) ? 0u : (acc << (32 - (k)));   /* candidate t at bit 31 - t */       \
    const int lowT = pieceLo - aBase, hiT = pieceHi - aBase;                                    \
    if (lowT > 0) m &= 0xffffffffu >> lowT;               /* slots before the piece (first chunk) */ \
    if (hiT < 32) m &= ~(0xffffffffu >> hiT);             /* slots past the piece */            \
    if (i == 0) {                                         /* the particle itself */             \
      const unsigned ts = (unsigned)(selfSlot - aBase);                                         \
      if (ts < 32u) m &= ~(0x80000000u >> ts);                                                  \
    }                                                                                           \
    cnt += __popc(m);                                                                           \
    if (cnt > FN_LIST_CAP) m = 0u;                        /* overflow: the particle takes the exact walk, nothing more is written */ \
    while (m != 0u) {                                                                           \
      const int bpos = __clz((int)m);                                                           \
      *wr = (uint16_t)(aBase + bpos);                                                           \
      wr += 64;                                           /* next entry of this lane */         \
      m &= ~(0x80000000u >> bpos);                                                              \
    }                                                                                           \
  }
  uint16_t* wr = &myList[0][lane];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int pieceLo = pLo[i], pieceHi = pHi[i];
    if (pieceHi > pieceLo) {
      int a = pieceLo & ~3;  // aligned slot at or before the piece
      int aBase = a, k = 0;
      uint32_t acc = 0u;
      f32x4 X = *reinterpret_cast<const f32x4*>(&sh.x[a]);
      f32x4 Y = *reinterpret_cast<const f32x4*>(&sh.y[a]);
      f32x4 Z = *reinterpret_cast<const f32x4*>(&sh.z[a]);
      f32x4 Xb = *reinterpret_cast<const f32x4*>(&sh.x[a + 4]);
      f32x4 Yb = *reinterpret_cast<const f32x4*>(&sh.y[a + 4]);
      f32x4 Zb = *reinterpret_cast<const f32x4*>(&sh.z[a + 4]);
      while (a < pieceHi) {
        // two quads per trip; each quad's three LDS reads are issued a whole quad test before it is used (its registers are
        // refilled right after their last use). Reads and tests may run up to 15 slots past the piece (FN_CAND_PAD);
        // FN_FLUSH masks those bits. (A/B: hand-placed counted lgkmcnt waits instead of hipcc's changed nothing — the loop is
        // bound by the issue rate of its packed-f32 arithmetic, not by LDS latency.)
        FN_TEST(X, Y, Z)
        X = *reinterpret_cast<const f32x4*>(&sh.x[a + 8]);
        Y = *reinterpret_cast<const f32x4*>(&sh.y[a + 8]);
        Z = *reinterpret_cast<const f32x4*>(&sh.z[a + 8]);
        __builtin_amdgcn_sched_barrier(0);  // (keeps hipcc from sinking the reads to the end of the trip, next to their use; without: same time)
        FN_TEST(Xb, Yb, Zb)
        Xb = *reinterpret_cast<const f32x4*>(&sh.x[a + 12]);
        Yb = *reinterpret_cast<const f32x4*>(&sh.y[a + 12]);
        Zb = *reinterpret_cast<const f32x4*>(&sh.z[a + 12]);
        __builtin_amdgcn_sched_barrier(0);
        a += 8; k += 8;
        if (k == 32) { FN_FLUSH(32) aBase = a; k = 0; }
      }
      FN_FLUSH(k)
    }
    segEnd[i] = cnt;
  }
#undef FN_TEST
#undef FN_FLUSH
  FN_STAMP(5)
#ifdef DIAG_NO_REPLAY  // timing only: the lists are walked and then dropped (a 1-entry list per lane keeps the walk alive)
  { const int keep = min(cnt, 1); cnt = keep;
#pragma unroll
    for (int i = 0; i < 4; i++) segEnd[i] = min(segEnd[i], keep); }
#endif
  int over = cnt > FN_LIST_CAP ? 1 : 0;
  over |= grp_other_half(over);  // all lanes of the particle take part
  over |= grp_other_pair(over);
  if (over) {
    if (quadLane == 0 && !slow) atomicAdd(&dbg[1], 1u);
    slow = true;
#pragma unroll
    for (int i = 0; i < 4; i++) segEnd[i] = 0;
  }

  // ---- 2. replay of the reference's two passes over the short lists, entirely in registers: the list (<= 24 LDS slots
  // per lane) is expanded once into d2v[] with static indices, 8 entries at a time with a wave-uniform skip. The slots are
  // kept (two per register): the list area is about to be reused.
  const int total = segEnd[3];
  float d2v[FN_LIST_CAP];
  uint32_t slotPk[FN_LIST_CAP / 2];
#pragma unroll
  for (int c0 = 0; c0 < FN_LIST_CAP; c0 += FN_CHUNK) {
    if (__any(c0 < total)) {
      int slotv[FN_CHUNK];
#pragma unroll
      for (int u = 0; u < FN_CHUNK; u++) slotv[u] = min((int)myList[c0 + u][lane], FN_CAND_CAP + FN_CAND_PAD - 1);
#pragma unroll
      for (int u = 0; u < FN_CHUNK; u++) {
        const int sl = slotv[u];
        const float ex = me.x - sh.x[sl], ey = me.y - sh.y[sl], ez = me.z - sh.z[sl];
        const float v = ex * ex + ey * ey + ez * ez;
        d2v[c0 + u] = (c0 + u < total) ? v : __builtin_inff();  // +inf never passes a `<` / `<=` test below
      }
#pragma unroll
      for (int u = 0; u < FN_CHUNK; u += 2) slotPk[(c0 + u) >> 1] = (uint32_t)slotv[u] | ((uint32_t)slotv[u + 1] << 16);
    } else {
#pragma unroll
      for (int u = 0; u < FN_CHUNK; u++) d2v[c0 + u] = __builtin_inff();
#pragma unroll
      for (int u = 0; u < FN_CHUNK; u += 2) slotPk[(c0 + u) >> 1] = 0u;
    }
  }
  FN_STAMP(6)
  // ---- 2a. pass 0 + threshold (sphFluid.cl:157-161,310-323) without building the histogram: C(j) = number of hits in
  // bins 0..j = number of hits with d^2 < U[j] (U precomputed exactly on the host). The reference's loop stops at
  // j* = min{ j : C(j) >= 32 } with jb = j* if C(j*) == 32, j* - 1 if it overshoots, and jb = 30 if no such j exists.
  int liveEnd = FN_CHUNK;  // wave-uniform: entries at or past it are +inf in every lane, and the counting loops skip them
#pragma unroll
  for (int c0 = FN_CHUNK; c0 < FN_LIST_CAP; c0 += FN_CHUNK) liveEnd = __any(c0 < total) ? c0 + FN_CHUNK : liveEnd;
  int lo = 0, hi = SPH_RSEG, cAtHi = 0;
#pragma unroll 1
  for (int it = 0; it < 5; it++) {
    const int mid = (lo + hi) >> 1;  // lo == hi on converged lanes: harmless re-evaluation
    const float U = sh.binU[min(mid, SPH_RSEG - 1)];
    int c = 0;
#pragma unroll
    for (int c0 = 0; c0 < FN_LIST_CAP; c0 += FN_CHUNK) {
      if (c0 < liveEnd) {
#pragma unroll
        for (int u = 0; u < FN_CHUNK; u++) c += (d2v[c0 + u] < U) ? 1 : 0;
      }
    }
    c += grp_other_half(c);
    c += grp_other_pair(c);
    if (lo < hi) {
      if (c >= SPH_MAXN) { hi = mid; cAtHi = c; } else lo = mid + 1;
    }
  }
  const int jb = (lo >= SPH_RSEG) ? SPH_RSEG : ((cAtHi == SPH_MAXN) ? lo : lo - 1);
  const float r2 = sh.binU[32 + jb];  // r_thr^2 with r_thr = (float)(jb + 1) * h / 30 (sphFluid.cl:313-321), from the host
  FN_STAMP(7)

  // ---- 2b. pass 1: hits with d^2 <= r_thr^2 as a bit mask; per-piece counts by popcount, exchanged inside the quad as one
  // packed word; every hit goes to (hits in earlier pieces of the merged order) + (rank inside its piece). Slots >= 32
  // are dropped, which is what the reference's `break` / `spaceLeft` logic amounts to (sphFluid.cl:145,168-169).
  fn_mask_t acc = 0;
#pragma unroll
  for (int c0 = 0; c0 < FN_LIST_CAP; c0 += FN_CHUNK) {
    if (c0 < liveEnd) {
#pragma unroll
      for (int u = 0; u < FN_CHUNK; u++) acc |= (d2v[c0 + u] <= r2) ? ((fn_mask_t)1 << (c0 + u)) : (fn_mask_t)0;
    }
  }
  fn_mask_t below[4];  // bits of the entries before the end of piece i
#pragma unroll
  for (int i = 0; i < 4; i++) below[i] = (segEnd[i] >= (int)(8 * sizeof(fn_mask_t))) ? ~(fn_mask_t)0 : (((fn_mask_t)1 << segEnd[i]) - 1);
  int mine[4];
  mine[0] = fn_popc(acc & below[0]);
#pragma unroll
  for (int i = 1; i < 4; i++) mine[i] = fn_popc(acc & below[i] & ~below[i - 1]);
  const int packed = mine[0] | (mine[1] << 8) | (mine[2] << 16) | (mine[3] << 24);  // each <= 48
  const int partner = grp_other_half(packed);   // the other half of the same four cells (none with two lanes per particle)
  const int cellTot = packed + partner;         // whole-cell counts of this pair (bytes <= 48: no carry)
  const int otherTot = grp_other_pair(cellTot); // whole-cell counts of the other pair
  const int zLoOther = grp_other_pair(zLoMine);  // (a DPP move: executed by all lanes of the particle)
  // (the wave's list area is free from here on: every lane has its slots in slotPk, and LDS operations of a wave retire in order)
  if (!slow) {
    const int c0b = cellTot & 255, c1b = (cellTot >> 8) & 255, c2b = (cellTot >> 16) & 255, c3b = (cellTot >> 24) & 255;
    const int o0b = otherTot & 255;
    const int sumOther = o0b + ((otherTot >> 8) & 255) + ((otherTot >> 16) & 255) + ((otherTot >> 24) & 255);
    const int run = min(c0b + c1b + c2b + c3b + sumOther, SPH_MAXN);
    // pair A (cells 0 | 5 6 7): cell 0 starts at 0, cells 5.. after cell 0 and all of pair B;
    // pair B (cells 1 2 3 4): starts after cell 0 (= pair A's first cell)
    const int first = pairB ? o0b : 0, afterFirst = pairB ? 0 : sumOther;
    int start[4];
    start[0] = first + (sub ? (partner & 255) : 0);
    start[1] = first + c0b + afterFirst + (sub ? ((partner >> 8) & 255) : 0);
    start[2] = first + c0b + afterFirst + c1b + (sub ? ((partner >> 16) & 255) : 0);
    start[3] = first + c0b + afterFirst + c1b + c2b + (sub ? ((partner >> 24) & 255) : 0);
    // Ids go straight to the tiled map (4-byte stores); the d^2 of the accepted neighbours go to the wave's staging area
    // [particle][slot], from which every lane of the quad then takes 8 consecutive slots: 8 square roots per lane instead of
    // one per list entry, and the distances leave as two 16-byte stores per lane (which also write the -1 of the unused slots).
    // The 16-bit copy of the row (sph_common.h, SPH_N16_*): offsets from the particle itself, or — cells one z layer away: pair A's
    // 5 6 7, pair B's 3 — from the first particle of the z-neighbour cell. A piece's entries are its LDS slots + one constant, so
    // the range check is two comparisons per piece, not one per entry.
    const int zBase = pairB ? zLoMine : zLoOther;
    int rel16[4];
    int wide = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bool zf = pairB ? (i == 2) : (i != 0);
      rel16[i] = absDelta[i] - (zf ? zBase : idNow) + SPH_N16_BIAS;
      if (pHi[i] > pLo[i] && (pLo[i] + rel16[i] < 0 || pHi[i] - 1 + rel16[i] > SPH_N16_MAX)) wide = 1;
    }
    // (per entry below: the offset from the particle itself if that fits, else the flagged one — which fits where the checks above
    // passed; the reader only follows the flag, so no per-piece bookkeeping is needed)
    int own16 = SPH_N16_BIAS - idNow, far16 = SPH_N16_BIAS + 0x8000 - zBase;
    asm volatile("" : "+v"(own16), "+v"(far16));  // (kept in registers: rematerialised per entry they cost two more instructions each)
    wide |= grp_other_half(wide);  // (both lanes are in this branch: `slow` is agreed inside the particle)
    wide |= grp_other_pair(wide);
    int tidNow = tid;  // (opaque: keeps the address below from being hoisted to the kernel's prologue and spilled)
    asm volatile("" : "+v"(tidNow));
    float* const waveArea = reinterpret_cast<float*>(&sh.list[0][0][0]) + (tidNow >> 6) * (FN_LIST_ROWS * 64 / 2);
    float* const dstRow = waveArea + ((tidNow & 63) >> FN_LOG_LANES) * FN_DST_STRIDE;
    // the row's 16-bit entries, [particle of the wave][slot], behind the d^2 area; pre-filled with "empty" by the lane that will store them
    uint16_t* const idRow = reinterpret_cast<uint16_t*>(waveArea + FN_PER_WAVE * FN_DST_STRIDE) + ((tidNow & 63) >> FN_LOG_LANES) * SPH_MAXN;
#pragma unroll
    for (int v = 0; v < FN_SLOTS_PER_LANE / 8; v++)
      *reinterpret_cast<uint4*>(idRow + FN_SLOTS_PER_LANE * quadLane + 8 * v) = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    const uint32_t mapBase = ((uint32_t)(idNow >> 6) * (8u * 64u) + (uint32_t)(idNow & 63)) << 2;  // element index of slot 0 (< 2^32: N <= 2^27)
    // walk the entries in list order with a running (piece start, index delta, rank inside the piece)
    int curStart = start[0], curDelta = absDelta[0], rank = 0;
#pragma unroll
    for (int c0 = 0; c0 < FN_LIST_CAP; c0 += FN_CHUNK) {
      if (!__any(c0 < total)) continue;  // wave-uniform skip of empty chunks
#pragma unroll
      for (int u = 0; u < FN_CHUNK; u++) {
        const int e = c0 + u;
        if (e == segEnd[0]) { curStart = start[1]; curDelta = absDelta[1]; rank = 0; }  // (empty pieces cascade in order)
        if (e == segEnd[1]) { curStart = start[2]; curDelta = absDelta[2]; rank = 0; }
        if (e == segEnd[2]) { curStart = start[3]; curDelta = absDelta[3]; rank = 0; }
        if ((acc >> e) & 1) {
          const int pos = curStart + rank;
          rank++;
          if (pos < SPH_MAXN) {
            const int jn = (int)((slotPk[e >> 1] >> ((e & 1) * 16)) & 0xffffu) + curDelta;
            const int o16 = jn + own16;
            idRow[pos] = (uint16_t)((uint32_t)o16 <= (uint32_t)SPH_N16_MAX ? o16 : jn + far16);
            dstRow[pos] = d2v[e];
          }
        }
      }
    }
    if (__any(wide != 0)) {
    // (rare) rows without a 16-bit copy: the same walk once more, for the 32-bit ids
    int curStart = start[0], curDelta = absDelta[0], rank = 0;
#pragma unroll
    for (int c0 = 0; c0 < FN_LIST_CAP; c0 += FN_CHUNK) {
      if (!__any(c0 < total)) continue;  // wave-uniform skip of empty chunks
#pragma unroll
      for (int u = 0; u < FN_CHUNK; u++) {
        const int e = c0 + u;
        if (e == segEnd[0]) { curStart = start[1]; curDelta = absDelta[1]; rank = 0; }  // (empty pieces cascade in order)
        if (e == segEnd[1]) { curStart = start[2]; curDelta = absDelta[2]; rank = 0; }
        if (e == segEnd[2]) { curStart = start[3]; curDelta = absDelta[3]; rank = 0; }
        if ((acc >> e) & 1) {
          const int pos = curStart + rank;
          rank++;
          if (pos < SPH_MAXN) {
            const int jn = (int)((slotPk[e >> 1] >> ((e & 1) * 16)) & 0xffffu) + curDelta;
            if (wide) nbrId[mapBase + (uint32_t)(((pos >> 2) << 8) + (pos & 3))] = jn;  // tiled map: group stride 64 lanes * 4 slots
          }
        }
      }
    }
    }
    if (wide) {  // K1 folded in: unused slots of the 32-bit row (the 16-bit ones are pre-filled)
      for (int k = run + quadLane; k < SPH_MAXN; k += FN_LANES) nbrId[mapBase + (uint32_t)(((k >> 2) << 8) + (k & 3))] = -1;
    }
    if (quadLane == 0) nbrBase[idNow] = zBase;
    const uint32_t mapMine = mapBase + (uint32_t)((FN_SLOTS_PER_LANE / 4) * quadLane) * 256u;  // this lane's first group of 4 slots
    // the 16-bit ids: every lane takes FN_SLOTS_PER_LANE consecutive slots of the particle from the staging area, 8 bytes per group of 4
#pragma unroll
    for (int v = 0; v < FN_SLOTS_PER_LANE / 8; v++) {
      const uint4 q = *reinterpret_cast<const uint4*>(idRow + FN_SLOTS_PER_LANE * quadLane + 8 * v);
      *reinterpret_cast<uint2*>(nbr16 + (size_t)(mapMine + (uint32_t)(2 * v) * 256u)) = make_uint2(q.x, q.y);
      *reinterpret_cast<uint2*>(nbr16 + (size_t)(mapMine + (uint32_t)(2 * v + 1) * 256u)) = make_uint2(q.z, q.w);
    }
    if (wide && quadLane == 0) {  // (after the entries: stores of one wave to one address keep their order)
      nbr16[mapBase] = (uint16_t)SPH_N16_WIDE;
      atomicAdd(&dbg[2], 1u);
    }
#pragma unroll
    for (int gq = 0; gq < FN_SLOTS_PER_LANE / 4; gq++) {
      const f32x4 dq = *reinterpret_cast<const f32x4*>(dstRow + FN_SLOTS_PER_LANE * quadLane + 4 * gq);
      const float dv[4] = {dq.x, dq.y, dq.z, dq.w};
      float out[4];
#pragma unroll
      for (int j = 0; j < 4; j++) out[j] = (FN_SLOTS_PER_LANE * quadLane + 4 * gq + j < run) ? sqrtf(dv[j]) * d.simScale : -1.f;
      *reinterpret_cast<f32x4*>(nbrDist + (size_t)(mapMine + (uint32_t)gq * 256u)) = f32x4{out[0], out[1], out[2], out[3]};
    }
  }
  FN_STAMP(8)
  }  // mine_now

  // ---- the rare particles the fast path cannot serve: the exact walk, by the wave that owns them (uniform control flow)
  {
    unsigned long long todo = __ballot(mine_now && slow && quadLane == 0);
    while (todo != 0ull) {
      const int b = __ffsll((long long)todo) - 1;
      todo &= todo - 1ull;
      fn_exact_walk(d, g, sh, p0 + wave * FN_PER_WAVE + (b >> FN_LOG_LANES), wave, lane);
    }
  }
  FN_STAMP(9)
  batchLo = batchHi;
  if (batchLo < blockHi) __syncthreads();  // (uniform) the next batch overwrites the staging area
  }  // batches
#ifdef FN_STAMPS
  __syncthreads();
  if (tid < 16) atomicAdd(&dbg[16 + tid], sh.stamps[tid] >> 6);
  if (tid == 0) {  // residency trace: when and where this workgroup ran
    trace[4 * blockIdx.x + 0] = (uint32_t)wgStart_;
    trace[4 * blockIdx.x + 1] = (uint32_t)__builtin_amdgcn_s_memrealtime();
    trace[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
    trace[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
  }
#endif
}

int sphk_find_neighbors(sph_solver* s, int ghostDepth) {
  {  // opt in to > 64 KB of dynamic LDS once per DEVICE (the attribute lives on the device's function object)
    static std::mutex mu;
    static bool attrSet[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    const int dev = s->cfg.device;
    if (dev < 0 || dev >= 64 || !attrSet[dev]) {
      SPH_HIP(hipFuncSetAttribute((const void*)k_find_neighbors, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(FnShared)));
      if (dev >= 0 && dev < 64) attrSet[dev] = true;
    }
  }
  const SphDev r = sph_ranged(s, ghostDepth);
  FnParams a;
  a.G = r.G; a.gx = r.gx; a.gy = r.gy; a.rangeLo = r.rangeLo; a.rangeHi = r.rangeHi;
  a.h = r.h; a.cellSize = r.cellSize; a.cellSizeInv = r.cellSizeInv; a.simScale = r.simScale;
  a.xmin = r.xmin; a.ymin = r.ymin; a.zmin = r.zmin;
  hipLaunchKernelGGL(k_find_neighbors, dim3(sph_blocks(s->d.N, FN_PART)), dim3(FN_THREADS), sizeof(FnShared), s->stream, a,
                     (const float4*)r.sortedPos, (const uint32_t*)r.keys, (const uint32_t*)r.cellStart, r.binU, r.nbrId, r.nbrDist, r.nbr16,
                     r.nbrBase, r.dbg,
                     r.valsAlt /* idle between the sort and the next step's sort */);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}
