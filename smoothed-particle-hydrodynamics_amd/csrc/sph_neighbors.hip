// K1 clearBuffers + K5 findNeighbors (sphFluid.cl:64-92, 94-184, 207-329) for gfx950.
//
// Semantics reproduced exactly: per particle, two passes over the same 8 cells in the order
// own, x, y, z, xy, xz, yz, xyz (towards the nearer half of the cell on each axis); pass 0 builds a 30-bin radial
// histogram of candidates with d^2 <= h^2, from which r_thr is chosen so that at most 32 remain (or 31h/30 if fewer
// than 32 exist); pass 1 appends candidates with d^2 <= r_thr^2 in traversal order, at most 32.
//
// MI355X design: the reference walks ~640 candidates twice per particle and does the expensive part
// (sqrt, IEEE divide, histogram / store) under a branch that ~7 % of the lanes take but ~99 % of the waves execute.
// Here a 256-thread workgroup owns 128 consecutive sorted particles (two lanes per particle), stages the <= 9
// contiguous runs of sorted particles that can contain their candidates (cells are contiguous along x in the sorted
// order) into LDS once as SoA x/y/z, and each lane
//   1. walks 4 of the particle's 8 cells ONCE, four candidates per trip (16-byte aligned LDS reads, packed-f32 math,
//      next quad prefetched), with the cheap filter d^2 <= max(h, 31h/30)^2 — a superset of both reference passes — and
//      appends the LDS slot of every hit to a private u16 list in LDS (compaction: ~45 of ~640 candidates survive),
//   2. replays pass 0 / threshold / pass 1 of the reference over that short list in registers, in traversal order,
//      with exactly the reference's float expressions, so r_thr, slot order and distances are bit-identical. Pass 0 does
//      not build the histogram: the cumulative counts it needs are "d^2 < U[j]" tests against 30 thresholds computed
//      exactly on the host (SphDev::binU), searched by bisection.
// Particles whose list overflows or whose cells are not fully staged (wrapped / aliased cells, LDS capacity) are queued
// and served by k_find_neighbors_fallback, the literal two-pass walk over global memory (one wave per particle), so the
// result is exact for any input.
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
  const size_t n4 = (size_t)((s->d.N + SPH_TILE - 1) / SPH_TILE) * 64 * 8;
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

#ifndef FN_PART
#define FN_PART 128         // particles per workgroup (A/B on config #2: 128 -> 0.89 ms, 256 -> 0.96 ms)
#endif
#define FN_THREADS (2 * FN_PART)  // two lanes per particle
#ifndef FN_CAND_CAP
#define FN_CAND_CAP 4096    // staged candidates per workgroup (SoA x/y/z: 48 KB; + lists 24 KB -> two workgroups per CU)
#endif
#define FN_WIN 16           // cell-table window per row: covers batches that span up to 13 cells (else: direct table reads)
#define FN_CAND_PAD 12      // the aligned, prefetching 4-wide walk reads (never uses) up to 11 slots past a cell
#ifndef FN_LIST_CAP
#define FN_LIST_CAP 48      // compaction list entries per lane (u16 [entry][lane])
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FnShared {
  float x[FN_CAND_CAP + FN_CAND_PAD], y[FN_CAND_CAP + FN_CAND_PAD], z[FN_CAND_CAP + FN_CAND_PAD];  // staged candidates
  uint16_t list[FN_LIST_CAP][FN_THREADS];         // [entry][lane] LDS slots of the filter hits, traversal order
  float binU[32];                                 // U[j]: d^2 < U[j]  <=>  counted in histogram bins 0..j (see SphDev::binU)
  int rowLo[9], rowHi[9], rowBase[9];             // staged runs: sorted-index range and first LDS slot
  int win[9][FN_WIN];                             // cellStart[] of the cells cLo-1 .. of every row (see the staging code)
  int segEnd[9], segDelta[9];                     // flat staging: LDS slots < segEnd[q] belong to run q; sorted index = slot + segDelta[q]
  int total, winOk;                               // staged candidates; 1 if the window covers the batch's cells
  int batchLo, batchHi, retry;                    // current batch of particles; retry: small-batch mode
};

// d.dbg layout: [0] particles handed to the fallback kernel because a cell was not staged, [1] because a list
// overflowed, [3] candidate runs dropped for LDS capacity, [4] fallback queue length.
//
// Lane pair (2p, 2p+1) serves particle p: lane 0 walks the cells k = 0,5,6,7 of the reference's order and lane 1 the
// cells 1,2,3,4, so the merged traversal order is A0 B1 B2 B3 B4 A5 A6 A7 and only four per-cell hit counts have to
// cross lanes (one DPP swap each) to place every neighbour in the reference's slot.
__global__ __launch_bounds__(FN_THREADS, 2) void k_find_neighbors(SphDev d, uint32_t* __restrict__ slowQueue) {
  extern __shared__ __align__(16) unsigned char fn_smem[];
  FnShared& sh = *reinterpret_cast<FnShared*>(fn_smem);
  const int tid = threadIdx.x;
  const int p = tid >> 1, half = tid & 1;
  const int rangeBegin = (int)d.cellStart[d.rangeLo], rangeEnd = (int)d.cellStart[d.rangeHi];  // all particles, or fewer ghost layers
  const int p0 = rangeBegin + blockIdx.x * FN_PART;
  if (p0 >= rangeEnd) return;  // uniform
  const int id = p0 + p;
  const bool alive = id < rangeEnd;

  // ---- stage the candidate runs. Cells of a batch of particles lie in [cLo, cHi]; row r = (sy+1) + 3*(sz+1) holds the
  // cells [cLo - 1, cHi + 1] shifted by sy*gx + sz*gx*gy, which is one contiguous run of sorted particles. Normally the
  // batch is the whole workgroup (128 particles). Where those runs would not fit the LDS budget — the workgroup straddles
  // two x-rows of cells (the runs then span whole rows) or sits in very dense cells — it serves its particles in smaller
  // batches instead: at most 32 particles, never across an x-row boundary.
  const int blockHi = min(p0 + FN_PART, rangeEnd);
  // own record and cell: issued now, needed only after the staging
  const float4 myPos = alive ? d.sortedPos[id] : make_float4(0.f, 0.f, 0.f, 0.f);
  const int myCell = alive ? (int)d.keys[id] : 0;
  if (tid == 0) { sh.batchLo = p0; sh.retry = 0; }
  while (true) {
  __syncthreads();  // (also protects the LDS of the previous batch)
  const int batchLo = sh.batchLo;
  if (batchLo >= blockHi) break;  // uniform
  if (tid < 64) {
    int len = blockHi - batchLo;
    if (sh.retry) {
      const int i = batchLo + tid;
      const unsigned row0 = d.keys[batchLo] / (unsigned)d.gx;
      const bool stop = (tid >= 32) || (i >= blockHi) || (d.keys[min(i, blockHi - 1)] / (unsigned)d.gx != row0);
      len = __ffsll((long long)__ballot(stop)) - 1;  // first lane that must not join the batch (lane 32 at the latest)
    }
    if (tid == 0) sh.batchHi = batchLo + len;
  }
  __syncthreads();
  const int batchHi = sh.batchHi;
  // Every global round trip below is on the workgroup's critical path (two workgroups per CU cannot hide them), so the
  // staging is three trips deep: (1) the batch's first / last cell, (2) the cell table of the 9 rows into LDS, (3) all
  // candidate records at once, flat over the concatenated runs.
  const int cLo = (int)d.keys[batchLo], cHi = (int)d.keys[batchHi - 1];
  const int nc = cHi - cLo + 3;  // cells per row: cLo-1 .. cHi+1; the window holds nc+1 table entries
  const bool winOk = nc + 1 <= FN_WIN;
  if (tid < 9 * FN_WIN) {
    const int r = tid / FN_WIN, i = tid % FN_WIN;
    const int sy = r % 3 - 1, sz = r / 3 - 1;
    const int shift = sy * d.gx + sz * d.gx * d.gy;
    if (winOk) {
      if (i <= nc) sh.win[r][i] = (int)d.cellStart[min(max(cLo - 1 + shift + i, 0), d.G)];
    } else if (i < 2) {  // sparse cells: only the two ends of the run (particle_cells reads the table directly)
      const int a = max(cLo - 1 + shift, 0), b = min(cHi + 1 + shift, d.G - 1);
      sh.win[r][i] = (a <= b) ? (int)d.cellStart[i == 0 ? a : b + 1] : 0;
    }
  }
  if (tid >= 192 && tid < 224) sh.binU[tid - 192] = d.binU[tid - 192];
  __syncthreads();
  if (tid == 0) {
    int total = 0;
    for (int r = 0; r < 9; r++) {
      const int lo = sh.win[r][0], hi = winOk ? sh.win[r][nc] : sh.win[r][1];
      sh.rowLo[r] = lo; sh.rowHi[r] = max(hi, lo);  // (clamped table reads: an out-of-grid run is empty)
      total += sh.rowHi[r] - lo;
    }
    sh.winOk = winOk ? 1 : 0;
    if (total > FN_CAND_CAP && !sh.retry) sh.retry = 2;  // 2 = "switch now": redo this workgroup in small batches
    else {
      int base = 0;
      // the own row (4) first, then the others: if LDS still runs out, the most-used runs are the ones that are staged
      const int order[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8};
      for (int q = 0; q < 9; q++) {
        const int r = order[q];
        int n = sh.rowHi[r] - sh.rowLo[r];
        sh.rowBase[r] = base;
        if (base + n > FN_CAND_CAP) { sh.rowHi[r] = sh.rowLo[r]; n = 0; atomicAdd(&d.dbg[3], 1u); }  // not staged (rare)
        sh.segDelta[q] = sh.rowLo[r] - base;
        base += n;
        sh.segEnd[q] = base;
      }
      sh.total = base;
    }
  }
  __syncthreads();
  if (sh.retry == 2) {  // uniform
    __syncthreads();
    if (tid == 0) sh.retry = 1;
    continue;
  }
  {
    const int total = sh.total;
    int se[8], sd[9];
#pragma unroll
    for (int q = 0; q < 8; q++) se[q] = sh.segEnd[q];
#pragma unroll
    for (int q = 0; q < 9; q++) sd[q] = sh.segDelta[q];
    constexpr int PER = 8;  // records per thread and round, all loads of a round in flight together
#pragma unroll 1
    for (int f0 = 0; f0 < total; f0 += PER * FN_THREADS) {
      float4 rec[PER];
#pragma unroll
      for (int u = 0; u < PER; u++) {
        const int f = f0 + tid + u * FN_THREADS;
        int delta = sd[0];
#pragma unroll
        for (int q = 0; q < 8; q++) delta = (f >= se[q]) ? sd[q + 1] : delta;  // runs are laid out in order of q
        if (f < total) rec[u] = d.sortedPos[f + delta];
      }
#pragma unroll
      for (int u = 0; u < PER; u++) {
        const int f = f0 + tid + u * FN_THREADS;
        if (f < total) { sh.x[f] = rec[u].x; sh.y[f] = rec[u].y; sh.z[f] = rec[u].z; }
      }
    }
  }
  __syncthreads();
  if (tid == 0) sh.batchLo = batchHi;  // next batch (every thread has read batchLo / batchHi by now)
  const bool mine_now = alive && id >= batchLo && id < batchHi;  // this lane's particle belongs to the current batch
  if (!mine_now) continue;

  float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
  bool slow = false;
  int ldsLo[4], num[4], absDelta[4];  // absDelta: sorted index = LDS slot + absDelta
  int selfSlot = -1;
#pragma unroll
  for (int i = 0; i < 4; i++) { num[i] = 0; ldsLo[i] = 0; absDelta[i] = 0; }
  if (alive) {
    me = myPos;
    CellSet cs;
    {  // particle_cells() with the cell table read from the LDS window where possible
      const float px = me.x - d.xmin, py = me.y - d.ymin, pz = me.z - d.zmin;
      const float cfx = (float)(int)(me.x * d.cellSizeInv) * d.cellSize;
      const float cfy = (float)(int)(me.y * d.cellSizeInv) * d.cellSize;
      const float cfz = (float)(int)(me.z * d.cellSizeInv) * d.cellSize;
      const int dx = ((px - cfx) < d.h) ? -1 : 1;
      const int dy = ((py - cfy) < d.h) ? -1 : 1;
      const int dz = ((pz - cfz) < d.h) ? -1 : 1;
      const int sy = dy * d.gx, sz = dz * d.gx * d.gy;
      const int off[8] = {0, dx, sy, sz, dx + sy, dx + sz, sy + sz, dx + sy + sz};
      const int xs[8] = {0, dx, 0, 0, dx, dx, 0, dx};  // the x part of off[]: position inside the row's window
      const int ry = dy + 1, rz = dz + 1;
      const int rows[8] = {4, 4, ry + 3, 1 + 3 * rz, ry + 3, 1 + 3 * rz, ry + 3 * rz, ry + 3 * rz};
      const bool useWin = sh.winOk != 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int raw = myCell + off[k];
        const int c = wrap_cell(raw, d.G);
        if (c == raw && useWin) {
          const int i = myCell - cLo + 1 + xs[k];  // in [0, nc): the batch's cells are cLo..cHi, the window starts at cLo-1
          cs.row[k] = rows[k];
          cs.lo[k] = sh.win[rows[k]][i];
          cs.hi[k] = sh.win[rows[k]][i + 1];
        } else {  // wrapped cells (never staged: only their emptiness matters) and batches wider than the window
          const int cc = min(max(c, 0), d.G - 1);
          cs.row[k] = (c == raw) ? rows[k] : -1;
          cs.lo[k] = (int)d.cellStart[cc];
          cs.hi[k] = (int)d.cellStart[cc + 1];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int n = cs.hi[k] - cs.lo[k];
      int base = 0;
      if (n > 0) {
        const int r = cs.row[k];
        if (r >= 0 && cs.lo[k] >= sh.rowLo[r] && cs.hi[k] <= sh.rowHi[r]) base = sh.rowBase[r] + (cs.lo[k] - sh.rowLo[r]);
        else slow = true;  // a non-empty cell of this particle is not in LDS (both lanes of the pair see this)
      }
      // lane 0 of the pair walks cells {0,5,6,7}, lane 1 cells {1,2,3,4}: both get four cells and about half of the hits
      // (own cell ~42 %, each face neighbour ~14 %, edge ~5 %, corner ~2 %), so neither list is much longer than the other
      const int owner = (k == 0 || k >= 5) ? 0 : 1, slot = (k == 0) ? 0 : (k >= 5 ? k - 4 : k - 1);
      if (owner == half) { num[slot] = n; ldsLo[slot] = base; absDelta[slot] = cs.lo[k] - base; }
      if (k == 0 && half == 0) selfSlot = base + (id - cs.lo[0]);  // the particle itself sits in its own cell
    }
  }
  if (slow) {
#pragma unroll
    for (int i = 0; i < 4; i++) num[i] = 0;
    if (half == 0) atomicAdd(&d.dbg[0], 1u);
  }

  // ---- 1. single walk with the cheap filter; r_max covers pass 0 (h) and every possible pass-1 radius (<= 31h/30).
  // Four candidates per trip from 16-byte-aligned SoA reads (the walk starts at the aligned slot at or before the cell and
  // masks what lies outside [0, n)); the arithmetic is packed f32 (v_pk_*), IEEE per component, same order as the reference.
  const float rA = d.h, rB = (float)(SPH_RSEG + 1) * d.h / (float)SPH_RSEG;
  const float r2max = fmaxf(rA * rA, rB * rB);
  int cnt = 0;
  int segEnd[4];
  const f32x2 mx = {me.x, me.x}, my = {me.y, me.y}, mz = {me.z, me.z};
  // The filter only has to be a superset of both reference passes (the replay below decides with the reference's exact
  // expressions), so it may use fused multiply-adds: its d^2 differs from the reference's by < 2^-21 relative, which the
  // 2^-20 margin on the radius absorbs. A hit is the SIGN BIT of d^2 - r^2 (two distinct floats never subtract to zero with
  // denormals on), shifted into a per-lane bit mask with one v_alignbit per candidate; every 32 candidates, and at the end
  // of a cell, the few set bits are turned into list entries in traversal order.
  const float r2f = r2max * (1.f + 0x1p-20f);
  const f32x2 thr = {r2f, r2f};
#define FN_LOAD(a, X, Y, Z)                                                                     \
  const f32x4 X = *reinterpret_cast<const f32x4*>(&sh.x[a]);                                    \
  const f32x4 Y = *reinterpret_cast<const f32x4*>(&sh.y[a]);                                    \
  const f32x4 Z = *reinterpret_cast<const f32x4*>(&sh.z[a]);
#define FN_TEST(X, Y, Z)                                                                        \
  {                                                                                             \
    const f32x2 ex0 = mx - X.xy, ex1 = mx - X.zw, ey0 = my - Y.xy, ey1 = my - Y.zw;             \
    const f32x2 ez0 = mz - Z.xy, ez1 = mz - Z.zw;                                               \
    const f32x2 q0 = __builtin_elementwise_fma(ez0, ez0, __builtin_elementwise_fma(ey0, ey0, ex0 * ex0)); \
    const f32x2 q1 = __builtin_elementwise_fma(ez1, ez1, __builtin_elementwise_fma(ey1, ey1, ex1 * ex1)); \
    const f32x2 s0 = q0 - thr, s1 = q1 - thr;                                                   \
    acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(s0.x), 31);                            \
    acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(s0.y), 31);                            \
    acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(s1.x), 31);                            \
    acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(s1.y), 31);                            \
  }
  // k candidates (slots aBase .. aBase+k-1, first one in the highest of the k low bits of acc) -> list entries
#define FN_FLUSH(k)                                                                             \
  {                                                                                             \
    uint32_t m = ((k) == 0) ? 0u : (acc << (32 - (k)));   /* candidate t at bit 31 - t */       \
    const int lowT = cellLo - aBase, hiT = cellHi - aBase;                                      \
    if (lowT > 0) m &= 0xffffffffu >> lowT;               /* slots before the cell (first chunk) */ \
    if (hiT < 32) m &= ~(0xffffffffu >> hiT);             /* slots past the cell */             \
    if (i == 0) {                                         /* the particle itself */             \
      const unsigned ts = (unsigned)(selfSlot - aBase);                                         \
      if (ts < 32u) m &= ~(0x80000000u >> ts);                                                  \
    }                                                                                           \
    while (m != 0u) {                                                                           \
      const int bpos = __clz((int)m);                                                           \
      sh.list[min(cnt, FN_LIST_CAP - 1)][tid] = (uint16_t)(aBase + bpos);                       \
      cnt++;                                                                                    \
      m &= ~(0x80000000u >> bpos);                                                              \
    }                                                                                           \
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int n = num[i], cellLo = ldsLo[i], cellHi = cellLo + n;
    if (n > 0) {
      int a = cellLo & ~3;  // aligned slot at or before the cell
      int aBase = a, k = 0;
      uint32_t acc = 0u;
      f32x4 X = *reinterpret_cast<const f32x4*>(&sh.x[a]);
      f32x4 Y = *reinterpret_cast<const f32x4*>(&sh.y[a]);
      f32x4 Z = *reinterpret_cast<const f32x4*>(&sh.z[a]);
      while (a < cellHi) {
        // two quads per trip, ping-pong: the next quad's three LDS reads are issued before the current quad is tested
        // (the wave has only one partner on its SIMD to hide LDS latency). Reads and tests may run up to 11 slots past
        // the cell (FN_CAND_PAD); FN_FLUSH masks those bits.
        FN_LOAD(a + 4, Xb, Yb, Zb)
        FN_TEST(X, Y, Z)
        X = *reinterpret_cast<const f32x4*>(&sh.x[a + 8]);
        Y = *reinterpret_cast<const f32x4*>(&sh.y[a + 8]);
        Z = *reinterpret_cast<const f32x4*>(&sh.z[a + 8]);
        FN_TEST(Xb, Yb, Zb)
        a += 8; k += 8;
        if (k == 32) { FN_FLUSH(32) aBase = a; k = 0; }
      }
      FN_FLUSH(k)
    }
    segEnd[i] = cnt;
  }
#undef FN_LOAD
#undef FN_TEST
#undef FN_FLUSH
  bool over = cnt > FN_LIST_CAP;
  const int partnerOver = __shfl_xor((int)over, 1);  // unconditional: both lanes of the pair must take part in the swap
  over = over || (partnerOver != 0);
  if (over) {
    if (half == 0 && !slow) atomicAdd(&d.dbg[1], 1u);
    slow = true;
#pragma unroll
    for (int i = 0; i < 4; i++) segEnd[i] = 0;
  }

  // ---- 2. replay of the reference's two passes over the short lists, entirely in registers: the list (<= 48 LDS slots
  // per lane) is expanded once into d2v[] with static indices (the slots are re-read from LDS for the few hits that are stored), 8 entries at a time with a wave-uniform skip.
  const int total = segEnd[3];
  float d2v[FN_LIST_CAP];
#pragma unroll
  for (int c0 = 0; c0 < FN_LIST_CAP; c0 += 8) {
    if (__any(c0 < total)) {
      int slotv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) slotv[u] = min((int)sh.list[c0 + u][tid], FN_CAND_CAP + FN_CAND_PAD - 1);
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int sl = slotv[u];
        const float ex = me.x - sh.x[sl], ey = me.y - sh.y[sl], ez = me.z - sh.z[sl];
        const float v = ex * ex + ey * ey + ez * ez;
        d2v[c0 + u] = (c0 + u < total) ? v : __builtin_inff();  // +inf never passes a `<` / `<=` test below
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; u++) d2v[c0 + u] = __builtin_inff();
    }
  }
  // ---- 2a. pass 0 + threshold (sphFluid.cl:157-161,310-323) without building the histogram: C(j) = number of hits in
  // bins 0..j = number of hits with d^2 < U[j] (U precomputed exactly on the host). The reference's loop stops at
  // j* = min{ j : C(j) >= 32 } with jb = j* if C(j*) == 32, j* - 1 if it overshoots, and jb = 30 if no such j exists.
  int liveEnd = 8;  // wave-uniform: entries at or past it are +inf in every lane, and the counting loops skip them
#pragma unroll
  for (int c0 = 8; c0 < FN_LIST_CAP; c0 += 8) liveEnd = __any(c0 < total) ? c0 + 8 : liveEnd;
  int lo = 0, hi = SPH_RSEG, cAtHi = 0;
#pragma unroll 1
  for (int it = 0; it < 5; it++) {
    const int mid = (lo + hi) >> 1;  // lo == hi on converged lanes: harmless re-evaluation
    const float U = sh.binU[min(mid, SPH_RSEG - 1)];
    int c = 0;
#pragma unroll
    for (int c0 = 0; c0 < FN_LIST_CAP; c0 += 8) {
      if (c0 < liveEnd) {
#pragma unroll
        for (int u = 0; u < 8; u++) c += (d2v[c0 + u] < U) ? 1 : 0;
      }
    }
    c += __shfl_xor(c, 1);
    if (lo < hi) {
      if (c >= SPH_MAXN) { hi = mid; cAtHi = c; } else lo = mid + 1;
    }
  }
  const int jb = (lo >= SPH_RSEG) ? SPH_RSEG : ((cAtHi == SPH_MAXN) ? lo : lo - 1);
  const float r_thr = (float)(jb + 1) * d.h / (float)SPH_RSEG;
  const float r2 = r_thr * r_thr;

  // ---- 2b. pass 1: hits with d^2 <= r_thr^2 as a bit mask; per-cell counts by popcount; swap the four counts inside the
  // pair; every hit goes to (hits in earlier cells of the merged order) + (rank inside its cell). Slots >= 32 are
  // dropped, which is what the reference's `break` / `spaceLeft` logic amounts to (sphFluid.cl:145,168-169).
  unsigned long long acc = 0ull;
#pragma unroll
  for (int c0 = 0; c0 < FN_LIST_CAP; c0 += 8) {
    if (c0 < liveEnd) {
#pragma unroll
      for (int u = 0; u < 8; u++) acc |= (d2v[c0 + u] <= r2) ? (1ull << (c0 + u)) : 0ull;
    }
  }
  unsigned long long below[4];  // bits of the entries before the end of cell i
#pragma unroll
  for (int i = 0; i < 4; i++) below[i] = (segEnd[i] >= 64) ? ~0ull : ((1ull << segEnd[i]) - 1ull);
  int mine[4], theirs[4];
  mine[0] = __popcll(acc & below[0]);
#pragma unroll
  for (int i = 1; i < 4; i++) mine[i] = __popcll(acc & below[i] & ~below[i - 1]);
#pragma unroll
  for (int i = 0; i < 4; i++) theirs[i] = __shfl_xor(mine[i], 1);
  if (alive && !slow) {
    // merged (reference) order of the 8 cells: A0 | B1 B2 B3 B4 | A5 A6 A7, where A = lane 0 and B = lane 1 of the pair
    int start[4];
    const int sumMine = mine[0] + mine[1] + mine[2] + mine[3], sumTheirs = theirs[0] + theirs[1] + theirs[2] + theirs[3];
    const int run = sumMine + sumTheirs;
    if (half == 0) { start[0] = 0; start[1] = mine[0] + sumTheirs; }
    else { start[0] = theirs[0]; start[1] = start[0] + mine[0]; }
    start[2] = start[1] + mine[1];
    start[3] = start[2] + mine[2];
    // walk the entries in list order with a running (cell start, index delta, rank inside the cell)
    int curStart = start[0], curDelta = absDelta[0], rank = 0;
    int32_t* const idBase = d.nbrId + (((size_t)(id >> 6) * 8 * 64 + (size_t)(id & 63)) << 2);
    float* const distBase = d.nbrDist + (((size_t)(id >> 6) * 8 * 64 + (size_t)(id & 63)) << 2);
#pragma unroll
    for (int c0 = 0; c0 < FN_LIST_CAP; c0 += 8) {
      if (!__any(c0 < total)) continue;  // wave-uniform skip of empty chunks
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int e = c0 + u;
        if (e == segEnd[0]) { curStart = start[1]; curDelta = absDelta[1]; rank = 0; }  // (empty cells cascade in order)
        if (e == segEnd[1]) { curStart = start[2]; curDelta = absDelta[2]; rank = 0; }
        if (e == segEnd[2]) { curStart = start[3]; curDelta = absDelta[3]; rank = 0; }
        if ((acc >> e) & 1ull) {
          const int pos = curStart + rank;
          rank++;
          if (pos < SPH_MAXN) {
            const int off = ((pos >> 2) << 8) + (pos & 3);  // tiled map: group stride 64 lanes * 4 slots
            idBase[off] = (int)sh.list[e][tid] + curDelta;
            distBase[off] = sqrtf(d2v[e]) * d.simScale;
          }
        }
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
  }  // batches
}

// The queued particles, one WAVE per particle: the 64 lanes take consecutive candidates of a cell, so a particle costs
// ~2 x 8 x 2 dependent memory round trips instead of ~1300. The reference's semantics, literally: pass 0 fills a
// 30-bin histogram (LDS atomics), every lane derives the same r_thr, pass 1 assigns slots in traversal order with a
// ballot prefix (lane order == candidate order inside a chunk) and stops at 32.
__global__ __launch_bounds__(SPH_BLOCK) void k_find_neighbors_fallback(SphDev d, const uint32_t* __restrict__ slowQueue) {
  __shared__ uint32_t hist[SPH_BLOCK / 64][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t count = d.dbg[4];
  const unsigned long long ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (uint32_t q = blockIdx.x * (SPH_BLOCK / 64) + wave; q < count; q += gridDim.x * (SPH_BLOCK / 64)) {
    const int id = (int)slowQueue[q];
    const float4 me = d.sortedPos[id];
    CellSet cs;
    particle_cells(d, me, (int)d.keys[id], cs);
    if (lane < 32) hist[wave][lane] = 0u;
    const float h2 = d.h * d.h;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      for (int j0 = cs.lo[k]; j0 < cs.hi[k]; j0 += 64) {
        const int j = j0 + lane;
        if (j < cs.hi[k] && j != id) {
          const float4 o = d.sortedPos[j];
          const float ex = me.x - o.x, ey = me.y - o.y, ez = me.z - o.z;
          const float d2 = ex * ex + ey * ey + ez * ez;
          if (d2 <= h2) {
            const float dist = sqrtf(d2);
            const int bin = (int)(dist * (float)SPH_RSEG / d.h);
            if (bin < SPH_RSEG) atomicAdd(&hist[wave][bin], 1u);
          }
        }
      }
    }
    int jb = 0, sum = 0;  // threshold, sphFluid.cl:310-323 (LDS ops of one wave retire in order: the adds are visible)
    while (jb < SPH_RSEG) {
      sum += (int)hist[wave][jb];
      if (sum == SPH_MAXN) break;
      if (sum > SPH_MAXN) { jb--; break; }
      jb++;
    }
    const float r_thr = (float)(jb + 1) * d.h / (float)SPH_RSEG;
    const float r2 = r_thr * r_thr;
    int found = 0;  // wave-uniform
#pragma unroll
    for (int k = 0; k < 8; k++) {
      for (int j0 = cs.lo[k]; j0 < cs.hi[k] && found < SPH_MAXN; j0 += 64) {
        const int j = j0 + lane;
        float d2 = 0.f;
        bool hit = false;
        if (j < cs.hi[k] && j != id) {
          const float4 o = d.sortedPos[j];
          const float ex = me.x - o.x, ey = me.y - o.y, ez = me.z - o.z;
          d2 = ex * ex + ey * ey + ez * ez;
          hit = d2 <= r2;
        }
        const unsigned long long m = __ballot(hit);
        const int pos = found + __popcll(m & ltMask);
        if (hit && pos < SPH_MAXN) {
          const size_t idx = nbr_index(id, pos);
          d.nbrId[idx] = j;
          d.nbrDist[idx] = sqrtf(d2) * d.simScale;
        }
        found += __popcll(m);
      }
    }
    if (lane >= min(found, SPH_MAXN) && lane < SPH_MAXN) {
      const size_t idx = nbr_index(id, lane);
      d.nbrId[idx] = -1;
      d.nbrDist[idx] = -1.f;
    }
  }
}

int sphk_find_neighbors(sph_solver* s, int ghostDepth) {
  static bool attrSet = false;
  if (!attrSet) {
    SPH_HIP(hipFuncSetAttribute((const void*)k_find_neighbors, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(FnShared)));
    attrSet = true;
  }
  // the fallback queue reuses keysAlt (N words, idle between the sort and the next step's sort)
  SPH_HIP(hipMemsetAsync(&s->d.dbg[4], 0, sizeof(uint32_t), s->stream));
  hipLaunchKernelGGL(k_find_neighbors, dim3(sph_blocks(s->d.N, FN_PART)), dim3(FN_THREADS), sizeof(FnShared), s->stream, sph_ranged(s, ghostDepth), s->d.keysAlt);
  hipLaunchKernelGGL(k_find_neighbors_fallback, dim3(min(sph_blocks(s->d.N, 4), 1024)), dim3(SPH_BLOCK), 0, s->stream, s->d, s->d.keysAlt);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}
