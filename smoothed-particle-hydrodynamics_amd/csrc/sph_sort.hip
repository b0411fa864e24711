// Cell binning on the device: hash (K2), stable LSD radix sort by cell id (replaces the reference's host qsort,
// owOpenCLSolver.cpp:255-261), gather into sorted order (K3) and the cell-start table (K4 + the host fix-up loop,
// owOpenCLSolver.cpp:305-319). No device->host round trip, unlike the reference.
#include "sph_common.h"

// ------------------------------------------------------------------ K2 hashParticles (sphFluid.cl:187-201,332-383)
__global__ __launch_bounds__(SPH_BLOCK) void k_hash(SphDev d) {
  const int id = blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (id >= d.N) return;
  const float4 p = d.posOrig[id];
  // A non-finite coordinate means the simulation has blown up; such particles all hash into one cell and the search turns
  // quadratic. Counted here, reported by the next blocking call (sph_synchronize / sph_read_*) as SPH_ERR_INVALID.
  if (!(fabsf(p.x) <= 3.0e38f && fabsf(p.y) <= 3.0e38f && fabsf(p.z) <= 3.0e38f)) atomicAdd(&d.dbg[6], 1u);
  // cellFactors ignores xmin/ymin/zmin (sphFluid.cl:197-199)
  const int cx = (int)(p.x * d.cellSizeInv), cy = (int)(p.y * d.cellSizeInv), cz = (int)(p.z * d.cellSizeInv);
  const int cell = cx + cy * d.gx + cz * d.gx * d.gy;
  d.keys[id] = (uint32_t)cell & d.cellMask;  // `& 0xffff` at sphFluid.cl:377 in reference mode
  d.vals[id] = (uint32_t)id;
}

// Slab mode (fused step): the sort key is the cell id COMPACTED to the cells a slab can hold — the declared grid is sized by h
// but hashed by 2h, so only ~1/8 of its index space is reachable, and a slab spans few z layers. The
// mapping (cx, cy, cz) -> cx + usedX*(cy + usedY*(cz - czBase)) is monotone in the real cell id, so the stable sort gives the
// same order with 2 radix passes instead of 3; k_sort_post_rekey puts the real cell id back into keys[].
struct CompactKey { int usedX, usedY, czBase, layers; };

__global__ __launch_bounds__(SPH_BLOCK) void k_hash_compact(SphDev d, CompactKey c) {
  const int id = blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (id >= d.N) return;
  const float4 p = d.posOrig[id];
  if (!(fabsf(p.x) <= 3.0e38f && fabsf(p.y) <= 3.0e38f && fabsf(p.z) <= 3.0e38f)) atomicAdd(&d.dbg[6], 1u);  // (see k_hash)
  const int cx = min(max((int)(p.x * d.cellSizeInv), 0), c.usedX - 1), cy = min(max((int)(p.y * d.cellSizeInv), 0), c.usedY - 1);
  const int cz = min(max((int)(p.z * d.cellSizeInv) - c.czBase, 0), c.layers - 1);
  d.keys[id] = (uint32_t)(cx + c.usedX * (cy + c.usedY * cz));
  d.vals[id] = (uint32_t)id;
}

__global__ __launch_bounds__(SPH_BLOCK) void k_sort_post_rekey(SphDev d) {
  const int i = blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (i >= d.N) return;
  const uint32_t src = d.vals[i];
  const float4 p = d.posOrig[src];
  d.sortedPos[i] = p;
  d.sortedVel[i] = d.velOrig[src];
  d.backIndex[src] = (uint32_t)i;
  const int cx = (int)(p.x * d.cellSizeInv), cy = (int)(p.y * d.cellSizeInv), cz = (int)(p.z * d.cellSizeInv);
  d.keys[i] = (uint32_t)(cx + cy * d.gx + cz * d.gx * d.gy) & d.cellMask;  // the real cell id, as k_hash computes it
}

static int bits_for(long long n) { int b = 1; while ((1LL << b) < n) b++; return b; }

// The compaction of the current domain (the slab's layers in slab mode, the whole box otherwise) and the bits its keys need.
static CompactKey compact_plan(const sph_solver* s, int* bits) {
  const SphDev& d = s->d;
  CompactKey c;
  c.usedX = min((int)(d.xmax * d.cellSizeInv) + 1, d.gx);
  c.usedY = min((int)(d.ymax * d.cellSizeInv) + 1, d.gy);
  long long lo = 0, hi = min((long long)(d.zmax * d.cellSizeInv) + 1, (long long)d.gz);
  if (s->hasSlab) {
    lo = max((long long)s->slab.layerLo - s->slab.ghostLayers - 2, 0LL);
    hi = min((long long)s->slab.layerHi + s->slab.ghostLayers + 2, (long long)d.gz);
  }
  c.czBase = (int)lo; c.layers = (int)max(hi - lo, 1LL);
  *bits = bits_for((long long)c.usedX * c.usedY * c.layers);
  return c;
}

// K2 of the fused step. Wide cell ids only (the reference's 16-bit ids alias, so no monotone compaction exists), and only
// when the compacted keys take fewer radix passes than the real ones.
int sphk_step_sort_bits(const sph_solver* s, bool* compact) {
  int bits = s->sortBits;
  const CompactKey c = compact_plan(s, &bits);
  // The compacted key is monotone in the real cell id only while k_hash_compact's clamps never bite: every coordinate a particle
  // of the box can have must hash to cx, cy, cz inside [0, used). A box that starts below zero (negative coordinates truncate
  // towards zero and alias in the real ids) or whose `used` was capped by the declared grid keeps the real keys.
  const SphDev& d = s->d;
  const bool clampFree = d.xmin >= 0.f && d.ymin >= 0.f && d.zmin >= 0.f && (int)(d.xmax * d.cellSizeInv) < c.usedX &&
                         (int)(d.ymax * d.cellSizeInv) < c.usedY && (s->hasSlab || (int)(d.zmax * d.cellSizeInv) < c.czBase + c.layers);
  *compact = clampFree && s->d.cellMask == 0xffffffffu && sph_sort_passes(bits) < sph_sort_passes(s->sortBits);
  return *compact ? bits : s->sortBits;
}

int sphk_hash_for_step(sph_solver* s, int* sortBits, bool* compact) {
  *sortBits = sphk_step_sort_bits(s, compact);
  if (!*compact) return sphk_hash(s);
  int bits;
  const CompactKey c = compact_plan(s, &bits);
  hipLaunchKernelGGL(k_hash_compact, dim3(sph_blocks(s->d.N)), dim3(SPH_BLOCK), 0, s->stream, s->d, c);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

int sphk_sort_post_rekey(sph_solver* s) {
  hipLaunchKernelGGL(k_sort_post_rekey, dim3(sph_blocks(s->d.N)), dim3(SPH_BLOCK), 0, s->stream, s->d);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

// hash + sort + gather of the fused step in slab mode
int sphk_hash_sort_post_slab(sph_solver* s) {
  int bits;
  bool compact;
  int rc = sphk_hash_for_step(s, &bits, &compact);
  if (rc == SPH_OK) rc = sphk_sort_pairs(s, s->d.N, bits);
  if (rc != SPH_OK) return rc;
  return compact ? sphk_sort_post_rekey(s) : sphk_sort_post(s);
}

int sphk_hash(sph_solver* s) {
  hipLaunchKernelGGL(k_hash, dim3(sph_blocks(s->d.N)), dim3(SPH_BLOCK), 0, s->stream, s->d);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

// ------------------------------------------------------------------ stable LSD radix sort, 8 or 9 bits per pass
// Tile = 256 threads x 16 keys. Wave w of a block owns the contiguous 1024 keys [w*1024, (w+1)*1024) of the tile and
// walks them in 16 rounds of 64, so "earlier in memory" == (block, wave, round, lane) order and ranks stay stable:
// myCompare orders by cell only and glibc's qsort keeps ties in input (= ascending orig id) order.
// BITS = 9 (512 digits) is used when it saves a pass: 17-18 key bits take 2 passes instead of 3 (the compacted cell ids of
// the 16.5 M box need 18), 25-27 bits 3 instead of 4.
#define RS_ITEMS 16
#define RS_TILE (SPH_BLOCK * RS_ITEMS)

template <int BITS>
__global__ __launch_bounds__(SPH_BLOCK) void k_radix_hist(const uint32_t* __restrict__ keys, int N, int shift,
                                                           uint32_t* __restrict__ blockHist, int numBlocks) {
  constexpr int D = 1 << BITS;
  __shared__ uint32_t hist[D];
  for (int i = threadIdx.x; i < D; i += SPH_BLOCK) hist[i] = 0;
  __syncthreads();
  const int base = blockIdx.x * RS_TILE;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const int i = base + r * SPH_BLOCK + threadIdx.x;
    if (i < N) atomicAdd(&hist[(keys[i] >> shift) & (uint32_t)(D - 1)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += SPH_BLOCK) blockHist[(size_t)i * numBlocks + blockIdx.x] = hist[i];
}

// Offsets in two levels, no single-block pass over the whole table:
//   k_radix_scan_rows: one workgroup per digit turns that digit's row blockHist[digit][0..numBlocks) into an exclusive
//                      scan in place and leaves the row total in digitTotal[digit];
//   k_radix_scatter:   every workgroup scans the digit totals itself (LDS) to get the digit bases.
__global__ __launch_bounds__(SPH_BLOCK) void k_radix_scan_rows(uint32_t* __restrict__ blockHist, int numBlocks,
                                                                uint32_t* __restrict__ digitTotal) {
  __shared__ uint32_t part[SPH_BLOCK];
  uint32_t* row = blockHist + (size_t)blockIdx.x * numBlocks;
  const int per = (numBlocks + SPH_BLOCK - 1) / SPH_BLOCK;
  const int lo = min((int)threadIdx.x * per, numBlocks), hi = min(lo + per, numBlocks);
  uint32_t sum = 0;
  for (int i = lo; i < hi; i++) sum += row[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < SPH_BLOCK; off <<= 1) {  // Hillis-Steele inclusive scan of the per-thread sums
    const uint32_t v = (threadIdx.x >= (unsigned)off) ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - sum;
  for (int i = lo; i < hi; i++) { const uint32_t v = row[i]; row[i] = run; run += v; }
  if (threadIdx.x == SPH_BLOCK - 1) digitTotal[blockIdx.x] = part[SPH_BLOCK - 1];
}

template <int BITS>
__global__ __launch_bounds__(SPH_BLOCK) void k_radix_scatter(const uint32_t* __restrict__ keysIn,
                                                              const uint32_t* __restrict__ valsIn,
                                                              uint32_t* __restrict__ keysOut,
                                                              uint32_t* __restrict__ valsOut, int N, int shift,
                                                              const uint32_t* __restrict__ blockHist, int numBlocks,
                                                              const uint32_t* __restrict__ digitTotal) {
  constexpr int D = 1 << BITS, PER = D / SPH_BLOCK;  // digits, digits per thread in the per-digit steps
  __shared__ volatile uint32_t waveCount[4][D];  // running count of digit d inside wave w's range
  __shared__ uint32_t digitBase[4][D];           // global output index of the first key with digit d of wave w
  __shared__ uint32_t scan[SPH_BLOCK];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4 * D; i += SPH_BLOCK) waveCount[i / D][i % D] = 0;
  uint32_t tot[PER], mySum = 0;  // thread t owns the digits t*PER .. t*PER + PER-1
#pragma unroll
  for (int p = 0; p < PER; p++) { tot[p] = digitTotal[threadIdx.x * PER + p]; mySum += tot[p]; }
  scan[threadIdx.x] = mySum;
  __syncthreads();
  for (int off = 1; off < SPH_BLOCK; off <<= 1) {  // inclusive scan of the per-thread sums of the digit totals
    const uint32_t v = (threadIdx.x >= (unsigned)off) ? scan[threadIdx.x - off] : 0u;
    __syncthreads();
    scan[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t digitStart[PER];  // keys with a smaller digit, whole array
  {
    uint32_t run = scan[threadIdx.x] - mySum;
#pragma unroll
    for (int p = 0; p < PER; p++) { digitStart[p] = run; run += tot[p]; }
  }
  const int base = blockIdx.x * RS_TILE + wave * (64 * RS_ITEMS);
  const unsigned long long ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint32_t key[RS_ITEMS], val[RS_ITEMS], rank[RS_ITEMS];
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const int i = base + r * 64 + lane;
    const bool valid = i < N;
    key[r] = valid ? keysIn[i] : 0u;
    val[r] = valid ? valsIn[i] : 0u;
    const uint32_t dgt = (key[r] >> shift) & (uint32_t)(D - 1);
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; b++) {
      const bool bit = (dgt >> b) & 1u;
      const unsigned long long m = __ballot(valid && bit);
      peers &= bit ? m : ~m;
    }
    // all peers read the running count, then the lowest peer bumps it (LDS ops of one wave retire in order)
    const uint32_t before = valid ? waveCount[wave][dgt] : 0u;
    rank[r] = before + (uint32_t)__popcll(peers & ltMask);
    if (valid && (peers & ltMask) == 0ull) waveCount[wave][dgt] = before + (uint32_t)__popcll(peers);
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < PER; p++) {
    const int dgt = threadIdx.x * PER + p;
    uint32_t run = digitStart[p] + blockHist[(size_t)dgt * numBlocks + blockIdx.x];
#pragma unroll
    for (int w = 0; w < 4; w++) { digitBase[w][dgt] = run; run += waveCount[w][dgt]; }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const int i = base + r * 64 + lane;
    if (i < N) {
      const uint32_t dst = digitBase[wave][(key[r] >> shift) & (uint32_t)(D - 1)] + rank[r];
      keysOut[dst] = key[r];
      valsOut[dst] = val[r];
    }
  }
}

int sphk_sort(sph_solver* s) { return sphk_sort_pairs(s, s->d.N, s->sortBits); }

// passes needed for `bits` key bits with the cheaper digit width (9 bits only when that saves a pass)
int sph_sort_digit_bits(int bits) { return ((bits + 8) / 9 < (bits + 7) / 8) ? 9 : 8; }
int sph_sort_passes(int bits) { const int w = sph_sort_digit_bits(bits); return (bits + w - 1) / w; }

template <int BITS>
static void radix_pass(sph_solver* s, int n, int nb, int shift) {
  SphDev& d = s->d;
  constexpr int D = 1 << BITS;
  uint32_t* digitTotal = s->blockHist + (size_t)SPH_SORT_MAX_DIGITS * s->maxSortBlocks;
  hipLaunchKernelGGL((k_radix_hist<BITS>), dim3(nb), dim3(SPH_BLOCK), 0, s->stream, d.keys, n, shift, s->blockHist, nb);
  hipLaunchKernelGGL(k_radix_scan_rows, dim3(D), dim3(SPH_BLOCK), 0, s->stream, s->blockHist, nb, digitTotal);
  hipLaunchKernelGGL((k_radix_scatter<BITS>), dim3(nb), dim3(SPH_BLOCK), 0, s->stream, d.keys, d.vals, d.keysAlt, d.valsAlt,
                     n, shift, s->blockHist, nb, digitTotal);
  uint32_t* t = d.keys; d.keys = d.keysAlt; d.keysAlt = t;
  t = d.vals; d.vals = d.valsAlt; d.valsAlt = t;
}

int sphk_sort_pairs(sph_solver* s, int n, int bits) {
  const int nb = (n + RS_TILE - 1) / RS_TILE;
  if (nb > s->maxSortBlocks) { sph_set_error("sort of %d keys exceeds the solver's capacity", n); return SPH_ERR_SIZE; }
  const int w = sph_sort_digit_bits(bits);
  for (int shift = 0; shift < bits; shift += w) {
    if (w == 9) radix_pass<9>(s, n, nb, shift);
    else radix_pass<8>(s, n, nb, shift);
  }
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

// ------------------------------------------------------------------ K3 sortPostPass (sphFluid.cl:441-466)
__global__ __launch_bounds__(SPH_BLOCK) void k_sort_post(SphDev d) {
  const int i = blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (i >= d.N) return;
  const uint32_t src = d.vals[i];
  d.sortedPos[i] = d.posOrig[src];  // .w stays the particle type; the cell id lives in keys[]
  d.sortedVel[i] = d.velOrig[src];
  d.backIndex[src] = (uint32_t)i;
}

// K4 indexx (sphFluid.cl:385-439) + the host fix-up loop (owOpenCLSolver.cpp:305-319) in one kernel, one lane per cell:
// cellStart[c] = number of particles whose cell id is < c (lower bound in the sorted keys). For a non-empty cell that
// is its first sorted index (what the reference's binary search finds); for an empty one it is the start of the next
// non-empty cell (what the backward fill writes); [0] = 0 and [G] = N. The key array is L2-resident (4 B/particle).
// Slab mode: only cells [first, last) are computed — the layers the slab's particles and their neighbour cells can lie in — plus
// the two ends of the table, which ranged launches read (sph_ranged: cellStart[0], cellStart[G]).
__global__ __launch_bounds__(SPH_BLOCK) void k_cell_start(SphDev d, int first, int last) {
  int c = first + blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (c >= last) {
    if (c == last) c = 0; else if (c == last + 1) c = d.G; else return;  // two spare lanes take the table's ends
  }
  if (c > d.G) return;
  int lo = 0, hi = d.N;  // first index in [0, N] whose key is >= c
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (d.keys[mid] < (uint32_t)c) lo = mid + 1; else hi = mid;
  }
  d.cellStart[c] = (c == d.G) ? (uint32_t)d.N : (uint32_t)lo;
}

int sphk_sort_post(sph_solver* s) {
  hipLaunchKernelGGL(k_sort_post, dim3(sph_blocks(s->d.N)), dim3(SPH_BLOCK), 0, s->stream, s->d);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}
int sphk_index_fixed(sph_solver* s) {
  int first = 0, last = s->d.G + 1;
  if (s->hasSlab) {
    // A slab's particles lie in its own layers +- ghostLayers; the search reads the table for their cells, the cells one layer
    // up and down and two cells beyond a run's end: layers [layerLo - W - 2, layerHi + W + 3) cover that. The declared grid is the
    // global one (39 us of binary searches per step at 16.5 M cells, whatever the rank's share of the particles).
    const long long layerCells = (long long)s->d.gx * s->d.gy, W = s->slab.ghostLayers;
    const long long lo = max((long long)s->slab.layerLo - W - 2, 0LL) * layerCells;
    const long long hi = min(((long long)s->slab.layerHi + W + 3) * layerCells, (long long)s->d.G + 1);
    first = (int)min(lo, (long long)s->d.G + 1); last = (int)max(hi, (long long)first);
  }
  hipLaunchKernelGGL(k_cell_start, dim3(sph_blocks(last - first + 2)), dim3(SPH_BLOCK), 0, s->stream, s->d, first, last);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}
int sphk_sort_post_and_index(sph_solver* s) {
  int rc = sphk_sort_post(s);
  return rc == SPH_OK ? sphk_index_fixed(s) : rc;
}

// K4 alone, for stage-by-stage parity: first sorted index of each non-empty cell, 0xffffffff (NO_PARTICLE_ID) for empty
// cells, [0] = 0 and [G] = N as sphFluid.cl:398-406 sets them.
__global__ __launch_bounds__(SPH_BLOCK) void k_index_raw(SphDev d) {
  const int i = blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (i >= d.N) return;
  const uint32_t key = d.keys[i];
  if (key < (uint32_t)d.G && key > 0u && (i == 0 || d.keys[i - 1] != key)) d.cellStartRaw[key] = (uint32_t)i;
}
__global__ void k_index_raw_ends(SphDev d) {
  d.cellStartRaw[0] = 0u;
  d.cellStartRaw[d.G] = (uint32_t)d.N;
}

int sphk_index_raw(sph_solver* s) {
  SPH_HIP(hipMemsetAsync(s->d.cellStartRaw, 0xff, sizeof(uint32_t) * ((size_t)s->d.G + 1), s->stream));
  hipLaunchKernelGGL(k_index_raw, dim3(sph_blocks(s->d.N)), dim3(SPH_BLOCK), 0, s->stream, s->d);
  hipLaunchKernelGGL(k_index_raw_ends, dim3(1), dim3(1), 0, s->stream, s->d);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}
