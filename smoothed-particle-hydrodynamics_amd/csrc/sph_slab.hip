// Slab decomposition support (include/sphmi.h "Spatial decomposition"; no reference counterpart — SURVEY.md 8e):
// select the authoritative particles after a step, build the halo messages for the two neighbouring slabs, and rebuild
// the local particle set (kept + received) sorted by global id. Everything stays in HBM; the caller moves the two
// messages with RCCL send/recv (smoothed-particle-hydrodynamics_amd/sphmi/slab.py).
#include "sph_common.h"

// Message record. Full form (SPH_SLAB_RECORD_WORDS = 9 words): position (x, y, z, type), velocity (vx, vy, vz, w), global id.
// Compact form (SPH_SLAB_COMPACT_WORDS = 7 words: x, y, z, vx, vy, vz, global id), used when every rank has verified that all
// non-boundary particles carry the same type word and velocity.w == +0 (sph_slab_liquid_signature / sph_slab_set_record_format):
// neither ever changes (integrate copies position.w and maps velocity.w = 0 to 0), so they travel as one constant.
struct RecFmt { int words; uint32_t typeBits; };

__device__ __forceinline__ void put_record(uint32_t* msg, uint32_t j, const float4 p, const float4 v, uint32_t g, const RecFmt f) {
  uint32_t* r = msg + (size_t)j * f.words;
  if (f.words == SPH_SLAB_COMPACT_WORDS) {
    r[0] = __float_as_uint(p.x); r[1] = __float_as_uint(p.y); r[2] = __float_as_uint(p.z);
    r[3] = __float_as_uint(v.x); r[4] = __float_as_uint(v.y); r[5] = __float_as_uint(v.z);
    r[6] = g;
  } else {
    r[0] = __float_as_uint(p.x); r[1] = __float_as_uint(p.y); r[2] = __float_as_uint(p.z); r[3] = __float_as_uint(p.w);
    r[4] = __float_as_uint(v.x); r[5] = __float_as_uint(v.y); r[6] = __float_as_uint(v.z); r[7] = __float_as_uint(v.w);
    r[8] = g;
  }
}
__device__ __forceinline__ uint32_t record_gid(const uint32_t* msg, int j, const RecFmt f) { return msg[(size_t)j * f.words + (f.words - 1)]; }
__device__ __forceinline__ void get_record(const uint32_t* msg, int j, const RecFmt f, float4& p, float4& v) {
  const uint32_t* r = msg + (size_t)j * f.words;
  if (f.words == SPH_SLAB_COMPACT_WORDS) {
    p = make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(f.typeBits));
    v = make_float4(__uint_as_float(r[3]), __uint_as_float(r[4]), __uint_as_float(r[5]), 0.f);
  } else {
    p = make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
    v = make_float4(__uint_as_float(r[4]), __uint_as_float(r[5]), __uint_as_float(r[6]), __uint_as_float(r[7]));
  }
}
static RecFmt rec_fmt(const sph_solver* s) { return RecFmt{s->slabRecWords ? s->slabRecWords : SPH_SLAB_RECORD_WORDS, s->slabTypeBits}; }

// Owned particles (flag set at the last rebuild) are the authoritative ones. All of them stay in the local set (they move
// less than one cell layer per step); those now within W layers of a cut are also copied into the neighbour's message —
// except boundary particles, which never move: every rank holds the boundary particles of its ghost layers from sph_slab_init
// on and KEEPS them (a ghost that is a boundary particle goes to the kept set), so they are never sent again.
// The compaction is ORDER-PRESERVING: the local arrays are sorted by global id, so the kept set and both messages come
// out sorted by global id and the receiver can rebuild with a three-way merge instead of a sort. Three launches:
//   k_slab_pack<false>  per workgroup of 2048 particles, how many go to each of the three destinations
//   k_slab_scan         exclusive scan of those counts over the workgroups (one workgroup) + the three totals
//   k_slab_pack<true>   the same flags again, slot = workgroup offset + rank inside the workgroup (ballot prefixes)
// owned[] holds 0 for ghosts and 1 + (z layer - slab_base) for owned particles, so that the pack can tell how far a particle moved
__host__ __device__ static inline int slab_base(const sph_slab& slab) { return slab.hasLower ? slab.layerLo - 1 : 0; }
__device__ __forceinline__ uint32_t owned_code(const sph_slab& slab, int layer) {
  return (layer >= slab.layerLo && layer < slab.layerHi) ? (uint32_t)(layer - slab_base(slab) + 1) : 0u;
}

#define PACK_ITEMS 8
#define PACK_SPAN (SPH_BLOCK * PACK_ITEMS)

template <bool WRITE>
__global__ __launch_bounds__(SPH_BLOCK) void k_slab_pack(SphDev d, sph_slab slab, uint32_t* __restrict__ blockCounts,
                                                         const uint32_t* __restrict__ blockOffsets, uint32_t* __restrict__ msgDown,
                                                         uint32_t* __restrict__ msgUp, int capRecords, SlabPart part,
                                                         uint32_t* __restrict__ moved, RecFmt fmt) {
  __shared__ uint32_t tot[PACK_ITEMS * (SPH_BLOCK / 64)][3];  // hits per (round, wave), then their exclusive prefix
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int first = blockIdx.x * PACK_SPAN + tid;
  int a0 = 0, a1 = 0, b0 = 0, b1 = 0;  // messages-only pass: sorted-index ranges of the particles integrated so far
  if (part.mode == SLAB_PART_MESSAGES) {
    a0 = (int)d.cellStart[part.a0]; a1 = (int)d.cellStart[part.a1]; b0 = (int)d.cellStart[part.b0]; b1 = (int)d.cellStart[part.b1];
  }
  uint32_t flags[PACK_ITEMS], rank[PACK_ITEMS];  // bit c: goes to destination c; 10 bits of rank per destination
#pragma unroll
  for (int u = 0; u < PACK_ITEMS; u++) {  // element order inside the workgroup: (round u, thread) = ascending particle index
    const int i = first + u * SPH_BLOCK;
    uint32_t f = 0u;
    const uint32_t ownedAt = (i < d.N) ? d.owned[i] : 0u;  // 0: ghost; otherwise 1 + (z layer at the last rebuild) - slabBase
    if (ownedAt) {
      bool final_ = true;  // messages-only pass of the overlapped step: only particles that have been integrated already
      if (part.mode == SLAB_PART_MESSAGES) {
        const int sid = (int)d.backIndex[i];
        final_ = (sid >= a0 && sid < a1) || (sid >= b0 && sid < b1);
      }
      if (final_) {
        const float4 pz = d.posOrig[i];
        const int layer = (int)(pz.z * d.cellSizeInv);  // the z cell coordinate hashParticles uses (sphFluid.cl:199)
        // The 4-layer halo, the ranged stage launches and the overlapped tail all rest on "a particle moves less than one cell
        // layer per step": count the owned particles that did not (reported as SPH_ERR_INVALID by the host side of the pack).
        if (!WRITE && abs(layer - ((int)ownedAt - 1 + slab_base(slab))) > 1) atomicAdd(moved, 1u);
        if (part.mode != SLAB_PART_MESSAGES) f = 1u;
        if (part.mode != SLAB_PART_KEPT && (int)pz.w != SPH_BOUNDARY_PARTICLE) {  // the boundary shell is static: never sent
          if (slab.hasLower && layer < slab.layerLo + slab.ghostLayers) f |= 2u;
          if (slab.hasUpper && layer >= slab.layerHi - slab.ghostLayers) f |= 4u;
        }
      }
    } else if (i < d.N && part.mode != SLAB_PART_MESSAGES) {
      if ((int)d.posOrig[i].w == SPH_BOUNDARY_PARTICLE) f = 1u;  // a boundary particle of the ghost layers: stays for good
    }
    flags[u] = f;
    uint32_t r = 0u;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const unsigned long long b = __ballot((f >> c) & 1u);
      r |= (uint32_t)__popcll(b & lt) << (10 * c);
      if (lane == 0) tot[u * (SPH_BLOCK / 64) + wave][c] = (uint32_t)__popcll(b);
    }
    rank[u] = r;
  }
  __syncthreads();
  if (tid < 3) {  // exclusive prefix over the 32 (round, wave) groups; the block total goes out in counting mode
    uint32_t run = 0u;
    for (int g = 0; g < PACK_ITEMS * (SPH_BLOCK / 64); g++) { const uint32_t v = tot[g][tid]; tot[g][tid] = run; run += v; }
    if (!WRITE) blockCounts[(size_t)blockIdx.x * 4 + tid] = run;
  }
  if (!WRITE) return;
  __syncthreads();
  const uint32_t base0 = blockOffsets[(size_t)blockIdx.x * 4 + 0], base1 = blockOffsets[(size_t)blockIdx.x * 4 + 1],
                 base2 = blockOffsets[(size_t)blockIdx.x * 4 + 2];
#pragma unroll
  for (int u = 0; u < PACK_ITEMS; u++) {
    const uint32_t f = flags[u];
    if (!f) continue;
    const int i = first + u * SPH_BLOCK;
    const int g = u * (SPH_BLOCK / 64) + wave;
    const float4 p = d.posOrig[i], v = d.velOrig[i];
    const uint32_t gid = d.gid[i];
    if (f & 1u) {
      const uint32_t k = base0 + tot[g][0] + (rank[u] & 1023u);
      d.sortedPos[k] = p; d.sortedVel[k] = v; d.keys[k] = gid;  // sorted* / keys are free between two steps: staging area
    }
    if (f & 2u) { const uint32_t j = base1 + tot[g][1] + ((rank[u] >> 10) & 1023u); if ((int)j < capRecords) put_record(msgDown, j, p, v, gid, fmt); }
    if (f & 4u) { const uint32_t j = base2 + tot[g][2] + ((rank[u] >> 20) & 1023u); if ((int)j < capRecords) put_record(msgUp, j, p, v, gid, fmt); }
  }
}

// blockOffsets[b][c] = sum of blockCounts[b'][c] over b' < b; counts[c] = grand totals. One workgroup, 256 blocks per trip.
__global__ __launch_bounds__(SPH_BLOCK) void k_slab_scan(const uint32_t* __restrict__ blockCounts, uint32_t* __restrict__ blockOffsets,
                                                         int nb, uint32_t* __restrict__ counts, uint32_t* __restrict__ headDown,
                                                         uint32_t* __restrict__ headUp, int recWords) {
  __shared__ uint32_t buf[SPH_BLOCK][3];
  __shared__ uint32_t carry[3];
  const int tid = threadIdx.x;
  if (tid < 3) carry[tid] = 0u;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += SPH_BLOCK) {
    const int b = b0 + tid;
    uint32_t v[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { v[c] = (b < nb) ? blockCounts[(size_t)b * 4 + c] : 0u; buf[tid][c] = v[c]; }
    __syncthreads();
    for (int off = 1; off < SPH_BLOCK; off <<= 1) {  // inclusive Hillis-Steele scan of the three columns
      uint32_t add[3] = {0u, 0u, 0u};
      if (tid >= off) { add[0] = buf[tid - off][0]; add[1] = buf[tid - off][1]; add[2] = buf[tid - off][2]; }
      __syncthreads();
      buf[tid][0] += add[0]; buf[tid][1] += add[1]; buf[tid][2] += add[2];
      __syncthreads();
    }
    if (b < nb) {
#pragma unroll
      for (int c = 0; c < 3; c++) blockOffsets[(size_t)b * 4 + c] = carry[c] + buf[tid][c] - v[c];
    }
    __syncthreads();
    if (tid < 3) carry[tid] += buf[SPH_BLOCK - 1][tid];
    __syncthreads();
  }
  if (tid < 3) counts[tid] = carry[tid];
  // framed messages ([payload words | payload | padding], sph_slab_pack_framed): the count word is written here, on the device
  if (tid == 1 && headDown) *headDown = carry[1] * (uint32_t)recWords;
  if (tid == 2 && headUp) *headUp = carry[2] * (uint32_t)recWords;
}

int sphk_slab_pack(sph_solver* s, uint32_t* msgDown, uint32_t* msgUp, int capRecords, uint32_t* headDown, uint32_t* headUp,
                   SlabPart part, uint32_t* counts) {
  const int nb = sph_blocks(s->d.N, PACK_SPAN);
  uint32_t* blockCounts = s->blockHist;  // the radix-sort workspace is idle between two steps: >= capacity/16 words
  uint32_t* blockOffsets = s->blockHist + (size_t)nb * 4;
  if (!counts) counts = s->slabCounts;
  const RecFmt fmt = rec_fmt(s);
  hipLaunchKernelGGL((k_slab_pack<false>), dim3(nb), dim3(SPH_BLOCK), 0, s->stream, s->d, s->slab, blockCounts, blockOffsets, msgDown, msgUp, capRecords, part, s->slabCounts + 7, fmt);
  hipLaunchKernelGGL(k_slab_scan, dim3(1), dim3(SPH_BLOCK), 0, s->stream, blockCounts, blockOffsets, nb, counts, headDown, headUp, fmt.words);
  hipLaunchKernelGGL((k_slab_pack<true>), dim3(nb), dim3(SPH_BLOCK), 0, s->stream, s->d, s->slab, blockCounts, blockOffsets, msgDown, msgUp, capRecords, part, s->slabCounts + 7, fmt);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
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
  d.owned[i] = owned_code(slab, (int)(p.z * d.cellSizeInv));
}

// ---- rebuild by three-way merge. The kept set (staging: sortedPos / sortedVel / keys) and both received messages are
// sorted by global id and global ids are unique, so the final index of an element is its own index plus the number of
// smaller ids in the other two lists (two binary searches). A message that is not sorted raises slabCounts[3], which the
// next sph_slab_pack reports as an error.
// The three list lengths may live on the DEVICE (sph_slab_rebuild_framed): the kept count where the pack left it, the message
// lengths in word 0 of the received frames — the host then enqueues the rebuild without knowing them and reads the totals back
// once, afterwards (SlabIn::*Ptr non-null; the launch grids cover the capacities). A frame that claims more records than its
// buffer holds, or a total beyond the solver's capacity, makes every merge kernel return at once and is reported by the totals.
struct SlabIn {
  const uint32_t* keptPtr; int keptHost;                        // kept count: *keptPtr if non-null
  const uint32_t *down, *up;                                    // payloads (records)
  const uint32_t *downWordsPtr, *upWordsPtr; int nDownHost, nUpHost;  // lengths: *wordsPtr / record words if non-null
  int capDown, capUp, capacity;                                 // records the payload buffers hold; particles the solver holds
};
struct SlabCounts { int kept, nDown, nUp; bool ok; };
__device__ __forceinline__ SlabCounts slab_counts(const SlabIn& in, const RecFmt f) {
  SlabCounts c;
  c.kept = in.keptPtr ? (int)*in.keptPtr : in.keptHost;
  c.nDown = in.down ? (in.downWordsPtr ? (int)(*in.downWordsPtr / (uint32_t)f.words) : in.nDownHost) : 0;
  c.nUp = in.up ? (in.upWordsPtr ? (int)(*in.upWordsPtr / (uint32_t)f.words) : in.nUpHost) : 0;
  c.ok = c.nDown <= in.capDown && c.nUp <= in.capUp && (long long)c.kept + c.nDown + c.nUp <= (long long)in.capacity;
  return c;
}
__device__ __forceinline__ int lower_bound_keys(const uint32_t* __restrict__ keys, int n, uint32_t g) {
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < g) lo = mid + 1; else hi = mid; }
  return lo;
}
__device__ __forceinline__ int lower_bound_msg(const uint32_t* __restrict__ msg, int n, uint32_t g, const RecFmt f) {
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (record_gid(msg, mid, f) < g) lo = mid + 1; else hi = mid; }
  return lo;
}
__device__ __forceinline__ void place_particle(const SphDev& d, const sph_slab& slab, int at, const float4 p, const float4 v, uint32_t g) {
  d.posOrig[at] = p;
  d.velOrig[at] = v;
  d.gid[at] = g;
  d.owned[at] = owned_code(slab, (int)(p.z * d.cellSizeInv));
}

// totals: [0] kept, [1] records from below, [2] from above, [3] 0 = merged / 1 = nothing done (a frame or the capacity overflowed)
__global__ __launch_bounds__(SPH_BLOCK) void k_slab_merge_kept(SphDev d, sph_slab slab, SlabIn in, RecFmt f, uint32_t* __restrict__ totals) {
  const SlabCounts c = slab_counts(in, f);
  const int i = blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (i == 0 && totals) { totals[0] = (uint32_t)c.kept; totals[1] = (uint32_t)c.nDown; totals[2] = (uint32_t)c.nUp; totals[3] = c.ok ? 0u : 1u; }
  if (!c.ok || i >= c.kept) return;
  const uint32_t g = d.keys[i];
  const int at = i + lower_bound_msg(in.down, c.nDown, g, f) + lower_bound_msg(in.up, c.nUp, g, f);
  place_particle(d, slab, at, d.sortedPos[i], d.sortedVel[i], g);
}

template <bool UP>
__global__ __launch_bounds__(SPH_BLOCK) void k_slab_merge_msg(SphDev d, sph_slab slab, SlabIn in, RecFmt f, uint32_t* __restrict__ unsorted) {
  const SlabCounts c = slab_counts(in, f);
  const int j = blockIdx.x * SPH_BLOCK + threadIdx.x;
  const uint32_t* msg = UP ? in.up : in.down;
  const uint32_t* other = UP ? in.down : in.up;
  const int n = UP ? c.nUp : c.nDown, nOther = UP ? c.nDown : c.nUp;
  if (!c.ok || j >= n) return;
  const uint32_t g = record_gid(msg, j, f);
  if (j > 0 && record_gid(msg, j - 1, f) >= g) atomicOr(unsorted, 1u);
  const int at = j + lower_bound_keys(d.keys, c.kept, g) + lower_bound_msg(other, nOther, g, f);
  float4 p, v;
  get_record(msg, j, f, p, v);
  place_particle(d, slab, at, p, v, g);
}

static int launch_merge(sph_solver* s, const SlabIn& in, int keptGrid, int downGrid, int upGrid, uint32_t* totals) {
  const RecFmt fmt = rec_fmt(s);
  { const int g = sph_guard_position_write(s); if (g != SPH_OK) return g; }
  hipLaunchKernelGGL(k_slab_merge_kept, dim3(sph_blocks(keptGrid > 0 ? keptGrid : 1)), dim3(SPH_BLOCK), 0, s->stream, s->d, s->slab, in, fmt, totals);
  if (in.down && downGrid > 0) hipLaunchKernelGGL((k_slab_merge_msg<false>), dim3(sph_blocks(downGrid)), dim3(SPH_BLOCK), 0, s->stream, s->d, s->slab, in, fmt, s->slabCounts + 3);
  if (in.up && upGrid > 0) hipLaunchKernelGGL((k_slab_merge_msg<true>), dim3(sph_blocks(upGrid)), dim3(SPH_BLOCK), 0, s->stream, s->d, s->slab, in, fmt, s->slabCounts + 3);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

int sphk_slab_rebuild(sph_solver* s, const uint32_t* recvDown, int nDown, const uint32_t* recvUp, int nUp, int kept) {
  const int total = kept + nDown + nUp;
  SlabIn in{nullptr, kept, nDown ? recvDown : nullptr, nUp ? recvUp : nullptr, nullptr, nullptr, nDown, nUp, nDown, nUp, s->capacity};
  const int rc = launch_merge(s, in, kept, nDown, nUp, nullptr);
  if (rc != SPH_OK) return rc;
  s->d.N = total;
  s->progress = 0;
  return SPH_OK;
}

// The same with every length on the device: frames are [payload words | payload]; the kept count is at keptPtr. s->d.N is NOT
// updated here (the host does not know the total yet): sph_api.hip reads `totals` back and finishes the rebuild.
int sphk_slab_rebuild_framed(sph_solver* s, const uint32_t* frameDown, int capDown, const uint32_t* frameUp, int capUp,
                             const uint32_t* keptPtr, uint32_t* totals) {
  SlabIn in{keptPtr, 0, frameDown ? frameDown + 1 : nullptr, frameUp ? frameUp + 1 : nullptr, frameDown, frameUp, 0, 0, capDown, capUp, s->capacity};
  return launch_merge(s, in, s->d.N, capDown, capUp, totals);  // (the kept set is a subset of the current local set)
}

// Initial set in any order (sph_slab_init): staging area holds n records; sort them by global id.
int sphk_slab_sort_rebuild(sph_solver* s, int total) {
  hipLaunchKernelGGL(k_iota, dim3(sph_blocks(total)), dim3(SPH_BLOCK), 0, s->stream, s->d.vals, total);
  int rc = sphk_sort_pairs(s, total, s->slab.globalIdBits);  // stable LSD radix sort by global id (keys), vals follow
  if (rc != SPH_OK) return rc;
  rc = sph_guard_position_write(s);
  if (rc != SPH_OK) return rc;
  hipLaunchKernelGGL(k_slab_gather, dim3(sph_blocks(total)), dim3(SPH_BLOCK), 0, s->stream, s->d, s->slab, total);
  SPH_HIP(hipGetLastError());
  s->d.N = total;
  s->progress = 0;
  return SPH_OK;
}
