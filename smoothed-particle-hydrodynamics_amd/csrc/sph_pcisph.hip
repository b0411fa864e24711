// PCISPH kernels K6, K7, K9-K13 (sphFluid.cl:472-708, 824-1212, 1684-1808) for gfx950.
//
// One lane per sorted particle (the reference's `id = particleIndexBack[get_global_id]` is a pure thread permutation and
// is dropped — SURVEY App. B #11). Every kernel streams its own 32 neighbour slots from the tiled maps — the ids as 16-bit
// offsets, eight coalesced 8-byte loads per lane (NbrTile; sph_common.h, SPH_N16_*), the distances where it needs them as eight
// 16-byte loads — and gathers neighbour state as single 12/16-byte transactions.
// Arithmetic keeps the reference's operand types and evaluation order: no FMA contraction, IEEE division/sqrt,
// f32 denormals kept, f64 where sphFluid.cl uses double.
#include "sph_common.h"
#include "sph_fastmath.h"

#define TYPE_OF(p4) ((int)(p4).w)
typedef float f32x3 __attribute__((ext_vector_type(3)));
// Predicted positions are packed (x, y, z), 12 bytes per particle: 10.7 records per 128-byte line instead of 8 — the gathers of
// predictDensity touch a quarter fewer lines (tools/micro/gather_lanes.hip: -25 % of the gather cost) and every stream of them is
// 12 bytes instead of 16. (The .w the reference keeps there is dead data.)
typedef f32x3 f32x3a __attribute__((aligned(4)));
__device__ __forceinline__ float4 load_pred(const SphDev& d, int id) {
  const f32x3a v = *reinterpret_cast<const f32x3a*>(d.predPos + 3 * (size_t)id);
  return make_float4(v.x, v.y, v.z, 0.f);
}
__device__ __forceinline__ void store_pred(const SphDev& d, int id, const float4 p) {
  const f32x3a v = {p.x, p.y, p.z};
  *reinterpret_cast<f32x3a*>(d.predPos + 3 * (size_t)id) = v;
}
// DIAGNOSTIC build only (timings, results invalid): DIAG_OWN_GATHER reads the lane's own record instead of the neighbour's, so
// every gather becomes a coalesced load. A/B on MI355X, config #2: forces 0.148 -> 0.061 ms, predictDensity x3 0.115 -> 0.083,
// pressure force x3 0.284 -> 0.199: the 16-byte neighbour gathers (64 different cache lines per wave instruction) cost about
// a fifth of the step; the kernels take the same time per particle at 1 M (cache resident) and at 16.5 M particles.
#ifdef DIAG_OWN_GATHER
#define NBR_INDEX(j) (id)
#else
#define NBR_INDEX(j) (max((j), 0))
#endif

// neighbour gathers kept in flight per lane by the branch-free kernels (A/B on MI355X, config #2)
#ifndef FC_BATCH
#define FC_BATCH 8   // forces: two 16-B gathers per neighbour
#endif
#ifndef PF_BATCH
#define PF_BATCH 8   // pressure force: 16 B + 4 B per neighbour
#endif
#ifndef PD_BATCH
#define PD_BATCH 16  // predicted density: 16 B per neighbour
#endif

// index into SphDev::gatherRec: groups of four particles, [4 x part 0][4 x part 1] (one 128-byte line; see k_pack_gather_records)
__device__ __forceinline__ size_t rec_index(int j, int part) { return ((size_t)(j >> 2) << 3) + (size_t)(part << 2) + (size_t)(j & 3); }

// the 8 x (int4 ids, float4 dists) of particle `id`, and its ids again as 8 x (4 x 16-bit offsets) (sph_common.h, SPH_N16_*)
struct NbrTile {
  const int4* ids;
  const float4* dist;
  const uint2* v16;
  int self, zOff;  // the particle and (base of its flagged offsets) - (the particle)
  __device__ __forceinline__ NbrTile(const SphDev& d, int id) {
    const size_t base = ((size_t)(id >> 6) * 8) * 64 + (size_t)(id & 63);
    ids = reinterpret_cast<const int4*>(d.nbrId) + base;
    dist = reinterpret_cast<const float4*>(d.nbrDist) + base;
    v16 = reinterpret_cast<const uint2*>(d.nbr16) + base;
    self = id;
    zOff = d.nbrBase[id] - id;
  }
  __device__ __forceinline__ int4 id4(int g) const { return ids[(size_t)g * 64]; }
  // The rows are streamed once per kernel: non-temporal loads keep them from displacing the records the gathers live on
  // (A/B at 16.5 M, profiles/r03/gather_rows_nontemporal_ab.txt: predictDensity 0.498 -> 0.490 ms, forces 1.549 -> 1.529, pressure force unchanged)
  __device__ __forceinline__ float4 dist4(int g) const {
    typedef float nt4 __attribute__((ext_vector_type(4)));
    const nt4 q = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(&dist[(size_t)g * 64]));
    return make_float4(q.x, q.y, q.z, q.w);
  }
  __device__ __forceinline__ uint2 vec16(int g) const {  // slots 4g .. 4g+3
    typedef unsigned int nt2 __attribute__((ext_vector_type(2)));
    const nt2 q = __builtin_nontemporal_load(reinterpret_cast<const nt2*>(&v16[(size_t)g * 64]));
    return make_uint2(q.x, q.y);
  }
  // the row has no 16-bit copy (an offset did not fit, or findNeighbors served the particle by its exact walk): use id_wide()
  __device__ __forceinline__ static bool wide(const uint2& group0) { return (group0.x & 0xffffu) == SPH_N16_WIDE; }
  __device__ __forceinline__ int id_wide(int slot) const {
    return reinterpret_cast<const int32_t*>(ids)[((size_t)(slot >> 2) * 64) * 4 + (size_t)(slot & 3)];
  }
  // entry k (0..3) of a group -> sorted index of the neighbour, -1 for an empty slot
  __device__ __forceinline__ int decode(const uint2& v, int k) const {
    const uint32_t w = (k >> 1) == 0 ? v.x : v.y;
    const uint32_t e = (k & 1) ? (w >> 16) : (w & 0xffffu);
    const int j = self - SPH_N16_BIAS + (int)(e & 0x7fffu) + ((e & 0x8000u) ? zOff : 0);
    return e == SPH_N16_EMPTY ? -1 : j;
  }
};

// XCD-aware block order: the hardware deals consecutive workgroups round-robin over the 8 XCDs; remap so that each
// XCD works on one contiguous eighth of the sorted particle range and its L2 holds only that slab's neighbourhood.
// Range-aware form for launches restricted to a cell range (slab mode): the remap must run over the blocks that HAVE work.
// Remapping over the whole grid would hand the first eighth of the logical blocks — for a short range, all of the work — to
// one XCD (a 5-layer launch then took as long as the full one: 120 us instead of 17).
__device__ __forceinline__ bool xcd_range_id(const SphDev& d, int& id) {
  const int begin = (int)d.cellStart[d.rangeLo], end = (int)d.cellStart[d.rangeHi];
  const int active = (end - begin + SPH_BLOCK - 1) / SPH_BLOCK;
  int b = blockIdx.x;
  if (b >= active) return false;
#ifndef NO_XCD_REMAP
  const int per = active >> 3, even = per << 3;
  if (b < even) {
    const int x = b & 7;
    int k = b >> 3;  // k-th workgroup dealt to XCD x
    // (a CU-local sub-order inside the XCD — every CU a contiguous sub-range, hoping its resident workgroups share L1 lines — was
    // measured: 0-7 % slower on all three gather kernels)
    b = x * per + k;
  }
#endif
  id = begin + b * SPH_BLOCK + threadIdx.x;
  return id < end;
}

__device__ __forceinline__ int xcd_block(int nblocks) {
#ifdef NO_XCD_REMAP
  return blockIdx.x;
#endif
  const int b = blockIdx.x;
  const int per = nblocks >> 3;  // blocks per XCD in the evenly divisible part
  const int even = per << 3;
  if (b >= even) return b;       // tail blocks keep their index
  return (b & 7) * per + (b >> 3);
}

// ------------------------------------------------------------------ K6 pcisph_computeDensity (sphFluid.cl:472-518)
// The HBM-roofline-graded pass: reads 32 x 4 B of distances, writes 4 B. The empty-slot test uses the distance
// sentinel (-1), which is set iff the id is -1 (both are written together), so the id half of the map is not read.
// A/B on MI355X (bench.py roofline, 16.5 M particles): XCD-remapped blocks + plain loads 0.66 of 8 TB/s, plain block
// order 0.67-0.69, plain order + non-temporal loads 0.75 (6.0 TB/s = 96 % of this part's 6.29 TB/s copy rate). The pass has
// no reuse, so neither an XCD-local L2 nor keeping the stream in cache helps.
#ifndef DENSITY_ITEMS
#define DENSITY_ITEMS 1   // particles per thread, all of their 8 x 16-B loads in flight together (A/B in profiles/README.md)
#endif
__global__ __launch_bounds__(SPH_BLOCK) void k_density(SphDev d, int nblocks) {
  typedef float nt4 __attribute__((ext_vector_type(4)));
  const int begin = (int)d.cellStart[d.rangeLo], end = (int)d.cellStart[d.rangeHi];
  const int first = begin + blockIdx.x * (SPH_BLOCK * DENSITY_ITEMS) + threadIdx.x;
  float4 r[DENSITY_ITEMS][8];
#pragma unroll
  for (int it = 0; it < DENSITY_ITEMS; it++) {
    const int id = first + it * SPH_BLOCK;
    if (id < end) {
      const NbrTile t(d, id);
#pragma unroll
      for (int g = 0; g < 8; g++) {  // all loads in flight before the first use
        const nt4 q = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(&t.dist[(size_t)g * 64]));
        r[it][g] = make_float4(q.x, q.y, q.z, q.w);
      }
    }
  }
#pragma unroll
  for (int it = 0; it < DENSITY_ITEMS; it++) {
    const int id = first + it * SPH_BLOCK;
    if (id >= end) continue;
    double density = 0.0;
#pragma unroll
    for (int g = 0; g < 8; g++) {
      const float rr[4] = {r[it][g].x, r[it][g].y, r[it][g].z, r[it][g].w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (rr[k] != -1.f) {
          const float r2 = rr[k] * rr[k];
          const float a = d.hs2 - r2;
          density += (double)(a * a * a);  // no r < hScaled test here
        }
      }
    }
    if (density < (double)d.hs6) density = (double)d.hs6;
    density *= d.massWpoly6;
#ifdef DIAG_DENSITY_INTO_REC  // timing experiment (results invalid): the density as a 4-byte store into the gather record's half-line
    reinterpret_cast<float*>(&d.gatherRec[rec_index(id, 1)])[3] = (float)density;
#else
    d.rho[id] = (float)density;
#endif
  }
}

int sphk_density(sph_solver* s, int ghostDepth) {
  const int nb = sph_blocks(s->d.N, SPH_BLOCK * DENSITY_ITEMS);
  hipLaunchKernelGGL(k_density, dim3(nb), dim3(SPH_BLOCK), 0, s->stream, sph_ranged(s, ghostDepth), nb);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

// ------------------------------------------------------------------ K9 pcisph_predictPositions (sphFluid.cl:889-979)
__device__ __forceinline__ float4 predict_position(const SphDev& d, const float4 x, const float4 v, const float4 ap) {
  if (TYPE_OF(x) == SPH_BOUNDARY_PARTICLE) return x;
  // v* = v + dt*a_p (pressure acceleration only, :924); x* = x + (dt/simScale)*v*
  float4 o;
  o.x = x.x + d.posTimeStep * (v.x + d.dt * ap.x);
  o.y = x.y + d.posTimeStep * (v.y + d.dt * ap.y);
  o.z = x.z + d.posTimeStep * (v.z + d.dt * ap.z);
  o.w = x.w;  // dead in the reference (holds cell id + ...); we keep the type
  return o;
}

__global__ __launch_bounds__(SPH_BLOCK) void k_predict_positions(SphDev d) {
  const int id = blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (id >= d.N) return;
  store_pred(d, id, predict_position(d, d.sortedPos[id], d.sortedVel[id], d.accP[id]));
}

int sphk_predict_positions(sph_solver* s) {
  hipLaunchKernelGGL(k_predict_positions, dim3(sph_blocks(s->d.N)), dim3(SPH_BLOCK), 0, s->stream, s->d);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

// ------------------------------------------------------------------ K7 pcisph_computeForcesAndInitPressure (sphFluid.cl:589-708)
template <bool FUSE_PREDICT>
__global__ __launch_bounds__(SPH_BLOCK, 4) void k_forces(SphDev d, int nblocks) {  // <= 128 VGPRs: four waves per SIMD
  int id;
  if (!xcd_range_id(d, id)) return;
  const float4 xi = d.sortedPos[id];
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  if (TYPE_OF(xi) == SPH_BOUNDARY_PARTICLE) {
    d.acc[id] = zero;
    if (FUSE_PREDICT) store_pred(d, id, xi);
    else { d.accP[id] = zero; d.rp[id].y = 0.f; }  // (fused step: nobody reads either before the next kernel overwrites it)
    return;
  }
  const float4 vi = d.sortedVel[id];
  const NbrTile t(d, id);
  float sx = 0.f, sy = 0.f, sz = 0.f, tx = 0.f, ty = 0.f, tz = 0.f;
  uint32_t bnd = 0u, ela = 0u;  // which neighbour slots hold boundary / elastic particles: saves integrate and the
                                // membrane kernel 32 type gathers per particle
  bool wideRow = false;
  // Branch-free (see k_predict_density): per batch of FC_BATCH neighbours the map loads, then all gathers in flight together.
  // The map is read batch by batch (not all 32 entries up front) and the velocity gather takes 12 bytes: at 193 VGPRs the
  // kernel ran two waves per SIMD, too few to cover its gathers.
#pragma unroll 1  // (a real loop: unrolled, hipcc hoists the loads of all four batches to the top and needs ~190 VGPRs)
  for (int b = 0; b < 32 / FC_BATCH; b++) {
    int jj[FC_BATCH];
    float rr[FC_BATCH];
#pragma unroll
    for (int q = 0; q < FC_BATCH / 4; q++) {  // (recomputing r from the gathered positions, as K12 does, is not faster here: 1.505 vs 1.521 ms)
      const float4 rq = t.dist4(b * (FC_BATCH / 4) + q);
      rr[4 * q] = rq.x; rr[4 * q + 1] = rq.y; rr[4 * q + 2] = rq.z; rr[4 * q + 3] = rq.w;
    }
#pragma unroll
    for (int q = 0; q < FC_BATCH / 4; q++) {
      const uint2 v = t.vec16(b * (FC_BATCH / 4) + q);
      if (b == 0 && q == 0) wideRow = NbrTile::wide(v);
#pragma unroll
      for (int k = 0; k < 4; k++) jj[4 * q + k] = t.decode(v, k);
    }
    if (wideRow) {  // rare
#pragma unroll
      for (int k = 0; k < FC_BATCH; k++) jj[k] = t.id_wide(b * FC_BATCH + k);
    }
    float4 xj[FC_BATCH], vr[FC_BATCH];
#pragma unroll
    for (int k = 0; k < FC_BATCH; k++) {
      const int jc = NBR_INDEX(jj[k]);
      xj[k] = d.gatherRec[rec_index(jc, 0)];
      vr[k] = d.gatherRec[rec_index(jc, 1)];  // (v.xyz, rho); for a boundary neighbour v is its wall normal (sphFluid.cl:653)
    }
#pragma unroll
    for (int k = 0; k < FC_BATCH; k++) {
      const int slot = b * FC_BATCH + k;
      const bool valid = jj[k] != -1;
      if (valid && TYPE_OF(xj[k]) == SPH_BOUNDARY_PARTICLE) bnd |= 1u << slot;
      if (valid && TYPE_OF(xj[k]) == SPH_ELASTIC_PARTICLE) ela |= 1u << slot;
      const bool use = valid && rr[k] < d.hs;
      const float rj = vr[k].w;
      const float w = d.hs - rr[k];
      sx = use ? sx + (vr[k].x - vi.x) * w / rj : sx;
      sy = use ? sy + (vr[k].y - vi.y) * w / rj : sy;
      sz = use ? sz + (vr[k].z - vi.z) * w / rj : sz;
      tx = use ? tx + d.surfTens * (xi.x - xj[k].x) : tx;
      ty = use ? ty + d.surfTens * (xi.y - xj[k].y) : ty;
      tz = use ? tz + d.surfTens * (xi.z - xj[k].z) : tz;
    }
  }
  d.bndMask[id] = bnd;
  if (d.hasElastic) d.elasticMask[id] = ela;
  const float scale = d.massMu * (float)(d.del2W / (double)d.rho[id]);
  float4 a;
  a.x = sx * scale + d.gravx + tx;
  a.y = sy * scale + d.gravy + ty;
  a.z = sz * scale + d.gravz + tz;
  a.w = 0.f;  // acceleration.w is never read (integrate zeroes it, sphFluid.cl:1721)
  d.acc[id] = a;
  // The staged API needs pressure = 0 and a zero pressure acceleration in memory; in the fused step the first predictDensity
  // starts from p = 0 by itself and the pressure acceleration is not read before the last pressure-force kernel writes it.
  if (FUSE_PREDICT) store_pred(d, id, predict_position(d, xi, vi, zero));
  else { d.accP[id] = zero; d.rp[id].y = 0.f; }
}

// Gather records of the forces kernel: what it needs of a neighbour — (x, y, z, type) and (v.xyz, rho) — in ONE 128-byte line,
// groups of four particles: [4 x (x,y,z,type)][4 x (vx,vy,vz,rho)]. A/B on MI355X at 16.5 M particles (tools/time_stage.py):
// three gathers from three arrays (position, velocity, density) 2.06 ms; position and velocity in one line, density apart
// 1.54 ms; everything in one line 1.22 ms + this pack pass (68 B per particle, 0.2 ms). A wave's gather instruction costs by
// the number of different lines it touches. Kept out of k_density so that the roofline-graded pass moves exactly its 132 B.

__global__ __launch_bounds__(SPH_BLOCK) void k_pack_gather_records(SphDev d) {
  const int id = blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (id >= d.N) return;
  const float4 v = d.sortedVel[id];
  d.gatherRec[rec_index(id, 0)] = d.sortedPos[id];
  d.gatherRec[rec_index(id, 1)] = make_float4(v.x, v.y, v.z, d.rho[id]);
}

// Slab mode: forces are only needed on the owned layers, but every local particle needs what K7 does besides the
// acceleration — the iteration-0 predicted position (pressure = 0 is implied by the first fused predictDensity).
__global__ __launch_bounds__(SPH_BLOCK) void k_ghost_init(SphDev d) {
  const int id = blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (id >= d.N) return;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  store_pred(d, id, predict_position(d, d.sortedPos[id], d.sortedVel[id], zero));
}

int sphk_ghost_init(sph_solver* s) {
  hipLaunchKernelGGL(k_ghost_init, dim3(sph_blocks(s->d.N)), dim3(SPH_BLOCK), 0, s->stream, s->d);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

int sphk_forces(sph_solver* s, bool fusePredict, int ghostDepth) {
  const int nb = sph_blocks(s->d.N);
  const SphDev d = sph_ranged(s, ghostDepth);
  hipLaunchKernelGGL(k_pack_gather_records, dim3(nb), dim3(SPH_BLOCK), 0, s->stream, s->d);
  if (fusePredict) hipLaunchKernelGGL((k_forces<true>), dim3(nb), dim3(SPH_BLOCK), 0, s->stream, d, nb);
  else hipLaunchKernelGGL((k_forces<false>), dim3(nb), dim3(SPH_BLOCK), 0, s->stream, d, nb);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

// ------------------------------------------------------------------ K10 pcisph_predictDensity (+ K11 correctPressure)
// (sphFluid.cl:982-1059, 1062-1098)
__device__ __forceinline__ float corrected_pressure(const SphDev& d, float p, float rhoPred) {
  const float rho_err = rhoPred - d.rho0;
  float p_corr = rho_err * d.delta;
  if (p_corr < 0.f) p_corr = 0.f;
  return p + p_corr;
}

// FIRST (fused step, iteration 0): the pressure being corrected is the 0 that K7 sets, without reading it.
template <bool FUSE_CORRECT, bool FIRST>
__global__ __launch_bounds__(SPH_BLOCK) void k_predict_density(SphDev d, int nblocks) {
  int id;
  if (!xcd_range_id(d, id)) return;
  const float4 xi = load_pred(d, id);
  const NbrTile t(d, id);
  // Branch-free: all 8 id loads first, then the gathers in batches of 8 with an always-valid index (empty slots read
  // record 0 and are masked out of the sum), so a wave keeps 8 gathers in flight instead of one per `if`.
  uint2 v16[8];  // the ids as 16-bit offsets: 64 bytes per particle instead of 128 (this kernel streams little else)
#pragma unroll
  for (int g = 0; g < 8; g++) v16[g] = t.vec16(g);
  const bool wideRow = NbrTile::wide(v16[0]);
  double density = 0.0;
#pragma unroll
  for (int b = 0; b < 32 / PD_BATCH; b++) {
    int jj[PD_BATCH];
#pragma unroll
    for (int k = 0; k < PD_BATCH; k++) {
      const int slot = b * PD_BATCH + k;
      jj[k] = t.decode(v16[slot >> 2], slot & 3);
    }
    if (wideRow) {  // rare
#pragma unroll
      for (int k = 0; k < PD_BATCH; k++) jj[k] = t.id_wide(b * PD_BATCH + k);
    }
    float4 xj[PD_BATCH];
#pragma unroll
    for (int k = 0; k < PD_BATCH; k++) xj[k] = load_pred(d, NBR_INDEX(jj[k]));
#pragma unroll
    for (int k = 0; k < PD_BATCH; k++) {
      const float rx = xi.x - xj[k].x, ry = xi.y - xj[k].y, rz = xi.z - xj[k].z;
      const float r2 = (rx * rx + ry * ry + rz * rz) * d.simScale * d.simScale;
      const float a = d.hs2 - r2;
      // a skipped term is added as +0.0: the sum is >= +0 (it starts there and only grows), so x + 0.0 == x bit for bit
      const float term = (jj[k] != -1 && r2 < d.hs2) ? a * a * a : 0.f;
      density += (double)term;
    }
  }
  if (density < (double)d.hs6) density = (double)d.hs6;
  density *= d.massWpoly6;
  const float rhoP = (float)density;
  if (FUSE_CORRECT) {  // (rho*, p) is one 8-byte record: what the pressure-force kernel gathers per neighbour besides the position
    const float pOld = FIRST ? 0.f : d.rp[id].y;
    d.rp[id] = make_float2(rhoP, corrected_pressure(d, pOld, rhoP));
  } else {
    d.rp[id].x = rhoP;
  }
}

int sphk_predict_density(sph_solver* s, bool fuseCorrect, int ghostDepth, bool first) {
  const int nb = sph_blocks(s->d.N);
  const SphDev d = sph_ranged(s, ghostDepth);
  if (fuseCorrect && first) hipLaunchKernelGGL((k_predict_density<true, true>), dim3(nb), dim3(SPH_BLOCK), 0, s->stream, d, nb);
  else if (fuseCorrect) hipLaunchKernelGGL((k_predict_density<true, false>), dim3(nb), dim3(SPH_BLOCK), 0, s->stream, d, nb);
  else hipLaunchKernelGGL((k_predict_density<false, false>), dim3(nb), dim3(SPH_BLOCK), 0, s->stream, d, nb);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

__global__ __launch_bounds__(SPH_BLOCK) void k_correct_pressure(SphDev d) {
  const int id = blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (id >= d.N) return;
  const float2 v = d.rp[id];
  d.rp[id].y = corrected_pressure(d, v.y, v.x);
}

int sphk_correct_pressure(sph_solver* s) {
  hipLaunchKernelGGL(k_correct_pressure, dim3(sph_blocks(s->d.N)), dim3(SPH_BLOCK), 0, s->stream, s->d);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

// ------------------------------------------------------------------ K13 pcisph_integrate (sphFluid.cl:1684-1808)
// + computeInteractionWithBoundaryParticles (sphFluid.cl:824-887), tangVel = true.
// Boundary neighbours are read from the sorted arrays: boundary particles never move, so sortedPos/sortedVel hold the
// same bits as position[]/velocity[] of the reference and the particleIndex indirection is unnecessary.
__device__ __forceinline__ void integrate_particle(const SphDev& d, int id, const float4 x, const float4 v, const float4 a0,
                                                   const float4 a1) {
  const float ax = a0.x + a1.x, ay = a0.y + a1.y, az = a0.z + a1.z;  // acceleration_.w = 0
  float nvx = v.x + d.dt * ax, nvy = v.y + d.dt * ay, nvz = v.z + d.dt * az, nvw = v.w + d.dt * 0.f;
  float nx = x.x + d.posTimeStep * nvx, ny = x.y + d.posTimeStep * nvy, nz = x.z + d.posTimeStep * nvz;
  if (nx < d.xmin) nx = d.xmin;
  if (ny < d.ymin) ny = d.ymin;
  if (nz < d.zmin) nz = d.zmin;
  if (nx > d.xmax - 0.000001f) nx = d.xmax - 0.000001f;
  if (ny > d.ymax - 0.000001f) ny = d.ymax - 0.000001f;
  if (nz > d.zmax - 0.000001f) nz = d.zmax - 0.000001f;
  nvx = (v.x + nvx) * 0.5f; nvy = (v.y + nvy) * 0.5f; nvz = (v.z + nvz) * 0.5f; nvw = (v.w + nvw) * 0.5f;

  const NbrTile t(d, id);
  float ncx = 0.f, ncy = 0.f, ncz = 0.f, ncw = 0.f, wsum = 0.f, wsum2 = 0.f;
  const uint32_t bnd = d.bndMask[id];  // boundary neighbours, found by the forces kernel of this step
  if (__any(bnd != 0u)) {              // waves in the bulk of the liquid skip the loop (and its 32 id loads) entirely
    const bool wideRow = NbrTile::wide(t.vec16(0));
#pragma unroll 2
    for (int g = 0; g < 8; g++) {
      const uint2 v = t.vec16(g);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (!((bnd >> (g * 4 + k)) & 1u)) continue;
        const int jb = wideRow ? t.id_wide(g * 4 + k) : t.decode(v, k);
        const float4 pb = d.sortedPos[jb];
        float dist = (nx - pb.x) * (nx - pb.x);
        dist += (ny - pb.y) * (ny - pb.y);
        dist += (nz - pb.z) * (nz - pb.z);
        dist = sqrtf(dist);
        const float w = fmaxf(0.f, (d.r0 - dist) / d.r0);  // Ihmsen 2010 (10)
        const float4 nb = d.sortedVel[jb];                  // wall normal, stored in the velocity slot
        ncx += nb.x * w; ncy += nb.y * w; ncz += nb.z * w; ncw += nb.w * w;
        wsum += w;
        wsum2 += w * (d.r0 - dist);
      }
    }
  }
  float len = ((ncx * ncx + ncy * ncy) + ncz * ncz) + ncw * ncw;  // dot(float4,float4) incl. the .w lane
  if (len != 0.f) {
    len = sqrtf(len);
    nx += ((ncx / len) * wsum2) / wsum;
    ny += ((ncy / len) * wsum2) / wsum;
    nz += ((ncz / len) * wsum2) / wsum;
    const float vn = ncx * nvx + ncy * nvy + ncz * nvz;  // un-normalised n_c_i (sphFluid.cl:878)
    if (vn < 0.f) {
      nvx -= ncx * vn; nvy -= ncy * vn; nvz -= ncz * vn;
      nvx *= 0.99f; nvy *= 0.99f; nvz *= 0.99f; nvw *= 0.99f;
    }
  }
  if (!d.hasElastic) {
    // no elastic matter: the three membrane kernels reduce to `position += 0` for non-boundary particles
    // (sphFluid.cl:1673 with an all-zero scratch half); fold that add in so the bits match (-0 + 0 = +0).
    nx += 0.f; ny += 0.f; nz += 0.f;
  }
  const uint32_t src = d.vals[id];
  d.velOrig[src] = make_float4(nvx, nvy, nvz, nvw);
  d.posOrig[src] = make_float4(nx, ny, nz, x.w);  // (x,y,z,type) in one store
}

__global__ __launch_bounds__(SPH_BLOCK) void k_integrate(SphDev d, int nblocks) {
  const int id = xcd_block(nblocks) * SPH_BLOCK + threadIdx.x;
  if (id >= d.N) return;
  const float4 x = d.sortedPos[id];
  if (TYPE_OF(x) == SPH_BOUNDARY_PARTICLE) return;
  integrate_particle(d, id, x, d.sortedVel[id], d.acc[id], d.accP[id]);
}

int sphk_integrate(sph_solver* s) {
  const int nb = sph_blocks(s->d.N);
  { const int g = sph_guard_position_write(s); if (g != SPH_OK) return g; }
  hipLaunchKernelGGL(k_integrate, dim3(nb), dim3(SPH_BLOCK), 0, s->stream, s->d, nb);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

// ------------------------------------------------------------------ K12 pcisph_computePressureForceAcceleration
// (sphFluid.cl:1101-1212). FUSE: 0 = alone, 1 = + next iteration's predictPositions, 2 = + integrate (last iteration).
// One batch of PF_BATCH neighbours of the pressure-force sum. FAST: square root and the divisions by r through sph_fastmath.h;
// returns false (wave-uniform) if any lane had an operand outside the ranges those sequences are valid in.
template <bool FAST>
__device__ __forceinline__ bool pf_batch(const SphDev& d, const float4 xi, const float pi_, const float hq, const float4 (&xj)[PF_BATCH],
                                         const float2 (&rpj)[PF_BATCH], const int (&jj)[PF_BATCH], float& rx, float& ry, float& rz,
                                         const bool ownOk) {
  bool ok = ownOk;
  // FAST: value, the three numerators and the three quotients carry the factor 2^40 of sph_fastmath.h (the 0.5 of the numerator
  // becomes 2^39; the sums take the quotients through fma(q, 2^-40, sum) — the same single rounding as sum + q / 2^40)
  const float half = FAST ? SPH_FAST_HALF_SCALED : 0.5f;
#pragma unroll
  for (int k = 0; k < PF_BATCH; k++) {
    const float ex_ = xi.x - xj[k].x, ey_ = xi.y - xj[k].y, ez_ = xi.z - xj[k].z;
    const float d2_ = ex_ * ex_ + ey_ * ey_ + ez_ * ez_;
    float sq_;
    if (FAST) ok = sph_sqrt_fast(d2_, d.fastD2Min, d.fastD2Max, &sq_) && ok;  // (the bounds also keep r inside the division's range)
    else sq_ = sqrtf(d2_);
    const float r = sq_ * d.simScale;  // == the stored neighborMap distance (sphFluid.cl:131-136,172), which is therefore not read
    // value = -(hs-r)^2*0.5*(p_i+p_j)/rho*_j, or for very close pairs -(hs/4-r)^2*0.5*(rho0*delta)/rho*_j (:1166-1168):
    // the numerator is selected first, so only one IEEE division is spent
    float num = -(d.hs - r) * (d.hs - r) * half * (pi_ + rpj[k].y);
    if (__any(r < d.closeRf)) num = (r < d.closeRf) ? -(hq - r) * (hq - r) * half * d.rho0delta : num;  // (pairs closer than h/4: rare, so wave-uniformly skipped)
#ifdef PF_IEEE_VALUE  // A/B: the compiler's division for value (13-15 instructions instead of 8)
    const float value = num / rpj[k].x;
#else
    const float value = FAST ? sph_div1_by(num, rpj[k].x) : num / rpj[k].x;
#endif
    const float vx = (xi.x - xj[k].x) * d.simScale, vy = (xi.y - xj[k].y) * d.simScale, vz = (xi.z - xj[k].z) * d.simScale;
    const bool use = jj[k] != -1 && r < d.hs;
    const float ax_ = value * vx, ay_ = value * vy, az_ = value * vz;
    float q_[3];
    // (the division's guard only matters for terms that are used: beyond the support radius, where value is tiny, nothing is added.
    // Guarding every lane and dropping unused terms by scaling them with 0 — one select instead of three — was measured: no change)
    if (FAST) ok = (sph_div3_by<false>(ax_, ay_, az_, value, d.fastValueMin, r, q_) || !use) && ok;
    else { q_[0] = ax_ / r; q_[1] = ay_ / r; q_[2] = az_ / r; }
    if (FAST) {
      rx = use ? __builtin_fmaf(q_[0], SPH_FAST_UNSCALE, rx) : rx;
      ry = use ? __builtin_fmaf(q_[1], SPH_FAST_UNSCALE, ry) : ry;
      rz = use ? __builtin_fmaf(q_[2], SPH_FAST_UNSCALE, rz) : rz;
    } else {
      rx = use ? rx + q_[0] : rx;
      ry = use ? ry + q_[1] : ry;
      rz = use ? rz + q_[2] : rz;
    }
  }
  return !FAST || !__any(!ok);
}

template <int FUSE>
__global__ __launch_bounds__(SPH_BLOCK) void k_pressure_force(SphDev d, int nblocks) {
  int id;
  if (!xcd_range_id(d, id)) return;
  const float4 xi = d.sortedPos[id];
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  if (TYPE_OF(xi) == SPH_BOUNDARY_PARTICLE) {
    if (FUSE == 1) store_pred(d, id, xi);
    else d.accP[id] = zero;  // (FUSE == 1, a middle iteration of the fused step: the next pressure-force kernel overwrites it unread)
    return;
  }
  const float2 rpi = d.rp[id];  // (rho*, p) of this particle
  const float pi_ = rpi.y;
  // precondition of the short division sequence (sph_fastmath.h): own coordinates not within 2^-4 of zero
  const bool ownOk = sph_exp_in(xi.x, SPH_FAST_COORD_EXP_LO, 100) && sph_exp_in(xi.y, SPH_FAST_COORD_EXP_LO, 100) &&
                     sph_exp_in(xi.z, SPH_FAST_COORD_EXP_LO, 100);
  const NbrTile t(d, id);
  float rx = 0.f, ry = 0.f, rz = 0.f;
  const float hq = d.hs * 0.25f;
  // Branch-free, like k_predict_density: map loads first, gathers in batches of PF_BATCH with an always-valid index,
  // masked accumulation (a skipped term leaves the sum untouched, exactly as the reference's `if`).
  // The distances are NOT read: r_ij is recomputed from the two position records this kernel has anyway, with the expression
  // findNeighbors stored it with (sphFluid.cl:131-136,172; IEEE sqrt) — bit-identical, and 128 of the 272 bytes the kernel
  // streamed per particle are gone (1.24 -> 1.17 ms per launch at 16.5 M particles; re-measured in round 3 with the short
  // arithmetic in place: reading the rows instead of taking the square root is 1.26 vs 1.00 ms per launch).
  uint2 v16[8];
#pragma unroll
  for (int g = 0; g < 8; g++) v16[g] = t.vec16(g);
  const bool wideRow = NbrTile::wide(v16[0]);
#pragma unroll
  for (int b = 0; b < 32 / PF_BATCH; b++) {
    int jj[PF_BATCH];
#pragma unroll
    for (int k = 0; k < PF_BATCH; k++) {
      const int slot = b * PF_BATCH + k;
      jj[k] = t.decode(v16[slot >> 2], slot & 3);
    }
    if (wideRow) {  // rare
#pragma unroll
      for (int k = 0; k < PF_BATCH; k++) jj[k] = t.id_wide(b * PF_BATCH + k);
    }
    float4 xj[PF_BATCH];
    float2 rpj[PF_BATCH];
#pragma unroll
    for (int k = 0; k < PF_BATCH; k++) {
      const int jc = NBR_INDEX(jj[k]);
      xj[k] = d.sortedPos[jc];
      rpj[k] = d.rp[jc];
    }
    // The kernel is bound by its arithmetic: the IEEE square root and the three IEEE divisions by r take the short sequences of
    // sph_fastmath.h, bit-identical inside their operand ranges; a batch in which any lane of the wave leaves those ranges is
    // simply computed again with the compiler's sqrtf and `/` (same results for the lanes that were inside).
    const float bx = rx, by = ry, bz = rz;
    if (!pf_batch<true>(d, xi, pi_, hq, xj, rpj, jj, rx, ry, rz, ownOk)) {
      if ((threadIdx.x & 63) == 0) atomicAdd(&d.dbg[8], 1u);  // (wave, batch) pairs recomputed with sqrtf and `/`: rare (tests assert it)
      rx = bx; ry = by; rz = bz;
      pf_batch<false>(d, xi, pi_, hq, xj, rpj, jj, rx, ry, rz, true);
    }
  }
  const float scale = (float)(d.massGradW / (double)rpi.x);
  const float4 ap = make_float4(rx * scale, ry * scale, rz * scale, 0.f);
  if (FUSE != 1) d.accP[id] = ap;
  if (FUSE == 1) store_pred(d, id, predict_position(d, xi, d.sortedVel[id], ap));
  if (FUSE == 2) integrate_particle(d, id, xi, d.sortedVel[id], d.acc[id], ap);
}

static int launch_pressure_force(sph_solver* s, int fuse, const SphDev& d) {
  const int nb = sph_blocks(s->d.N);
  if (fuse == 2) { const int g = sph_guard_position_write(s); if (g != SPH_OK) return g; }  // (+ integrate: writes posOrig)
  if (fuse == 0) hipLaunchKernelGGL((k_pressure_force<0>), dim3(nb), dim3(SPH_BLOCK), 0, s->stream, d, nb);
  else if (fuse == 1) hipLaunchKernelGGL((k_pressure_force<1>), dim3(nb), dim3(SPH_BLOCK), 0, s->stream, d, nb);
  else hipLaunchKernelGGL((k_pressure_force<2>), dim3(nb), dim3(SPH_BLOCK), 0, s->stream, d, nb);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

int sphk_pressure_force(sph_solver* s, int fuse, int ghostDepth) { return launch_pressure_force(s, fuse, sph_ranged(s, ghostDepth)); }

int sphk_pressure_force_layers(sph_solver* s, int fuse, long long loLayer, long long hiLayer) {
  return launch_pressure_force(s, fuse, sph_ranged_layers(s, loLayer, hiLayer));
}
