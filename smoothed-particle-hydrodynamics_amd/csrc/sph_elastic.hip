// Elastic matter: K8 pcisph_computeElasticForces (sphFluid.cl:709-815) and the membrane kernels K14-K16
// (sphFluid.cl:1214-1682). These touch only the elastic / near-membrane particles (a few % of a scene), so they are
// straight one-lane-per-particle kernels; quirks are reproduced on purpose.
#include "sph_common.h"

// ------------------------------------------------------------------ K8
__global__ __launch_bounds__(SPH_BLOCK) void k_elastic(SphDev d) {
  const int index = blockIdx.x * SPH_BLOCK + threadIdx.x;  // index among the elastic particles, not a particle id
  if (index >= d.numElastic) return;
  const int id = (int)d.backIndex[index + d.elasticOffset];
  const float4 xi = d.sortedPos[id];
  float4 a = d.acc[id];
  const float kSpring = 600000000.f;
  for (int nc = 0; nc < SPH_MAXN; nc++) {
    const float4 conn = d.elastic[(size_t)index * SPH_MAXN + nc];  // (j + 0.1, r0_ij, muscle id . colour, 0)
    int jd = (int)conn.x;
    if (jd == -1) break;  // first NO_PARTICLE_ID ends the list (sphFluid.cl:802-803)
    jd = (int)d.backIndex[jd];
    const float4 xj = d.sortedPos[jd];
    const float vx = (xi.x - xj.x) * d.simScale, vy = (xi.y - xj.y) * d.simScale, vz = (xi.z - xj.z) * d.simScale;
    const float r = sqrtf(((vx * vx + vy * vy) + vz * vz) + 0.f * 0.f);  // dot(float4,float4) with .w = 0
    const float dr = r - conn.y;
    if (r != 0.f) {
      a.x += -(vx / r) * dr * kSpring;
      a.y += -(vy / r) * dr * kSpring;
      a.z += -(vz / r) * dr * kSpring;
      const int m = (int)conn.z;  // the reference scans i = 0..MUSCLE_COUNT-1 for (int)conn.z == i+1 (:777-784)
      if (m >= 1 && m <= d.muscleCount) {
        const float sig = d.muscle[m - 1];
        if (sig > 0.f) {
          a.x += -(vx / r) * sig * 800.f;
          a.y += -(vy / r) * sig * 800.f;
          a.z += -(vz / r) * sig * 800.f;
        }
      }
    }
  }
  d.acc[id] = a;
}

int sphk_elastic(sph_solver* s) {
  if (s->d.numElastic == 0) return SPH_OK;  // owOpenCLSolver.cpp:422-423
  hipLaunchKernelGGL(k_elastic, dim3(sph_blocks(s->d.numElastic)), dim3(SPH_BLOCK), 0, s->stream, s->d);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

// ------------------------------------------------------------------ K14
int sphk_clear_membranes(sph_solver* s) {
  if (!s->d.membDelta) return SPH_OK;  // no elastic matter: the scratch half does not exist (folded into integrate)
  SPH_HIP(hipMemsetAsync(s->d.membDelta, 0, sizeof(float4) * (size_t)s->d.N, s->stream));
  return SPH_OK;
}

// ------------------------------------------------------------------ K15
struct f3 { float x, y, z; };
__device__ __forceinline__ float det3(f3 c1, f3 c2, f3 c3) {  // sphFluid.cl:1229-1247
  return c1.x * c2.y * c3.z + c1.y * c2.z * c3.x + c1.z * c2.x * c3.y - c1.z * c2.y * c3.x - c1.x * c2.z * c3.y -
         c1.y * c2.x * c3.z;
}
// sphFluid.cl:1250-1308. `b = (0, b_1, b_2, b_3)` there, and the determinant reads .x/.y/.z = (0, b_1, b_2).
__device__ __forceinline__ bool project_to_plane(const float4 ps, const float4 pa, const float4 pb, const float4 pc, f3* pm) {
  const float b_1 = pa.x * ((pb.y - pa.y) * (pc.z - pa.z) - (pb.z - pa.z) * (pc.y - pa.y)) +
                    pa.y * ((pb.z - pa.z) * (pc.x - pa.x) - (pb.x - pa.x) * (pc.z - pa.z)) +
                    pa.z * ((pb.x - pa.x) * (pc.y - pa.y) - (pb.y - pa.y) * (pc.x - pa.x));
  const float b_2 = ps.x * (pb.x - pa.x) + ps.y * (pb.y - pa.y) + ps.z * (pb.z - pa.z);
  const f3 a_1 = {(pb.y - pa.y) * (pc.z - pa.z) - (pb.z - pa.z) * (pc.y - pa.y), pb.x - pa.x, pc.x - pa.x};
  const f3 a_2 = {(pb.z - pa.z) * (pc.x - pa.x) - (pb.x - pa.x) * (pc.z - pa.z), pb.y - pa.y, pc.y - pa.y};
  const f3 a_3 = {(pb.x - pa.x) * (pc.y - pa.y) - (pb.y - pa.y) * (pc.x - pa.x), pb.z - pa.z, pc.z - pa.z};
  const f3 b = {0.f, b_1, b_2};
  const float den = det3(a_1, a_2, a_3);
  if (den != 0.f) {
    pm->x = det3(b, a_2, a_3) / den;
    pm->y = det3(a_1, b, a_3) / den;
    pm->z = det3(a_1, a_2, b) / den;
    return true;
  }
  return false;  // the reference prints and abandons this particle (:1296-1298,1468-1472)
}

// Two kernels. k_membrane_collect queues the liquid particles that have an elastic neighbour (bit mask from the forces
// kernel); k_membranes then gives every queued particle HALF A WAVE: lane = neighbour slot, so the expensive part — up to
// 7 triangle projections per elastic neighbour — runs in parallel over the slots instead of serially in one lane (the
// one-lane-per-particle form left 63 lanes waiting for the one next to a membrane: 0.27 ms of the worm scene's 0.81 ms).
// The per-slot results are then combined by one lane in slot order, i.e. with the reference's summation order.
__global__ __launch_bounds__(SPH_BLOCK) void k_membrane_collect(SphDev d, uint32_t* __restrict__ queue, uint32_t* __restrict__ count) {
  __shared__ uint32_t local, base;
  if (threadIdx.x == 0) local = 0u;
  __syncthreads();
  const int id = blockIdx.x * SPH_BLOCK + threadIdx.x;
  uint32_t slot = 0xffffffffu;
  if (id < d.N && (int)d.sortedPos[id].w == SPH_LIQUID_PARTICLE && d.elasticMask[id] != 0u) slot = atomicAdd(&local, 1u);
  __syncthreads();
  if (threadIdx.x == 0) base = local ? atomicAdd(count, local) : 0u;
  __syncthreads();
  if (slot != 0xffffffffu) queue[base + slot] = (uint32_t)id;
}

__global__ __launch_bounds__(SPH_BLOCK) void k_membranes(SphDev d, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count) {
  __shared__ float sm[SPH_BLOCK / 32][32][4];  // per half-wave: (mx, my, mz, dist) of each slot
  const int lane = threadIdx.x & 31, group = threadIdx.x >> 5;            // 32 lanes = the 32 neighbour slots of one particle
  const uint32_t total = *count;
  const uint32_t groups = gridDim.x * (SPH_BLOCK / 32);
  for (uint32_t q = blockIdx.x * (SPH_BLOCK / 32) + group; q < ((total + 1u) & ~1u); q += groups) {  // both halves of a wave iterate together
    const bool live = q < total;
    const int id = live ? (int)queue[q] : 0;
    const uint32_t ela = live ? d.elasticMask[id] : 0u;
    const uint32_t src = d.vals[id];
    const float4 me = d.posOrig[src];  // the just-integrated position (:1436,1466)
    bool active = live && ((ela >> lane) & 1u);
    bool failed = false;  // projection onto a degenerate triangle / zero-length normal: the reference abandons the particle
    float mx = 0.f, my = 0.f, mz = 0.f, dist = 0.f;
    int ijk = 0;
    if (active) {
      const int jd = nbr_decode(d.nbr16, d.nbrBase, d.nbrId, id, lane);
      const uint32_t jsrc = d.vals[jd];
      const float4 pj = d.posOrig[jsrc];
      const float vx = me.x - pj.x, vy = me.y - pj.y, vw = me.w - pj.w;  // .z zeroed, .w kept (:1436-1438)
      dist = sqrtf(((vx * vx + vy * vy) + 0.f * 0.f) + vw * vw);
      for (int mli = 0; mli < SPH_MAX_MEMBRANES_INCLUDING_SAME_PARTICLE; mli++) {
        const int mdi = d.pml[(size_t)jsrc * SPH_MAX_MEMBRANES_INCLUDING_SAME_PARTICLE + mli];
        if (!(mdi > -1)) break;
        const float4 pi_ = d.posOrig[d.membraneData[mdi * 3 + 0]];
        const float4 pj_ = d.posOrig[d.membraneData[mdi * 3 + 1]];
        const float4 pk_ = d.posOrig[d.membraneData[mdi * 3 + 2]];
        f3 pp;
        if (!project_to_plane(me, pi_, pj_, pk_, &pp)) { failed = true; break; }
        const float nx = me.x - pp.x, ny = me.y - pp.y, nz = me.z - pp.z;
        const float len = sqrtf(nx * nx + ny * ny + nz * nz);
        if (!(len > 0.f)) { failed = true; break; }  // "error #001" path (:1501-1505)
        mx += nx / len; my += ny / len; mz += nz / len;
        ijk++;
      }
      if (ijk > 0) { mx /= (float)ijk; my /= (float)ijk; mz /= (float)ijk; }
    }
    // The reference walks the slots in order and returns at the FIRST failure, before anything is written; a failure in
    // any slot therefore abandons the particle (later slots cannot undo it, earlier ones have written nothing yet).
    const unsigned long long failMask = __ballot(failed);
    const unsigned long long halfMask = 0xffffffffull << (group & 1 ? 32 : 0);
    sm[group][lane][0] = mx; sm[group][lane][1] = my; sm[group][lane][2] = mz; sm[group][lane][3] = dist;
    const unsigned long long useMask = __ballot(active && ijk > 0);
    __builtin_amdgcn_wave_barrier();  // LDS writes above are read by lane 0 of the half below (same wave: in order)
    if (lane == 0 && live && !(failMask & halfMask)) {
      const uint32_t use = (uint32_t)((useMask & halfMask) >> (group & 1 ? 32 : 0));
      float ncx = 0.f, ncy = 0.f, ncz = 0.f, wsum = 0.f, wsum2 = 0.f;
      for (int nc = 0; nc < SPH_MAXN; nc++) {  // second loop of the reference (:1578-1591): slot order
        if (!((use >> nc) & 1u)) continue;
        const float dd = sm[group][nc][3];
        const float w = fmaxf(0.f, (d.r0 - dd) / d.r0);
        ncx += sm[group][nc][0] * w; ncy += sm[group][nc][1] * w; ncz += sm[group][nc][2] * w;
        wsum += w;
        wsum2 += w * (d.r0 - dd);
      }
      if (use) {
        float len = ((ncx * ncx + ncy * ncy) + ncz * ncz) + 0.f * 0.f;
        if (len != 0.f) {
          len = sqrtf(len);
          float4 o = d.membDelta[src];
          o.x += 1.0f * ((ncx / len) * wsum2) / wsum;
          o.y += 1.0f * ((ncy / len) * wsum2) / wsum;
          o.z += 1.0f * ((ncz / len) * wsum2) / wsum;
          d.membDelta[src] = o;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

int sphk_membranes(sph_solver* s) {
  if (!s->d.hasElastic) return SPH_OK;  // no elastic neighbours can exist: the kernel would touch nothing
  // queue: keysAlt is idle after the sort; counter: dbg[5]
  uint32_t* queue = s->d.keysAlt;
  uint32_t* count = s->d.dbg + 5;
  SPH_HIP(hipMemsetAsync(count, 0, sizeof(uint32_t), s->stream));
  hipLaunchKernelGGL(k_membrane_collect, dim3(sph_blocks(s->d.N)), dim3(SPH_BLOCK), 0, s->stream, s->d, queue, count);
  hipLaunchKernelGGL(k_membranes, dim3(min(sph_blocks(s->d.N, SPH_BLOCK / 32), 2048)), dim3(SPH_BLOCK), 0, s->stream, s->d, queue, count);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}

// ------------------------------------------------------------------ K16
__global__ __launch_bounds__(SPH_BLOCK) void k_membranes_finalize(SphDev d) {
  const int src = blockIdx.x * SPH_BLOCK + threadIdx.x;  // id -> src is a bijection; walk orig order for coalescing
  if (src >= d.N) return;
  float4 p = d.posOrig[src];
  if ((int)p.w == SPH_BOUNDARY_PARTICLE) return;
  const float4 dl = d.membDelta[src];
  p.x += dl.x; p.y += dl.y; p.z += dl.z; p.w += dl.w;
  d.posOrig[src] = p;
}

int sphk_membranes_finalize(sph_solver* s) {
  if (!s->d.hasElastic) return SPH_OK;  // already folded into integrate (`+ 0.f`)
  { const int g = sph_guard_position_write(s); if (g != SPH_OK) return g; }
  hipLaunchKernelGGL(k_membranes_finalize, dim3(sph_blocks(s->d.N)), dim3(SPH_BLOCK), 0, s->stream, s->d);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}
