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

__global__ __launch_bounds__(SPH_BLOCK) void k_membranes(SphDev d) {
  const int id = blockIdx.x * SPH_BLOCK + threadIdx.x;
  if (id >= d.N) return;
  if ((int)d.sortedPos[id].w != SPH_LIQUID_PARTICLE) return;  // only liquid particles are displaced (:1395)
  const uint32_t src = d.vals[id];
  const float4 me = d.posOrig[src];  // the just-integrated position (:1436,1466)
  float ncx = 0.f, ncy = 0.f, ncz = 0.f, wsum = 0.f, wsum2 = 0.f;
  int jc = 0;
  for (int nc = 0; nc < SPH_MAXN; nc++) {
    const int jd = d.nbrId[nbr_index(id, nc)];
    if (jd == -1) break;  // stops at the first empty slot (:1557)
    if ((int)d.sortedPos[jd].w != SPH_ELASTIC_PARTICLE) continue;
    const uint32_t jsrc = d.vals[jd];
    const float4 pj = d.posOrig[jsrc];
    const float vx = me.x - pj.x, vy = me.y - pj.y, vw = me.w - pj.w;  // .z zeroed, .w kept (:1436-1438)
    const float dist = sqrtf(((vx * vx + vy * vy) + 0.f * 0.f) + vw * vw);
    float mx = 0.f, my = 0.f, mz = 0.f;
    int ijk = 0;
    for (int mli = 0; mli < SPH_MAX_MEMBRANES_INCLUDING_SAME_PARTICLE; mli++) {
      const int mdi = d.pml[(size_t)jsrc * SPH_MAX_MEMBRANES_INCLUDING_SAME_PARTICLE + mli];
      if (!(mdi > -1)) break;
      const float4 pi_ = d.posOrig[d.membraneData[mdi * 3 + 0]];
      const float4 pj_ = d.posOrig[d.membraneData[mdi * 3 + 1]];
      const float4 pk_ = d.posOrig[d.membraneData[mdi * 3 + 2]];
      f3 pp;
      if (!project_to_plane(me, pi_, pj_, pk_, &pp)) return;
      const float nx = me.x - pp.x, ny = me.y - pp.y, nz = me.z - pp.z;
      const float len = sqrtf(nx * nx + ny * ny + nz * nz);
      if (!(len > 0.f)) return;  // "error #001" path (:1501-1505)
      mx += nx / len; my += ny / len; mz += nz / len;
      ijk++;
    }
    if (ijk > 0) {
      mx /= (float)ijk; my /= (float)ijk; mz /= (float)ijk;
      // second loop of the reference (:1578-1591), applied as each membrane neighbour is finished: same order, same sums
      const float w = fmaxf(0.f, (d.r0 - dist) / d.r0);
      ncx += mx * w; ncy += my * w; ncz += mz * w;
      wsum += w;
      wsum2 += w * (d.r0 - dist);
      jc++;
    }
  }
  if (jc == 0) return;
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

int sphk_membranes(sph_solver* s) {
  if (!s->d.hasElastic) return SPH_OK;  // no elastic neighbours can exist: the kernel would touch nothing
  hipLaunchKernelGGL(k_membranes, dim3(sph_blocks(s->d.N)), dim3(SPH_BLOCK), 0, s->stream, s->d);
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
  hipLaunchKernelGGL(k_membranes_finalize, dim3(sph_blocks(s->d.N)), dim3(SPH_BLOCK), 0, s->stream, s->d);
  SPH_HIP(hipGetLastError());
  return SPH_OK;
}
