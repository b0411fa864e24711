/* TEST INFRASTRUCTURE — NOT PRODUCT CODE. See sph_oracle.h for scope, pinning and who may load this.
 *
 * One function per reference stage; each cites the reference lines it restates. Arithmetic keeps the
 * reference's operand types and evaluation order (build with -ffp-contract=off, no fast-math):
 * float terms, double accumulators where sphFluid.cl has them, IEEE sqrtf and division.
 */
#include "sph_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define MAXN 32          /* MAX_NEIGHBOR_COUNT, owOpenCLConstant.h:4 */
#define MAXMEMB 7        /* MAX_MEMBRANES_INCLUDING_SAME_PARTICLE, owOpenCLConstant.h:6 */
#define RSEG 30          /* radius_segments, sphFluid.cl:116 */
#define T_LIQUID 1
#define T_ELASTIC 2
#define T_BOUNDARY 3

typedef struct { float x, y, z, w; } f4;
typedef struct { float x, y; } f2;
typedef struct { uint32_t x, y; } u2;

struct sph_oracle {
  sph_oracle_params p;
  f4 *position, *velocity;             /* 2N, orig order; 2nd half = membrane scratch */
  f4 *sortedPosition;                  /* 2N, sorted; 2nd half = predicted */
  f4 *sortedVelocity;                  /* N */
  f4 *acceleration;                    /* 2N: non-pressure | pressure */
  f2 *neighborMap;                     /* 32N, reference float2 encoding (export view) */
  int32_t *nbrId;                      /* 32N integer ids (authoritative) */
  u2 *particleIndex;                   /* N (+1 in front for the idx-1 read of indexx) */
  uint32_t *particleIndexBack;         /* N */
  uint32_t *gridCellIndex, *gridCellIndexFixedUp; /* G+1 */
  float *pressure, *rho;               /* N, 2N */
  f4 *elastic;                         /* 32*E */
  int32_t *membraneData, *pml;         /* 3M, 7E */
  float *muscle;                       /* muscleCount */
  uint32_t *scratchCount;              /* G+1 counting-sort scratch */
  u2 *scratchPairs;
  double stageSeconds[SO_STAGE_COUNT];
};

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static inline int ptype_sorted(const sph_oracle* s, int id) {
  /* (int)position[ particleIndex[id].y ].w  — sphFluid.cl:615-616 */
  return (int)s->position[s->particleIndex[id].y].w;
}

float sph_oracle_surf_tens_coeff(double Wpoly6Coefficient, float h, float simulationScale) {
  /* sphFluid.cl:662: -1.5e-09f * 0.3f * (float)(Wpoly6Coefficient * pow(hScaled2/2.0,3.0)) * simulationScale */
  float hScaled = h * simulationScale;
  float hScaled2 = hScaled * hScaled;
  return (-1.5e-09f * 0.3f * (float)(Wpoly6Coefficient * pow(hScaled2 / 2.0, 3.0)) * simulationScale);
}

/* ------------------------------------------------------------------ K1 clearBuffers (sphFluid.cl:64-92) */
static void st_clear(sph_oracle* s) {
  const int N = s->p.N;
#pragma omp parallel for schedule(static) num_threads(s->p.threads)
  for (int i = 0; i < N * MAXN; i++) {
    s->neighborMap[i].x = -1.f; s->neighborMap[i].y = -1.f; s->nbrId[i] = -1;
  }
}

/* ------------------------------------------------------------------ K2 hashParticles (sphFluid.cl:187-201,332-383) */
static void st_hash(sph_oracle* s) {
  const sph_oracle_params* p = &s->p;
#pragma omp parallel for schedule(static) num_threads(s->p.threads)
  for (int id = 0; id < p->N; id++) {
    f4 q = s->position[id];
    int cx = (int)(q.x * p->hashGridCellSizeInv);
    int cy = (int)(q.y * p->hashGridCellSizeInv);
    int cz = (int)(q.z * p->hashGridCellSizeInv);
    int cell = cx + cy * p->gridCellsX + cz * p->gridCellsX * p->gridCellsY;
    s->particleIndex[id].x = (uint32_t)cell & p->cellIdMask;
    s->particleIndex[id].y = (uint32_t)id;
  }
}

/* ------------------------------------------------------------------ H1 _runSort (owOpenCLSolver.cpp:255-261,690-696)
 * qsort by cell id only; glibc's qsort is a stable merge sort, so ties keep ascending orig id
 * (SURVEY App. B #4). A stable counting sort gives the identical permutation. Cells that do not fit
 * the table (possible only for positions outside the box) fall back to a stable merge sort. */
static int cmp_cell(const void* a, const void* b) {
  int x = (int)((const u2*)a)->x, y = (int)((const u2*)b)->x;
  return (x < y) ? -1 : (x > y);
}
static void merge_sort_pairs(u2* a, u2* tmp, int n) {
  if (n < 2) return;
  int m = n / 2;
  merge_sort_pairs(a, tmp, m); merge_sort_pairs(a + m, tmp, n - m);
  int i = 0, j = m, k = 0;
  while (i < m && j < n) tmp[k++] = (cmp_cell(&a[j], &a[i]) < 0) ? a[j++] : a[i++];
  while (i < m) tmp[k++] = a[i++];
  while (j < n) tmp[k++] = a[j++];
  memcpy(a, tmp, sizeof(u2) * (size_t)n);
}
static void st_sort(sph_oracle* s) {
  const int N = s->p.N, G = s->p.gridCellCount;
  uint32_t maxc = 0;
  for (int i = 0; i < N; i++) if (s->particleIndex[i].x > maxc) maxc = s->particleIndex[i].x;
  uint32_t bins = (s->p.cellIdMask == 0xffffu) ? 65536u : (uint32_t)G + 1u;
  if (maxc >= bins || (int)maxc < 0) { merge_sort_pairs(s->particleIndex, s->scratchPairs, N); return; }
  uint32_t* cnt = s->scratchCount;  /* sized max(65536, G+1)+1 */
  memset(cnt, 0, sizeof(uint32_t) * ((size_t)bins + 1));
  for (int i = 0; i < N; i++) cnt[s->particleIndex[i].x + 1]++;
  for (uint32_t c = 0; c < bins; c++) cnt[c + 1] += cnt[c];
  for (int i = 0; i < N; i++) s->scratchPairs[cnt[s->particleIndex[i].x]++] = s->particleIndex[i];
  memcpy(s->particleIndex, s->scratchPairs, sizeof(u2) * (size_t)N);
}

/* ------------------------------------------------------------------ K3 sortPostPass (sphFluid.cl:441-466) */
static void st_sortpost(sph_oracle* s) {
  const int N = s->p.N;
#pragma omp parallel for schedule(static) num_threads(s->p.threads)
  for (int id = 0; id < N; id++) {
    u2 spi = s->particleIndex[id];
    f4 q = s->position[spi.y];
    q.w = (float)(int)spi.x;
    s->sortedVelocity[id] = s->velocity[spi.y];
    s->sortedPosition[id] = q;
    s->particleIndexBack[spi.y] = (uint32_t)id;
  }
}

/* ------------------------------------------------------------------ K4 indexx (sphFluid.cl:385-439)
 * first sorted index holding cell c, else -1; [0]=0, [G]=N. (The binary search of the reference finds
 * exactly the first occurrence; a linear sweep over the sorted keys gives the same table.) */
static void st_indexx(sph_oracle* s) {
  const int N = s->p.N, G = s->p.gridCellCount;
  uint32_t* g = s->gridCellIndex;
  for (int c = 0; c <= G; c++) g[c] = 0xffffffffu;
  for (int i = N - 1; i >= 0; i--) {
    uint32_t c = s->particleIndex[i].x;
    if (c < (uint32_t)G && (int)c > 0) g[c] = (uint32_t)i;  /* ids > G are never searched (sphFluid.cl:394) */
  }
  g[0] = 0; g[G] = (uint32_t)N;
}

/* ------------------------------------------------------------------ H2 _runIndexPostPass (owOpenCLSolver.cpp:305-319) */
static void st_indexpost(sph_oracle* s) {
  const int G = s->p.gridCellCount;
  uint32_t* b = s->gridCellIndexFixedUp;
  memcpy(b, s->gridCellIndex, sizeof(uint32_t) * ((size_t)G + 1));
  uint32_t recent = (uint32_t)G;
  for (int i = G; i >= 0; i--) {
    if (b[i] == 0xffffffffu) b[i] = recent; else recent = b[i];
  }
}

/* ------------------------------------------------------------------ K5 findNeighbors (sphFluid.cl:94-184,207-329) */
static inline int search_cell(int cell, int dx, int dy, int dz, int gx, int gy, int G) {
  int c = cell + dx + dy * gx + dz * gx * gy;
  if (c < 0) c += G;
  if (c >= G) c -= G;      /* sequential SELECTs, sphFluid.cl:109-110 */
  return c;
}
static int scan_cell(sph_oracle* s, int cell, f4 me, int myId, int spaceLeft, int mode, int* hist, float r_thr) {
  const sph_oracle_params* p = &s->p;
  int base = (int)s->gridCellIndexFixedUp[cell];
  int count = (int)s->gridCellIndexFixedUp[cell + 1] - base;
  int found = 0;
  float r2thr = r_thr * r_thr;
  if (spaceLeft > 0) {
    for (int i = 0; i < count; i++) {
      int j = base + i;
      if (j == myId) continue;
      f4 o = s->sortedPosition[j];
      float dx = me.x - o.x, dy = me.y - o.y, dz = me.z - o.z;
      float d2 = dx * dx + dy * dy + dz * dz;
      if (d2 <= r2thr) {
        float d = sqrtf(d2);
        int bin = (int)(d * RSEG / p->h);
        if (bin < RSEG) hist[bin]++;
        if (mode) {
          int off = MAXN - spaceLeft + found;
          if (off >= MAXN) break;
          size_t k = (size_t)myId * MAXN + (size_t)off;
          s->nbrId[k] = j;
          s->neighborMap[k].x = (float)j;
          s->neighborMap[k].y = d * p->simulationScale;
          found++;
        }
      }
    }
  }
  return found;
}
static void st_find(sph_oracle* s) {
  const sph_oracle_params* p = &s->p;
  const int G = p->gridCellCount, gx = p->gridCellsX, gy = p->gridCellsY;
#pragma omp parallel for schedule(dynamic, 256) num_threads(s->p.threads)
  for (int id = 0; id < p->N; id++) {
    f4 me = s->sortedPosition[id];
    int myCell = (int)((uint32_t)(int)me.w & p->cellIdMask);
    int hist[RSEG];
    for (int i = 0; i < RSEG; i++) hist[i] = 0;
    float r_thr = p->h;
    int found = 0, sum = 0;
    for (int mode = 0; mode < 2; mode++) {
      float px = me.x - p->xmin, py = me.y - p->ymin, pz = me.z - p->zmin;
      float cfx = (float)(int)(me.x * p->hashGridCellSizeInv) * p->hashGridCellSize;
      float cfy = (float)(int)(me.y * p->hashGridCellSizeInv) * p->hashGridCellSize;
      float cfz = (float)(int)(me.z * p->hashGridCellSizeInv) * p->hashGridCellSize;
      int dx = ((px - cfx) < p->h) ? -1 : 1;
      int dy = ((py - cfy) < p->h) ? -1 : 1;
      int dz = ((pz - cfz) < p->h) ? -1 : 1;
      static const int order[8][3] = {{0,0,0},{1,0,0},{0,1,0},{0,0,1},{1,1,0},{1,0,1},{0,1,1},{1,1,1}};
      for (int k = 0; k < 8; k++) {
        int c = (k == 0) ? myCell
                         : search_cell(myCell, order[k][0] * dx, order[k][1] * dy, order[k][2] * dz, gx, gy, G);
        found += scan_cell(s, c, me, id, MAXN - found, mode, hist, r_thr);
      }
      if (mode == 0) {
        int j = 0;
        while (j < RSEG) {
          sum += hist[j];
          if (sum == MAXN) break;
          if (sum > MAXN) { j--; break; }
          j++;
        }
        r_thr = (float)(j + 1) * p->h / (float)RSEG;
      }
    }
  }
}

/* ------------------------------------------------------------------ K6 pcisph_computeDensity (sphFluid.cl:472-518) */
static void st_density(sph_oracle* s) {
  const sph_oracle_params* p = &s->p;
  const float hs = p->h * p->simulationScale, hs2 = hs * hs, hs6 = hs2 * hs2 * hs2;
  const double mW = ((double)p->mass) * p->Wpoly6Coefficient;
#pragma omp parallel for schedule(static) num_threads(s->p.threads)
  for (int id = 0; id < p->N; id++) {
    double density = 0.0;
    for (int nc = 0; nc < MAXN; nc++) {
      size_t k = (size_t)id * MAXN + nc;
      if (s->nbrId[k] != -1) {
        float r2 = s->neighborMap[k].y;
        r2 *= r2;
        density += (hs2 - r2) * (hs2 - r2) * (hs2 - r2);
      }
    }
    if (density < hs6) density = hs6;
    density *= mW;
    s->rho[id] = (float)density;
  }
}

/* ------------------------------------------------------------------ K7 pcisph_computeForcesAndInitPressure (sphFluid.cl:589-708) */
static void st_forces(sph_oracle* s) {
  const sph_oracle_params* p = &s->p;
  const int N = p->N;
  const float hs = p->h * p->simulationScale;
  const float massMu = (float)(p->mass * p->viscosity);
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma omp parallel for schedule(static) num_threads(s->p.threads)
  for (int id = 0; id < N; id++) {
    if (ptype_sorted(s, id) == T_BOUNDARY) {
      s->acceleration[id] = zero; s->acceleration[N + id] = zero; s->pressure[id] = 0.f;
      continue;
    }
    f4 sum = zero, st = zero;
    const f4 vi = s->sortedVelocity[id], xi = s->sortedPosition[id];
    for (int nc = 0; nc < MAXN; nc++) {
      size_t k = (size_t)id * MAXN + nc;
      int jd = s->nbrId[k];
      if (jd == -1) continue;
      float r = s->neighborMap[k].y;
      if (r < hs) {
        const f4 vj = s->sortedVelocity[jd], xj = s->sortedPosition[jd];
        const float rj = s->rho[jd], w = hs - r;
        sum.x += (vj.x - vi.x) * w / rj; sum.y += (vj.y - vi.y) * w / rj;
        sum.z += (vj.z - vi.z) * w / rj; sum.w += (vj.w - vi.w) * w / rj;
        st.x += p->surfTensCoeff * (xi.x - xj.x); st.y += p->surfTensCoeff * (xi.y - xj.y);
        st.z += p->surfTensCoeff * (xi.z - xj.z);
      }
    }
    const float scale = massMu * (float)(p->del2WviscosityCoefficient / s->rho[id]);
    f4 a;
    a.x = sum.x * scale; a.y = sum.y * scale; a.z = sum.z * scale; a.w = sum.w * scale;
    a.x += p->gravity_x; a.y += p->gravity_y; a.z += p->gravity_z; a.w += 0.0f;
    a.x += st.x; a.y += st.y; a.z += st.z; a.w += 0.f;
    s->acceleration[id] = a;
    s->acceleration[N + id] = zero;
    s->pressure[id] = 0.f;
  }
}

/* ------------------------------------------------------------------ K8 pcisph_computeElasticForces (sphFluid.cl:709-815) */
static void st_elastic(sph_oracle* s) {
  const sph_oracle_params* p = &s->p;
  if (p->numOfElasticP == 0) return;  /* owOpenCLSolver.cpp:422-423 */
  const float kSpring = 600000000.f;
#pragma omp parallel for schedule(static) num_threads(s->p.threads)
  for (int index = 0; index < p->numOfElasticP; index++) {
    int id = (int)s->particleIndexBack[index + p->elasticOffset];
    for (int nc = 0; nc < MAXN; nc++) {
      f4 conn = s->elastic[(size_t)index * MAXN + nc];
      int jd = (int)conn.x;
      if (jd == -1) break;
      jd = (int)s->particleIndexBack[jd];
      f4 xi = s->sortedPosition[id], xj = s->sortedPosition[jd];
      float vx = (xi.x - xj.x) * p->simulationScale, vy = (xi.y - xj.y) * p->simulationScale,
            vz = (xi.z - xj.z) * p->simulationScale, vw = 0.f;
      float r = sqrtf(((vx * vx + vy * vy) + vz * vz) + vw * vw);
      float dr = r - conn.y;
      if (r != 0.f) {
        f4* a = &s->acceleration[id];
        a->x += -(vx / r) * dr * kSpring; a->y += -(vy / r) * dr * kSpring;
        a->z += -(vz / r) * dr * kSpring; a->w += -(vw / r) * dr * kSpring;
        for (int i = 0; i < p->muscleCount; i++) {
          if ((int)conn.z == (i + 1)) {
            if (s->muscle[i] > 0.f) {
              a->x += -(vx / r) * s->muscle[i] * 800.f; a->y += -(vy / r) * s->muscle[i] * 800.f;
              a->z += -(vz / r) * s->muscle[i] * 800.f; a->w += -(vw / r) * s->muscle[i] * 800.f;
            }
          }
        }
      }
    }
  }
}

/* ------------------------------------------------------------------ K9 pcisph_predictPositions (sphFluid.cl:889-979) */
static void st_predictpos(sph_oracle* s) {
  const sph_oracle_params* p = &s->p;
  const int N = p->N;
  const float posTimeStep = p->timeStep * p->simulationScaleInv;
#pragma omp parallel for schedule(static) num_threads(s->p.threads)
  for (int id = 0; id < N; id++) {
    f4 x = s->sortedPosition[id];
    if (ptype_sorted(s, id) == T_BOUNDARY) { s->sortedPosition[N + id] = x; continue; }
    f4 a = s->acceleration[N + id], v = s->sortedVelocity[id], nv, nx;   /* pressure half only (:924) */
    nv.x = v.x + p->timeStep * a.x; nv.y = v.y + p->timeStep * a.y; nv.z = v.z + p->timeStep * a.z; nv.w = v.w + p->timeStep * a.w;
    nx.x = x.x + posTimeStep * nv.x; nx.y = x.y + posTimeStep * nv.y; nx.z = x.z + posTimeStep * nv.z; nx.w = x.w + posTimeStep * nv.w;
    s->sortedPosition[N + id] = nx;
  }
}

/* ------------------------------------------------------------------ K10 pcisph_predictDensity (sphFluid.cl:982-1059) */
static void st_predictdens(sph_oracle* s) {
  const sph_oracle_params* p = &s->p;
  const int N = p->N;
  const float hs = p->h * p->simulationScale, hs2 = hs * hs, hs6 = hs2 * hs2 * hs2;
  const double mW = ((double)p->mass) * p->Wpoly6Coefficient;
#pragma omp parallel for schedule(static) num_threads(s->p.threads)
  for (int id = 0; id < N; id++) {
    double density = 0.0;
    const f4 xi = s->sortedPosition[N + id];
    for (int nc = 0; nc < MAXN; nc++) {
      int jd = s->nbrId[(size_t)id * MAXN + nc];
      if (jd == -1) continue;
      const f4 xj = s->sortedPosition[N + jd];
      float rx = xi.x - xj.x, ry = xi.y - xj.y, rz = xi.z - xj.z;
      float r2 = (rx * rx + ry * ry + rz * rz) * p->simulationScale * p->simulationScale;
      if (r2 < hs2) density += (hs2 - r2) * (hs2 - r2) * (hs2 - r2);
    }
    if (density < hs6) density = hs6;
    density *= mW;
    s->rho[N + id] = (float)density;
  }
}

/* ------------------------------------------------------------------ K11 pcisph_correctPressure (sphFluid.cl:1062-1098) */
static void st_correctp(sph_oracle* s) {
  const sph_oracle_params* p = &s->p;
  const int N = p->N;
#pragma omp parallel for schedule(static) num_threads(s->p.threads)
  for (int id = 0; id < N; id++) {
    float rho_err = s->rho[N + id] - p->rho0;
    float p_corr = rho_err * p->delta;
    if (p_corr < 0) p_corr = 0;
    s->pressure[id] += p_corr;
  }
}

/* ------------------------------------------------------------------ K12 pcisph_computePressureForceAcceleration (sphFluid.cl:1101-1212) */
static void st_pressureforce(sph_oracle* s) {
  const sph_oracle_params* p = &s->p;
  const int N = p->N;
  const float hs = p->h * p->simulationScale;
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma omp parallel for schedule(static) num_threads(s->p.threads)
  for (int id = 0; id < N; id++) {
    if (ptype_sorted(s, id) == T_BOUNDARY) { s->acceleration[N + id] = zero; continue; }
    f4 res = zero;
    const f4 xi = s->sortedPosition[id];
    for (int nc = 0; nc < MAXN; nc++) {
      size_t k = (size_t)id * MAXN + nc;
      int jd = s->nbrId[k];
      if (jd == -1) continue;
      float r = s->neighborMap[k].y;
      if (r < hs) {
        const f4 xj = s->sortedPosition[jd];
        float value = -(hs - r) * (hs - r) * 0.5f * (s->pressure[id] + s->pressure[jd]) / s->rho[N + jd];
        float vx = (xi.x - xj.x) * p->simulationScale, vy = (xi.y - xj.y) * p->simulationScale,
              vz = (xi.z - xj.z) * p->simulationScale;
        if (r < 0.5 * (hs / 2)) {
          value = -(hs * 0.25f - r) * (hs * 0.25f - r) * 0.5f * (p->rho0 * p->delta) / s->rho[N + jd];
        }
        res.x += value * vx / r; res.y += value * vy / r; res.z += value * vz / r; res.w += value * 0.f / r;
      }
    }
    const float scale = (float)(((double)p->mass) * p->gradWspikyCoefficient / ((double)s->rho[N + id]));
    res.x *= scale; res.y *= scale; res.z *= scale; res.w *= scale;
    s->acceleration[N + id] = res;
  }
}

/* ------------------------------------------------------------------ K13 pcisph_integrate + boundary handling (sphFluid.cl:824-887,1684-1808) */
static void st_integrate(sph_oracle* s) {
  const sph_oracle_params* p = &s->p;
  const int N = p->N;
  const float posTimeStep = p->timeStep * p->simulationScaleInv, r0 = p->r0;
  /* Work-items read position[] of boundary particles (never written) and the type of any neighbour
   * (restored by the writer) — SURVEY App. B #24; one 16-byte store of (x,y,z,type) has the serial semantics. */
#pragma omp parallel for schedule(static) num_threads(s->p.threads)
  for (int id = 0; id < N; id++) {
    const int src = (int)s->particleIndex[id].y;
    const float particleType = s->position[src].w;
    if ((int)particleType == T_BOUNDARY) continue;
    const f4 x = s->sortedPosition[id], v = s->sortedVelocity[id];
    const f4 a0 = s->acceleration[id], a1 = s->acceleration[N + id];
    f4 a, nv, nx;
    a.x = a0.x + a1.x; a.y = a0.y + a1.y; a.z = a0.z + a1.z; a.w = 0.f;
    nv.x = v.x + p->timeStep * a.x; nv.y = v.y + p->timeStep * a.y; nv.z = v.z + p->timeStep * a.z; nv.w = v.w + p->timeStep * a.w;
    nx.x = x.x + posTimeStep * nv.x; nx.y = x.y + posTimeStep * nv.y; nx.z = x.z + posTimeStep * nv.z;
    if (nx.x < p->xmin) nx.x = p->xmin;
    if (nx.y < p->ymin) nx.y = p->ymin;
    if (nx.z < p->zmin) nx.z = p->zmin;
    if (nx.x > p->xmax - 0.000001f) nx.x = p->xmax - 0.000001f;
    if (nx.y > p->ymax - 0.000001f) nx.y = p->ymax - 0.000001f;
    if (nx.z > p->zmax - 0.000001f) nx.z = p->zmax - 0.000001f;
    nv.x = (v.x + nv.x) * 0.5f; nv.y = (v.y + nv.y) * 0.5f; nv.z = (v.z + nv.z) * 0.5f; nv.w = (v.w + nv.w) * 0.5f;
    /* computeInteractionWithBoundaryParticles(..., tangVel = true) */
    f4 n = {0.f, 0.f, 0.f, 0.f};
    float wsum = 0.f, wsum2 = 0.f;
    for (int nc = 0; nc < MAXN; nc++) {
      int jb = s->nbrId[(size_t)id * MAXN + nc];
      if (jb == -1) continue;
      int srcb = (int)s->particleIndex[jb].y;
      if ((int)s->position[srcb].w == T_BOUNDARY) {
        const f4 pb = s->position[srcb];
        float d = (nx.x - pb.x) * (nx.x - pb.x);
        d += (nx.y - pb.y) * (nx.y - pb.y);
        d += (nx.z - pb.z) * (nx.z - pb.z);
        d = sqrtf(d);
        float w = fmaxf(0.f, (r0 - d) / r0);
        const f4 nb = s->velocity[srcb];
        n.x += nb.x * w; n.y += nb.y * w; n.z += nb.z * w; n.w += nb.w * w;
        wsum += w;
        wsum2 += w * (r0 - d);
      }
    }
    float len = ((n.x * n.x + n.y * n.y) + n.z * n.z) + n.w * n.w;
    if (len != 0) {
      len = sqrtf(len);
      nx.x += ((n.x / len) * wsum2) / wsum; nx.y += ((n.y / len) * wsum2) / wsum; nx.z += ((n.z / len) * wsum2) / wsum;
      float vn = n.x * nv.x + n.y * nv.y + n.z * nv.z;
      if (vn < 0) {
        nv.x -= n.x * vn; nv.y -= n.y * vn; nv.z -= n.z * vn;
        nv.x = nv.x * 0.99f; nv.y = nv.y * 0.99f; nv.z = nv.z * 0.99f; nv.w = nv.w * 0.99f;
      }
    }
    s->velocity[src] = nv;
    nx.w = particleType;
    s->position[src] = nx;
  }
}

/* ------------------------------------------------------------------ K14 clearMembraneBuffers (sphFluid.cl:1214-1227) */
static void st_clearmemb(sph_oracle* s) {
  const int N = s->p.N;
  memset(s->position + N, 0, sizeof(f4) * (size_t)N);
  memset(s->velocity + N, 0, sizeof(f4) * (size_t)N);
}

/* sphFluid.cl:1229-1247 — only .x/.y/.z take part */
static inline float det3(f4 c1, f4 c2, f4 c3) {
  return c1.x * c2.y * c3.z + c1.y * c2.z * c3.x + c1.z * c2.x * c3.y
       - c1.z * c2.y * c3.x - c1.x * c2.z * c3.y - c1.y * c2.x * c3.z;
}
/* sphFluid.cl:1250-1308. Note b = (0, b_1, b_2, b_3): the determinant reads (.x,.y,.z) = (0,b_1,b_2). */
static f4 project_to_plane(f4 ps, f4 pa, f4 pb, f4 pc) {
  f4 pm = {0.f, 0.f, 0.f, 0.f};
  float b_1 = pa.x * ((pb.y - pa.y) * (pc.z - pa.z) - (pb.z - pa.z) * (pc.y - pa.y))
            + pa.y * ((pb.z - pa.z) * (pc.x - pa.x) - (pb.x - pa.x) * (pc.z - pa.z))
            + pa.z * ((pb.x - pa.x) * (pc.y - pa.y) - (pb.y - pa.y) * (pc.x - pa.x));
  float b_2 = ps.x * (pb.x - pa.x) + ps.y * (pb.y - pa.y) + ps.z * (pb.z - pa.z);
  float b_3 = ps.x * (pc.x - pa.x) + ps.y * (pc.y - pa.y) + ps.z * (pc.z - pa.z);
  f4 a_1 = {(pb.y - pa.y) * (pc.z - pa.z) - (pb.z - pa.z) * (pc.y - pa.y), pb.x - pa.x, pc.x - pa.x, 0.f};
  f4 a_2 = {(pb.z - pa.z) * (pc.x - pa.x) - (pb.x - pa.x) * (pc.z - pa.z), pb.y - pa.y, pc.y - pa.y, 0.f};
  f4 a_3 = {(pb.x - pa.x) * (pc.y - pa.y) - (pb.y - pa.y) * (pc.x - pa.x), pb.z - pa.z, pc.z - pa.z, 0.f};
  f4 b = {0.f, b_1, b_2, b_3};
  float den = det3(a_1, a_2, a_3);
  if (den != 0) {
    pm.x = det3(b, a_2, a_3) / den;
    pm.y = det3(a_1, b, a_3) / den;
    pm.z = det3(a_1, a_2, b) / den;
  } else {
    pm.w = -1;
  }
  return pm;
}

/* ------------------------------------------------------------------ K15 computeInteractionWithMembranes (sphFluid.cl:1369-1650) */
static void st_memb(sph_oracle* s) {
  const sph_oracle_params* p = &s->p;
  const int N = p->N;
  if (!s->pml || !s->membraneData) return;  /* no membrane lists: the reference has no buffers to run this kernel on */
  const float r0 = p->r0;
#pragma omp parallel for schedule(static) num_threads(s->p.threads)
  for (int id = 0; id < N; id++) {
    const int src = (int)s->particleIndex[id].y;
    if ((int)s->position[src].w != T_LIQUID) continue;
    f4 nvec[MAXN];
    float dist[MAXN];
    int jc = 0, aborted = 0;
    for (int i = 0; i < MAXN; i++) { nvec[i].x = nvec[i].y = nvec[i].z = nvec[i].w = 0.f; }
    const f4 me = s->position[src];
    for (int nc = 0; nc < MAXN && !aborted; nc++) {
      int jd = s->nbrId[(size_t)id * MAXN + nc];
      if (jd == -1) break;
      int jsrc = (int)s->particleIndex[jd].y;
      if ((int)s->position[jsrc].w != T_ELASTIC) continue;
      int ijk = 0;
      const f4 pj = s->position[jsrc];
      float vx = me.x - pj.x, vy = me.y - pj.y, vz = 0.f, vw = me.w - pj.w;   /* `.z = 0` slip, :1437 */
      float d = sqrtf(((vx * vx + vy * vy) + vz * vz) + vw * vw);
      for (int mli = 0; mli < MAXMEMB; mli++) {
        int mdi = s->pml[(size_t)jsrc * MAXMEMB + mli];
        if (!(mdi > -1)) break;
        f4 pi_ = s->position[s->membraneData[mdi * 3 + 0]];
        f4 pj_ = s->position[s->membraneData[mdi * 3 + 1]];
        f4 pk_ = s->position[s->membraneData[mdi * 3 + 2]];
        f4 pp = project_to_plane(me, pi_, pj_, pk_);
        if (pp.w == -1) { aborted = 1; break; }
        f4 nrm = {me.x - pp.x, me.y - pp.y, me.z - pp.z, me.w - pp.w};
        float len = sqrtf(nrm.x * nrm.x + nrm.y * nrm.y + nrm.z * nrm.z);
        if (len > 0) {
          nvec[jc].x += nrm.x / len; nvec[jc].y += nrm.y / len; nvec[jc].z += nrm.z / len; nvec[jc].w += nrm.w / len;
          ijk++;
        } else { aborted = 1; break; }
      }
      if (aborted) break;
      if (ijk > 0) {
        nvec[jc].x /= (float)ijk; nvec[jc].y /= (float)ijk; nvec[jc].z /= (float)ijk; nvec[jc].w /= (float)ijk;
        dist[jc] = d;
        jc++;
      }
    }
    if (aborted || jc == 0) continue;
    f4 n = {0.f, 0.f, 0.f, 0.f};
    float wsum = 0.f, wsum2 = 0.f;
    for (int nc = 0; nc < jc; nc++) {
      float w = fmaxf(0.f, (r0 - dist[nc]) / r0);
      n.x += nvec[nc].x * w; n.y += nvec[nc].y * w; n.z += nvec[nc].z * w; n.w += nvec[nc].w * w;
      wsum += w;
      wsum2 += w * (r0 - dist[nc]);
    }
    n.w = 0;
    float len = ((n.x * n.x + n.y * n.y) + n.z * n.z) + n.w * n.w;
    if (len != 0) {
      len = sqrtf(len);
      f4* o = &s->position[N + src];
      o->x += 1.0f * ((n.x / len) * wsum2) / wsum;
      o->y += 1.0f * ((n.y / len) * wsum2) / wsum;
      o->z += 1.0f * ((n.z / len) * wsum2) / wsum;
    }
  }
}

/* ------------------------------------------------------------------ K16 ..._finalize (sphFluid.cl:1652-1682) */
static void st_membfin(sph_oracle* s) {
  const int N = s->p.N;
#pragma omp parallel for schedule(static) num_threads(s->p.threads)
  for (int src = 0; src < N; src++) {   /* id -> src is a bijection; iterate src directly */
    if ((int)s->position[src].w == T_BOUNDARY) continue;
    f4* q = &s->position[src];
    const f4 d = s->position[N + src];
    q->x += d.x; q->y += d.y; q->z += d.z; q->w += d.w;
  }
}

/* ================================================================== driver */
typedef void (*stage_fn)(sph_oracle*);
static const stage_fn STAGES[SO_STAGE_COUNT] = {
  st_clear, st_hash, st_sort, st_sortpost, st_indexx, st_indexpost, st_find, st_density, st_forces, st_elastic,
  st_predictpos, st_predictdens, st_correctp, st_pressureforce, st_integrate, st_clearmemb, st_memb, st_membfin};

int sph_oracle_run(sph_oracle* s, int stage) {
  if (stage < 0 || stage >= SO_STAGE_COUNT) return -1;
  double t0 = now_s();
  STAGES[stage](s);
  s->stageSeconds[stage] += now_s() - t0;
  return 0;
}

/* owPhysicsFluidSimulator.cpp:79-149 */
int sph_oracle_step(sph_oracle* s) {
  static const int pre[] = {SO_CLEAR, SO_HASH, SO_SORT, SO_SORTPOST, SO_INDEXX, SO_INDEXPOST, SO_FIND,
                            SO_DENSITY, SO_FORCES, SO_ELASTIC};
  for (size_t i = 0; i < sizeof(pre) / sizeof(pre[0]); i++) sph_oracle_run(s, pre[i]);
  int iter = 0;
  do {
    sph_oracle_run(s, SO_PREDICTPOS); sph_oracle_run(s, SO_PREDICTDENS);
    sph_oracle_run(s, SO_CORRECTP); sph_oracle_run(s, SO_PRESSUREFORCE);
    iter++;
  } while (iter < s->p.maxIteration);
  sph_oracle_run(s, SO_INTEGRATE);
  sph_oracle_run(s, SO_CLEARMEMB); sph_oracle_run(s, SO_MEMB); sph_oracle_run(s, SO_MEMBFIN);
  return 0;
}

void sph_oracle_update_muscles(sph_oracle* s, const float* signal) {
  memcpy(s->muscle, signal, sizeof(float) * (size_t)s->p.muscleCount);
}

void sph_oracle_stage_seconds(sph_oracle* s, double* out) {
  memcpy(out, s->stageSeconds, sizeof(s->stageSeconds));
}

static void* zalloc(size_t n) { void* q = calloc(n ? n : 1, 1); if (!q) { fprintf(stderr, "sph_oracle: out of memory\n"); abort(); } return q; }

sph_oracle* sph_oracle_create(const sph_oracle_params* p, const float* position, const float* velocity,
                              const float* elasticConnections, const int32_t* membraneData,
                              const int32_t* particleMembranesList) {
  sph_oracle* s = (sph_oracle*)zalloc(sizeof(*s));
  s->p = *p;
  if (s->p.threads < 1) s->p.threads = 1;
  const size_t N = (size_t)p->N, G = (size_t)p->gridCellCount;
  s->position = zalloc(sizeof(f4) * 2 * N); s->velocity = zalloc(sizeof(f4) * 2 * N);
  memcpy(s->position, position, sizeof(f4) * N); memcpy(s->velocity, velocity, sizeof(f4) * N);
  s->sortedPosition = zalloc(sizeof(f4) * 2 * N); s->sortedVelocity = zalloc(sizeof(f4) * N);
  s->acceleration = zalloc(sizeof(f4) * 2 * N);
  s->neighborMap = zalloc(sizeof(f2) * MAXN * N); s->nbrId = zalloc(sizeof(int32_t) * MAXN * N);
  s->particleIndex = (u2*)zalloc(sizeof(u2) * (N + 1)) + 1;
  s->particleIndexBack = zalloc(sizeof(uint32_t) * N);
  s->gridCellIndex = zalloc(sizeof(uint32_t) * (G + 1)); s->gridCellIndexFixedUp = zalloc(sizeof(uint32_t) * (G + 1));
  s->pressure = zalloc(sizeof(float) * N); s->rho = zalloc(sizeof(float) * 2 * N);
  s->muscle = zalloc(sizeof(float) * (size_t)(p->muscleCount > 0 ? p->muscleCount : 1));
  size_t bins = (G + 1 > 65536 ? G + 1 : 65536) + 2;
  s->scratchCount = zalloc(sizeof(uint32_t) * bins); s->scratchPairs = zalloc(sizeof(u2) * N);
  if (p->numOfElasticP > 0 && elasticConnections) {
    s->elastic = zalloc(sizeof(f4) * MAXN * (size_t)p->numOfElasticP);
    memcpy(s->elastic, elasticConnections, sizeof(f4) * MAXN * (size_t)p->numOfElasticP);
  }
  if (p->numOfMembranes > 0 && membraneData) {
    s->membraneData = zalloc(sizeof(int32_t) * 3 * (size_t)p->numOfMembranes);
    memcpy(s->membraneData, membraneData, sizeof(int32_t) * 3 * (size_t)p->numOfMembranes);
  }
  if (p->numOfElasticP > 0 && particleMembranesList) {
    s->pml = zalloc(sizeof(int32_t) * MAXMEMB * (size_t)p->numOfElasticP);
    memcpy(s->pml, particleMembranesList, sizeof(int32_t) * MAXMEMB * (size_t)p->numOfElasticP);
  }
  return s;
}

void sph_oracle_destroy(sph_oracle* s) {
  if (!s) return;
  free(s->position); free(s->velocity); free(s->sortedPosition); free(s->sortedVelocity); free(s->acceleration);
  free(s->neighborMap); free(s->nbrId); free(s->particleIndex - 1); free(s->particleIndexBack);
  free(s->gridCellIndex); free(s->gridCellIndexFixedUp); free(s->pressure); free(s->rho); free(s->muscle);
  free(s->scratchCount); free(s->scratchPairs); free(s->elastic); free(s->membraneData); free(s->pml);
  free(s);
}

size_t sph_oracle_buffer(sph_oracle* s, const char* name, void** ptr) {
  const size_t N = (size_t)s->p.N, G = (size_t)s->p.gridCellCount;
#define BUF(n, v, bytes) if (!strcmp(name, n)) { *ptr = (void*)(v); return (bytes); }
  BUF("position", s->position, sizeof(f4) * 2 * N) BUF("velocity", s->velocity, sizeof(f4) * 2 * N)
  BUF("sortedPosition", s->sortedPosition, sizeof(f4) * 2 * N) BUF("sortedVelocity", s->sortedVelocity, sizeof(f4) * N)
  BUF("acceleration", s->acceleration, sizeof(f4) * 2 * N) BUF("neighborMap", s->neighborMap, sizeof(f2) * MAXN * N)
  BUF("neighborIds", s->nbrId, sizeof(int32_t) * MAXN * N) BUF("particleIndex", s->particleIndex, sizeof(u2) * N)
  BUF("particleIndexBack", s->particleIndexBack, sizeof(uint32_t) * N)
  BUF("gridCellIndex", s->gridCellIndex, sizeof(uint32_t) * (G + 1))
  BUF("gridCellIndexFixedUp", s->gridCellIndexFixedUp, sizeof(uint32_t) * (G + 1))
  BUF("pressure", s->pressure, sizeof(float) * N) BUF("rho", s->rho, sizeof(float) * 2 * N)
#undef BUF
  *ptr = NULL;
  return 0;
}
