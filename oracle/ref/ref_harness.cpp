// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// Host harness around the *reference's own* OpenCL kernels (/root/reference/src/sphFluid.cl),
// which oracle/ref/Makefile compiles unmodified to an x86-64 object with ROCm clang (-x cl).
// This file supplies (a) the 7 OpenCL builtins that object leaves undefined and (b) the ~40
// lines of host logic the reference keeps in owOpenCLSolver.cpp / owPhysicsFluidSimulator.cpp
// (stage order, qsort + myCompare, cell-index fix-up, calcDelta), restated because those files
// need an OpenCL CPU device / Python 2.7 and cannot be built here.
//
// Output: oracle/_ref/libsphref.so (git-ignored). It exists only in the build container (the
// reference sources do not travel); tests use it to pin oracle/sph_oracle.c and to generate the
// committed fixtures under tests/golden/. Nothing in the product path links or loads it.
//
// Conventions that OpenCL leaves implementation-defined and that we fix here:
//   dot(float4,float4) = ((x*x + y*y) + z*z) + w*w ; native_sqrt = sqrt = correctly rounded sqrtf ;
//   pow(double,double) = libm pow ; select(a,b,c) = c ? b : a ; max(float,float) = fmaxf.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>

#include "owPhysicsConstant.h"  // reference constants, compiled where they lie (-I/root/reference/src)

typedef float float4v __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- OpenCL builtins
static thread_local size_t g_gid = 0;
size_t cl_get_global_id(unsigned) __asm__("_Z13get_global_idj");
size_t cl_get_global_id(unsigned) { return g_gid; }
float cl_native_sqrt(float) __asm__("_Z11native_sqrtf");
float cl_native_sqrt(float x) { return sqrtf(x); }
float cl_sqrt(float) __asm__("_Z4sqrtf");
float cl_sqrt(float x) { return sqrtf(x); }
float cl_dot4(float4v, float4v) __asm__("_Z3dotDv4_fS_");
float cl_dot4(float4v a, float4v b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
float cl_maxf(float, float) __asm__("_Z3maxff");
float cl_maxf(float a, float b) { return fmaxf(a, b); }
double cl_pow(double, double) __asm__("_Z3powdd");
double cl_pow(double a, double b) { return pow(a, b); }
int cl_select(int, int, unsigned) __asm__("_Z6selectiij");
int cl_select(int a, int b, unsigned c) { return c ? b : a; }

// ---------------------------------------------------------------- kernel prototypes (sphFluid.cl)
struct f2 { float x, y; };
struct f4 { float x, y, z, w; };
struct u2 { unsigned x, y; };
extern "C" {
void clearBuffers(f2* neighborMap, int N);
void hashParticles(f4* position, int gx, int gy, int gz, float cellInv, float xmin, float ymin, float zmin,
                   u2* particleIndex, int N);
void indexx(u2* particleIndex, int gridCellCount, unsigned* gridCellIndex, int N);
void sortPostPass(u2* particleIndex, unsigned* particleIndexBack, f4* position, f4* velocity, f4* sortedPosition,
                  f4* sortedVelocity, int N);
void findNeighbors(unsigned* gridCellIndexFixedUp, f4* sortedPosition, int gridCellCount, int gx, int gy, int gz,
                   float h, float cellSize, float cellInv, float simScale, float xmin, float ymin, float zmin,
                   f2* neighborMap, int N);
void pcisph_computeDensity(f2* neighborMap, double Wpoly6, float h, float mass, float rho0, float simScale,
                           float stiffness, f4* sortedPosition, float* pressure, float* rho,
                           unsigned* particleIndexBack, float delta, int N);
void pcisph_computeForcesAndInitPressure(f2* neighborMap, float* rho, float* pressure, f4* sortedPosition,
                                         f4* sortedVelocity, f4* acceleration, unsigned* particleIndexBack,
                                         double Wpoly6, double del2Wvisc, float h, float mass, float mu,
                                         float simScale, float gx_, float gy_, float gz_, f4* position,
                                         u2* particleIndex, int N);
void pcisph_computeElasticForces(f2* neighborMap, f4* sortedPosition, f4* sortedVelocity, f4* acceleration,
                                 unsigned* particleIndexBack, f4* velocity, float h, float mass, float simScale,
                                 int numOfElasticP, f4* elasticConnectionsData, int offset, int N, int MUSCLE_COUNT,
                                 float* muscle_activation_signal, f4* position);
void pcisph_predictPositions(f4* acceleration, f4* sortedPosition, f4* sortedVelocity, u2* particleIndex,
                             unsigned* particleIndexBack, float gx_, float gy_, float gz_, float simScaleInv,
                             float timeStep, float xmin, float xmax, float ymin, float ymax, float zmin, float zmax,
                             float damping, f4* position, f4* velocity, float r0, f2* neighborMap, int N);
void pcisph_predictDensity(f2* neighborMap, unsigned* particleIndexBack, double Wpoly6, float h, float mass,
                           float rho0, float simScale, float stiffness, f4* sortedPosition, float* pressure,
                           float* rho, float delta, int N);
void pcisph_correctPressure(f2* neighborMap, unsigned* particleIndexBack, float h, float mass, float rho0,
                            float simScale, float stiffness, f4* sortedPosition, float* pressure, float* rho,
                            float delta, f4* position, u2* particleIndex, int N);
void pcisph_computePressureForceAcceleration(f2* neighborMap, float* pressure, float* rho, f4* sortedPosition,
                                             f4* sortedVelocity, unsigned* particleIndexBack, float delta,
                                             double gradWspiky, float h, float mass, float mu, float simScale,
                                             f4* acceleration, float rho0, f4* position, u2* particleIndex, int N);
void clearMembraneBuffers(f4* position, f4* velocity, f4* sortedPosition, int N);
void computeInteractionWithMembranes(f4* position, f4* velocity, f4* sortedPosition, u2* particleIndex,
                                     unsigned* particleIndexBack, f2* neighborMap, int* particleMembranesList,
                                     int* membraneData, int N, int numOfElasticP, float r0);
void computeInteractionWithMembranes_finalize(f4* position, f4* velocity, u2* particleIndex,
                                              unsigned* particleIndexBack, int N);
void pcisph_integrate(f4* acceleration, f4* sortedPosition, f4* sortedVelocity, u2* particleIndex,
                      unsigned* particleIndexBack, float gx_, float gy_, float gz_, float simScaleInv, float timeStep,
                      float xmin, float xmax, float ymin, float ymax, float zmin, float zmax, float damping,
                      f4* position, f4* velocity, float* rho, float r0, f2* neighborMap, int N, int iterationCount);
}

// ---------------------------------------------------------------- restated host logic
// calcDelta: /root/reference/src/owPhysicsFluidSimulator.cpp:164-203 (same expression types:
// float sums, double sum1/sum2, `beta` is the float-subnormal constant of owPhysicsConstant.h:68).
static float ref_calcDelta() {
  static const float x[] = {1, 1, 0, -1, -1, -1, 0, 1, 1, 1, 0, -1, -1, -1, 0, 1, 1, 1, 0, -1, -1, -1, 0, 1, 2, -2, 0, 0, 0, 0, 0, 0};
  static const float y[] = {0, 1, 1, 1, 0, -1, -1, -1, 0, 1, 1, 1, 0, -1, -1, -1, 0, 1, 1, 1, 0, -1, -1, -1, 0, 0, 2, -2, 0, 0, 0, 0};
  static const float z[] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 0, 0, 0, 2, -2, 1, -1};
  float sx = 0.f, sy = 0.f, sz = 0.f;
  double sum1 = 0.0, sum2 = 0.0;
  float particleRadius = pow(mass / rho0, 1.f / 3.f);
  for (int i = 0; i < 32; i++) {
    float vx = x[i] * 0.8f * particleRadius, vy = y[i] * 0.8f * particleRadius, vz = z[i] * 0.8f * particleRadius;
    float dist = sqrt(vx * vx + vy * vy + vz * vz);
    if (dist <= h * simulationScale) {
      float h_r_2 = pow((h * simulationScale - dist), 2);
      sx += h_r_2 * vx / dist;
      sy += h_r_2 * vy / dist;
      sz += h_r_2 * vz / dist;
      sum2 += h_r_2 * h_r_2;
    }
  }
  sum1 = sx * sx + sy * sy + sz * sz;
  double result = 1.0 / (beta * gradWspikyCoefficient * gradWspikyCoefficient * (sum1 + sum2));
  return (float)result;
}
extern const float delta = ref_calcDelta();

// myCompare: owOpenCLSolver.cpp:690-696 (compares the cell id only; glibc qsort keeps ties stable).
static int ref_compare(const void* a, const void* b) {
  const int* f1 = (const int*)a;
  const int* f2 = (const int*)b;
  if (f1[0] < f2[0]) return -1;
  if (f1[0] > f2[0]) return +1;
  return 0;
}

struct RefSolver {
  int N, roundedUp;
  int gx, gy, gz, gridCellCount;
  float xmin, xmax, ymin, ymax, zmin, zmax;
  int numOfElasticP, elasticOffset, muscleCount, numOfMembranes;
  int threads;
  std::vector<f4> position, velocity, sortedPosition, sortedVelocity, acceleration, elastic;
  std::vector<f2> neighborMap;
  std::vector<u2> particleIndex;  // one spare element in front for indexx's idx-1 read (quirk #10)
  std::vector<unsigned> particleIndexBack, gridCellIndex, gridCellIndexFixedUp;
  std::vector<float> pressure, rho, muscle;
  std::vector<int> membraneData, particleMembranesList;
};

#define FOR_ITEMS(S, COUNT, CALL)                                              \
  do {                                                                         \
    int _n = (COUNT);                                                          \
    _Pragma("omp parallel for schedule(static) num_threads(S->threads)")       \
    for (int _g = 0; _g < _n; _g++) { g_gid = (size_t)_g; CALL; }              \
  } while (0)
#define FOR_ITEMS_SERIAL(COUNT, CALL)                                          \
  do { int _n = (COUNT); for (int _g = 0; _g < _n; _g++) { g_gid = (size_t)_g; CALL; } } while (0)

extern "C" {

// Constants exactly as the reference's compiler evaluates them.
struct ref_constants {
  float rho0, mass, timeStep, simulationScale, h, hashGridCellSize, hashGridCellSizeInv, simulationScaleInv, r0;
  float stiffness, viscosity, damping, gravity_x, gravity_y, gravity_z, delta;
  float xmin, xmax, ymin, ymax, zmin, zmax;
  int gridCellsX, gridCellsY, gridCellsZ, gridCellCount, maxIteration;
  double beta, Wpoly6Coefficient, gradWspikyCoefficient, del2WviscosityCoefficient;
};

void ref_get_constants(ref_constants* c) {
  c->rho0 = rho0; c->mass = mass; c->timeStep = timeStep; c->simulationScale = simulationScale; c->h = h;
  c->hashGridCellSize = hashGridCellSize; c->hashGridCellSizeInv = hashGridCellSizeInv;
  c->simulationScaleInv = simulationScaleInv; c->r0 = r0;
  c->stiffness = stiffness; c->viscosity = viscosity; c->damping = damping;
  c->gravity_x = gravity_x; c->gravity_y = gravity_y; c->gravity_z = gravity_z; c->delta = delta;
  // box + grid: owOpenCLSolver.cpp:7-17
  const float xmin_ = XMIN, xmax_ = XMAX, ymin_ = YMIN, ymax_ = YMAX, zmin_ = ZMIN, zmax_ = ZMAX;
  c->xmin = xmin_; c->xmax = xmax_; c->ymin = ymin_; c->ymax = ymax_; c->zmin = zmin_; c->zmax = zmax_;
  c->gridCellsX = (int)((XMAX - XMIN) / h) + 1;
  c->gridCellsY = (int)((YMAX - YMIN) / h) + 1;
  c->gridCellsZ = (int)((ZMAX - ZMIN) / h) + 1;
  c->gridCellCount = c->gridCellsX * c->gridCellsY * c->gridCellsZ;
  c->maxIteration = maxIteration;
  c->beta = beta; c->Wpoly6Coefficient = Wpoly6Coefficient; c->gradWspikyCoefficient = gradWspikyCoefficient;
  c->del2WviscosityCoefficient = del2WviscosityCoefficient;
}

// The box (and hence the grid) is a run-time parameter of the harness so that the synthetic boxes of
// SURVEY §8(d) can be run through the reference kernels; all physics constants stay the reference's.
RefSolver* ref_create(int N, float xmax_, float ymax_, float zmax_, int gx, int gy, int gz, const float* pos,
                      const float* vel, int numOfElasticP, int elasticOffset, const float* elastic,
                      int numOfMembranes, const int* membranes, const int* particleMembranesList, int threads) {
  RefSolver* S = new RefSolver();
  S->N = N; S->roundedUp = ((N - 1) / 256 + 1) * 256;
  S->gx = gx; S->gy = gy; S->gz = gz; S->gridCellCount = gx * gy * gz;
  S->xmin = 0; S->ymin = 0; S->zmin = 0; S->xmax = xmax_; S->ymax = ymax_; S->zmax = zmax_;
  S->numOfElasticP = numOfElasticP; S->elasticOffset = elasticOffset; S->muscleCount = 100;
  S->numOfMembranes = numOfMembranes; S->threads = threads > 0 ? threads : 1;
  S->position.assign(2 * (size_t)N, f4{0, 0, 0, 0}); S->velocity.assign(2 * (size_t)N, f4{0, 0, 0, 0});
  memcpy(S->position.data(), pos, sizeof(f4) * N); memcpy(S->velocity.data(), vel, sizeof(f4) * N);
  S->sortedPosition.assign(2 * (size_t)N, f4{0, 0, 0, 0}); S->sortedVelocity.assign(N, f4{0, 0, 0, 0});
  S->acceleration.assign(2 * (size_t)N, f4{0, 0, 0, 0});
  S->neighborMap.assign(32 * (size_t)N, f2{0, 0});
  S->particleIndex.assign((size_t)N + 1, u2{0, 0});
  S->particleIndexBack.assign(N, 0);
  S->gridCellIndex.assign((size_t)S->gridCellCount + 1, 0); S->gridCellIndexFixedUp.assign((size_t)S->gridCellCount + 1, 0);
  S->pressure.assign(N, 0.f); S->rho.assign(2 * (size_t)N, 0.f);
  S->muscle.assign(S->muscleCount, 0.f);  // quirk #18: never uploaded before step 0; zeros in practice
  if (numOfElasticP > 0 && elastic) { S->elastic.resize((size_t)numOfElasticP * 32); memcpy(S->elastic.data(), elastic, sizeof(f4) * 32 * numOfElasticP); }
  if (numOfMembranes > 0 && membranes) S->membraneData.assign(membranes, membranes + 3 * (size_t)numOfMembranes);
  if (numOfElasticP > 0 && particleMembranesList) S->particleMembranesList.assign(particleMembranesList, particleMembranesList + 7 * (size_t)numOfElasticP);
  return S;
}
void ref_destroy(RefSolver* S) { delete S; }

enum { ST_CLEAR = 0, ST_HASH, ST_SORT, ST_SORTPOST, ST_INDEXX, ST_INDEXPOST, ST_FIND, ST_DENSITY, ST_FORCES, ST_ELASTIC,
       ST_PREDICTPOS, ST_PREDICTDENS, ST_CORRECTP, ST_PRESSUREFORCE, ST_INTEGRATE, ST_CLEARMEMB, ST_MEMB, ST_MEMBFIN };

// One stage == one owOpenCLSolver::_run* (owOpenCLSolver.cpp:213-687); argument binding as there.
int ref_run(RefSolver* S, int stage, int iterationCount) {
  const int N = S->N, R = S->roundedUp;
  f4 *position = S->position.data(), *velocity = S->velocity.data(), *sp = S->sortedPosition.data(), *sv = S->sortedVelocity.data(), *acc = S->acceleration.data();
  f2* nm = S->neighborMap.data();
  u2* pi = S->particleIndex.data() + 1;
  unsigned* pib = S->particleIndexBack.data();
  float *pressure = S->pressure.data(), *rho = S->rho.data();
  switch (stage) {
    case ST_CLEAR: FOR_ITEMS(S, R, clearBuffers(nm, N)); break;
    case ST_HASH: FOR_ITEMS(S, R, hashParticles(position, S->gx, S->gy, S->gz, hashGridCellSizeInv, S->xmin, S->ymin, S->zmin, pi, N)); break;
    case ST_SORT: qsort(pi, N, 2 * sizeof(int), ref_compare); break;  // owOpenCLSolver.cpp:255-261
    case ST_SORTPOST: FOR_ITEMS(S, R, sortPostPass(pi, pib, position, velocity, sp, sv, N)); break;
    case ST_INDEXX: {
      int Gr = ((S->gridCellCount - 1) / 256 + 1) * 256;
      unsigned* gci = S->gridCellIndex.data();
      FOR_ITEMS(S, Gr, indexx(pi, S->gridCellCount, gci, N));
    } break;
    case ST_INDEXPOST: {  // owOpenCLSolver.cpp:305-319
      unsigned* b = S->gridCellIndexFixedUp.data();
      memcpy(b, S->gridCellIndex.data(), sizeof(unsigned) * ((size_t)S->gridCellCount + 1));
      int recent = S->gridCellCount;
      for (int i = S->gridCellCount; i >= 0; i--) {
        if (b[i] == (unsigned)NO_CELL_ID) b[i] = recent; else recent = b[i];
      }
    } break;
    case ST_FIND: FOR_ITEMS(S, R, findNeighbors(S->gridCellIndexFixedUp.data(), sp, S->gridCellCount, S->gx, S->gy, S->gz, h, hashGridCellSize, hashGridCellSizeInv, simulationScale, S->xmin, S->ymin, S->zmin, nm, N)); break;
    case ST_DENSITY: FOR_ITEMS(S, R, pcisph_computeDensity(nm, Wpoly6Coefficient, h, mass, rho0, simulationScale, stiffness, sp, pressure, rho, pib, delta, N)); break;
    case ST_FORCES: FOR_ITEMS(S, R, pcisph_computeForcesAndInitPressure(nm, rho, pressure, sp, sv, acc, pib, Wpoly6Coefficient, del2WviscosityCoefficient, h, mass, viscosity, simulationScale, gravity_x, gravity_y, gravity_z, position, pi, N)); break;
    case ST_ELASTIC:
      if (S->numOfElasticP == 0) return 0;
      FOR_ITEMS(S, ((S->numOfElasticP - 1) / 256 + 1) * 256, pcisph_computeElasticForces(nm, sp, sv, acc, pib, velocity, h, mass, simulationScale, S->numOfElasticP, S->elastic.data(), S->elasticOffset, N, S->muscleCount, S->muscle.data(), position));
      break;
    case ST_PREDICTPOS: FOR_ITEMS(S, R, pcisph_predictPositions(acc, sp, sv, pi, pib, gravity_x, gravity_y, gravity_z, simulationScaleInv, timeStep, S->xmin, S->xmax, S->ymin, S->ymax, S->zmin, S->zmax, damping, position, velocity, r0, nm, N)); break;
    case ST_PREDICTDENS: FOR_ITEMS(S, R, pcisph_predictDensity(nm, pib, Wpoly6Coefficient, h, mass, rho0, simulationScale, stiffness, sp, pressure, rho, delta, N)); break;
    case ST_CORRECTP: FOR_ITEMS(S, R, pcisph_correctPressure(nm, pib, h, mass, rho0, simulationScale, stiffness, sp, pressure, rho, delta, position, pi, N)); break;
    case ST_PRESSUREFORCE: FOR_ITEMS(S, R, pcisph_computePressureForceAcceleration(nm, pressure, rho, sp, sv, pib, delta, gradWspikyCoefficient, h, mass, viscosity, simulationScale, acc, rho0, position, pi, N)); break;
    // integrate and the membrane kernels write `position` while other work-items read it (quirk #24):
    // run them serially so the oracle has the serial semantics.
    case ST_INTEGRATE: FOR_ITEMS_SERIAL(R, pcisph_integrate(acc, sp, sv, pi, pib, gravity_x, gravity_y, gravity_z, simulationScaleInv, timeStep, S->xmin, S->xmax, S->ymin, S->ymax, S->zmin, S->zmax, damping, position, velocity, rho, r0, nm, N, iterationCount)); break;
    case ST_CLEARMEMB: FOR_ITEMS(S, R, clearMembraneBuffers(position, velocity, sp, N)); break;
    case ST_MEMB: FOR_ITEMS_SERIAL(R, computeInteractionWithMembranes(position, velocity, sp, pi, pib, nm, S->particleMembranesList.data(), S->membraneData.data(), N, S->numOfElasticP, r0)); break;
    case ST_MEMBFIN: FOR_ITEMS_SERIAL(R, computeInteractionWithMembranes_finalize(position, velocity, pi, pib, N)); break;
    default: return -1;
  }
  return 0;
}

// owPhysicsFluidSimulator::simulationStep (owPhysicsFluidSimulator.cpp:79-149), minus timing / I/O.
int ref_step(RefSolver* S, int iterationCount) {
  static const int pre[] = {ST_CLEAR, ST_HASH, ST_SORT, ST_SORTPOST, ST_INDEXX, ST_INDEXPOST, ST_FIND, ST_DENSITY, ST_FORCES, ST_ELASTIC};
  for (int s : pre) ref_run(S, s, iterationCount);
  int iter = 0;
  do {
    ref_run(S, ST_PREDICTPOS, iterationCount); ref_run(S, ST_PREDICTDENS, iterationCount);
    ref_run(S, ST_CORRECTP, iterationCount); ref_run(S, ST_PRESSUREFORCE, iterationCount);
    iter++;
  } while (iter < maxIteration);
  ref_run(S, ST_INTEGRATE, iterationCount);
  ref_run(S, ST_CLEARMEMB, iterationCount); ref_run(S, ST_MEMB, iterationCount); ref_run(S, ST_MEMBFIN, iterationCount);
  return 0;
}

void ref_update_muscles(RefSolver* S, const float* signal) { memcpy(S->muscle.data(), signal, sizeof(float) * S->muscleCount); }

// name → raw buffer (reference layout, table 2.2 of SURVEY.md). Returns byte size, 0 if unknown.
size_t ref_buffer(RefSolver* S, const char* name, void** ptr) {
#define BUF(n, v, off) if (!strcmp(name, n)) { *ptr = (void*)((v).data() + (off)); return sizeof((v)[0]) * ((v).size() - (off)); }
  BUF("position", S->position, 0) BUF("velocity", S->velocity, 0) BUF("sortedPosition", S->sortedPosition, 0)
  BUF("sortedVelocity", S->sortedVelocity, 0) BUF("acceleration", S->acceleration, 0) BUF("neighborMap", S->neighborMap, 0)
  BUF("particleIndex", S->particleIndex, 1) BUF("particleIndexBack", S->particleIndexBack, 0)
  BUF("gridCellIndex", S->gridCellIndex, 0) BUF("gridCellIndexFixedUp", S->gridCellIndexFixedUp, 0)
  BUF("pressure", S->pressure, 0) BUF("rho", S->rho, 0)
#undef BUF
  *ptr = nullptr;
  return 0;
}

}  // extern "C"
