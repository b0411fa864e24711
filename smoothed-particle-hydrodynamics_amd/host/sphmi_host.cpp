// libsphmi_host.so — host-side mirror of the reference's constants, loaders and scene helpers (no GPU code).
// Every function cites the reference lines whose behaviour it reproduces. Build with g++ -O2 -ffp-contract=off:
// the constants below depend on IEEE gradual underflow and on float-typed sub-expressions.
#include "sphmi_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <string>
#include <vector>

namespace {

// ---- owPhysicsConstant.h:12-27,62-76, evaluated in the same operand types ---------------------------------
struct PhysConst {
  float rho0, mass, timeStep, simulationScale, h, hashGridCellSize, hashGridCellSizeInv, simulationScaleInv, r0;
  float viscosity, gravity_x, gravity_y, gravity_z, delta;
  double beta, Wpoly6, gradWspiky, del2Wvisc;
};

// calcDelta — owPhysicsFluidSimulator.cpp:164-203. Float running sums, double sum1/sum2, `beta` a float subnormal.
float calc_delta(const PhysConst& c) {
  static const float x[] = {1, 1, 0, -1, -1, -1, 0, 1, 1, 1, 0, -1, -1, -1, 0, 1, 1, 1, 0, -1, -1, -1, 0, 1, 2, -2, 0, 0, 0, 0, 0, 0};
  static const float y[] = {0, 1, 1, 1, 0, -1, -1, -1, 0, 1, 1, 1, 0, -1, -1, -1, 0, 1, 1, 1, 0, -1, -1, -1, 0, 0, 2, -2, 0, 0, 0, 0};
  static const float z[] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 0, 0, 0, 2, -2, 1, -1};
  float sx = 0.f, sy = 0.f, sz = 0.f;
  double sum1 = 0.0, sum2 = 0.0;
  const float particleRadius = powf(c.mass / c.rho0, 1.f / 3.f);
  const float hs = c.h * c.simulationScale;
  for (int i = 0; i < 32; i++) {
    float vx = x[i] * 0.8f * particleRadius, vy = y[i] * 0.8f * particleRadius, vz = z[i] * 0.8f * particleRadius;
    float dist = sqrtf(vx * vx + vy * vy + vz * vz);
    if (dist <= hs) {
      float t = hs - dist;
      float h_r_2 = (float)((double)t * (double)t);  // pow(float,2) promotes to double, then narrows (:187)
      sx += h_r_2 * vx / dist;
      sy += h_r_2 * vy / dist;
      sz += h_r_2 * vz / dist;
      sum2 += h_r_2 * h_r_2;
    }
  }
  sum1 = sx * sx + sy * sy + sz * sz;
  double result = 1.0 / (c.beta * c.gradWspiky * c.gradWspiky * (sum1 + sum2));
  return (float)result;
}

PhysConst phys_const() {
  PhysConst c;
  c.rho0 = 1000.0f;
  c.mass = 3.25e-14f;
  c.timeStep = 5.0e-06f;
  c.simulationScale = 0.004f * powf(c.mass, 1.f / 3.f) / powf(0.00025f, 1.f / 3.f);
  c.h = 3.34f;
  c.hashGridCellSize = 2.0f * c.h;
  c.hashGridCellSizeInv = 1.0f / c.hashGridCellSize;
  c.simulationScaleInv = 1.0f / c.simulationScale;
  c.r0 = 0.5f * c.h;
  c.viscosity = 0.00005f;
  c.gravity_x = 0.0f; c.gravity_y = -9.8f; c.gravity_z = 0.0f;
  // `const double beta = timeStep*timeStep*mass*mass*2/(rho0*rho0)`: a float expression (subnormal), widened after
  volatile float b = c.timeStep * c.timeStep * c.mass * c.mass * 2 / (c.rho0 * c.rho0);
  c.beta = (double)b;
  c.Wpoly6 = 315.0 / (64.0 * M_PI * pow((double)(c.h * c.simulationScale), 9.0));
  c.gradWspiky = -45.0 / (M_PI * pow((double)(c.h * c.simulationScale), 6.0));
  c.del2Wvisc = -c.gradWspiky;
  c.delta = calc_delta(c);
  return c;
}

// sphFluid.cl:662
float surf_tens_coeff(double Wpoly6, float h, float simulationScale) {
  float hScaled = h * simulationScale;
  float hScaled2 = hScaled * hScaled;
  return (-1.5e-09f * 0.3f * (float)(Wpoly6 * pow(hScaled2 / 2.0, 3.0)) * simulationScale);
}

// PCG32 (XSH-RR), the generator SURVEY §8(d) names for the optional lattice jitter.
struct Pcg32 {
  uint64_t state, inc;
  explicit Pcg32(uint64_t seed) : state(0), inc((54u << 1) | 1u) { next(); state += seed; next(); }
  uint32_t next() {
    uint64_t old = state;
    state = old * 6364136223846793005ULL + inc;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((32u - rot) & 31u));
  }
  float uniform_pm1() { return 2.0f * ((float)(next() >> 8) * (1.0f / 16777216.0f)) - 1.0f; }
};

struct BoundaryDims { int nx, ny, nz; };
// owHelper.cpp:717-719 — `(int)((XMAX - XMIN)/r0)`: XMAX is the *double* macro `30.0*h`, not the narrowed float xmax
BoundaryDims boundary_dims(const sph_config* cfg, double xm, double ym, double zm) {
  BoundaryDims d;
  const double h = (double)cfg->h;
  d.nx = (int)((xm * h - 0) / cfg->r0);
  d.ny = (int)((ym * h - 0) / cfg->r0);
  d.nz = (int)((zm * h - 0) / cfg->r0);
  return d;
}

inline void put(float* a, size_t i, float x, float y, float z, float w) {
  a[4 * i + 0] = x; a[4 * i + 1] = y; a[4 * i + 2] = z; a[4 * i + 3] = w;
}

// Visits every particle of the synthetic box in generation order (liquid lattice, x fastest; then the boundary shell in the
// visiting order and float expressions of owHelper.cpp:776-927): emit(index, x, y, z, type, vx, vy, vz, vw).
template <typename Emit>
size_t walk_box(const sph_config* cfg, double xm, double ym, double zm, int lx, int ly, int lz, float spacing, float ox,
                       float oy, float oz, float jitter, uint64_t seed, Emit emit) {
  size_t i = 0;
  Pcg32 rng(seed);
  // liquid lattice (type 1.1 as in configuration/position*.txt), x fastest
  for (int iz = 0; iz < lz; iz++)
    for (int iy = 0; iy < ly; iy++)
      for (int ix = 0; ix < lx; ix++) {
        float x = ox + (float)ix * spacing, y = oy + (float)iy * spacing, z = oz + (float)iz * spacing;
        if (jitter > 0.f) { x += jitter * rng.uniform_pm1(); y += jitter * rng.uniform_pm1(); z += jitter * rng.uniform_pm1(); }
        emit(i, x, y, z, 1.1f, 0.f, 0.f, 0.f, 0.f);
        i++;
      }
  // boundary shell — owHelper.cpp:776-927 (same visiting order, same float expressions)
  const BoundaryDims d = boundary_dims(cfg, xm, ym, zm);
  const int nx = d.nx, ny = d.ny, nz = d.nz;
  const float r0 = cfg->r0, t = (float)SPH_BOUNDARY_PARTICLE;
  const float s3 = sqrtf(3.f), s2 = sqrtf(2.f);
  for (int ix = 0; ix < nx; ix++)
    for (int iy = 0; iy < ny; iy++) {  // 1 - top and bottom
      const bool ex = (ix == 0) || (ix == nx - 1), ey = (iy == 0) || (iy == ny - 1);
      const float px = ix * r0 + r0 / 2, py = iy * r0 + r0 / 2;
      const float z0 = 0 * r0 + r0 / 2, z1 = (nz - 1) * r0 + r0 / 2;
      const float sx = 1.f * (ix == 0) - 1 * (ix == nx - 1), sy = 1.f * (iy == 0) - 1 * (iy == ny - 1);
      if (ex || ey) {
        const float q = (ex && ey) ? s3 : s2;  // corners / edges
        emit(i, px, py, z0, t, sx / q, sy / q, 1.f / q, t); i++;
        emit(i, px, py, z1, t, sx / q, sy / q, -1.f / q, t); i++;
      } else {  // planes
        emit(i, px, py, z0, t, 0.f, 0.f, 1.f, t); i++;
        emit(i, px, py, z1, t, 0.f, 0.f, -1.f, t); i++;
      }
    }
  for (int ix = 0; ix < nx; ix++)
    for (int iz = 1; iz < nz - 1; iz++) {  // 2 - side walls OX-OZ and opposite
      const float px = ix * r0 + r0 / 2, pz = iz * r0 + r0 / 2;
      const float y0 = 0 * r0 + r0 / 2, y1 = (ny - 1) * r0 + r0 / 2;
      if ((ix == 0) || (ix == nx - 1)) {  // edges; the z component is 0 there because 1 <= iz <= nz-2 (:866,:875)
        const float nzc = 1.f * ((iz == 0) - (iz == nz - 1)) / s2;
        emit(i, px, y0, pz, t, 0.f, 1.f / s2, nzc, t); i++;
        emit(i, px, y1, pz, t, 0.f, -1.f / s2, nzc, t); i++;
      } else {
        emit(i, px, y0, pz, t, 0.f, 1.f, 0.f, t); i++;
        emit(i, px, y1, pz, t, 0.f, -1.f, 0.f, t); i++;
      }
    }
  for (int iy = 1; iy < ny - 1; iy++)
    for (int iz = 1; iz < nz - 1; iz++) {  // 3 - side walls OY-OZ and opposite
      const float py = iy * r0 + r0 / 2, pz = iz * r0 + r0 / 2;
      emit(i, 0 * r0 + r0 / 2, py, pz, t, 1.f, 0.f, 0.f, t); i++;
      emit(i, (nx - 1) * r0 + r0 / 2, py, pz, t, -1.f, 0.f, 0.f, t); i++;
    }
  return i;
}

}  // namespace

extern "C" {

int sphmi_default_config(sph_config* cfg) {
  if (!cfg) return SPH_ERR_INVALID;
  memset(cfg, 0, sizeof(*cfg));
  const PhysConst c = phys_const();
  cfg->abi_version = SPHMI_ABI_VERSION;
  cfg->h = c.h; cfg->hashGridCellSize = c.hashGridCellSize; cfg->hashGridCellSizeInv = c.hashGridCellSizeInv;
  cfg->simulationScale = c.simulationScale; cfg->simulationScaleInv = c.simulationScaleInv;
  cfg->r0 = c.r0; cfg->mass = c.mass; cfg->rho0 = c.rho0; cfg->timeStep = c.timeStep; cfg->viscosity = c.viscosity;
  cfg->delta = c.delta;
  cfg->gravity_x = c.gravity_x; cfg->gravity_y = c.gravity_y; cfg->gravity_z = c.gravity_z;
  cfg->Wpoly6Coefficient = c.Wpoly6; cfg->gradWspikyCoefficient = c.gradWspiky; cfg->del2WviscosityCoefficient = c.del2Wvisc;
  cfg->surfTensCoeff = surf_tens_coeff(c.Wpoly6, c.h, c.simulationScale);
  cfg->muscleCount = 100;  // owWorldSimulation.cpp:31
  cfg->maxIteration = 3;   // owPhysicsConstant.h:76
  return sphmi_config_set_box(cfg, 30.0, 20.0, 250.0, 0xffffu);  // owPhysicsConstant.h:32-37
}

int sphmi_config_set_box(sph_config* cfg, double xm, double ym, double zm, uint32_t cellIdMask) {
  if (!cfg || !(xm > 0) || !(ym > 0) || !(zm > 0)) return SPH_ERR_INVALID;
  const double h = (double)cfg->h;
  // `#define XMAX 30.0*h` is a double expression; `const float xmax = XMAX` narrows it (owOpenCLSolver.cpp:7-12)
  const double X = xm * h, Y = ym * h, Z = zm * h;
  cfg->xmin = 0.f; cfg->ymin = 0.f; cfg->zmin = 0.f;
  cfg->xmax = (float)X; cfg->ymax = (float)Y; cfg->zmax = (float)Z;
  // owOpenCLSolver.cpp:14-17 — divides by h (not by the cell size 2h): SURVEY App. B #2
  cfg->gridCellsX = (int)((X - 0) / cfg->h) + 1;
  cfg->gridCellsY = (int)((Y - 0) / cfg->h) + 1;
  cfg->gridCellsZ = (int)((Z - 0) / cfg->h) + 1;
  const long long g = (long long)cfg->gridCellsX * cfg->gridCellsY * cfg->gridCellsZ;
  if (g >= 0x7fffffffLL) return SPH_ERR_INVALID;
  cfg->gridCellCount = (int)g;
  cfg->cellIdMask = cellIdMask;
  return SPH_OK;
}

int sphmi_count_particles(const char* positionFile) {
  std::ifstream f(positionFile);
  if (!f.is_open()) return -1;
  int count = 0;
  while (f.good()) {  // owHelper.cpp:1441-1447
    float x, y, z, p_type = -1.1f;
    f >> x >> y >> z >> p_type;
    if (p_type >= 0) count++; else break;
  }
  return count;
}

int sphmi_load_configuration(const char* positionFile, const char* velocityFile, int count, float* position,
                             float* velocity, int* nLiquid, int* nElastic, int* nBoundary) {
  if (!position || !velocity || count <= 0) return SPH_ERR_INVALID;
  int nl = 0, ne = 0, nb = 0;
  std::ifstream pf(positionFile);
  if (!pf.is_open()) return SPH_ERR_INVALID;
  int i = 0;
  float x, y, z, w;
  while (pf.good() && i < count) {  // owHelper.cpp:1470-1489
    pf >> x >> y >> z >> w;
    put(position, (size_t)i, x, y, z, w);
    switch ((int)w) {
      case SPH_LIQUID_PARTICLE: nl++; break;
      case SPH_ELASTIC_PARTICLE: ne++; break;
      case SPH_BOUNDARY_PARTICLE: nb++; break;
    }
    i++;
  }
  if (i != count) return SPH_ERR_SIZE;
  std::ifstream vf(velocityFile);
  if (!vf.is_open()) return SPH_ERR_INVALID;
  i = 0;
  while (vf.good() && i < count) {  // owHelper.cpp:1498-1506
    vf >> x >> y >> z >> w;
    put(velocity, (size_t)i, x, y, z, w);
    i++;
  }
  if (i != count) return SPH_ERR_SIZE;
  if (nLiquid) *nLiquid = nl;
  if (nElastic) *nElastic = ne;
  if (nBoundary) *nBoundary = nb;
  return SPH_OK;
}

int sphmi_load_elastic_connections(const char* file, int numOfElasticP, float* out) {
  if (!out || numOfElasticP <= 0) return SPH_ERR_INVALID;
  std::ifstream f(file);
  if (!f.is_open()) return SPH_ERR_INVALID;
  const int cap = numOfElasticP * SPH_MAX_NEIGHBOR_COUNT;
  int i = 0;
  while (f.good() && i < cap) {  // owHelper.cpp:1526-1538
    float jd = -10, rij0 = 0, v1 = 0, v2 = 0;
    f >> jd >> rij0 >> v1 >> v2;
    if (jd >= -1) { put(out, (size_t)i, jd, rij0, v1, v2); i++; }
  }
  return i;
}

int sphmi_box_counts(const sph_config* cfg, double xm, double ym, double zm, int lx, int ly, int lz, int* nLiquid,
                     int* nBoundary) {
  if (!cfg || lx < 0 || ly < 0 || lz < 0) return SPH_ERR_INVALID;
  const BoundaryDims d = boundary_dims(cfg, xm, ym, zm);
  if (d.nx < 2 || d.ny < 2 || d.nz < 2) return SPH_ERR_INVALID;
  const long long liq = (long long)lx * ly * lz;
  const long long bnd = 2LL * ((long long)d.nx * d.ny + (long long)(d.nx + d.ny - 2) * (d.nz - 2));  // owHelper.cpp:770
  if (liq + bnd >= 0x7fffffffLL) return SPH_ERR_INVALID;
  if (nLiquid) *nLiquid = (int)liq;
  if (nBoundary) *nBoundary = (int)bnd;
  return SPH_OK;
}

int sphmi_generate_box(const sph_config* cfg, double xm, double ym, double zm, int lx, int ly, int lz, float spacing,
                       float ox, float oy, float oz, float jitter, uint64_t seed, float* position, float* velocity) {
  int nl = 0, nb = 0;
  int rc = sphmi_box_counts(cfg, xm, ym, zm, lx, ly, lz, &nl, &nb);
  if (rc != SPH_OK || !position || !velocity) return rc != SPH_OK ? rc : SPH_ERR_INVALID;
  const size_t n = walk_box(cfg, xm, ym, zm, lx, ly, lz, spacing, ox, oy, oz, jitter, seed,
                            [&](size_t i, float x, float y, float z, float w, float vx, float vy, float vz, float vw) {
                              put(position, i, x, y, z, w);
                              put(velocity, i, vx, vy, vz, vw);
                            });
  return (n == (size_t)nl + (size_t)nb) ? SPH_OK : SPH_ERR_SIZE;
}

// z cell layer of a particle exactly as hashParticles computes it (sphFluid.cl:199): (int)(z * hashGridCellSizeInv) in float
static inline int box_layer(const sph_config* cfg, float z) { return (int)(z * cfg->hashGridCellSizeInv); }

int sphmi_box_layer_histogram(const sph_config* cfg, double xm, double ym, double zm, int lx, int ly, int lz, float spacing,
                              float ox, float oy, float oz, float jitter, uint64_t seed, int64_t* hist, int layers) {
  int nl = 0, nb = 0;
  int rc = sphmi_box_counts(cfg, xm, ym, zm, lx, ly, lz, &nl, &nb);
  if (rc != SPH_OK || !hist || layers <= 0) return rc != SPH_OK ? rc : SPH_ERR_INVALID;
  for (int i = 0; i < layers; i++) hist[i] = 0;
  bool outside = false;
  walk_box(cfg, xm, ym, zm, lx, ly, lz, spacing, ox, oy, oz, jitter, seed,
           [&](size_t, float, float, float z, float, float, float, float, float) {
             const int l = box_layer(cfg, z);
             if (l < 0 || l >= layers) outside = true; else hist[l]++;
           });
  return outside ? SPH_ERR_SIZE : SPH_OK;
}

int sphmi_generate_box_slice(const sph_config* cfg, double xm, double ym, double zm, int lx, int ly, int lz, float spacing,
                             float ox, float oy, float oz, float jitter, uint64_t seed, int layerLo, int layerHi,
                             float* position, float* velocity, uint32_t* globalIds, int capacity, int* count) {
  int nl = 0, nb = 0;
  int rc = sphmi_box_counts(cfg, xm, ym, zm, lx, ly, lz, &nl, &nb);
  if (rc != SPH_OK || !count) return rc != SPH_OK ? rc : SPH_ERR_INVALID;
  const bool fill = position && velocity && globalIds;
  size_t k = 0;
  walk_box(cfg, xm, ym, zm, lx, ly, lz, spacing, ox, oy, oz, jitter, seed,
           [&](size_t i, float x, float y, float z, float w, float vx, float vy, float vz, float vw) {
             const int l = box_layer(cfg, z);
             if (l < layerLo || l >= layerHi) return;
             if (fill && k < (size_t)capacity) {
               put(position, k, x, y, z, w);
               put(velocity, k, vx, vy, vz, vw);
               globalIds[k] = (uint32_t)i;
             }
             k++;
           });
  *count = (int)k;
  return (fill && k > (size_t)capacity) ? SPH_ERR_SIZE : SPH_OK;
}

int sphmi_save_configuration(const char* dir, const float* position, int count, int numOfElasticP, int numOfLiquidP,
                             const float* connections, const int32_t* membranes, int numOfMembranes, int firstIteration) {
  if (!dir || !position || count <= 0) return SPH_ERR_INVALID;
  const std::string base = std::string(dir) + "/";
  std::ofstream positionFile;
  if (firstIteration) {  // owHelper.cpp:1643-1649
    positionFile.open((base + "position_buffer.txt").c_str(), std::ofstream::trunc);
    positionFile << numOfElasticP << "\n";
    positionFile << numOfLiquidP << "\n";
  } else {
    positionFile.open((base + "position_buffer.txt").c_str(), std::ofstream::app);
  }
  if (!positionFile.is_open()) return SPH_ERR_INVALID;
  for (int i = 0; i < count; i++) {
    if ((int)position[4 * i + 3] != SPH_BOUNDARY_PARTICLE)
      positionFile << position[i * 4 + 0] << "\t" << position[i * 4 + 1] << "\t" << position[i * 4 + 2] << "\t"
                   << position[i * 4 + 3] << "\n";
  }
  positionFile.close();
  if (firstIteration) {
    if (connections && numOfElasticP > 0) {
      std::ofstream connectionFile((base + "connection_buffer.txt").c_str(), std::ofstream::trunc);
      const int con_num = SPH_MAX_NEIGHBOR_COUNT * numOfElasticP;
      for (int i = 0; i < con_num; i++)
        connectionFile << connections[4 * i + 0] << "\t" << connections[4 * i + 1] << "\t" << connections[4 * i + 2] << "\t"
                       << connections[4 * i + 3] << "\n";
    }
    if (membranes && numOfMembranes > 0) {
      std::ofstream membranesFile((base + "membranes_buffer.txt").c_str(), std::ofstream::trunc);
      membranesFile << numOfMembranes << "\n";
      for (int i = 0; i < numOfMembranes; i++)
        membranesFile << membranes[3 * i + 0] << "\t" << membranes[3 * i + 1] << "\t" << membranes[3 * i + 2] << "\t" << 0 << "\n";
    }
  }
  return SPH_OK;
}

// owHelper::loadConfigurationFromFile (owHelper.cpp:1674-1741), the `-l_from` playback input: the files written above.
// The position file is "numOfElasticP\nnumOfLiquidP\n" followed by frames of numOfElasticP + numOfLiquidP rows
// `x y z type`; the reference slurps all rows (its `while(good())` loop duplicates the last one, which the integer division
// `rows / PARTICLE_COUNT` then drops) and serves frame `iteration` from memory.
int sphmi_trajectory_info(const char* dir, int* numOfElasticP, int* numOfLiquidP, int* frames, int* numOfMembranes) {
  if (!dir) return SPH_ERR_INVALID;
  const std::string base = std::string(dir) + "/";
  std::ifstream positionFile((base + "position_buffer.txt").c_str());
  if (!positionFile.is_open()) return SPH_ERR_INVALID;
  int ne = 0, nl = 0;
  positionFile >> ne >> nl;
  if (!positionFile.good() || ne < 0 || nl < 0 || ne + nl <= 0) return SPH_ERR_INVALID;
  long long rows = 0;
  float x, y, z, t;
  while (positionFile >> x >> y >> z >> t) rows++;
  if (numOfElasticP) *numOfElasticP = ne;
  if (numOfLiquidP) *numOfLiquidP = nl;
  if (frames) *frames = (int)(rows / (ne + nl));
  if (numOfMembranes) {
    *numOfMembranes = 0;
    std::ifstream membranesFile((base + "membranes_buffer.txt").c_str());
    if (membranesFile.is_open()) membranesFile >> *numOfMembranes;
  }
  return SPH_OK;
}

int sphmi_trajectory_frame(const char* dir, int frame, float* position) {
  int ne = 0, nl = 0, frames = 0;
  int rc = sphmi_trajectory_info(dir, &ne, &nl, &frames, nullptr);
  if (rc != SPH_OK) return rc;
  if (!position || frame < 0 || frame >= frames) return SPH_ERR_INVALID;
  std::ifstream positionFile((std::string(dir) + "/position_buffer.txt").c_str());
  int a, b;
  positionFile >> a >> b;
  const long long P = (long long)ne + nl, skip = P * frame;
  float v[4];
  for (long long i = 0; i < skip + P; i++) {
    if (!(positionFile >> v[0] >> v[1] >> v[2] >> v[3])) return SPH_ERR_SIZE;
    if (i >= skip) memcpy(position + 4 * (i - skip), v, sizeof(v));
  }
  return SPH_OK;
}

int sphmi_trajectory_connections(const char* dir, int numOfElasticP, float* connections) {
  if (!dir || !connections || numOfElasticP <= 0) return SPH_ERR_INVALID;
  std::ifstream connectionFile((std::string(dir) + "/connection_buffer.txt").c_str());
  if (!connectionFile.is_open()) return SPH_ERR_INVALID;
  const long long n = (long long)SPH_MAX_NEIGHBOR_COUNT * numOfElasticP;
  for (long long i = 0; i < n; i++)
    if (!(connectionFile >> connections[4 * i] >> connections[4 * i + 1] >> connections[4 * i + 2] >> connections[4 * i + 3])) return SPH_ERR_SIZE;
  return SPH_OK;
}

int sphmi_trajectory_membranes(const char* dir, int numOfMembranes, int32_t* membranes) {
  if (!dir || !membranes || numOfMembranes <= 0) return SPH_ERR_INVALID;
  std::ifstream membranesFile((std::string(dir) + "/membranes_buffer.txt").c_str());
  int count = 0;
  if (!membranesFile.is_open() || !(membranesFile >> count) || count < numOfMembranes) return SPH_ERR_INVALID;
  for (int i = 0; i < numOfMembranes; i++) {  // rows of four ints as the reference reads them (i, j, k, unused)
    int32_t* m = membranes + 4 * (size_t)i;
    if (!(membranesFile >> m[0] >> m[1] >> m[2] >> m[3])) return SPH_ERR_SIZE;
  }
  return SPH_OK;
}

int sphmi_muscle_signal(int step, float* out, int muscleCount) {
  if (!out || muscleCount < 96) return SPH_ERR_INVALID;
  for (int i = 0; i < muscleCount; i++) out[i] = 0.f;
  float w1[24], w2[24];
  for (int k = 0; k < 12; k++) {  // main_sim.py:19-37: linspace(0, 1.5*2*pi, 12), velocity 1e-4, phase pi
    const double pos = (double)k * ((1.5 * 2 * M_PI) / 11.0);
    const double a = (sin(pos - 0.0001 * (double)step) + 1) / 2;
    const double b = (sin(pos + M_PI - 0.0001 * (double)step) + 1) / 2;
    w1[2 * k] = w1[2 * k + 1] = (float)a;
    w2[2 * k] = w2[2 * k + 1] = (float)b;
  }
  // main_sim.py:49-52: [W1, W2, W2, W1]
  memcpy(out, w1, sizeof(w1)); memcpy(out + 24, w2, sizeof(w2)); memcpy(out + 48, w2, sizeof(w2)); memcpy(out + 72, w1, sizeof(w1));
  return SPH_OK;
}

}  // extern "C"
