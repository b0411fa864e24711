// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// Drives the reference's own scene generator (owHelper::generateConfiguration, compiled from
// /root/reference/src/owHelper.cpp where it lies) the way owPhysicsFluidSimulator.cpp:27-66 does,
// and hands the five arrays back through a C interface. Part of oracle/_ref/libsphref.so.
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include "owHelper.h"
#include "owPhysicsConstant.h"

// globals the reference's owHelper.cpp expects (owWorldSimulation.cpp:29-31,47; owPhysicsFluidSimulator.cpp:9-14)
int PARTICLE_COUNT = 0;
int PARTICLE_COUNT_RoundedUp = 0;
int local_NDRange_size = 256;
int numOfElasticConnections = 0;
int numOfMembranes = 0;
int numOfElasticP = 0;
int numOfLiquidP = 0;
int iterationCount = 0;

extern "C" {

struct ref_scene {
  int N, numOfLiquidP, numOfElasticP, numOfBoundaryP, numOfMembranes;
  float* position;               // 4N
  float* velocity;               // 4N
  float* elasticConnections;     // 4*32*numOfElasticP
  int* membraneData;             // 3*numOfMembranes
  int* particleMembranesList;    // 7*numOfElasticP
};

int ref_generate_scene(ref_scene* out) {
  float *position_cpp = nullptr, *velocity_cpp = nullptr, *elastic = nullptr;
  int *membraneData_cpp = nullptr, *pml = nullptr;
  int numOfBoundaryP = 0;
  numOfLiquidP = numOfElasticP = numOfMembranes = numOfElasticConnections = 0;
  owHelper::generateConfiguration(0, position_cpp, velocity_cpp, elastic, membraneData_cpp, numOfLiquidP, numOfElasticP,
                                  numOfBoundaryP, numOfElasticConnections, numOfMembranes, pml);
  position_cpp = new float[4 * (size_t)PARTICLE_COUNT];
  velocity_cpp = new float[4 * (size_t)PARTICLE_COUNT];
  membraneData_cpp = numOfMembranes > 0 ? new int[3 * (size_t)numOfMembranes] : nullptr;
  pml = numOfElasticP > 0 ? new int[7 * (size_t)numOfElasticP] : nullptr;
  owHelper::generateConfiguration(1, position_cpp, velocity_cpp, elastic, membraneData_cpp, numOfLiquidP, numOfElasticP,
                                  numOfBoundaryP, numOfElasticConnections, numOfMembranes, pml);
  out->N = PARTICLE_COUNT; out->numOfLiquidP = numOfLiquidP; out->numOfElasticP = numOfElasticP;
  out->numOfBoundaryP = numOfBoundaryP; out->numOfMembranes = numOfMembranes;
  out->position = position_cpp; out->velocity = velocity_cpp; out->elasticConnections = elastic;
  out->membraneData = membraneData_cpp; out->particleMembranesList = pml;
  return 0;
}

// The reference's own text loader (owHelper::preLoadConfiguration + loadConfiguration, owHelper.cpp:1431-1545). It reads the
// hard-coded relative paths ./configuration/position.txt, velocity.txt and (if there are elastic particles)
// elasticconnections.txt, so the caller passes the directory that holds `configuration/` and we chdir there for the call.
int ref_load_configuration(const char* dir, ref_scene* out) {
  char cwd[4096];
  if (!getcwd(cwd, sizeof(cwd)) || chdir(dir) != 0) return -1;
  owHelper::preLoadConfiguration();
  float* position_cpp = new float[4 * (size_t)PARTICLE_COUNT];
  float* velocity_cpp = new float[4 * (size_t)PARTICLE_COUNT];
  float* elastic = nullptr;
  int nl = 0, ne = 0, nb = 0, nc = 0;
  owHelper::loadConfiguration(position_cpp, velocity_cpp, elastic, nl, ne, nb, nc);
  if (chdir(cwd) != 0) return -2;
  out->N = PARTICLE_COUNT; out->numOfLiquidP = nl; out->numOfElasticP = ne; out->numOfBoundaryP = nb; out->numOfMembranes = 0;
  out->position = position_cpp; out->velocity = velocity_cpp; out->elasticConnections = elastic;
  out->membraneData = nullptr; out->particleMembranesList = nullptr;
  return 0;
}

// The reference's own `-l_to` writer (owHelper::loadConfigurationToFile, owHelper.cpp:1640-1672) into <dir>/buffers/. It reads
// the globals PARTICLE_COUNT, numOfElasticP, numOfLiquidP, numOfMembranes and indexes `membranes` with stride 4.
int ref_save_configuration(const char* dir, float* position, int count, float* connections, int* membranes4, int nElastic,
                           int nLiquid, int nMembranes, int firstIteration) {
  char cwd[4096];
  if (!getcwd(cwd, sizeof(cwd)) || chdir(dir) != 0) return -1;
  PARTICLE_COUNT = count; numOfElasticP = nElastic; numOfLiquidP = nLiquid; numOfMembranes = nMembranes;
  owHelper::loadConfigurationToFile(position, connections, membranes4, firstIteration != 0);
  return chdir(cwd) != 0 ? -2 : 0;
}

void ref_free_scene(ref_scene* s) {
  delete[] s->position; delete[] s->velocity; delete[] s->elasticConnections; delete[] s->membraneData;
  delete[] s->particleMembranesList;
  memset(s, 0, sizeof(*s));
}

}  // extern "C"
