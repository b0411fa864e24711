// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// Drives the reference's own scene generator (owHelper::generateConfiguration, compiled from
// /root/reference/src/owHelper.cpp where it lies) the way owPhysicsFluidSimulator.cpp:27-66 does,
// and hands the five arrays back through a C interface. Part of oracle/_ref/libsphref.so.
#include <cstdlib>
#include <cstring>
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

void ref_free_scene(ref_scene* s) {
  delete[] s->position; delete[] s->velocity; delete[] s->elasticConnections; delete[] s->membraneData;
  delete[] s->particleMembranesList;
  memset(s, 0, sizeof(*s));
}

}  // extern "C"
