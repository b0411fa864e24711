/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's per-step hot path
 *   owPhysicsFluidSimulator::simulationStep()  (/root/reference/src/owPhysicsFluidSimulator.cpp:79-149)
 *   -> owOpenCLSolver::_run*                    (/root/reference/src/owOpenCLSolver.cpp:213-687)
 *   -> the 16 kernels of                        (/root/reference/src/sphFluid.cl)
 * with every constant a run-time parameter. It is the checker for the HIP path and the timed
 * `cpu_baseline` ("port") of bench.py. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (libsphmi.so) never does.
 *
 * PARITY PIN: bit-exact against oracle/_ref/libsphref.so (the reference's own kernels compiled
 * for x86-64) on configs #1/#3 and synthetic boxes in the build container (tests/test_oracle_vs_ref.py),
 * and against the committed fixtures tests/golden/ *.npz that libsphref.so generated
 * (tests/golden/make_golden.py). The reference itself ships no tests or golden vectors.
 *
 * Buffers are kept in the reference's layouts except that neighbour ids
 * and cell ids are kept as integers beside their float encodings, so that "wide" mode
 * (cell_id_mask = 0xffffffff, N > 2^24) has an exact oracle too.
 */
#ifndef SPH_ORACLE_H
#define SPH_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sph_oracle_params {
  int32_t N;
  int32_t gridCellsX, gridCellsY, gridCellsZ, gridCellCount;
  uint32_t cellIdMask; /* 0xffff = reference ("ref16"), 0xffffffff = wide */
  float h, hashGridCellSize, hashGridCellSizeInv, simulationScale, simulationScaleInv;
  float xmin, xmax, ymin, ymax, zmin, zmax;
  float r0, mass, rho0, timeStep, viscosity, delta;
  float gravity_x, gravity_y, gravity_z;
  float surfTensCoeff; /* sphFluid.cl:662 coefficient, evaluated on the host in the reference's types */
  double Wpoly6Coefficient, gradWspikyCoefficient, del2WviscosityCoefficient;
  int32_t numOfElasticP, elasticOffset, muscleCount, numOfMembranes, maxIteration;
  int32_t threads; /* OpenMP threads used by every stage */
} sph_oracle_params;

typedef struct sph_oracle sph_oracle;

enum sph_oracle_stage {
  SO_CLEAR = 0, SO_HASH, SO_SORT, SO_SORTPOST, SO_INDEXX, SO_INDEXPOST, SO_FIND, SO_DENSITY, SO_FORCES, SO_ELASTIC,
  SO_PREDICTPOS, SO_PREDICTDENS, SO_CORRECTP, SO_PRESSUREFORCE, SO_INTEGRATE, SO_CLEARMEMB, SO_MEMB, SO_MEMBFIN,
  SO_STAGE_COUNT
};

sph_oracle* sph_oracle_create(const sph_oracle_params* p, const float* position, const float* velocity,
                              const float* elasticConnections, const int32_t* membraneData,
                              const int32_t* particleMembranesList);
void sph_oracle_destroy(sph_oracle* s);
int sph_oracle_run(sph_oracle* s, int stage);  /* one owOpenCLSolver::_run* */
int sph_oracle_step(sph_oracle* s);            /* one simulationStep() */
void sph_oracle_update_muscles(sph_oracle* s, const float* signal);
/* Buffer by reference name, reference layout. Returns bytes; *ptr stays owned by the oracle.
 * Names: position velocity sortedPosition sortedVelocity acceleration neighborMap particleIndex
 * particleIndexBack gridCellIndex gridCellIndexFixedUp pressure rho neighborIds(int32[N*32]) */
size_t sph_oracle_buffer(sph_oracle* s, const char* name, void** ptr);
/* seconds spent in each stage since creation (accumulated), SO_STAGE_COUNT doubles */
void sph_oracle_stage_seconds(sph_oracle* s, double* out);

/* Host-side constant evaluation in the reference's expression types
 * (owPhysicsConstant.h:12-76, owPhysicsFluidSimulator.cpp:164-203). */
float sph_oracle_surf_tens_coeff(double Wpoly6Coefficient, float h, float simulationScale);

#ifdef __cplusplus
}
#endif
#endif
