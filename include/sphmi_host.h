/* sphmi_host.h — C ABI of libsphmi_host.so: the host-side pieces either side of the device path that the
 * reference keeps in owPhysicsConstant.h, owOpenCLSolver.cpp:7-17, owPhysicsFluidSimulator.cpp:164-203 (calcDelta),
 * owHelper.cpp (text loaders, boundary-shell generator) and main_sim.py (muscle signals). Pure C++/libm, no GPU.
 */
#ifndef SPHMI_HOST_H
#define SPHMI_HOST_H

#include "sphmi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Physics constants exactly as the reference's compiler evaluates them (owPhysicsConstant.h:12-76),
 * delta from calcDelta (owPhysicsFluidSimulator.cpp:164-203), the shipped box 30h x 20h x 250h and its grid
 * (owOpenCLSolver.cpp:7-17), cellIdMask = 0xffff, no elastic matter, muscleCount = 100, maxIteration = 3. */
int sphmi_default_config(sph_config* cfg);

/* Replace the box: maxima given in units of h as *double* multipliers, evaluated like the reference's macros
 * `XMAX 30.0*h` (double product narrowed to float) and grid dims `(int)((MAX-MIN)/h)+1` (owOpenCLSolver.cpp:7-17). */
int sphmi_config_set_box(sph_config* cfg, double xmax_in_h, double ymax_in_h, double zmax_in_h, uint32_t cellIdMask);

/* owHelper::preLoadConfiguration (owHelper.cpp:1431-1458): number of particles in a position file, -1 on error. */
int sphmi_count_particles(const char* positionFile);
/* owHelper::loadConfiguration (owHelper.cpp:1460-1545), position + velocity part. Returns 0 or negative. */
int sphmi_load_configuration(const char* positionFile, const char* velocityFile, int count, float* position4N,
                             float* velocity4N, int* numOfLiquidP, int* numOfElasticP, int* numOfBoundaryP);
/* elasticconnections.txt reader (owHelper.cpp:1512-1540): rows of `jd rij0 val1 val2`; returns rows read. */
int sphmi_load_elastic_connections(const char* file, int numOfElasticP, float* out4x32xE);

/* Synthetic box of SURVEY §8(d): liquid lattice first, then the boundary shell exactly as
 * owHelper::generateConfiguration stage 1 (owHelper.cpp:717-719,770-928; normals in `velocity`, type 3).
 * Liquid: lx*ly*lz lattice, spacing `spacing`, origin (ox,oy,oz), type 1.1, zero velocity, optional uniform jitter
 * in [-jitter, jitter] per axis from PCG32(seed). The box maxima are given again as the double multipliers of h,
 * because the reference sizes the shell from the un-narrowed macros `XMAX 30.0*h` (owHelper.cpp:717-719).
 * sphmi_box_counts gives the sizes to allocate. */
int sphmi_box_counts(const sph_config* cfg, double xmax_in_h, double ymax_in_h, double zmax_in_h, int lx, int ly, int lz,
                     int* numOfLiquidP, int* numOfBoundaryP);
int sphmi_generate_box(const sph_config* cfg, double xmax_in_h, double ymax_in_h, double zmax_in_h, int lx, int ly, int lz,
                       float spacing, float ox, float oy, float oz, float jitter, uint64_t seed, float* position4N,
                       float* velocity4N);

/* The same box for one rank of a z-slab decomposition (bench.py --gpus N, sphmi/slab.py), without materialising the whole
 * scene per rank: _layer_histogram counts the particles per z cell layer (layer = (int)(z * hashGridCellSizeInv), as
 * hashParticles computes it, sphFluid.cl:199) into hist[0..layers), returning SPH_ERR_SIZE if a particle lies outside;
 * _slice writes only the particles of layers [layerLo, layerHi) — in generation order, i.e. ascending global id — together
 * with their global ids (index in the full scene). Call it with position == NULL first to get *count. Same particles, bit for
 * bit, as the corresponding rows of sphmi_generate_box. No reference counterpart (the reference is single-device). */
int sphmi_box_layer_histogram(const sph_config* cfg, double xmax_in_h, double ymax_in_h, double zmax_in_h, int lx, int ly, int lz,
                              float spacing, float ox, float oy, float oz, float jitter, uint64_t seed, int64_t* hist, int layers);
int sphmi_generate_box_slice(const sph_config* cfg, double xmax_in_h, double ymax_in_h, double zmax_in_h, int lx, int ly, int lz,
                             float spacing, float ox, float oy, float oz, float jitter, uint64_t seed, int layerLo, int layerHi,
                             float* position4n, float* velocity4n, uint32_t* globalIds, int capacity, int* count);

/* The worm scene owHelper::generateConfiguration builds (owHelper.cpp:104-1429): an elastic worm shell
 * (two radial layers, 199 cross-sections) with its triangulated membrane, springs to every elastic / boundary particle within
 * r0*sqrt(2.7) (rest length 0.95 of the distance; lengthwise springs between "green" shell particles carry one of 96 muscle
 * group colours), liquid inside the worm and on the floor of the box, and the boundary shell. Particle order: elastic,
 * liquid, boundary (elasticOffset = 0). Bit-identical to the compiled reference generator for the shipped box
 * (tests/golden/worm_input.npz). sphmi_worm_counts gives the sizes: position / velocity 4*(E+L+B) floats,
 * elasticConnections 4*32*E floats, membraneData 3*M ints, particleMembranesList 7*E ints. */
int sphmi_worm_counts(const sph_config* cfg, double xmax_in_h, double ymax_in_h, double zmax_in_h, int* numOfElasticP,
                      int* numOfLiquidP, int* numOfBoundaryP, int* numOfMembranes);
int sphmi_generate_worm(const sph_config* cfg, double xmax_in_h, double ymax_in_h, double zmax_in_h, float* position4N,
                        float* velocity4N, float* elasticConnections, int32_t* membraneData, int32_t* particleMembranesList);

/* owHelper::loadConfigurationToFile (owHelper.cpp:1640-1672), the `-l_to` trajectory dump: `<dir>/position_buffer.txt` gets
 * "numOfElasticP\nnumOfLiquidP\n" when firstIteration, then one `x\ty\tz\ttype` line per NON-boundary particle (appended
 * on later calls; the reference calls it every 10th step, owPhysicsFluidSimulator.cpp:121-129); on the first call also
 * `connection_buffer.txt` (32*numOfElasticP rows of 4 floats) and `membranes_buffer.txt` (count, then one row per
 * triangle). Numbers are formatted by operator<< with the stream defaults, as in the reference. Deviation: the reference
 * indexes the stride-3 membrane array with stride 4 (a bug, owHelper.cpp:1665); here each row is `i\tj\tk\t0`. */
int sphmi_save_configuration(const char* dir, const float* position4N, int count, int numOfElasticP, int numOfLiquidP,
                             const float* connections, const int32_t* membranes, int numOfMembranes, int firstIteration);

/* owHelper::loadConfigurationFromFile (owHelper.cpp:1674-1741): the `-l_from` playback input, i.e. the files
 * sphmi_save_configuration writes. _info gives the particle counts, the number of complete frames in position_buffer.txt and
 * the membrane count; _frame fills 4*(numOfElasticP+numOfLiquidP) floats; _connections 4*32*numOfElasticP floats;
 * _membranes 4*numOfMembranes ints (rows `i j k unused`, as the reference reads them). */
int sphmi_trajectory_info(const char* dir, int* numOfElasticP, int* numOfLiquidP, int* frames, int* numOfMembranes);
int sphmi_trajectory_frame(const char* dir, int frame, float* position);
int sphmi_trajectory_connections(const char* dir, int numOfElasticP, float* connections);
int sphmi_trajectory_membranes(const char* dir, int numOfMembranes, int32_t* membranes);

/* Muscle activation of main_sim.py:4-53 / PyramidalSimulation.cpp:68-93 in closed form:
 * 96 values for step t, entries 96..muscleCount-1 left 0. */
int sphmi_muscle_signal(int step, float* out, int muscleCount);

#ifdef __cplusplus
}
#endif
#endif
