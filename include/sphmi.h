/* sphmi.h — C ABI of libsphmi.so, the MI355X (gfx950) PCISPH / Electrofluid step solver.
 *
 * Drop-in boundary: this library replaces the reference's C++ class `owOpenCLSolver`
 * (/root/reference/src/owOpenCLSolver.h:28-62), which is the only seam between the step
 * orchestrator owPhysicsFluidSimulator::simulationStep() (src/owPhysicsFluidSimulator.cpp:79-149)
 * and the device kernels (src/sphFluid.cl). Each entry point names the reference member it replaces.
 * Plain pointers and sizes only; no C++/torch types cross this boundary. All functions return
 * 0 on success or a negative sph_status; nothing throws. sph_last_error() gives the text.
 *
 * The reference keeps its inputs in mutable globals (PARTICLE_COUNT, gridCells*, numOf*, delta, the
 * box macros — owOpenCLSolver.h:14-17, owOpenCLSolver.cpp:7-23, owPhysicsConstant.h); here they are
 * the explicit `sph_config`.
 *
 * After each sph_run_* the solver holds what the reference's buffers would hold after the same
 * _run* call; sph_read_buffer() exports any of them in the reference's own
 * layout and index space, so stage-by-stage parity can be checked. Internally the data is laid out
 * for CDNA4, not as the reference's AoS buffers.
 */
#ifndef SPHMI_H
#define SPHMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPHMI_ABI_VERSION 2
#define SPH_MAX_PARTICLES ((1 << 27) - 1) /* capacity ceiling of one solver: the tiled neighbour map is addressed with 32-bit
                                             element indices (32 slots x 2^27 particles = 2^32); sph_create rejects more */
#define SPH_MAX_NEIGHBOR_COUNT 32 /* owOpenCLConstant.h:4 */
#define SPH_MAX_MEMBRANES_INCLUDING_SAME_PARTICLE 7 /* owOpenCLConstant.h:6 */
#define SPH_LIQUID_PARTICLE 1   /* owOpenCLConstant.h:8-10 */
#define SPH_ELASTIC_PARTICLE 2
#define SPH_BOUNDARY_PARTICLE 3

typedef enum sph_status {
  SPH_OK = 0,
  SPH_ERR_INVALID = -1,  /* bad argument / configuration the kernels cannot represent */
  SPH_ERR_HIP = -2,      /* a HIP runtime call failed (no GPU, out of memory, launch failure) */
  SPH_ERR_ORDER = -3,    /* stage called before the stage it depends on */
  SPH_ERR_UNKNOWN_BUFFER = -4,
  SPH_ERR_SIZE = -5
} sph_status;

/* Everything the reference passes as kernel arguments or globals. Field names follow the reference. */
typedef struct sph_config {
  int32_t abi_version;      /* SPHMI_ABI_VERSION */
  int32_t particleCount;    /* PARTICLE_COUNT */
  int32_t capacity;         /* buffers are sized for this many particles (slab decomposition: the count varies per step);
                               0 = particleCount. At most SPH_MAX_PARTICLES (2^27 - 1 = 134,217,727) */
  int32_t gridCellsX, gridCellsY, gridCellsZ, gridCellCount; /* owOpenCLSolver.cpp:14-17 */
  uint32_t cellIdMask;      /* 0xffff = reference behaviour (sphFluid.cl:229,377); 0xffffffff = wide */
  float h, hashGridCellSize, hashGridCellSizeInv, simulationScale, simulationScaleInv; /* owPhysicsConstant.h:19-24 */
  float xmin, xmax, ymin, ymax, zmin, zmax;                                           /* owOpenCLSolver.cpp:7-12 */
  float r0, mass, rho0, timeStep, viscosity, delta;                                   /* owPhysicsConstant.h:12-27,63,75 */
  float gravity_x, gravity_y, gravity_z;                                              /* owPhysicsConstant.h:72-74 */
  float surfTensCoeff;      /* sphFluid.cl:662, evaluated once on the host in the reference's types */
  double Wpoly6Coefficient, gradWspikyCoefficient, del2WviscosityCoefficient;         /* owPhysicsConstant.h:69-71 */
  int32_t numOfElasticP;    /* rows of elasticConnectionsData */
  int32_t elasticOffset;    /* numOfBoundaryP*(!generateInitialConfiguration), owOpenCLSolver.cpp:435 */
  int32_t muscleCount;      /* MUSCLE_COUNT (owWorldSimulation.cpp:31) */
  int32_t numOfMembranes;
  int32_t maxIteration;     /* owPhysicsConstant.h:76 */
  int32_t device;           /* HIP device ordinal */
  void* stream;             /* hipStream_t to launch on; NULL = the solver creates its own */
} sph_config;

typedef struct sph_solver sph_solver;

/* owOpenCLSolver::owOpenCLSolver(position_cpp, velocity_cpp, elasticConnectionsData_cpp, membraneData_cpp,
 * particleMembranesList_cpp) — owOpenCLSolver.cpp:27-92. Host arrays are copied; the caller keeps ownership
 * of all five (unlike the reference, particleMembranesList is NOT freed here). position/velocity: 4*N floats
 * (x,y,z,type) / (vx,vy,vz,w); elastic: 4*32*numOfElasticP floats; membranes: 3*numOfMembranes ints;
 * particleMembranesList: 7*numOfElasticP ints. */
int sph_create(const sph_config* cfg, const float* position, const float* velocity, const float* elasticConnections,
               const int32_t* membraneData, const int32_t* particleMembranesList, sph_solver** out);
/* owOpenCLSolver::~owOpenCLSolver — owOpenCLSolver.cpp:744-748 */
int sph_destroy(sph_solver* s);

/* One per owOpenCLSolver::_run* (owOpenCLSolver.h:37-56, owOpenCLSolver.cpp:213-687), same order contract as
 * simulationStep(). Asynchronous on the solver's stream. */
int sph_run_clear_buffers(sph_solver* s);                          /* _runClearBuffers            :213 */
int sph_run_hash_particles(sph_solver* s);                         /* _runHashParticles           :229 */
int sph_run_sort(sph_solver* s);                                   /* _runSort (host qsort there) :255 */
int sph_run_sort_post_pass(sph_solver* s);                         /* _runSortPostPass            :262 */
int sph_run_indexx(sph_solver* s);                                 /* _runIndexx                  :284 */
int sph_run_index_post_pass(sph_solver* s);                        /* _runIndexPostPass (host)    :305 */
int sph_run_find_neighbors(sph_solver* s);                         /* _runFindNeighbors           :320 */
int sph_run_pcisph_compute_density(sph_solver* s);                 /* _run_pcisph_computeDensity  :357 */
int sph_run_pcisph_compute_forces_and_init_pressure(sph_solver* s);/* ..._computeForcesAndInitPressure :386 */
int sph_run_pcisph_compute_elastic_forces(sph_solver* s);          /* ..._computeElasticForces    :420 */
int sph_run_pcisph_predict_positions(sph_solver* s);               /* ..._predictPositions        :454 */
int sph_run_pcisph_predict_density(sph_solver* s);                 /* ..._predictDensity          :490 */
int sph_run_pcisph_correct_pressure(sph_solver* s);                /* ..._correctPressure         :519 */
int sph_run_pcisph_compute_pressure_force_acceleration(sph_solver* s); /* ..._computePressureForceAcceleration :549 */
int sph_run_pcisph_integrate(sph_solver* s, int iterationCount);   /* _run_pcisph_integrate       :649 */
int sph_run_clear_membrane_buffers(sph_solver* s);                 /* _run_clearMembraneBuffers   :583 */
int sph_run_compute_interaction_with_membranes(sph_solver* s);     /* _run_computeInteractionWithMembranes :602 */
int sph_run_compute_interaction_with_membranes_finalize(sph_solver* s); /* ..._finalize           :628 */

/* The whole stage sequence of owPhysicsFluidSimulator::simulationStep() (owPhysicsFluidSimulator.cpp:88-113) as one
 * call on the solver's stream, with the fusions of DESIGN.md §4. Same results as calling the sph_run_* above in order. */
int sph_step(sph_solver* s, int iterationCount);

/* owOpenCLSolver::updateMuscleActivityData — owOpenCLSolver.cpp:738-742 (n must equal muscleCount) */
int sph_update_muscles(sph_solver* s, const float* signal, int n);

/* owOpenCLSolver::read_position_buffer / read_density_buffer / read_particleIndex_buffer — owOpenCLSolver.h:60-62.
 * Blocking. position: 4N floats, orig order. density: N floats, SORTED order (as the reference).
 * particleIndex: 2N uints (cell, origId), sorted. velocity has no reference getter; provided for checks. */
int sph_read_position(sph_solver* s, float* out4N);
int sph_read_velocity(sph_solver* s, float* out4N);
int sph_read_density(sph_solver* s, float* outN);
int sph_read_particle_index(sph_solver* s, uint32_t* out2N);

/* read_position_buffer without the wait. The reference ends every simulationStep() with a blocking 16N-byte read
 * (owPhysicsFluidSimulator.cpp:115, owOpenCLSolver.h:60); here the copy can run on a stream of its own under the NEXT step's
 * search and PCISPH stages: sph_read_position_async returns at once, the positions of the steps enqueued so far land in
 * out4N in the background, and the next kernel that writes positions (integrate, the last kernel of a step) waits for the
 * copy on the device. sph_read_position_wait blocks until the last requested copy has landed (and reports a blown-up state
 * like sph_read_position); a second sph_read_position_async waits for the previous one first. out4N must stay valid until
 * then. For the copy to overlap, out4N must be DMA-able: memory that is already pinned (hipHostMalloc / hipHostRegister) is
 * used as it is, any other buffer is page-locked in place with hipHostRegister on first use (kept until sph_destroy or
 * sph_host_unregister — call that before freeing the buffer); if that fails the copy goes through a pinned staging area and
 * sph_read_position_wait does the final memcpy. */
int sph_read_position_async(sph_solver* s, float* out4N);
int sph_read_position_wait(sph_solver* s);
int sph_host_unregister(sph_solver* s, void* buffer);

/* Test/inspection export in the reference's layout. Names: position[2N f4] velocity[2N f4]
 * sortedPosition[2N f4] sortedVelocity[N f4] acceleration[2N f4] neighborMap[32N f2] neighborIds[32N i32]
 * particleIndex[N u2] particleIndexBack[N u32] gridCellIndex[G+1 u32] gridCellIndexFixedUp[G+1 u32]
 * pressure[N f32] rho[2N f32]. `bytes` must equal the buffer's size (query with out == NULL: returns the size
 * through *needed). Blocking. */
int sph_read_buffer(sph_solver* s, const char* name, void* out, size_t bytes, size_t* needed);
/* The neighbour lists of the SORTED particles [first, first + count) only (a 64 M-particle map is 16 GB): ids[count][32]
 * (sorted index of the neighbour, -1 = empty slot: neighborMap[].x of the reference, sphFluid.cl:170) and dist[count][32]
 * (neighborMap[].y: distance * simulationScale, -1 = empty). Either pointer may be NULL. Blocking. */
int sph_read_neighbor_rows(sph_solver* s, int32_t first, int32_t count, int32_t* ids, float* dist);

int sph_synchronize(sph_solver* s);

/* Per-stage device timing with hipEvents on the solver's stream (the reference prints per-stage wall time,
 * owPhysicsFluidSimulator.cpp:88-115). Stage ids = sph_stage. Times accumulate until reset. */
typedef enum sph_stage {
  SPH_ST_HASH = 0, SPH_ST_SORT, SPH_ST_SORT_POST, SPH_ST_INDEX, SPH_ST_FIND_NEIGHBORS, SPH_ST_DENSITY,
  SPH_ST_FORCES, SPH_ST_ELASTIC, SPH_ST_PREDICT_DENSITY, SPH_ST_PRESSURE_FORCE, SPH_ST_INTEGRATE, SPH_ST_MEMBRANES,
  SPH_ST_COUNT
} sph_stage;
int sph_set_stage_timing(sph_solver* s, int enable);
int sph_get_stage_times(sph_solver* s, double* ms_total, int64_t* launches, int n); /* n = SPH_ST_COUNT */
int sph_reset_stage_times(sph_solver* s);
/* Radix passes (8- or 9-bit digits) the sort of sph_step() takes for this solver: 24 algorithmic bytes per particle each. */
int sph_step_sort_passes(sph_solver* s);

/* ---- Spatial decomposition (no reference counterpart: the reference is single-device) -----------------
 * The global box is cut into z-slabs of whole cell layers, one solver (one GPU, one process) per slab. A solver holds the
 * particles of its own layers [layerLo, layerHi) plus `ghostLayers` layers on each side and advances ALL of them with the
 * ordinary sph_step(); only the particles it owned when the set was last rebuilt are authoritative afterwards. Once per
 * step the authoritative particles near a cut are sent to the neighbouring solver (sph_slab_pack -> the caller's
 * RCCL send/recv -> sph_slab_rebuild), which replaces its whole ghost zone with them. ghostLayers = 4 cell layers (8h)
 * cover the 6 neighbour hops (6 x 31h/30) that one PCISPH step propagates information, so owned particles get bit-identical
 * results to a single-solver run; the local arrays are kept sorted by global id so that the within-cell order (ascending
 * orig id) is the global one. Requires cellIdMask = 0xffffffff.
 * Message = n records of 9 words: position (x,y,z,type), velocity (vx,vy,vz,w), global id (or 7 words, see
 * sph_slab_set_record_format). All pointers below are DEVICE pointers on the solver's device. */
typedef struct sph_slab {
  int32_t layerLo, layerHi;   /* owned cell layers along z: cz = (int)(z * hashGridCellSizeInv) */
  int32_t ghostLayers;        /* W */
  int32_t hasLower, hasUpper; /* a neighbouring slab exists below / above */
  int32_t globalIdBits;       /* bit length of the largest global id */
} sph_slab;
#define SPH_SLAB_RECORD_WORDS 9
#define SPH_SLAB_COMPACT_WORDS 7 /* x, y, z, vx, vy, vz, global id: see sph_slab_set_record_format */
int sph_slab_init(sph_solver* s, const sph_slab* slab, const uint32_t* globalIds /* host, particleCount entries */);
/* counts[0] = authoritative particles kept, counts[1] / counts[2] = records written to msgDown / msgUp (each has room for
 * `capRecords`), every list in ascending global-id order (order-preserving compaction of the sorted local set).
 * Blocking (the counts come back to the host). */
int sph_slab_pack(sph_solver* s, void* msgDown, void* msgUp, int32_t capRecords, int32_t counts[3]);
/* The same, for callers that ship a message as ONE transfer `[count word | payload | padding]` (sphmi/slab.py): frame buffers
 * have room for 1 + 9*capRecords words; word 0 receives the number of payload words (9 x records), written on the device, and
 * the payload starts at word 1 — the frame can be handed to RCCL as it is. */
int sph_slab_pack_framed(sph_solver* s, void* frameDown, void* frameUp, int32_t capRecords, int32_t counts[3]);
/* New local set = kept + nDown records received from below + nUp from above, sorted by global id (three-way merge: the
 * received messages must be in ascending global-id order, as sph_slab_pack writes them; a message that is not makes the
 * next sph_slab_pack fail with SPH_ERR_INVALID). */
int sph_slab_rebuild(sph_solver* s, const void* recvDown, int32_t nDown, const void* recvUp, int32_t nUp);
/* Overlapped variant of "sph_step, then sph_slab_pack_framed": sph_slab_step_begin enqueues the whole step with its last stage
 * (pressure force + integrate) running first on the owned layers next to the cuts, packs the two frames as soon as those are
 * integrated, and returns without waiting; sph_slab_step_messages blocks only until the frames are ready and gives the
 * record counts (down, up) — the caller can hand the frames to RCCL while the rest of the step is still running; the following
 * sph_slab_rebuild waits for the step and takes the kept count from it. capRecords overflow is reported by sph_slab_step_messages. */
int sph_slab_step_begin(sph_solver* s, int iterationCount, void* frameDown, void* frameUp, int32_t capRecords);
int sph_slab_step_messages(sph_solver* s, int32_t counts[2]);
/* The same rebuild with NO host round trip: frameDown / frameUp are complete received frames `[payload words | payload]` with room
 * for capXRecords records (NULL where there is no neighbour), the kept count is taken where the preceding pack left it on the
 * device. Everything is enqueued and the call returns; the new particle count reaches the host with sph_slab_rebuild_finish
 * (blocking; also done implicitly by the next call that needs it): counts = kept, records from below, from above, and
 * counts[3] = 1 if NOTHING was merged because a frame announced more records than its buffer holds — fetch the rest and call
 * sph_slab_rebuild with the complete messages. A total beyond the solver's capacity is SPH_ERR_SIZE. */
int sph_slab_rebuild_framed(sph_solver* s, const void* frameDown, int32_t capDownRecords, const void* frameUp, int32_t capUpRecords);
int sph_slab_rebuild_finish(sph_solver* s, int32_t counts[4]);
/* Boundary particles never move: every solver keeps the boundary particles of its ghost layers from sph_slab_init on, and no
 * message ever carries one. The other particles' records can drop two more words: *typeBits = the position.w bit pattern common
 * to all non-boundary particles this solver was created with, provided their velocity.w is +0 (neither value ever changes);
 * 0 = it holds none, 0xffffffff = not uniform. When EVERY rank reports the same pattern (or 0), all of them may call
 * sph_slab_set_record_format(s, SPH_SLAB_COMPACT_WORDS, pattern) before the first pack: records are then 7 words
 * (x, y, z, vx, vy, vz, global id), 28 instead of 36 bytes. Default: SPH_SLAB_RECORD_WORDS. */
int sph_slab_liquid_signature(sph_solver* s, uint32_t* typeBits);
int sph_slab_set_record_format(sph_solver* s, int32_t recordWords, uint32_t typeBits);
/* Make the solver's stream wait for a hipEvent_t recorded on another stream (e.g. behind the RCCL receive on the caller's
 * communication stream) — stream-to-stream, the host does not wait. */
int sph_stream_wait_event(sph_solver* s, void* hipEvent);
int sph_particle_count(sph_solver* s);
/* Blocking read of the local set in its current order: positions, velocities (4 floats each), global ids and the
 * ownership flag (1 = in the owned layers when the set was last rebuilt); arrays sized sph_particle_count(). */
int sph_slab_read(sph_solver* s, float* position4, float* velocity4, uint32_t* globalIds, uint32_t* owned);

const char* sph_last_error(void);
int sph_abi_version(void);
/* What the loaded binary was built as: "libsphmi gfx950 abi 2" for the product build; diagnostic (timing-only, INVALID results)
 * variants add the names of their switches, each containing "DIAG". */
const char* sph_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* SPHMI_H */
