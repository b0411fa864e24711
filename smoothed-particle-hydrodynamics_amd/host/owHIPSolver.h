// owHIPSolver.h — the reference's `owOpenCLSolver` (src/owOpenCLSolver.h:28-62) re-implemented over the C ABI of
// libsphmi.so. Same constructor, same method names, same return/exception behaviour, so owPhysicsFluidSimulator.cpp
// needs only `#include "owHIPSolver.h"` and `typedef owHIPSolver owOpenCLSolver;`.
//
// Differences that the C ABI makes explicit: the globals the reference reads (PARTICLE_COUNT, numOfElasticP, delta, box
// macros ...) arrive in a `sph_config`; `particleMembranesList_cpp` is NOT freed by the solver.
#pragma once
#include <stdexcept>
#include <string>

#include "sphmi.h"
#include "sphmi_host.h"

class owHIPSolver {
 public:
  owHIPSolver(const sph_config& cfg, const float* position_cpp, const float* velocity_cpp,
              const float* elasticConnectionsData_cpp = nullptr, const int* membraneData_cpp = nullptr,
              const int* particleMembranesList_cpp = nullptr)
      : cfg_(cfg), s_(nullptr) {
    // owOpenCLSolver.cpp:88-91: setup failures surface as std::exception ("ERROR: ..." + exit(-1) in the caller)
    check(sph_create(&cfg_, position_cpp, velocity_cpp, elasticConnectionsData_cpp, membraneData_cpp,
                     particleMembranesList_cpp, &s_), "sph_create");
  }
  ~owHIPSolver() { sph_destroy(s_); }
  owHIPSolver(const owHIPSolver&) = delete;
  owHIPSolver& operator=(const owHIPSolver&) = delete;

  // PCISPH kernels for data structures support and management — owOpenCLSolver.h:37-43. The reference returns the
  // cl_int of the enqueue (0 = OK) and callers ignore it; here: the sph_status.
  unsigned int _runClearBuffers() { return (unsigned)sph_run_clear_buffers(s_); }
  unsigned int _runHashParticles() { return (unsigned)sph_run_hash_particles(s_); }
  unsigned int _runSort() { return (unsigned)sph_run_sort(s_); }
  unsigned int _runSortPostPass() { return (unsigned)sph_run_sort_post_pass(s_); }
  unsigned int _runIndexx() { return (unsigned)sph_run_indexx(s_); }
  unsigned int _runIndexPostPass() { return (unsigned)sph_run_index_post_pass(s_); }
  unsigned int _runFindNeighbors() { return (unsigned)sph_run_find_neighbors(s_); }
  // PCISPH kernels for physics-related calculations — owOpenCLSolver.h:45-52
  unsigned int _run_pcisph_computeDensity() { return (unsigned)sph_run_pcisph_compute_density(s_); }
  unsigned int _run_pcisph_computeForcesAndInitPressure() { return (unsigned)sph_run_pcisph_compute_forces_and_init_pressure(s_); }
  unsigned int _run_pcisph_computeElasticForces() { return (unsigned)sph_run_pcisph_compute_elastic_forces(s_); }
  unsigned int _run_pcisph_predictPositions() { return (unsigned)sph_run_pcisph_predict_positions(s_); }
  unsigned int _run_pcisph_predictDensity() { return (unsigned)sph_run_pcisph_predict_density(s_); }
  unsigned int _run_pcisph_correctPressure() { return (unsigned)sph_run_pcisph_correct_pressure(s_); }
  unsigned int _run_pcisph_computePressureForceAcceleration() { return (unsigned)sph_run_pcisph_compute_pressure_force_acceleration(s_); }
  unsigned int _run_pcisph_integrate(int iterationCount) { return (unsigned)sph_run_pcisph_integrate(s_, iterationCount); }
  // owOpenCLSolver.h:54-56
  unsigned int _run_clearMembraneBuffers() { return (unsigned)sph_run_clear_membrane_buffers(s_); }
  unsigned int _run_computeInteractionWithMembranes() { return (unsigned)sph_run_compute_interaction_with_membranes(s_); }
  unsigned int _run_computeInteractionWithMembranes_finalize() { return (unsigned)sph_run_compute_interaction_with_membranes_finalize(s_); }
  // owOpenCLSolver.h:58
  unsigned int updateMuscleActivityData(float* _muscle_activation_signal_cpp) {
    return (unsigned)sph_update_muscles(s_, _muscle_activation_signal_cpp, cfg_.muscleCount);
  }
  // owOpenCLSolver.h:60-62 — copy failures throw std::runtime_error there (owOpenCLSolver.cpp:727-736)
  void read_position_buffer(float* position_cpp) { check(sph_read_position(s_, position_cpp), "read_position_buffer"); }
  // The same read without the wait (sph_read_position_async): the 16N-byte copy runs on its own stream under the NEXT step's
  // kernels and position_cpp is complete after wait_position_buffer() — call that where the data is consumed (the reference:
  // owPhysicsFluidSimulator::getPosition_cpp(), which the viewer calls once per frame). position_cpp is page-locked in place on
  // first use. INTEGRATION.md §1 shows the two lines.
  void read_position_buffer_async(float* position_cpp) { check(sph_read_position_async(s_, position_cpp), "read_position_buffer_async"); }
  void wait_position_buffer() { check(sph_read_position_wait(s_), "wait_position_buffer"); }
  void read_density_buffer(float* density_cpp) { check(sph_read_density(s_, density_cpp), "read_density_buffer"); }
  void read_particleIndex_buffer(unsigned int* particleIndexBuffer) {
    check(sph_read_particle_index(s_, particleIndexBuffer), "read_particleIndex_buffer");
  }

  // beyond the reference: the whole stage sequence of simulationStep() as one call, and per-stage device timing
  unsigned int step(int iterationCount) { return (unsigned)sph_step(s_, iterationCount); }
  sph_solver* handle() { return s_; }
  const sph_config& config() const { return cfg_; }

 private:
  static void check(int rc, const char* what) {
    if (rc != SPH_OK) throw std::runtime_error(std::string(what) + ": " + sph_last_error());
  }
  sph_config cfg_;
  sph_solver* s_;
};
