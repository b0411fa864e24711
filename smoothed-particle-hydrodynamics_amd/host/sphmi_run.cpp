// sphmi_run — headless driver shaped like owPhysicsFluidSimulator (src/owPhysicsFluidSimulator.cpp:27-149): loads
// configuration/position.txt + velocity.txt style files (or generates a synthetic box), constructs the solver through
// the owOpenCLSolver-compatible facade, and issues the reference's stage sequence with its per-stage timing printout.
//
//   sphmi_run --position P.txt --velocity V.txt [--steps N] [--staged] [--out positions.bin] [--quiet] [--blocking-readback]
//   sphmi_run --box 50 50 50 --lattice 100 100 100 [--wide] ...
//   sphmi_run --worm [--muscles] ...      the generated worm scene of the reference's default start-up (owHelper.cpp:709)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "owHIPSolver.h"

typedef owHIPSolver owOpenCLSolver;  // the one line a reference maintainer changes

static double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

struct Watch {  // owHelper::refreshTime / watch_report (owHelper.cpp:44-57,1806-1841)
  double t0, t1; bool quiet;
  void refresh() { t0 = t1 = now_ms(); }
  void report(const char* fmt) { double t = now_ms(); if (!quiet) printf(fmt, t - t1); t1 = t; }
  double elapsed() const { return t1 - t0; }
};

int main(int argc, char** argv) {
  const char *posFile = nullptr, *velFile = nullptr, *outFile = nullptr;
  int steps = 10; bool staged = false, wide = false, quiet = false, muscles = false, worm = false, blockingRead = false;
  double box[3] = {0, 0, 0}; int lat[3] = {0, 0, 0};
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--position") && i + 1 < argc) posFile = argv[++i];
    else if (!strcmp(argv[i], "--velocity") && i + 1 < argc) velFile = argv[++i];
    else if (!strcmp(argv[i], "--out") && i + 1 < argc) outFile = argv[++i];
    else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--box") && i + 3 < argc) { for (int k = 0; k < 3; k++) box[k] = atof(argv[++i]); }
    else if (!strcmp(argv[i], "--lattice") && i + 3 < argc) { for (int k = 0; k < 3; k++) lat[k] = atoi(argv[++i]); }
    else if (!strcmp(argv[i], "--staged")) staged = true;
    else if (!strcmp(argv[i], "--wide")) wide = true;
    else if (!strcmp(argv[i], "--quiet")) quiet = true;
    else if (!strcmp(argv[i], "--muscles")) muscles = true;
    else if (!strcmp(argv[i], "--worm")) worm = true;
    else if (!strcmp(argv[i], "--blocking-readback")) blockingRead = true;  // the reference's blocking read_position_buffer
    else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
  }
  try {
    sph_config cfg;
    sphmi_default_config(&cfg);
    std::vector<float> position_cpp, velocity_cpp, elasticConnectionsData_cpp;
    std::vector<int> membraneData_cpp, particleMembranesList_cpp;
    int numOfLiquidP = 0, numOfElasticP = 0, numOfBoundaryP = 0;
    if (posFile && velFile) {  // owHelper::preLoadConfiguration + loadConfiguration
      int n = sphmi_count_particles(posFile);
      if (n <= 0) throw std::runtime_error(std::string("Could not open file ") + posFile);
      cfg.particleCount = n;
      position_cpp.resize(4 * (size_t)n); velocity_cpp.resize(4 * (size_t)n);
      if (sphmi_load_configuration(posFile, velFile, n, position_cpp.data(), velocity_cpp.data(), &numOfLiquidP, &numOfElasticP, &numOfBoundaryP))
        throw std::runtime_error("could not load configuration");
      if (numOfElasticP) throw std::runtime_error("elastic particles need elasticconnections.txt, which the reference repository does not ship");
    } else if (worm) {  // owHelper::generateConfiguration, shipped box (owPhysicsFluidSimulator.cpp:36-53)
      int numOfMembranes = 0;
      if (sphmi_worm_counts(&cfg, 30.0, 20.0, 250.0, &numOfElasticP, &numOfLiquidP, &numOfBoundaryP, &numOfMembranes)) throw std::runtime_error("worm scene: bad configuration");
      cfg.particleCount = numOfElasticP + numOfLiquidP + numOfBoundaryP;
      cfg.numOfElasticP = numOfElasticP; cfg.numOfMembranes = numOfMembranes; cfg.elasticOffset = 0;
      position_cpp.resize(4 * (size_t)cfg.particleCount); velocity_cpp.resize(4 * (size_t)cfg.particleCount);
      elasticConnectionsData_cpp.resize((size_t)4 * SPH_MAX_NEIGHBOR_COUNT * numOfElasticP);
      membraneData_cpp.resize(3 * (size_t)numOfMembranes);
      particleMembranesList_cpp.resize((size_t)SPH_MAX_MEMBRANES_INCLUDING_SAME_PARTICLE * numOfElasticP);
      if (sphmi_generate_worm(&cfg, 30.0, 20.0, 250.0, position_cpp.data(), velocity_cpp.data(), elasticConnectionsData_cpp.data(),
                              membraneData_cpp.data(), particleMembranesList_cpp.data()))
        throw std::runtime_error("worm scene generation failed");
    } else if (box[0] > 0 && lat[0] > 0) {  // synthetic pure-liquid box, SURVEY 8(d)
      if (sphmi_config_set_box(&cfg, box[0], box[1], box[2], wide ? 0xffffffffu : 0xffffu)) throw std::runtime_error("bad box");
      if (sphmi_box_counts(&cfg, box[0], box[1], box[2], lat[0], lat[1], lat[2], &numOfLiquidP, &numOfBoundaryP)) throw std::runtime_error("bad lattice");
      cfg.particleCount = numOfLiquidP + numOfBoundaryP;
      position_cpp.resize(4 * (size_t)cfg.particleCount); velocity_cpp.resize(4 * (size_t)cfg.particleCount);
      const float sp = 0.93f * cfg.r0, o = 3.0f * cfg.r0;
      if (sphmi_generate_box(&cfg, box[0], box[1], box[2], lat[0], lat[1], lat[2], sp, o, o, o, 0.f, 20261004ull, position_cpp.data(), velocity_cpp.data()))
        throw std::runtime_error("box generation failed");
    } else {
      fprintf(stderr, "usage: sphmi_run (--position P --velocity V | --box X Y Z --lattice A B C | --worm [--muscles]) [--steps N] [--staged] [--wide] [--out F]\n");
      return 2;
    }
    printf("particles: %d (liquid %d, elastic %d, boundary %d), grid %d x %d x %d\n", cfg.particleCount, numOfLiquidP,
           numOfElasticP, numOfBoundaryP, cfg.gridCellsX, cfg.gridCellsY, cfg.gridCellsZ);
    owOpenCLSolver* ocl_solver = new owOpenCLSolver(cfg, position_cpp.data(), velocity_cpp.data(),
                                                    elasticConnectionsData_cpp.empty() ? nullptr : elasticConnectionsData_cpp.data(),
                                                    membraneData_cpp.empty() ? nullptr : membraneData_cpp.data(),
                                                    particleMembranesList_cpp.empty() ? nullptr : particleMembranesList_cpp.data());
    std::vector<float> muscle_activation_signal_cpp(cfg.muscleCount, 0.f);
    Watch helper; helper.quiet = quiet;
    double total = 0;
    for (int iterationCount = 0; iterationCount < steps; iterationCount++) {
      helper.refresh();
      if (!quiet) printf("\n[[ Step %d ]]\n", iterationCount);
      if (staged) {  // owPhysicsFluidSimulator.cpp:88-113, call for call
        ocl_solver->_runClearBuffers();       sph_synchronize(ocl_solver->handle()); helper.report("_runClearBuffers: \t%9.3f ms\n");
        ocl_solver->_runHashParticles();      sph_synchronize(ocl_solver->handle()); helper.report("_runHashParticles: \t%9.3f ms\n");
        ocl_solver->_runSort();               sph_synchronize(ocl_solver->handle()); helper.report("_runSort: \t\t%9.3f ms\n");
        ocl_solver->_runSortPostPass();       sph_synchronize(ocl_solver->handle()); helper.report("_runSortPostPass: \t%9.3f ms\n");
        ocl_solver->_runIndexx();             sph_synchronize(ocl_solver->handle()); helper.report("_runIndexx: \t\t%9.3f ms\n");
        ocl_solver->_runIndexPostPass();      sph_synchronize(ocl_solver->handle()); helper.report("_runIndexPostPass: \t%9.3f ms\n");
        ocl_solver->_runFindNeighbors();      sph_synchronize(ocl_solver->handle()); helper.report("_runFindNeighbors: \t%9.3f ms\n");
        ocl_solver->_run_pcisph_computeDensity();
        ocl_solver->_run_pcisph_computeForcesAndInitPressure();
        ocl_solver->_run_pcisph_computeElasticForces();
        int iter = 0;
        do {
          ocl_solver->_run_pcisph_predictPositions();
          ocl_solver->_run_pcisph_predictDensity();
          ocl_solver->_run_pcisph_correctPressure();
          ocl_solver->_run_pcisph_computePressureForceAcceleration();
          iter++;
        } while (iter < cfg.maxIteration);
        ocl_solver->_run_pcisph_integrate(iterationCount);
        sph_synchronize(ocl_solver->handle()); helper.report("_runPCISPH: \t\t%9.3f ms\t3 iteration(s)\n");
        ocl_solver->_run_clearMembraneBuffers();
        ocl_solver->_run_computeInteractionWithMembranes();
        ocl_solver->_run_computeInteractionWithMembranes_finalize();
      } else {
        ocl_solver->step(iterationCount);
        sph_synchronize(ocl_solver->handle()); helper.report("sph_step (fused): \t%9.3f ms\n");
      }
      // owPhysicsFluidSimulator.cpp:115. Default: the copy is only started here and overlaps the next step (the positions are
      // consumed after the loop); --blocking-readback waits for it like the reference does.
      if (blockingRead) ocl_solver->read_position_buffer(position_cpp.data());
      else ocl_solver->read_position_buffer_async(position_cpp.data());
      helper.report("_readBuffer: \t\t%9.3f ms\n");
      if (!quiet) printf("------------------------------------\n_Total_step_time:\t%9.3f ms\n------------------------------------\n", helper.elapsed());
      total += helper.elapsed();
      if (muscles) {  // signals computed after step t drive step t+1 (owPhysicsFluidSimulator.cpp:134-141)
        sphmi_muscle_signal(iterationCount, muscle_activation_signal_cpp.data(), cfg.muscleCount);
        ocl_solver->updateMuscleActivityData(muscle_activation_signal_cpp.data());
      }
    }
    {
      helper.refresh();
      ocl_solver->wait_position_buffer();  // the last step's copy
      helper.report("");
      total += helper.elapsed();
    }
    printf("%d steps, %.3f ms/step incl. the 16N-byte position read-back, %.3e particle-steps/s\n", steps, total / steps,
           cfg.particleCount * 1000.0 / (total / steps));
    if (outFile) {
      FILE* f = fopen(outFile, "wb");
      if (!f || fwrite(position_cpp.data(), sizeof(float), position_cpp.size(), f) != position_cpp.size()) throw std::runtime_error("cannot write --out file");
      fclose(f);
    }
    delete ocl_solver;
  } catch (std::exception& e) {  // owPhysicsFluidSimulator.cpp:73-76,144-148
    std::cout << "ERROR: " << e.what() << std::endl;
    exit(-1);
  }
  return 0;
}
