"""Time ONE staged kernel alone (it is re-run on the same state, nothing is integrated, so diagnostic builds whose results are
invalid cannot drive the scene into a non-finite state): tools/time_stage.py forces|predict_density|pressure_force [reps] [16M]."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import scenes
stage = sys.argv[1] if len(sys.argv) > 1 else "pressure_force"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
big = len(sys.argv) > 3 and sys.argv[3] == "16M"
sc = scenes.liquid_box((78.0, 50.0, 470.0), (160, 100, 1000), mask=0xffffffff) if big else scenes.liquid_box((50.0, 50.0, 50.0), (100, 100, 100))
h = scenes.hip_for(sc)
# (no h.step() here: a diagnostic build would integrate its invalid forces, and a non-finite state makes the search quadratic)
h._runClearBuffers(); h._runHashParticles(); h._runSort(); h._runSortPostPass(); h._runIndexx(); h._runIndexPostPass(); h._runFindNeighbors()
h._run_pcisph_computeDensity(); h._run_pcisph_computeForcesAndInitPressure(); h._run_pcisph_predictPositions()
h._run_pcisph_predictDensity(); h._run_pcisph_correctPressure(); h._run_pcisph_computePressureForceAcceleration()
call = {"forces": h._run_pcisph_computeForcesAndInitPressure, "predict_density": h._run_pcisph_predictDensity,
        "pressure_force": h._run_pcisph_computePressureForceAcceleration}[stage]
for _ in range(3): call()
h.synchronize(); h.set_stage_timing(True); h.reset_stage_times()
for _ in range(reps): call()
h.synchronize()
ms, n = h.stage_times()[stage]
print("%-28s %s ms/launch %.4f" % (os.path.basename(os.environ.get("SPHMI_LIB", "default")), stage, ms / n))
