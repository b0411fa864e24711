"""Time findNeighbors alone (staged API) on the config #2 cube: hash/sort/index once, then the search `reps` times.
Used for A/B builds (SPHMI_LIB=...), including ablations whose neighbour maps are not valid and must not be consumed."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import scenes
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
jitter = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
big = len(sys.argv) > 3 and sys.argv[3] == "16M"   # the config #4 box instead of the config #2 cube
sc = (scenes.liquid_box((78.0, 50.0, 470.0), (160, 100, 1000), mask=0xffffffff, jitter_in_r0=jitter) if big else
      scenes.liquid_box((50.0, 50.0, 50.0), (100, 100, 100), mask=0xffff, jitter_in_r0=jitter))
h = scenes.hip_for(sc)
h._runClearBuffers(); h._runHashParticles(); h._runSort(); h._runSortPostPass(); h._runIndexx(); h._runIndexPostPass()
for _ in range(3): h._runFindNeighbors()
h.synchronize(); h.set_stage_timing(True); h.reset_stage_times()
for _ in range(reps): h._runFindNeighbors()
h.synchronize()
ms, n = h.stage_times()["find_neighbors"]
print(os.environ.get("SPHMI_LIB", "default"), "find_neighbors ms/launch %.4f" % (ms / n))
