"""Time pcisph_computeDensity alone (staged API) after a few real steps; A/B builds via SPHMI_LIB."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import scenes
from profile_step import WORK  # noqa
name = sys.argv[1] if len(sys.argv) > 1 else "1M"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
box, lat, mask = WORK[name]
sc = scenes.liquid_box(box, lat, mask=mask)
h = scenes.hip_for(sc)
if not os.environ.get("SPHMI_NO_STEPS"):  # (diagnostic builds with invalid results must not integrate)
    for it in range(3): h.step(it)
h._runClearBuffers(); h._runHashParticles(); h._runSort(); h._runSortPostPass(); h._runIndexx(); h._runIndexPostPass(); h._runFindNeighbors()
for _ in range(3): h._run_pcisph_computeDensity()
h.synchronize(); h.set_stage_timing(True); h.reset_stage_times()
for _ in range(reps): h._run_pcisph_computeDensity()
h.synchronize()
ms, n = h.stage_times()["density"]
N = sc["cfg"].particleCount
us = ms / n * 1e3
print(os.path.basename(os.environ.get("SPHMI_LIB", "default")), name, "density us/launch %.2f  algorithmic GB/s %.0f  frac %.3f" % (us, N * 132 / us / 1e3, N * 132 / us / 1e3 / 8000))
