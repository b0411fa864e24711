"""Where a findNeighbors wave spends its cycles: run the DIAGNOSTIC build (make variant NAME=stamps EXTRA=-DFN_STAMPS, selected
with SPHMI_LIB) on the config #2 cube and print the share of every phase (s_memtime stamps summed over waves; read shares, not
lengths — the stamps forbid overlaps the real kernel has)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ.setdefault("SPHMI_LIB", os.path.join(ROOT, "smoothed-particle-hydrodynamics_amd", "libsphmi_stamps.so"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sc = scenes.liquid_box((50.0, 50.0, 50.0), (100, 100, 100), mask=0xffff)
h = scenes.hip_for(sc)
h._runClearBuffers(); h._runHashParticles(); h._runSort(); h._runSortPostPass(); h._runIndexx(); h._runIndexPostPass()
h._runFindNeighbors(); h.synchronize(); h.reset_stage_times()
for _ in range(reps): h._runFindNeighbors()
h.synchronize()
c = h.buffer("debugCounters").astype(float)
names = ["batch bounds", "cell-table window", "run table", "candidate loads", "setup", "walk", "expand", "bisect", "pass1+store", "exact walks+barrier"]
tot = c[16:26].sum()
waves = reps * ((sc["cfg"].particleCount + 127) // 128) * 8
print("phase shares (cycles per wave in brackets):")
for i, n in enumerate(names):
    print("  %-22s %5.1f %%  [%7.0f]" % (n, 100 * c[16 + i] / max(tot, 1), 64 * c[16 + i] / waves))
print("  total cycles per wave %.0f" % (64 * tot / waves))
