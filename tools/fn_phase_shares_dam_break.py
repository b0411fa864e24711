"""findNeighbors phase shares on an EVOLVED state (1 M-particle dam break after N steps), with the DIAGNOSTIC stamps build
(make variant NAME=stamps EXTRA=-DFN_STAMPS): tools/fn_phase_shares_dam_break.py [steps=1500]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ.setdefault("SPHMI_LIB", os.path.join(ROOT, "smoothed-particle-hydrodynamics_amd", "libsphmi_stamps.so"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
sc = scenes.liquid_box((120.0, 66.0, 56.0), (70, 120, 100), mask=0xffffffff, origin_in_r0=(3.0, 3.0, 3.0))
h = scenes.hip_for(sc)
names = ["head (bounds, run table)", "-", "-", "candidate loads + barrier", "setup", "walk", "expand", "bisect", "pass1+store", "exact walks"]
nwg = (sc["cfg"].particleCount + 127) // 128
for target in (0, steps):
    for it in range(target if target == 0 else 0, target):
        h.step(it)
    h.synchronize()
    h.set_stage_timing(True); h.reset_stage_times()  # (also clears the debug counters)
    before = h.buffer("debugCounters").astype(np.uint32)
    reps = 5
    for it in range(reps):
        h.step(target + it)
    h.synchronize()
    c = (h.buffer("debugCounters").astype(np.uint32) - before).astype(float)  # (the counters are 32-bit and wrap over a long run)
    ms, n = h.stage_times()["find_neighbors"]
    h.set_stage_timing(False)
    tot = c[16:26].sum()
    waves = reps * nwg * 4
    print("after %d steps: find_neighbors %.3f ms/launch (stamps build); per step: no-LDS-cell %d, list overflow %d, dropped runs %d"
          % (target, ms / n, c[0] / reps, c[1] / reps, c[3] / reps))
    for i, nm in enumerate(names):
        if nm != "-":
            print("  %-26s %5.1f %%  [%7.0f cycles per wave]" % (nm, 100 * c[16 + i] / max(tot, 1), 64 * c[16 + i] / waves))
