"""One-off (not collected by pytest): config #2 (or, with a second argument "config4", the 16.5 M-particle box of config #4) at full
size, 100 steps on the GPU and on the oracle, every 25 steps all positions / velocities / neighbour ids / densities / pressures
compared bit for bit. A third argument gives the lattice spacing in r0 (0.85: pressure-active from the first step; default 0.93).
python tests/long_run_parity.py [steps] [config2|config4] [spacing]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scenes
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
big = len(sys.argv) > 2 and sys.argv[2] == "config4"
spacing = float(sys.argv[3]) if len(sys.argv) > 3 else 0.93
sc = (scenes.liquid_box((78.0, 50.0, 470.0), (160, 100, 1000), mask=0xffffffff, spacing_in_r0=spacing) if big else
      scenes.liquid_box((50.0, 50.0, 50.0), (100, 100, 100), mask=0xffff, spacing_in_r0=spacing))
N = sc["cfg"].particleCount
hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc, threads=int(os.environ.get("ORACLE_THREADS", "16")))
t0 = time.time()
for it in range(steps):
    hip.step(it); ora.step()
    if (it + 1) % 25 == 0:
        ok = (scenes.bits_equal(hip.read_position_buffer(), ora.buffer("position").reshape(-1, 4)[:N])
              and scenes.bits_equal(hip.read_velocity_buffer(), ora.buffer("velocity").reshape(-1, 4)[:N])
              and np.array_equal(hip.buffer("neighborIds"), ora.buffer("neighborIds"))
              and scenes.bits_equal(hip.read_density_buffer(), ora.buffer("rho").reshape(-1)[:N])
              and scenes.bits_equal(hip.buffer("pressure").reshape(-1)[:N], ora.buffer("pressure").reshape(-1)[:N]))
        p = ora.buffer("pressure").reshape(-1)[:N]
        print("step %d: %s  | particles with p > 0: %d, max p %.3g, slow batches of K12 so far %d  (%.0f s)" % (
            it + 1, "bit-identical" if ok else "MISMATCH", int((p > 0).sum()), float(p.max()), int(hip.buffer("debugCounters")[8]), time.time() - t0), flush=True)
        if not ok:
            sys.exit(1)
print("ok")
