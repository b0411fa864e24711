"""Time the fused step on one of the fixture scenes (config1 / worm / any tests.scenes.SCENES entry) with stage timing."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import scenes, sphmi
name = sys.argv[1] if len(sys.argv) > 1 else "worm"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
sc = scenes.worm_scene() if name == "worm" else scenes.config1() if name == "config1" else scenes.SCENES[name]()
sim = sphmi.owPhysicsFluidSimulator(sc["cfg"], sc["position"], sc["velocity"], sc["elastic"], sc["membranes"],
                                    sc["particle_membranes"], muscles=sc["elastic"] is not None)
h = sim.ocl_solver
for _ in range(5): sim.simulationStep(read_back=False)
h.synchronize()
t0 = time.perf_counter()
for _ in range(steps): sim.simulationStep(read_back=False)
h.synchronize(); dt = time.perf_counter() - t0
h.set_stage_timing(True); h.reset_stage_times()  # second pass: per-stage events (kept out of the step time above)
for _ in range(steps): sim.simulationStep(read_back=False)
h.synchronize()
st = h.stage_times(); c = h.buffer("debugCounters")
N = sc["cfg"].particleCount
print(name, "N", N, "ms/step %.3f" % (dt * 1e3 / steps), "particle-steps/s %.3e" % (N * steps / dt))
print({k: round(ms / steps, 4) for k, (ms, n) in st.items() if n})
print("particles per step left to the in-kernel exact walk: unstaged %d, list overflow %d" % (c[0] // steps, c[1] // steps))
