"""ms/step and fallback counters along a dam break (disordered states), 1.1 M particles."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import scenes
sc = scenes.liquid_box((120.0, 50.0, 50.0), (80, 90, 100), mask=0xffffffff)
N = sc["cfg"].particleCount
h = scenes.hip_for(sc)
it = 0
for chunk in range(8):
    h.synchronize(); h.set_stage_timing(False); t0 = time.perf_counter()
    for _ in range(400):
        h.step(it); it += 1
    h.synchronize(); dt = (time.perf_counter() - t0) / 400 * 1e3
    h.set_stage_timing(True); h.reset_stage_times()
    for _ in range(10):
        h.step(it); it += 1
    h.synchronize()
    st = h.stage_times(); c = h.buffer("debugCounters")
    pos = h.read_position_buffer()
    print("step %5d  %.3f ms/step  fn %.3f  unstaged/step %d  overflow/step %d  retry-dropped %d  ymean %.1f xmax %.1f finite %s" % (
        it, dt, st["find_neighbors"][0] / 10, c[0] // 10, c[1] // 10, c[3] // 10, pos[:720000, 1].mean(), pos[:720000, 0].max(),
        bool(np.isfinite(pos).all())), flush=True)
