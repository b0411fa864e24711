"""Wall time per fused step on small scenes (launch-bound regime): tools/small_scene_step_time.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import scenes
for name, sc in (("tiny 3k", scenes.SCENES["tiny"]()), ("config1 61k", scenes.config1()),
                 ("cube 250k", scenes.liquid_box((32.0, 32.0, 32.0), (60, 60, 60)))):
    h = scenes.hip_for(sc)
    for it in range(20): h.step(it)
    h.synchronize()
    t0 = time.perf_counter()
    K = 200
    for it in range(K): h.step(20 + it)
    h.synchronize()
    wall = (time.perf_counter() - t0) / K * 1e3
    h.set_stage_timing(True); h.reset_stage_times()
    for it in range(50): h.step(300 + it)
    h.synchronize()
    st = h.stage_times()
    dev = sum(ms / max(n, 1) for ms, n in st.values() if n)
    print("%-12s N=%7d  wall %.4f ms/step   sum of stage device times %.4f ms" % (name, sc["cfg"].particleCount, wall, dev))
