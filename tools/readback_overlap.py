"""Does the asynchronous position read-back (sph_read_position_async) slow the kernels it runs under? Per-stage device times of
the config #4 box with and without a 264 MB copy in flight, and the wall time per step of both loops.
  tools/readback_overlap.py [steps=20] [16M|1M]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import scenes
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
big = not (len(sys.argv) > 2 and sys.argv[2] == "1M")
sc = scenes.liquid_box((78.0, 50.0, 470.0), (160, 100, 1000), mask=0xffffffff) if big else scenes.liquid_box((50.0, 50.0, 50.0), (100, 100, 100))
N = sc["cfg"].particleCount
h = scenes.hip_for(sc)
if len(sys.argv) > 3 and sys.argv[3] == "pinned":   # hipHostMalloc'ed by torch instead of page-locked in place by hipHostRegister
    import torch
    keep = torch.empty((N, 4), dtype=torch.float32, pin_memory=True)
    host = keep.numpy()
else:
    host = np.empty((N, 4), np.float32)
print("destination:", "torch pinned (hipHostMalloc)" if len(sys.argv) > 3 and sys.argv[3] == "pinned" else "numpy, page-locked in place (hipHostRegister)")
it = 0
for _ in range(5):
    h.step(it); it += 1
h.read_position_buffer_async(host); h.wait_position_buffer(); h.synchronize()


def loop(copy, timing):
    global it
    h.set_stage_timing(timing); h.reset_stage_times(); h.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        h.step(it); it += 1
        if copy:
            h.read_position_buffer_async(host)
    t1 = time.perf_counter()
    if copy:
        h.wait_position_buffer()
    h.synchronize()
    t2 = time.perf_counter()
    st = h.stage_times() if timing else {}
    h.set_stage_timing(False)
    return (t2 - t0) * 1e3 / steps, (t1 - t0) * 1e3 / steps, {k: round(ms / steps, 4) for k, (ms, n) in st.items() if n}


for copy in (False, True, False, True):
    wall, host_ms, _ = loop(copy, False)
    print("copy in flight %-5s  wall %.3f ms/step (host returned from the loop after %.3f ms/step)" % (copy, wall, host_ms), flush=True)
for copy in (False, True):
    wall, _, st = loop(copy, True)
    print("copy in flight %-5s  stage-timed wall %.3f ms/step, sum of stages %.3f: %s" % (copy, wall, sum(st.values()), st), flush=True)
