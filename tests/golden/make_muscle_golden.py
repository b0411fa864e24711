"""Generates tests/golden/muscle_signal.npz from the reference's own signal generator (src/main_sim.py:4-53).

Run in the build container only (needs /root/reference):   python tests/golden/make_muscle_golden.py

main_sim.py is Python 2: `j = n/2` is an int there and a float under Python 3, which numpy's linspace refuses as `num`.
The only accommodation made here is a linspace that casts `num` to int (what Python 2's integer division would have
passed); every arithmetic statement executed is the reference's own. The values are narrowed to float32 exactly as
PyramidalSimulation::unpackPythonList does (`float value = PyFloat_AsDouble(...)`, src/PyramidalSimulation.cpp:55-66).
"""
import importlib.util
import os
import sys

import numpy as np

REF = "/root/reference/src/main_sim.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "muscle_signal.npz")


def main():
    if not os.path.exists(REF):
        sys.exit("reference not present: fixtures can only be regenerated in the build container")
    real_linspace = np.linspace
    np.linspace = lambda start, stop, num=50, *a, **k: real_linspace(start, stop, int(num), *a, **k)
    try:
        spec = importlib.util.spec_from_file_location("main_sim", REF)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        steps = [0, 1, 2, 3, 10, 99, 1000, 31415, 62831, 100000]
        sim = mod.muscle_simulation()
        seq = [np.asarray(sim.run(), np.float64) for _ in range(4)]          # the object's own clock: t = 0, 1, 2, 3
        by_time = {}
        for t in steps:
            w1, w2 = mod.parallel_waves(time=t)
            by_time[t] = np.concatenate([np.asarray(list(w1), np.float64), np.asarray(list(w2), np.float64)])
    finally:
        np.linspace = real_linspace
    for i in range(4):  # run() == [W1, W2, W2, W1] of parallel_waves(time = i)
        w1, w2 = by_time[i][:24], by_time[i][24:]
        assert np.array_equal(seq[i], np.concatenate([w1, w2, w2, w1]))
    sig = np.stack([np.concatenate([by_time[t][:24], by_time[t][24:], by_time[t][24:], by_time[t][:24]]) for t in steps])
    np.savez(OUT, steps=np.asarray(steps, np.int64), signal_f64=sig, signal_f32=sig.astype(np.float32))
    print("wrote", OUT, sig.shape)


if __name__ == "__main__":
    main()
