"""Profiling driver: a few fused steps of one workload, nothing else (use under rocprofv3; see profiles/README.md)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import scenes  # noqa: E402

WORK = {"1M": ((50.0, 50.0, 50.0), (100, 100, 100), 0xffff), "4M": ((80.0, 80.0, 80.0), (160, 160, 160), 0xffffffff),
        "8M": ((100.0, 100.0, 100.0), (200, 200, 200), 0xffffffff),
        "16M": ((78.0, 50.0, 470.0), (160, 100, 1000), 0xffffffff),
        "64M": ((240.0, 200.0, 310.0), (250, 400, 640), 0xffffffff)}

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "1M"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    box, lat, mask = WORK[name]
    sc = scenes.liquid_box(box, lat, mask=mask)
    h = scenes.hip_for(sc)
    for it in range(steps):
        h.step(it)
    h.synchronize()
    print("done", name, sc["cfg"].particleCount, steps)


if __name__ == "__main__":
    main()
