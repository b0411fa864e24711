"""Soak: a 1 M-particle liquid column collapses for thousands of steps on the GPU; every CHECK steps the state is handed to the
oracle (TEST INFRASTRUCTURE) and one more step of both is compared word for word. Prints the fallback counters of findNeighbors.
  tools/soak_dam_break.py [steps=3000] [check=500]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import scenes
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
check = int(sys.argv[2]) if len(sys.argv) > 2 else 500
lat = (70, 120, 100)
sc = scenes.liquid_box((120.0, 66.0, 56.0), lat, mask=0xffffffff, origin_in_r0=(3.0, 3.0, 3.0))
cfg = sc["cfg"]
N, L = cfg.particleCount, lat[0] * lat[1] * lat[2]
print("particles", N, "liquid", L, flush=True)
hip = scenes.hip_for(sc)
y0, x0 = sc["position"][:L, 1].mean(), sc["position"][:L, 0].max()
t0 = time.time()
done = 0
while done < steps:
    for it in range(done, done + check):
        hip.step(it)
    done += check
    pos, vel = hip.read_position_buffer(), hip.read_velocity_buffer()
    assert np.all(np.isfinite(pos)) and np.all(np.isfinite(vel)), "state not finite at step %d" % done
    state = dict(sc, position=pos.copy(), velocity=vel.copy())
    ora = scenes.oracle_for(state, threads=16)
    hip.step(done)
    ora.step()
    done += 1
    got, want = scenes.canonical(hip.buffer, N), scenes.canonical(ora.buffer, N)
    bad = [k for k in want if k != "gridCellIndex" and not scenes.bits_equal(got[k], want[k])]
    c = hip.buffer("debugCounters")
    occ = np.bincount(want["particleIndex"].reshape(-1, 2)[:, 0].astype(np.int64))
    nn = (want["neighborIds"].reshape(-1, 32) >= 0).sum(1)
    print("step %5d: %s | mean y of the liquid %.1f (start %.1f), front x %.1f (start %.1f) | max cell occupancy %d, mean neighbours %.1f,"
          " full lists %d | fn counters: no-LDS-cell %d, list overflow %d, dropped runs %d | %.0f s"
          % (done, "bit-identical" if not bad else "DIFFERENT: %s" % bad, pos[:L, 1].mean(), y0, pos[:L, 0].max(), x0, occ.max(),
             nn[:L].mean(), int((nn == 32).sum()), c[0], c[1], c[3], time.time() - t0), flush=True)
    ora.close()
    if bad:
        sys.exit(1)
print("ok")
