"""Fuzz (TEST INFRASTRUCTURE: the oracle is the checker): seeded random scenes — box shape, lattice density, jitter, cell-id
mode, blobs of hundreds of particles, particles on cell faces — HIP against the oracle, bit for bit.
  tools/fuzz_parity.py [first_seed=100] [count=40] [max_lattice=60]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import scenes
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
maxlat = int(sys.argv[3]) if len(sys.argv) > 3 else 60
bad_total = 0
t0 = time.time()
import sphmi
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    if seed % 5 == 4:  # every fifth scene: an elastic sheet with springs, a muscle row and membranes inside a liquid lattice
        sheet = (int(rng.integers(3, 22)), int(rng.integers(3, 22)))
        box = (float(max(8, (sheet[0] + 9) // 2 + rng.integers(0, 6))), float(rng.integers(8, 14)), float(max(8, (sheet[1] + 9) // 2 + rng.integers(0, 6))))
        lattice = tuple(int(max(2, (2.0 * b - 7.0) / 0.93 * rng.uniform(0.5, 1.0))) for b in box)
        muscles = bool(rng.integers(0, 2))
        sc = scenes.elastic_sheet_box(box, lattice, sheet, muscles)
        N = sc["cfg"].particleCount
        hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc, threads=16)
        bad = []
        for it in range(6):
            hip.step(it)
            ora.step()
            sig = sphmi.muscle_signal(it)
            hip.updateMuscleActivityData(sig)
            ora.update_muscles(sig)
            got, want = scenes.canonical(hip.buffer, N), scenes.canonical(ora.buffer, N)
            bad += [k for k in want if k != "gridCellIndex" and not scenes.bits_equal(got[k], want[k])]
        print("seed %4d N %7d box %s lattice %s elastic sheet %s muscles %s : %s"
              % (seed, N, box, lattice, sheet, muscles, "ok" if not bad else "DIFFERENT %s" % sorted(set(bad))), flush=True)
        bad_total += bool(bad)
        hip.close(); ora.close()
        continue
    box = tuple(float(rng.integers(8, 40)) for _ in range(3))
    spacing = float(rng.choice([0.5, 0.6, 0.8, 0.93, 1.1, 1.4]))
    lattice = tuple(int(max(2, min(maxlat, (2.0 * b - 7.0) / spacing * rng.uniform(0.5, 1.0)))) for b in box)
    wide = bool(rng.integers(0, 2))
    blobs = bool(rng.integers(0, 2))
    sc = scenes.liquid_box(box, lattice, spacing_in_r0=spacing, jitter_in_r0=float(rng.uniform(0.0, 0.45)),
                           mask=0xffffffff if wide else 0xffff, origin_in_r0=(4.0, 4.0, 4.0), seed=seed)
    cfg = sc["cfg"]
    N, nl = cfg.particleCount, sc["numOfLiquidP"]
    pos = sc["position"].copy()
    lo, hi = pos[:nl, :3].min(0), pos[:nl, :3].max(0)
    if blobs:
        for _ in range(int(rng.integers(1, 6))):
            centre = rng.uniform(lo, hi).astype(np.float32)
            members = rng.choice(nl, size=min(nl // 4, int(rng.integers(100, 3000))), replace=False)
            pos[members, :3] = centre + rng.normal(0.0, rng.uniform(0.3, 2.0) * cfg.r0, size=(members.size, 3)).astype(np.float32)
        pos[:nl, :3] = np.clip(pos[:nl, :3], lo, hi)
    cell = np.float32(cfg.hashGridCellSize)
    on_face = rng.choice(nl, size=min(nl, 256), replace=False)
    axis = rng.integers(0, 3, size=on_face.size)
    pos[on_face, axis] = np.maximum(np.round(pos[on_face, axis] / cell), 1.0).astype(np.float32) * cell
    sc["position"] = pos
    hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc, threads=16)
    bad = []
    if blobs:
        for st in scenes.STAGE_SEQUENCE[:scenes.STAGE_SEQUENCE.index("computeDensity") + 1]:
            getattr(hip, scenes.HIP_STAGE_METHOD[st])()
            ora.run(st)
        got, want = scenes.canonical(hip.buffer, N), scenes.canonical(ora.buffer, N)
        bad = [k for k in ("particleIndex", "gridCellIndexFixedUp", "neighborIds", "neighborDist", "rho") if not scenes.bits_equal(got[k], want[k])]
    else:
        for it in range(2):
            hip.step(it)
            ora.step()
            got, want = scenes.canonical(hip.buffer, N), scenes.canonical(ora.buffer, N)
            bad += [k for k in want if k != "gridCellIndex" and not scenes.bits_equal(got[k], want[k])]
    c = hip.buffer("debugCounters")
    occ = np.bincount(want["particleIndex"].reshape(-1, 2)[:, 0].astype(np.int64)).max()
    print("seed %4d N %7d box %s lattice %s spacing %.2f %s %s max occupancy %4d fn counters %d/%d/%d : %s"
          % (seed, N, box, lattice, spacing, "wide" if wide else "ref16", "blobs" if blobs else "steps", occ, c[0], c[1], c[3],
             "ok" if not bad else "DIFFERENT %s" % bad), flush=True)
    bad_total += bool(bad)
    hip.close(); ora.close()
print("%d scenes, %d different, %.0f s" % (count, bad_total, time.time() - t0))
sys.exit(1 if bad_total else 0)
