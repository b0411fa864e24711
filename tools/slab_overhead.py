"""Time pack + rebuild (no communication) on one GPU: the per-step cost the slab decomposition adds besides RCCL."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import scenes
from sphmi import slab as S
sc = scenes.liquid_box((50.0, 50.0, 50.0), (100, 100, 100), mask=0xffffffff)
cfg = sc["cfg"]; n = cfg.particleCount
lay = S.particle_layers(sc["position"], cfg)
slab = S.make_slab([int(lay.min()), int(lay.max()) + 1], 0, 1, n)
be = S.HipSlabBackend(cfg, sc["position"], sc["velocity"], np.arange(n, dtype=np.uint32), slab)
dd = S.SlabDecomposition(be, 0, 1)
for it in range(3): dd.step(it)
be.solver.synchronize()
t0 = time.perf_counter()
for it in range(20): be.step(it)
be.solver.synchronize(); t1 = time.perf_counter()
for it in range(20): dd.step(it)
be.solver.synchronize(); t2 = time.perf_counter()
print("step only %.3f ms, step + pack/rebuild %.3f ms -> overhead %.3f ms" % ((t1 - t0) * 50, (t2 - t1) * 50, (t2 - t1 - t1 + t0) * 50))
