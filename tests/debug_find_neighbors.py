"""Debug helper (test infrastructure: uses the oracle as the checker): first particles whose neighbour lists differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, scenes
sc = scenes.liquid_box((8.0, 8.0, 8.0), (18, 18, 18), spacing_in_r0=0.45, origin_in_r0=(4.0, 4.0, 4.0))
N = sc["cfg"].particleCount
hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc, threads=8)
seq = scenes.STAGE_SEQUENCE[:scenes.STAGE_SEQUENCE.index("findNeighbors") + 1]
for st in seq:
    getattr(hip, scenes.HIP_STAGE_METHOD[st])(); ora.run(st)
a = hip.buffer("neighborIds").reshape(-1, 32); b = ora.buffer("neighborIds").reshape(-1, 32)
pi = ora.buffer("particleIndex").reshape(-1, 2)
bad = np.flatnonzero((a != b).any(1))
print("N", N, "bad particles", bad.size, "first", bad[:10])
c = hip.buffer("debugCounters"); print("counters", c[:5])
cells = pi[:, 0].astype(int)
occ = np.bincount(cells); print("max cell occupancy", occ.max(), "cells used", (occ > 0).sum())
print("bad cells", np.unique(cells[bad])[:20], "count per cell of bad", np.bincount(cells[bad])[np.unique(cells[bad])][:20])
for p in bad[:3]:
    print("p", p, "cell", cells[p], "got", a[p][:8], "want", b[p][:8], "n got", (a[p] >= 0).sum(), "n want", (b[p] >= 0).sum())
    print("   got cells", cells[a[p][a[p] >= 0]][:8], "want cells", cells[b[p][b[p] >= 0]][:8])
da = hip.buffer("neighborMap").reshape(-1, 32, 2)[:, :, 1]; db = ora.buffer("neighborMap").reshape(-1, 32, 2)[:, :, 1]
badd = np.flatnonzero((da.view(np.uint32) != db.view(np.uint32)).any(1))
print("particles with a distance mismatch", badd.size, badd[:20])
print("bad sorted ids mod 128:", bad % 128, " // 128:", bad // 128)
sp = ora.buffer("sortedPosition").reshape(-1, 4)[:N]
for p in bad[:6]:
    print("p", p, "type", sp[p, 3] if False else "", "pos", sp[p, :3], "cell", cells[p], "occ", occ[cells[p]])
    print("   got ", a[p]); print("   want", b[p])
    print("   dgot ", da[p][:12]); print("   dwant", db[p][:12])
