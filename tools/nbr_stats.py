import sys; import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import scenes, sphmi, numpy as np
sc = scenes.liquid_box((50.0,50.0,50.0),(100,100,100))
h = scenes.hip_for(sc)
h.reset_stage_times()
h.step(0); h.synchronize()
c = h.buffer('debugCounters'); N = sc['cfg'].particleCount
print('N', N, 'fallback particles: cell not staged', c[0], ', list overflow', c[1], '; candidate runs dropped', c[3])
nm = h.buffer('neighborIds').reshape(-1,32); print('mean nbrs', (nm>=0).sum(1).mean())
