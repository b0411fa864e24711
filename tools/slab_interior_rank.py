"""What one step costs an INTERIOR rank of the weak-scaling run, measured on one GPU.

Three z-slabs of the 3x config-#2 column live in one process as three HipSlabBackends; frames are handed over by device
copies (no RCCL), the two outer slabs are advanced and synchronised first, and only the middle slab's step -> frames ->
(emulated transfer: device copies of the neighbours' frames + an optional host delay) -> rebuild is timed. The same bits as a
real 3-rank run (the exchange logic is sphmi/slab.py's, minus the transport), so the particle counts, ghost trimming and the
overlapped tail are the real ones.   usage: slab_interior_rank.py [steps] [host_delay_us]
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch
import scenes
from sphmi import slab as S

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
delay = float(sys.argv[2]) * 1e-6 if len(sys.argv) > 2 else 150e-6
overlap = os.environ.get("SPHMI_SLAB_OVERLAP", "1") != "0"
world = 3
# default: three slabs of the config #2 column (weak scaling). SPHMI_SLAB_BOX="78,50,176.25" SPHMI_SLAB_LATTICE="160,100,375" is
# three eighths of the config #4 box: the middle slab then carries what an interior rank of `bench.py --gpus 8` carries.
BOX = tuple(float(v) for v in os.environ.get("SPHMI_SLAB_BOX", "50,50,%g" % (50.0 * world)).split(","))
LAT = tuple(int(v) for v in os.environ.get("SPHMI_SLAB_LATTICE", "100,100,%d" % (100 * world)).split(","))
sc = scenes.liquid_box(BOX, LAT, mask=0xffffffff)
n_global = sc["cfg"].particleCount
layers = S.particle_layers(sc["position"], sc["cfg"])
cuts = S.balanced_cuts(layers, world)
backs = []
for r in range(world):
    cfg = scenes.liquid_box(BOX, (2, 2, 2), mask=0xffffffff)["cfg"]  # a config object per solver
    slab = S.make_slab(cuts, r, world, n_global)
    idx = S.local_indices(layers, slab)
    backs.append(S.HipSlabBackend(cfg, sc["position"][idx], sc["velocity"][idx], idx, slab))
sigs = {b.liquid_signature() for b in backs} - {0}
compact = len(sigs) == 1 and 0xffffffff not in sigs and os.environ.get("SPHMI_SLAB_FULL_RECORDS", "0") == "0"
if compact:  # what SlabDecomposition agrees on across ranks: 7-word records
    for b in backs:
        b.set_record_format(7, next(iter(sigs)))
REC = backs[0].record_words
framed = os.environ.get("SPHMI_SLAB_SYNC_EXCHANGE", "0") == "0"  # rebuild with device-side counts (no host round trip before the merge)
mid = backs[1]
print("record words", REC, "| rebuild:", "framed, counts on the device" if framed else "host counts")
print("local particles per slab:", [b.count for b in backs], "owned ~", n_global // world, "overlap", overlap)


def produce(b, it):
    if overlap:
        return b.step_and_pack_framed(it)
    b.step(it)
    return b.pack_framed()


def received(frames, r, whole=False):
    """(from lower, from upper) payload views for slab r out of everybody's frames (whole: the frames [count | payload])."""
    a = 0 if whole else 1
    lo = None if r == 0 else frames[r - 1][3][a:1 + frames[r - 1][4]]      # lower neighbour's UP frame
    up = None if r == world - 1 else frames[r + 1][1][a:1 + frames[r + 1][2]]  # upper neighbour's DOWN frame
    return lo, up


times = []
parts = []
for it in range(steps + 3):
    frames = [None] * world
    for r in (0, 2):
        frames[r] = produce(backs[r], it)
    for r in (0, 2):
        backs[r].solver.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames[1] = produce(mid, it)
    tp = time.perf_counter()
    lo, up = received(frames, 1, whole=framed)
    lo, up = lo.clone(), up.clone()          # the transfer: two device copies of the size RCCL would move ...
    if os.environ.get("SPHMI_DEBUG_SORTED"):
        for nm, t in (("lo", lo), ("up", up)):
            g = t.cpu().numpy().view(np.uint32)[(1 if framed else 0):].reshape(-1, REC)[:, REC - 1].astype(np.int64)
            print(it, nm, g.size, bool(np.all(np.diff(g) > 0)), flush=True)
    torch.cuda.current_stream().synchronize()
    if delay:
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < delay:  # ... plus the host-side latency of issuing and completing it
            pass
    tr = time.perf_counter()
    if framed:   # enqueued behind the transfer without the host knowing any count; the count is collected once, at the end
        mid.wait_for(torch.cuda.current_stream())
        mid.rebuild_framed(lo, up)
        assert not mid.rebuild_finish()[3]
    else:
        mid.rebuild(lo, up)
    mid.solver.synchronize()
    dt = time.perf_counter() - t0
    if it >= 3:
        times.append(dt)
        parts.append((tp - t0, tr - tp, time.perf_counter() - tr))
    for r in (0, 2):
        a, b = received(frames, r)
        a, b = (None if a is None else a.clone()), (None if b is None else b.clone())
        torch.cuda.current_stream().synchronize()  # the copies run on torch's stream, the rebuild on the solver's
        backs[r].rebuild(a, b)
if os.environ.get("SPHMI_STAGES"):  # per-stage device times of the middle slab (plain path), a few more steps
    mid.solver.set_stage_timing(True); mid.solver.reset_stage_times()
    k = 5
    for it2 in range(steps + 3, steps + 3 + k):
        frames = [produce(backs[r], it2) if r != 1 else None for r in range(world)]
        mid.step(it2)
        frames[1] = mid.pack_framed()
        torch.cuda.synchronize()
        for r in range(world):
            a, b = received(frames, r)
            a, b = (None if a is None else a.clone()), (None if b is None else b.clone())
            torch.cuda.current_stream().synchronize()
            backs[r].rebuild(a, b)
    st = mid.solver.stage_times()
    print("  stages (ms/step):", {n: round(ms / k, 4) for n, (ms, c) in st.items() if c})
times = np.array(times) * 1e3
print("  until frames ready %.3f ms, transfer emulation %.3f ms, rebuild incl. wait for the step %.3f ms" % tuple(np.array(parts).mean(0) * 1e3))
print("  bytes per message to the middle slab: %d + %d" % (4 * lo.numel(), 4 * up.numel()))
print("interior rank: %.3f ms/step (p50 %.3f) for %d local / %d owned particles, emulated transfer delay %.0f us" % (
    times.mean(), np.median(times), mid.count, n_global // world, delay * 1e6))
