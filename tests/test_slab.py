"""Multi-GPU path: slab decomposition + halo exchange of sphmi/slab.py.

CPU: two gloo ranks advance their slabs with the oracle as numerical backend; the union of the owned particles must be
bit-identical to a single-domain oracle run — this checks the decomposition itself (cuts, 4-layer ghost zone, ownership
hand-over, global-id ordering, message format). GPU (-m gpu): the same with libsphmi's pack / rebuild kernels and
HBM-resident messages, two ranks sharing the one card of the test box (gloo, host-staged transport; RCCL needs one GPU per
rank and is exercised by bench.py --gpus N)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import scenes
import sphmi
from sphmi import slab as S

HERE = os.path.dirname(os.path.abspath(__file__))
STEPS = 4


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(backend, world, out, steps=STEPS, env=None):
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "slab_worker.py"), "--rank", str(r), "--world", str(world),
                               "--port", str(port), "--backend", backend, "--steps", str(steps), "--out", str(out)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              env=dict(os.environ, **(env or {}))) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(world)]


def single_domain_reference(steps=STEPS, long_scene=False, max_iteration=None, env=None):
    sys.path.insert(0, HERE)
    import slab_worker
    extra = dict(env or {})
    if long_scene:
        extra["SPHMI_TEST_LONG_SCENE"] = "1"
    if max_iteration:
        extra["SPHMI_TEST_MAXITER"] = str(max_iteration)
    extra = {k: v for k, v in extra.items() if k.startswith("SPHMI_TEST_")}
    os.environ.update(extra)
    try:
        sc = slab_worker.scene()
    finally:
        for k in extra:
            os.environ.pop(k, None)
    o = scenes.oracle_for(sc, threads=8)
    for _ in range(steps):
        o.step()
    n = sc["cfg"].particleCount
    return sc, o.buffer("position").reshape(-1, 4)[:n], o.buffer("velocity").reshape(-1, 4)[:n]


def check_union(results, sc, pos_ref, vel_ref):
    n = sc["cfg"].particleCount
    gid = np.concatenate([r["gid"] for r in results])
    assert gid.size == n and np.unique(gid).size == n, "owned sets must partition the particles (%d of %d)" % (gid.size, n)
    pos = np.concatenate([r["pos"] for r in results])
    vel = np.concatenate([r["vel"] for r in results])
    order = np.argsort(gid)
    assert scenes.bits_equal(pos[order], pos_ref), scenes.diff_report(pos[order], pos_ref)
    assert scenes.bits_equal(vel[order], vel_ref), scenes.diff_report(vel[order], vel_ref)
    # the halo really carried particles, and both ranks hold more than they own (ghost layers)
    assert all(int(r["sent"]) > 0 for r in results)
    assert all(r["counts"][-1] > r["gid"].size for r in results)


def test_partition_helpers():
    sc = scenes.liquid_box((8.0, 8.0, 60.0), (12, 10, 110), mask=0xffffffff)
    lay = S.particle_layers(sc["position"], sc["cfg"])
    cuts = S.balanced_cuts(lay, 2)
    assert len(cuts) == 3 and cuts[1] - cuts[0] >= 8 and cuts[2] - cuts[1] >= 8
    slabs = [S.make_slab(cuts, r, 2, sc["cfg"].particleCount) for r in range(2)]
    idx = [S.local_indices(lay, s) for s in slabs]
    owned = [(lay[i] >= s.layerLo) & (lay[i] < s.layerHi) for i, s in zip(idx, slabs)]
    assert sum(int(o.sum()) for o in owned) == sc["cfg"].particleCount  # owned sets partition the scene
    assert all(i.size > o.sum() for i, o in zip(idx, owned))            # plus ghosts
    assert np.all(np.diff(idx[0].astype(np.int64)) > 0)                 # ascending global id
    with pytest.raises(ValueError):
        S.balanced_cuts(lay, 16)                                        # slabs thinner than 2 * GHOST_LAYERS
    # cost balance: an interior slab pays for two ghost zones, an end slab for one -> the ends get more layers, and the largest
    # cost (owned + 0.75 x ghosts) is never above what equal owned counts would give
    def worst(cuts):
        out = []
        for r in range(len(cuts) - 1):
            own = int(((lay >= cuts[r]) & (lay < cuts[r + 1])).sum())
            gh = (int(((lay >= cuts[r] - S.GHOST_LAYERS) & (lay < cuts[r])).sum()) if r else 0) + \
                 (int(((lay >= cuts[r + 1]) & (lay < cuts[r + 1] + S.GHOST_LAYERS)).sum()) if r < len(cuts) - 2 else 0)
            out.append(own + S.GHOST_COST * gh)
        return max(out)
    c3, e3 = S.balanced_cuts(lay, 3), S.balanced_cuts(lay, 3, ghost_cost=0.0)
    assert worst(c3) <= worst(e3)
    assert (c3[2] - c3[1]) <= min(c3[1] - c3[0], c3[3] - c3[2])  # the middle slab is the thinnest


def test_two_rank_halo_exchange_matches_single_domain_cpu(tmp_path):
    results = run_ranks("oracle", 2, tmp_path)
    sc, pos_ref, vel_ref = single_domain_reference()
    check_union(results, sc, pos_ref, vel_ref)
    # framing: the first exchange needs a second transfer (no agreed bound yet), every later step exactly one
    assert all(int(r["transfers"]) == STEPS + 1 for r in results)
    assert all(int(r["record_words"]) == sphmi.SLAB_COMPACT_WORDS for r in results)  # generated scene: one type word, velocity.w = 0


def test_record_formats_cpu(tmp_path):
    """Full 9-word records on request, and automatically when one non-boundary particle has velocity.w != 0 (every rank must
    agree: the odd particle lives on one rank only). Same bits as the single domain either way; compact records send fewer bytes."""
    sc, pos_ref, vel_ref = single_domain_reference()
    dirs = [tmp_path / name for name in ("compact", "full", "odd")]  # (np.load is lazy: every run keeps its own files)
    for d in dirs:
        d.mkdir()
    compact = run_ranks("oracle", 2, dirs[0])
    full = run_ranks("oracle", 2, dirs[1], env={"SPHMI_SLAB_FULL_RECORDS": "1"})
    check_union(full, sc, pos_ref, vel_ref)
    assert all(int(r["record_words"]) == sphmi.SLAB_RECORD_WORDS for r in full)
    assert all(int(c["sent"]) < int(f["sent"]) for c, f in zip(compact, full))
    odd = run_ranks("oracle", 2, dirs[2], env={"SPHMI_TEST_ODD_W": "1"})
    sc, pos_ref, vel_ref = single_domain_reference(env={"SPHMI_TEST_ODD_W": "1"})
    check_union(odd, sc, pos_ref, vel_ref)
    assert all(int(r["record_words"]) == sphmi.SLAB_RECORD_WORDS for r in odd)


def test_backend_owned_frames_are_sent_in_place_cpu(tmp_path):
    """The zero-copy path of the exchange (frames owned by the backend and sent as they are — what HipSlabBackend does under
    RCCL), exercised here with CPU frames; also with bounds smaller than the payload (remainder taken from inside the frame)."""
    sc, pos_ref, vel_ref = single_domain_reference(steps=3)
    for extra in ({}, {"SPHMI_TEST_BOUND_WORDS": str(4 * sphmi.SLAB_RECORD_WORDS)}):
        results = run_ranks("oracle", 3, tmp_path, steps=3, env=dict({"SPHMI_TEST_FRAMED": "1"}, **extra))
        check_union(results, sc, pos_ref, vel_ref)


def test_halo_payload_larger_than_the_agreed_bound_cpu(tmp_path):
    """Bounds forced to 4 records: every step takes the remainder path (count word says more than fits), results unchanged."""
    results = run_ranks("oracle", 2, tmp_path, steps=3, env={"SPHMI_TEST_BOUND_WORDS": str(4 * sphmi.SLAB_RECORD_WORDS)})
    sc, pos_ref, vel_ref = single_domain_reference(steps=3)
    check_union(results, sc, pos_ref, vel_ref)
    assert all(int(r["transfers"]) == 2 * 3 for r in results)


@pytest.mark.parametrize("env", [{}, {"SPHMI_TEST_BOUND_WORDS": "36"}, {"SPHMI_TEST_VZ": "0.6"}])
def test_asynchronous_exchange_protocol_cpu(tmp_path, env):
    """The protocol of the RCCL path — frames sent at their agreed length without the host knowing a count, rebuild from the frames,
    counts collected at the start of the next step, a frame that was too short completed by a second transfer there — run over
    gloo with the oracle backend offering the same four calls as HipSlabBackend (the GPU suite runs it on real RCCL)."""
    steps = 5
    results = run_ranks("oracle", 3, tmp_path, steps=steps, env=dict(env, SPHMI_TEST_ASYNC="1"))
    sc, pos_ref, vel_ref = single_domain_reference(steps=steps, env=env)
    check_union(results, sc, pos_ref, vel_ref)
    assert all(bool(r["asynchronous"]) for r in results)
    expect = 2 * steps if "SPHMI_TEST_BOUND_WORDS" in env else steps + 1
    assert all(int(r["transfers"]) == expect for r in results)


def test_three_rank_halo_exchange_matches_single_domain_cpu(tmp_path):
    results = run_ranks("oracle", 3, tmp_path, steps=3)
    sc, pos_ref, vel_ref = single_domain_reference(steps=3)
    check_union(results, sc, pos_ref, vel_ref)


DRIFT = {"SPHMI_TEST_VZ": "0.6"}  # 1.5 units = 0.22 cell layers per step along z: particles cross the cuts within a few steps


def test_ownership_changes_hands_cpu(tmp_path):
    """The liquid drifts across the cuts: particles that started in one slab end up owned by its neighbour (kept as ghost on one
    side, received as owned on the other); the union is still bit-identical to the single domain."""
    results = run_ranks("oracle", 3, tmp_path, steps=5, env=DRIFT)
    sc, pos_ref, vel_ref = single_domain_reference(steps=5, env=DRIFT)
    check_union(results, sc, pos_ref, vel_ref)
    assert sum(int(r["adopted"]) for r in results) > 20, "no particle changed owner: the scene does not test the hand-over"


def test_eight_rank_halo_exchange_matches_single_domain_cpu(tmp_path):
    """World size 8 (the node the benchmark targets) on a long thin box: six interior ranks with two neighbours each and two end
    ranks, nine-layer slabs (one interior layer between the two 4-layer cut zones), liquid drifting across the cuts."""
    env = dict(DRIFT, SPHMI_TEST_EIGHT_SLABS="1")
    results = run_ranks("oracle", 8, tmp_path, steps=4, env=env)
    sc, pos_ref, vel_ref = single_domain_reference(steps=4, env=env)
    check_union(results, sc, pos_ref, vel_ref)
    assert sum(int(r["adopted"]) for r in results) > 20
    assert len(results[0]["cuts"]) == 9


def test_balanced_cuts_on_the_benchmark_boxes():
    """balanced_cuts on the real layer histograms of BASELINE configs #4 (16.5 M box, 235 occupied layers) and #5 (64 M dam break,
    155 layers) at world 2 / 4 / 8: every rank's cost (owned + 0.75 x ghost particles) within 5 % of the mean, slabs at least
    2 x GHOST_LAYERS thick, cuts ascending and covering the occupied range."""
    for box, lattice in (((78.0, 50.0, 470.0), (160, 100, 1000)), ((240.0, 200.0, 310.0), (250, 400, 640))):
        sc = scenes.liquid_box(box, lattice, mask=0xffffffff)
        lay = S.particle_layers(sc["position"], sc["cfg"])
        del sc
        lo, hi = int(lay.min()), int(lay.max()) + 1
        hist = np.bincount(lay - lo, minlength=hi - lo).astype(np.int64)
        del lay
        cum = np.concatenate([[0], np.cumsum(hist)])
        for world in (2, 4, 8):
            cuts = S.balanced_cuts_hist(hist, lo, world)
            assert cuts[0] == lo and cuts[-1] == hi and all(b - a >= 2 * S.GHOST_LAYERS for a, b in zip(cuts, cuts[1:]))
            cnt = lambda a, b: int(cum[min(max(b, lo), hi) - lo] - cum[min(max(a, lo), hi) - lo]) if b > a else 0
            cost = []
            for r in range(world):
                a, b = cuts[r], cuts[r + 1]
                gh = (cnt(a - S.GHOST_LAYERS, a) if r else 0) + (cnt(b, b + S.GHOST_LAYERS) if r < world - 1 else 0)
                cost.append(cnt(a, b) + S.GHOST_COST * gh)
            assert max(cost) / (sum(cost) / world) <= 1.05, (box, world, cuts, cost)


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", ["1", "0"])
def test_ownership_changes_hands_gpu(tmp_path, overlap):
    """The same drift through libsphmi's pack / merge kernels, with the overlapped tail (its 'W + 1 layers next to the cut' rule)
    and with the plain step-then-pack path."""
    env = dict(DRIFT, SPHMI_SLAB_OVERLAP=overlap)
    results = run_ranks("hip", 3, tmp_path, steps=5, env=env)
    sc, pos_ref, vel_ref = single_domain_reference(steps=5, env=env)
    check_union(results, sc, pos_ref, vel_ref)
    assert sum(int(r["adopted"]) for r in results) > 20


@pytest.mark.gpu
def test_particle_faster_than_one_layer_per_step_is_reported():
    """The halo depth assumes < 1 cell layer of motion per step: a particle that breaks it makes the next pack fail loudly."""
    sc = scenes.liquid_box((8.0, 8.0, 40.0), (12, 10, 60), mask=0xffffffff)
    cfg = sc["cfg"]
    n = cfg.particleCount
    lay = S.particle_layers(sc["position"], cfg)
    lo, hi = int(lay.min()), int(lay.max()) + 1
    top = int(lay[sc["position"][:, 3].astype(int) == 1].max())  # highest liquid layer; the box above it is empty
    assert top + 4 < hi
    slab = S.make_slab([lo, top + 4, hi], 0, 2, n)  # rank 0 of 2: owns the liquid and three empty layers above it
    idx = S.local_indices(lay, slab)
    vel = sc["velocity"][idx].copy()
    liquid = np.flatnonzero(sc["position"][idx, 3].astype(int) == 1)
    fast = int(liquid[np.argmax(sc["position"][idx][liquid, 2])])  # on the free top face of the liquid block: nothing in its way
    vel[fast, 2] = 6.0  # 6.0 * timeStep * simulationScaleInv = 14.8 units = 2.2 cell layers in one step
    be = S.HipSlabBackend(cfg, sc["position"][idx], vel, idx, slab)
    be.step(0)
    with pytest.raises(sphmi.SphError, match="more than one cell layer"):
        be.pack()


@pytest.mark.gpu
def test_two_rank_halo_exchange_matches_single_domain_gpu(tmp_path):
    results = run_ranks("hip", 2, tmp_path)
    sc, pos_ref, vel_ref = single_domain_reference()
    check_union(results, sc, pos_ref, vel_ref)


@pytest.mark.gpu
def test_three_rank_halo_exchange_matches_single_domain_gpu(tmp_path):
    """Three HIP ranks on the one card: the middle rank has a neighbour on both sides (both message merges, both frames)."""
    results = run_ranks("hip", 3, tmp_path, steps=6)
    sc, pos_ref, vel_ref = single_domain_reference(steps=6)
    check_union(results, sc, pos_ref, vel_ref)
    assert all(int(r["transfers"]) == 6 + 1 for r in results)


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", ["1", "0"])
def test_three_rank_long_scene_overlapped_and_plain_step_gpu(tmp_path, overlap):
    """Slabs of 14 layers: the middle rank integrates two cut zones of 5 layers first, packs, then its 4 interior layers
    (sph_slab_step_begin); with SPHMI_SLAB_OVERLAP=0 the plain step-then-pack path. Both bit-identical to the single domain."""
    env = {"SPHMI_TEST_LONG_SCENE": "1", "SPHMI_SLAB_OVERLAP": overlap}
    results = run_ranks("hip", 3, tmp_path, steps=5, env=env)
    sc, pos_ref, vel_ref = single_domain_reference(steps=5, long_scene=True)
    check_union(results, sc, pos_ref, vel_ref)
    cuts = results[1]["cuts"]
    assert cuts[2] - cuts[1] >= 11, "the middle slab must be thicker than its two cut zones"


@pytest.mark.gpu
def test_other_iteration_counts_change_the_ghost_depths_gpu(tmp_path):
    """maxIteration = 2: every stage's ghost depth changes (4 hops instead of 6); still bit-identical to the single domain.
    maxIteration = 4 needs 8 hops = 4.13 layers > the 4 ghost layers: sph_slab_init must refuse it."""
    results = run_ranks("hip", 2, tmp_path, steps=4, env={"SPHMI_TEST_MAXITER": "2"})
    sc, pos_ref, vel_ref = single_domain_reference(steps=4, max_iteration=2)
    assert sc["cfg"].maxIteration == 2
    check_union(results, sc, pos_ref, vel_ref)
    sc4 = scenes.liquid_box((8.0, 8.0, 20.0), (12, 10, 30), mask=0xffffffff)
    sc4["cfg"].maxIteration = 4
    n = sc4["cfg"].particleCount
    lay = S.particle_layers(sc4["position"], sc4["cfg"])
    with pytest.raises(sphmi.SphError):
        S.HipSlabBackend(sc4["cfg"], sc4["position"], sc4["velocity"], np.arange(n, dtype=np.uint32),
                         S.make_slab([int(lay.min()), int(lay.max()) + 1], 0, 1, n))


@pytest.mark.gpu
def test_two_slabs_exchange_frames_over_rccl_in_one_process(tmp_path):
    """The RCCL path on the 1-GPU test box (tests/rccl_pair_worker.py): two slabs, real `batch_isend_irecv` on a world-1 NCCL
    group (self sends), HipSlabBackend frames sent in place. Union of the owned particles == single-domain oracle, bit for bit."""
    steps = 6
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_pair_worker.py"), "--steps", str(steps), "--port", str(free_port()),
                        "--out", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    results = [np.load(os.path.join(tmp_path, "rank%d.npz" % k)) for k in range(2)]
    sc, pos_ref, vel_ref = single_domain_reference(steps=steps)
    check_union(results, sc, pos_ref, vel_ref)
    assert all(int(x["transfers"]) == steps + 1 for x in results)   # one framed transfer per step after the first exchange
    assert all(bool(x["asynchronous"]) and int(x["record_words"]) == sphmi.SLAB_COMPACT_WORDS for x in results)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"SPHMI_TEST_BOUND_WORDS": "36"}, {"SPHMI_SLAB_FULL_RECORDS": "1"},
                                 {"SPHMI_SLAB_SYNC_EXCHANGE": "1"}, dict(DRIFT, SPHMI_TEST_LONG_SCENE="1")])
def test_rccl_exchange_variants_in_one_process(tmp_path, env):
    """The asynchronous RCCL exchange (counts read on the device, one host wait per step) with frames too short for their
    messages in EVERY step (the rest is fetched by finish() and the rebuild repeated from the complete payloads), with full 9-word
    records, against the synchronous exchange, and with liquid drifting across the cut."""
    steps = 5
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_pair_worker.py"), "--steps", str(steps), "--port", str(free_port()),
                        "--out", str(tmp_path)], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    results = [np.load(os.path.join(tmp_path, "rank%d.npz" % k)) for k in range(2)]
    ref_env = {k: v for k, v in env.items() if k.startswith("SPHMI_TEST_") and k != "SPHMI_TEST_BOUND_WORDS"}
    sc, pos_ref, vel_ref = single_domain_reference(steps=steps, long_scene="SPHMI_TEST_LONG_SCENE" in env, env=ref_env)
    check_union(results, sc, pos_ref, vel_ref)
    if "SPHMI_TEST_BOUND_WORDS" in env:
        assert all(int(x["transfers"]) == 2 * steps for x in results)
    assert all(bool(x["asynchronous"]) == ("SPHMI_SLAB_SYNC_EXCHANGE" not in env) for x in results)


@pytest.mark.gpu
def test_single_rank_slab_backend_equals_plain_solver():
    """world = 1: pack + rebuild every step (sort by global id) must not change anything."""
    sc = scenes.liquid_box((8.0, 8.0, 20.0), (12, 10, 30), mask=0xffffffff, jitter_in_r0=0.05)
    cfg = sc["cfg"]
    n = cfg.particleCount
    plain = scenes.hip_for(sc)
    lay = S.particle_layers(sc["position"], cfg)
    slab = S.make_slab([int(lay.min()), int(lay.max()) + 1], 0, 1, n)
    cfg2 = scenes.liquid_box((8.0, 8.0, 20.0), (12, 10, 30), mask=0xffffffff, jitter_in_r0=0.05)["cfg"]
    be = S.HipSlabBackend(cfg2, sc["position"], sc["velocity"], np.arange(n, dtype=np.uint32), slab)
    dd = S.SlabDecomposition(be, 0, 1)
    for it in range(3):
        plain.step(it)
        assert dd.step(it) == n
    gid, pos, vel = be.owned_state()
    assert np.array_equal(gid, np.arange(n))
    assert scenes.bits_equal(pos, plain.read_position_buffer())
    assert scenes.bits_equal(vel, plain.read_velocity_buffer())


@pytest.mark.gpu
def test_initial_ids_in_any_order_and_unsorted_message_is_rejected():
    """sph_slab_init sorts an arbitrary initial order by global id; the per-step rebuild is a three-way merge and needs
    messages in ascending global-id order (as sph_slab_pack writes them): an unsorted one must fail loudly."""
    import torch
    import sphmi
    sc = scenes.liquid_box((8.0, 8.0, 20.0), (12, 10, 30), mask=0xffffffff, jitter_in_r0=0.05)
    cfg = sc["cfg"]
    n = cfg.particleCount
    lay = S.particle_layers(sc["position"], cfg)
    lo, hi = int(lay.min()), int(lay.max()) + 1
    mid = (lo + hi) // 2
    slab = S.make_slab([lo, mid, hi], 0, 2, n)  # rank 0 of 2: has an upper neighbour
    idx = S.local_indices(lay, slab)
    perm = np.random.default_rng(3).permutation(idx.size)
    be = S.HipSlabBackend(cfg, sc["position"][idx][perm], sc["velocity"][idx][perm], idx[perm], slab)
    pos, vel, gid, owned = be.solver.slab_read()
    assert np.array_equal(gid, idx) and scenes.bits_equal(pos, sc["position"][idx])
    be.step(0)
    kept, down, up = be.pack()
    assert down.numel() == 0 and up.numel() > 0
    rec = up.cpu().numpy().view(np.uint32).reshape(-1, S.SLAB_RECORD_WORDS if hasattr(S, "SLAB_RECORD_WORDS") else 9)
    assert np.all(np.diff(rec[:, 8].astype(np.int64)) > 0)            # messages leave sorted by global id
    assert not np.any(rec[:, 3].view(np.float32).astype(np.int32) == sphmi.BOUNDARY_PARTICLE)  # the static shell is never sent
    # a well-formed message from "above": what this rank would receive = its own ghosts minus the boundary ones, which it keeps
    is_bnd = pos[:, 3].astype(np.int32) == sphmi.BOUNDARY_PARTICLE
    ghosts = np.flatnonzero((owned == 0) & ~is_bnd)
    assert kept == int(((owned != 0) | is_bnd).sum())                    # kept = owned + the boundary particles of the ghost layers
    msg = np.zeros((ghosts.size, 9), np.uint32)
    msg[:, 0:4] = pos[ghosts].view(np.uint32); msg[:, 4:8] = vel[ghosts].view(np.uint32); msg[:, 8] = gid[ghosts]
    good = torch.from_numpy(msg.view(np.int32).reshape(-1)).to(be.device)
    assert be.rebuild(None, good) == kept + ghosts.size == idx.size
    _, _, gid2, _ = be.solver.slab_read()
    assert np.all(np.diff(gid2.astype(np.int64)) > 0)                  # merged set is sorted
    be.step(1)
    kept, down, up = be.pack()
    bad = torch.from_numpy(msg[::-1].copy().view(np.int32).reshape(-1)).to(be.device)
    be.rebuild(None, bad)
    with pytest.raises(sphmi.SphError):
        be.pack()


@pytest.mark.gpu
def test_bench_self_launch_two_ranks_on_one_card():
    """`python bench.py --gpus 2` from a plain shell (no WORLD_SIZE): bench.py starts its two ranks itself, before any GPU call.
    The test box has one card, so the ranks share it and the transport falls back to gloo (RCCL refuses two ranks on one
    device); everything else — per-rank scene slices, slab solvers, overlapped step, exchange, the relayed result line — is the
    multi-GPU path."""
    import json
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SPHMI_DIST_BACKEND"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(scenes.ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny_long", "--steps", "4",
                        "--warmup", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["halo"]["owned_sets_partition_all_particles"] is True
    assert out["dist_backend"] in ("gloo", "nccl") and len(out["devices"]) == 2
    assert "DIAG" not in out["lib"]["build"] and len(out["lib"]["sha256"]) == 64
    assert out["roofline"]["frac"] > 0
