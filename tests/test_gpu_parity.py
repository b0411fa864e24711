"""GPU parity tests (run with -m gpu on an MI355X). Everything goes through the C ABI of libsphmi.so
(include/sphmi.h) via the owOpenCLSolver-shaped binding, and is compared with the oracle on the same inputs.

Bars (BASELINE.json north_star): bit-exact for cell assignments, sort permutation, cell table and neighbour ids;
particle positions within 1e-5 relative after N steps. In practice every buffer is expected bit-identical, because
the kernels keep the reference's operand types and evaluation order (no FMA contraction, IEEE div/sqrt, denormals
on); the tests therefore assert bit equality and report the 1e-5 criterion as a second, weaker check."""
import json
import os

import numpy as np
import pytest

import scenes
import sphmi

pytestmark = pytest.mark.gpu

POSITION_RTOL = 1e-5  # north_star: "particle positions within 1e-5 relative after N steps"
# K4's intermediate table (first index per cell, -1 for empty cells) is only materialised by sph_run_indexx();
# the fused sph_step() writes the fixed-up table directly (sph_sort.hip, k_sort_post), so it is not compared there.
FUSED_SKIP = ("gridCellIndex",)


def canon_hip(h, N):
    return scenes.canonical(h.buffer, N)


def canon_ora(o, N):
    return scenes.canonical(o.buffer, N)


def assert_same(got, want, where, skip=()):
    bad = []
    for k in want:
        if k in skip:
            continue
        if not scenes.bits_equal(got[k], want[k]):
            bad.append("%s: %s" % (k, scenes.diff_report(got[k], want[k])))
    assert not bad, "%s\n  " % where + "\n  ".join(bad)


def position_rel_err(got, want, r0):
    """max_i |x - x_ref| / max(|x_ref|, r0) per component."""
    g, w = got[:, :3].astype(np.float64), want[:, :3].astype(np.float64)
    return float((np.abs(g - w) / np.maximum(np.abs(w), r0)).max())


@pytest.mark.parametrize("name", list(scenes.SCENES))
def test_stage_by_stage_matches_oracle(name):
    """Every sph_run_* leaves the buffers the reference's _run* would leave (exported in the reference's layout)."""
    sc = scenes.SCENES[name]()
    N = sc["cfg"].particleCount
    big = N > 20000
    hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc, threads=8)
    for it in range(2 if big else 4):
        staged = it in (0, 3) and not (big and it > 0)
        if staged:
            for k, st in enumerate(scenes.STAGE_SEQUENCE):
                m = getattr(hip, scenes.HIP_STAGE_METHOD[st])
                m(it) if st == "integrate" else m()
                ora.run(st)
                if big and st not in ("sort", "indexPostPass", "findNeighbors", "computeDensity", "integrate"):
                    continue
                # The export rebuilds sortedPosition.w (the cell id) from the key array, which hashParticles and sort
                # rewrite before sortPostPass refreshes sortedPosition: skip that one buffer for those two stages.
                skip = ("sortedPosition",) if st in ("hashParticles", "sort") else ()
                if k < scenes.STAGE_SEQUENCE.index("indexx"):
                    skip += FUSED_SKIP  # still the table of the last *staged* step until indexx runs again
                assert_same(canon_hip(hip, N), canon_ora(ora, N), "%s step %d stage %d %s" % (name, it, k, st), skip)
        else:
            hip.step(it)
            ora.step()
        if sc["elastic"] is not None:
            sig = sphmi.muscle_signal(it)
            hip.updateMuscleActivityData(sig)
            ora.update_muscles(sig)
        assert_same(canon_hip(hip, N), canon_ora(ora, N), "%s after step %d" % (name, it), () if staged else FUSED_SKIP)


@pytest.mark.parametrize("name", ["tiny", "tiny_compressed", "tiny_jitter", "tiny_elastic"])
def test_fused_step_matches_reference_fixture(name):
    """sph_step() x 11 against the arrays the reference's own kernels produced (tests/golden/<name>.npz)."""
    z = np.load(os.path.join(scenes.GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    sc = scenes.SCENES[name]()
    N = sc["cfg"].particleCount
    hip = scenes.hip_for(sc)
    for it in range(meta["steps"]):
        hip.step(it)
        if sc["elastic"] is not None:
            hip.updateMuscleActivityData(sphmi.muscle_signal(it))
        if it in (0, 10):
            c = canon_hip(hip, N)
            for key in ("position", "velocity", "rho", "neighborIds", "pressure"):
                assert scenes.bits_equal(c[key], z["%s_%d" % (key, it)]), \
                    "%s %s after step %d: %s" % (name, key, it, scenes.diff_report(c[key], z["%s_%d" % (key, it)]))
            got = {b: scenes.sha(v) for b, v in c.items()}
            bad = [b for b in got if got[b] != meta["step_hashes"][str(it)][b] and b not in FUSED_SKIP]
            assert not bad, (name, it, bad)
    err = position_rel_err(hip.read_position_buffer(), z["position_10"], sc["cfg"].r0)
    assert err <= POSITION_RTOL


def test_config1_100_steps_against_reference_fixture():
    """BASELINE config #1: configuration/positionPureLiquid.txt, 100 steps (pressure becomes active after ~step 70)."""
    z = np.load(os.path.join(scenes.GOLDEN, "config1.npz"))
    meta = json.loads(str(z["meta"]))
    sc = scenes.config1()
    N = sc["cfg"].particleCount
    sim = sphmi.owPhysicsFluidSimulator(sc["cfg"], sc["position"], sc["velocity"])
    for it in range(100):
        sim.simulationStep(read_back=(it in (0, 9, 99)))
        if str(it) in meta["step_hashes"]:
            pos = sim.getPosition_cpp()
            want = z["position_sample_%d" % it]
            err = position_rel_err(pos[z["sample_ids"]], want, sc["cfg"].r0)
            assert err <= POSITION_RTOL, "step %d: rel err %g" % (it, err)
            got = {b: scenes.sha(v) for b, v in canon_hip(sim.ocl_solver, N).items()}
            bad = [b for b in got if got[b] != meta["step_hashes"][str(it)][b] and b not in FUSED_SKIP]
            assert not bad, "config1 step %d: %s differ from the reference" % (it, bad)


def test_worm_scene_against_reference_fixture():
    """BASELINE config #3: generated C. elegans scene (springs, muscles, membranes, 16-bit aliasing), 10 steps."""
    z = np.load(os.path.join(scenes.GOLDEN, "worm.npz"))
    meta = json.loads(str(z["meta"]))
    sc = scenes.worm_scene()
    N = sc["cfg"].particleCount
    sim = sphmi.owPhysicsFluidSimulator(sc["cfg"], sc["position"], sc["velocity"], sc["elastic"], sc["membranes"],
                                        sc["particle_membranes"], muscles=True)
    for it in range(10):
        sim.simulationStep(read_back=True)
        if str(it) in meta["step_hashes"]:
            err = position_rel_err(sim.getPosition_cpp()[z["sample_ids"]], z["position_sample_%d" % it], sc["cfg"].r0)
            assert err <= POSITION_RTOL, "step %d: rel err %g" % (it, err)
            got = {b: scenes.sha(v) for b, v in canon_hip(sim.ocl_solver, N).items()}
            bad = [b for b in got if got[b] != meta["step_hashes"][str(it)][b] and b not in FUSED_SKIP]
            assert not bad, "worm step %d: %s differ from the reference" % (it, bad)


def test_staged_and_fused_paths_agree():
    sc = scenes.SCENES["tiny_elastic"]()
    a = sphmi.owPhysicsFluidSimulator(sc["cfg"], sc["position"], sc["velocity"], sc["elastic"], sc["membranes"],
                                      sc["particle_membranes"], fused=True, muscles=True)
    b = sphmi.owPhysicsFluidSimulator(sc["cfg"], sc["position"], sc["velocity"], sc["elastic"], sc["membranes"],
                                      sc["particle_membranes"], fused=False, muscles=True)
    for it in range(6):
        a.simulationStep(); b.simulationStep()
    assert scenes.bits_equal(a.getPosition_cpp(), b.getPosition_cpp())
    assert scenes.bits_equal(a.ocl_solver.read_velocity_buffer(), b.ocl_solver.read_velocity_buffer())
    assert scenes.bits_equal(a.getDensity_cpp(), b.getDensity_cpp())
    assert scenes.bits_equal(a.getParticleIndex_cpp(), b.getParticleIndex_cpp())


def test_viewer_frame_feed_matches_oracle():
    """density per ORIGINAL particle through the viewer's inverse-permutation contract
    (getDensity_cpp is in sorted order, owWorldSimulation.cpp:112-127) equals the oracle's."""
    from sphmi import frames
    sc = scenes.SCENES["tiny_jitter"]()
    sim = sphmi.owPhysicsFluidSimulator(sc["cfg"], sc["position"], sc["velocity"])
    ora = scenes.oracle_for(sc)
    for _ in range(3):
        sim.simulationStep(); ora.step()
    pos, rho = frames.frame(sim)
    N = sc["cfg"].particleCount
    opi = ora.buffer("particleIndex").reshape(-1, 2)[:N]
    want = ora.buffer("rho")[:N][frames.invert_particle_index(opi)]
    assert scenes.bits_equal(rho, want)
    assert scenes.bits_equal(pos, ora.buffer("position").reshape(-1, 4)[:N])
    ora.close()


def _check_search_structures(hip, cfg, N):
    """Size-independent properties of the binning outputs."""
    pi = hip.read_particleIndex_buffer()
    keys, vals = pi[:, 0].astype(np.int64), pi[:, 1].astype(np.int64)
    assert np.all(np.diff(keys) >= 0), "keys not sorted"
    same = keys[1:] == keys[:-1]
    assert np.all(vals[1:][same] > vals[:-1][same]), "ties not in ascending orig id order (SURVEY App. B #4)"
    assert np.array_equal(np.sort(vals), np.arange(N)), "vals is not a permutation"
    back = hip.buffer("particleIndexBack").astype(np.int64)
    assert np.array_equal(back[vals], np.arange(N)), "particleIndexBack is not the inverse permutation"
    G = cfg.gridCellCount
    table = hip.buffer("gridCellIndexFixedUp").astype(np.int64)
    want = np.searchsorted(keys, np.arange(G + 1), side="left")  # #particles with cell < c
    assert np.array_equal(table, want), "cell table != lower_bound(keys)"
    return keys, vals


def test_config2_one_million_cube_against_oracle():
    """BASELINE config #2 (1M-particle pure-liquid cube, N = 1,058,808, reference-exact 16-bit mode) at full size:
    search-structure properties, then two full steps compared with the oracle word for word."""
    sc = scenes.liquid_box((50.0, 50.0, 50.0), (100, 100, 100))
    cfg = sc["cfg"]
    N = cfg.particleCount
    assert N == 1058808
    hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc, threads=16)
    for it in range(2):
        hip.step(it)
        ora.step()
        _check_search_structures(hip, cfg, N)
        got = hip.buffer("neighborIds")
        assert np.array_equal(got, ora.buffer("neighborIds")), "neighbour ids differ at step %d" % it
        assert scenes.bits_equal(hip.buffer("neighborMap"), ora.buffer("neighborMap"))
        assert scenes.bits_equal(hip.buffer("rho"), ora.buffer("rho"))
        assert scenes.bits_equal(hip.read_position_buffer(), ora.buffer("position").reshape(-1, 4)[:N])
        assert scenes.bits_equal(hip.read_velocity_buffer(), ora.buffer("velocity").reshape(-1, 4)[:N])
    # density read-back contract: sorted order (owOpenCLSolver.h:61)
    assert scenes.bits_equal(hip.read_density_buffer(), ora.buffer("rho")[:N])
    # boundary particles are never written
    bnd = sc["position"][:, 3].astype(int) == 3
    assert scenes.bits_equal(hip.read_position_buffer()[bnd], sc["position"][bnd])


def test_wide_mode_large_grid_properties():
    """Wide ids on a grid with > 2^16 cells (3 radix passes): properties + oracle equality for one step."""
    sc = scenes.liquid_box((60.0, 40.0, 60.0), (60, 40, 60), mask=0xffffffff, jitter_in_r0=0.02)
    cfg = sc["cfg"]
    N = cfg.particleCount
    assert cfg.gridCellCount > 65536
    hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc, threads=16)
    hip.step(0)
    ora.step()
    keys, _ = _check_search_structures(hip, cfg, N)
    assert keys.max() > 65535
    assert np.array_equal(hip.buffer("neighborIds"), ora.buffer("neighborIds"))
    assert scenes.bits_equal(hip.read_position_buffer(), ora.buffer("position").reshape(-1, 4)[:N])


def test_config4_sixteen_million_box_properties():
    """BASELINE config #4 at full size (16,507,704 particles, wide cell ids, 1.9 M cells): too large for the oracle in a test,
    so size-independent properties of one step: sorted keys with ascending-id ties, permutation + inverse, cell table =
    lower_bound, every neighbour distance recomputable bit for bit from the sorted positions and inside the search radius,
    no self / duplicate neighbours, and the density recomputed from the stored distances in slot order (f64, like
    pcisph_computeDensity) equal bit for bit."""
    sc = scenes.liquid_box((78.0, 50.0, 470.0), (160, 100, 1000), mask=0xffffffff)
    cfg = sc["cfg"]
    N = cfg.particleCount
    assert N == 16507704
    hip = scenes.hip_for(sc)
    hip._runClearBuffers(); hip._runHashParticles(); hip._runSort(); hip._runSortPostPass(); hip._runIndexx()
    hip._runIndexPostPass(); hip._runFindNeighbors(); hip._run_pcisph_computeDensity()
    keys, vals = _check_search_structures(hip, cfg, N)
    assert keys.max() > 65535
    nm = hip.buffer("neighborMap").reshape(N, 32, 2)
    ids = hip.buffer("neighborIds").reshape(N, 32)
    dist = nm[:, :, 1]
    assert np.array_equal(nm[:, :, 0].astype(np.int32), ids)            # float-coded ids of the reference layout
    valid = ids >= 0
    assert np.array_equal(valid, dist != -1.0)
    assert np.all(np.cumsum(~valid, axis=1)[valid] == 0), "empty slots must come last"
    assert valid.sum() > 30 * (N - 600000), "the liquid bulk must have (nearly) full lists"
    spos = hip.buffer("sortedPosition").reshape(-1, 4)[:N]
    simScale, h = np.float32(cfg.simulationScale), np.float32(cfg.h)
    rmax2 = (np.float32(31) * h / np.float32(30)) ** 2
    rho = np.zeros(N, np.float64)
    hs2 = (h * simScale) * (h * simScale)
    own = np.arange(N, dtype=np.int64)
    for k in range(32):                                                  # slot by slot: 16.5 M-element vector operations
        v = valid[:, k]
        j = ids[:, k].astype(np.int64)
        assert not np.any(j[v] == own[v]), "a particle lists itself"
        if k:
            assert not np.any(v & (ids[:, k] == ids[:, k - 1])), "duplicate neighbour"
        e = spos[own[v], :3] - spos[j[v], :3]
        d2 = (e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2]
        assert np.all(d2 <= rmax2)
        assert scenes.bits_equal(np.sqrt(d2) * simScale, dist[v, k]), "slot %d: stored distance != sqrt(d2)*simulationScale" % k
        r2 = dist[:, k] * dist[:, k]
        a = hs2 - r2
        rho += np.where(v, (a * a * a).astype(np.float64), 0.0)
    hs6 = np.float64((hs2 * hs2) * hs2)
    rho = np.maximum(rho, hs6) * (np.float64(np.float32(cfg.mass)) * np.float64(cfg.Wpoly6Coefficient))
    assert scenes.bits_equal(rho.astype(np.float32), hip.read_density_buffer())


def test_config4_sixteen_million_box_full_steps_against_oracle():
    """BASELINE config #4 at full size, FULL steps, against the oracle itself (16 host threads, ~2 s per step): positions, velocities,
    densities and pressures of all 16,507,704 particles after each of three fused steps, bit for bit. (The other buffers at this size are
    covered by the properties test above and, word for word, by the 1 M wide-mode test below.)"""
    sc = scenes.liquid_box((78.0, 50.0, 470.0), (160, 100, 1000), mask=0xffffffff)
    N = sc["cfg"].particleCount
    hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc, threads=16)
    for it in range(3):
        hip.step(it)
        ora.step()
        assert scenes.bits_equal(hip.read_position_buffer(), ora.buffer("position").reshape(-1, 4)[:N]), it
        assert scenes.bits_equal(hip.read_velocity_buffer(), ora.buffer("velocity").reshape(-1, 4)[:N]), it
        assert scenes.bits_equal(hip.read_density_buffer(), ora.buffer("rho").reshape(-1)[:N]), it
        assert scenes.bits_equal(hip.buffer("pressure").reshape(-1)[:N], ora.buffer("pressure").reshape(-1)[:N]), it
    ora.close()


def test_wide_mode_million_particles_full_steps_against_oracle():
    """Wide cell ids at scale, FULL steps: 1.07 M particles on a grid of 152,561 declared cells (3 radix passes, ids > 2^16),
    three fused steps compared with the oracle word for word in every buffer (search structures, neighbour ids and distances,
    densities, pressures, accelerations, predicted / sorted / final positions, velocities)."""
    sc = scenes.liquid_box((60.0, 40.0, 60.0), (110, 80, 110), mask=0xffffffff, jitter_in_r0=0.02)
    cfg = sc["cfg"]
    N = cfg.particleCount
    assert N > 1000000 and cfg.gridCellCount > 65536
    hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc, threads=16)
    for it in range(3):
        hip.step(it)
        ora.step()
        keys, _ = _check_search_structures(hip, cfg, N)
        assert keys.max() > 65535
        assert_same(canon_hip(hip, N), canon_ora(ora, N), "wide 1M, step %d" % it, FUSED_SKIP)
    ora.close()


def test_config5_sixty_four_million_dam_break_properties():
    """BASELINE config #5 at full size on one GPU (65,469,208 particles: a 250 x 400 x 640 liquid column at low x in a 240h x 200h x
    310h box, wide cell ids, 15 M declared cells, 31 GB of device state). The oracle cannot run it in a test, so: the search
    structures in full (sorted keys with ascending-id ties, permutation + inverse, cell table = lower_bound); in three windows
    of 2^18 sorted particles (start, liquid bulk, end) every stored neighbour distance recomputed bit for bit from the sorted
    positions, inside the search radius, no self / duplicate neighbours, empty slots last, and the density recomputed from the
    stored distances in slot order equal bit for bit; then two full fused steps: everything finite, boundary particles
    untouched, the column falling (mean stored vy of the liquid = g dt to 1e-3)."""
    sc = scenes.liquid_box((240.0, 200.0, 310.0), (250, 400, 640), mask=0xffffffff)
    cfg = sc["cfg"]
    N = cfg.particleCount
    assert N == 65469208
    nl = sc["numOfLiquidP"]
    hip = scenes.hip_for(sc)
    hip._runClearBuffers(); hip._runHashParticles(); hip._runSort(); hip._runSortPostPass(); hip._runIndexx()
    hip._runIndexPostPass(); hip._runFindNeighbors(); hip._run_pcisph_computeDensity()
    keys, vals = _check_search_structures(hip, cfg, N)
    assert keys.max() > 65535
    is_liquid = vals < nl  # (liquid_box puts the liquid lattice first; the exported sortedPosition.w is the cell id)
    del vals
    spos = hip.buffer("sortedPosition").reshape(-1, 4)[:N, :3].copy()
    rho_gpu = hip.read_density_buffer()
    simScale, h = np.float32(cfg.simulationScale), np.float32(cfg.h)
    rmax2 = (np.float32(31) * h / np.float32(30)) ** 2
    hs2 = (h * simScale) * (h * simScale)
    hs6 = np.float64((hs2 * hs2) * hs2)
    W = 1 << 18
    liquid_sorted = np.flatnonzero(is_liquid)
    mid = int(liquid_sorted[liquid_sorted.size // 2]) // 64 * 64
    for first in (0, mid, N - W):
        ids, dist = hip.neighbor_rows(first, W)
        valid = ids >= 0
        assert np.array_equal(valid, dist != -1.0)
        assert np.all(np.cumsum(~valid, axis=1)[valid] == 0), "empty slots must come last"
        own = np.arange(first, first + W, dtype=np.int64)
        rho = np.zeros(W, np.float64)
        for k in range(32):
            v = valid[:, k]
            j = ids[:, k].astype(np.int64)
            assert not np.any(j[v] == own[v]), "a particle lists itself"
            if k:
                assert not np.any(v & (ids[:, k] == ids[:, k - 1])), "duplicate neighbour"
            e = spos[own[v]] - spos[j[v]]
            d2 = (e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2]
            assert np.all(d2 <= rmax2)
            assert scenes.bits_equal(np.sqrt(d2) * simScale, dist[v, k]), "slot %d: stored distance != sqrt(d2)*simulationScale" % k
            a = hs2 - dist[:, k] * dist[:, k]
            rho += np.where(v, (a * a * a).astype(np.float64), 0.0)
        rho = np.maximum(rho, hs6) * (np.float64(np.float32(cfg.mass)) * np.float64(cfg.Wpoly6Coefficient))
        assert scenes.bits_equal(rho.astype(np.float32), rho_gpu[first:first + W])
        if first == mid:
            liq = is_liquid[first:first + W]
            assert liq.sum() > W // 4 and valid[liq].sum() > 24 * liq.sum(), "the liquid bulk must have (nearly) full lists"
    del spos, keys
    for it in range(2):
        hip.step(it)
    pos, vel = hip.read_position_buffer(), hip.read_velocity_buffer()
    assert np.all(np.isfinite(pos)) and np.all(np.isfinite(vel))
    assert scenes.bits_equal(pos[nl:], sc["position"][nl:]), "boundary particles must not move"
    g_dt = np.float64(cfg.gravity_y) * np.float64(cfg.timeStep)
    # the stored velocity is the average of old and new: from rest 0.5 g dt after one step, (0.5 + 1.5) / 2 = 1.0 g dt
    # after two; pressure and viscosity forces are pairwise and cancel in the mean
    assert abs(vel[:nl, 1].astype(np.float64).mean() / g_dt - 1.0) < 1e-3


def test_config5_sixty_four_million_full_steps_against_oracle():
    """BASELINE config #5 at full size against the oracle itself (65,469,208 particles; ~40 GB of host memory and ~10 s per step for
    the oracle on 16 threads): positions, velocities, densities and pressures after each of two fused steps, bit for bit."""
    sc = scenes.liquid_box((240.0, 200.0, 310.0), (250, 400, 640), mask=0xffffffff)
    N = sc["cfg"].particleCount
    assert N == 65469208
    hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc, threads=16)
    for it in range(2):
        hip.step(it)
        ora.step()
        assert scenes.bits_equal(hip.read_position_buffer(), ora.buffer("position").reshape(-1, 4)[:N]), it
        assert scenes.bits_equal(hip.read_velocity_buffer(), ora.buffer("velocity").reshape(-1, 4)[:N]), it
        assert scenes.bits_equal(hip.read_density_buffer(), ora.buffer("rho").reshape(-1)[:N]), it
        assert scenes.bits_equal(hip.buffer("pressure").reshape(-1)[:N], ora.buffer("pressure").reshape(-1)[:N]), it
    ora.close()


def test_two_solvers_in_one_process():
    """Two solvers side by side in one process (own streams; on a second device too where the box has one): the dynamic-LDS opt-in
    of the search kernel is per device, and neither solver disturbs the other's state."""
    import torch
    sc = scenes.SCENES["tiny_jitter"]()
    N = sc["cfg"].particleCount
    ora = scenes.oracle_for(sc)
    for _ in range(2):
        ora.step()
    want = ora.buffer("position").reshape(-1, 4)[:N]
    devices = [0, 1] if torch.cuda.device_count() > 1 else [0, 0]
    solvers = []
    for dev in devices:
        s2 = scenes.SCENES["tiny_jitter"]()
        s2["cfg"].device = dev
        solvers.append(scenes.hip_for(s2))
    for it in range(2):          # interleaved: a, b, a, b
        for hs in solvers:
            hs.step(it)
    for hs in solvers:
        assert scenes.bits_equal(hs.read_position_buffer(), want)
        assert np.array_equal(hs.buffer("neighborIds"), ora.buffer("neighborIds"))
    ora.close()


def test_ragged_sizes_and_single_particle_cells():
    """N not a multiple of 64/256/4096, and a sparse scene where most cells hold one particle."""
    for lattice, spacing in (((7, 5, 3), 0.93), ((3, 3, 3), 2.9), ((1, 1, 1), 0.93)):
        sc = scenes.liquid_box((6.0, 5.0, 7.0), lattice, spacing_in_r0=spacing)
        N = sc["cfg"].particleCount
        hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc)
        for it in range(3):
            hip.step(it)
            ora.step()
        assert_same(canon_hip(hip, N), canon_ora(ora, N), "lattice %s" % (lattice,), FUSED_SKIP)


def test_overcrowded_cells_take_the_fallback_paths():
    """Far more than 32 candidates within h and ~700 particles per cell: every compaction list overflows (fallback kernel,
    one wave per particle), candidate runs exceed the LDS budget (small batches), radial bins overflow 32 at the first bin."""
    sc = scenes.liquid_box((8.0, 8.0, 8.0), (18, 18, 18), spacing_in_r0=0.45, origin_in_r0=(4.0, 4.0, 4.0))
    N = sc["cfg"].particleCount
    hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc, threads=8)
    hip.reset_stage_times()
    for it in range(2):
        hip.step(it)
        ora.step()
        assert_same(canon_hip(hip, N), canon_ora(ora, N), "overcrowded step %d" % it, FUSED_SKIP)
    c = hip.buffer("debugCounters")
    assert c[1] > 1000, "the list-overflow fallback was expected to be exercised (got %d)" % c[1]
    nm = hip.buffer("neighborIds").reshape(-1, 32)
    assert (nm[:, 31] >= 0).sum() > 1000  # many full lists


def test_evolved_dam_break_state_parity():
    """Parity on DISORDERED states: a liquid column (110 k particles + 35 k boundary) collapses for 1,500 steps on the GPU — free fall,
    impact on the floor, a front running along it; at three checkpoints the GPU state is handed to the oracle and one more
    step of both is compared word for word (neighbour ids and distances, densities, pressures, accelerations, positions,
    velocities). Lattice scenes never produce the cell occupancies and list lengths this does."""
    sc = scenes.liquid_box((60.0, 30.0, 30.0), (40, 55, 50), origin_in_r0=(3.0, 3.0, 3.0))
    cfg = sc["cfg"]
    N, L = cfg.particleCount, 40 * 55 * 50
    hip = scenes.hip_for(sc)
    y0, x0 = sc["position"][:L, 1].mean(), sc["position"][:L, 0].max()
    done = 0
    for checkpoint in (500, 1000, 1500):
        for it in range(done, checkpoint):
            hip.step(it)
        done = checkpoint
        pos, vel = hip.read_position_buffer(), hip.read_velocity_buffer()
        assert np.all(np.isfinite(pos)) and np.all(np.isfinite(vel))
        state = dict(sc, position=pos.copy(), velocity=vel.copy())
        ora = scenes.oracle_for(state, threads=16)
        hip.reset_stage_times()
        hip.step(checkpoint)
        ora.step()
        done += 1
        assert_same(canon_hip(hip, N), canon_ora(ora, N), "dam break, step %d" % checkpoint, FUSED_SKIP)
        ora.close()
    pos = hip.read_position_buffer()
    assert y0 - pos[:L, 1].mean() > 20.0, "the column must have collapsed (oracle alone: 47.6 -> 20.9)"
    assert pos[:L, 0].max() - x0 > 60.0, "the front must have run along the floor (oracle alone: 65 -> 140)"


def test_coincident_particles_neighbour_search():
    """Particles at identical positions (d = 0): found as each other's first-bin neighbours with distance 0; compared up to
    the density pass (beyond it the reference divides by r = 0, sphFluid.cl:1172-1178)."""
    sc = scenes.liquid_box((8.0, 8.0, 8.0), (12, 10, 12), jitter_in_r0=0.03)
    pos = sc["position"].copy()
    pos[5:10, :3] = pos[200:205, :3]  # five coincident pairs
    sc["position"] = pos
    N = sc["cfg"].particleCount
    hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc)
    for st in scenes.STAGE_SEQUENCE[:scenes.STAGE_SEQUENCE.index("computeDensity") + 1]:
        getattr(hip, scenes.HIP_STAGE_METHOD[st])()
        ora.run(st)
    got, want = canon_hip(hip, N), canon_ora(ora, N)
    for k in ("particleIndex", "gridCellIndexFixedUp", "neighborIds", "neighborDist", "rho"):
        assert scenes.bits_equal(got[k], want[k]), (k, scenes.diff_report(got[k], want[k]))
    d = want["neighborDist"].reshape(-1, 32)
    assert (d == 0.0).sum() >= 10


@pytest.mark.parametrize("seed", range(24))
def test_random_heterogeneous_scenes_against_oracle(seed):
    """Seeded random scenes the hand-made ones do not cover together: non-cubic boxes, sparse to dense lattices, strong jitter,
    both cell-id modes, blobs of hundreds of particles inside one cell next to empty space, and particles sitting exactly on cell
    faces (coordinate = k x cell size, where the reference's truncation decides the cell). Search, cell table, neighbour lists and
    densities are compared bit for bit; scenes without blobs also run three full steps."""
    rng = np.random.default_rng(1000 + seed)
    box = tuple(float(rng.integers(6, 15)) for _ in range(3))
    spacing = float(rng.choice([0.6, 0.8, 0.93, 1.4]))
    lattice = tuple(int(max(2, min(40, (2.0 * b - 7.0) / spacing * rng.uniform(0.6, 1.0)))) for b in box)  # (the box is 2b r0 long)
    wide = bool(seed % 2)
    sc = scenes.liquid_box(box, lattice, spacing_in_r0=spacing, jitter_in_r0=float(rng.uniform(0.0, 0.45)),
                           mask=0xffffffff if wide else 0xffff, origin_in_r0=(4.0, 4.0, 4.0), seed=77 + seed)
    cfg = sc["cfg"]
    N, nl = cfg.particleCount, sc["numOfLiquidP"]
    pos = sc["position"].copy()
    blobs = seed % 6 >= 3
    lo = pos[:nl, :3].min(0); hi = pos[:nl, :3].max(0)
    if blobs:
        for _ in range(3):
            centre = rng.uniform(lo, hi).astype(np.float32)
            members = rng.choice(nl, size=min(nl // 4, int(rng.integers(100, 700))), replace=False)
            pos[members, :3] = centre + rng.normal(0.0, rng.uniform(0.3, 1.2) * cfg.r0, size=(members.size, 3)).astype(np.float32)
        pos[:nl, :3] = np.clip(pos[:nl, :3], lo, hi)
    cell = np.float32(cfg.hashGridCellSize)
    on_face = rng.choice(nl, size=min(nl, 64), replace=False)
    axis = rng.integers(0, 3, size=on_face.size)
    k = np.maximum(np.round(pos[on_face, axis] / cell), 1.0).astype(np.float32)
    pos[on_face, axis] = k * cell
    sc["position"] = pos
    hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc, threads=8)
    if blobs:  # (inside a blob pairs get arbitrarily close: beyond the density pass the reference's forces overflow)
        for st in scenes.STAGE_SEQUENCE[:scenes.STAGE_SEQUENCE.index("computeDensity") + 1]:
            getattr(hip, scenes.HIP_STAGE_METHOD[st])()
            ora.run(st)
        got, want = canon_hip(hip, N), canon_ora(ora, N)
        for key in ("particleIndex", "gridCellIndexFixedUp", "neighborIds", "neighborDist", "rho"):
            assert scenes.bits_equal(got[key], want[key]), (seed, key, scenes.diff_report(got[key], want[key]))
    else:
        for it in range(3):
            hip.step(it)
            ora.step()
            assert_same(canon_hip(hip, N), canon_ora(ora, N), "random scene %d step %d" % (seed, it), FUSED_SKIP)


def test_compacted_keys_and_nine_bit_radix_digits():
    """Wide cell ids, a box whose reachable cells (42 x 41 x 43 = 74 k) need 17 key bits while the declared grid needs 20: the
    fused step sorts compacted keys in two passes of 9-bit digits instead of three of 8 (sph_sort.hip) and puts the real cell
    ids back afterwards. Permutation, cell ids, cell table and everything downstream must equal the oracle's."""
    sc = scenes.liquid_box((82.0, 80.0, 84.0), (30, 20, 30), mask=0xffffffff, jitter_in_r0=0.2, origin_in_r0=(40.0, 30.0, 50.0))
    N = sc["cfg"].particleCount
    hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc, threads=16)
    for it in range(2):
        hip.step(it)
        ora.step()
        assert_same(canon_hip(hip, N), canon_ora(ora, N), "compacted keys, step %d" % it, FUSED_SKIP)


def test_blown_up_state_is_reported():
    """Coincident particles make the reference divide by r = 0 (sphFluid.cl:1172-1178): their coordinates become NaN in the first
    step. The next step's hash kernel counts non-finite coordinates and the next blocking call fails loudly instead of the caller
    finding out from a search that has turned quadratic."""
    sc = scenes.liquid_box((8.0, 8.0, 8.0), (12, 10, 12), jitter_in_r0=0.03)
    pos = sc["position"].copy()
    pos[5:10, :3] = pos[200:205, :3]
    sc["position"] = pos
    hip = scenes.hip_for(sc)
    hip.step(0)
    hip.synchronize()  # the NaNs exist now, but nothing has hashed them yet
    hip.step(1)
    with pytest.raises(sphmi.SphError, match="not finite"):
        hip.synchronize()


@pytest.mark.parametrize("iterations", [1, 2, 4])
def test_other_predict_correct_iteration_counts(iterations):
    """maxIteration (owPhysicsConstant.h:76) is a run-time parameter here: the fused step must follow the oracle for
    1, 2 and 4 predict-correct iterations too (different fusion pattern: the last pressure force carries integrate)."""
    sc = scenes.SCENES["tiny_compressed"]()
    sc["cfg"].maxIteration = iterations
    N = sc["cfg"].particleCount
    hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc)
    for it in range(3):
        hip.step(it)
        ora.step()
    assert_same(canon_hip(hip, N), canon_ora(ora, N), "maxIteration=%d" % iterations, FUSED_SKIP)


def test_elastic_particles_behind_the_boundary_block():
    """elasticOffset = numOfBoundaryP (file-mode particle order, owOpenCLSolver.cpp:435), springs + muscle, no membranes."""
    sc = scenes.elastic_offset_box()
    N = sc["cfg"].particleCount
    hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc)
    for it in range(5):
        hip.step(it)
        ora.step()
        sig = sphmi.muscle_signal(it)
        hip.updateMuscleActivityData(sig)
        ora.update_muscles(sig)
    assert_same(canon_hip(hip, N), canon_ora(ora, N), "elastic offset", FUSED_SKIP)
    acc = canon_ora(ora, N)["acceleration"]
    assert np.abs(acc).max() > 0


def test_error_behaviour():
    """C-ABI error convention (include/sphmi.h): status codes, no exceptions across the ABI, order contract."""
    sc = scenes.SCENES["tiny"]()
    hip = scenes.hip_for(sc)
    with pytest.raises(sphmi.SphError, match="simulationStep order"):
        hip._runSort()  # before _runHashParticles
    hip._runHashParticles()
    with pytest.raises(sphmi.SphError):
        hip._runFindNeighbors()  # before sort / index
    with pytest.raises(sphmi.SphError):
        hip.updateMuscleActivityData(np.zeros(7, np.float32))
    with pytest.raises(sphmi.SphError):
        hip.buffer_raw = hip._chk(hip._L.sph_read_buffer(hip._h, b"noSuchBuffer", None, 0, None))
    bad = sphmi.default_config()
    bad.particleCount = 10
    bad.gridCellCount += 1
    with pytest.raises(sphmi.SphError):
        sphmi.owHIPSolver(bad, np.zeros((10, 4), np.float32), np.zeros((10, 4), np.float32))
    # the solver still works after the failed calls
    hip.step(0)
    assert np.isfinite(hip.read_position_buffer()).all()


def test_stage_timing_reports_every_stage():
    sc = scenes.SCENES["tiny"]()
    hip = scenes.hip_for(sc)
    hip.set_stage_timing(True)
    for it in range(3):
        hip.step(it)
    t = hip.stage_times()
    assert t["density"][1] == 3 and t["predict_density"][1] == 9 and t["find_neighbors"][1] == 3
    assert all(ms >= 0 for ms, _ in t.values())


def test_cpp_driver_matches_reference_fixture(tmp_path):
    """host/sphmi_run.cpp: the owOpenCLSolver-compatible C++ facade driven like owPhysicsFluidSimulator, on config #1
    text files (written here from the committed fixture), staged and fused, against the reference's positions."""
    import subprocess
    z = np.load(os.path.join(scenes.GOLDEN, "config1.npz"))
    sc = scenes.config1()
    pf, vf = tmp_path / "position.txt", tmp_path / "velocity.txt"
    np.savetxt(pf, sc["position"], fmt="%.9e", delimiter="\t")
    np.savetxt(vf, sc["velocity"], fmt="%.9e", delimiter="\t")
    exe = os.path.join(scenes.PKG, "sphmi_run")
    for extra in (["--staged"], []):
        out = tmp_path / "pos.bin"
        r = subprocess.run([exe, "--position", str(pf), "--velocity", str(vf), "--steps", "10", "--quiet", "--out", str(out)]
                           + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        pos = np.fromfile(out, np.float32).reshape(-1, 4)
        assert scenes.bits_equal(pos[z["sample_ids"]], z["position_sample_9"])


def test_cpp_driver_generated_worm_scene_matches_reference_fixture(tmp_path):
    """sphmi_run --worm: scene from the restated generator, muscle signals from sphmi_muscle_signal, fused
    steps through the C++ facade; positions after 10 steps against the compiled reference (same fixture as the Python test)."""
    import subprocess
    z = np.load(os.path.join(scenes.GOLDEN, "worm.npz"))
    out = tmp_path / "pos.bin"
    r = subprocess.run([os.path.join(scenes.PKG, "sphmi_run"), "--worm", "--muscles", "--steps", "10", "--quiet", "--out", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    pos = np.fromfile(out, np.float32).reshape(-1, 4)
    assert pos.shape[0] == 232887
    assert scenes.bits_equal(pos[z["sample_ids"]], z["position_sample_9"])


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the fields the driver reads (a tiny workload here; the default is config #4)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "tiny", "--steps", "3", "--warmup", "1",
                        "--cpu-steps", "2", "--no-extra"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "stages_ms", "stages_frac"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["scaling"] == "strong" and d["config"]["workload"] == "tiny" and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert abs(d["value"] - d["config"]["particles"] * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-2
    rf, cb = d["roofline"], d["cpu_baseline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "traffic" in rf and rf["bytes_per_launch"] == d["config"]["particles"] * 132
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb and cb["unit"] == "particle-steps/s"


# ---------------------------------------------------------------------------------------------- round 3
def _pressure_force_fallbacks(hip):
    """(wave, batch) pairs of k_pressure_force recomputed with sqrtf and `/` since the last reset_stage_times (dbg[8])."""
    return int(hip.buffer("debugCounters")[8])


def test_pressure_active_wide_million_particles_every_buffer():
    """Pressure-ACTIVE state at scale (the lattices at 0.93 r0 keep p = 0 for ~70 steps): 1.33 M liquid particles at 0.85 r0 — the
    `tiny_compressed` recipe — in a wide-mode box of 152,561 declared cells, three fused steps, EVERY buffer against the oracle.
    pcisph_correctPressure and the value / division arithmetic of pcisph_computePressureForceAcceleration
    (sphFluid.cl:1062-1098, 1160-1180) see non-zero pressures from the first iteration on, and they go through the short division /
    square-root sequences of sph_fastmath.h: the fallback counter must stay (almost) idle."""
    sc = scenes.liquid_box((60.0, 40.0, 60.0), (125, 85, 125), spacing_in_r0=0.85, mask=0xffffffff)
    cfg = sc["cfg"]
    N = cfg.particleCount
    assert N > 1300000 and cfg.gridCellCount > 65536
    hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc, threads=16)
    hip.reset_stage_times()
    for it in range(3):
        hip.step(it)
        ora.step()
        got, want = canon_hip(hip, N), canon_ora(ora, N)
        assert want["pressure"].max() > 0, "step %d: the scene must be pressure-active" % it
        assert_same(got, want, "compressed wide 1.3M, step %d" % it, FUSED_SKIP)
    assert (want["pressure"] > 0).sum() > 1000000
    acc_p = ora.buffer("acceleration").reshape(-1, 4)[N:, :3]
    assert np.abs(acc_p).max() > 0, "the pressure acceleration must be non-zero"
    batches = 3 * 3 * 4 * (N // 64)  # steps x launches x batches per wave x waves
    assert _pressure_force_fallbacks(hip) <= batches // 100, "the short division path must be the one that runs"
    ora.close()


def test_pressure_force_slow_path_is_taken_and_exact():
    """The guards of the short division / square-root path: a liquid particle whose x coordinate is closer to zero than 2^-4 (outside
    the boundary shell, inside the box) fails `ownOk`, so its wave recomputes every batch with sqrtf and `/` — same bits as the
    oracle, and the fallback counter says the slow path ran."""
    sc = scenes.SCENES["tiny_compressed"]()
    pos = sc["position"].copy()
    pos[0, 0] = np.float32(0.03)
    sc["position"] = pos
    N = sc["cfg"].particleCount
    hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc)
    hip.reset_stage_times()
    for it in range(3):
        hip.step(it)
        ora.step()
        assert_same(canon_hip(hip, N), canon_ora(ora, N), "slow path, step %d" % it, FUSED_SKIP)
    assert canon_ora(ora, N)["pressure"].max() > 0
    assert _pressure_force_fallbacks(hip) > 0
    ora.close()


def test_config4_box_pressure_active_full_steps_against_oracle():
    """The config #4 box (16,507,704 particles) with its lattice at 0.85 r0: pressure-active from the first step. Two fused steps,
    positions, velocities, densities and pressures of every particle against the oracle, bit for bit."""
    sc = scenes.liquid_box((78.0, 50.0, 470.0), (160, 100, 1000), spacing_in_r0=0.85, mask=0xffffffff)
    N = sc["cfg"].particleCount
    assert N == 16507704
    hip, ora = scenes.hip_for(sc), scenes.oracle_for(sc, threads=16)
    hip.reset_stage_times()
    for it in range(2):
        hip.step(it)
        ora.step()
        assert scenes.bits_equal(hip.read_position_buffer(), ora.buffer("position").reshape(-1, 4)[:N]), it
        assert scenes.bits_equal(hip.read_velocity_buffer(), ora.buffer("velocity").reshape(-1, 4)[:N]), it
        assert scenes.bits_equal(hip.read_density_buffer(), ora.buffer("rho").reshape(-1)[:N]), it
        p = ora.buffer("pressure").reshape(-1)[:N]
        assert scenes.bits_equal(hip.buffer("pressure").reshape(-1)[:N], p), it
        assert (p > 0).sum() > 10000000, "step %d: pressure-active" % it
    assert _pressure_force_fallbacks(hip) <= 2 * 3 * 4 * (N // 64) // 100
    ora.close()


def test_short_division_and_square_root_sequences_are_exact():
    """sph_fastmath.h replaces IEEE `/` and sqrtf in the hottest arithmetic kernel (k_pressure_force). tools/micro/exact_div_sqrt
    (built by build(); hipcc also exists on the GPU box) compares them with the compiler's correctly rounded code on the GPU:
    (i) the square root for EVERY float of [fastD2Min, fastD2Max] of the shipped simulationScale (~8e8 floats) plus the guard's
    edges, (ii) the division for every mantissa of r at three exponents with numerators placed next to rounding midpoints of the
    quotient and next to / at floats (~1.8e9 quotients), (iii) 1.07e9 random operand sets. Zero differences."""
    import subprocess
    tool_dir = os.path.join(scenes.ROOT, "tools", "micro")
    exe = os.path.join(tool_dir, "exact_div_sqrt")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", tool_dir, "exact_div_sqrt"])
    scale = float(np.float32(sphmi.default_config().simulationScale))
    p = subprocess.run([exe, "suite", repr(scale), "1"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 6 and all(" 0 differ" in l or " 0 results differ" in l for l in lines[1:]), p.stdout
    assert "every float of the guarded range" in lines[1]


def test_async_position_read_back_overlaps_and_matches():
    """sph_read_position_async / sph_read_position_wait: the copy of step t runs under step t+1 and the next integrate waits for it
    on the device. Step-by-step the asynchronously read positions must equal the blocking read of the same step (bit for bit),
    whether the caller waits at once, a step late, or only at the end; getPosition_cpp() of the simulator waits by itself."""
    sc = scenes.liquid_box((30.0, 20.0, 30.0), (40, 25, 40), jitter_in_r0=0.05)
    N = sc["cfg"].particleCount
    ref = scenes.hip_for(sc)
    want = []
    for it in range(6):
        ref.step(it)
        want.append(ref.read_position_buffer().copy())
    ref.close()
    hip = scenes.hip_for(sc)
    a, b = np.empty((N, 4), np.float32), np.empty((N, 4), np.float32)
    hip.step(0); hip.read_position_buffer_async(a); hip.wait_position_buffer()
    assert scenes.bits_equal(a, want[0])
    hip.step(1); hip.read_position_buffer_async(a)
    hip.step(2)                                  # enqueued while the copy of step 1 is in flight: integrate(2) waits for it
    hip.wait_position_buffer()
    assert scenes.bits_equal(a, want[1]), "the copy of step 1 must not see step 2's positions"
    hip.read_position_buffer_async(b)            # step 2, second (page-locked) buffer
    hip.step(3); hip.read_position_buffer_async(a)   # waits for b's copy first
    assert scenes.bits_equal(b, want[2])
    hip.step(4); hip.step(5)
    hip.wait_position_buffer()
    assert scenes.bits_equal(a, want[3])
    assert scenes.bits_equal(hip.read_position_buffer(), want[5])
    hip.close()
    sim = sphmi.owPhysicsFluidSimulator(sc["cfg"], sc["position"], sc["velocity"])
    for it in range(3):
        sim.simulationStep(async_read_back=True)
    assert scenes.bits_equal(sim.getPosition_cpp(), want[2])


def test_async_position_read_back_with_membranes():
    """The worm scene: positions are written by integrate AND by the membrane finalize pass of the same step; both must wait for a
    read-back that is still in flight. Copies collected one step late equal the blocking reads of a twin run."""
    sc = scenes.worm_scene()
    N = sc["cfg"].particleCount
    ref, hip = scenes.hip_for(sc), scenes.hip_for(scenes.worm_scene())
    want = []
    for it in range(4):
        ref.step(it)
        ref.updateMuscleActivityData(sphmi.muscle_signal(it))
        want.append(ref.read_position_buffer().copy())
    bufs = [np.empty((N, 4), np.float32), np.empty((N, 4), np.float32)]
    for it in range(4):
        hip.step(it)
        hip.updateMuscleActivityData(sphmi.muscle_signal(it))
        hip.read_position_buffer_async(bufs[it & 1])      # (waits for the copy of step it - 1 first)
        if it:
            assert scenes.bits_equal(bufs[(it - 1) & 1], want[it - 1]), "step %d" % (it - 1)
    hip.wait_position_buffer()
    assert scenes.bits_equal(bufs[1], want[3])


def test_box_that_starts_below_zero_keeps_real_sort_keys():
    """Wide cell ids with xmin, ymin, zmin < 0 (no particle there): the compacted sort keys of the fused step are only monotone in
    the real cell ids while their clamps cannot bite, so such a box must sort the real keys — and still follow the oracle."""
    def build():
        sc = scenes.liquid_box((82.0, 80.0, 84.0), (30, 20, 30), mask=0xffffffff, jitter_in_r0=0.2, origin_in_r0=(40.0, 30.0, 50.0))
        sc["cfg"].xmin = sc["cfg"].ymin = sc["cfg"].zmin = -np.float32(sc["cfg"].h)
        return sc
    plain = scenes.liquid_box((82.0, 80.0, 84.0), (30, 20, 30), mask=0xffffffff, jitter_in_r0=0.2, origin_in_r0=(40.0, 30.0, 50.0))
    sc = build()
    N = sc["cfg"].particleCount
    hip, ora, hp = scenes.hip_for(sc), scenes.oracle_for(sc, threads=16), scenes.hip_for(plain)
    assert hp.step_sort_passes() == 2 and hip.step_sort_passes() == 3  # compacted (17 bits, 9-bit digits) vs real keys (20 bits)
    for it in range(2):
        hip.step(it)
        ora.step()
        assert_same(canon_hip(hip, N), canon_ora(ora, N), "box from -h, step %d" % it, FUSED_SKIP)


def test_blown_up_state_stays_reported():
    """The non-finite-coordinate error is sticky (a caller that ignores one SPH_ERR_INVALID must not continue silently) and a timing
    reset between the step and the check does not erase it."""
    sc = scenes.liquid_box((8.0, 8.0, 8.0), (12, 10, 12), jitter_in_r0=0.03)
    pos = sc["position"].copy()
    pos[5:10, :3] = pos[200:205, :3]
    sc["position"] = pos
    hip = scenes.hip_for(sc)
    hip.step(0)
    hip.step(1)
    hip.reset_stage_times()
    for _ in range(2):
        with pytest.raises(sphmi.SphError, match="not finite"):
            hip.synchronize()
    with pytest.raises(sphmi.SphError, match="not finite"):
        hip.read_position_buffer()
