"""CPU tests that pin the oracle (oracle/sph_oracle.c):
  * against the committed fixtures that the reference's own kernels produced (tests/golden/*.npz), and
  * where oracle/_ref/libsphref.so exists (the build container), live against those kernels, buffer by buffer.
The reference ships no tests or golden vectors of its own."""
import json
import os

import numpy as np
import pytest

import scenes
import sphmi

REF_SCENES = [n for n in scenes.SCENES if n != "wide"]


def load_golden(name):
    z = np.load(os.path.join(scenes.GOLDEN, name + ".npz"))
    return json.loads(str(z["meta"])), z


@pytest.mark.parametrize("name", REF_SCENES)
def test_oracle_matches_reference_fixture_stage_by_stage(name):
    meta, z = load_golden(name)
    sc = scenes.SCENES[name]()
    N = sc["cfg"].particleCount
    assert N == meta["N"]
    assert scenes.sha(sc["position"]) == meta["input_position_sha"], "scene generator drifted from the fixture"
    assert scenes.sha(sc["velocity"]) == meta["input_velocity_sha"]
    T = scenes.oracle_for(sc)
    for it in range(meta["steps"]):
        per_stage = meta["stage_hashes"].get(str(it))
        if per_stage is not None:
            for k, st in enumerate(scenes.STAGE_SEQUENCE):
                T.run(st)
                got = {b: scenes.sha(v) for b, v in scenes.canonical(T.buffer, N).items()}
                bad = [b for b in got if got[b] != per_stage[k][b]]
                assert not bad, "step %d stage %d (%s): %s differ from the reference" % (it, k, st, bad)
        else:
            T.step()
        if sc["elastic"] is not None:
            T.update_muscles(sphmi.muscle_signal(it))
        got = {b: scenes.sha(v) for b, v in scenes.canonical(T.buffer, N).items()}
        bad = [b for b in got if got[b] != meta["step_hashes"][str(it)][b]]
        assert not bad, "after step %d: %s differ from the reference" % (it, bad)
    if "position_10" in z:
        c = scenes.canonical(T.buffer, N)
        assert scenes.bits_equal(c["position"], z["position_10"])
        assert scenes.bits_equal(c["velocity"], z["velocity_10"])
        assert scenes.bits_equal(c["neighborIds"], z["neighborIds_10"])


def test_oracle_config1_against_reference_fixture():
    """BASELINE config #1 (shipped pure-liquid files, N = 61,440): hashes after steps 0 and 9 and the survey's
    density statistic (liquid mean rho at step 0 = 844.214029)."""
    meta, z = load_golden("config1")
    sc = scenes.config1()
    N = sc["cfg"].particleCount
    T = scenes.oracle_for(sc, threads=8)
    for it in range(10):
        T.step()
        if it == 0:
            rho = T.buffer("rho")[:N]
            pib = T.buffer("particleIndexBack")
            liq = sc["position"][:, 3].astype(int) == 1
            mean = float(rho[pib[liq]].astype(np.float64).mean())
            assert abs(mean - 844.214029) < 1e-5 and abs(mean - meta["liquid_rho_mean_step0"]) < 1e-9
        if str(it) in meta["step_hashes"]:
            got = {b: scenes.sha(v) for b, v in scenes.canonical(T.buffer, N).items()}
            bad = [b for b in got if got[b] != meta["step_hashes"][str(it)][b]]
            assert not bad, "config1 step %d: %s" % (it, bad)
            pos = T.buffer("position").reshape(-1, 4)
            assert scenes.bits_equal(pos[z["sample_ids"]], z["position_sample_%d" % it])


def test_oracle_worm_scene_against_reference_fixture():
    """BASELINE config #3 (generated C. elegans scene: liquid + springs + muscles + membranes, 41,700 particles aliased
    by the 16-bit cell ids), 5 steps with the muscle signal switched on."""
    meta, z = load_golden("worm")
    sc = scenes.worm_scene()
    N = sc["cfg"].particleCount
    assert N == 232887 and meta["counts"]["numOfElasticP"] == 10143 and meta["counts"]["numOfMembranes"] == 11386
    for k in ("position", "velocity", "elastic", "membranes", "particle_membranes"):
        assert scenes.sha(sc[k]) == meta["input_sha"][k]
    T = scenes.oracle_for(sc, threads=8)
    for it in range(5):
        T.step()
        T.update_muscles(sphmi.muscle_signal(it))
        if str(it) in meta["step_hashes"]:
            got = {b: scenes.sha(v) for b, v in scenes.canonical(T.buffer, N).items()}
            bad = [b for b in got if got[b] != meta["step_hashes"][str(it)][b]]
            assert not bad, "worm step %d: %s" % (it, bad)
            pos = T.buffer("position").reshape(-1, 4)
            assert scenes.bits_equal(pos[z["sample_ids"]], z["position_sample_%d" % it])


def test_oracle_thread_count_does_not_change_results():
    sc = scenes.SCENES["tiny_elastic"]()
    N = sc["cfg"].particleCount
    A, B = scenes.oracle_for(sc, threads=1), scenes.oracle_for(sc, threads=8)
    for it in range(4):
        A.step(); B.step()
    ca, cb = scenes.canonical(A.buffer, N), scenes.canonical(B.buffer, N)
    for k in ca:
        assert scenes.bits_equal(ca[k], cb[k]), k


def test_wide_mode_equals_reference_mode_when_nothing_aliases():
    """Wide ids (cellIdMask = 0xffffffff) change nothing while raw cell ids stay below 65,536: the transitive
    argument that validates the wide-mode oracle used for the 16M / 64M boxes."""
    a = scenes.liquid_box((8.0, 8.0, 8.0), (12, 10, 12), jitter_in_r0=0.03)
    b = scenes.liquid_box((8.0, 8.0, 8.0), (12, 10, 12), jitter_in_r0=0.03, mask=0xffffffff)
    N = a["cfg"].particleCount
    A, B = scenes.oracle_for(a), scenes.oracle_for(b)
    for it in range(5):
        A.step(); B.step()
    ca, cb = scenes.canonical(A.buffer, N), scenes.canonical(B.buffer, N)
    for k in ca:
        assert scenes.bits_equal(ca[k], cb[k]), k


def test_wide_mode_differs_where_reference_aliases():
    a, b = scenes.SCENES["alias16"](), scenes.SCENES["wide"]()
    N = a["cfg"].particleCount
    A, B = scenes.oracle_for(a, 8), scenes.oracle_for(b, 8)
    A.step(); B.step()
    ka = A.buffer("particleIndex").reshape(-1, 2)[:, 0]
    kb = B.buffer("particleIndex").reshape(-1, 2)[:, 0]
    assert ka.max() < 65536 <= kb.max()
    # sortedness + permutation properties hold in both modes
    for S in (A, B):
        pi = S.buffer("particleIndex").reshape(-1, 2)
        assert np.all(np.diff(pi[:, 0].astype(np.int64)) >= 0)
        same = pi[1:, 0] == pi[:-1, 0]
        assert np.all(pi[1:, 1][same] > pi[:-1, 1][same])  # ascending orig id inside a cell (App. B #4)
        assert np.array_equal(np.sort(pi[:, 1]), np.arange(N, dtype=np.uint32))


def _ref_available():
    from oracle import refbind
    return refbind.available()


@pytest.mark.skipif(not _ref_available(), reason="oracle/_ref/libsphref.so only exists in the build container")
@pytest.mark.parametrize("name", ["tiny_jitter", "tiny_elastic", "elastic_offset"])
def test_oracle_live_against_reference_kernels(name):
    from oracle import refbind as R
    sc = scenes.elastic_offset_box() if name == "elastic_offset" else scenes.SCENES[name]()
    c = sc["cfg"]
    N = c.particleCount
    S = R.RefSolver(sc["position"], sc["velocity"], (c.xmax, c.ymax, c.zmax), (c.gridCellsX, c.gridCellsY, c.gridCellsZ),
                    elastic=sc["elastic"], membranes=sc["membranes"], particle_membranes=sc["particle_membranes"],
                    elastic_offset=c.elasticOffset, threads=4)
    T = scenes.oracle_for(sc)
    names = ["position", "velocity", "sortedPosition", "sortedVelocity", "acceleration", "neighborMap", "particleIndex",
             "particleIndexBack", "gridCellIndex", "gridCellIndexFixedUp", "pressure", "rho"]
    stages = list(scenes.STAGE_SEQUENCE)
    if sc["particle_membranes"] is None and sc["elastic"] is not None:
        # elastic matter without membrane lists: the reference creates no membrane buffers (owOpenCLSolver.cpp:68-81) and
        # its kernel would dereference them; the membrane stage is left out on both sides
        stages.remove("computeInteractionWithMembranes")
    for it in range(3):
        for st in stages:
            S.run(st, it); T.run(st)
            for b in names:  # every word of every buffer, including the dead .w lanes
                assert scenes.bits_equal(S.buffer(b), T.buffer(b)), (it, st, b, scenes.diff_report(S.buffer(b), T.buffer(b)))
        if sc["elastic"] is not None:
            sig = sphmi.muscle_signal(it)
            S.update_muscles(sig); T.update_muscles(sig)
