"""CPU tests of the host layer: constants pinned to the reference compiler's evaluation, loaders, scene helpers,
and that both shared libraries load and export every symbol the headers declare (no compute calls without a GPU)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import scenes
import sphmi

GOLDEN = scenes.GOLDEN


def test_constants_match_reference_compiler():
    """include/sphmi_host.h: sphmi_default_config == owPhysicsConstant.h as evaluated by the reference build
    (tests/golden/constants.json, written by make_golden.py from oracle/_ref)."""
    ref = json.load(open(os.path.join(GOLDEN, "constants.json")))
    cfg = sphmi.default_config()
    for name, bits in ref["float_bits"].items():
        if name in ("stiffness", "damping"):  # passed to kernels but never used in arithmetic
            continue
        got = int(np.float32(getattr(cfg, name)).view(np.uint32))
        assert got == bits, "%s: %08x != %08x" % (name, got, bits)
    for name, hx in ref["double_hex"].items():
        if name == "beta":
            continue
        assert float(getattr(cfg, name)).hex() == hx, name
    for name in ("gridCellsX", "gridCellsY", "gridCellsZ", "gridCellCount", "maxIteration"):
        assert getattr(cfg, name) == ref[name]
    # pinned bit patterns of SURVEY Appendix A
    assert int(np.float32(cfg.delta).view(np.uint32)) == 0x3E80422C
    assert int(np.float32(cfg.simulationScale).view(np.uint32)) == 0x3607FBE2
    assert (cfg.gridCellsX, cfg.gridCellsY, cfg.gridCellsZ, cfg.gridCellCount) == (31, 21, 251, 163401)


def test_surface_tension_coefficient_matches_oracle():
    from oracle import oraclebind as O
    cfg = sphmi.default_config()
    want = O.lib().sph_oracle_surf_tens_coeff(cfg.Wpoly6Coefficient, cfg.h, cfg.simulationScale)
    assert np.float32(cfg.surfTensCoeff).view(np.uint32) == np.float32(want).view(np.uint32)


def test_box_generator_reproduces_reference_boundary_shell():
    """The boundary part of sphmi_generate_box == the last numOfBoundaryP particles of the reference generator's
    output (tests/golden/worm_input.npz), bit for bit: positions, normals stored in velocity, type 3 in both .w."""
    w = np.load(os.path.join(GOLDEN, "worm_input.npz"))
    cfg = sphmi.default_config()
    pos, vel, cnt = sphmi.generate_box(cfg, 0, 0, 0)
    nb = cnt["numOfBoundaryP"]
    assert nb == 102408
    assert scenes.bits_equal(pos, w["position"][-nb:])
    assert scenes.bits_equal(vel, w["velocity"][-nb:])


def test_box_sizes_of_survey_configs():
    """config #2 N=1,058,808 (grid 51^3), config #4 N=16,507,704, config #5 N=65,469,208."""
    for box, lat, n, grid in (((50.0, 50.0, 50.0), (100, 100, 100), 1058808, (51, 51, 51)),
                              ((78.0, 50.0, 470.0), (160, 100, 1000), 16507704, (79, 51, 471)),
                              ((240.0, 200.0, 310.0), (250, 400, 640), 65469208, (241, 201, 311))):
        cfg = sphmi.default_config()
        sphmi.set_box(cfg, *box, 0xffffffff)
        nl, nb = ctypes.c_int(), ctypes.c_int()
        assert sphmi.host_lib().sphmi_box_counts(ctypes.byref(cfg), box[0], box[1], box[2], lat[0], lat[1], lat[2],
                                                 ctypes.byref(nl), ctypes.byref(nb)) == 0
        assert nl.value + nb.value == n
        assert (cfg.gridCellsX, cfg.gridCellsY, cfg.gridCellsZ) == grid


def test_text_loader_roundtrip(tmp_path):
    """owHelper::preLoadConfiguration/loadConfiguration format: `x y z type` per line, tab separated %e,
    last line without a newline is still a particle (owHelper.cpp:1441-1447)."""
    z = np.load(os.path.join(GOLDEN, "config1_input.npz"))
    pos, vel = z["position"][:300], z["velocity"][:300]
    pf, vf = tmp_path / "position.txt", tmp_path / "velocity.txt"
    pf.write_text("\n".join("\t".join("%e" % v for v in row) for row in pos))
    vf.write_text("\n".join("\t".join("%e" % v for v in row) for row in vel) + "\n")
    p2, v2, cnt = sphmi.load_configuration(str(pf), str(vf))
    assert p2.shape == (300, 4) and cnt["numOfBoundaryP"] + cnt["numOfLiquidP"] == 300
    np.testing.assert_allclose(p2, pos, rtol=1e-6)
    np.testing.assert_allclose(v2, vel, rtol=1e-6, atol=1e-12)
    with pytest.raises(sphmi.SphError):
        sphmi.load_configuration(str(tmp_path / "missing.txt"), str(vf))


def test_loaders_match_the_reference_loader_fixture():
    """sphmi_load_configuration / sphmi_load_elastic_connections against owHelper::loadConfiguration ITSELF (compiled into
    oracle/_ref; fixture tests/golden/ref_loader.npz made by tests/golden/make_io_golden.py): a configuration in the file order of
    the shipped position.txt (boundary, elastic, liquid) with awkward number formats, blanks instead of tabs, leading blanks and
    a last line without newline; the connection file is one the reference's own writer produced. Bit for bit."""
    lo = os.path.join(GOLDEN, "ref_loader")
    want = np.load(os.path.join(GOLDEN, "ref_loader.npz"))
    pos, vel, cnt = sphmi.load_configuration(os.path.join(lo, "position.txt"), os.path.join(lo, "velocity.txt"))
    assert scenes.bits_equal(pos, want["position"]) and scenes.bits_equal(vel, want["velocity"])
    assert [cnt["numOfLiquidP"], cnt["numOfElasticP"], cnt["numOfBoundaryP"]] == want["counts"].tolist()
    ne = cnt["numOfElasticP"]
    con = np.full((32 * ne, 4), np.float32(7.0))
    con = con.astype(np.float32)
    rows = sphmi.host_lib().sphmi_load_elastic_connections(os.path.join(lo, "elasticconnections.txt").encode(), ne, con.ctypes.data)
    assert rows == 32 * ne
    assert scenes.bits_equal(con, want["elastic"])
    # the committed config #1 inputs are what the reference loader reads from the shipped PureLiquid files
    c1 = np.load(os.path.join(GOLDEN, "config1_input.npz"))
    assert c1["position"].shape == (61440, 4) and int((c1["position"][:, 3].astype(int) == 3).sum()) == 32834


def test_trajectory_dump_matches_the_reference_writer_fixture(tmp_path):
    """sphmi_save_configuration against owHelper::loadConfigurationToFile ITSELF (tests/golden/ref_dump/*.txt, written by the
    reference's code in oracle/_ref from the arrays in ref_dump_input.npz): header, two frames, connection and membrane files,
    byte for byte — including operator<< number formatting (1e-07, 123457, 0). Documented deviation: the reference reads its
    membrane array with stride 4; it was given [i, j, k, 0] rows, sphmi_save_configuration takes the stride-3 table."""
    z = np.load(os.path.join(GOLDEN, "ref_dump_input.npz"))
    sphmi.save_configuration(str(tmp_path), z["position"], 3, 5, z["connections"], z["membranes"], first=True)
    sphmi.save_configuration(str(tmp_path), z["position2"], 3, 5, first=False)
    for f in ("position_buffer.txt", "connection_buffer.txt", "membranes_buffer.txt"):
        assert (tmp_path / f).read_bytes() == open(os.path.join(GOLDEN, "ref_dump", f), "rb").read(), f


def test_trajectory_dump_format(tmp_path):
    """buffers/*.txt of the reference's -l_to mode (owHelper.cpp:1640-1672): header, non-boundary particles only, appended
    frames, default operator<< float formatting (6 significant digits)."""
    sc = scenes.SCENES["tiny_elastic"]()
    n_el, n_liq = sc["numOfElasticP"], sc["numOfLiquidP"]
    sphmi.save_configuration(str(tmp_path), sc["position"], n_el, n_liq, sc["elastic"], sc["membranes"], first=True)
    sphmi.save_configuration(str(tmp_path), sc["position"], n_el, n_liq, first=False)
    lines = (tmp_path / "position_buffer.txt").read_text().split("\n")
    assert lines[0] == str(n_el) and lines[1] == str(n_liq)
    body = [l for l in lines[2:] if l]
    assert len(body) == 2 * (n_el + n_liq)  # two frames, boundary particles skipped
    first = np.array([float(v) for v in body[0].split("\t")], np.float32)
    np.testing.assert_allclose(first, sc["position"][0], rtol=1e-5)
    assert body[0].split("\t")[3] == "2.1"  # 6 significant digits, no padding
    con = (tmp_path / "connection_buffer.txt").read_text().strip().split("\n")
    assert len(con) == 32 * n_el and con[-1].split("\t")[0] == "-1"
    mem = (tmp_path / "membranes_buffer.txt").read_text().strip().split("\n")
    assert int(mem[0]) == sc["membranes"].shape[0] and [int(v) for v in mem[1].split("\t")] == list(sc["membranes"][0]) + [0]


def test_trajectory_playback_reader_roundtrip(tmp_path):
    """owHelper::loadConfigurationFromFile (owHelper.cpp:1674-1741): the `-l_from` reader gets back what the `-l_to` writer
    dumped — counts, frame count, every row to the 6 significant digits the text holds, connections and membranes."""
    sc = scenes.SCENES["tiny_elastic"]()
    n_el, n_liq = sc["numOfElasticP"], sc["numOfLiquidP"]
    moved = sc["position"].copy()
    moved[:, :3] += np.float32(0.125)
    sphmi.save_configuration(str(tmp_path), sc["position"], n_el, n_liq, sc["elastic"], sc["membranes"], first=True)
    sphmi.save_configuration(str(tmp_path), moved, n_el, n_liq, first=False)
    sphmi.save_configuration(str(tmp_path), sc["position"], n_el, n_liq, first=False)
    tr = sphmi.load_trajectory(str(tmp_path))
    assert (tr["numOfElasticP"], tr["numOfLiquidP"]) == (n_el, n_liq) and tr["frames"].shape == (3, n_el + n_liq, 4)
    keep = sc["position"][:, 3].astype(np.int32) != 3
    np.testing.assert_allclose(tr["frames"][0], sc["position"][keep], rtol=1e-5)
    np.testing.assert_allclose(tr["frames"][1], moved[keep], rtol=1e-5)
    assert np.array_equal(tr["frames"][2], tr["frames"][0])
    np.testing.assert_allclose(tr["connections"], sc["elastic"], rtol=1e-5)
    assert np.array_equal(tr["membranes"][:, :3], sc["membranes"]) and np.all(tr["membranes"][:, 3] == 0)
    with pytest.raises(sphmi.SphError):
        sphmi.load_trajectory(str(tmp_path / "missing"))


def test_muscle_signal_matches_reference_generator():
    """sphmi_muscle_signal against values produced by the reference's own main_sim.py (fixture
    tests/golden/muscle_signal.npz, generated by tests/golden/make_muscle_golden.py), narrowed to float as
    PyramidalSimulation::unpackPythonList does: bit-identical at every sampled step."""
    z = np.load(os.path.join(scenes.GOLDEN, "muscle_signal.npz"))
    for t, want in zip(z["steps"], z["signal_f32"]):
        s = sphmi.muscle_signal(int(t))
        assert s.shape == (100,) and np.all(s[96:] == 0)      # MUSCLE_COUNT = 100, entries 96..99 stay 0
        assert scenes.bits_equal(s[:96], want), "step %d" % t
    s0, s5 = sphmi.muscle_signal(0), sphmi.muscle_signal(50000)
    assert np.abs(s5[:96] - s0[:96]).max() > 0.1  # the wave travels


def _declared(header, prefix):
    txt = open(os.path.join(scenes.ROOT, "include", header)).read()
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, txt)))


def test_host_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(sphmi.HOST_LIB_PATH)
    names = _declared("sphmi_host.h", "sphmi_")
    assert set(names) == set(sphmi.HOST_EXPORTED_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n


def test_device_library_loads_and_exports_every_declared_symbol():
    """libsphmi.so must load on a machine without a GPU (symbols only; no compute call is made here)."""
    lib = ctypes.CDLL(sphmi.LIB_PATH)
    names = [n for n in _declared("sphmi.h", "sph_") if n not in ("sph_status", "sph_stage")]
    assert set(names) == set(sphmi.EXPORTED_SYMBOLS), set(names) ^ set(sphmi.EXPORTED_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n
    lib.sph_abi_version.restype = ctypes.c_int
    assert lib.sph_abi_version() == sphmi.ABI_VERSION


def test_create_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a HIP device the constructor raises (SPH_ERR_HIP) instead of computing."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sc = scenes.SCENES["tiny"]()
    with pytest.raises(sphmi.SphError):
        scenes.hip_for(sc)


def test_create_rejects_non_finite_and_out_of_box_input():
    """sph_create validates before it touches the device: non-finite coordinates in any mode, out-of-box coordinates in wide mode
    (the radix sort orders only the bits an in-grid cell id can have)."""
    for mask, bad, what in ((0xffff, np.nan, "not finite"), (0xffffffff, -1.0, "outside the box"), (0xffffffff, np.inf, "not finite")):
        sc = scenes.liquid_box((8.0, 8.0, 8.0), (6, 6, 6), mask=mask)
        pos = sc["position"].copy()
        pos[5, 1] = bad
        with pytest.raises(sphmi.SphError, match=what):
            sphmi.owHIPSolver(sc["cfg"], pos, sc["velocity"])


def test_config_struct_layout_matches_header():
    """ctypes mirror of sph_config has the size the C compiler gives the header's struct."""
    import subprocess, tempfile
    src = '#include <stdio.h>\n#include "sphmi.h"\nint main(){printf("%zu %zu\\n", sizeof(sph_config), __builtin_offsetof(sph_config, stream));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(scenes.ROOT, "include"), c, "-o", exe])
        size, off = map(int, subprocess.check_output([exe]).split())
    assert ctypes.sizeof(sphmi.SphConfig) == size
    assert sphmi.SphConfig.stream.offset == off


def test_frame_feed_inversion_and_writers(tmp_path):
    """the viewer's inverse-permutation contract and the headless frame writers (no GPU needed)."""
    from sphmi import frames
    rng = np.random.default_rng(5)
    n = 1000
    perm = rng.permutation(n).astype(np.uint32)            # sorted position -> orig id
    pi = np.stack([np.sort(rng.integers(0, 50, n)).astype(np.uint32), perm], axis=1)
    back = frames.invert_particle_index(pi)
    # the reference's loop: p_indexb[2*p_indexb[2*i+1]+0] = i  (owWorldSimulation.cpp:112-116)
    ref = np.empty(n, np.uint32)
    for i in range(n):
        ref[pi[i, 1]] = i
    assert np.array_equal(back, ref)
    pos = rng.random((n, 4)).astype(np.float32)
    pos[:, 3] = np.where(np.arange(n) % 3 == 0, np.float32(3.1), np.float32(1.1))
    rho = (1000 + 60 * rng.random(n)).astype(np.float32)
    kept = frames.write_vtk(tmp_path / "f.vtk", pos, rho)
    assert kept == int((pos[:, 3].astype(np.int32) != 3).sum())
    raw = open(tmp_path / "f.vtk", "rb").read()
    off = raw.index(b"POINTS") + len(("POINTS %d float\n" % kept).encode())
    xyz = np.frombuffer(raw[off:off + 12 * kept], ">f4").reshape(-1, 3)
    assert np.array_equal(xyz.astype(np.float32), pos[pos[:, 3].astype(np.int32) != 3][:, :3])
    frames.write_npz(tmp_path / "f.npz", pos, rho, step=7)
    z = np.load(tmp_path / "f.npz")
    assert np.array_equal(z["position"], pos) and np.array_equal(z["density"], rho) and int(z["step"]) == 7
    rgb = frames.density_colour(np.float32([990, 1005, 1015, 1025, 1035, 1100]), 1000.0)
    assert np.allclose(rgb[0], (0, 0, 1)) and np.allclose(rgb[1], (0, 0.5, 1)) and np.allclose(rgb[2], (0, 1, 0.5))
    assert np.allclose(rgb[3], (0.5, 1, 0)) and np.allclose(rgb[4], (1, 0.5, 0)) and np.allclose(rgb[5], (1, 0, 0))


def test_worm_generator_matches_reference_fixture():
    """sphmi_generate_worm restates owHelper::generateConfiguration (owHelper.cpp:104-1429). Pinned against the
    output of the compiled reference generator (tests/golden/worm_input.npz, made by tests/golden/make_golden.py): particle
    counts, every position / velocity bit, all 137,804 springs with their rest lengths and muscle colours, the 11,386
    membrane triangles and the per-particle triangle lists."""
    z = np.load(scenes.worm_scene_path())
    cfg = sphmi.default_config()
    sc = sphmi.generate_worm(cfg)
    assert (sc["numOfElasticP"], sc["numOfLiquidP"], sc["numOfBoundaryP"]) == (10143, 120336, 102408)
    assert cfg.particleCount == 232887 and cfg.numOfElasticP == 10143 and cfg.numOfMembranes == 11386
    for mine, ref in (("position", "position"), ("velocity", "velocity"), ("elastic", "elastic")):
        assert scenes.bits_equal(sc[mine], z[ref]), mine
    assert np.array_equal(sc["membranes"], z["membranes"])
    assert np.array_equal(sc["particle_membranes"], z["particle_membranes"])
    el = sc["elastic"]
    live = el[:, 0] >= 0
    assert int(live.sum()) == 137804
    muscles = np.unique(el[live, 2].astype(np.int32))
    assert muscles[0] == 0 and set(range(1, 97)) <= set(muscles.tolist())  # 96 muscle groups, quadrants of 24


def test_worm_generator_other_box():
    """A different box moves the worm axis and changes the liquid / boundary counts; structure stays consistent."""
    cfg = sphmi.default_config()
    sphmi.set_box(cfg, 30.0, 20.0, 260.0, 0xffff)
    sc = sphmi.generate_worm(cfg)
    n = cfg.particleCount
    assert n == sc["numOfElasticP"] + sc["numOfLiquidP"] + sc["numOfBoundaryP"] and sc["numOfElasticP"] == 10143
    t = sc["position"][:, 3].astype(np.int32)
    assert np.all(t[:10143] == 2) and np.all(t[10143:10143 + sc["numOfLiquidP"]] == 1) and np.all(t[-sc["numOfBoundaryP"]:] == 3)
    assert sc["membranes"].min() >= 0 and sc["membranes"].max() < 10143
    ids = sc["elastic"][:, 0]
    assert np.all((ids == -1) | ((ids >= 0) & (ids < n)))
