"""Generates the text I/O fixtures with the REFERENCE's own loader and writer — owHelper::loadConfiguration
(owHelper.cpp:1431-1545) and owHelper::loadConfigurationToFile (:1640-1672), compiled unmodified into oracle/_ref/libsphref.so —
so that sphmi_load_configuration / sphmi_load_elastic_connections / sphmi_save_configuration are pinned against the reference's
code, not against themselves. Build container only (needs /root/reference and oracle/_ref). Outputs, all small:

  tests/golden/config1_input.npz       positionPureLiquid.txt + velocityPureLiquid.txt as the reference loader reads them
  tests/golden/ref_loader/*.txt        a small hand-made configuration (boundary, elastic, liquid; tricky number formats)
  tests/golden/ref_loader.npz          what the reference loader returns for it (positions, velocities, connections, counts)
  tests/golden/ref_dump/*.txt          what the reference writer produces for a small scene, two frames
  tests/golden/ref_dump_input.npz      the arrays that were handed to it
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import refbind as R  # noqa: E402

REF_CFG = "/root/reference/configuration"


def stage(files):
    """A temporary directory with configuration/<name> -> the given source files (the reference's paths are hard-coded)."""
    d = tempfile.mkdtemp(prefix="sphmi_io_")
    os.makedirs(os.path.join(d, "configuration"))
    os.makedirs(os.path.join(d, "buffers"))
    for name, src in files.items():
        shutil.copy(src, os.path.join(d, "configuration", name))
    return d


def main():
    # ---- config #1 through the reference loader
    d = stage({"position.txt": os.path.join(REF_CFG, "positionPureLiquid.txt"),
               "velocity.txt": os.path.join(REF_CFG, "velocityPureLiquid.txt")})
    c1 = R.load_configuration(d)
    shutil.rmtree(d)
    np.savez_compressed(os.path.join(HERE, "config1_input.npz"), position=c1["position"], velocity=c1["velocity"])
    print("config1: N %d liquid %d boundary %d" % (c1["N"], c1["numOfLiquidP"], c1["numOfBoundaryP"]))

    # ---- the writer: a small scene (3 elastic, 5 liquid, 4 boundary particles, 2 membranes), two frames
    rng = np.random.RandomState(7)
    pos = np.zeros((12, 4), np.float32)
    pos[:, :3] = rng.uniform(0.0, 100.0, (12, 3)).astype(np.float32)
    pos[:3, 3], pos[3:8, 3], pos[8:, 3] = 2.1, 1.1, 3.1
    pos[4, 0], pos[5, 1], pos[6, 2] = 1.0e-7, 123456.789, 0.0   # formats: exponent, > 6 digits, zero
    con = np.full((3 * 32, 4), -1.0, np.float32)
    con[:, 1:] = 0.0
    con[0] = (1.1, 3.3816e-06, 1.2, 0.0)
    con[1] = (2.1, 2.0e-06, 0.0, 0.0)
    con[32] = (0.1, 3.3816e-06, 57.4, 0.0)
    mem4 = np.array([[0, 1, 2, 0], [2, 1, 0, 0]], np.int32)
    d = stage({})
    R.save_configuration(d, pos, con, mem4, 3, 5, True)
    pos2 = pos.copy()
    pos2[:8, :3] += np.float32(0.25)
    R.save_configuration(d, pos2, con, mem4, 3, 5, False)
    out = os.path.join(HERE, "ref_dump")
    os.makedirs(out, exist_ok=True)
    for f in ("position_buffer.txt", "connection_buffer.txt", "membranes_buffer.txt"):
        shutil.copy(os.path.join(d, "buffers", f), os.path.join(out, f))
    np.savez(os.path.join(HERE, "ref_dump_input.npz"), position=pos, position2=pos2, connections=con, membranes=mem4[:, :3])

    # ---- the loader: a hand-made configuration in the file order of the shipped position.txt (boundary, elastic, liquid);
    # elasticconnections.txt is the connection file the reference itself just wrote
    lines_p = ["1.67\t1.67\t1.67\t3.1", "0.835\t100.2\t8.35e+02\t3.1", "5.01000e+00\t5.010000e+00\t5.0100000e+00\t2.1",
               "6.68\t5.01\t5.01\t2.1", "8.35\t5.01\t5.01\t2.1", "1.5531e1\t2.e1\t3e1\t1.1", "17.0841\t20\t30\t1.1",
               "18.6372 20 30 1.1", "  20.1903\t20\t30\t1.1"]
    lines_v = ["0.57735\t0.57735\t0.57735\t0", "-1\t0\t0\t0", "0\t0\t0\t0", "1e-3\t-2e-3\t3e-3\t0", "0\t0\t0\t0",
               "0\t-9.8e-5\t0\t0", "0\t0\t0\t0", "0\t0\t0\t0", "0\t0\t0\t0"]
    lo = os.path.join(HERE, "ref_loader")
    os.makedirs(lo, exist_ok=True)
    open(os.path.join(lo, "position.txt"), "w").write("\n".join(lines_p) + "\n")
    open(os.path.join(lo, "velocity.txt"), "w").write("\n".join(lines_v))  # (no trailing newline: the last line may lack it)
    shutil.copy(os.path.join(out, "connection_buffer.txt"), os.path.join(lo, "elasticconnections.txt"))
    shutil.rmtree(d)
    d = stage({n: os.path.join(lo, n) for n in ("position.txt", "velocity.txt", "elasticconnections.txt")})
    L = R.load_configuration(d)
    shutil.rmtree(d)
    np.savez(os.path.join(HERE, "ref_loader.npz"), position=L["position"], velocity=L["velocity"], elastic=L["elastic"],
             counts=np.array([L["numOfLiquidP"], L["numOfElasticP"], L["numOfBoundaryP"]], np.int32))
    print("loader fixture: N %d (liquid %d elastic %d boundary %d), %d connection rows" %
          (L["N"], L["numOfLiquidP"], L["numOfElasticP"], L["numOfBoundaryP"], L["elastic"].shape[0]))


if __name__ == "__main__":
    main()
