"""bench.py --gpus N from a plain shell: the self-launcher (fresh ranks under torch.distributed.run before any GPU call),
the relayed result line, the per-rank scene slices and the watchdog. CPU-only here (gloo, --dry-run: no solver is built —
libsphmi has no CPU path); the GPU suite runs the same launcher with real solvers (test_slab.py)."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra_args, env_extra, timeout=240):
    env = dict(os.environ, SPHMI_DIST_BACKEND="gloo", **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    return p, time.time() - t0


def test_self_launch_two_ranks_dry_run():
    p, _ = run_bench(["--gpus", "2", "--workload", "tiny_long", "--dry-run"], {})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout  # exactly one result line reaches stdout; everything else went to stderr
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["dist_backend"] == "gloo"
    halo = out["halo"]
    assert halo["owned_sets_partition_all_particles"] is True
    assert sum(halo["owned_particles_per_rank"]) == out["config"]["particles"]
    # every rank holds its own layers plus the ghost layers only, never the whole scene
    assert all(n < out["config"]["particles"] for n in halo["local_particles_per_rank"])
    assert all(l > o for l, o in zip(halo["local_particles_per_rank"], halo["owned_particles_per_rank"]))


def test_a_failing_rank_is_relayed_as_a_failure():
    p, _ = run_bench(["--gpus", "4", "--workload", "tiny_long", "--dry-run"], {})
    assert p.returncode != 0  # 30 occupied layers are too few for four slabs of 8: the ranks say so and the launcher relays the failure
    assert "too few" in p.stderr


def test_silent_rank_ends_the_run_with_an_error():
    p, took = run_bench(["--gpus", "2", "--workload", "tiny_long", "--dry-run"],
                        {"SPHMI_BENCH_TEST_HANG_RANK": "1", "SPHMI_WATCHDOG_S": "4"}, timeout=120)
    assert p.returncode != 0
    assert "silent for" in p.stderr
    assert took < 90
    assert not [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]


def test_slices_equal_the_rows_of_the_full_scene():
    """sphmi_generate_box_slice / sphmi_box_layer_histogram against the full generator (jittered lattice: the PRNG stream must
    advance for the particles a slice skips)."""
    import scenes
    import sphmi
    from sphmi import slab as S
    sc = scenes.liquid_box((8.0, 8.0, 40.0), (12, 10, 60), mask=0xffffffff, jitter_in_r0=0.05)
    cfg = sc["cfg"]
    r0 = np.float32(cfg.r0)
    kw = dict(spacing=np.float32(0.93) * r0, origin=(np.float32(3) * r0,) * 3, jitter=float(np.float32(0.05) * r0))
    lay = S.particle_layers(sc["position"], cfg)
    assert np.array_equal(np.bincount(lay, minlength=cfg.gridCellsZ), sphmi.box_layer_histogram(cfg, 12, 10, 60, **kw))
    for lo, hi in ((S.OPEN_LO, 5), (3, 9), (8, S.OPEN_HI)):
        pos, vel, gid = sphmi.generate_box_slice(cfg, 12, 10, 60, lo, hi, **kw)
        idx = np.flatnonzero((lay >= lo) & (lay < hi))
        assert np.array_equal(gid, idx)
        assert scenes.bits_equal(pos, sc["position"][idx]) and scenes.bits_equal(vel, sc["velocity"][idx])
    assert sphmi.box_counts(cfg, 12, 10, 60) == (sc["numOfLiquidP"], sc["numOfBoundaryP"])
