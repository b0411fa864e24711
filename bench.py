#!/usr/bin/env python3
"""bench.py — particle-steps/s of the PCISPH step path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL. Either the driver starts the ranks (`python -m torch.distributed.run ... bench.py
--gpus N`: WORLD_SIZE is set, this file is a rank) or this file does: called from a plain shell with --gpus N > 1 and no
WORLD_SIZE it starts N fresh child processes with torch.distributed.run BEFORE touching any GPU, relays rank 0's JSON
line and exits with the children's status (a process that has initialised the GPU is never re-executed).

A "step" is one full simulationStep() (owPhysicsFluidSimulator.cpp:79-149: neighbour search + PCISPH predict-correct
loop + integration) of BASELINE config #4, the 16M-particle pure-liquid box of SURVEY §8(d) (N = 16,507,704 including
the boundary shell, box 78h x 50h x 470h, wide cell ids) — the configuration the metric and both north_star targets are
quoted on — through the C ABI of libsphmi.so with all state resident in HBM. The same box is used for every --gpus N
(N > 1: cut into N z-slabs, 4-layer halo exchanged once per step over RCCL), so the series is STRONG scaling.
Rank 0 prints ONE JSON line. Besides the contract fields it carries
  roofline      the density pass (pcisph_computeDensity): algorithmic bytes = 132 B/particle (32 distances + 1 write,
                SURVEY §8d) / its mean launch duration, measured with HIP events on the solver's stream inside real steps
                (2.18 GB per launch: 8x the Infinity Cache, i.e. an HBM figure)
  cpu_baseline  the CPU restatement (oracle/, "port") timed on this box's host cores on the same scene, few steps
  stages_ms     mean device time per stage per step (HIP events), findNeighbors reported as time
  stages_frac   per stage: algorithmic bytes (lean layout) / time / 8 TB/s
  lib           path, sha256 and sph_build_info() of the libsphmi.so that ran (a diagnostic build is refused)
  config2_1M_cube, config3_worm, evolved_dam_break   extra blocks, N = 1 only and outside `value`: the 1M cube of config #2;
                the worm scene of config #3 (elastic matter, membranes, muscles on); a 0.77 M-particle liquid column after
                1,500 steps of collapse (disordered state: the exact-walk counters of findNeighbors are reported)
  rccl_ranks, devices, halo   (N > 1) world size as the process group reports it, each rank's device, per-rank exchange figures
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "smoothed-particle-hydrodynamics_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
DENSITY_BYTES_PER_PARTICLE = 132  # 32 x 4 B distances read + 4 B density written

WORKLOADS = {
    # name: (box in h, liquid lattice, cell id mask)
    "config2_1M_cube": ((50.0, 50.0, 50.0), (100, 100, 100), 0xffff),
    "tiny": ((8.0, 8.0, 8.0), (12, 10, 12), 0xffff),
    "tiny_long": ((8.0, 8.0, 60.0), (12, 10, 110), 0xffffffff),  # 30 cell layers: the smallest box two slabs fit in (tests)
    "cube_4M": ((80.0, 80.0, 80.0), (160, 160, 160), 0xffffffff),
    "config4_16M_box": ((78.0, 50.0, 470.0), (160, 100, 1000), 0xffffffff),  # SURVEY 8(d) config #4, strong scaling
    "config5_64M_dambreak": ((240.0, 200.0, 310.0), (250, 400, 640), 0xffffffff),  # liquid column at low x
}
DAM_BREAK = ((120.0, 50.0, 50.0), (80, 90, 100), 0xffffffff)  # extra block: 0.72 M liquid particles in a column at low x

DEFAULT_WORKLOAD = "config4_16M_box"

# Algorithmic bytes per particle of every launch of a stage in one step (pure liquid, 3 predict-correct iterations), for the
# layout the kernels actually use: neighbour ids as 16-bit offsets (64 B per row + a 4-byte base), distances 128 B per row.
# sort = radix passes x (4 R hist + 8 R + 8 W scatter + 4); find_neighbors 16 + 4 in, 64 + 128 + 4 out; density 132; forces
# 64 + 4 + 128 in + own records and outputs 44, + 32 for the (v, rho) pack; predict_density 3 x (64 + 4 + 12 + 8; predicted positions are packed x, y, z); pressure_force
# (distances recomputed from the gathered positions: none read) 2 x (68 + 24 + 48 predictPositions + 16) + (68 + 24 + 200 integrate).
# SURVEY App. D's lean layout with 4-byte ids is 2.6 KB per particle and step: `whole_step_2600B` keeps that yardstick.
STAGE_ALGO_BYTES = {"hash": 20, "sort": 72,  # (sort: 24 B per radix pass, set from the solver's pass count below)
                    "sort_post": 69, "find_neighbors": 216, "density": 132, "forces": 268,
                    "predict_density": 264, "pressure_force": 596}


def weak_workload(world):
    """--workload weak: the config #2 column repeated `world` times along z (1.06M particles per GPU), wide cell ids,
    cut into `world` z-slabs with a 4-layer halo exchanged once per step (sphmi/slab.py)."""
    return (50.0, 50.0, 50.0 * world), (100, 100, 100 * world), 0xffffffff


def host_cores():
    """Threads the CPU baseline may use: the affinity mask, capped by the cgroup CPU quota and by the GPU box's
    per-GPU CPU share (16), overridable with SPHMI_CPU_THREADS."""
    if os.environ.get("SPHMI_CPU_THREADS"):
        return max(1, int(os.environ["SPHMI_CPU_THREADS"]))
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


# ----------------------------------------------------------------------------------------------- liveness
# Every rank reports progress ("beats"); a rank that is silent for SPHMI_WATCHDOG_S seconds (default 120) ends the run with a
# non-zero status instead of hanging until somebody's outer limit: inside each rank a daemon thread checks the rank's own last
# beat (covers ranks started by the driver's torch.distributed.run), and the self-launcher checks every rank's heartbeat file.
WATCHDOG_S = float(os.environ.get("SPHMI_WATCHDOG_S", "120"))
# start-up phases (the first `import torch` on a fresh box takes 1-2 minutes, an 8-rank RCCL rendezvous tens of seconds) may be
# silent for longer than a step may
STARTUP_S = float(os.environ.get("SPHMI_WATCHDOG_STARTUP_S", str(max(600.0, WATCHDOG_S))))
_last_beat = [time.monotonic(), "start", STARTUP_S]


def beat(phase, startup=False):
    limit = STARTUP_S if startup else WATCHDOG_S
    _last_beat[0], _last_beat[1], _last_beat[2] = time.monotonic(), phase, limit
    d = os.environ.get("SPHMI_HEARTBEAT_DIR")
    if d:
        try:
            with open(os.path.join(d, "rank%s" % os.environ.get("RANK", "0")), "w") as f:
                f.write("%g|%s" % (limit, phase))
        except OSError:
            pass


def start_rank_watchdog(rank):
    def run():
        while True:
            time.sleep(min(5.0, max(0.2, WATCHDOG_S / 4)))
            idle = time.monotonic() - _last_beat[0]
            if idle > _last_beat[2]:
                sys.stderr.write("bench.py watchdog: rank %d silent for %.0f s in phase '%s' — giving up\n" % (rank, idle, _last_beat[1]))
                sys.stderr.flush()
                os._exit(3)
    threading.Thread(target=run, daemon=True).start()


def self_launch(args, argv):
    """--gpus N > 1 from a plain shell: N fresh ranks under torch.distributed.run. Nothing in THIS process touches a GPU (torch
    is not even imported); the children are a new process group so that a hung run can be ended as a whole."""
    import signal
    import socket
    import subprocess
    import tempfile
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    hb = tempfile.mkdtemp(prefix="sphmi_bench_hb_")
    env = dict(os.environ, SPHMI_HEARTBEAT_DIR=hb, SPHMI_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True, text=True, bufsize=1)
    lines = []

    def pump():
        for line in proc.stdout:
            if line.startswith('{"metric"'):
                lines.append(line)
            else:  # anything else the ranks or the launcher print is not the result line
                sys.stderr.write(line)
    t = threading.Thread(target=pump, daemon=True)
    t.start()
    started = time.monotonic()
    hung = None
    while proc.poll() is None:
        time.sleep(0.5)
        now = time.time()
        for r in range(args.gpus):
            f = os.path.join(hb, "rank%d" % r)
            limit = STARTUP_S
            try:
                idle, text = now - os.path.getmtime(f), open(f).read()
                limit, phase = float(text.split("|", 1)[0]), text.split("|", 1)[1]
            except (OSError, ValueError, IndexError):
                idle, phase = time.monotonic() - started, "not started"
            if idle > limit + 10.0:  # (the rank's own watchdog fires first; this one covers a rank that cannot even do that)
                hung = (r, idle, phase)
        if hung:
            sys.stderr.write("bench.py: rank %d silent for %.0f s in phase '%s': ending the run\n" % hung)
            try:
                os.killpg(proc.pid, signal.SIGTERM)
                proc.wait(timeout=10)
            except Exception:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except Exception:
                    pass
            break
    rc = proc.wait()
    t.join(timeout=5)
    for f in os.listdir(hb):
        os.unlink(os.path.join(hb, f))
    os.rmdir(hb)
    if hung:
        return 3
    if rc == 0 and len(lines) != 1:
        sys.stderr.write("bench.py: expected one result line from rank 0, got %d\n" % len(lines))
        return 4
    for line in lines:
        sys.stdout.write(line)
    sys.stdout.flush()
    return rc


def lib_identity():
    """Path, sha256 and build info of the libsphmi.so this process loaded."""
    import sphmi
    h = hashlib.sha256()
    with open(sphmi.LIB_PATH, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return {"path": os.path.relpath(sphmi.LIB_PATH, ROOT), "sha256": h.hexdigest(), "build": sphmi.build_info(),
            "selected_by_SPHMI_LIB": bool(os.environ.get("SPHMI_LIB"))}


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS) + ["weak"],
                    help="default: config #4 for every --gpus N (strong scaling); 'weak': 1M-particle column per GPU")
    ap.add_argument("--cpu-steps", type=int, default=-1,
                    help="steps of the CPU baseline (default: ~10 s of work on 16 cores, e.g. 4 at 16.5 M; 0 = skip)")
    ap.add_argument("--no-stage-pass", action="store_true", help="skip the second, per-stage-timed pass")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra blocks (config #2, config #3, evolved dam break)")
    ap.add_argument("--allow-diagnostic-lib", action="store_true", help="run although SPHMI_LIB names a timing-only (invalid-results) build")
    ap.add_argument("--dry-run", action="store_true",
                    help="N > 1 only: launcher, rendezvous, cuts and per-rank scene slices, no solver (needs no GPU; tests)")
    return ap.parse_args(argv)


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, argv))
    rank_main(args)


def rank_main(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d but WORLD_SIZE=%d: start it with --nproc-per-node %d (or without a launcher)" % (args.gpus, world, args.gpus))
    beat("imports", startup=True)
    if world > 1:
        start_rank_watchdog(rank)

    import numpy as np
    import torch
    import scenes
    import sphmi

    dist = None
    backend_name = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # SPHMI_DIST_BACKEND=gloo lets several ranks share one card (functional rehearsal on a 1-GPU box: host-staged halo);
        # it is also what a run with fewer cards than ranks falls back to (RCCL refuses two ranks on one device)
        ndev = torch.cuda.device_count()
        backend_name = os.environ.get("SPHMI_DIST_BACKEND") or ("nccl" if ndev >= world else "gloo")
        beat("rendezvous (%s)" % backend_name, startup=True)
        if not args.dry_run:
            local_rank = local_rank % max(1, ndev)
            torch.cuda.set_device(local_rank)
        if backend_name == "nccl":  # bind the communicator to this rank's GPU explicitly (one process per GPU)
            dist.init_process_group(backend_name, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend_name, rank=rank, world_size=world)
        dist.barrier()  # creates the communicator with every rank present before the first point-to-point exchange
        beat("rendezvous done", startup=True)
    comm_dev = "cuda" if backend_name == "nccl" else "cpu"
    if args.dry_run:
        if world == 1:
            sys.exit("--dry-run is for --gpus N > 1")
    else:
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a GPU: libsphmi has no CPU path")
        torch.cuda.set_device(local_rank)
        lib = lib_identity()
        if "DIAG" in lib["build"] and not args.allow_diagnostic_lib:
            sys.exit("bench.py: %s is a diagnostic build (%s): its results are invalid; pass --allow-diagnostic-lib to time it anyway"
                     % (lib["path"], lib["build"]))

    strong = args.workload != "weak"
    if not strong:
        box, lattice, mask = weak_workload(world)
        workload_name = "weak_%dx_config2_column" % world
    else:
        box, lattice, mask = WORKLOADS[args.workload]
        if world > 1:
            mask = 0xffffffff  # the slab decomposition needs wide cell ids
        workload_name = args.workload

    decomposition = None
    stream = None
    beat("scene", startup=True)
    if world > 1:
        # Every rank builds only ITS slice of the scene (own layers + ghost layers) from the generator, after cutting the box with
        # the per-layer histogram — no rank ever holds the 16.5 M-particle arrays (N x 528 MB of host copies otherwise).
        from sphmi import slab as S
        cfg = scenes.liquid_box_config(box, mask=mask)
        nl, nb = sphmi.box_counts(cfg, *lattice)
        N = nl + nb
        hist = sphmi.box_layer_histogram(cfg, *lattice)
        occupied = np.flatnonzero(hist)
        lo_layer = int(occupied[0])
        cuts = S.balanced_cuts_hist(hist[lo_layer:int(occupied[-1]) + 1], lo_layer, world)
        slab = S.make_slab(cuts, rank, world, N)
        own_lo = slab.layerLo - slab.ghostLayers if slab.hasLower else S.OPEN_LO
        own_hi = slab.layerHi + slab.ghostLayers if slab.hasUpper else S.OPEN_HI
        pos_l, vel_l, gid_l = sphmi.generate_box_slice(cfg, *lattice, own_lo, own_hi)
        beat("scene slice: %d particles" % gid_l.size, startup=True)
        if args.dry_run:
            return dry_run_report(args, np, torch, dist, S, cfg, slab, pos_l, gid_l, N, rank, world, workload_name, backend_name)
        cfg.device = local_rank
        stream = torch.cuda.Stream(device=local_rank)
        cfg.stream = stream.cuda_stream
        backend = S.HipSlabBackend(cfg, pos_l, vel_l, gid_l, slab)
        del pos_l, vel_l
        decomposition = S.SlabDecomposition(backend, rank, world, dist)
        solver = backend.solver
        sc = None

        class _Stepper:  # step + halo exchange
            def step(self, it):
                decomposition.step(it)
                beat("step %d" % it)
        stepper = _Stepper()
    else:
        sc = scenes.liquid_box(box, lattice, mask=mask)
        cfg = sc["cfg"]
        cfg.device = local_rank
        stream = torch.cuda.Stream(device=local_rank)
        cfg.stream = stream.cuda_stream  # the solver launches on this stream; torch events below are recorded on it
        N = cfg.particleCount  # global particle count
        solver = sphmi.owHIPSolver(cfg, sc["position"], sc["velocity"])
        stepper = solver
    beat("solver ready", startup=True)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    it = 0
    for _ in range(args.warmup):
        stepper.step(it); it += 1
    barrier()
    beat("warm-up done")
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        ev0.record(stream)
        for _ in range(args.steps):
            stepper.step(it); it += 1
        ev1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    beat("timed region done")
    dev_ms = ev0.elapsed_time(ev1)
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    ms_per_step = wall * 1e3 / args.steps
    value = N * args.steps / wall  # N = particles of the whole job (all ranks)

    # second pass: the same steps with a HIP event pair around every stage (kept out of `value`)
    stages_ms, stages_frac, roofline, sort_passes = {}, {}, None, None
    if not args.no_stage_pass:
        solver.set_stage_timing(True)
        solver.reset_stage_times()
        k2 = max(5, min(args.steps, 20))
        for _ in range(k2):
            stepper.step(it); it += 1
        if decomposition is not None:
            decomposition.finish()  # (collects the last asynchronous exchange: the local particle count)
        st = solver.stage_times()
        solver.set_stage_timing(False)
        stages_ms = {k: round(ms / k2, 5) for k, (ms, cnt) in st.items() if cnt}
        d_ms, d_cnt = st["density"]
        n_local = solver.N  # particles this rank's density kernel processes per launch (incl. ghost layers when N > 1)
        if d_cnt:
            achieved = n_local * DENSITY_BYTES_PER_PARTICLE / (d_ms / d_cnt * 1e-3) / 1e9
            roofline = {"kernel": "k_density (pcisph_computeDensity)", "bound": "hbm", "achieved": round(achieved, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": None, "avg_launch_us": round(d_ms / d_cnt * 1e3, 2),
                        "bytes_per_launch": n_local * DENSITY_BYTES_PER_PARTICLE}
            # HBM bytes from the committed rocprofv3 PMC passes of this workload (profiles/README.md) — NOT measured in this
            # run (counters cannot be read from inside the process), and only quoted while the kernel sources are the ones the
            # passes were taken on
            traffic_file = os.path.join(ROOT, "profiles", "density_traffic.json")
            if os.path.exists(traffic_file) and world == 1:
                tr = json.load(open(traffic_file))
                entry = tr if tr.get("workload") == workload_name else tr.get("other_workloads", {}).get(workload_name)
                if entry is not None:
                    current = density_sources_sha()
                    if tr.get("kernel_sources_sha256") in (None, current):
                        roofline["traffic"] = entry.get("hbm_bytes_per_launch")
                        roofline["traffic_source"] = "profiles/density_traffic.json (committed rocprofv3 PMC passes; not this run)"
                    else:
                        roofline["traffic_source"] = ("profiles/density_traffic.json is stale: measured on other kernel sources (%s..., now %s...)"
                                                      % (tr["kernel_sources_sha256"][:12], current[:12]))
        # per stage: algorithmic bytes of all its launches in one step / its time / 8 TB/s (pure-liquid scenes)
        sort_passes = solver.step_sort_passes()
        algo = dict(STAGE_ALGO_BYTES, sort=24 * sort_passes)  # 24 B per particle and radix pass (2 or 3 passes)
        # (single domain only: in slab mode the stages are launched on different layer ranges and grouped differently)
        stages_frac = {} if world > 1 else {k: round(n_local * algo[k] / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                            for k, v in stages_ms.items() if k in algo and v > 0}
        step_ms = sum(stages_ms.values())
        if step_ms > 0 and world == 1:
            stages_frac["whole_step_2600B"] = round(n_local * 2600.0 / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            moved = sum(algo[k] for k in stages_ms if k in algo)
            stages_frac["whole_step_bytes_moved"] = moved
            stages_frac["whole_step_moved"] = round(n_local * moved / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    beat("stage pass done")

    # SURVEY 8(d) extras, single GPU only and outside `value` (a box's clocks drift while it is loaded — the same plain loop is up
    # to 8 % slower a second or two into the run — so everything here is compared with a plain loop of its own, not with
    # `ms_per_step`): per-step p50 from one event pair per step, and
    # the step with the reference's mandatory 16N-byte position read-back (read_position_buffer after every step,
    # owPhysicsFluidSimulator.cpp:115) — started asynchronously so that the copy runs under the next step
    # (sph_read_position_async), timed INTERLEAVED with the same loop without it (plain, async, plain, async; 50 steps each), and
    # waited for like the reference does.
    #   ms_per_step_with_position_readback        steady state: device time per step of the async loop (events on the solver's stream,
    #                                             from before the first step to after the last step's kernels — what a caller that
    #                                             keeps stepping pays per step; the next integrate waits for the copy inside it)
    #   ..._incl_final_copy                       wall time of the same loop / 50 including the wait for the LAST copy (~6 ms at
    #                                             16.5 M that nothing can hide, i.e. +0.13 ms per step over 50 steps)
    #   ms_per_step_same_pass_without_readback    the interleaved plain loop, same method as the steady-state figure
    p50_ms, readback_ms, readback_blocking_ms, readback_plain_ms, readback_wall_ms = None, None, None, None, None
    if world == 1:
        host_pos = np.empty((N, 4), np.float32)
        solver.read_position_buffer_async(host_pos)  # (page-locks host_pos once, outside the timed loops)
        solver.wait_position_buffer()
        seg = {"plain": [], "async": []}
        wall_async = []
        for _ in range(2):
            for kind in ("plain", "async"):
                solver.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                r0 = time.perf_counter()
                with torch.cuda.stream(stream):
                    e0.record(stream)
                    for _ in range(50):
                        stepper.step(it); it += 1
                        if kind == "async":
                            solver.read_position_buffer_async(host_pos)
                    e1.record(stream)
                if kind == "async":
                    solver.wait_position_buffer()
                solver.synchronize()
                if kind == "async":
                    wall_async.append((time.perf_counter() - r0) * 1e3 / 50)
                seg[kind].append(e0.elapsed_time(e1) / 50)
        readback_ms = sum(seg["async"]) / 2
        readback_plain_ms = sum(seg["plain"]) / 2
        readback_wall_ms = sum(wall_async) / 2
        kb = 10
        r0 = time.perf_counter()
        for _ in range(kb):
            stepper.step(it); it += 1
            solver.read_position_buffer(host_pos)
        readback_blocking_ms = (time.perf_counter() - r0) * 1e3 / kb
        del host_pos
        k3 = max(5, min(args.steps, 50))
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(k3 + 1)]
        with torch.cuda.stream(stream):
            evs[0].record(stream)
            for i in range(k3):
                stepper.step(it); it += 1
                evs[i + 1].record(stream)
        torch.cuda.synchronize()
        per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(k3))
        p50_ms = per_step[len(per_step) // 2]
    beat("read-back pass done")

    # Multi-GPU sanity after the timed region (outside `value`): the ranks' owned sets must still partition the particles and
    # every owned particle must be finite — a silent halo failure would show here.
    partition_ok = None
    if decomposition is not None:
        decomposition.finish()
        pos_l, vel_l, gid_l, owned_l = solver.slab_read()
        m = owned_l.astype(bool)
        stats = torch.tensor([float(m.sum()), float(np.isfinite(pos_l[m]).all() and np.isfinite(vel_l[m]).all()),
                              float(gid_l[m].astype(np.float64).sum())], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        partition_ok = bool(int(stats[0].item()) == N and int(stats[1].item()) == world
                            and abs(stats[2].item() - N * (N - 1) / 2.0) < 0.5)  # every global id exactly once

    per_rank = None  # every rank's local particle count, bytes sent, host time spent in the exchange (waits included), device
    if decomposition is not None:
        mine = torch.tensor([float(solver.N), float(decomposition.bytes_sent), float(getattr(decomposition, "exchange_seconds", 0.0)),
                             float(torch.cuda.current_device())], dtype=torch.float64, device=comm_dev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank = np.stack([g.cpu().numpy() for g in gathered])


    # CPU baseline: the oracle (bit-identical CPU restatement) on the same scene, all host cores, rank 0 only
    cpu = None
    cpu_steps = args.cpu_steps
    if cpu_steps < 0:  # ~10 s on 16 cores at the 6.9e6 particle-steps/s of round 1: 4 steps at 16.5 M, 80 at 1 M
        cpu_steps = int(max(2, min(80, round(6.9e7 / N))))
    if rank == 0 and cpu_steps > 0 and world == 1:
        cores = host_cores()
        ora = scenes.oracle_for(sc, threads=cores)
        ora.step()  # warm-up (page faults, first-touch)
        c0 = time.perf_counter()
        for _ in range(cpu_steps):
            ora.step()
        cw = time.perf_counter() - c0
        cpu = {"value": round(N * cpu_steps / cw, 1), "unit": "particle-steps/s", "cores": cores, "kind": "port",
               "sample": "%d steps of the same %d-particle scene after 1 warm-up step (oracle/sph_oracle.c, OpenMP, %d threads)"
                         % (cpu_steps, N, cores), "ms_per_step": round(cw * 1e3 / cpu_steps, 2)}
        ora.close()
        del ora
        # the 1-thread figure SURVEY 8(d) asks for beside the all-cores one, on the <= 1 M-particle cube of config #2
        # (a 1-thread step of the 16.5 M box takes ~26 s)
        sc1 = sc if N <= 1100000 else scenes.liquid_box(*WORKLOADS["config2_1M_cube"][:2], mask=WORKLOADS["config2_1M_cube"][2])
        one = scenes.oracle_for(sc1, threads=1)
        one.step()
        c0 = time.perf_counter()
        n1 = 2
        for _ in range(n1):
            one.step()
        c1 = time.perf_counter() - c0
        one.close()
        cpu["single_thread_value"] = round(sc1["cfg"].particleCount * n1 / c1, 1)
        cpu["single_thread_sample"] = "%d steps of the %d-particle cube of config #2, 1 thread" % (n1, sc1["cfg"].particleCount)

    # extra blocks, GPU only, N = 1 only, outside `value`
    extra = {}
    if rank == 0 and world == 1 and not args.no_extra:
        solver.close()
        sc = None
        if workload_name != "config2_1M_cube":
            extra["config2_1M_cube"] = extra_config2(scenes, sphmi, torch, local_rank, stream)
        extra["config3_worm"] = extra_worm(np, sphmi, torch, local_rank, stream)
        extra["evolved_dam_break"] = extra_dam_break(np, scenes, sphmi, torch, local_rank, stream)

    if rank == 0:
        out = {
            "metric": "particle-steps/sec (whole node)", "value": round(value, 1), "unit": "particle-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload_name, "particles": N, "particles_per_gpu": N // world, "box_in_h": list(box),
                       "liquid_lattice": list(lattice), "cell_ids": "ref16" if mask == 0xffff else "wide",
                       "pcisph_iterations": cfg.maxIteration,
                       "parallelism": "1 GPU" if world == 1 else
                       "%d z-slabs, 4-layer halo, one RCCL send + recv per neighbour per step" % world},
            "device_ms_per_step": round(dev_ms / args.steps, 4),
            "ms_per_step_p50": None if p50_ms is None else round(p50_ms, 4),
            "ms_per_step_with_position_readback": None if readback_ms is None else round(readback_ms, 4),
            "ms_per_step_with_blocking_position_readback": None if readback_blocking_ms is None else round(readback_blocking_ms, 4),
            "ms_per_step_with_position_readback_incl_final_copy": None if readback_wall_ms is None else round(readback_wall_ms, 4),
            "ms_per_step_same_pass_without_readback": None if readback_plain_ms is None else round(readback_plain_ms, 4),
            "roofline": roofline, "cpu_baseline": cpu, "stages_ms": stages_ms, "stages_frac": stages_frac,
            "radix_sort_passes": sort_passes, "lib": lib,
        }
        out.update(extra)
        if cpu:
            out["gpu_over_cpu"] = round(value / cpu["value"], 1)
        if decomposition is not None:
            out["dist_backend"] = backend_name
            out["rccl_ranks"] = dist.get_world_size() if backend_name == "nccl" else 0
            out["devices"] = per_rank[:, 3].astype(int).tolist()
            out["halo"] = {"local_particles_per_rank": per_rank[:, 0].astype(int).tolist(),
                           "bytes_sent_per_step_per_rank": (per_rank[:, 1] / max(1, it)).astype(int).tolist(),
                           "exchange_host_ms_per_step_per_rank": [round(v * 1e3 / max(1, it), 4) for v in per_rank[:, 2]],
                           "p2p_groups_per_step_rank0": round(decomposition.transfers / max(1, it), 3),
                           "record_bytes": 4 * decomposition.rec, "boundary_shell": "static (kept per rank, never sent)",
                           "host_waits_per_step": 1 if decomposition._can_run_async() else 3,
                           "owned_sets_partition_all_particles": partition_ok}
        print(json.dumps(out), flush=True)
    beat("done")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def density_sources_sha():
    """Hash of the source text k_density is compiled from — its section of sph_pcisph.hip (K6 header comment up to the K9 header),
    the NbrTile accessor it reads the map through and nbr_index(): profiles/density_traffic.json is only quoted for these."""
    src = open(os.path.join(ROOT, "smoothed-particle-hydrodynamics_amd", "csrc", "sph_pcisph.hip")).read()
    common = open(os.path.join(ROOT, "smoothed-particle-hydrodynamics_amd", "csrc", "sph_common.h")).read()
    parts = [src[src.index("// ------------------------------------------------------------------ K6 pcisph_computeDensity"):
                 src.index("// ------------------------------------------------------------------ K9 pcisph_predictPositions")],
             src[src.index("struct NbrTile {"):src.index("// XCD-aware block order")],
             common[common.index("// index of (sorted particle id, slot) in the tiled neighbour map"):common.index("// The neighbour ids a second time")]]
    return hashlib.sha256("\n".join(parts).encode()).hexdigest()


def dry_run_report(args, np, torch, dist, S, cfg, slab, pos_l, gid_l, N, rank, world, workload_name, backend_name):
    """--dry-run: what every rank would hand its solver, checked across ranks — the owned sets of the slices partition the
    particles — and reported as a line without a value. Exercises launcher, rendezvous, cuts, slices, heartbeats."""
    beat("dry run")
    if os.environ.get("SPHMI_BENCH_TEST_HANG_RANK") == str(rank):  # test hook: this rank goes silent (watchdog test)
        time.sleep(3600)
    lay = S.particle_layers(pos_l, cfg)
    owned = (lay >= slab.layerLo) & (lay < slab.layerHi)
    stats = torch.tensor([float(owned.sum()), float(gid_l[owned].astype(np.float64).sum()), float(gid_l.size)], dtype=torch.float64)
    gathered = [torch.zeros_like(stats) for _ in range(world)]
    dist.all_gather(gathered, stats)
    g = np.stack([t.numpy() for t in gathered])
    ok = bool(int(g[:, 0].sum()) == N and abs(g[:, 1].sum() - N * (N - 1) / 2.0) < 0.5)
    if rank == 0:
        print(json.dumps({"metric": "particle-steps/sec (whole node)", "value": None, "unit": "particle-steps/s", "n_gpus": world,
                          "steps": 0, "warmup": 0, "dry_run": True, "dist_backend": backend_name,
                          "config": {"workload": workload_name, "particles": N},
                          "halo": {"local_particles_per_rank": g[:, 2].astype(int).tolist(),
                                   "owned_particles_per_rank": g[:, 0].astype(int).tolist(),
                                   "owned_sets_partition_all_particles": ok}}), flush=True)
    beat("done")
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        sys.exit(5)


def _timed_steps(solver, torch, first, k):
    torch.cuda.synchronize()
    e0 = time.perf_counter()
    for i in range(k):
        solver.step(first + i)
    torch.cuda.synchronize()
    return time.perf_counter() - e0


def _stage_pass(solver, first, k):
    solver.set_stage_timing(True)
    solver.reset_stage_times()
    for i in range(k):
        solver.step(first + i)
    st = solver.stage_times()
    solver.set_stage_timing(False)
    return {a: round(ms / k, 5) for a, (ms, cnt) in st.items() if cnt}


def extra_config2(scenes, sphmi, torch, device, stream):
    """BASELINE config #2 (1M cube, reference-exact 16-bit cell ids)."""
    b2, l2, m2 = WORKLOADS["config2_1M_cube"]
    sc2 = scenes.liquid_box(b2, l2, mask=m2)
    sc2["cfg"].device = device
    sc2["cfg"].stream = stream.cuda_stream
    s2 = sphmi.owHIPSolver(sc2["cfg"], sc2["position"], sc2["velocity"])
    n2 = sc2["cfg"].particleCount
    for i in range(5):
        s2.step(i)
    k = 50
    w2 = _timed_steps(s2, torch, 5, k)
    out = {"particles": n2, "cell_ids": "ref16", "steps": k, "ms_per_step": round(w2 * 1e3 / k, 4),
           "value": round(n2 * k / w2, 1), "stages_ms": _stage_pass(s2, 55, 20)}
    s2.close()
    return out


def extra_worm(np, sphmi, torch, device, stream):
    """BASELINE config #3: the reference's generated worm scene (owHelper.cpp:709-1429; elastic matter, membranes), muscles driven
    by the signal of main_sim.py after every step (owPhysicsFluidSimulator.cpp:134-141), shipped box, 16-bit cell ids."""
    cfg = sphmi.default_config()
    w = sphmi.generate_worm(cfg)
    cfg.device = device
    cfg.stream = stream.cuda_stream
    s = sphmi.owHIPSolver(cfg, w["position"], w["velocity"], w["elastic"], w["membranes"], w["particle_membranes"])
    n = cfg.particleCount

    def run(first, k):
        for i in range(first, first + k):
            s.step(i)
            s.updateMuscleActivityData(sphmi.muscle_signal(i, cfg.muscleCount))
    run(0, 5)
    torch.cuda.synchronize()
    k = 50
    e0 = time.perf_counter()
    run(5, k)
    torch.cuda.synchronize()
    wall = time.perf_counter() - e0
    s.set_stage_timing(True)
    s.reset_stage_times()
    run(55, 20)
    st = s.stage_times()
    s.set_stage_timing(False)
    pos = s.read_position_buffer()
    out = {"particles": n, "elastic": w["numOfElasticP"], "liquid": w["numOfLiquidP"], "boundary": w["numOfBoundaryP"],
           "membranes": int(cfg.numOfMembranes), "muscles": "on", "steps": k, "ms_per_step": round(wall * 1e3 / k, 4),
           "value": round(n * k / wall, 1), "stages_ms": {a: round(ms / 20, 5) for a, (ms, cnt) in st.items() if cnt},
           "finite": bool(np.isfinite(pos).all())}
    s.close()
    return out


def extra_dam_break(np, scenes, sphmi, torch, device, stream, evolve=1500):
    """A disordered state: a 0.72 M-particle liquid column (1.1 M with the shell) that has been collapsing for `evolve` steps on
    the GPU — dense cells, long candidate lists, the exact walk of findNeighbors in use (dbg[0..3] per step)."""
    b, l, m = DAM_BREAK
    sc = scenes.liquid_box(b, l, mask=m)
    sc["cfg"].device = device
    sc["cfg"].stream = stream.cuda_stream
    s = sphmi.owHIPSolver(sc["cfg"], sc["position"], sc["velocity"])
    n, nl = sc["cfg"].particleCount, sc["numOfLiquidP"]
    k = 50
    rest = _timed_steps(s, torch, 0, k)  # the resting column, for comparison
    for i in range(k, evolve):
        s.step(i)
    wall = _timed_steps(s, torch, evolve, k)
    s.set_stage_timing(True)
    s.reset_stage_times()  # (also zeroes the fallback counters)
    for i in range(20):
        s.step(evolve + k + i)
    st = s.stage_times()
    s.set_stage_timing(False)
    c = s.buffer("debugCounters")
    pos = s.read_position_buffer()
    out = {"particles": n, "liquid": nl, "evolved_steps": evolve, "steps": k, "ms_per_step": round(wall * 1e3 / k, 4),
           "value": round(n * k / wall, 1), "ms_per_step_at_rest": round(rest * 1e3 / k, 4),
           "stages_ms": {a: round(ms / 20, 5) for a, (ms, cnt) in st.items() if cnt},
           "find_neighbors_exact_walks_per_step": {"cell_not_staged": int(c[0]) // 20, "list_overflow": int(c[1]) // 20,
                                                   "rows_without_16bit_ids": int(c[2]) // 20, "runs_dropped": int(c[3]) // 20},
           # (wave, batch of 8 neighbours) pairs of k_pressure_force that left the short division / square-root path, of 3 x 4 per wave
           "pressure_force_slow_batches_per_step": int(c[8]) // 20, "pressure_force_batches_per_step": 12 * ((n + 63) // 64),
           "mean_y_of_liquid_in_h": round(float(pos[:nl, 1].mean() / sc["cfg"].h), 2),
           "front_x_in_h": round(float(pos[:nl, 0].max() / sc["cfg"].h), 2), "finite": bool(np.isfinite(pos).all())}
    s.close()
    return out


if __name__ == "__main__":
    main()
