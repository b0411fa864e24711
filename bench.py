#!/usr/bin/env python3
"""bench.py — particle-steps/s of the PCISPH step path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

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
  config2_1M_cube   the 1M-particle cube of config #2 as an extra block (N = 1 only): value, ms/step, stages
"""
import argparse
import json
import os
import sys
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
    "cube_4M": ((80.0, 80.0, 80.0), (160, 160, 160), 0xffffffff),
    "config4_16M_box": ((78.0, 50.0, 470.0), (160, 100, 1000), 0xffffffff),  # SURVEY 8(d) config #4, strong scaling
    "config5_64M_dambreak": ((240.0, 200.0, 310.0), (250, 400, 640), 0xffffffff),  # liquid column at low x
}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS) + ["weak"],
                    help="default: config #4 for every --gpus N (strong scaling); 'weak': 1M-particle column per GPU")
    ap.add_argument("--cpu-steps", type=int, default=-1,
                    help="steps of the CPU baseline (default: ~10 s of work on 16 cores, e.g. 4 at 16.5 M; 0 = skip)")
    ap.add_argument("--no-stage-pass", action="store_true", help="skip the second, per-stage-timed pass")
    ap.add_argument("--no-extra", action="store_true", help="skip the config #2 extra block")
    args = ap.parse_args()

    import numpy as np
    import torch
    import scenes
    import sphmi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # SPHMI_DIST_BACKEND=gloo lets several ranks share one card (functional rehearsal on a 1-GPU box: host-staged halo)
        backend_name = os.environ.get("SPHMI_DIST_BACKEND", "nccl")
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend_name == "nccl":  # bind the communicator to this rank's GPU explicitly (one process per GPU)
            dist.init_process_group(backend_name, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend_name, rank=rank, world_size=world)
        dist.barrier()  # creates the communicator with every rank present before the first point-to-point exchange
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: libsphmi has no CPU path")
    torch.cuda.set_device(local_rank)

    strong = args.workload != "weak"
    if not strong:
        box, lattice, mask = weak_workload(world)
        workload_name = "weak_%dx_config2_column" % world
    else:
        box, lattice, mask = WORKLOADS[args.workload]
        if world > 1:
            mask = 0xffffffff  # the slab decomposition needs wide cell ids
        workload_name = args.workload
    sc = scenes.liquid_box(box, lattice, mask=mask)
    cfg = sc["cfg"]
    cfg.device = local_rank
    stream = torch.cuda.Stream(device=local_rank)
    cfg.stream = stream.cuda_stream  # the solver launches on this stream; torch events below are recorded on it
    N = cfg.particleCount  # global particle count
    decomposition = None
    if world > 1:
        from sphmi import slab as S
        layers = S.particle_layers(sc["position"], cfg)
        cuts = S.balanced_cuts(layers, world)
        slab = S.make_slab(cuts, rank, world, N)
        idx = S.local_indices(layers, slab)
        backend = S.HipSlabBackend(cfg, sc["position"][idx], sc["velocity"][idx], idx, slab)
        decomposition = S.SlabDecomposition(backend, rank, world, dist)
        solver = backend.solver

        class _Stepper:  # step + halo exchange
            def step(self, it):
                decomposition.step(it)
        stepper = _Stepper()
    else:
        solver = sphmi.owHIPSolver(cfg, sc["position"], sc["velocity"])
        stepper = solver

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    it = 0
    for _ in range(args.warmup):
        stepper.step(it); it += 1
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        ev0.record(stream)
        for _ in range(args.steps):
            stepper.step(it); it += 1
        ev1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
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
            # run: counters cannot be read from inside the process
            traffic_file = os.path.join(ROOT, "profiles", "density_traffic.json")
            if os.path.exists(traffic_file) and world == 1:
                tr = json.load(open(traffic_file))
                if tr.get("workload") == workload_name:
                    roofline["traffic"] = tr.get("hbm_bytes_per_launch")
                elif workload_name in tr.get("other_workloads", {}):
                    roofline["traffic"] = tr["other_workloads"][workload_name]["hbm_bytes_per_launch"]
                if roofline["traffic"] is not None:
                    roofline["traffic_source"] = "profiles/density_traffic.json (committed rocprofv3 PMC passes; not this run)"
        # per stage: algorithmic bytes of all its launches in one step / its time / 8 TB/s (pure-liquid scenes)
        sort_passes = solver.step_sort_passes()
        algo = dict(STAGE_ALGO_BYTES, sort=24 * sort_passes)  # 24 B per particle and radix pass (2 or 3 passes)
        stages_frac = {k: round(n_local * algo[k] / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                       for k, v in stages_ms.items() if k in algo and v > 0}
        step_ms = sum(stages_ms.values())
        if step_ms > 0:
            stages_frac["whole_step_2600B"] = round(n_local * 2600.0 / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            moved = sum(algo[k] for k in stages_ms if k in algo)
            stages_frac["whole_step_bytes_moved"] = moved
            stages_frac["whole_step_moved"] = round(n_local * moved / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)

    # Multi-GPU sanity after the timed region (outside `value`): the ranks' owned sets must still partition the particles and
    # every owned particle must be finite — a silent halo failure would show here.
    partition_ok = None
    if decomposition is not None:
        pos_l, vel_l, gid_l, owned_l = solver.slab_read()
        m = owned_l.astype(bool)
        stats = torch.tensor([float(m.sum()), float(np.isfinite(pos_l[m]).all() and np.isfinite(vel_l[m]).all()),
                              float(gid_l[m].astype(np.float64).sum())], dtype=torch.float64,
                             device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        partition_ok = bool(int(stats[0].item()) == N and int(stats[1].item()) == world
                            and abs(stats[2].item() - N * (N - 1) / 2.0) < 0.5)  # every global id exactly once

    per_rank = None  # every rank's local particle count, bytes sent and host time spent in the exchange (waits included)
    if decomposition is not None:
        mine = torch.tensor([float(solver.N), float(decomposition.bytes_sent), float(getattr(decomposition, "exchange_seconds", 0.0))],
                            dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank = np.stack([g.cpu().numpy() for g in gathered])

    # SURVEY 8(d) extras, single GPU only and outside `value`: per-step p50 from one event pair per step, and the step
    # with the reference's mandatory 16N-byte position read-back (read_position_buffer after every step).
    p50_ms, readback_ms = None, None
    if world == 1:
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
        host_pos = np.empty((N, 4), np.float32)
        solver.synchronize()
        r0 = time.perf_counter()
        for _ in range(k3):
            stepper.step(it); it += 1
            solver.read_position_buffer(host_pos)
        readback_ms = (time.perf_counter() - r0) * 1e3 / k3

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

    # extra block: BASELINE config #2 (1M cube, reference-exact 16-bit cell ids), GPU only, outside `value`
    extra = None
    if rank == 0 and world == 1 and not args.no_extra and workload_name != "config2_1M_cube":
        solver.close()
        b2, l2, m2 = WORKLOADS["config2_1M_cube"]
        sc2 = scenes.liquid_box(b2, l2, mask=m2)
        sc2["cfg"].device = local_rank
        sc2["cfg"].stream = stream.cuda_stream
        s2 = sphmi.owHIPSolver(sc2["cfg"], sc2["position"], sc2["velocity"])
        n2 = sc2["cfg"].particleCount
        for i in range(5):
            s2.step(i)
        torch.cuda.synchronize()
        k = 50
        e0 = time.perf_counter()
        for i in range(k):
            s2.step(5 + i)
        torch.cuda.synchronize()
        w2 = time.perf_counter() - e0
        s2.set_stage_timing(True)
        s2.reset_stage_times()
        for i in range(20):
            s2.step(55 + i)
        st2 = s2.stage_times()
        s2.set_stage_timing(False)
        extra = {"particles": n2, "cell_ids": "ref16", "steps": k, "ms_per_step": round(w2 * 1e3 / k, 4),
                 "value": round(n2 * k / w2, 1), "stages_ms": {a: round(ms / 20, 5) for a, (ms, cnt) in st2.items() if cnt}}
        s2.close()

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
            "roofline": roofline, "cpu_baseline": cpu, "stages_ms": stages_ms, "stages_frac": stages_frac,
            "radix_sort_passes": sort_passes,
        }
        if extra:
            out["config2_1M_cube"] = extra
        if cpu:
            out["gpu_over_cpu"] = round(value / cpu["value"], 1)
        if decomposition is not None:
            out["halo"] = {"local_particles_per_rank": per_rank[:, 0].astype(int).tolist(),
                           "bytes_sent_per_step_per_rank": (per_rank[:, 1] / max(1, it)).astype(int).tolist(),
                           "exchange_host_ms_per_step_per_rank": [round(v * 1e3 / max(1, it), 4) for v in per_rank[:, 2]],
                           "p2p_groups_per_step_rank0": round(decomposition.transfers / max(1, it), 3),
                           "owned_sets_partition_all_particles": partition_ok}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
