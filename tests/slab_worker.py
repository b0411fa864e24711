"""Worker for tests/test_slab.py: one rank of a 2-rank (or N-rank) slab-decomposed run.

  python slab_worker.py --rank R --world W --port P --backend oracle|hip --steps K --out DIR

backend "oracle" is a TEST-ONLY stand-in that advances the local particle set with oracle/sph_oracle.c (so the exchange
logic in sphmi/slab.py runs on CPU under gloo); backend "hip" is the product path (libsphmi.so, HBM-resident messages).
Each rank writes its OWNED particles (global id, position, velocity) to DIR/rank<R>.npz.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import scenes  # noqa: E402
import sphmi  # noqa: E402
from sphmi import slab as S  # noqa: E402

REC = sphmi.SLAB_RECORD_WORDS
BOUNDARY = sphmi.BOUNDARY_PARTICLE


def scene():
    sc = _scene()
    if os.environ.get("SPHMI_TEST_MAXITER"):  # other predict-correct iteration counts change every stage's ghost depth
        sc["cfg"].maxIteration = int(os.environ["SPHMI_TEST_MAXITER"])
    if os.environ.get("SPHMI_TEST_FUZZ_SEED"):  # every liquid particle its own velocity, up to 0.37 cell layers per step along z
        rng = np.random.default_rng(int(os.environ["SPHMI_TEST_FUZZ_SEED"]) + 7)
        nl = sc["numOfLiquidP"]
        sc["velocity"][:nl, :3] = rng.uniform(-1.0, 1.0, size=(nl, 3)).astype(np.float32)
    if os.environ.get("SPHMI_TEST_ODD_W"):  # one liquid particle with velocity.w != 0: compact 7-word records would lose it
        sc["velocity"][5, 3] = np.float32(0.25)
    if os.environ.get("SPHMI_TEST_VZ"):  # the liquid drifts along z (across the cuts): ownership must change hands during the run
        nl = sc["numOfLiquidP"]
        sc["velocity"][:nl, 2] = np.float32(os.environ["SPHMI_TEST_VZ"])
    return sc


def _scene():
    # wide-mode box, long in z: 30 cell layers, lattice with a little jitter so that particles cross the cut
    if os.environ.get("SPHMI_TEST_FUZZ_SEED"):  # tools/fuzz_slab.py: a random long box (40-100 cell layers), random fill and jitter
        rng = np.random.default_rng(int(os.environ["SPHMI_TEST_FUZZ_SEED"]))
        box = (float(rng.integers(6, 11)), float(rng.integers(6, 11)), float(2 * rng.integers(40, 101)))
        lattice = tuple(int(max(2, (2.0 * b - 7.0) / 0.93 * f)) for b, f in zip(box, rng.uniform(0.6, 1.0, size=3)))
        return scenes.liquid_box(box, lattice, mask=0xffffffff, jitter_in_r0=float(rng.uniform(0.0, 0.3)), seed=int(rng.integers(1, 1 << 30)))
    if os.environ.get("SPHMI_TEST_EIGHT_SLABS"):  # 72 layers, thin in x and y: eight slabs of nine layers
        return scenes.liquid_box((6.0, 6.0, 144.0), (8, 8, 290), mask=0xffffffff, jitter_in_r0=0.05)
    if os.environ.get("SPHMI_TEST_LONG_SCENE"):  # 42 layers: three slabs of 14, so the middle one has an interior between its two cut zones
        return scenes.liquid_box((8.0, 8.0, 84.0), (12, 10, 156), mask=0xffffffff, jitter_in_r0=0.05)
    return scenes.liquid_box((8.0, 8.0, 60.0), (12, 10, 110), mask=0xffffffff, jitter_in_r0=0.05)


class OracleSlabBackend:
    """numpy + oracle implementation of the backend interface of sphmi/slab.py (test infrastructure)."""

    def __init__(self, cfg, position, velocity, global_ids, slab):
        import torch
        self.torch = torch
        self.cfg, self.slab = cfg, slab
        self.pos, self.vel, self.gid = position.copy(), velocity.copy(), global_ids.copy()
        self.owned = self._owned(self.pos)
        if os.environ.get("SPHMI_TEST_FRAMED") or os.environ.get("SPHMI_TEST_ASYNC"):
            self.frame_device = torch.device("cpu")
        if os.environ.get("SPHMI_TEST_ASYNC"):  # the four calls of the asynchronous exchange (sphmi/slab.py), in numpy
            self.step_and_pack_framed = self._step_and_pack_framed
            self.rebuild_framed = self._rebuild_framed
            self.rebuild_finish = self._rebuild_finish
            self.wait_for = lambda stream: None

    def _owned(self, pos):
        lay = S.particle_layers(pos, self.cfg)
        return (lay >= self.slab.layerLo) & (lay < self.slab.layerHi)

    @property
    def count(self):
        return self.pos.shape[0]

    def step(self, iteration):
        from oracle import oraclebind as O
        d = sphmi.config_dict(self.cfg)
        d["N"] = d["particleCount"] = self.count
        o = O.OracleSolver(d, self.pos, self.vel, threads=2)
        o.step()
        self.pos = o.buffer("position").reshape(-1, 4)[:self.count].copy()
        self.vel = o.buffer("velocity").reshape(-1, 4)[:self.count].copy()
        o.close()

    # ---- the record formats of include/sphmi.h: 9 words, or 7 (x, y, z, vx, vy, vz, gid) with the type word as a constant
    record_words = REC

    def liquid_signature(self):
        nb = self.pos[:, 3].astype(np.int32) != BOUNDARY
        if not nb.any():
            return 0
        t = np.unique(self.pos[nb, 3].view(np.uint32))
        if t.size != 1 or np.any(self.vel[nb, 3].view(np.uint32) != 0):
            return 0xffffffff
        return int(t[0])

    def set_record_format(self, words, type_bits=0):
        self.record_words, self.type_bits = words, np.uint32(type_bits)

    def _records(self, m):
        p, v, g = self.pos[m].view(np.int32), self.vel[m].view(np.int32), self.gid[m].view(np.int32)[:, None]
        rec = np.concatenate([p, v, g], axis=1) if self.record_words == REC else np.concatenate([p[:, :3], v[:, :3], g], axis=1)
        return self.torch.from_numpy(np.ascontiguousarray(rec).reshape(-1))

    def pack(self):
        lay = S.particle_layers(self.pos, self.cfg)
        a = self.owned
        liquid = self.pos[:, 3].astype(np.int32) != BOUNDARY   # the boundary shell is static: kept by every rank, never sent
        down = a & liquid & (lay < self.slab.layerLo + self.slab.ghostLayers) & bool(self.slab.hasLower)
        up = a & liquid & (lay >= self.slab.layerHi - self.slab.ghostLayers) & bool(self.slab.hasUpper)
        keep = a | ~liquid
        self._kept = (self.pos[keep], self.vel[keep], self.gid[keep])
        return int(keep.sum()), self._records(down), self._records(up)

    def pack_framed(self):
        """[payload word count | payload] frames on the CPU, like HipSlabBackend's in HBM (exercises the zero-copy path)."""
        kept, down, up = self.pack()
        frames = []
        for t in (down, up):
            f = self.torch.zeros(1 + 2 * t.numel() + 8192 * REC, dtype=self.torch.int32)  # room for the agreed padding
            f[0] = t.numel()
            f[1:1 + t.numel()] = t
            frames.append(f)
        return kept, frames[0], int(down.numel()), frames[1], int(up.numel())

    def _step_and_pack_framed(self, iteration):
        self.step(iteration)
        kept, fd, nd, fu, nu = self.pack_framed()
        return None, fd, nd, fu, nu

    def _rebuild_framed(self, frame_down, frame_up):
        """What sph_slab_rebuild_framed does: lengths from word 0 of the frames; nothing is merged if a frame is too short."""
        W = self.record_words
        counts, payloads, ok = [], [], True
        for f in (frame_down, frame_up):
            n = 0 if f is None else int(f[0]) // W
            cap = 0 if f is None else (f.numel() - 1) // W
            ok = ok and n <= cap
            counts.append(n)
            payloads.append(None if f is None else f[1:1 + min(n, cap) * W])
        self._framed = (int(self._kept[0].shape[0]), counts[0], counts[1], not ok)
        if ok:
            self.rebuild(payloads[0], payloads[1])

    def _rebuild_finish(self):
        return self._framed

    def rebuild(self, recv_down, recv_up):
        parts = [self._kept]
        W = self.record_words
        for t in (recv_down, recv_up):
            if t is not None and t.numel():
                r = t.cpu().numpy().reshape(-1, W)
                if W == REC:
                    parts.append((r[:, 0:4].view(np.float32), r[:, 4:8].view(np.float32), r[:, 8].view(np.uint32)))
                else:
                    n = r.shape[0]
                    p = np.concatenate([r[:, 0:3], np.full((n, 1), self.type_bits, np.uint32).view(np.int32)], axis=1)
                    v = np.concatenate([r[:, 3:6], np.zeros((n, 1), np.int32)], axis=1)
                    parts.append((np.ascontiguousarray(p).view(np.float32), np.ascontiguousarray(v).view(np.float32), r[:, 6].view(np.uint32)))
        pos = np.concatenate([p[0] for p in parts]); vel = np.concatenate([p[1] for p in parts]); gid = np.concatenate([p[2] for p in parts])
        order = np.argsort(gid, kind="stable")
        assert np.unique(gid).size == gid.size, "duplicate particle after the halo exchange"
        self.pos, self.vel, self.gid = np.ascontiguousarray(pos[order]), np.ascontiguousarray(vel[order]), gid[order]
        self.owned = self._owned(self.pos)
        return self.count

    def owned_state(self):
        return self.gid[self.owned], self.pos[self.owned], self.vel[self.owned]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--backend", choices=["oracle", "hip"], required=True)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % a.port, rank=a.rank, world_size=a.world)
    sc = scene()
    cfg = sc["cfg"]
    n_global = cfg.particleCount
    layers = S.particle_layers(sc["position"], cfg)
    cuts = S.balanced_cuts(layers, a.world)
    slab = S.make_slab(cuts, a.rank, a.world, n_global)
    idx = S.local_indices(layers, slab)
    pos, vel = sc["position"][idx], sc["velocity"][idx]
    if a.backend == "hip":
        cfg.device = 0
        backend = S.HipSlabBackend(cfg, pos, vel, idx, slab)
    else:
        backend = OracleSlabBackend(cfg, pos, vel, idx, slab)
    if os.environ.get("SPHMI_TEST_BOUND_WORDS"):  # force the "payload outgrew its bound" path: tiny agreed bounds
        words = int(os.environ["SPHMI_TEST_BOUND_WORDS"])
        S.SlabDecomposition.next_bound = staticmethod(lambda n: words)
    dd = S.SlabDecomposition(backend, a.rank, a.world, dist, comm_device="cpu")  # gloo: messages staged through host
    counts = []
    for it in range(a.steps):
        n = dd.step(it)
        counts.append(-1 if n is None else n)  # (asynchronous exchange: the count arrives with finish())
    counts[-1] = dd.finish()
    gid, p, v = backend.owned_state()
    first_owner = np.searchsorted(np.array(cuts[1:-1]), layers, side="right")  # rank that owned each particle at the start
    np.savez(os.path.join(a.out, "rank%d.npz" % a.rank), gid=gid, pos=p, vel=v, counts=np.array(counts),
             cuts=np.array(cuts), sent=dd.bytes_sent, transfers=dd.transfers, record_words=dd.rec, asynchronous=dd._can_run_async(),
             adopted=int((first_owner[gid.astype(np.int64)] != a.rank).sum()))  # owned now, owned by another rank at the start
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
