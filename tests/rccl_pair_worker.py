"""GPU test worker: two slabs of one scene in ONE process on one GPU, halo frames moved by RCCL.

A 1-GPU box cannot run two RCCL ranks (one rank per device), but a world-size-1 NCCL group can send to itself. Each slab runs
the unmodified SlabDecomposition (sphmi/slab.py) in its own thread with a tiny stand-in for torch.distributed that pairs the
two sides' point-to-point operations and issues them as ONE real `batch_isend_irecv` of self sends / receives on the NCCL
group. What this exercises on real hardware: HipSlabBackend's device-written frames sent in place, RCCL's stream versus the
solvers' streams, the rebuild from RCCL-written buffers. What it cannot: xGMI links and inter-process rendezvous.
Writes both slabs' owned particles to --out.
"""
import argparse
import os
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import scenes  # noqa: E402,F401
import slab_worker  # noqa: E402
import sphmi  # noqa: E402
from sphmi import slab as S  # noqa: E402


class PairedDist:
    """The subset of torch.distributed that SlabDecomposition uses, for two in-process 'ranks' over a world-1 NCCL group."""

    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    isend, irecv = "isend", "irecv"

    class _Done:
        def wait(self):
            return True

    def __init__(self, real):
        self.real = real
        self.local = threading.local()
        self.barrier = threading.Barrier(2)
        self.pending = {}
        self.groups = 0

    def get_backend(self):
        return "nccl"

    def batch_isend_irecv(self, ops):
        me = self.local.rank
        self.pending[me] = ops
        if self.barrier.wait() == 0:  # one of the two threads issues the combined group
            real, realops = self.real, []
            for r in (0, 1):
                for snd in (o for o in self.pending[r] if o.op == self.isend):
                    rcv = [o for o in self.pending[snd.peer] if o.op == self.irecv and o.peer == r]
                    assert len(rcv) == 1 and rcv[0].tensor.numel() == snd.tensor.numel(), "unmatched halo transfer"
                    realops += [real.P2POp(real.isend, snd.tensor, 0), real.P2POp(real.irecv, rcv[0].tensor, 0)]
            for w in real.batch_isend_irecv(realops):
                w.wait()
            self.groups += 1
        self.barrier.wait()
        return [self._Done() for _ in ops]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % a.port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    paired = PairedDist(dist)
    if os.environ.get("SPHMI_TEST_BOUND_WORDS"):  # force the "payload outgrew its frame" path: tiny agreed bounds
        words = int(os.environ["SPHMI_TEST_BOUND_WORDS"])
        S.SlabDecomposition.next_bound = staticmethod(lambda n: words)
    sc = slab_worker.scene()
    n_global = sc["cfg"].particleCount
    layers = S.particle_layers(sc["position"], sc["cfg"])
    cuts = S.balanced_cuts(layers, 2)
    decs = []
    for rank in (0, 1):
        cfg = slab_worker.scene()["cfg"]  # a config object of its own per solver
        cfg.device = 0
        slab = S.make_slab(cuts, rank, 2, n_global)
        idx = S.local_indices(layers, slab)
        backend = S.HipSlabBackend(cfg, sc["position"][idx], sc["velocity"][idx], idx, slab)
        decs.append(S.SlabDecomposition(backend, rank, 2, paired, record_format=None))
    # (the in-process stand-in has no all_gather: agree on the record format by hand, as SlabDecomposition does across ranks)
    sigs = {d.backend.liquid_signature() for d in decs} - {0}
    if len(sigs) == 1 and 0xffffffff not in sigs and not os.environ.get("SPHMI_SLAB_FULL_RECORDS"):
        for d in decs:
            d.backend.set_record_format(sphmi.SLAB_COMPACT_WORDS, next(iter(sigs)))
    assert all(d.comm_device.type == "cuda" for d in decs)
    errors = []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            paired.local.rank = rank
            for it in range(a.steps):
                decs[rank].step(it)
            decs[rank].finish()
        except BaseException as e:  # surface worker failures in the parent instead of hanging the barrier
            errors.append(e)
            paired.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    for rank in (0, 1):
        gid, p, v = decs[rank].backend.owned_state()
        np.savez(os.path.join(a.out, "rank%d.npz" % rank), gid=gid, pos=p, vel=v, counts=np.array([decs[rank].backend.count]),
                 cuts=np.array(cuts), sent=decs[rank].bytes_sent, transfers=decs[rank].transfers, record_words=decs[rank].rec,
                 asynchronous=decs[rank]._can_run_async())
    print("rccl groups issued:", paired.groups)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
