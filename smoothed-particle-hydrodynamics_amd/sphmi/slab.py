"""Slab decomposition over torch.distributed (RCCL on MI355X; gloo for CPU tests) — SURVEY.md §8e, DESIGN.md §5.

One process per GPU. The global box is cut into z-slabs of whole cell layers; each rank advances its own layers plus
`GHOST_LAYERS` ghost layers per side with the ordinary fused step and, once per step, sends the authoritative particles
that now lie within `GHOST_LAYERS` of a cut to the neighbouring rank (one point-to-point message per neighbour per step — a
count word, the payload and a little padding up to a length both ends derived from the previous step — each riding one
xGMI link; no collective on the data path). Ghost zones are replaced wholesale by what arrives.

Why 4 ghost layers: one PCISPH step propagates information over 6 neighbour hops (density 1, forces 2, and +1 per
predict/correct half-iteration up to 6 for the last pressure force), each hop <= 31h/30, i.e. 6.2h < 4 cell layers (8h).
Particles owned at the last rebuild therefore see exactly the inputs a single-GPU run would give them, and because the
local arrays are kept sorted by global id the within-cell order is the global one: owned results are bit-identical to
the single-solver run (tests/test_slab.py).

Boundary particles never move, so every rank keeps the boundary particles of its ghost layers from the start and the
messages carry the other particles only; when all ranks find one common type word and velocity.w == 0 on those (always the
case for generated scenes) a record shrinks from 9 to 7 words (28 B: x, y, z, vx, vy, vz, global id).

The numerical work sits behind a tiny backend interface so that the exchange logic itself can be tested on CPU:
  backend.count                         current local particle count
  backend.record_words                  words per message record (9, or 7 once set_record_format(7, type_bits) was called)
  backend.liquid_signature()            common type word of the local non-boundary particles (0 none, 0xffffffff not uniform)
  backend.step(iteration)               advance every local particle one step
  backend.pack() -> (kept, msg_down, msg_up)   int32 tensors of record_words words per record
  backend.rebuild(recv_down, recv_up)   new local set = kept (owned particles + the static boundary ghosts) + received,
                                        sorted by global id

Under RCCL with the HIP backend a step costs the host ONE wait (for the packed frames): the frames are sent in place, the solver's
stream waits for the receive through an event, the rebuild kernels take the received counts from the frames on the device
(sph_slab_rebuild_framed), and the new particle count is collected at the start of the next step.
"""
import ctypes as C

import numpy as np

from . import SLAB_COMPACT_WORDS, SLAB_RECORD_WORDS, SphSlab, owHIPSolver

def os_environ_flag(name):
    import os
    return os.environ.get(name, "0") not in ("", "0")


GHOST_LAYERS = 4
OPEN_LO, OPEN_HI = -(1 << 30), (1 << 30)  # the first / last slab own everything below / above


def particle_layers(position, cfg):
    """z cell coordinate exactly as hashParticles computes it (sphFluid.cl:199): (int)(z * hashGridCellSizeInv), f32."""
    z = np.ascontiguousarray(position[:, 2], np.float32)
    return (z * np.float32(cfg.hashGridCellSizeInv)).astype(np.int32)


GHOST_COST = 0.75  # what a ghost particle costs relative to an owned one (measured: 8 ghost layers of 25 add ~24 % to the step)


def balanced_cuts(layers, world, min_thickness=2 * GHOST_LAYERS, ghost_cost=GHOST_COST):
    """Cut the occupied layer range into `world` slabs of (nearly) equal COST = owned particles + ghost_cost x ghost particles
    (the GHOST_LAYERS layers beyond each cut that the rank also advances): the two end slabs have one ghost zone instead of
    two and get correspondingly more layers. Returns world+1 cut layers; every slab is at least 2*GHOST_LAYERS thick so that
    halos only ever involve direct neighbours. Deterministic: every rank computes the same cuts from the same histogram."""
    lo, hi = int(layers.min()), int(layers.max()) + 1
    hist = np.bincount(layers - lo, minlength=hi - lo).astype(np.int64)
    return balanced_cuts_hist(hist, lo, world, min_thickness, ghost_cost)


def balanced_cuts_hist(hist, lo, world, min_thickness=2 * GHOST_LAYERS, ghost_cost=GHOST_COST):
    """balanced_cuts from the per-layer particle histogram (layer `lo` first) — what the ranks all-gather in a real run."""
    hist = np.asarray(hist, np.int64)
    hi = lo + hist.size
    cum = np.concatenate([[0], np.cumsum(hist)])

    def count(a, b):  # particles in layers [a, b), clipped to the occupied range
        a, b = min(max(a, lo), hi), min(max(b, lo), hi)
        return int(cum[b - lo] - cum[a - lo]) if b > a else 0

    def cost(cuts, r):
        a, b = cuts[r], cuts[r + 1]
        ghosts = (count(a - GHOST_LAYERS, a) if r > 0 else 0) + (count(b, b + GHOST_LAYERS) if r < world - 1 else 0)
        return count(a, b) + ghost_cost * ghosts

    def sweep(limit):
        """Greedy left-to-right assignment with every slab's cost <= limit; None if the last slab would exceed it."""
        cuts = [lo]
        for r in range(world - 1):
            a = cuts[-1]
            b = a + min_thickness
            if b > hi - min_thickness * (world - 1 - r):
                return None
            trial = cuts + [b]
            if cost(trial + [hi] * (world - len(trial)), r) > limit:
                return None  # even the thinnest admissible slab is too expensive
            while b + 1 <= hi - min_thickness * (world - 1 - r) and cost(cuts + [b + 1] + [hi] * (world - len(cuts) - 1), r) <= limit:
                b += 1
            cuts.append(b)
        cuts.append(hi)
        return cuts if cost(cuts, world - 1) <= limit else None

    if (hi - lo) < min_thickness * world:
        raise ValueError("%d occupied layers are too few for %d slabs of at least %d layers" % (hi - lo, world, min_thickness))
    lo_t, hi_t = 0.0, float(cum[-1]) * (1.0 + 2.0 * ghost_cost)
    best = sweep(hi_t)
    if best is None:
        raise ValueError("no admissible slab decomposition into %d ranks" % world)
    for _ in range(60):  # bisection on the admissible maximum cost
        mid = 0.5 * (lo_t + hi_t)
        c = sweep(mid)
        if c is None:
            lo_t = mid
        else:
            hi_t, best = mid, c
    return best


def make_slab(cuts, rank, world, n_global):
    s = SphSlab()
    s.layerLo = OPEN_LO if rank == 0 else cuts[rank]
    s.layerHi = OPEN_HI if rank == world - 1 else cuts[rank + 1]
    s.ghostLayers = GHOST_LAYERS
    s.hasLower = 1 if rank > 0 else 0
    s.hasUpper = 1 if rank < world - 1 else 0
    s.globalIdBits = max(1, int(n_global - 1).bit_length())
    return s


def local_indices(layers, slab):
    """Global ids (ascending) of the particles a rank holds initially: its own layers plus the ghost layers."""
    lo = slab.layerLo - slab.ghostLayers if slab.hasLower else OPEN_LO
    hi = slab.layerHi + slab.ghostLayers if slab.hasUpper else OPEN_HI
    return np.flatnonzero((layers >= lo) & (layers < hi)).astype(np.uint32)


class HipSlabBackend:
    """The MI355X backend: an owHIPSolver with head-room for a varying particle count; halo messages are torch tensors
    in HBM that libsphmi's pack / rebuild kernels write and read directly (sph_slab_pack / sph_slab_rebuild)."""

    def __init__(self, cfg, position, velocity, global_ids, slab, headroom=1.3):
        import torch
        self.torch = torch
        n = position.shape[0]
        cfg.particleCount = n
        cfg.capacity = int(n * headroom) + 4096
        self.solver = owHIPSolver(cfg, position, velocity)
        self.solver.slab_init(slab, global_ids)
        self.cap_records = cfg.capacity // 2
        self.record_words = SLAB_RECORD_WORDS
        dev = torch.device("cuda", cfg.device)
        # message frames: [payload word count | records ...]; libsphmi writes both parts on the device (sph_slab_pack_framed)
        self.frame_down = torch.empty(1 + self.cap_records * SLAB_RECORD_WORDS, dtype=torch.int32, device=dev)
        self.frame_up = torch.empty_like(self.frame_down)
        self.device = self.frame_device = dev

    def liquid_signature(self):
        return self.solver.slab_liquid_signature()

    def set_record_format(self, words, type_bits=0):
        self.solver.slab_set_record_format(words, type_bits)
        self.record_words = words

    @property
    def count(self):
        return self.solver.N

    def step(self, iteration):
        self.solver.step(iteration)

    def step_and_pack_framed(self, iteration):
        """The overlapped form of step() + pack_framed(): returns as soon as the two frames are packed; the rest of the step
        (the interior's last stage, the kept set) is still running and is waited for by rebuild(). kept is not known yet."""
        self.solver.slab_step_begin(iteration, C.c_void_p(self.frame_down.data_ptr()), C.c_void_p(self.frame_up.data_ptr()),
                                    self.cap_records)
        nd, nu = self.solver.slab_step_messages()
        return None, self.frame_down, nd * self.record_words, self.frame_up, nu * self.record_words

    def pack_framed(self):
        """(kept, frame_down, payload words, frame_up, payload words): frames ready to be sent from word 0."""
        kept, nd, nu = self.solver.slab_pack_framed(C.c_void_p(self.frame_down.data_ptr()), C.c_void_p(self.frame_up.data_ptr()),
                                                    self.cap_records)
        return kept, self.frame_down, nd * self.record_words, self.frame_up, nu * self.record_words

    def pack(self):
        kept, fd, nd, fu, nu = self.pack_framed()
        return kept, fd[1:1 + nd], fu[1:1 + nu]

    def rebuild(self, recv_down, recv_up):
        keep = []  # the received tensors must outlive the asynchronous rebuild kernels: released at the next rebuild

        def ptr_n(t):
            if t is None or t.numel() == 0:
                return None, 0
            t = t.to(self.device).contiguous()
            keep.append(t)
            return C.c_void_p(t.data_ptr()), t.numel() // self.record_words
        pd, nd = ptr_n(recv_down)
        pu, nu = ptr_n(recv_up)
        n = self.solver.slab_rebuild(pd, nd, pu, nu)
        self._keep = keep
        return n

    def rebuild_framed(self, frame_down, frame_up):
        """Asynchronous rebuild from complete frames [payload words | payload] (None where there is no neighbour); every count is
        read on the device. rebuild_finish() collects the result."""
        keep = []

        def ptr_cap(t):
            if t is None:
                return None, 0
            t = t.to(self.device).contiguous()
            keep.append(t)
            return C.c_void_p(t.data_ptr()), (t.numel() - 1) // self.record_words
        pd, cd = ptr_cap(frame_down)
        pu, cu = ptr_cap(frame_up)
        self.solver.slab_rebuild_framed(pd, cd, pu, cu)
        self._keep = keep

    def rebuild_finish(self):
        """(kept, records from below, from above, nothing_merged): nothing_merged means a frame was shorter than its message —
        fetch the rest and call rebuild() with the complete payloads."""
        return self.solver.slab_rebuild_finish()

    def wait_for(self, stream):
        """The solver's stream waits (on the device) for everything enqueued so far on the torch stream `stream`."""
        ev = self.torch.cuda.Event()
        ev.record(stream)
        self.solver.stream_wait_event(ev.cuda_event)
        self._keep_event = ev

    def owned_state(self):
        pos, vel, gid, owned = self.solver.slab_read()
        m = owned.astype(bool)
        return gid[m], pos[m], vel[m]


class SlabDecomposition:
    """Drives one backend per rank: step, then the halo exchange with the two neighbouring ranks."""

    def __init__(self, backend, rank, world, dist=None, comm_device=None, record_format="auto"):
        import torch
        self.torch = torch
        self.backend, self.rank, self.world, self.dist = backend, rank, world, dist
        self.lower = rank - 1 if rank > 0 else None
        self.upper = rank + 1 if rank < world - 1 else None
        if comm_device is None:
            comm_device = "cuda" if (dist is not None and dist.get_backend() == "nccl") else "cpu"
        self.comm_device = torch.device(comm_device) if comm_device == "cpu" else getattr(backend, "device", torch.device("cuda"))
        self.bytes_sent = 0
        self.transfers = 0          # batched point-to-point groups issued so far (1 per step in steady state)
        self._bound_out, self._bound_in, self._frames = {}, {}, {}
        # overlap the exchange with the tail of the step (sph_slab_step_begin); SPHMI_SLAB_OVERLAP=0 restores step-then-pack
        import os
        self.overlap = os.environ.get("SPHMI_SLAB_OVERLAP", "1") != "0"
        self._pending = None        # asynchronous exchange in flight (RCCL path): collected by finish()
        if record_format == "auto":
            record_format = self._agree_on_record_format()
        if record_format is not None and hasattr(backend, "set_record_format"):
            backend.set_record_format(*record_format)

    @property
    def rec(self):
        return getattr(self.backend, "record_words", SLAB_RECORD_WORDS)

    def _agree_on_record_format(self):
        """Compact 7-word records if every rank's non-boundary particles carry the same type word and velocity.w == 0 (the
        solvers checked their own particles when they were created; one all-gather of a word settles it for the run)."""
        if not hasattr(self.backend, "liquid_signature") or os_environ_flag("SPHMI_SLAB_FULL_RECORDS"):
            return None
        mine = int(self.backend.liquid_signature())
        sigs = [mine]
        if self.dist is not None and self.world > 1:
            if not hasattr(self.dist, "all_gather"):
                return None
            t = self.torch.tensor([mine], dtype=self.torch.int64, device=self.comm_device)
            out = [self.torch.zeros_like(t) for _ in range(self.world)]
            self.dist.all_gather(out, t)
            sigs = [int(o.item()) for o in out]
        real = {v for v in sigs if v != 0}
        if 0xffffffff in real or len(real) != 1:
            return None
        return SLAB_COMPACT_WORDS, real.pop()

    def _to_comm(self, t):
        return t if t.device == self.comm_device else t.to(self.comm_device)

    # ---- message framing. A link direction carries ONE transfer per step: [count word | payload | padding] of an agreed
    # length, so that the receiver can post its receive without first learning the size. Both ends of a link know the
    # payload length of the previous step and derive the same bound for the next one (12.5 % + 1024 records of slack);
    # the very first exchange, and a payload that outgrows its bound (the count word says so), use an explicit second
    # transfer of exactly the missing words.
    @staticmethod
    def next_bound(words, record_words=SLAB_RECORD_WORDS):
        rec = words // record_words
        rec = rec + rec // 8 + 1024
        return ((rec + 255) // 256) * 256 * record_words

    def _next_bound(self, words):
        try:
            return self.next_bound(words, self.rec)
        except TypeError:  # (tests replace next_bound by a one-argument function)
            return self.next_bound(words)

    def _frame(self, key, words):
        """A reusable int32 buffer of 1 + words entries on the communication device."""
        t = self._frames.get(key)
        if t is None or t.numel() < 1 + words:
            t = self.torch.empty(1 + words, dtype=self.torch.int32, device=self.comm_device)
            self._frames[key] = t
        return t[:1 + words]

    def _transfer(self, sends, recvs):
        """One batched group of point-to-point operations: sends = [(tensor, peer)], recvs likewise."""
        dist = self.dist
        ops = [dist.P2POp(dist.isend, t, p) for t, p in sends] + [dist.P2POp(dist.irecv, t, p) for t, p in recvs]
        if not ops:
            return
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if self.comm_device.type == "cuda":
            # RCCL completes on torch's communication stream; the host reads the count words and the rebuild kernels run
            # on the solver's own stream
            self.torch.cuda.synchronize(self.comm_device)

    def exchange(self, prepacked=None):
        torch = self.torch
        # Zero-copy framing when the backend keeps [count | payload] frames on the communication device (HipSlabBackend under
        # RCCL): the frame is sent as it is. Otherwise (CPU-staged tests, other backends) frames are assembled here.
        zero_copy = hasattr(self.backend, "pack_framed") and getattr(self.backend, "frame_device", None) == self.comm_device
        if prepacked is not None or zero_copy:
            kept, frame_down, nd, frame_up, nu = prepacked if prepacked is not None else self.backend.pack_framed()
            frames = {self.lower: frame_down, self.upper: frame_up}
            n_words = {self.lower: nd, self.upper: nu}
            payload = None
            if not zero_copy:  # frames live on another device than the communication runs on (CPU-staged tests)
                payload = {p: f[1:1 + n_words[p]] for p, f in frames.items()}
        else:
            kept, msg_down, msg_up = self.backend.pack()
            payload = {self.lower: msg_down, self.upper: msg_up}
            n_words = {p: int(t.numel()) for p, t in payload.items()}
        if self.world == 1:
            return self.backend.rebuild(None, None)
        peers = [p for p in (self.lower, self.upper) if p is not None]
        n_out = {p: n_words[p] for p in peers}
        first = not self._bound_out
        if first:  # no agreed bounds yet: the transfer carries only the count word
            for p in peers:
                self._bound_out[p] = 0
                self._bound_in[p] = 0
        out, inn = {}, {}
        for p in peers:
            b = self._bound_out[p]
            if zero_copy and 1 + b <= frames[p].numel():
                out[p] = frames[p][:1 + b]
            else:  # assemble the frame here (also when the agreed length outgrows the backend's own buffer)
                src = frames[p][1:1 + n_out[p]] if zero_copy else payload[p]
                f = self._frame(("out", p), b)
                f[0:1] = torch.tensor([n_out[p]], dtype=torch.int32)
                k = min(n_out[p], b)
                if k:
                    f[1:1 + k] = self._to_comm(src[:k])
                out[p] = f
            inn[p] = self._frame(("in", p), max(self._bound_in[p], 0))
        self._transfer([(out[p], p) for p in peers], [(inn[p], p) for p in peers])
        self.bytes_sent += sum(4 * out[p].numel() for p in peers)
        heads = torch.stack([inn[p][0] for p in peers]).cpu().tolist()  # one device->host copy for all count words
        n_in = dict(zip(peers, (int(v) for v in heads)))
        # what did not fit the agreed bounds (always the case in the first exchange): exact sizes are now known to both ends
        sends, recvs, recv = [], [], {}
        for p in peers:
            b_out, b_in = self._bound_out[p], self._bound_in[p]
            if n_out[p] > b_out:
                rest = frames[p][1 + b_out:1 + n_out[p]] if zero_copy else self._to_comm(payload[p][b_out:]).contiguous()
                sends.append((rest, p))
                self.bytes_sent += 4 * rest.numel()
            if n_in[p] > b_in:
                full = torch.empty(n_in[p], dtype=torch.int32, device=self.comm_device)
                if b_in:
                    full[:b_in] = inn[p][1:1 + b_in]
                recvs.append((full[b_in:], p))
                recv[p] = full
            else:
                recv[p] = inn[p][1:1 + n_in[p]]
        self._transfer(sends, recvs)
        for p in peers:
            self._bound_out[p] = self._next_bound(n_out[p])
            self._bound_in[p] = self._next_bound(n_in[p])
        self.transfers += 1 + (1 if (sends or recvs) else 0)
        if hasattr(self.backend, "rebuild_framed") and not (sends or recvs) and all(1 + n_in[p] <= inn[p].numel() for p in peers):
            # the received frames are complete: let the backend read the counts from the frames on the device, as the RCCL path does
            self.backend.rebuild_framed(inn.get(self.lower), inn.get(self.upper))
            kept, n_down, n_up, nothing = self.backend.rebuild_finish()
            assert not nothing and n_down * self.rec == n_in.get(self.lower, 0) and n_up * self.rec == n_in.get(self.upper, 0)
            return self.backend.count
        return self.backend.rebuild(recv.get(self.lower), recv.get(self.upper))

    # ---- RCCL path: one host wait per step. The frames the backend packed on the device are sent in place; the solver's stream
    # waits for the receive through an event; the rebuild reads the received counts on the device; the new local count (and
    # whether a message outgrew its agreed frame) is collected by finish() at the start of the next step.
    def _can_run_async(self):
        # (HipSlabBackend under RCCL; the protocol itself does not care where the frames live, and the CPU tests run it over gloo
        # with a backend that offers the same four calls)
        return (self.overlap and self.world > 1 and all(hasattr(self.backend, m) for m in ("rebuild_framed", "rebuild_finish",
                                                                                           "step_and_pack_framed", "wait_for"))
                and getattr(self.backend, "frame_device", None) == self.comm_device and not os_environ_flag("SPHMI_SLAB_SYNC_EXCHANGE"))

    def _exchange_async(self, prepacked):
        torch, dist = self.torch, self.dist
        _, frame_down, nd, frame_up, nu = prepacked
        frames = {self.lower: frame_down, self.upper: frame_up}
        n_out = {self.lower: nd, self.upper: nu}
        peers = [p for p in (self.lower, self.upper) if p is not None]
        if not self._bound_out:  # first exchange: the frames carry the count word only; finish() fetches the payloads
            for p in peers:
                self._bound_out[p] = self._bound_in[p] = 0
        out, inn = {}, {}
        for p in peers:
            b = min(self._bound_out[p], frames[p].numel() - 1)
            self._bound_out[p] = b
            out[p] = frames[p][:1 + b]
            inn[p] = self._frame(("in", p), self._bound_in[p])
        ops = [dist.P2POp(dist.isend, out[p], p) for p in peers] + [dist.P2POp(dist.irecv, inn[p], p) for p in peers]
        for w in dist.batch_isend_irecv(ops):
            w.wait()  # torch's current stream waits for RCCL — the host does not
        # ... and the solver's stream waits for that
        self.backend.wait_for(torch.cuda.current_stream(self.comm_device) if self.comm_device.type == "cuda" else None)
        self.backend.rebuild_framed(inn.get(self.lower), inn.get(self.upper))
        self.bytes_sent += sum(4 * out[p].numel() for p in peers)
        self.transfers += 1
        self._pending = dict(frames=frames, n_out={p: n_out[p] for p in peers}, inn=inn, peers=peers)

    def finish(self):
        """Collect the asynchronous exchange of the last step (no-op otherwise): the new local count, the next frame lengths,
        and — rare — the part of a message that did not fit its agreed frame. Returns the local particle count."""
        pend, self._pending = self._pending, None
        if pend is None:
            return self.backend.count
        torch = self.torch
        kept, n_down, n_up, nothing = self.backend.rebuild_finish()
        peers, frames, inn, n_out = pend["peers"], pend["frames"], pend["inn"], pend["n_out"]
        n_in = {self.lower: n_down * self.rec, self.upper: n_up * self.rec}
        sends, recvs, recv = [], [], {}
        for p in peers:
            b_out, b_in = self._bound_out[p], self._bound_in[p]
            if n_out[p] > b_out:  # both ends of a link see the same two numbers, so both take part in the second transfer
                rest = frames[p][1 + b_out:1 + n_out[p]]
                sends.append((rest, p))
                self.bytes_sent += 4 * rest.numel()
            if n_in[p] > b_in:
                full = torch.empty(n_in[p], dtype=torch.int32, device=self.comm_device)
                if b_in:
                    full[:b_in] = inn[p][1:1 + b_in]
                recvs.append((full[b_in:], p))
                recv[p] = full
            else:
                recv[p] = inn[p][1:1 + n_in[p]]
        if sends or recvs:
            self._transfer(sends, recvs)
            self.transfers += 1
        if nothing:  # a received frame was shorter than its message: rebuild from the complete payloads
            self.backend.rebuild(recv.get(self.lower), recv.get(self.upper))
        for p in peers:
            self._bound_out[p] = self._next_bound(n_out[p])
            self._bound_in[p] = self._next_bound(n_in[p])
        return self.backend.count

    def step(self, iteration):
        import time
        if self._can_run_async():
            t0 = time.perf_counter()  # (includes the wait for the step itself up to the packed messages)
            self.finish()
            self._exchange_async(self.backend.step_and_pack_framed(iteration))
            self.exchange_seconds = getattr(self, "exchange_seconds", 0.0) + time.perf_counter() - t0
            return None  # the count arrives with finish()
        if self.overlap and self.world > 1 and hasattr(self.backend, "step_and_pack_framed"):
            t0 = time.perf_counter()  # (includes the wait for the step itself up to the packed messages)
            n = self.exchange(self.backend.step_and_pack_framed(iteration))
            self.exchange_seconds = getattr(self, "exchange_seconds", 0.0) + time.perf_counter() - t0
            return n
        self.backend.step(iteration)
        t0 = time.perf_counter()
        n = self.exchange()
        self.exchange_seconds = getattr(self, "exchange_seconds", 0.0) + time.perf_counter() - t0  # host time incl. waits
        return n
