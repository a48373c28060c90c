"""Multi-GPU plumbing: one process per GPU, torch.distributed.

The prove path shards in two ways:
  * independent proofs (the reference's own concurrency model: one credential per task,
    sample/client_helper/src/main.rs:177-216) -> every rank holds a full replica, no collective;
  * one proof, every MSM range-sharded (the five independent sums of forks/groth16/src/prover.rs:66,74,266):
    each rank computes partial sums over its contiguous range of every query, and the only exchange
    step is an all_gather of 384 bytes per rank (four G1 points + one G2 point), after which every rank
    (or just rank 0) finishes A, B, C exactly as prover.rs:76-135.

Two process groups, on purpose:
  * the CONTROL plane (barriers, max-over-ranks of a timing, agreement on what to do next) always runs over gloo on host
    tensors: it carries a few bytes, and nothing that is measured may depend on a GPU collective having come up;
  * the DATA plane - the 384-byte all_gather of a sharded proof - runs over RCCL ("nccl") on device tensors, on a group
    that is created lazily by `open_data_group`, in a helper thread with a deadline, and agreed on over the control plane:
    if RCCL does not come up on every rank (or hangs), every rank falls back to the gloo group together and says so.
"""
from __future__ import annotations

import threading
import time
from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

PARTIAL_BYTES = 384
POISON = b"\xff" * PARTIAL_BYTES     # no partial-sum record looks like this: coordinates are canonical (< q < 2^254)

_control = None
# set (to a description) when open_data_group gave up on a helper thread that was still inside RCCL: the thread cannot be
# cancelled, keeps whatever it holds and may finish or abort later - results measured afterwards in this process say so
RCCL_TAINTED = None


def shard_range(n: int, rank: int, count: int) -> Tuple[int, int]:
    """Contiguous range of a length-n query owned by `rank` (same formula as csrc/prover.hip shard_range)."""
    if count <= 1:
        return 0, n
    return n * rank // count, n * (rank + 1) // count


def control_group():
    """the gloo group the control plane runs on: the default group when that is gloo, else a gloo group made once"""
    global _control
    if not dist.is_initialized():
        return None
    if _control is None:
        _control = dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
    return _control


def _on_host(group) -> bool:
    return dist.get_backend(group) == "gloo"


def barrier_sync(world: int, group=None) -> None:
    """device idle, every rank here, device idle.  `group` defaults to the control plane."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier(group=group if group is not None else control_group())
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(x: float, world: int, device=None, group=None) -> float:
    if world <= 1:
        return x
    group = group if group is not None else control_group()
    t = torch.tensor([x], dtype=torch.float64, device=torch.device("cpu") if _on_host(group) else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def min_over_ranks(x: float, world: int, device=None, group=None) -> float:
    return -max_over_ranks(-x, world, device, group)


def gather_over_ranks(x: float, world: int, group=None) -> list:
    """every rank's value, rank order (control plane)"""
    if world <= 1:
        return [x]
    group = group if group is not None else control_group()
    t = torch.tensor([x], dtype=torch.float64)
    out = torch.empty(world, dtype=torch.float64)
    dist.all_gather_into_tensor(out, t, group=group)
    return [float(v) for v in out]


def open_data_group(device, backend: str = "nccl", deadline_s: float = 90.0):
    """-> (group, backend actually in use, error or None).  backend "nccl" = RCCL: the group is created and one
    device all_gather is pushed through it in a helper thread; whether it worked is agreed over the control plane (min
    over ranks), so that either every rank uses RCCL or every rank uses the gloo control group.  A helper thread that is
    still inside RCCL at the deadline is left behind (daemon) and counts as a failure."""
    world = dist.get_world_size()
    ctl = control_group()
    if backend == "gloo":
        return ctl, "gloo", None
    res = {}

    def attempt():
        try:
            torch.cuda.set_device(device)        # a new thread starts on device 0, whatever the main thread chose
            g = None
            if backend == "nccl":
                try:        # bound to this rank's device: the communicator is made here, not at the first collective
                    g = dist.new_group(backend=backend, device_id=device)
                except (TypeError, ValueError):
                    g = None
            if g is None:   # (a torch whose new_group takes no device_id, or refuses one under a gloo default group)
                g = dist.new_group(backend=backend)
            mine = torch.full((8,), float(dist.get_rank()), device=device)
            allv = torch.empty(8 * world, device=device)
            dist.all_gather_into_tensor(allv, mine, group=g)
            torch.cuda.current_stream(device).synchronize()
            got = allv.cpu().view(world, 8)[:, 0].tolist()
            if got != [float(k) for k in range(world)]:
                raise RuntimeError("first all_gather returned %r" % (got,))
            res["group"] = g
        except BaseException as e:   # noqa: BLE001 - reported, never raised into the measurement
            res["error"] = "%s: %s" % (type(e).__name__, str(e).strip().splitlines()[0][:300] if str(e).strip() else "")
    th = threading.Thread(target=attempt, daemon=True)
    th.start()
    th.join(deadline_s)
    if th.is_alive():
        global RCCL_TAINTED
        res["error"] = "no answer from the %s group after %.0f s" % (backend, deadline_s)
        RCCL_TAINTED = "rank %d: the helper thread that opens the %s group was still inside the library at the %.0f s deadline and " \
                       "was left behind (it cannot be cancelled)" % (dist.get_rank(), backend, deadline_s)
    ok = 1.0 if "group" in res and "error" not in res else 0.0
    all_ok = min_over_ranks(ok, world, group=ctl) > 0.5
    if all_ok:
        return res["group"], backend, None
    err = res.get("error") or "another rank could not open its %s group" % backend
    if "group" in res and not th.is_alive():
        # this rank's own group came up but another rank's did not: a communicator nobody will use is not left alive
        try:
            dist.destroy_process_group(res["group"])
        except BaseException as e:   # noqa: BLE001 - reported with the rest
            err += " (and destroying this rank's own %s group failed: %s)" % (backend, type(e).__name__)
    return ctl, "gloo", err


class PartialGather:
    """The one exchange step of a sharded proof: an all_gather of 384 bytes per rank (four G1 partial sums + one G2), or of
    `batch` such records at once.  The tensors are allocated once and reused for every proof; with RCCL ("nccl") they
    live on the device and the collective runs over xGMI, with gloo they are host tensors."""

    def __init__(self, device, group=None, batch: int = 1):
        self.group = group
        self.world = dist.get_world_size(group)
        self.on_host = _on_host(group)
        self.batch = batch
        n = PARTIAL_BYTES * batch
        dev = torch.device("cpu") if self.on_host else device
        self._mine = torch.empty(n, dtype=torch.uint8, device=dev)
        self._all = torch.empty(self.world * n, dtype=torch.uint8, device=dev)
        if self.on_host:
            self._stage_in = self._mine
            self._stage_out = self._all
        else:   # page-locked staging on the host side of the two small copies
            self._stage_in = torch.empty(n, dtype=torch.uint8).pin_memory()
            self._stage_out = torch.empty(self.world * n, dtype=torch.uint8).pin_memory()

    def __call__(self, partial: bytes) -> bytes:
        """partial: batch x 384 bytes -> world x batch x 384 bytes, rank-major"""
        self._stage_in.numpy()[:] = np.frombuffer(partial, dtype=np.uint8)
        if not self.on_host:
            self._mine.copy_(self._stage_in, non_blocking=True)
        dist.all_gather_into_tensor(self._all, self._mine, group=self.group)
        if not self.on_host:
            self._stage_out.copy_(self._all, non_blocking=True)
            torch.cuda.current_stream(self._all.device).synchronize()
        return self._stage_out.numpy().tobytes()


def gather_partials(partial: bytes, device, group=None) -> bytes:
    """all_gather of one rank's 384-byte partial-sum record -> world x 384 bytes, rank order (one-off form of
    PartialGather)."""
    return PartialGather(device, group)(partial)


class WitnessMapFailed(Exception):
    """the source rank's witness map failed - raised by HScalarScatter.exchange AFTER the scatter every rank takes part in, so
    the sequence of collectives is intact and the job can be poisoned through the gather; `cause` is the original exception.
    Anything else exchange() raises comes from the collective itself: the transport is gone and nothing more can be exchanged."""

    def __init__(self, cause):
        super().__init__(repr(cause))
        self.cause = cause


class HScalarScatter:
    """The exchange step of the "scatter" arrangement: ONE rank computes every shard's h scalars (witness_map_coset), each rank
    receives its slice.  Chunks are padded to the longest slice (strided shards are all domain_size / world long); buffers are
    allocated once; with RCCL they live on the device - 32 B x domain_size / world per peer, each over its own xGMI link - with
    gloo on the host.  `slots` receive buffers let several proofs' slices be held at once (ShardedProver.prove_stream); with
    `rotate` any rank may be asked to be the source (it then holds the 32 B x domain_size send buffer too)."""

    def __init__(self, prover, device, group, rank: int, world: int, slots: int = 1, rotate: bool = False, src_ranks=(0,), spans=None):
        self.prover, self.group, self.rank, self.world, self.rotate = prover, group, rank, world, rotate
        self.on_host = (world <= 1) or _on_host(group)
        if spans is not None:        # unequal shares (cg_options.shard_span): shard p owns [D·lo/10000, D·hi/10000) of the coset values
            D = prover.domain_size
            self.slices = [(D * lo // 10000, D * hi // 10000 - D * lo // 10000) for lo, hi in spans]
        else:
            self.slices = [prover.h_scalars_slice(p) for p in range(world)]
        self.chunk = max(c for _, c in self.slices) * 32
        total = max(o + c for o, c in self.slices)
        dev = torch.device("cpu") if self.on_host else device
        self._recv = [torch.zeros(max(1, self.chunk), dtype=torch.uint8, device=dev) for _ in range(max(1, slots))]
        sends = rank in src_ranks or rotate          # ranks that may be asked to be the source hold the send buffer
        self._all = torch.zeros(total * 32, dtype=torch.uint8, device=dev) if sends else None
        # a real context (cg_witness_map_coset) can write into host memory it is handed; page-locked when a GPU is there
        self._direct_host = (self.on_host and sends and hasattr(prover, "_h") and hasattr(prover, "domain_size")
                             and total == getattr(prover, "domain_size", -1))
        if self._direct_host and torch.cuda.is_available():
            try:
                self._all = self._all.pin_memory()
            except Exception:
                pass
        equal = all(c * 32 == self.chunk for _, c in self.slices)
        # equal slices: the scatter list is views of the one vector; otherwise padded copies
        self._pad = None if equal or not sends else [torch.zeros(max(1, self.chunk), dtype=torch.uint8, device=dev) for _ in range(world)]
        self._failure = None

    def _global(self, r: int) -> int:
        return dist.get_global_rank(self.group, r) if (self.world > 1 and self.group is not None) else r

    def compute(self, assignment, on_device: bool, seconds: dict, wm=None):
        """the source's part of an exchange on its own: run the witness map into the send buffer.  A failure is kept and
        raised by the exchange(computed=True) that follows, after its scatter."""
        t0 = time.perf_counter()
        self._failure = None
        try:
            run = wm if wm is not None else (lambda **kw: self.prover.witness_map_coset(assignment, on_device=on_device, **kw))
            if self.on_host:
                if self._direct_host:       # the library writes straight into the (page-locked) tensor the scatter sends from
                    run(out_host=self._all.data_ptr())
                else:
                    got = np.frombuffer(bytes(run()), dtype=np.uint8)
                    self._all.numpy()[:] = got[:self._all.numel()]    # (a stand-in may return more than its shards' slices cover)
            else:
                run(out_dev=self._all.data_ptr())
        except BaseException as e:   # noqa: BLE001
            self._failure = e
        seconds["witness_map"] += time.perf_counter() - t0

    def exchange(self, assignment, on_device: bool, seconds: dict, src: int = 0, slot: int = 0, wm=None, computed: bool = False):
        """Rank `src` (of the group) runs the witness map on `assignment` and scatters; every rank's slice lands in receive
        buffer `slot`.  -> (this rank's slice: a device address or numpy bytes, whether it is on the device).  A witness
        map that fails on the source is raised there AFTER the scatter the other ranks are already waiting in.
        wm (optional): the witness map of an OPEN two-call proof (OpenPartial.witness_map_coset) to use instead of the
        context's own - it runs on the working set the open proof holds.
        computed: the source has run compute() already (two sources working at the same time before their scatters)."""
        t0 = time.perf_counter()
        off, cnt = self.slices[self.rank]
        failure = None
        if self.rank == src:
            if not computed:
                self.compute(assignment, on_device, {"witness_map": 0.0}, wm)
            failure, self._failure = self._failure, None
        t1 = time.perf_counter()
        recv = self._recv[slot]
        if self.world > 1:
            lst = None
            if self.rank == src:
                if self._pad is None:
                    lst = [self._all[o * 32:o * 32 + self.chunk] for o, _ in self.slices]
                else:
                    for p, (o, c) in enumerate(self.slices):
                        self._pad[p][:c * 32].copy_(self._all[o * 32:(o + c) * 32])
                    lst = self._pad
            dist.scatter(recv, lst, src=self._global(src), group=self.group)
            if not self.on_host:
                torch.cuda.current_stream(recv.device).synchronize()
            mine = recv
        else:
            mine = self._all[off * 32:(off + cnt) * 32]
        if failure is not None:
            raise WitnessMapFailed(failure)     # the caller turns it into a poison record for the gather that follows
        t2 = time.perf_counter()
        seconds["witness_map"] += t1 - t0
        seconds["scatter"] += t2 - t1
        if self.on_host:
            return mine.numpy()[:cnt * 32], False
        return mine.data_ptr(), True

    def __call__(self, assignment, on_device: bool, seconds: dict, wm=None):
        return self.exchange(assignment, on_device, seconds, 0, 0, wm)


class ShardedProver:
    """One proof across all ranks.  `prover` is any object with prove_partial(assignment, r, on_device) and
    assemble(partials, n_shards, r, s) — a crescent_credentials_amd.Prover loaded with shard_rank/shard_count
    on the GPU, or a stand-in in the CPU (gloo) tests.  `seconds` accumulates where the wall time of the proofs went
    (this rank's partial sums, the all_gather, the host finish).

    `prove*` is one proof at a time (latency).  `prove_stream` keeps several sharded proofs in flight per rank, as the
    reference's host does with whole proofs (one task per credential, sample/client_helper/src/main.rs:177-216): worker
    threads compute this rank's partial sums of proofs k, k+1, ... concurrently (a shard context with proof_slots > 1),
    ONE gather thread exchanges the 384-byte records strictly in proof order (every rank issues the same sequence of
    collectives whatever order its partial sums finish in), and the host finish of proof k runs while the partial sums
    of the next proofs are on the GPU."""

    def __init__(self, prover, device, group=None, arrangement: str = "recompute", rotate: bool = False, stream_slots: int = 8,
                 two_call: bool = False, split_map: bool = False, spans=None):
        """arrangement ("run the witness map on GPU 0 and scatter h, or recompute it redundantly on every GPU -
        measure both"):
          "recompute" - every rank runs the witness map for its own share of the h MSM (cg_prove_partial); one collective
                        per proof, the 384-byte all_gather;
          "scatter"   - rank 0 runs the witness map once for all shards (cg_witness_map_coset: domain_size x 32 B, shard-major),
                        a scatter hands every rank its slice, and the ranks prove with it (cg_prove_partial_q; their contexts
                        may be loaded with CG_FLAG_H_SCALARS_EXTERNAL and then hold no witness-map memory at all); two
                        collectives per proof.  `prover` needs witness_map_coset / h_scalars_slice / prove_partial_q.
                        `rotate`: in a STREAM of sharded proofs (prove_stream) the rank that runs the witness map of proof k is
                        k mod world instead of always rank 0 - every rank then needs a context with witness-map resources -
                        so that the one full witness map per proof is spread over the ranks; `stream_slots` bounds the proofs in
                        flight of such a stream (a receive buffer each)."""
        if arrangement not in ("recompute", "scatter"):
            raise ValueError("arrangement must be 'recompute' or 'scatter'")
        if two_call and arrangement != "scatter":
            raise ValueError("two_call belongs to the 'scatter' arrangement")
        # two_call ("scatter" only): every rank OPENS the proof first (cg_prove_partial_q_begin: its l, a, b1, b2 partial sums are
        # queued and run) and only then takes part in the witness map + scatter; the h share follows with the slice
        # (cg_prove_partial_q_finish).  The assignment-driven MSMs leave the critical path: witness map -> scatter -> h share.
        # `prover` needs prove_partial_q_begin (-> an object with witness_map_coset / finish / abort).
        self.two_call = bool(two_call)
        # split_map (with two_call): the witness map in two halves on two ranks - rank 0 computes the a side, vinv·a(g w^j), rank 1
        # the b side, b(g w^j) (one sparse product and two transforms each: half the witness map's time, at the same time) - each
        # scatters its side, and every shard multiplies its two slices itself (cg_prove_partial_q_finish2).  Two scatters and one
        # gather per proof.  `prover` on ranks 0 and 1 needs witness-map resources; its open proofs need witness_map_coset_half
        # and finish2.
        if split_map and not two_call:
            raise ValueError("split_map belongs to the two-call form of the 'scatter' arrangement")
        self.split_map = bool(split_map)
        # spans ("scatter" only): the (lo, hi) every rank's context was loaded with (cg_options.shard_span, 1/10000 of every query),
        # rank order - unequal shares, so that the ranks that also compute (half of) the witness map carry less of the MSMs
        self.spans = list(spans) if spans is not None else None
        if self.spans is not None and arrangement != "scatter":
            raise ValueError("spans belong to the 'scatter' arrangement")
        self.prover = prover
        self.device = device
        self.group = group
        self.arrangement = arrangement
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.src_b = min(1, self.world - 1)      # split_map: the rank that computes the b side
        self.last_parts = None
        self.all_gathers = 0          # all_gathers issued so far (one per proof when world > 1)
        self.scatters = 0             # scatters issued so far ("scatter" arrangement: one per proof)
        self.proofs = 0
        self.seconds = {"partial": 0.0, "gather": 0.0, "assemble": 0.0, "witness_map": 0.0, "scatter": 0.0}
        self._gather = PartialGather(device, group) if self.world > 1 else None
        self.rotate = bool(rotate)
        self._stream_slots = max(1, stream_slots)
        self._scatter = (HScalarScatter(prover, device, group, self.rank, self.world, slots=self._stream_slots, rotate=self.rotate,
                                        spans=self.spans) if arrangement == "scatter" else None)
        self._scatter_b = (HScalarScatter(prover, device, group, self.rank, self.world, slots=1, src_ranks=(self.src_b,), spans=self.spans)
                           if self.split_map else None)
        # A stream of scatter-arrangement proofs issues BOTH its collectives - the scatter of job k and the gather of job
        # k - in_flight - from ONE communication thread on a fixed schedule (_prove_stream_scatter), on the one group.  (Round 5
        # issued them from two free-running threads on two groups: each group saw its own sequence in order, but nothing ordered
        # the two against each other across ranks - rank A could enqueue scatter-then-gather where rank B enqueued
        # gather-then-scatter, which RCCL documents as a deadlock hazard for communicators used concurrently.)

    def _prove(self, assignment, on_device: bool, r: int, s: int):
        """one sharded proof.  A shard that fails here does not leave the other ranks waiting in the collective either: the
        failing rank sends a poison record and then raises its own exception, every other rank raises a RuntimeError
        naming it (the protocol of prove_stream, for one proof)."""
        t0 = time.perf_counter()
        failure = None
        part = None
        if self._scatter is not None and self.split_map:
            opened = self.prover.prove_partial_q_begin(assignment, r, on_device=on_device)     # a failure here precedes every collective
            qa = qb = None
            q_on_device = False
            # the two sources compute their sides FIRST - at the same time, each on its own GPU - and only then the scatters: a
            # source that went into the other's scatter before computing would start when that one had finished
            for side, sc, src in ((0, self._scatter, 0), (1, self._scatter_b, self.src_b)):
                if self.rank == src:
                    sc.compute(assignment, on_device, self.seconds, wm=lambda side=side, **kw: opened.witness_map_coset_half(side, **kw))
            for side, sc, src in ((0, self._scatter, 0), (1, self._scatter_b, self.src_b)):    # both scatters, whatever fails
                try:
                    got, q_on_device = sc.exchange(assignment, on_device, self.seconds, src, 0, computed=True)
                    if side == 0:
                        qa = got
                    else:
                        qb = got
                except WitnessMapFailed as e:
                    failure = failure or e.cause
                except BaseException:        # noqa: BLE001 - the collective failed: give the slot back, nothing can follow
                    opened.abort()
                    raise
                self.scatters += 1
            t0 = time.perf_counter()
            if failure is None:
                try:
                    part = opened.finish2(qa, qb, q_on_device)
                except BaseException as e:   # noqa: BLE001 - raised below, after the gather
                    failure = e
            else:
                opened.abort()
        elif self._scatter is not None and self.two_call:
            opened = self.prover.prove_partial_q_begin(assignment, r, on_device=on_device)     # a failure here precedes every collective
            q = None
            try:
                q, q_on_device = self._scatter(assignment, on_device, self.seconds, wm=opened.witness_map_coset)
            except WitnessMapFailed as e:
                failure = e.cause
            except BaseException:            # noqa: BLE001 - the collective failed: give the slot back, nothing can follow
                opened.abort()
                raise
            self.scatters += 1
            t0 = time.perf_counter()
            if failure is None:
                try:
                    part = opened.finish(q, q_on_device)
                except BaseException as e:   # noqa: BLE001 - raised below, after the gather
                    failure = e
            else:
                opened.abort()
        elif self._scatter is not None:
            q = None
            try:
                q, q_on_device = self._scatter(assignment, on_device, self.seconds)
            except WitnessMapFailed as e:            # the scatter itself was done: the job is poisoned through the gather below
                failure = e.cause
            # (anything else the scatter raises comes from the collective: no gather can follow, it propagates)
            self.scatters += 1
            t0 = time.perf_counter()
            if failure is None:
                try:
                    part = self.prover.prove_partial_q(assignment, q, r, on_device=on_device, q_on_device=q_on_device)
                except BaseException as e:           # noqa: BLE001 - raised below, after the collective every rank is about to enter
                    failure = e
        else:
            try:
                part = self.prover.prove_partial(assignment, r, on_device=on_device)
            except BaseException as e:               # noqa: BLE001
                failure = e
        if failure is not None:
            if self.world <= 1:
                raise failure
            part = POISON
        t1 = time.perf_counter()
        if self.world > 1:
            parts = self._gather(part)
            self.all_gathers += 1
            if failure is not None:
                raise failure
            bad = [q for q in range(self.world) if parts[PARTIAL_BYTES * q:PARTIAL_BYTES * (q + 1)] == POISON]
            if bad:
                raise RuntimeError("sharded proof: " + ", ".join("rank %d failed" % q for q in bad))
        else:
            parts = part
        t2 = time.perf_counter()
        proof = self.prover.assemble(parts, self.world, r, s)
        t3 = time.perf_counter()
        self.seconds["partial"] += t1 - t0
        self.seconds["gather"] += t2 - t1
        self.seconds["assemble"] += t3 - t2
        self.proofs += 1
        self.last_parts = parts          # the records the latest proof was assembled from, rank order (what a checker looks at first)
        return proof

    def prove(self, full_assignment, r: int, s: int):
        return self._prove(full_assignment, False, r, s)

    def prove_dev(self, d_ptr: int, r: int, s: int):
        return self._prove(d_ptr, True, r, s)

    def prove_stream(self, jobs, in_flight: int, on_device: bool = True, done_times: Optional[list] = None):
        """jobs: list of (assignment, r, s) - the SAME list, in the same order, on every rank.  -> proofs in job order.
        `in_flight` sharded proofs are kept going on this rank.  done_times (optional list) receives perf_counter() of
        every proof's completion, job order.

        A shard that fails does not leave the other ranks waiting in a collective: the failing rank still takes part in
        every all_gather, sending a poison record for the job it could not do; every rank sees the poison, skips that job's
        host finish, and at the end of the stream the failing rank raises its own exception and the others a RuntimeError
        naming the rank and the job.  The sequence of collectives is the same on every rank whatever fails."""
        n = len(jobs)
        if n == 0:
            return []
        if self._scatter is not None:
            return self._prove_stream_scatter(jobs, in_flight, on_device, done_times)
        # the poison check below reads one 384-byte record per rank: a batched gather (PartialGather(batch > 1), rank-major
        # world x batch x 384) would be mis-parsed
        assert self._gather is None or self._gather.batch == 1, "prove_stream exchanges one record per collective"
        in_flight = max(1, min(in_flight, n))
        parts = [None] * n           # this rank's 384-byte record of proof k
        gathered = [None] * n        # all ranks' records of proof k (None: some rank failed on it)
        proofs = [None] * n
        if done_times is not None:
            done_times[:] = [0.0] * n
        have_part = [threading.Event() for _ in range(n)]
        have_all = [threading.Event() for _ in range(n)]
        nxt = [0]
        lock = threading.Lock()
        mine, theirs, fatal = [], [], []      # (job, exception) of this rank; (job, rank) poisoned by others; the gather itself

        def worker():
            while not fatal:
                with lock:
                    k = nxt[0]
                    nxt[0] += 1
                if k >= n:
                    return
                a, r, s = jobs[k]
                t0 = time.perf_counter()
                try:
                    parts[k] = self.prover.prove_partial(a, r, on_device=on_device)
                except BaseException as e:   # noqa: BLE001 - surfaces below, after the stream
                    mine.append((k, e))
                    parts[k] = POISON
                t1 = time.perf_counter()
                have_part[k].set()
                have_all[k].wait()
                if fatal or gathered[k] is None:
                    continue
                try:
                    t2 = time.perf_counter()
                    proofs[k] = self.prover.assemble(gathered[k], self.world, r, s)
                    t3 = time.perf_counter()
                except BaseException as e:   # noqa: BLE001
                    mine.append((k, e))
                    continue
                if done_times is not None:
                    done_times[k] = t3
                with lock:
                    self.seconds["partial"] += t1 - t0
                    self.seconds["assemble"] += t3 - t2
                    self.proofs += 1

        def gatherer():
            try:
                if self._gather is not None and not self._gather.on_host:
                    torch.cuda.set_device(self.device)   # a new thread starts on device 0
                for k in range(n):
                    have_part[k].wait()
                    t0 = time.perf_counter()
                    if self.world > 1:
                        allp = self._gather(parts[k])
                        self.all_gathers += 1
                    else:
                        allp = parts[k]
                    bad = [q for q in range(self.world) if allp[PARTIAL_BYTES * q:PARTIAL_BYTES * (q + 1)] == POISON]
                    if bad:
                        theirs.extend((k, q) for q in bad)
                    else:
                        gathered[k] = allp
                    self.seconds["gather"] += time.perf_counter() - t0
                    have_all[k].set()
            except BaseException as e:       # noqa: BLE001 - the transport itself failed: nothing more can be exchanged
                fatal.append(e)
                for ev in have_all:
                    ev.set()

        ts = [threading.Thread(target=worker) for _ in range(in_flight)] + [threading.Thread(target=gatherer)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if fatal:
            raise fatal[0]
        if mine:
            raise mine[0][1]
        if theirs:
            raise RuntimeError("sharded proof stream: " + ", ".join("rank %d failed on job %d" % (q, k) for k, q in theirs[:8]))
        return proofs

    def _prove_stream_scatter(self, jobs, in_flight: int, on_device: bool, done_times: Optional[list]):
        """prove_stream in the "scatter" arrangement.  Two kinds of thread per rank: ONE communication thread issues every
        collective of the stream on a schedule that is the same on every rank - step t: the gather of job t - d's 384-byte records
        (d = in_flight), then the scatter of job t (on the job's source rank, k mod world with `rotate`, else rank 0, the witness map
        runs first); `in_flight` workers prove with the slices as they arrive (cg_prove_partial_q) and finish the proofs.  Job t's
        scatter reuses the receive buffer of job t - d, whose partial sums are done by then - the same event the gather of job t - d
        waits for, so the schedule costs no extra wait.  A job whose witness map or partial sums fail on some rank is poisoned
        for every rank through the gather, as in prove_stream; a collective that raises ends the stream on this rank."""
        n = len(jobs)
        in_flight = max(1, min(in_flight, n, self._stream_slots))
        d = in_flight
        world, rank, sc = self.world, self.rank, self._scatter
        gather = self._gather
        src_of = (lambda k: k % world) if self.rotate else (lambda k: 0)
        parts, gathered, proofs = [None] * n, [None] * n, [None] * n
        slices = [None] * n
        if done_times is not None:
            done_times[:] = [0.0] * n
        q_ready = [threading.Event() for _ in range(n)]
        have_part = [threading.Event() for _ in range(n)]
        have_all = [threading.Event() for _ in range(n)]
        nxt = [0]
        lock = threading.Lock()
        mine, theirs, fatal = [], [], []

        def comm():
            try:
                if not sc.on_host or (gather is not None and not gather.on_host):
                    torch.cuda.set_device(self.device)   # a new thread starts on device 0
                for t in range(n + d):
                    kg = t - d
                    if kg >= 0:                              # the records of job t - d: every rank's partial sums are awaited alike
                        have_part[kg].wait()
                        if fatal:
                            return
                        t0 = time.perf_counter()
                        if world > 1:
                            allp = gather(parts[kg])
                            self.all_gathers += 1
                        else:
                            allp = parts[kg]
                        bad = [q for q in range(world) if allp[PARTIAL_BYTES * q:PARTIAL_BYTES * (q + 1)] == POISON]
                        if bad:
                            theirs.extend((kg, q) for q in bad)
                        else:
                            gathered[kg] = allp
                        self.seconds["gather"] += time.perf_counter() - t0
                        have_all[kg].set()
                    if t < n:                                # the slices of job t (receive buffer t mod d: free since the wait above)
                        try:
                            slices[t] = sc.exchange(jobs[t][0], on_device, self.seconds, src_of(t), t % d)
                        except WitnessMapFailed as e:        # this rank's witness map failed; the scatter itself was done
                            mine.append((t, e.cause))
                            slices[t] = None
                        self.scatters += 1
                        q_ready[t].set()
            except BaseException as e:                       # noqa: BLE001 - a collective failed: nothing more can be exchanged
                fatal.append(e)
                for ev in q_ready + have_all + have_part:
                    ev.set()

        def worker():
            while not fatal:
                with lock:
                    k = nxt[0]
                    nxt[0] += 1
                if k >= n:
                    return
                a, r, s = jobs[k]
                q_ready[k].wait()
                if fatal:
                    return
                t0 = time.perf_counter()
                if slices[k] is None:
                    parts[k] = POISON
                else:
                    try:
                        q, q_dev = slices[k]
                        parts[k] = self.prover.prove_partial_q(a, q, r, on_device=on_device, q_on_device=q_dev)
                    except BaseException as e:   # noqa: BLE001
                        mine.append((k, e))
                        parts[k] = POISON
                t1 = time.perf_counter()
                have_part[k].set()
                have_all[k].wait()
                if fatal or gathered[k] is None:
                    continue
                try:
                    t2 = time.perf_counter()
                    proofs[k] = self.prover.assemble(gathered[k], world, r, s)
                    t3 = time.perf_counter()
                except BaseException as e:   # noqa: BLE001
                    mine.append((k, e))
                    continue
                if done_times is not None:
                    done_times[k] = t3
                with lock:
                    self.seconds["partial"] += t1 - t0
                    self.seconds["assemble"] += t3 - t2
                    self.proofs += 1

        ts = [threading.Thread(target=comm)] + [threading.Thread(target=worker) for _ in range(in_flight)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if fatal:
            raise fatal[0]
        if mine:
            raise mine[0][1]
        if theirs:
            raise RuntimeError("sharded proof stream: " + ", ".join("rank %d failed on job %d" % (q, k) for k, q in theirs[:8]))
        return proofs

    def breakdown_ms(self) -> dict:
        """mean milliseconds per proof spent in each step since construction (or the last reset_breakdown)"""
        n = max(1, self.proofs)
        return {k: round(v / n * 1e3, 3) for k, v in self.seconds.items()}

    def reset_breakdown(self) -> None:
        self.proofs = 0
        for k in self.seconds:
            self.seconds[k] = 0.0
