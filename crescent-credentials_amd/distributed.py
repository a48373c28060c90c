"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL ("nccl").

The prove path shards in two ways:
  * independent proofs (the reference's own concurrency model: one credential per task,
    sample/client_helper/src/main.rs:177-216) -> every rank holds a full replica, no collective;
  * one proof, every MSM range-sharded (the five independent sums of forks/groth16/src/prover.rs:66,74,266):
    each rank computes partial sums over its contiguous range of every query, and the only exchange
    step is an all_gather of 384 bytes per rank (four G1 points + one G2 point), after which every rank
    (or just rank 0) finishes A, B, C exactly as prover.rs:76-135.
"""
from __future__ import annotations

import time
from typing import Tuple

import numpy as np
import torch
import torch.distributed as dist

PARTIAL_BYTES = 384


def shard_range(n: int, rank: int, count: int) -> Tuple[int, int]:
    """Contiguous range of a length-n query owned by `rank` (same formula as csrc/prover.hip shard_range)."""
    if count <= 1:
        return 0, n
    return n * rank // count, n * (rank + 1) // count


def barrier_sync(world: int) -> None:
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(x: float, world: int, device) -> float:
    if world <= 1:
        return x
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class PartialGather:
    """The one exchange step of a sharded proof: an all_gather of 384 bytes per rank (four G1 partial sums + one G2).
    The tensors are allocated once and reused for every proof; with RCCL ("nccl") they live on the device and the
    collective runs over xGMI, with gloo they are host tensors."""

    def __init__(self, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.on_host = dist.get_backend(group) == "gloo"
        dev = torch.device("cpu") if self.on_host else device
        self._mine = torch.empty(PARTIAL_BYTES, dtype=torch.uint8, device=dev)
        self._all = torch.empty(self.world * PARTIAL_BYTES, dtype=torch.uint8, device=dev)
        if self.on_host:
            self._stage_in = self._mine
            self._stage_out = self._all
        else:   # page-locked staging on the host side of the two 384-byte copies
            self._stage_in = torch.empty(PARTIAL_BYTES, dtype=torch.uint8).pin_memory()
            self._stage_out = torch.empty(self.world * PARTIAL_BYTES, dtype=torch.uint8).pin_memory()

    def __call__(self, partial: bytes) -> bytes:
        self._stage_in.numpy()[:] = np.frombuffer(partial, dtype=np.uint8)
        if not self.on_host:
            self._mine.copy_(self._stage_in, non_blocking=True)
        dist.all_gather_into_tensor(self._all, self._mine, group=self.group)
        if not self.on_host:
            self._stage_out.copy_(self._all, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        return self._stage_out.numpy().tobytes()


def gather_partials(partial: bytes, device, group=None) -> bytes:
    """all_gather of one rank's 384-byte partial-sum record -> world x 384 bytes, rank order (one-off form of
    PartialGather)."""
    return PartialGather(device, group)(partial)


class ShardedProver:
    """One proof across all ranks.  `prover` is any object with prove_partial(assignment, r, on_device) and
    assemble(partials, n_shards, r, s) — a crescent_credentials_amd.Prover loaded with shard_rank/shard_count
    on the GPU, or a stand-in in the CPU (gloo) tests.  `seconds` accumulates where the wall time of the proofs went
    (this rank's partial sums, the all_gather, the host finish)."""

    def __init__(self, prover, device, group=None):
        self.prover = prover
        self.device = device
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.all_gathers = 0          # collectives issued so far (one per proof when world > 1)
        self.proofs = 0
        self.seconds = {"partial": 0.0, "gather": 0.0, "assemble": 0.0}
        self._gather = PartialGather(device, group) if self.world > 1 else None

    def _prove(self, assignment, on_device: bool, r: int, s: int):
        t0 = time.perf_counter()
        part = self.prover.prove_partial(assignment, r, on_device=on_device)
        t1 = time.perf_counter()
        if self.world > 1:
            parts = self._gather(part)
            self.all_gathers += 1
        else:
            parts = part
        t2 = time.perf_counter()
        proof = self.prover.assemble(parts, self.world, r, s)
        t3 = time.perf_counter()
        self.seconds["partial"] += t1 - t0
        self.seconds["gather"] += t2 - t1
        self.seconds["assemble"] += t3 - t2
        self.proofs += 1
        return proof

    def prove(self, full_assignment, r: int, s: int):
        return self._prove(full_assignment, False, r, s)

    def prove_dev(self, d_ptr: int, r: int, s: int):
        return self._prove(d_ptr, True, r, s)

    def breakdown_ms(self) -> dict:
        """mean milliseconds per proof spent in each step since construction (or the last reset_breakdown)"""
        n = max(1, self.proofs)
        return {k: round(v / n * 1e3, 3) for k, v in self.seconds.items()}

    def reset_breakdown(self) -> None:
        self.proofs = 0
        for k in self.seconds:
            self.seconds[k] = 0.0
