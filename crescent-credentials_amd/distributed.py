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


def gather_partials(partial: bytes, device, group=None) -> bytes:
    """all_gather of one rank's 384-byte partial-sum record -> world x 384 bytes, rank order."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":
        device = torch.device("cpu")
    mine = torch.frombuffer(bytearray(partial), dtype=torch.uint8).to(device)
    out = [torch.empty(PARTIAL_BYTES, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    return b"".join(bytes(t.cpu().numpy().tobytes()) for t in out)


class ShardedProver:
    """One proof across all ranks.  `prover` is any object with prove_partial(assignment, r, on_device) and
    assemble(partials, n_shards, r, s) — a crescent_credentials_amd.Prover loaded with shard_rank/shard_count
    on the GPU, or a stand-in in the CPU (gloo) tests."""

    def __init__(self, prover, device, group=None):
        self.prover = prover
        self.device = device
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.all_gathers = 0          # collectives issued so far (one per proof when world > 1)

    def _finish(self, part: bytes, r: int, s: int):
        if self.world > 1:
            parts = gather_partials(part, self.device, self.group)
            self.all_gathers += 1
        else:
            parts = part
        return self.prover.assemble(parts, self.world, r, s)

    def prove(self, full_assignment, r: int, s: int):
        return self._finish(self.prover.prove_partial(full_assignment, r, on_device=False), r, s)

    def prove_dev(self, d_ptr: int, r: int, s: int):
        return self._finish(self.prover.prove_partial(d_ptr, r, on_device=True), r, s)
