"""The two-halves arrangement of a sharded proof (cg_prove_partial_q_begin / cg_partial_witness_map_coset_half / two scatters /
cg_prove_partial_q_finish2) in a loop, N processes on the one GPU over gloo, EVERY piece checked on EVERY rank and proof against
this rank's own recomputation: the a-side and b-side slices that arrived (against cg_witness_map_coset_half on a plain shard
context), and the 384-byte record (against cg_prove_partial on that context), point by point.  Written to find the one-in-forty
"the two-halves scatter arrangement's proof differs" of the 8-process bench test; `--load` runs whole proofs on another context
of the same process from two threads meanwhile, `--host-load` spins CPU threads (both widen timing windows).
usage: python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 tools/stress_two_halves.py [--proofs 200]"""
import argparse
import os
import random
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

import crescent_credentials_amd as cc
from crescent_credentials_amd import workloads as wl
from crescent_credentials_amd.distributed import HScalarScatter, PartialGather


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--proofs", type=int, default=200)
    ap.add_argument("--shape", default="medium")
    ap.add_argument("--load", action="store_true")
    ap.add_argument("--host-load", type=int, default=0)
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cc.lib().cg_init(0, None)
    dev = torch.device("cuda:0")
    R = cc.api.FR_MODULUS
    l, m, M = wl.SHAPES[a.shape]
    cm, w = wl.synthetic_circuit(11, l, m, M, 0.9, 3, profile="gates")
    rng = random.Random(5)
    pk = cc.generate_parameters_with_qap(cm, *[rng.randrange(1, R) for _ in range(4)])
    ws = [torch.from_numpy(w).to(dev)]
    w2 = w.copy().reshape(-1, 32)
    w2[l + 1:] = w2[l + 1:][np.random.RandomState(3).permutation(len(w2) - l - 1)]
    ws.append(torch.from_numpy(w2.reshape(-1)).to(dev))
    ctx = cc.Prover(pk, cm, shard_rank=rank, shard_count=world, h_scalars_external=(rank > 1))
    plain = cc.Prover(pk, cm, shard_rank=rank, shard_count=world)
    sc_a = HScalarScatter(ctx, dev, None, rank, world, slots=1, src_ranks=(0,))
    sc_b = HScalarScatter(ctx, dev, None, rank, world, slots=1, src_ranks=(min(1, world - 1),))
    gather = PartialGather(dev, None) if world > 1 else None
    off, cnt = plain.h_scalars_slice(rank)
    stop = []
    if a.load:
        other = cc.Prover(pk, cm, proof_slots=2)
        def hammer():
            k = 0
            while not stop:
                other.prove_dev(ws[k % 2].data_ptr(), 1 + k, 2 + k)
                k += 1
        for _ in range(2):
            threading.Thread(target=hammer, daemon=True).start()
    for _ in range(a.host_load):
        def spin():
            x = 0
            while not stop:
                x += 1
        threading.Thread(target=spin, daemon=True).start()
    seconds = {"witness_map": 0.0, "scatter": 0.0}
    names = ("h", "l", "a", "b1", "b2")
    sizes = (64, 64, 64, 64, 128)
    bad = 0
    srng = random.Random(99)
    t0 = time.time()
    for k in range(a.proofs):
        wd = ws[k % 2]
        r = srng.randrange(R) if k % 7 else 0
        opened = ctx.prove_partial_q_begin(wd.data_ptr(), r, on_device=True)
        qa, _ = sc_a.exchange(wd.data_ptr(), True, seconds, 0, 0, wm=lambda **kw: opened.witness_map_coset_half(0, **kw))
        qb, _ = sc_b.exchange(wd.data_ptr(), True, seconds, min(1, world - 1), 0, wm=lambda **kw: opened.witness_map_coset_half(1, **kw))
        qa_c, qb_c = bytes(qa), bytes(qb)            # what arrived, before anything else touches the buffers
        part = opened.finish2(qa, qb, False)
        want_a = bytes(plain.witness_map_coset_half(wd.data_ptr(), 0, on_device=True))[off * 32:(off + cnt) * 32]
        want_b = bytes(plain.witness_map_coset_half(wd.data_ptr(), 1, on_device=True))[off * 32:(off + cnt) * 32]
        want = plain.prove_partial(wd.data_ptr(), r, on_device=True)
        msgs = []
        if qa_c != want_a:
            d = np.flatnonzero(np.frombuffer(qa_c, np.uint8) != np.frombuffer(want_a, np.uint8))
            msgs.append("a slice differs in %d bytes (first at element %d of %d)" % (d.size, d[0] // 32, cnt))
        if qb_c != want_b:
            d = np.flatnonzero(np.frombuffer(qb_c, np.uint8) != np.frombuffer(want_b, np.uint8))
            msgs.append("b slice differs in %d bytes (first at element %d of %d)" % (d.size, d[0] // 32, cnt))
        if bytes(qa) != qa_c or bytes(qb) != qb_c:
            msgs.append("a receive buffer changed during finish2")
        if part != want:
            o = 0
            for nm, sz in zip(names, sizes):
                if part[o:o + sz] != want[o:o + sz]:
                    msgs.append("partial %s differs" % nm)
                o += sz
        if msgs:
            bad += 1
            print("[rank %d] proof %d (r %s): %s" % (rank, k, "= 0" if r == 0 else "!= 0", "; ".join(msgs)), flush=True)
        if gather is not None:
            gather(part)
    stop.append(1)
    tot = torch.tensor([bad])
    dist.all_reduce(tot)
    if rank == 0:
        print("two halves, %d ranks, %d proofs in %.1f s: %d (rank, proof) pairs with a mismatch" % (world, a.proofs, time.time() - t0, int(tot[0])), flush=True)
    ctx.close(); plain.close()
    dist.barrier()
    sys.exit(1 if int(tot[0]) else 0)


if __name__ == "__main__":
    main()
