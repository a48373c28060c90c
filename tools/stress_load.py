"""A shard context loaded over and over (cg_circuit_load: the fold of the h query and of C into the l query, the window tables)
and ONE record made on each load, compared point by point with the record of the first loads: a table that a load got wrong
shows as a wrong point for that context.  Run several of these at once, with tools/hold_queues.py beside them, to put the
process's hardware queues under time-slicing (what the eight-process bench test meets inside the full GPU suite).
usage: python tools/stress_load.py <shard_rank> <loads> [shape=medium] [shards=8]"""
import os
import random
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import crescent_credentials_amd as cc
from crescent_credentials_amd import workloads as wl

rank = int(sys.argv[1])
loads = int(sys.argv[2])
shape = sys.argv[3] if len(sys.argv) > 3 else "medium"
shards = int(sys.argv[4]) if len(sys.argv) > 4 else 8
cc.lib().cg_init(0, None)
R = cc.api.FR_MODULUS
l, m, M = wl.SHAPES[shape]
cm, w = wl.synthetic_circuit(11, l, m, M, 0.9, 3, profile="gates")
rng = random.Random(5)
pk = cc.generate_parameters_with_qap(cm, *[rng.randrange(1, R) for _ in range(4)])
wd = torch.from_numpy(w).cuda()
names, sizes = ("h", "l", "a", "b1", "b2"), (64, 64, 64, 64, 128)


external = os.environ.get("STRESS_EXTERNAL", "1") == "1"      # a shard that never runs the witness map (CG_FLAG_H_SCALARS_EXTERNAL)
proofs = int(os.environ.get("STRESS_PROOFS", "3"))            # per load: the first triggers the window re-tune, the others run on its tables
if external:
    plain = cc.Prover(pk, cm, shard_rank=rank, shard_count=shards)
    q_all = torch.empty(plain.domain_size * 32, dtype=torch.uint8, device="cuda")
    plain.witness_map_coset(wd.data_ptr(), on_device=True, out_dev=q_all.data_ptr())
    off, cnt = plain.h_scalars_slice(rank)
    q_ptr = q_all.data_ptr() + off * 32
    torch.cuda.synchronize()
    plain.close()


def one():
    ctx = cc.Prover(pk, cm, shard_rank=rank, shard_count=shards, h_scalars_external=external)
    recs = []
    for _ in range(proofs):
        if external:
            recs.append(ctx.prove_partial_q(wd.data_ptr(), q_ptr, 5, on_device=True, q_on_device=True))
        else:
            recs.append(ctx.prove_partial(wd.data_ptr(), 5, on_device=True))
    ctx.close()
    return b"".join(recs)


first = [one() for _ in range(3)]
ref = max(set(first), key=first.count)
bad = sum(1 for r in first if r != ref)
t0 = time.time()
for k in range(loads):
    rec = one()
    if rec != ref:
        bad += 1
        o, which = 0, []
        for j in range(len(rec) // 384):
            for nm, sz in zip(names, sizes):
                if rec[o:o + sz] != ref[o:o + sz]:
                    which.append("%s(proof %d)" % (nm, j))
                o += sz
        print("[shard %d] load %d: points %s differ" % (rank, k, ", ".join(which)), flush=True)
print("[shard %d] %d loads in %.0f s: %d wrong" % (rank, loads + 3, time.time() - t0, bad), flush=True)
