#!/usr/bin/env python3
"""Soak of the LONE slots: N bursts of one to four proofs arriving together on an otherwise idle sixteen-slot throughput context
(the first two of a burst take the lone slots, five streams each, on streams borrowed from the last one-stream slots; the
others take one-stream slots next to them), assignments alternately from pageable host memory and device memory, every proof
compared byte for byte with what a single-slot (latency) context made for the same (assignment, r, s).
usage: python tools/soak_bursts.py [bursts=1200]"""
import os, random, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
import crescent_credentials_amd as cc
from crescent_credentials_amd import workloads as wl

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
assert cc.lib().cg_init(0, None) == 0
R = cc.api.FR_MODULUS
l, m, M = wl.SHAPES["rs256-sd"]
cm, w = wl.synthetic_circuit(0xC5E5CE47 + 3, l, m, M, 0.9, 3, profile="gates")
rng = random.Random(77)
pk = cc.generate_parameters_with_qap(cm, *[rng.randrange(1, R) for _ in range(4)])
W = w.reshape(-1, 32)
ws = [w]
for seed in (1, 2):
    p = W.copy(); p[l:] = W[l:][np.random.default_rng(seed).permutation(M - l)]; ws.append(p.reshape(-1).copy())
wd = [torch.from_numpy(x).cuda() for x in ws]
ref = cc.Prover(pk, cm)
par = cc.Prover(pk, cm, proof_slots=16)
info = par.info()
assert info["lone_slots"] == 2, info
cases = [(j, rng.randrange(R), rng.randrange(R)) for j in range(len(ws)) for _ in range(4)] + [(0, 0, 5)]
want = {c: ref.prove_dev(wd[c[0]].data_ptr(), c[1], c[2]).data for c in cases}
for c in cases[:3]:
    par.prove_dev(wd[c[0]].data_ptr(), c[1], c[2])          # the one-time re-tune
bad = done = 0
t0 = time.time()
with ThreadPoolExecutor(max_workers=4) as ex:
    for b in range(N):
        k = 1 + (b % 4)
        pick = [cases[rng.randrange(len(cases))] for _ in range(k)]
        def one(c, host=(b & 1)):
            j, r, s = c
            got = par.prove(ws[j], r, s).data if host else par.prove_dev(wd[j].data_ptr(), r, s).data
            return got == want[c]
        res = list(ex.map(one, pick))
        bad += res.count(False)
        done += k
print("%d bursts, %d proofs in %.0f s on a context with 2 lone slots: %d mismatches" % (N, done, time.time() - t0, bad))
sys.exit(1 if bad else 0)
