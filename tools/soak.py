#!/usr/bin/env python3
"""Soak: N proofs of the full-size headline workload on a throughput context with T proof slots and T + 2 host threads, every
one compared byte for byte with the proof a single-slot (latency) context made for the same (assignment, r, s).  Odd
proofs take their assignment from page-locked HOST memory (cg_prove: the upload buffers and their copy streams under
load), even ones from device memory.  Catches rare order-dependence in the grouping / accumulation / upload under load.
usage: python tools/soak.py [N=4000] [T=16]"""
import os, random, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
import crescent_credentials_amd as cc
from crescent_credentials_amd import workloads as wl

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
assert cc.lib().cg_init(0, None) == 0
R = cc.api.FR_MODULUS
l, m, M = wl.SHAPES["rs256-sd"]
cm, w = wl.synthetic_circuit(0xC5E5CE47 + 3, l, m, M, 0.9, 3, profile="gates")
rng = random.Random(2024)
pk = cc.generate_parameters_with_qap(cm, *[rng.randrange(1, R) for _ in range(4)])
W = w.reshape(-1, 32)
ws = [w]
for seed in (1, 2, 3):
    p = W.copy(); p[l:] = W[l:][np.random.default_rng(seed).permutation(M - l)]; ws.append(p.reshape(-1).copy())
wd = [torch.from_numpy(x).cuda() for x in ws]
wh = [cc.HostBuffer(x.size) for x in ws]
for hb, x in zip(wh, ws):
    hb.array[:] = x
jobs = [(k % 4, rng.randrange(R) if k % 7 else 0, rng.randrange(R)) for k in range(32)]
alone = cc.Prover(pk, cm)
expect = [alone.prove_dev(wd[a].data_ptr(), r, s).data for a, r, s in jobs]
alone.close()
prover = cc.Prover(pk, cm, proof_slots=T)
order = [rng.randrange(32) for _ in range(N)]
t0 = time.time()
def one(ij):
    i, j = ij
    a, r, s = jobs[j]
    return (prover.prove_host_ptr(wh[a].ptr, r, s) if i & 1 else prover.prove_dev(wd[a].data_ptr(), r, s)).data
with ThreadPoolExecutor(max_workers=T + 2) as ex:
    got = list(ex.map(one, enumerate(order)))
dt = time.time() - t0
bad = sum(1 for j, g in zip(order, got) if g != expect[j])
prover.close()
print("soak: %d proofs (half from pinned host memory), %d slots, %d threads, %.1f s (%.1f proofs/s), mismatches: %d" % (N, T, T + 2, dt, N / dt, bad))
sys.exit(1 if bad else 0)
