#!/usr/bin/env python3
"""Full-size (S21) byte parity of the HIP path against the C restatement on several assignments and (r, s):
satisfying circom-like, satisfying uniform, random (non-satisfying) and sparse assignments.  ~4 s of CPU per case."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import crescent_credentials_amd as cc
from crescent_credentials_amd import workloads as wl
import cpu_ref

assert cc.lib().cg_init(0, None) == 0
R = cc.api.FR_MODULUS
l, m, M = wl.SHAPES["rs256-sd"]
rng = random.Random(2026)
nprng = np.random.default_rng(2026)
trap = [rng.randrange(1, R) for _ in range(4)]
results = []
for name, bits in (("circom-like", 0.9), ("uniform", 0.0)):
    cm, w = wl.synthetic_circuit(0xC5E5CE47 + (3 if bits else 4), l, m, M, bits, 3)
    pk = cc.generate_parameters_with_qap(cm, *trap)
    prover = cc.Prover(pk, cm, proof_slots=1)
    cases = [(name + ", satisfying", w)]
    if bits:
        a = nprng.integers(0, 256, (M, 32), dtype=np.uint8); a[:, 31] %= 0x30; a[0] = 0; a[0, 0] = 1
        cases.append((name + " key, random assignment (not satisfying)", a.reshape(-1).copy()))
        b = a.copy(); b[nprng.random(M) < 0.97] = 0; b[0] = 0; b[0, 0] = 1
        cases.append((name + " key, sparse assignment", b.reshape(-1).copy()))
    for label, wv in cases:
        for r, s in ((0, 0), (rng.randrange(R), rng.randrange(R))):
            t = time.time()
            g = prover.prove(wv, r, s).data
            tg = time.time() - t
            t = time.time()
            c = cpu_ref.prove(pk, (cm.a, cm.b, cm.c), l, m, M, wv, r, s, nthreads=32)
            tc = time.time() - t
            ok = g == c
            results.append(ok)
            print("%-58s r,s %s: GPU %.3f s  CPU %.2f s  bytes identical: %s" % (label, "zero" if r == 0 else "random", tg, tc, ok), flush=True)
    prover.close()
print("ALL IDENTICAL" if all(results) else "MISMATCH")
sys.exit(0 if all(results) else 1)
