"""Quick timing probe: synthetic circuit -> GPU setup -> load -> a few proofs with phase timings."""
import argparse, time, random, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crescent_credentials_amd as cc
from crescent_credentials_amd import workloads as wl

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="rs256-sd")
ap.add_argument("--bits", type=float, default=0.9)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--window", type=int, default=0)
a = ap.parse_args()
l, m, M = wl.SHAPES[a.shape]
t = time.time(); cm, w = wl.synthetic_circuit(1, l, m, M, a.bits, 3); print("synth %.1fs nnz=%d" % (time.time() - t, cm.a.nnz + cm.b.nnz + cm.c.nnz), flush=True)
assert cc.lib().cg_init(0, None) == 0
rng = random.Random(1)
R = cc.api.FR_MODULUS
t = time.time(); pk = cc.generate_parameters_with_qap(cm, *(rng.randrange(1, R) for _ in range(4))); print("setup %.1fs" % (time.time() - t), flush=True)
t = time.time(); p = cc.Prover(pk, cm, window_bits=a.window); print("load %.1fs D=%d" % (time.time() - t, p.domain_size), flush=True)
for i in range(a.iters):
    t = time.time(); proof, tm = p.prove(w, rng.randrange(R), rng.randrange(R), timings=True); dt = time.time() - t
    print("prove %.1f ms" % (dt * 1e3), {k: (round(v, 2) if isinstance(v, float) else v) for k, v in tm.items()}, flush=True)
print(proof.data.hex()[:32])
