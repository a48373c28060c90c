#!/usr/bin/env python3
"""The shader clock the chip holds under different loads of this library (cg_probe_shader_clock from a side thread):
idle, a stream of whole proofs (the headline arrangement), G1 MSMs only, transforms only.  MI355X is power-limited under dense
integer multiply-adds: the denser the vector-ALU work, the lower the clock (tools/ubench/valu_rates --json: 1.34 GHz under
nothing but v_mad_u64_u32).  usage: python tools/probe_clock_vs_load.py [seconds per load]"""
import os
import random
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import crescent_credentials_amd as cc
from crescent_credentials_amd import workloads as wl

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
cc.lib().cg_init(0, None)
R = cc.api.FR_MODULUS


def sample_while(fn_threads, secs):
    """run the callables in threads for `secs`, sampling the clock meanwhile -> (median GHz, calls per second)"""
    stop = threading.Event()
    counts = [0] * len(fn_threads)

    def loop(i, fn):
        while not stop.is_set():
            fn()
            counts[i] += 1
    ts = [threading.Thread(target=loop, args=(i, f)) for i, f in enumerate(fn_threads)]
    for t in ts:
        t.start()
    time.sleep(0.5)
    c0, t0 = sum(counts), time.perf_counter()
    clocks = []
    while time.perf_counter() - t0 < secs:
        clocks.append(cc.probe_shader_clock(0, 20000))
        time.sleep(0.03)
    rate = (sum(counts) - c0) / (time.perf_counter() - t0)
    stop.set()
    for t in ts:
        t.join()
    clocks.sort()
    return round(clocks[len(clocks) // 2], 3), round(rate, 1)


print("idle:", sample_while([lambda: time.sleep(0.01)], 1.0)[0], "GHz")
l, m, M = wl.SHAPES["rs256-sd"]
cm, w = wl.synthetic_circuit(0xC5E5CE47 + 3, l, m, M, 0.9, 3, profile="gates")
rng = random.Random(1)
pk = cc.generate_parameters_with_qap(cm, *[rng.randrange(1, R) for _ in range(4)])
prover = cc.Prover(pk, cm, proof_slots=16)
wd = torch.from_numpy(w).cuda()
prover.prove_dev(wd.data_ptr(), 5, 7)
prover.prove_dev(wd.data_ptr(), 5, 7)
ghz, rate = sample_while([lambda: prover.prove_dev(wd.data_ptr(), 5, 7)] * 16, SECS)
print("whole proofs, 16 in flight: %.3f GHz at %.1f proofs/s" % (ghz, rate))
prover.close()
# G1 MSMs only: the h query's size, uniform scalars, four resident table sets proving in turn
n = 1 << 21
scal = torch.from_numpy(np.frombuffer(b"".join(rng.randrange(R).to_bytes(32, "little") for _ in range(4096)), np.uint8).copy()).cuda().repeat(n // 4096)
bases = pk.h_query[:64 * (n - 1)]
ctxs = [cc.MsmContext(bases, group=1) for _ in range(4)]
for c in ctxs:
    c.run_dev(scal.data_ptr(), n - 1)
ghz, rate = sample_while([(lambda c=c: c.run_dev(scal.data_ptr(), n - 1)) for c in ctxs], SECS)
print("G1 MSMs of 2^21 uniform scalars, 4 in flight: %.3f GHz at %.1f MSMs/s" % (ghz, rate))
for c in ctxs:
    c.close()
nts = [cc.NttContext(21) for _ in range(4)]
bufs = [torch.zeros(32 << 21, dtype=torch.uint8, device="cuda") for _ in nts]
ghz, rate = sample_while([(lambda t=t, b=b: t.run_dev(b.data_ptr())) for t, b in zip(nts, bufs)], SECS)
print("transforms of 2^21 elements, 4 in flight: %.3f GHz at %.1f transforms/s" % (ghz, rate))
