"""one 1/2-shard context: wall time of consecutive cg_prove_partial calls after an idle gap (clock ramp?), and with a
second, idle, 12-slot context alive in the process"""
import os, sys, time, random
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import crescent_credentials_amd as cc
from crescent_credentials_amd import workloads as wl
cc.lib().cg_init(0, None)
R = cc.api.FR_MODULUS
l, m, M = wl.SHAPES["rs256-sd"]
cm, w = wl.synthetic_circuit(3, l, m, M, 0.9, 3, profile="gates")
rng = random.Random(1)
pk = cc.generate_parameters_with_qap(cm, *[rng.randrange(1, R) for _ in range(4)])
p = cc.Prover(pk, cm, shard_rank=0, shard_count=2)
wd = torch.from_numpy(w).cuda()
f = lambda: p.prove_partial(wd.data_ptr(), 5, on_device=True)
for _ in range(4):
    f()
def series(tag, gap):
    time.sleep(gap)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); f(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    print(tag, "gap", gap, ts, "clock", round(cc.probe_shader_clock(-1, 2000), 3), flush=True)
for gap in (0.0, 0.05, 0.3, 1.0):
    series("alone", gap)
big = cc.Prover(pk, cm, proof_slots=12)
big.prove_dev(wd.data_ptr(), 1, 2); big.prove_dev(wd.data_ptr(), 1, 2)
for gap in (0.0, 0.3):
    series("with idle 12-slot context", gap)
big.close()
for gap in (0.0, 0.3):
    series("after closing the 12-slot context", gap)
