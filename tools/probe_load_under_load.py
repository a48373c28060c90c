import os, sys, time, random, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import crescent_credentials_amd as cc
from crescent_credentials_amd import workloads as wl
cc.lib().cg_init(0, None)
R = cc.api.FR_MODULUS
l, m, M = wl.SHAPES["rs256-sd"]
cm, w = wl.synthetic_circuit(3, l, m, M, 0.9, 3, profile="gates")
rng = random.Random(1)
pk = cc.generate_parameters_with_qap(cm, *[rng.randrange(1, R) for _ in range(4)])
busy = cc.Prover(pk, cm, proof_slots=16)
wd = torch.from_numpy(w).cuda()
busy.prove_dev(wd.data_ptr(), 5, 7); busy.prove_dev(wd.data_ptr(), 5, 7)
p = cc.Prover(pk, cm, proof_slots=16); print("idle GPU: load timings", p.load_timings()); p.close()
stop = threading.Event()
def hammer():
    while not stop.is_set():
        busy.prove_dev(wd.data_ptr(), 5, 7)
ts = [threading.Thread(target=hammer) for _ in range(16)]
for t in ts: t.start()
time.sleep(1.0)
p = cc.Prover(pk, cm, proof_slots=16); print("busy GPU (16 proofs in flight on another context): load timings", p.load_timings()); p.close()
stop.set()
for t in ts: t.join()
