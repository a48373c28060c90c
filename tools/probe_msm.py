"""One MSM over resident bases, for kernel traces: python tools/probe_msm.py --k 10 --group 1 [--bits 0.9]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import crescent_credentials_amd as cc
from sweep import random_scalars
ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--group", type=int, default=1)
ap.add_argument("--bits", type=float, default=0.0)
ap.add_argument("--window", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
assert cc.lib().cg_init(0, None) == 0
n = 1 << a.k
seeds = np.random.default_rng(1).integers(0, 256, (n, 32), dtype=np.uint8)
seeds[:, 31] %= 0x30
seeds[:, 0] |= 1
bases = (cc.fixed_base_g1 if a.group == 1 else cc.fixed_base_g2)(seeds)
ctx = cc.MsmContext(bases, group=a.group, window_bits=a.window)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
sc = random_scalars(n, gen, a.bits)
torch.cuda.synchronize()
for _ in range(a.reps):
    out, tm = ctx.run_dev(sc.data_ptr(), n, timings=True)
print({k: v for k, v in tm.items() if v})
