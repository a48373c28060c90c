"""scaling of the C restatement's witness map and one G1 MSM with the thread count on this host"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import cpu_ref
from crescent_credentials_amd import workloads as wl
l, m, M = wl.SHAPES["rs256-sd"]
cm, w = wl.synthetic_circuit(3, l, m, M, 0.9, 3, profile="gates")
print("procs", cpu_ref.num_procs(), "affinity", len(os.sched_getaffinity(0)))
for nt in (8, 16, 32, 64, 128, 256):
    if nt > cpu_ref.num_procs():
        break
    t = time.perf_counter(); cpu_ref.witness_map((cm.a, cm.b, cm.c), l, m, M, w, nthreads=nt); d = time.perf_counter() - t
    print("witness_map threads %3d: %.2f s" % (nt, d), flush=True)
