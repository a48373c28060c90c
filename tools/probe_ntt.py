"""kernel time of resident transforms (cg_ntt_run, HIP events) over a few sizes: python tools/probe_ntt.py [logn ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crescent_credentials_amd as cc
cc.lib().cg_init(0, None)
out = []
for logn in [int(x) for x in sys.argv[1:]] or [18, 20, 21, 22]:
    ctx = cc.NttContext(logn)
    x = torch.randint(0, 256, (1 << logn, 32), dtype=torch.uint8, device="cuda")
    x[:, 31] %= 0x30
    torch.cuda.synchronize()
    for _ in range(3):
        ctx.run_dev(x.data_ptr(), inverse=False, coset=False)
    ms = min(ctx.run_dev(x.data_ptr(), inverse=False, coset=False) for _ in range(10))
    out.append("2^%d %.3f ms" % (logn, ms))
    ctx.close()
print("tile", os.environ.get("CG_NTT_TILE", "default"), " ".join(out))
