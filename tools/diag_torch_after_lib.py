import sys, os, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import crescent_credentials_amd as cc
print("init", cc.lib().cg_init(0, None))
maps = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l]
print("loaded before torch:", sorted(set(maps)))
step = sys.argv[1] if len(sys.argv) > 1 else "ntt"
if step == "ntt":
    ctx = cc.NttContext(10); ctx.run(np.zeros(32 << 10, np.uint8)); print("ntt ok")
import torch
print("torch", torch.__version__, torch.version.hip)
maps = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l]
print("loaded after torch:", sorted(set(maps)))
try:
    print("device_count", torch.cuda.device_count())
    x = torch.zeros(4).cuda(); print("cuda ok", x.sum().item())
except Exception as e:
    print("FAIL", repr(e)[:300])
