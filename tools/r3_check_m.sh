set -u
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ntt or prove or strided" 2>&1 | tail -2)
for i in 1 2; do
for t in 11 10; do
CG_NTT_TILE=$t python - <<PY
import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
import crescent_credentials_amd as cc
cc.lib().cg_init(0, None)
out = []
for logn in (18, 20, 21, 22):
    ctx = cc.NttContext(logn)
    x = torch.randint(0, 256, (1 << logn, 32), dtype=torch.uint8, device="cuda"); x[:, 31] %= 0x30
    for _ in range(3): ctx.run_dev(x.data_ptr(), inverse=False, coset=False)
    ms = min(ctx.run_dev(x.data_ptr(), inverse=False, coset=False) for _ in range(8))
    out.append("2^%d %.3f ms" % (logn, ms))
print("tile", os.environ["CG_NTT_TILE"], " ".join(out))
PY
done; done
for s in 1 8; do for t in 11 10; do echo -n "tile $t "; CG_NTT_TILE=$t python tools/probe_latency.py $s 2>/dev/null | cut -c1-100; done; done
run() { env "$@" python bench.py --steps 100 --blocks 5 --no-sweep --no-cpu-baseline --no-host-witness 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['timing']['spread_pct'])"; }
for i in 1 2; do run CG_NTT_TILE=11; run CG_NTT_TILE=10; done
