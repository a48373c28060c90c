set -u
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "msm or prove" 2>&1 | tail -2)
for i in 1 2 3; do
python tools/probe_msm.py --group 1 --k 21 --reps 4 2>/dev/null | tail -1 | python -c "
import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('staged', 'sort_ms', round(d['sort_ms'],3), 'msm_h_ms', round(d['msm_h_ms'],3))"
CG_PLACE_DIRECT=1 python tools/probe_msm.py --group 1 --k 21 --reps 4 2>/dev/null | tail -1 | python -c "
import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('direct', 'sort_ms', round(d['sort_ms'],3), 'msm_h_ms', round(d['msm_h_ms'],3))"
done
for s in 1 8; do python tools/probe_latency.py $s 2>/dev/null | cut -c1-60; CG_PLACE_DIRECT=1 python tools/probe_latency.py $s 2>/dev/null | cut -c1-60; done
run() { env "$@" python bench.py --steps 100 --blocks 5 --no-sweep --no-cpu-baseline --no-host-witness 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['timing']['spread_pct'])"; }
for i in 1 2; do run X=1; run CG_PLACE_DIRECT=1; done
