set -u
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e_files.py -x -q -k "prove or witness or qap or folded or strided or setup" 2>&1 | tail -2)
for s in 1; do python tools/probe_latency.py $s 2>/dev/null | cut -c1-110; CG_SELL_PLAIN=1 python tools/probe_latency.py $s 2>/dev/null | cut -c1-110; done
run() { env "$@" python bench.py --steps 100 --blocks 5 --no-sweep --no-cpu-baseline --no-host-witness 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['timing']['spread_pct'], d['phase_ms']['witness_map_ms'])"; }
for i in 1 2 3; do run X=1; run CG_SELL_PLAIN=1; done
