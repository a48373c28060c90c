set -u
for i in 1 2 3; do for s in 1 8; do echo -n "graph    "; python tools/probe_latency.py $s 2>/dev/null | cut -c1-60; echo -n "no graph "; CG_NO_GRAPH=1 python tools/probe_latency.py $s 2>/dev/null | cut -c1-60; done; done
run() { env "$@" python bench.py --steps 100 --blocks 5 --no-sweep --no-cpu-baseline --no-host-witness 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['timing']['spread_pct'], d['timing']['host_cpus_busy'])"; }
for i in 1 2; do run X=1; run CG_NO_GRAPH=1; done
