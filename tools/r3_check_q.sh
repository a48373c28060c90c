set -u
for i in 1 2 3 4 5; do (timeout 900 python -m pytest tests/test_gpu_host_and_ranks.py tests/test_gpu_e2e_files.py -x -q -k "come_and_go or cache" 2>&1 | tail -1); done
for i in 1 2; do for s in 1 8; do python tools/probe_latency.py $s 2>/dev/null | cut -c1-60; done; done
python bench.py --steps 100 --blocks 5 --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['timing']['spread_pct'], d['timing']['host_cpus_busy'], d['host_witness']['pinned']['proofs_per_s'], d['host_witness']['pinned']['host_cpus_busy'])"
