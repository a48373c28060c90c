set -u
for q in 32 24 16; do echo "== GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q python tools/exp_shard_idle.py 2>&1 | grep "gap" | tail -6; done
for i in 1 2; do for q in 32 24 16 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 100 --blocks 5 --no-sweep --no-cpu-baseline --no-host-witness 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q', d['value'], d['timing']['spread_pct'])"
done; done
