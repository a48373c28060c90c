set -u
run() { env "$@" python bench.py --steps 100 --blocks 5 --no-sweep --no-cpu-baseline --no-host-witness $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$* $EXTRA', d['value'], d['timing']['spread_pct'])"; }
for i in 1 2; do
EXTRA="" run GPU_MAX_HW_QUEUES=32
EXTRA="" run GPU_MAX_HW_QUEUES=16 CG_SERIAL_STREAMS=1
EXTRA="--inflight 16" run GPU_MAX_HW_QUEUES=16 CG_SERIAL_STREAMS=1
EXTRA="--inflight 16" run GPU_MAX_HW_QUEUES=24 CG_SERIAL_STREAMS=1
EXTRA="--inflight 16" run GPU_MAX_HW_QUEUES=32
done
